// explicit instantiations, group Q: precision-mode (fp64) general-row kernels, 24- and 32-wide (see myosim_inst_list.hpp)
#include "myosim_engine_kernel_f64.hpp"
#include "myosim_inst_list.hpp"
namespace mm64 {
MM_KERNELS_F64_Q(MM_INSTANTIATE)
}
