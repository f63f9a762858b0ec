// explicit instantiations, group P: the precision-mode (fp64) kernels (see myosim_inst_list.hpp, myosim_engine_kernel_f64.hpp)
#include "myosim_engine_kernel_f64.hpp"
#include "myosim_inst_list.hpp"
namespace mm64 {
MM_KERNELS_F64_P(MM_INSTANTIATE)
}
