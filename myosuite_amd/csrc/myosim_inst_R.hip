// explicit instantiations, group R: precision-mode (fp64) general-row kernels of the legs (36-wide; Euler and implicitfast)
#include "myosim_engine_kernel_f64.hpp"
#include "myosim_inst_list.hpp"
namespace mm64 {
MM_KERNELS_F64_R(MM_INSTANTIATE)
}
