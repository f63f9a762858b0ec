// explicit instantiations, group J (see myosim_inst_list.hpp)
#include "myosim_engine_kernel.hpp"
#include "myosim_inst_list.hpp"
MM_KERNELS_J(MM_INSTANTIATE)
