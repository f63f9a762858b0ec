// explicit instantiations, group K: the reset-observation kernels (see myosim_inst_list.hpp)
#include "myosim_engine_kernel.hpp"
#include "myosim_inst_list.hpp"
MM_KERNELS_OBS(MM_INSTANTIATE_OBS)
