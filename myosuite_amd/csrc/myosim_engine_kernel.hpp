#pragma once
// myosim_engine_kernel.hpp -- the fused engine kernel in fp32 (the shipped kernels): myosim_engine_body.inc with real = float at
// global scope.  Header comment, kernel-argument structs and model-table accessors: myosim_engine_common.hpp.
#include "myosim_engine_common.hpp"
#undef MM_REAL
#undef MM_F64
#define MM_REAL float
#define MM_F64 0
#include "myosim_engine_body.inc"
