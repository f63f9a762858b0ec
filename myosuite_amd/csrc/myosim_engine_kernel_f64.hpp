#pragma once
// myosim_engine_kernel_f64.hpp -- the precision-mode kernels: the same device code over real = double, in namespace mm64
// (`mm_model_set_option(m, "precision", MM_PREC_F64 | MM_PREC_F64_STATE)`, include/myosim.h).  Limit-rows-only models on the
// Euler integrator (the general-row / collision code is fp32-only and compiled out: MM_F64).  One wave per SIMD: the register
// state of a lane doubles, so these kernels are built for 256-thread blocks (up to 512 VGPRs + AGPRs per lane).
#include "myosim_engine_common.hpp"
namespace mm64 {
#undef MM_REAL
#undef MM_F64
#define MM_REAL double
#define MM_F64 1
// The loaders are templates over model tables (Tab<float>, a global type): argument-dependent lookup would also find the fp32
// body's ::ld3 / ::ldq / ::ldm from inside mm64 in a translation unit that holds both families (the host side), so the fp64
// copies get their own names.
#define ld3 ld3_f64
#define ldq ldq_f64
#define ldm ldm_f64
#include "myosim_engine_body.inc"
#undef ld3
#undef ldq
#undef ldm
}   // namespace mm64
#undef MM_REAL
#undef MM_F64
#define MM_REAL float
#define MM_F64 0
