// myosim_inst_list.hpp -- the ONE list of compiled k_engine<G, NVP, LM, GEN, INTEG> instantiations (both LM variants each).
// X(lanes_per_env, padded_nv, general_rows, integrator kernel: 0 Euler, 1 RK4, 2 implicitfast).  Used for: explicit instantiation (myosim_inst_*.hip, one group per
// translation unit so they build in parallel), extern declarations + have_kernel() + the launch table (myosim_engine.hip).
#pragma once
#define MM_KERNELS_A(X) X(4, 4, 0, 0) X(8, 4, 0, 0) X(16, 4, 0, 0) X(32, 4, 0, 0) X(64, 4, 0, 0)
#define MM_KERNELS_B(X) X(32, 24, 0, 0) X(64, 24, 0, 0) X(32, 32, 0, 0)
#define MM_KERNELS_C(X) X(64, 32, 0, 0) X(64, 40, 0, 0) X(16, 4, 1, 0)
#define MM_KERNELS_D(X) X(32, 24, 1, 0) X(64, 32, 1, 0) X(32, 32, 1, 0)
#define MM_KERNELS_E(X) X(64, 40, 1, 0) X(4, 4, 0, 1) X(32, 24, 0, 1)
#define MM_KERNELS_F(X) X(32, 32, 0, 1) X(64, 40, 0, 1) X(16, 4, 1, 1) X(32, 24, 1, 1)
#define MM_KERNELS_G(X) X(64, 32, 1, 1) X(64, 40, 1, 1)
#define MM_KERNELS_H(X) X(64, 36, 1, 0)   /* leg models: 34 dofs (the 40-wide tile wastes 20 % of the dense linear algebra) */
#define MM_KERNELS_I(X) X(64, 24, 1, 0)   /* row-rich models with <= 24 dofs (key turn, torso) */
#define MM_KERNELS_J(X) X(4, 4, 0, 2) X(32, 24, 0, 2) X(64, 36, 1, 2)   /* implicitfast: elbow, hand, leg at their default widths */
/* reset-observation pass as its own kernel (k_engine<..., OBS = true>, model through L2): the BASELINE workloads whose reset is not
   folded into the env-step launch -- reorient, leg-walk (Euler and implicitfast) */
#define MM_KERNELS_OBS(X) X(64, 32, 1, 0) X(64, 36, 1, 0) X(64, 36, 1, 2)
/* precision mode (real = double, namespace mm64; myosim_engine_kernel_f64.hpp): the limit-rows-only Euler kernels of the elbow-
   and hand-sized models -- BASELINE.json's accuracy target is stated on configs 2-3 (group P) -- and the general-row kernels of the
   contact / equality models, one env per wave: self-colliding hand / key turn / torso (24), reorient family (32), legs (36; Euler
   and implicitfast) (group Q) */
#define MM_KERNELS_F64_P(X) X(4, 4, 0, 0) X(8, 4, 0, 0) X(16, 4, 0, 0) X(32, 24, 0, 0) X(64, 24, 0, 0)
#define MM_KERNELS_F64_Q(X) X(64, 24, 1, 0) X(64, 32, 1, 0)
#define MM_KERNELS_F64_R(X) X(64, 36, 1, 0) X(64, 36, 1, 2)
#define MM_KERNELS_F64(X) MM_KERNELS_F64_P(X) MM_KERNELS_F64_Q(X) MM_KERNELS_F64_R(X)
#define MM_KERNEL_LIST(X) MM_KERNELS_J(X) MM_KERNELS_A(X) MM_KERNELS_B(X) MM_KERNELS_C(X) MM_KERNELS_D(X) MM_KERNELS_E(X) MM_KERNELS_F(X) MM_KERNELS_G(X) MM_KERNELS_H(X) MM_KERNELS_I(X)
#define MM_INSTANTIATE(G_, N_, GN_, RK_)                                        \
  template __global__ void k_engine<G_, N_, true, GN_ != 0, RK_>(KArgs);   \
  template __global__ void k_engine<G_, N_, false, GN_ != 0, RK_>(KArgs);
#define MM_DECLARE(G_, N_, GN_, RK_)                                                   \
  extern template __global__ void k_engine<G_, N_, true, GN_ != 0, RK_>(KArgs);   \
  extern template __global__ void k_engine<G_, N_, false, GN_ != 0, RK_>(KArgs);
#define MM_INSTANTIATE_OBS(G_, N_, GN_, RK_) template __global__ void k_engine<G_, N_, false, GN_ != 0, RK_, true>(KArgs);
#define MM_DECLARE_OBS(G_, N_, GN_, RK_) extern template __global__ void k_engine<G_, N_, false, GN_ != 0, RK_, true>(KArgs);
