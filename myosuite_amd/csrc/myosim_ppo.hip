// myosim_ppo.hip -- fused PPO learner kernels (include/myosim_ppo.h): the consumer of the batched env-step on the training path
// of the reference (benchmarks/mjx_benchmark_PPO.py:50-60, hyper-parameters myosuite/envs/myo/mjx/__init__.py:43-67).
//
// A PPO minibatch update on these networks ((64, 64, 64) MLPs, 1 280 ... 5 120 samples) is ~0.3 GFLOP: as ~100 torch launches it
// is pure launch latency (round 4, HIP-graphed: ~0.68 ms per update, 256 updates per 81 920-step iteration against 4 ms of
// rollout).  Here a workgroup of four waves owns S = 16 or 32 samples end to end: the gathered, normalised observation rows and
// every layer's pre-activations stay in LDS, each linear layer -- forward, backward-data and backward-weight -- is a sweep of
// v_mfma_f32_16x16x4_f32 tiles (fp32 in, fp32 accumulate: the same numbers as the torch fp32 path up to summation order) whose
// A operand comes from LDS as 128-bit reads and whose B operand is the weight matrix read through L2 (20 k ... 45 k floats: it
// never leaves the cache), the losses are evaluated in place on the output tile, and the workgroup leaves ONE partial gradient
// per parameter.  A second launch adds the partials in a fixed order (deterministic; no float atomics) and a third clips and
// applies Adam.  Policy and value networks run in different workgroups of the same launch.
//
// Tile bookkeeping (lane = 16 lk + lr of a wave): v_mfma_f32_16x16x4_f32 takes A[row = lr][k = lk], B[k = lk][col = lr] and
// returns D[row = 4 lk + v][col = lr] in v = 0..3.  Within a 16-wide k chunk lane group lk supplies k = 4 lk + m for the m-th of
// four back-to-back MFMAs (a permutation of the reduction index, applied to both operands), so that one 128-bit load feeds four
// MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/myosim.h"
#include "../../include/myosim_ppo.h"

namespace {

thread_local std::string g_perr;
int pfail(int code, const std::string& msg) { g_perr = msg; return code; }
#define PHIPCHK(x)                                                                                  \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) return pfail(MM_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));    \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int NTHREADS = 256;
constexpr float HALF_LOG_2PI = 0.91893853320467274178f;

// one network as the kernels see it: dimensions, parameter offsets (floats, relative to the network's first parameter) and the
// LDS plan of a workgroup of S samples (offsets in floats; every row stride is 4 mod 8 floats: 16 rows x 128-bit reads of an MFMA
// operand fall into distinct banks)
struct NetD {
  int nl, in, npar, pad_;
  int w[MM_PPO_MAX_LAYERS], woff[MM_PPO_MAX_LAYERS], boff[MM_PPO_MAX_LAYERS];
  int ldx, ldh, ldo, total;
  int ldz[MM_PPO_MAX_LAYERS], oZ[MM_PPO_MAX_LAYERS], oA[MM_PPO_MAX_LAYERS];   // hidden layer l: pre-activations Z, activations A = swish(Z)
  int oX, oOUT, oD0, oD1, oAUX, pad2_;          // oAUX: S x act_dim raw actions + S x act_dim entropy-sample noise of the policy's loss stage
};

#ifndef MM_PPO_PROF
#define MM_PPO_PROF 0        /* 1 (tools/build_variant.py ppoprof -DMM_PPO_PROF=1): k_ppo_grad stamps clock64() per stage into mm_ppo_debug_set_prof's buffer */
#endif
__device__ unsigned long long* g_ppo_prof = nullptr;
#define PPROF(i)                                                                                                       \
  do {                                                                                                                 \
    if (MM_PPO_PROF && threadIdx.x == 0 && g_ppo_prof) g_ppo_prof[(size_t)blockIdx.x * 64 + (i)] = clock64();         \
  } while (0)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_(float z) { return z * sigmoidf_(z); }
__device__ __forceinline__ float dsilu_(float z) { const float s = sigmoidf_(z); return s * (1.f + z * (1.f - s)); }
__device__ __forceinline__ float softplus_(float x) { return x > 20.f ? x : log1pf(expf(x)); }      // torch F.softplus (threshold 20)
__device__ __forceinline__ float dsoftplus_(float x) { return x > 20.f ? 1.f : sigmoidf_(x); }

// C[s][n] = bias[n] + sum_k A[s][k] W[n][k]: A in LDS (S = 16 RT rows, stride lda, columns K..K16 zero), W global row-major
// [N][K].  A wave owns output column tiles nt = wave, wave + 4, ... and all RT row tiles of each (the weight operand is loaded
// once per k chunk).  epi(s, n, n < N, value) for every s < S, n < N16.
template <int RT, class Epi>
__device__ __forceinline__ void gemm_F(const float* A, int lda, int K, const float* __restrict__ Wg, const float* __restrict__ bg, int N, Epi epi) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int K16 = (K + 15) & ~15, NT = (N + 15) >> 4;
  const bool vec = (K & 3) == 0;
  constexpr int PF = 8;
  for (int nt = wv; nt < NT; nt += 4) {
    f4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) acc[rt] = f4{0.f, 0.f, 0.f, 0.f};
    const int n = 16 * nt + lr;
    const bool nok = n < N;
    const float* wr = Wg + (size_t)(nok ? n : 0) * K;
    // the weight operand of PF k chunks is requested before the first MFMA of the batch: one L2 round trip per batch, not per chunk
    // (a wave has its SIMD to itself here -- nothing else hides the latency)
    for (int c0 = 0; c0 < K16; c0 += 16 * PF) {
      f4 b[PF];
#pragma unroll
      for (int u = 0; u < PF; u++) {
        const int k0 = c0 + 16 * u + 4 * lk;
        if (vec) {
          b[u] = (nok && k0 < K) ? (f4)(*(const f4u*)(wr + k0)) : f4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
          for (int m = 0; m < 4; m++) b[u][m] = (nok && k0 + m < K) ? wr[k0 + m] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < PF; u++) {
        if (c0 + 16 * u < K16) {
          const int k0 = c0 + 16 * u + 4 * lk;
#pragma unroll
          for (int rt = 0; rt < RT; rt++) {
            const f4 av = *(const f4*)(A + (16 * rt + lr) * lda + k0);
#pragma unroll
            for (int m = 0; m < 4; m++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b[u][m], acc[rt], 0, 0, 0);
          }
        }
      }
    }
    const float bias = nok ? bg[n] : 0.f;
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int v = 0; v < 4; v++) epi(16 * rt + 4 * lk + v, n, nok, acc[rt][v] + bias);
  }
}

// C[s][k] = sum_n dZ[s][n] W[n][k]: dZ in LDS (columns N..N16 zero), W global [N][K].  epi(s, k, k < K, value), k < K16.
template <int RT, class Epi>
__device__ __forceinline__ void gemm_B(const float* dZ, int ldz, int N, const float* __restrict__ Wg, int K, Epi epi) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int N16 = (N + 15) & ~15, KT = (K + 15) >> 4;
  constexpr int PF = 4;
  for (int kt = wv; kt < KT; kt += 4) {
    f4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) acc[rt] = f4{0.f, 0.f, 0.f, 0.f};
    const int kc = 16 * kt + lr;
    const bool kok = kc < K;
    for (int c0 = 0; c0 < N16; c0 += 16 * PF) {
      float b[PF][4];
#pragma unroll
      for (int u = 0; u < PF; u++)
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const int n = c0 + 16 * u + 4 * lk + m;
          b[u][m] = (kok && n < N) ? Wg[(size_t)n * K + kc] : 0.f;
        }
#pragma unroll
      for (int u = 0; u < PF; u++) {
        if (c0 + 16 * u < N16) {
          const int n0 = c0 + 16 * u + 4 * lk;
#pragma unroll
          for (int rt = 0; rt < RT; rt++) {
            const f4 av = *(const f4*)(dZ + (16 * rt + lr) * ldz + n0);
#pragma unroll
            for (int m = 0; m < 4; m++) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b[u][m], acc[rt], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int v = 0; v < 4; v++) epi(16 * rt + 4 * lk + v, kc, kok, acc[rt][v]);
  }
}

// dW[n][k] = sum_s dZ[s][n] A[s][k] over the S samples of the workgroup (both in LDS, padded columns zero), written to the
// workgroup's partial-gradient block dWg [N][K]; db[n] = sum_s dZ[s][n].
template <int RT>
__device__ __forceinline__ void gemm_G(const float* dZ, int ldz, int N, const float* A, int lda, int K, float* __restrict__ dWg, float* __restrict__ dbg) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int NT = (N + 15) >> 4, KT = (K + 15) >> 4;
  int nt = wv / KT, kt = wv - nt * KT;
  for (int t = wv; t < NT * KT; t += 4) {
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 16 * RT; c += 16) {
      const int s0 = c + 4 * lk;
#pragma unroll
      for (int m = 0; m < 4; m++)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dZ[(s0 + m) * ldz + 16 * nt + lr], A[(s0 + m) * lda + 16 * kt + lr], acc, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int n = 16 * nt + 4 * lk + v, k = 16 * kt + lr;
      if (n < N && k < K) dWg[(size_t)n * K + k] = acc[v];
    }
    kt += 4;
    while (kt >= KT) { kt -= KT; nt++; }
  }
  // bias gradient: column sums of dZ; 16 RT rows split over the lanes of a 4-lane group
  for (int i = threadIdx.x; i < 4 * ((N + 3) & ~3); i += NTHREADS) {
    const int n = i >> 2, q = i & 3;
    float sum = 0.f;
    if (n < N)
#pragma unroll
      for (int r = 0; r < 4 * RT; r++) sum += dZ[(4 * RT * q + r) * ldz + n];
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (q == 0 && n < N) dbg[n] = sum;
  }
}

// normalised observation rows of the workgroup's samples into LDS (rows >= cnt and columns >= in are zero).  The row indices go
// through LDS first and the element loop requests XU rows' worth of loads before it uses any: a gathered row is an HBM round trip,
// and the loop is 14 (hand) ... 50 (leg) elements per thread.
template <int RT, class RowF>
__device__ __forceinline__ void load_x(const NetD* d, float* L, const float* __restrict__ obs, int od, const float* __restrict__ mean,
                                       const float* __restrict__ sd, int cnt, RowF rowf, float* __restrict__ obs_copy, long long* rows) {
  const int in = d->in, in16 = (in + 15) & ~15, ldx = d->ldx;
  float* X = L + d->oX;
  if ((int)threadIdx.x < 16 * RT) rows[threadIdx.x] = (int)threadIdx.x < cnt ? (long long)rowf(threadIdx.x) : 0;
  __syncthreads();
  // a thread owns columns c = tid, tid + 256, ... of every row: no index arithmetic beyond adds, and the 16 RT row loads of a
  // column are requested together
  const bool nrm = mean != nullptr;
  for (int c = threadIdx.x; c < in16; c += NTHREADS) {
    const bool cok = c < in;
    const float m = (cok && nrm) ? mean[c] : 0.f, q = (cok && nrm) ? 1.f / sd[c] : 1.f;
    float x[16 * RT];
#pragma unroll
    for (int r = 0; r < 16 * RT; r++) x[r] = (cok && r < cnt) ? obs[(size_t)rows[r] * od + c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16 * RT; r++) {
      if (cok && r < cnt && obs_copy) obs_copy[(size_t)rows[r] * od + c] = x[r];
      X[r * ldx + c] = (nrm && r < cnt) ? fminf(fmaxf((x[r] - m) * q, -5.f), 5.f) : x[r];      // rows >= cnt stay zero with the normaliser on too
    }
  }
  __syncthreads();
}

// forward pass of one network over the workgroup's samples: every hidden layer's pre-activations Z[l] AND activations A[l] stay in
// LDS (the backward pass wants both: recomputing A costs a stage -- a barrier and a pass of exponentials -- per layer)
template <int RT>
__device__ __forceinline__ void forward(const NetD* d, float* L, const float* __restrict__ Pn) {
  const int nl = d->nl, ldo = d->ldo;
  const float* Ain = L + d->oX;
  int lda = d->ldx, K = d->in;
  for (int l = 0; l < nl; l++) {
    const int N = d->w[l];
    const float* Wg = Pn + d->woff[l];
    const float* bg = Pn + d->boff[l];
    if (l < nl - 1) {
      float* Z = L + d->oZ[l];
      float* A = L + d->oA[l];
      const int ldz = d->ldz[l];
      gemm_F<RT>(Ain, lda, K, Wg, bg, N, [&](int s, int n, bool ok, float z) {
        Z[s * ldz + n] = ok ? z : 0.f;
        A[s * ldz + n] = ok ? silu_(z) : 0.f;
      });
      __syncthreads();
      PPROF(20 + l);
      Ain = A; lda = ldz; K = N;
    } else {
      float* O = L + d->oOUT;
      gemm_F<RT>(Ain, lda, K, Wg, bg, N, [&](int s, int n, bool ok, float z) { O[s * ldo + n] = ok ? z : 0.f; });
      __syncthreads();
    }
  }
}

// backward pass: OUT holds d loss / d output; partial gradients of every layer go to `part` (the network's parameter layout).
// Per layer ONE stage: the weight-gradient tiles (LDS x LDS -> global) and the input-gradient tiles (LDS x weights -> LDS) have no
// dependence on each other.
template <int RT>
__device__ __forceinline__ void backward(const NetD* d, float* L, const float* __restrict__ Pn, float* __restrict__ part) {
  const int nl = d->nl, ldh = d->ldh;
  const float* dZ = L + d->oOUT;
  int ldd = d->ldo;
  float* Dn = L + d->oD0;
  float* Do = L + d->oD1;
  for (int l = nl - 1; l >= 0; l--) {
    const int N = d->w[l], K = l ? d->w[l - 1] : d->in;
    const float* Ain = l ? L + d->oA[l - 1] : L + d->oX;
    const int lda = l ? d->ldz[l - 1] : d->ldx;
    gemm_G<RT>(dZ, ldd, N, Ain, lda, K, part + d->woff[l], part + d->boff[l]);
    PPROF(31 + 3 * l);
    if (l) {
      const float* Z = L + d->oZ[l - 1];
      const int ldz = d->ldz[l - 1];
      float* D = Dn;
      gemm_B<RT>(dZ, ldd, N, Pn + d->woff[l], K, [&](int s, int k, bool ok, float v) { D[s * ldh + k] = ok ? v * dsilu_(Z[s * ldz + k]) : 0.f; });
      __syncthreads();
      PPROF(32 + 3 * l);
      dZ = D; ldd = ldh;
      float* t = Dn; Dn = Do; Do = t;
    }
  }
}

struct GradArgs {
  const float* P; const float* obs; const float* mean; const float* sd; const long long* idx;
  const float* raw; const float* lold; const float* adv; const float* ret;
  const float* enoise;      // [B][act_dim] standard-normal draws for the entropy's log-det-Jacobian sample, or null (pre-squash entropy only)
  float* part_pi; float* part_vf;
  const NetD* dpi; const NetD* dvf;
  int mb, od, ad, nb_pi, voff, squash;
  float eps, entc, vc;
};

template <int RT>
__global__ __launch_bounds__(NTHREADS) void k_ppo_grad(GradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float L[];
  constexpr int S = 16 * RT, LPS = NTHREADS / S;         // lanes per sample in the loss stage
  const bool is_pi = (int)blockIdx.x < a.nb_pi;
  const int b = is_pi ? blockIdx.x : blockIdx.x - a.nb_pi;
  // the network descriptor goes through LDS: every layer reads a handful of its fields, and from global memory each read is an L2
  // round trip in front of the layer's first weight load
  __shared__ NetD sdesc;
  __shared__ long long rows[32];
  __shared__ float s_old[32], s_adv[32];
  PPROF(0);
  {
    const int* src = (const int*)(is_pi ? a.dpi : a.dvf);
    for (int i = threadIdx.x; i < (int)(sizeof(NetD) / sizeof(int)); i += NTHREADS) ((int*)&sdesc)[i] = src[i];
    __syncthreads();
  }
  PPROF(1);
  const NetD* d = &sdesc;
  float* s_raw = L + d->oAUX;
  float* s_eps = s_raw + S * a.ad;      // noise rows of the entropy's log-det-Jacobian sample (a.enoise)
  const float* Pn = a.P + (is_pi ? 0 : a.voff);
  const int cnt = min(S, a.mb - b * S);
  const long long* idx = a.idx + (size_t)b * S;
  load_x<RT>(d, L, a.obs, a.od, a.mean, a.sd, cnt, [&](int s) { return idx[s]; }, nullptr, rows);
  PPROF(2);
  // the per-sample operands of the loss are requested now and used after the forward pass
  if (is_pi) {
    const int ad = a.ad;
    for (int q = threadIdx.x; q < ad; q += NTHREADS) {         // act_dim <= 128 columns: the 16 RT row loads of a column together
      float r[16 * RT];
#pragma unroll
      for (int u = 0; u < 16 * RT; u++) r[u] = u < cnt ? a.raw[(size_t)rows[u] * ad + q] : 0.f;
#pragma unroll
      for (int u = 0; u < 16 * RT; u++) s_raw[u * ad + q] = r[u];
      if (a.enoise) {
#pragma unroll
        for (int u = 0; u < 16 * RT; u++) r[u] = u < cnt ? a.enoise[(size_t)rows[u] * ad + q] : 0.f;
#pragma unroll
        for (int u = 0; u < 16 * RT; u++) s_eps[u * ad + q] = r[u];
      }
    }
    if ((int)threadIdx.x < cnt) { s_old[threadIdx.x] = a.lold[rows[threadIdx.x]]; s_adv[threadIdx.x] = a.adv[rows[threadIdx.x]]; }
  } else if ((int)threadIdx.x < cnt) {
    s_old[threadIdx.x] = a.ret[rows[threadIdx.x]];
  }
  PPROF(3);
  forward<RT>(d, L, Pn);
  PPROF(4);
  float* O = L + d->oOUT;
  const int ldo = d->ldo;
  const float inv_mb = 1.f / (float)a.mb;
  if (is_pi) {
    const int s = threadIdx.x / LPS, j = threadIdx.x - s * LPS, ad = a.ad;
    const bool valid = s < cnt;
    float lp = 0.f;
    if (valid)
      for (int q = j; q < ad; q += LPS) {
        const float m = O[s * ldo + q], sd = softplus_(O[s * ldo + ad + q]) + 1e-3f, r = s_raw[s * ad + q];
        const float z = (r - m) / sd;
        lp += -0.5f * z * z - logf(sd) - HALF_LOG_2PI;
        lp -= a.squash == MM_PPO_SQUASH_TANH ? 2.f * (0.69314718055994530942f - r - softplus_(-2.f * r)) : (-softplus_(-r) - softplus_(r));
      }
#pragma unroll
    for (int o = LPS >> 1; o; o >>= 1) lp += __shfl_xor(lp, o);
    float glp = 0.f;
    if (valid) {
      const float ratio = expf(lp - s_old[s]), A = s_adv[s];
      const float cl = fminf(fmaxf(ratio, 1.f - a.eps), 1.f + a.eps);
      glp = (ratio * A <= cl * A) ? -A * inv_mb * ratio : 0.f;       // d loss / d logp through min(r A, clip(r) A)
    }
    for (int q = j; q < ad; q += LPS) {
      float dm = 0.f, dro = 0.f;
      if (valid) {
        const float m = O[s * ldo + q], o = O[s * ldo + ad + q], sd = softplus_(o) + 1e-3f, r = s_raw[s * ad + q];
        const float isd = 1.f / sd, z = (r - m) * isd;
        dm = glp * z * isd;
        float dsd = glp * (z * z * isd - isd) - a.entc * inv_mb * isd;
        if (a.enoise) {
          // brax NormalTanhDistribution.entropy: + log|d squash / d x| at a reparametrised sample x = m + sd e; d/dx = -2 tanh(x)
          // (tanh) or 1 - 2 sigma(x) (sigmoid squashing)
          const float e = s_eps[s * ad + q], x = m + sd * e;
          const float dl = a.squash == MM_PPO_SQUASH_TANH ? -2.f * tanhf(x) : 1.f - 2.f / (1.f + expf(-x));
          dm -= a.entc * inv_mb * dl;
          dsd -= a.entc * inv_mb * dl * e;
        }
        dro = dsd * dsoftplus_(o);
      }
      O[s * ldo + q] = dm;
      O[s * ldo + ad + q] = dro;
    }
  } else {
    if ((int)threadIdx.x < S) {
      const int s = threadIdx.x;
      O[s * ldo] = s < cnt ? 2.f * a.vc * inv_mb * (O[s * ldo] - s_old[s]) : 0.f;
    }
  }
  __syncthreads();
  PPROF(5);
  backward<RT>(d, L, Pn, (is_pi ? a.part_pi : a.part_vf) + (size_t)b * d->npar);
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  const float t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

// grad[p] = sum over the workgroups' partials in a fixed order (deterministic): a workgroup owns 64 consecutive parameters, its four
// waves each add every fourth partial (four independent accumulators: 16 loads in flight per lane), the four sums meet in LDS.
// Sum of squares of the workgroup's slice -> blocksq[block].  64 parameters never straddle the two networks' partial arrays only if
// np_pi is a multiple of 64 -- it is not in general, so every lane picks its own source.
constexpr int RED_P = 64;
__global__ __launch_bounds__(NTHREADS) void k_ppo_reduce(const float* __restrict__ part_pi, int nb_pi, int np_pi, const float* __restrict__ part_vf,
                                                         int nb_vf, int np_vf, float* __restrict__ grad, float* __restrict__ blocksq) {
  __shared__ float sh[4][RED_P];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int p = blockIdx.x * RED_P + lane;
  float g = 0.f;
  if (p < np_pi + np_vf) {
    const bool pi = p < np_pi;
    const float* src = pi ? part_pi + p : part_vf + (p - np_pi);
    const int nb = pi ? nb_pi : nb_vf;
    const size_t st = pi ? np_pi : np_vf;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    int b = grp;
    for (; b + 12 < nb; b += 16) {
      g0 += src[(size_t)b * st]; g1 += src[(size_t)(b + 4) * st]; g2 += src[(size_t)(b + 8) * st]; g3 += src[(size_t)(b + 12) * st];
    }
    for (; b < nb; b += 4) g0 += src[(size_t)b * st];
    g = (g0 + g1) + (g2 + g3);
  }
  sh[grp][lane] = g;
  __syncthreads();
  float q = 0.f;
  if (grp == 0) {
    g = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    if (p < np_pi + np_vf) grad[p] = g;
    q = g * g;
#pragma unroll
    for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) blocksq[blockIdx.x] = q;
  }
}

__global__ __launch_bounds__(RED_P) void k_ppo_sumsq(const float* __restrict__ grad, int np, float* __restrict__ blocksq) {
  const int p = blockIdx.x * RED_P + threadIdx.x;
  const float g = p < np ? grad[p] : 0.f;
  float q = g * g;
#pragma unroll
  for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o);
  if (threadIdx.x == 0) blocksq[blockIdx.x] = q;
}

// torch.nn.utils.clip_grad_norm_ (coef = min(1, max_norm / (norm + 1e-6))) + torch.optim.Adam (bias-corrected, eps outside the root)
__global__ __launch_bounds__(NTHREADS) void k_ppo_adam(float* __restrict__ P, const float* __restrict__ grad, float* __restrict__ m1, float* __restrict__ m2,
                                                       const float* __restrict__ blocksq, int nbq, float* step, unsigned* done, int np, float gscale,
                                                       float lr, float b1, float b2, float eps, float max_norm) {
  __shared__ float sh[4];
  float q = 0.f;
  for (int i = threadIdx.x; i < nbq; i += NTHREADS) q += blocksq[i];
  const float norm = sqrtf(block_sum(q, sh)) * gscale;
  const float coef = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  const int p = blockIdx.x * NTHREADS + threadIdx.x;
  const float t = *step + 1.f;                 // this update's index; the LAST workgroup to finish stores it (no workgroup waits)
  if (p < np) {
    const float g = grad[p] * gscale * coef;
    const float a = b1 * m1[p] + (1.f - b1) * g, v = b2 * m2[p] + (1.f - b2) * g * g;
    m1[p] = a; m2[p] = v;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    P[p] -= (lr / bc1) * a / (sqrtf(v) / sqrtf(bc2) + eps);
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(done, 1u) == gridDim.x - 1) { *step = t; *done = 0u; }
}

struct ActArgs {
  const float* P; const float* obs; const float* mean; const float* sd; const float* noise;
  float* obs_out; float* raw_out; float* logp_out; float* value_out; float* action_out;
  const NetD* dpi; const NetD* dvf;
  int n, od, ad, nb_pi, voff, squash;
};

template <int RT>
__global__ __launch_bounds__(NTHREADS) void k_ppo_act(ActArgs a) {
  extern __shared__ __attribute__((aligned(16))) float L[];
  constexpr int S = 16 * RT, LPS = NTHREADS / S;
  const bool is_pi = (int)blockIdx.x < a.nb_pi;
  const int b = is_pi ? blockIdx.x : blockIdx.x - a.nb_pi;
  __shared__ NetD sdesc;
  __shared__ long long rows[32];
  {
    const int* src = (const int*)(is_pi ? a.dpi : a.dvf);
    for (int i = threadIdx.x; i < (int)(sizeof(NetD) / sizeof(int)); i += NTHREADS) ((int*)&sdesc)[i] = src[i];
    __syncthreads();
  }
  const NetD* d = &sdesc;
  const float* Pn = a.P + (is_pi ? 0 : a.voff);
  const int cnt = min(S, a.n - b * S), base = b * S;
  load_x<RT>(d, L, a.obs, a.od, a.mean, a.sd, cnt, [&](int s) { return base + s; }, is_pi ? a.obs_out : nullptr, rows);
  forward<RT>(d, L, Pn);
  const float* O = L + d->oOUT;
  const int ldo = d->ldo;
  if (is_pi) {
    const int s = threadIdx.x / LPS, j = threadIdx.x - s * LPS, ad = a.ad;
    const bool valid = s < cnt;
    const size_t row = (size_t)(base + (valid ? s : 0));
    float lp = 0.f;
    if (valid)
      for (int q = j; q < ad; q += LPS) {
        const float m = O[s * ldo + q], sd = softplus_(O[s * ldo + ad + q]) + 1e-3f, e = a.noise[row * ad + q];
        const float r = m + sd * e, z = (r - m) / sd;
        a.raw_out[row * ad + q] = r;
        a.action_out[row * ad + q] = a.squash == MM_PPO_SQUASH_TANH ? tanhf(r) : sigmoidf_(r);
        lp += -0.5f * z * z - logf(sd) - HALF_LOG_2PI;
        lp -= a.squash == MM_PPO_SQUASH_TANH ? 2.f * (0.69314718055994530942f - r - softplus_(-2.f * r)) : (-softplus_(-r) - softplus_(r));
      }
#pragma unroll
    for (int o = LPS >> 1; o; o >>= 1) lp += __shfl_xor(lp, o);
    if (valid && j == 0) a.logp_out[row] = lp;
  } else if ((int)threadIdx.x < cnt) {
    a.value_out[base + threadIdx.x] = O[threadIdx.x * ldo];
  }
}

__global__ void k_ppo_store(const float* __restrict__ rwd, int cols, int col, float scale, const uint8_t* __restrict__ ended,
                            const uint8_t* __restrict__ trunc, int n, float* __restrict__ rew, float* __restrict__ tro, float* __restrict__ teo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const float en = ended[e] ? 1.f : 0.f, tr = (trunc && trunc[e]) ? 1.f : 0.f;
  rew[e] = rwd[(size_t)e * cols + col] * scale;
  tro[e] = tr * en;
  teo[e] = en * (1.f - tr);
}

int r16(int x) { return (x + 15) & ~15; }

// LDS plan of a workgroup of S samples
void plan(NetD& d, int S, int aux_cols) {
  int maxh = 16;
  for (int l = 0; l + 1 < d.nl; l++) maxh = std::max(maxh, r16(d.w[l]));
  d.ldx = r16(d.in) + 4; d.ldh = maxh + 4; d.ldo = r16(d.w[d.nl - 1]) + 4;
  int o = 0;
  d.oX = o; o += S * d.ldx;
  for (int l = 0; l < MM_PPO_MAX_LAYERS; l++) { d.ldz[l] = 0; d.oZ[l] = 0; d.oA[l] = 0; }
  for (int l = 0; l + 1 < d.nl; l++) { d.ldz[l] = r16(d.w[l]) + 4; d.oZ[l] = o; o += S * d.ldz[l]; d.oA[l] = o; o += S * d.ldz[l]; }
  d.oOUT = o; o += S * d.ldo;
  d.oD0 = o; o += S * d.ldh;
  d.oD1 = o; o += S * d.ldh;
  d.oAUX = o; o += S * aux_cols;
  d.total = o;
}

}  // namespace

struct mm_ppo {
  mm_ppo_config cfg;
  int device = 0;
  NetD net[2][2];            // [policy, value][RT - 1]
  NetD* dnet = nullptr;      // the same four descriptors on the device
  int np_pi = 0, np_vf = 0, np = 0;
  int max_rt = 1;            // 2: the 32-sample plan fits in LDS for both networks
  int force_rt = 0;          // MYOSIM_PPO_SAMPLES=16|32 (A/B)
  size_t lds[2] = {0, 0};    // dynamic LDS bytes per RT
  float* part = nullptr;     // partial gradients: [nb_max][np_pi] then [nb_max][np_vf]
  int nb_max = 0;
  float *m1 = nullptr, *m2 = nullptr, *blocksq = nullptr, *step = nullptr;   // step: [0] update counter, [1] the finished-workgroup counter of k_ppo_adam
  int nbq = 0;
  const float* ent_noise = nullptr;   // mm_ppo_set_entropy_noise: [B][act_dim] draws for the entropy's squash log-det-Jacobian sample, or null
};

extern "C" const char* mm_ppo_last_error(void) { return g_perr.c_str(); }

// the launches of a handle go to ITS device (workspace, descriptors), whatever device is current in the calling thread
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

extern "C" void mm_ppo_destroy(mm_ppo* h) {
  if (!h) return;
  for (void* p : {(void*)h->dnet, (void*)h->part, (void*)h->m1, (void*)h->m2, (void*)h->blocksq, (void*)h->step}) (void)hipFree(p);
  delete h;
}

static int fill_net(NetD& d, int in, int nl, const int* w, const char* what) {
  if (nl < 1 || nl > MM_PPO_MAX_LAYERS) return pfail(MM_EUNSUPPORTED, std::string("mm_ppo_create: ") + what + " network has " + std::to_string(nl) + " layers (1.." + std::to_string(MM_PPO_MAX_LAYERS) + ")");
  d = NetD{};
  d.nl = nl; d.in = in;
  int o = 0, k = in;
  for (int l = 0; l < nl; l++) {
    const int lim = l + 1 < nl ? MM_PPO_MAX_WIDTH : MM_PPO_MAX_OUT;
    if (w[l] < 1 || w[l] > lim) return pfail(MM_EUNSUPPORTED, std::string("mm_ppo_create: ") + what + " layer " + std::to_string(l) + " has width " + std::to_string(w[l]) + " (fused kernels: hidden <= " + std::to_string(MM_PPO_MAX_WIDTH) + ", output <= " + std::to_string(MM_PPO_MAX_OUT) + ")");
    d.w[l] = w[l]; d.woff[l] = o; o += w[l] * k; d.boff[l] = o; o += w[l]; k = w[l];
  }
  d.npar = o;
  return MM_OK;
}

extern "C" int mm_ppo_create(const mm_ppo_config* c, int device, mm_ppo** out) {
  if (!c || !out) return pfail(MM_EARG, "mm_ppo_create: null argument");
  if (c->size < (int)sizeof(mm_ppo_config)) return pfail(MM_EARG, "mm_ppo_create: mm_ppo_config.size " + std::to_string(c->size) + " < " + std::to_string(sizeof(mm_ppo_config)));
  if (c->obs_dim < 1 || c->obs_dim > MM_PPO_MAX_OBS || c->act_dim < 1 || 2 * c->act_dim > MM_PPO_MAX_OUT) return pfail(MM_EUNSUPPORTED, "mm_ppo_create: obs_dim / act_dim outside the fused kernels' limits");
  if (c->pi_layers < 1 || c->pi_layers > MM_PPO_MAX_LAYERS || c->pi_widths[c->pi_layers - 1] != 2 * c->act_dim) return pfail(MM_EARG, "mm_ppo_create: the policy's last layer must have 2 act_dim outputs");
  if (c->vf_layers < 1 || c->vf_layers > MM_PPO_MAX_LAYERS || c->vf_widths[c->vf_layers - 1] != 1) return pfail(MM_EARG, "mm_ppo_create: the value network's last layer must have 1 output");
  if (c->squash != MM_PPO_SQUASH_TANH && c->squash != MM_PPO_SQUASH_SIGMOID) return pfail(MM_EARG, "mm_ppo_create: squash");
  if (c->max_minibatch < 1) return pfail(MM_EARG, "mm_ppo_create: max_minibatch");
  mm_ppo* h = new mm_ppo();
  h->cfg = *c; h->device = device;
  int rc;
  for (int r = 0; r < 2; r++) {
    if ((rc = fill_net(h->net[0][r], c->obs_dim, c->pi_layers, c->pi_widths, "policy")) || (rc = fill_net(h->net[1][r], c->obs_dim, c->vf_layers, c->vf_widths, "value"))) { delete h; return rc; }
    plan(h->net[0][r], 16 * (r + 1), 2 * c->act_dim); plan(h->net[1][r], 16 * (r + 1), 0);      // raw actions + the entropy sample's noise rows
    h->lds[r] = sizeof(float) * (size_t)std::max(h->net[0][r].total, h->net[1][r].total);
  }
  h->np_pi = h->net[0][0].npar; h->np_vf = h->net[1][0].npar; h->np = h->np_pi + h->np_vf;
  const size_t lds_limit = 150 * 1024;       // 160 KB per CU minus the kernels' static arrays (descriptor, row indices, per-sample scalars) and a margin
  if (h->lds[0] > lds_limit) { const size_t need = h->lds[0]; delete h; return pfail(MM_ELDS, "mm_ppo_create: a 16-sample workgroup needs " + std::to_string(need) + " B of LDS"); }
  h->max_rt = h->lds[1] <= lds_limit ? 2 : 1;
  if (const char* e = getenv("MYOSIM_PPO_SAMPLES")) h->force_rt = atoi(e) == 32 ? 2 : atoi(e) == 16 ? 1 : 0;
  int cur = 0;
  (void)hipGetDevice(&cur);
#define CREATE_CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::string m_ = std::string(#x) + ": " + hipGetErrorString(e_); (void)hipSetDevice(cur); mm_ppo_destroy(h); return pfail(MM_EHIP, m_); } } while (0)
  CREATE_CHK(hipSetDevice(device));
  CREATE_CHK(hipMalloc(&h->dnet, 4 * sizeof(NetD)));
  CREATE_CHK(hipMemcpy(h->dnet, &h->net[0][0], 4 * sizeof(NetD), hipMemcpyHostToDevice));
  h->nb_max = (c->max_minibatch + 15) / 16;
  CREATE_CHK(hipMalloc(&h->part, sizeof(float) * (size_t)h->nb_max * h->np));
  h->nbq = (h->np + RED_P - 1) / RED_P;          // workgroups of k_ppo_reduce / k_ppo_sumsq = entries of blocksq
  CREATE_CHK(hipMalloc(&h->m1, sizeof(float) * h->np)); CREATE_CHK(hipMalloc(&h->m2, sizeof(float) * h->np));
  CREATE_CHK(hipMalloc(&h->blocksq, sizeof(float) * h->nbq)); CREATE_CHK(hipMalloc(&h->step, 2 * sizeof(float)));
  CREATE_CHK(hipMemset(h->m1, 0, sizeof(float) * h->np)); CREATE_CHK(hipMemset(h->m2, 0, sizeof(float) * h->np));
  CREATE_CHK(hipMemset(h->blocksq, 0, sizeof(float) * h->nbq)); CREATE_CHK(hipMemset(h->step, 0, 2 * sizeof(float)));
  // The dynamic-LDS ceiling is per-KERNEL state of the process, not of this handle: set it to the fixed upper bound every handle is
  // checked against (lds_limit), so that creating a second handle with smaller networks never lowers it under a live one's launches.
  CREATE_CHK(hipFuncSetAttribute((const void*)k_ppo_grad<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
  CREATE_CHK(hipFuncSetAttribute((const void*)k_ppo_act<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
  CREATE_CHK(hipFuncSetAttribute((const void*)k_ppo_grad<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
  CREATE_CHK(hipFuncSetAttribute((const void*)k_ppo_act<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit));
  CREATE_CHK(hipDeviceSynchronize());
  (void)hipSetDevice(cur);
#undef CREATE_CHK
  *out = h;
  return MM_OK;
}

// tools builds only (-DMM_PPO_PROF=1): buf = [workgroups][64] uint64 stage stamps of the next mm_ppo_grad launches; NULL switches it off
extern "C" int mm_ppo_debug_set_prof(void* buf) {
  PHIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_ppo_prof), &buf, sizeof(buf)));
  return MM_OK;
}

extern "C" int mm_ppo_param_count(const mm_ppo* h) { return h ? h->np : 0; }
extern "C" int mm_ppo_value_offset(const mm_ppo* h) { return h ? h->np_pi : 0; }

extern "C" int mm_ppo_reset_optimizer(mm_ppo* h, void* stream) {
  if (!h) return pfail(MM_EARG, "mm_ppo_reset_optimizer: null handle");
  PHIPCHK(hipMemsetAsync(h->m1, 0, sizeof(float) * h->np, (hipStream_t)stream));
  PHIPCHK(hipMemsetAsync(h->m2, 0, sizeof(float) * h->np, (hipStream_t)stream));
  PHIPCHK(hipMemsetAsync(h->step, 0, 2 * sizeof(float), (hipStream_t)stream));
  return MM_OK;
}

// samples per workgroup: 16.  A workgroup is one dependent chain of ~20 stages (one wave per SIMD: nothing inside it hides a
// stage's latencies), so the launch takes as long as the chain as long as every workgroup is resident -- and the chain of 16 samples
// is the shorter one (three 16-sample workgroups fit a CU's LDS for the hand networks).  MYOSIM_PPO_SAMPLES=32 for A/B.
static int pick_rt(const mm_ppo* h, int rows) {
  (void)rows;
  return h->force_rt ? std::min(h->force_rt, h->max_rt) : 1;
}

extern "C" int mm_ppo_act(mm_ppo* h, const float* params, const float* obs, const float* obs_mean, const float* obs_std, const float* noise,
                          int nenv, float* obs_out, float* raw_out, float* logp_out, float* value_out, float* action_out, void* stream) {
  if (!h || !params || !obs || !value_out || nenv <= 0 || (obs_mean && !obs_std)) return pfail(MM_EARG, "mm_ppo_act: bad argument");
  if (action_out && (!noise || !raw_out || !logp_out)) return pfail(MM_EARG, "mm_ppo_act: noise / raw_out / logp_out are required with action_out");
  DeviceGuard guard(h->device);
  const int rt = pick_rt(h, nenv), S = 16 * rt, nb = (nenv + S - 1) / S;
  ActArgs a{params, obs, obs_mean, obs_std, noise, obs_out, raw_out, logp_out, value_out, action_out,
            h->dnet + (rt - 1), h->dnet + 2 + (rt - 1), nenv, h->cfg.obs_dim, h->cfg.act_dim, action_out ? nb : 0, h->np_pi, h->cfg.squash};
  const dim3 grid(action_out ? 2 * nb : nb), block(NTHREADS);
  if (rt == 2) hipLaunchKernelGGL(k_ppo_act<2>, grid, block, h->lds[1], (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_ppo_act<1>, grid, block, h->lds[0], (hipStream_t)stream, a);
  PHIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_ppo_store(const float* rwd, int rwd_cols, int rwd_col, float reward_scale, const uint8_t* ended, const uint8_t* truncated,
                            int nenv, float* reward_out, float* trunc_out, float* term_out, void* stream) {
  if (!rwd || !ended || !reward_out || !trunc_out || !term_out || nenv <= 0 || rwd_col < 0 || rwd_col >= rwd_cols) return pfail(MM_EARG, "mm_ppo_store: bad argument");
  hipLaunchKernelGGL(k_ppo_store, dim3((nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, rwd, rwd_cols, rwd_col, reward_scale, ended,
                     truncated, nenv, reward_out, trunc_out, term_out);
  PHIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_ppo_grad(mm_ppo* h, const float* params, const float* obs, const float* obs_mean, const float* obs_std, const int64_t* idx,
                           int mb, const float* raw, const float* logp_old, const float* adv, const float* ret, float* grad_out, void* stream) {
  if (!h || !params || !obs || !idx || !raw || !logp_old || !adv || !ret || !grad_out || (obs_mean && !obs_std)) return pfail(MM_EARG, "mm_ppo_grad: null argument");
  if (mb < 1 || mb > h->cfg.max_minibatch) return pfail(MM_EARG, "mm_ppo_grad: minibatch of " + std::to_string(mb) + " rows, workspace sized for " + std::to_string(h->cfg.max_minibatch));
  DeviceGuard guard(h->device);
  const int rt = pick_rt(h, mb), S = 16 * rt, nb = (mb + S - 1) / S;
  float* part_pi = h->part;
  float* part_vf = h->part + (size_t)h->nb_max * h->np_pi;
  GradArgs a{params, obs, obs_mean, obs_std, (const long long*)idx, raw, logp_old, adv, ret, h->ent_noise, part_pi, part_vf,
             h->dnet + (rt - 1), h->dnet + 2 + (rt - 1), mb, h->cfg.obs_dim, h->cfg.act_dim, nb, h->np_pi, h->cfg.squash,
             h->cfg.clipping_epsilon, h->cfg.entropy_cost, h->cfg.value_cost};
  if (rt == 2) hipLaunchKernelGGL(k_ppo_grad<2>, dim3(2 * nb), dim3(NTHREADS), h->lds[1], (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_ppo_grad<1>, dim3(2 * nb), dim3(NTHREADS), h->lds[0], (hipStream_t)stream, a);
  PHIPCHK(hipGetLastError());
  hipLaunchKernelGGL(k_ppo_reduce, dim3(h->nbq), dim3(NTHREADS), 0, (hipStream_t)stream, part_pi, nb, h->np_pi, part_vf, nb, h->np_vf, grad_out,
                     h->blocksq);
  PHIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_ppo_set_entropy_noise(mm_ppo* h, const float* noise) {
  if (!h) return pfail(MM_EARG, "mm_ppo_set_entropy_noise: null handle");
  h->ent_noise = noise;
  return MM_OK;
}

extern "C" int mm_ppo_adam(mm_ppo* h, float* params, const float* grad, float grad_scale, int recompute_norm, void* stream) {
  if (!h || !params || !grad) return pfail(MM_EARG, "mm_ppo_adam: null argument");
  DeviceGuard guard(h->device);
  if (recompute_norm) {
    hipLaunchKernelGGL(k_ppo_sumsq, dim3(h->nbq), dim3(RED_P), 0, (hipStream_t)stream, grad, h->np, h->blocksq);
    PHIPCHK(hipGetLastError());
  }
  const mm_ppo_config& c = h->cfg;
  hipLaunchKernelGGL(k_ppo_adam, dim3((h->np + NTHREADS - 1) / NTHREADS), dim3(NTHREADS), 0, (hipStream_t)stream, params, grad, h->m1, h->m2, h->blocksq, h->nbq, h->step,
                     (unsigned*)(h->step + 1), h->np, grad_scale, c.learning_rate, c.beta1, c.beta2, c.adam_eps, c.max_grad_norm);
  PHIPCHK(hipGetLastError());
  return MM_OK;
}
