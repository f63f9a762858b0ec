// Microbenchmark: issue cost of LDS float atomics vs plain LDS writes on gfx950 (one wave, cycles per instruction).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip && ./lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(long long* out, int active, int conflict, int iters) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = 0.f;
  __syncthreads();
  // address pattern: `conflict` lanes share one word
  const int slot = lane / (conflict > 0 ? conflict : 1);
  float* p = lds + slot;
  long long t0 = clock64();
  if (lane < active) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (MODE == 0) atomicAdd(p + 64 * u, 1.0f);                       // ds_add_f32, no return
        else if (MODE == 1) p[64 * u] = (float)it;                          // ds_write_b32
        else if (MODE == 2) { float r = atomicAdd(p + 64 * u, 1.0f); if (r < -1.f) p[1] = r; }   // ds_add_rtn_f32
        else { float r = p[64 * u]; p[64 * u] = r + 1.f; }                  // read-modify-write, dependent
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  long long t1 = clock64();
  if (lane == 0) out[0] = t1 - t0;
  if (lds[lane] == 12345.f) out[1] = 1;
}

int main() {
  long long* d; hipMalloc(&d, 16);
  const int iters = 2000;
  const char* names[] = {"ds_add_f32", "ds_write_b32", "ds_add_rtn_f32", "read+write"};
  for (int mode = 0; mode < 4; mode++)
    for (int active : {64, 32, 10, 1})
      for (int conflict : {1, 5, 32}) {
        if (conflict > active) continue;
        for (int rep = 0; rep < 2; rep++) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, active, conflict, iters);
          if (mode == 1) hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, active, conflict, iters);
          if (mode == 2) hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, active, conflict, iters);
          if (mode == 3) hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, active, conflict, iters);
          hipDeviceSynchronize();
        }
        long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-16s active %2d lanes/word %2d : %.1f clock64 ticks per instruction (16 per fence)\n", names[mode], active, conflict, (double)h[0] / (iters * 16.0));
      }
  return 0;
}
