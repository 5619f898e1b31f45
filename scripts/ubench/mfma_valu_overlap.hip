// Does ONE SIMD overlap the matrix pipe with VALU work of ANOTHER wavefront?  (round 5; the three MFMA-heavy matcher kernels all show
// MFMA-busy + VALU-issue ~ 0.94 of the SIMD cycles, i.e. the two seem to add.)
// One 512-thread workgroup per CU = two wavefronts per SIMD.  Roles by wavefront number w (mode 0: w < 4 matrix, w >= 4 VALU -- one of each
// per SIMD if wavefronts go round robin over the SIMDs; mode 1: by w & 1).  Runs: matrix only, VALU only, both; reports the kernel time.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench/mfma_valu_overlap.hip -o scripts/ubench/mfma_valu_overlap && scripts/ubench/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int VKIND>
__global__ void __launch_bounds__(512) k(float* out, int n_mfma, int n_valu, int roles, int split) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool matrix = split == 0 ? (w < 4) : ((w & 1) == 0);
  if (matrix) {
    if (!(roles & 1)) return;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(lane + i); y[i] = (__bf16)(float)(lane - i); }
    for (int i = 0; i < n_mfma; ++i) {  // four independent chains: the pipe is never waiting for a result
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!(roles & 2)) return;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 1e-3f * (lane + i);
    for (int i = 0; i < n_valu; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (VKIND == 0) v[j] = __builtin_fmaf(v[j], 1.0000001f, 1e-9f);        // full-rate VALU
        else v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.5f;                       // quarter-rate transcendental + a multiply
      }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <int VKIND>
float run(float* out, int grid, int n_mfma, int n_valu, int roles, int split) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) k<VKIND><<<grid, 512>>>(out, n_mfma, n_valu, roles, split);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<VKIND><<<grid, 512>>>(out, n_mfma, n_valu, roles, split);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)grid * 512 * 4);
  const int n_mfma = 20000;  // x 4 MFMAs of 8 passes
  for (int split = 0; split < 2; ++split) {
    printf("roles by %s\n", split == 0 ? "w < 4 (matrix) / w >= 4 (VALU)" : "w & 1");
    {
      const float tm = run<0>(out, grid, n_mfma, 0, 1, split);
      printf("  matrix only: %.3f ms  (%d x 4 MFMA 32x32x16 per wavefront = %.1f cycles per MFMA at 2.4 GHz)\n", tm, n_mfma, tm * 1e-3 * 2.4e9 / (4.0 * n_mfma));
      for (int vk = 0; vk < 2; ++vk)
        for (int n_valu : {20000, 40000, 80000}) {
          const float tv = vk == 0 ? run<0>(out, grid, 0, n_valu, 2, split) : run<1>(out, grid, 0, n_valu, 2, split);
          const float tb = vk == 0 ? run<0>(out, grid, n_mfma, n_valu, 3, split) : run<1>(out, grid, n_mfma, n_valu, 3, split);
          printf("  VALU %-12s x %6d x 8: alone %.3f ms, with the matrix wavefront %.3f ms   (max %.3f, sum %.3f)\n", vk == 0 ? "v_fma_f32" : "v_exp+v_mul", n_valu, tv,
                 tb, tm > tv ? tm : tv, tm + tv);
        }
    }
  }
  return 0;
}
