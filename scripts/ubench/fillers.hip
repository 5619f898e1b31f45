// What does one wavefront per SIMD hide behind a v_mfma_f32_32x32x16_f16?  Cycles per MFMA for streams of 24 MFMAs on 8
// accumulators (the NeRF K-step's shape) with different fillers per gap.  Build: hipcc --offload-arch=gfx950 -O3 fillers.hip -o fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)

// MODE 0: bare MFMAs.  1: + 4 independent VALU per gap.  2: + 4 VALU in two dependent pairs (add -> med3).  3: + chain of 4 dependent
// VALU.  4: + 1 ds_read_b128 per gap (consumed a K-step later).  5: mode 2 + 1 ds_read_b128 per gap.  6: mode 5 + one s_barrier
// per 24 MFMAs.  7: mode 0 + one s_barrier per 24.  8: 6 VALU per gap, independent.  9: 8 VALU per gap, independent.
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[8];
  for (int b = 0; b < 8; ++b)
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  u32x4 opq[8];
  for (int b = 0; b < 8; ++b) opq[b] = u32x4{0x3c003c00u + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + b};
  f16x8 x = __builtin_bit_cast(f16x8, opq[0]);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + lane * 1e-3f + i;
  const float cmax = 65504.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int b = m & 7;
      acc[b] = MFMA(__builtin_bit_cast(f16x8, opq[b]), x, acc[b]);
      SB;
      if (MODE == 1 || MODE == 8 || MODE == 9) {
        const int n = MODE == 1 ? 4 : (MODE == 8 ? 6 : 8);
#pragma unroll
        for (int i = 0; i < n; ++i) { v[i] = v[i] * 1.0001f + 0.5f; asm volatile("" : "+v"(v[i])); }
      }
      if (MODE == 2 || MODE == 5 || MODE == 6) {
        float a0 = v[0] + v[2], a1 = v[1] + v[3];
        asm volatile("" : "+v"(a0), "+v"(a1));
        v[0] = __builtin_amdgcn_fmed3f(a0, 0.f, cmax); v[1] = __builtin_amdgcn_fmed3f(a1, 0.f, cmax);
        asm volatile("" : "+v"(v[0]), "+v"(v[1]));
      }
      if (MODE == 3) {
        float a0 = v[0] + v[2];
        asm volatile("" : "+v"(a0));
        a0 = __builtin_amdgcn_fmed3f(a0, 0.f, cmax);
        asm volatile("" : "+v"(a0));
        a0 = a0 * 1.0001f;
        asm volatile("" : "+v"(a0));
        v[0] = a0 + 0.25f;
        asm volatile("" : "+v"(v[0]));
      }
      if (MODE == 4 || MODE == 5 || MODE == 6) {
        opq[(b + 4) & 7] = reinterpret_cast<const u32x4*>(lds)[((m * 64) & 1023) + lane];
      }
      SB;
      if ((MODE == 6 || MODE == 7) && m == 11) { __builtin_amdgcn_s_barrier(); SB; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float r = 0.f;
  for (int b = 0; b < 8; ++b)
    for (int i = 0; i < 16; ++i) r += acc[b][i];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(float* out, unsigned long long* cyc, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 256, 65536>>>(out, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) k<MODE><<<256, 256, 65536>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0;
  for (int i = 0; i < 256; ++i) c += (double)h[i];
  c /= 256;
  printf("%-66s %7.3f ms  %5.1f shader cycles/MFMA (s_memtime)  clock %.2f GHz  (%s)\n", name, ms, c / (iters * 24.0), c / (ms * 1e6),
         hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 256 * 8);
  run<0>(out, cyc, "0: bare MFMAs, 8 accumulators");
  run<1>(out, cyc, "1: + 4 independent VALU per gap");
  run<8>(out, cyc, "8: + 6 independent VALU per gap");
  run<9>(out, cyc, "9: + 8 independent VALU per gap");
  run<2>(out, cyc, "2: + 4 VALU per gap in two dependent pairs");
  run<3>(out, cyc, "3: + a chain of 4 dependent VALU per gap");
  run<4>(out, cyc, "4: + 1 ds_read_b128 per gap");
  run<5>(out, cyc, "5: + two dependent VALU pairs + 1 ds_read_b128 per gap");
  run<6>(out, cyc, "6: as 5 + one s_barrier per 24 MFMAs");
  run<7>(out, cyc, "7: bare + one s_barrier per 24 MFMAs");
  return 0;
}
