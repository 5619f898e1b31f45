// Microbenchmark for a DIFFERENT decomposition of the fused NeRF tile (round 3): can a K-loop without
// LDS-staged weights sustain the matrix pipe?
//
// Today (nerf_fwd_bf16.hip): wavefront = 32 samples x ALL 256 outputs; every wavefront reads the whole 16 KiB weight slot of a
// K-step from LDS (4 x redundant: 64 KiB of LDS reads + 16 KiB of LDS-DMA writes per 768 MFMA cycles = 104 B/clk of the CU's
// 128 B/clk if it ran at the matrix rate), one barrier + 4 DMA instructions per wavefront and K-step.  Timing-only ablations
// (scripts/ablate_nerf.sh): no operand reads -18 %, no DMA + barrier -14 %, no re-packing -15..22 %.
//
// "v4": wavefront = 2 output blocks (64 neurons) x ALL 128 samples (4 groups of 32).  Weights are private to a wavefront ->
// straight from L2 into registers (4 x 1 KiB loads per K-step, prefetched), no LDS staging, no per-K-step barrier; the
// activations (B operands, hi/lo packed, 128 KiB per layer) live in LDS and every wavefront reads all of them: 8 ds_read_b128
// per 24 MFMAs (32 KiB of LDS reads per K-step and CU: a third of today's traffic).
//
// MODE 0: the v4 K-loop alone (16 K-steps per "layer", no layer boundary work)
// MODE 1: + layer boundary: barrier, 128 accumulator reads + ~5 VALU per value (bias / relu / hi-lo split), 32 KiB of LDS
//         writes per wavefront, barrier  (the re-packing is EXPOSED in this design)
// MODE 2: MODE 1 with the weight loads removed (bound of the L2 stream)
// Build: hipcc --offload-arch=gfx950 -O3 v4_kloop.hip -o v4_kloop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int KS = 16;          // K-steps per layer
constexpr int NSLOT = 160;      // K-steps in the weight stream (~ the real blob: 159)

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const char* __restrict__ w, float* out, int nlayers) {
  extern __shared__ __attribute__((aligned(16))) float act[];  // [KS][4 groups][hi, lo][64 lanes][4 floats] = 128 KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < KS * 4 * 2 * 256; i += 256) act[i] = 1e-3f * (i & 1023);
  __syncthreads();
  f32x16 acc[2][4];
  for (int b = 0; b < 2; ++b)
    for (int g = 0; g < 4; ++g)
      for (int i = 0; i < 16; ++i) acc[b][g][i] = 0.f;
  // this wavefront's weight stream: [K-step][block 0 hi, block 0 lo, block 1 hi, block 1 lo][64 lanes][16 B] = 4 KiB per K-step
  const u32x4* wp = reinterpret_cast<const u32x4*>(w + (size_t)wave * NSLOT * 4096) + lane;
  u32x4 wq[4][4];  // four K-steps in flight (16 % 4 == 0: the rotation is layer invariant)
  int s = 0;       // position in the stream
  auto wload = [&](int slot, u32x4 (&d)[4]) {
    if (MODE == 2) return;
    const u32x4* p = wp + (size_t)(slot % NSLOT) * 256;
#pragma unroll
    for (int o = 0; o < 4; ++o) d[o] = __builtin_nontemporal_load(p + o * 64);
  };
  wload(0, wq[0]); wload(1, wq[1]); wload(2, wq[2]); wload(3, wq[3]);
  for (int l = 0; l < nlayers; ++l) {
    asm volatile("" ::: "memory");  // (the activations are re-read every layer: keeps the compiler from hoisting 512 registers of LDS reads)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // B operands of this K-step: 4 sample groups x (hi, lo)
      const u32x4* b4 = reinterpret_cast<const u32x4*>(act + ks * 2048) + lane;
      f16x8 bh[4], bl[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bh[g] = __builtin_bit_cast(f16x8, b4[(g * 2 + 0) * 64]);
        bl[g] = __builtin_bit_cast(f16x8, b4[(g * 2 + 1) * 64]);
      }
      u32x4 (&cur)[4] = wq[ks % 4];
      const f16x8 a0h = __builtin_bit_cast(f16x8, cur[0]), a0l = __builtin_bit_cast(f16x8, cur[1]);
      const f16x8 a1h = __builtin_bit_cast(f16x8, cur[2]), a1l = __builtin_bit_cast(f16x8, cur[3]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[0][g] = MFMA(a0h, bh[g], acc[0][g]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[1][g] = MFMA(a1h, bh[g], acc[1][g]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[0][g] = MFMA(a0h, bl[g], acc[0][g]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[1][g] = MFMA(a1h, bl[g], acc[1][g]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[0][g] = MFMA(a0l, bh[g], acc[0][g]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[1][g] = MFMA(a1l, bh[g], acc[1][g]);
      wload(s + 4, wq[ks % 4]);
      ++s;
    }
    if (MODE >= 1) {
      __syncthreads();  // everybody has read the layer's input
      // re-pack this wavefront's 2 blocks x 4 groups x 16 values and publish them as the next layer's K-steps 4 wave .. 4 wave + 3
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            f16x8 h8, l8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float v = __builtin_fmaxf(acc[b][g][8 * m + i] + 0.01f * i, 0.f);
              const _Float16 h = (_Float16)v;
              h8[i] = h;
              l8[i] = (_Float16)(v - (float)h);
              acc[b][g][8 * m + i] = 0.f;
            }
            u32x4* d = reinterpret_cast<u32x4*>(act + (4 * wave + 2 * b + m) * 2048) + lane;
            d[(g * 2 + 0) * 64] = __builtin_bit_cast(u32x4, h8);
            d[(g * 2 + 1) * 64] = __builtin_bit_cast(u32x4, l8);
          }
      __syncthreads();
    }
  }
  float r = 0.f;
  for (int b = 0; b < 2; ++b)
    for (int g = 0; g < 4; ++g)
      for (int i = 0; i < 16; ++i) r += acc[b][g][i];
  out[blockIdx.x * 256 + threadIdx.x] = r + act[threadIdx.x];
}

template <int MODE>
void run(const char* w, float* out, int nlayers, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  k<MODE><<<256, 256, 128 * 1024>>>(w, out, nlayers);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<MODE><<<256, 256, 128 * 1024>>>(w, out, nlayers);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double nmfma = (double)nlayers * KS * 24;  // per wavefront
  printf("%-58s %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz; %s)\n", name, ms, 256.0 * 4 * nmfma * 32768 / ms / 1e9,
         ms * 1e-3 * 2.4e9 / nmfma, hipGetErrorString(hipGetLastError()));
}

int main() {
  char* w; float* out;
  const size_t wb = (size_t)4 * NSLOT * 4096 + 65536;
  hipMalloc(&w, wb); hipMemset(w, 0, wb);
  hipMalloc(&out, 256 * 256 * 4);
  const int n = 1000;  // layers of 16 K-steps
  run<0>(w, out, n, "0: v4 K-loop (weights L2->regs, activations from LDS)");
  run<1>(w, out, n, "1: + exposed layer boundary (re-pack, 32 KiB LDS writes)");
  run<2>(w, out, n, "2: as 1 without the weight stream");
  return 0;
}
