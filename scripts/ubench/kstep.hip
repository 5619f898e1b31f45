// Replica of the NeRF K-step's instruction mix at one wavefront per SIMD, to see which ingredient makes the ring barrier expensive.
// Per round: 24 MFMAs on 8 accumulators; A operands double buffered in registers (opA for blocks 0-3, opB for blocks 4-7, hi and lo),
// fetched from LDS at least 8 MFMAs before their use.  Build: hipcc --offload-arch=gfx950 -O3 -w kstep.hip -o kstep
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), (c), 0, 0, 0)
#define SB __builtin_amdgcn_sched_barrier(0)

// bit 0: LDS operand reads (16 per round).  bit 1: 40 VALU in the second half.  bit 2: one s_barrier per round.  bit 3: 4 LDS-DMA pieces
// per round behind the barrier.  bit 4: barrier every SECOND round only.  bit 5: reads in two bursts of 8 instead of one per gap.
// bit 6: the DMA is waited for two rounds later (vmcnt(8)) instead of one.  bit 7: the DMA pieces go out behind the last four MFMAs of the
// first half instead of behind the barrier.  bit 8: with bit 4, both rounds' DMA (8 pieces) behind the one barrier.
template <int M>
__global__ void __launch_bounds__(256, 1) k(float* out, const char* w, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.f;
  __syncthreads();
  f32x16 acc[8];
  for (int b = 0; b < 8; ++b)
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  u32x4 ah[4], al[4], bh[4], bl[4];
  for (int b = 0; b < 4; ++b) { ah[b] = al[b] = bh[b] = bl[b] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; }
  u32x4 xh = u32x4{0x3c003c00u + lane, 0, 0, 0}, xl = u32x4{0x1c001c00u, 0, 0, 0};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + lane * 1e-3f + i;
  const u32x4* s4 = reinterpret_cast<const u32x4*>(lds) + lane;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const u32x4* sl = s4 + (it & 3) * 1024;
    // first half: blocks 0-3 with (ah, al); fetch (bh, bl)
    if ((M & 1) && (M & 32)) {
#pragma unroll
      for (int o = 0; o < 4; ++o) { bh[o] = sl[(8 + 2 * o) * 64]; bl[o] = sl[(9 + 2 * o) * 64]; }
      SB;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) { acc[o] = MFMA(ah[o], xh, acc[o]); SB; if ((M & 1) && !(M & 32)) bh[o] = sl[(8 + 2 * o) * 64]; SB; }
#pragma unroll
    for (int o = 0; o < 4; ++o) { acc[o] = MFMA(ah[o], xl, acc[o]); SB; if ((M & 1) && !(M & 32)) bl[o] = sl[(9 + 2 * o) * 64]; SB; }
    const auto* src = (const __attribute__((address_space(1))) void*)(w + (size_t)((it * 4 + wave) & 1023) * 4096 + lane * 16);
    auto* dst = (__attribute__((address_space(3))) void*)(lds + 8192 + ((it & 1) * 4 + wave) * 1024);
    const auto* src2 = (const __attribute__((address_space(1))) void*)(w + (size_t)((it * 4 + 4 + wave) & 1023) * 4096 + lane * 16);
    auto* dst2 = (__attribute__((address_space(3))) void*)(lds + 8192 + (((it + 1) & 1) * 4 + wave) * 1024);
    if ((M & 8) && (M & 128)) {
      acc[0] = MFMA(al[0], xh, acc[0]); SB; __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0); SB;
      acc[1] = MFMA(al[1], xh, acc[1]); SB; __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0); SB;
      acc[2] = MFMA(al[2], xh, acc[2]); SB; __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0); SB;
      acc[3] = MFMA(al[3], xh, acc[3]); SB; __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0); SB;
    } else {
#pragma unroll
      for (int o = 0; o < 4; ++o) { acc[o] = MFMA(al[o], xh, acc[o]); SB; }
    }
    const bool bar_now = !(M & 16) || (it & 1);
    if (M & 8) {
      if (!(M & 256) || bar_now) {
        if (M & 64) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (M & 128) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (M & 256) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if ((M & 4) && bar_now) { __builtin_amdgcn_s_barrier(); SB; }
    if ((M & 8) && !(M & 128)) {
      if (!(M & 256) || bar_now) {
        __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
        if (M & 256) {
          __builtin_amdgcn_global_load_lds(src2, dst2, 16, 0, 0);
          __builtin_amdgcn_global_load_lds(src2, dst2, 16, 1024, 0);
          __builtin_amdgcn_global_load_lds(src2, dst2, 16, 2048, 0);
          __builtin_amdgcn_global_load_lds(src2, dst2, 16, 3072, 0);
        }
      }
      if (!(M & 64) && !(M & 256)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      SB;
    }
    // second half: blocks 4-7 with (bh, bl); fetch (ah, al) of the next round
    auto work = [&](int n) {
      if (!(M & 2)) return;
#pragma unroll
      for (int i = 0; i < n; i += 2) {
        float a0 = v[i & 7] + v[(i + 2) & 7];
        asm volatile("" : "+v"(a0));
        v[i & 7] = __builtin_amdgcn_fmed3f(a0, 0.f, 65504.f);
        asm volatile("" : "+v"(v[i & 7]));
      }
    };
#pragma unroll
    for (int o = 0; o < 4; ++o) { acc[4 + o] = MFMA(bh[o], xh, acc[4 + o]); SB; work(4); SB; }
    if ((M & 1) && (M & 32)) {
#pragma unroll
      for (int o = 0; o < 4; ++o) { ah[o] = sl[(2 * o) * 64]; al[o] = sl[(1 + 2 * o) * 64]; }
      SB;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) { acc[4 + o] = MFMA(bh[o], xl, acc[4 + o]); SB; work(3); if ((M & 1) && !(M & 32)) ah[o] = sl[(2 * o) * 64]; SB; }
#pragma unroll
    for (int o = 0; o < 4; ++o) { acc[4 + o] = MFMA(bl[o], xh, acc[4 + o]); SB; work(3); if ((M & 1) && !(M & 32)) al[o] = sl[(1 + 2 * o) * 64]; SB; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float r = 0.f;
  for (int b = 0; b < 8; ++b)
    for (int i = 0; i < 16; ++i) r += acc[b][i];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int M>
void run(float* out, const char* w, unsigned long long* cyc, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<M><<<256, 256, 65536>>>(out, w, iters, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 3; ++i) k<M><<<256, 256, 65536>>>(out, w, iters, cyc);
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
  printf("%-64s %7.3f ms  %6.1f cycles per 24 MFMAs (%4.1f per MFMA)  clock %.2f GHz  (%s)\n", name, ms, c / iters, c / (iters * 24.0), c / (ms * 1e6),
         hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out; char* w; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&w, 1024 * 4096 + 65536); hipMemset(w, 0, 1024 * 4096 + 65536);
  run<0>(out, w, cyc, "MFMAs only");
  run<3>(out, w, cyc, "reads (spread) + VALU");
  run<35>(out, w, cyc, "reads (bursts) + VALU");
  run<11>(out, w, cyc, "spread + VALU + DMA behind 1st half, waited 1 round later, no barrier");
  run<15>(out, w, cyc, "spread + VALU + barrier + DMA behind barrier, waited next round");
  run<15 + 64>(out, w, cyc, "... DMA waited two rounds later (vmcnt 8)");
  run<15 + 64 + 128>(out, w, cyc, "... DMA pieces behind bare MFMAs, two rounds ahead");
  run<15 + 16 + 256>(out, w, cyc, "... barrier every 2nd round, 8 pieces behind it, vmcnt(0)");
  run<47>(out, w, cyc, "bursts + VALU + barrier + DMA behind barrier, waited next round");
  run<47 + 64>(out, w, cyc, "bursts ... DMA waited two rounds later");
  run<47 + 16 + 256>(out, w, cyc, "bursts ... barrier every 2nd round, 8 pieces, vmcnt(0)");
  return 0;
}
