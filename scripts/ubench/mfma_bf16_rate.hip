// Microbenchmark: sustained v_mfma_f32_32x32x16_bf16 rate of ONE wavefront per SIMD under the access patterns of
// nerf_fwd_bf16.hip.  Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_rate.hip -o mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// MODE 0: 8 accumulators round robin, register operands        (pure issue rate)
// MODE 1: 4 accumulators, each touched 3x in a row of 12        (the kernel's dependency pattern: distance 4)
// MODE 2: MODE 1 + 16 ds_read_b128 per 24 MFMAs (operands from LDS, consume-first order)
// MODE 3: MODE 2 + per-slot s_barrier
// MODE 4: MODE 3 + LDS DMA of 16 KiB per slot (global_load_lds) into a 4-slot ring
// MODE 5: MODE 4 with ONE address / M0 per slot and the 4 pieces told apart by the instruction's immediate offset
// MODE 7: MODE 5 with the 4 pieces spread over the slot (one piece after every 4th..6th MFMA)
// MODE 6: MODE 5 through buffer_load ... lds (SGPR descriptor + scalar slot offset + one 32-bit lane offset)
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const char* __restrict__ w, float* out, int nslots) {
  __shared__ __attribute__((aligned(16))) float ring[4 * 4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 4096; i += 256) ring[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc[8];
  for (int o = 0; o < 8; ++o)
    for (int i = 0; i < 16; ++i) acc[o][i] = 0.f;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 64 * 16384 + 65536, 0x00020000);
  bf16x8 xh, xl, a0[8];
  for (int i = 0; i < 8; ++i) { xh[i] = (__bf16)(0.5f + lane * 0.01f); xl[i] = (__bf16)(0.001f * i); }
  for (int o = 0; o < 8; ++o) a0[o] = xh;
  for (int g = 0; g < nslots; ++g) {
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ring + (g & 3) * 4096) + lane;
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = MFMA(a0[o], r == 1 ? xl : xh, acc[o]);
    } else if (MODE == 1) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[4 * p + o] = MFMA(a0[4 * p + o], r == 1 ? xl : xh, acc[4 * p + o]);
    } else {
      bf16x8 ah[4], al[4], bh[4], bl[4];
      const char* src7 = w + (size_t)((g + 2) % 64) * 16384 + wave * 4096 + lane * 16;
      float* dst7 = ring + ((g + 2) & 3) * 4096 + wave * 1024;
#define PIECE(off) if (MODE == 7 && g + 2 < nslots) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src7, (__attribute__((address_space(3))) void*)dst7, 16, off, 0); __builtin_amdgcn_sched_barrier(0); }
      if (MODE == 5 && g + 2 < nslots) {
        const char* src = w + (size_t)((g + 2) % 64) * 16384 + wave * 4096 + lane * 16;
        float* dst = ring + ((g + 2) & 3) * 4096 + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 3072, 0);
      } else if (MODE == 6 && g + 2 < nslots) {
        float* dst = ring + ((g + 2) & 3) * 4096 + wave * 1024;
        const int soff = ((g + 2) % 64) * 16384 + wave * 4096;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, soff, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, soff, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, soff, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, soff, 3072, 0);
      } else if (MODE == 4 && g + 2 < nslots) {
        const char* src = w + (size_t)((g + 2) % 64) * 16384 + wave * 4096 + lane * 16;
        float* dst = ring + ((g + 2) & 3) * 4096 + wave * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                           (__attribute__((address_space(3))) void*)(dst + q * 256), 16, 0, 0);
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) { ah[o] = __builtin_bit_cast(bf16x8, s4[(o * 2) * 64]); al[o] = __builtin_bit_cast(bf16x8, s4[(o * 2 + 1) * 64]); }
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] = MFMA(ah[o], xh, acc[o]);
      __builtin_amdgcn_sched_barrier(0);
      PIECE(0)
#pragma unroll
      for (int o = 0; o < 4; ++o) { bh[o] = __builtin_bit_cast(bf16x8, s4[((4 + o) * 2) * 64]); bl[o] = __builtin_bit_cast(bf16x8, s4[((4 + o) * 2 + 1) * 64]); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] = MFMA(ah[o], xl, acc[o]);
      PIECE(1024)
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] = MFMA(al[o], xh, acc[o]);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 3) {
        if (MODE == 7) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (MODE >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[4 + o] = MFMA(bh[o], xh, acc[4 + o]);
      PIECE(2048)
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[4 + o] = MFMA(bh[o], xl, acc[4 + o]);
      PIECE(3072)
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[4 + o] = MFMA(bl[o], xh, acc[4 + o]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int o = 0; o < 8; ++o)
    for (int i = 0; i < 16; ++i) s += acc[o][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* w, float* out, int nslots, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 256>>>(w, out, nslots);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<MODE><<<256, 256>>>(w, out, nslots);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double mfma = 256.0 * 4 * nslots * 24;
  printf("%-46s %8.3f ms  %7.1f TFLOP/s bf16  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", name, ms, mfma * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 / (nslots * 24.0));
}

int main() {
  char* w; float* out;
  hipMalloc(&w, 64 * 16384 + 65536); hipMemset(w, 0, 64 * 16384 + 65536);
  hipMalloc(&out, 256 * 256 * 4);
  const int n = 20000;
  run<0>(w, out, n, "0: 8 accumulators, register operands");
  run<1>(w, out, n, "1: 4+4 accumulators, distance-4 dependency");
  run<2>(w, out, n, "2: + 16 ds_read_b128 per slot");
  run<3>(w, out, n, "3: + s_barrier per slot");
  run<4>(w, out, n, "4: + 16 KiB LDS-DMA per slot");
  run<5>(w, out, n, "5: DMA, one address + immediate offsets");
  run<6>(w, out, n, "6: DMA via buffer_load lds");
  run<7>(w, out, n, "7: DMA imm offsets, pieces spread over slot");
  return 0;
}
