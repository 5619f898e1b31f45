// Fused per-ray NeRF evaluation on the bf16 matrix cores with fp32-accurate operand splitting ("bf16x3").
//
// Same computation and outputs as nerf_fwd.hip; what changes is
// the arithmetic of the layer products: every fp32 operand x is split into two bf16 values x = hi + lo (16 mantissa
// bits together) and each product is evaluated as  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi  with three
// v_mfma_f32_32x32x16_bf16 (fp32 accumulation).  The dropped lo*lo term is 2^-16 relative; measured against the fp32
// oracle the rendered features differ by < 1e-6 (tolerance 1e-4) -- see DESIGN.md section 3.1b.  Three bf16 MFMAs
// cost 3/16 of the fp32 MFMA they replace.
//
// Structure (one workgroup = 4 wavefronts = 128 samples, one wavefront per SIMD):
//   * activations stay in registers between layers, now as packed bf16 (hi, lo) B operands: the MFMA result layout
//     (lane = sample + 32*half, register r <-> neuron (r&3)+8*(r>>2)+4*half) maps 8 consecutive registers of a lane
//     onto the 8 K-slots of one 32x32x16 step, so re-packing is lane-local (relu, cvt, subtract, cvt);
//   * weights are pre-split and pre-ordered on the host into 16 KiB "slots" = one K-step for all 8 output blocks,
//     streamed by all 4 wavefronts with global_load_lds (LDS DMA, no VGPRs) into a 4-slot LDS ring, two slots ahead;
//     one s_barrier + one counted s_waitcnt vmcnt per slot; every wavefront then reads its A operands with
//     conflict-free ds_read_b128 (the ring layout is lane-linear, exactly what the DMA writes);
//   * the tapped activations (feature output) are kept in registers (AGPRs) until the compositing weights are known
//     and reduced over the 32 samples of a wavefront with cross-lane shuffles, so LDS is free for the weight ring;
//   * the integrated positional encoding is evaluated once per 128-sample chunk with an fp64 angle-doubling
//     recurrence (sin/cos(2^i x) from sin/cos(x)) and parked in LDS as ready-made B operands for layers 0 and 5.
#include "common.h"
#include <string.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 128;
constexpr int SLOT_BYTES = 16384;
constexpr int SLOT_FLOATS = SLOT_BYTES / 4;
constexpr int NRING = 4;
constexpr int WS_WORKGROUPS = 512;  // upper bound of the persistent grid (one workgroup per CU); sizes the workspace
constexpr int XS = 6;    // K-steps of the 90(->96)-d IPE input
constexpr int HS = 16;   // K-steps of a 256-d hidden input
constexpr int VS = 3;    // K-steps of the 43(->48)-d [direction PE | appearance] input of the views layer
constexpr int NSLOT_NORGB = XS + 4 * HS + (XS + HS) + 2 * HS;       // layers 0..7          = 124
constexpr int NSLOT_FULL = NSLOT_NORGB + HS + (HS + VS);             // + feature + views    = 159

// small-parameter block (fp32), same layout as nerf_fwd.hip
constexpr int OFF_BIAS = 0, OFF_BVIEWS = 2304, OFF_WALPHA = 2432, OFF_WRGB = 2688, OFF_MISC = 3072, SMALL = 3088;
constexpr int SMALL_PAD = 4096;  // floats reserved in the blob / LDS (16 KiB)
constexpr size_t BLOB_BYTES = (size_t)SMALL_PAD * 4 + (size_t)NSLOT_FULL * SLOT_BYTES;

// LDS map (floats)
constexpr int LDS_SMALL = 0;
constexpr int LDS_RING = SMALL_PAD;
constexpr int LDS_IPE = LDS_RING + NRING * SLOT_FLOATS;        // [4 waves][XS][2][64][4 floats]
constexpr int LDS_SCR = LDS_IPE + 4 * XS * 2 * 64 * 4;          // per-sample scratch, see below
constexpr int LDS_FEAT = LDS_SCR + TILE * 12 + 32;              // [4 waves][256] partial feature sums
constexpr int LDS_EX = LDS_FEAT + 4 * 256;                     // [4 ray slots][48] views-layer extra inputs
constexpr int LDS_TOTAL = LDS_EX + 4 * 48;

struct NerfArgs {
  const char* blob;
  const float* rays;
  const float* t;
  const float* app_row;
  float* weights;
  float* feat;
  float* pts;
  float* rgb;
  float* depth;
  float* acc;
  float* raw;
  float* sfeat;
  float* ws;  // [gridDim.x][4 wavefronts][32][64 lanes][4]: tapped activations of the tile in flight (fp32)
  int R, S, tap, white_bg, flags, ntiles;
  float var_scale;
};

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// timing-only ablation switches (results are wrong when set; used by scripts/ab_nerf.py to attribute time)
#ifndef NM_ABL
#define NM_ABL 0
#endif
// -DNM_TRACE: profiling build only -- the `raw` output becomes a [grid][32] table of s_memtime stamps of wavefront 0
#ifndef NM_TRACE
#define NM_TRACE 0
#endif

__host__ __device__ __forceinline__ constexpr int nrow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

#if NM_TRACE
#define TRACE(i)                                                                                               \
  do {                                                                                                         \
    if (a.raw && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.raw)[bid * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define TRACE(i) do { } while (0)
#endif

// x = hi + lo with hi, lo bf16 (round to nearest even): 16 bits of mantissa survive.
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// Explicit residency in the accumulator half of the register file: the tapped activations (128 dwords per lane) are
// written once and read once per chunk, so they are parked in AGPRs by hand -- left to the allocator they compete
// with the resident activations for the 256 architectural VGPRs and starve the A-operand staging of the MFMA loop.
__device__ __forceinline__ unsigned agpr_put(unsigned v) {
  unsigned a;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ unsigned agpr_get(unsigned a) {
  unsigned v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}

// LDS DMA of one 16 KiB weight slot: every wavefront moves 4 x 1 KiB (lane l: 16 bytes at chunk*1024 + 16*l).
// The 4 pieces share ONE global address and ONE M0 (LDS base) and differ only in the instruction's immediate offset,
// which the hardware adds on both sides -- measured 31 instead of 58 cycles of issue per piece beside the MFMAs.
__device__ __forceinline__ void dma_slot(const char* blob_slots, int g, float* ring, int wave, int lane) {
  const auto* src = (const __attribute__((address_space(1))) void*)(blob_slots + (size_t)g * SLOT_BYTES + wave * 4096 + lane * 16);
  auto* dst = (__attribute__((address_space(3))) void*)(ring + (g & (NRING - 1)) * SLOT_FLOATS + wave * 1024);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
}

// Ring protocol for slot g (identical sequence in all 4 wavefronts):
//   wait until this wavefront's DMA pieces of slot g have landed (at most the 4 instructions of slot g+1 may remain
//   in flight), barrier (=> every wavefront's pieces landed AND everybody finished reading slot g-1... g-2), then
//   start the DMA of slot g+2 into the ring position that slot g-2 occupied.
__device__ __forceinline__ void ring_acquire(const char* blob_slots, int g, int nslots, float* ring, int wave, int lane) {
  if (g + 1 < nslots) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!(NM_ABL & 8)) __builtin_amdgcn_s_barrier();
  if (g + 2 < nslots && !(NM_ABL & 16)) dma_slot(blob_slots, g + 2, ring, wave, lane);
}

// A operands of half a slot: 4 output blocks x (hi, lo) = 8 x 16 bytes per lane.
struct OpHalf {
  bf16x8 h[4], l[4];
};

__device__ __forceinline__ void load_half(OpHalf& d, const float* slot, int lane, int p) {
  const u32x4* s4 = reinterpret_cast<const u32x4*>(slot) + lane;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    d.h[o] = __builtin_bit_cast(bf16x8, s4[((4 * p + o) * 2 + 0) * 64]);
    d.l[o] = __builtin_bit_cast(bf16x8, s4[((4 * p + o) * 2 + 1) * 64]);
  }
}

// acc[4p .. 4p+3] += W_half . (xh + xl)  as  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi, issued in two parts (4 + 8 MFMAs)
template <int NOB>
__device__ __forceinline__ void mfma_head(f32x16 (&acc)[NOB], int p, const OpHalf& a, const bf16x8& xh) {
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[4 * p + o] = MFMA_BF16(a.h[o], xh, acc[4 * p + o]);
}
template <int NOB>
__device__ __forceinline__ void mfma_tail(f32x16 (&acc)[NOB], int p, const OpHalf& a, const bf16x8& xh, const bf16x8& xl) {
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[4 * p + o] = MFMA_BF16(a.h[o], xl, acc[4 * p + o]);
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[4 * p + o] = MFMA_BF16(a.l[o], xh, acc[4 * p + o]);
}

// One K-step (slot g) of an 8-block layer, software pipelined over half slots with a "consume first" order: every
// batch of LDS reads is issued right AFTER four MFMAs that use the previously fetched operands, so the wait the
// compiler places in front of those MFMAs only covers reads that were issued >= 8 MFMAs (256 cycles) earlier:
//   head(blocks 0-3, A) | fetch B = blocks 4-7 of slot g | tail(blocks 0-3, A)
//   ring barrier of slot g+1 (+ DMA of slot g+3)
//   head(blocks 4-7, B) | fetch A = blocks 0-3 of slot g+1 | tail(blocks 4-7, B)
// PREFETCH = false on the last slot of a layer part: the operands of the next slot are then fetched by the next part
// itself (one exposed LDS latency per part) instead of being kept live -- and spilled -- across the re-packing code.
template <bool PREFETCH>
__device__ __forceinline__ void slot_step8(f32x16 (&acc)[8], OpHalf& A, const char* blob_slots, int g, int nslots, float* ring,
                                           int wave, int lane, const bf16x8& xh, const bf16x8& xl) {
  OpHalf B;
  mfma_head<8>(acc, 0, A, xh);
  __builtin_amdgcn_sched_barrier(0);
  load_half(B, ring + (g & (NRING - 1)) * SLOT_FLOATS, lane, 1);
  __builtin_amdgcn_sched_barrier(0);
  mfma_tail<8>(acc, 0, A, xh, xl);
  __builtin_amdgcn_sched_barrier(0);
  if (g + 1 < nslots) ring_acquire(blob_slots, g + 1, nslots, ring, wave, lane);
  mfma_head<8>(acc, 1, B, xh);
  __builtin_amdgcn_sched_barrier(0);
  if (PREFETCH && g + 1 < nslots) load_half(A, ring + ((g + 1) & (NRING - 1)) * SLOT_FLOATS, lane, 0);
  __builtin_amdgcn_sched_barrier(0);
  mfma_tail<8>(acc, 1, B, xh, xl);
  __builtin_amdgcn_sched_barrier(0);
}

// Same for the 4-block views layer (a slot is a single half).
template <bool PREFETCH>
__device__ __forceinline__ void slot_step4(f32x16 (&acc)[4], OpHalf& A, const char* blob_slots, int g, int nslots, float* ring,
                                           int wave, int lane, const bf16x8& xh, const bf16x8& xl) {
  OpHalf C = A;
  if (g + 1 < nslots) ring_acquire(blob_slots, g + 1, nslots, ring, wave, lane);
  mfma_head<4>(acc, 0, C, xh);
  __builtin_amdgcn_sched_barrier(0);
  if (PREFETCH && g + 1 < nslots) load_half(A, ring + ((g + 1) & (NRING - 1)) * SLOT_FLOATS, lane, 0);
  __builtin_amdgcn_sched_barrier(0);
  mfma_tail<4>(acc, 0, C, xh, xl);
  __builtin_amdgcn_sched_barrier(0);
}

// relu / identity without the canonicalising v_max the compiler adds around fmaxf on MFMA results
__device__ __forceinline__ float vmax(float x, float floor_v) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(floor_v));
  return r;
}

// fp64 sin/cos of |x| <~ 1e3 (musl __sin / __cos kernels after a two-term Cody-Waite reduction)
__device__ __forceinline__ void sincos_f64(double x, double& s, double& c) {
  const double n = __builtin_rint(x * 0.63661977236758134308);
  double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
  r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
  const int q = (int)n;
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double sr = r + r * z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
  const double cr = 1.0 - 0.5 * z + z * z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double ss = (q & 1) ? cr : sr, cc = (q & 1) ? sr : cr;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

__global__ void __launch_bounds__(256, 1) nerf_fwd_bf16x3_kernel(NerfArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[LDS_TOTAL];
  float* const sm_small = sm + LDS_SMALL;
  float* const ring = sm + LDS_RING;
  float* const sm_ipe = sm + LDS_IPE;
  float* const sm_sigma = sm + LDS_SCR;       // [128]
  float* const sm_rgb = sm_sigma + TILE;      // [3][128]
  float* const sm_t0 = sm_rgb + 3 * TILE;
  float* const sm_t1 = sm_t0 + TILE;
  float* const sm_mean = sm_t1 + TILE;        // [3][128]
  float* const sm_dn = sm_mean + 3 * TILE;
  float* const sm_w = sm_dn + TILE;
  float* const sm_misc = sm_w + TILE;         // [32]
  float* const sm_feat = sm + LDS_FEAT;       // [4][256]
  float* const sm_part = sm_feat;             // [4 half wavefronts][8] partial per-ray sums: written and read by wavefront 0/1
                                              // before wavefront 0 stores its feature partials over them (program order)
  float* const sm_ex = sm + LDS_EX;           // [nr][48]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, hi = lane >> 5;
  const int S = a.S, R = a.R;
  const int SP = S < TILE ? S : TILE;
  const int nr = TILE / SP;
  const int nchunks = (S + TILE - 1) / TILE;
  const bool need_rgb = !(a.flags & NM_NERF_SKIP_RGB);
  const bool feat_max = (a.flags & NM_NERF_FEAT_MAX) != 0;
  const bool need_tap = (a.feat != nullptr) || (a.sfeat != nullptr);
  const int tap = (a.tap < 0 || a.tap > 7) ? 7 : a.tap;
  const int nslots = need_rgb ? NSLOT_FULL : NSLOT_NORGB;
  const char* const blob_slots = a.blob + (size_t)SMALL_PAD * 4;

  for (int i = tid; i < SMALL / 4; i += 256) reinterpret_cast<f32x4*>(sm_small)[i] = reinterpret_cast<const f32x4*>(a.blob)[i];

  // persistent workgroups: one per CU (the LDS footprint allows no more), tiles dealt round robin
#pragma unroll 1
  for (int bid = blockIdx.x; bid < a.ntiles; bid += gridDim.x) {
  TRACE(0);
  // extra inputs of the views layer, one value per thread (they depend on the ray only):
  // f = 0..11 sin(2^k d), 12..23 sin(2^k d + pi/2), 24..26 raw d, 27..42 appearance, 43..47 padding
  if (need_rgb && tid < nr * 48) {
    const int r2 = tid / 48, f = tid % 48;
    const int ray2 = bid * nr + r2;
    const float* rq = a.rays + (size_t)(ray2 < R ? ray2 : R - 1) * 12 + 8;
    float v = 0.f;
    if (f < 24) {
      const int k = (f % 12) / 3;
      const float xe = rq[f % 3] * (float)(1 << k);
      v = nm_sinf(f < 12 ? xe : xe + 1.57079637050628662109375f);
    } else if (f < 27) {
      v = rq[f - 24];
    } else if (f < 43) {
      v = a.app_row ? a.app_row[f - 27] : 0.f;
    }
    sm_ex[tid] = v;
  }
  __syncthreads();  // biases are read before the first ring barrier
  TRACE(1);

  const int js = wave * 32 + s;
  const int rl = js / SP;
  const int ray = bid * nr + rl;
  const int rc = ray < R ? ray : R - 1;
  const float* rp = a.rays + (size_t)rc * 12;
  const float o0 = rp[0], o1 = rp[1], o2 = rp[2], d0 = rp[3], d1 = rp[4], d2 = rp[5], radius = rp[11];
  const float dsq0 = d0 * d0, dsq1 = d1 * d1, dsq2 = d2 * d2;
  const float dmag = fmaxf(1e-10f, (dsq0 + dsq1) + dsq2);
  const float dnorm = sqrtf((dsq0 + dsq1) + dsq2);
  const float nul0 = 1.0f - dsq0 / dmag, nul1 = 1.0f - dsq1 / dmag, nul2 = 1.0f - dsq2 / dmag;

  float red_acc = 0.f;
  float carryT = 1.f;
  float best_w = -1.f;
  float feat_run = 0.f;  // thread t: running feature channel t of the (single) ray when S > 128

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int sidx = chunk * TILE + (js % SP);
    const float t0 = a.t[(size_t)rc * (S + 1) + sidx];
    const float t1 = a.t[(size_t)rc * (S + 1) + sidx + 1];
    const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
    const float denom = fmaxf(1.1920928955078125e-07f, 3.0f * mu2 + hw2);
    const float t_mean = mu + (2.0f * mu * hw2) / denom;
    const float t_var = hw2 / 3.0f - (float)(4.0 / 15.0) * ((hw4 * (12.0f * mu2 - hw2)) / (denom * denom));
    const float r_var = (radius * radius) * ((mu2 / 4.0f + (float)(5.0 / 12.0) * hw2) - (float)(4.0 / 15.0) * hw4 / denom);
    float mean[3] = {d0 * t_mean + o0, d1 * t_mean + o1, d2 * t_mean + o2};
    float var[3] = {t_var * dsq0 + r_var * nul0, t_var * dsq1 + r_var * nul1, t_var * dsq2 + r_var * nul2};
    if (a.var_scale > 0.f) {
      var[0] *= a.var_scale; var[1] *= a.var_scale; var[2] *= a.var_scale;
    }
    if (hi == 0) {
      sm_t0[js] = t0; sm_t1[js] = t1;
      sm_mean[js] = mean[0]; sm_mean[TILE + js] = mean[1]; sm_mean[2 * TILE + js] = mean[2];
      sm_dn[js] = dnorm;
    }

    // start the weight stream: slots 0 and 1
    dma_slot(blob_slots, 0, ring, wave, lane);
    dma_slot(blob_slots, 1, ring, wave, lane);

    // ---- integrated positional encoding -> B operands of the 6 IPE K-steps, parked in LDS -------------------------
    // K-slot (step m, half h, i) <-> encoding index f = 16 m + 8 h + i in the reference's order
    // f = part*45 + scale*3 + axis  (part 0: sin(2^scale x), part 1: sin(2^scale x + pi/2)); f >= 90 is padding.
    {
      float ipe[2][15][3];  // [part][scale][axis] for THIS lane's sample
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        double sd, cd;
        if (NM_ABL & 2) { sd = mean[ax]; cd = var[ax]; }
        else sincos_f64((double)mean[ax], sd, cd);
#pragma unroll
        for (int i = 0; i < 15; ++i) {
          const float sc = (float)(1 << i);
          const float xe = mean[ax] * sc;
          const float damp = (NM_ABL & 2) ? var[ax] * sc : expf(-0.5f * (var[ax] * (sc * sc)));
          // reference: sin(fl32(xe + fl32(pi/2))): the rounded sum deviates from xe + pi/2 by eps
          const float argc = xe + 1.57079637050628662109375f;
          const double eps = ((double)argc - (double)xe) - 1.57079632679489661923;
          const double e2 = eps * eps;
          const double ce = 1.0 - 0.5 * e2 + e2 * e2 * (1.0 / 24.0);
          const double se = eps * (1.0 - e2 * (1.0 / 6.0) + e2 * e2 * (1.0 / 120.0));
          ipe[0][i][ax] = damp * (float)sd;
          ipe[1][i][ax] = damp * (float)(cd * ce - sd * se);
          const double s2 = 2.0 * sd * cd, c2 = 1.0 - 2.0 * sd * sd;  // angle doubling
          sd = s2;
          cd = c2;
        }
      }
      float* dst = sm_ipe + wave * (XS * 2 * 64 * 4) + lane * 4;
#pragma unroll
      for (int m = 0; m < XS; ++m) {
        float v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int f0 = 16 * m + i, f1 = 16 * m + 8 + i;  // half 0 / half 1 candidates (compile time)
          const float v0 = f0 < 90 ? ipe[f0 / 45][(f0 % 45) / 3][f0 % 3] : 0.f;
          const float v1 = f1 < 90 ? ipe[f1 / 45][(f1 % 45) / 3][f1 % 3] : 0.f;
          v8[i] = hi ? v1 : v0;
        }
        bf16x8 h8, l8;
        split8(v8, h8, l8);
        *reinterpret_cast<u32x4*>(dst + (m * 2 + 0) * 256) = __builtin_bit_cast(u32x4, h8);
        *reinterpret_cast<u32x4*>(dst + (m * 2 + 1) * 256) = __builtin_bit_cast(u32x4, l8);
      }
    }

    TRACE(2);
    // ---- 8 pts layers + feature_linear --------------------------------------------------------------------------
    bf16x8 xh[HS], xl[HS];     // resident activations as B operands: K-step ks = 2*block + half-of-block
    f32x4* const tapw = reinterpret_cast<f32x4*>(a.ws) + ((size_t)blockIdx.x * 4 + wave) * 32 * 64 + lane;
    float sig_part = 0.f;
    int g = 0;                 // slot counter of this chunk
    OpHalf opA;                // operands of the next half slot, fetched one half slot ahead
    ring_acquire(blob_slots, 0, nslots, ring, wave, lane);
    const float* ipe_src = sm_ipe + wave * (XS * 2 * 64 * 4) + lane * 4;
#pragma unroll 1
    for (int l = 0; l < 9; ++l) {
      if (l == 8 && !need_rgb) break;
      f32x16 acc[8];
      const float* bl = sm_small + OFF_BIAS + l * 256 + 4 * hi;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q);
          acc[ob][4 * q + 0] = b[0]; acc[ob][4 * q + 1] = b[1]; acc[ob][4 * q + 2] = b[2]; acc[ob][4 * q + 3] = b[3];
        }
      if (l == 0 || l == 5) {
        load_half(opA, ring + (g & (NRING - 1)) * SLOT_FLOATS, lane, 0);
#pragma unroll
        for (int m = 0; m < XS; ++m) {
          const bf16x8 ph = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ipe_src + (m * 2 + 0) * 256));
          const bf16x8 pl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ipe_src + (m * 2 + 1) * 256));
          if (m + 1 < XS) slot_step8<true>(acc, opA, blob_slots, g, nslots, ring, wave, lane, ph, pl);
          else slot_step8<false>(acc, opA, blob_slots, g, nslots, ring, wave, lane, ph, pl);
          ++g;
        }
      }
      if (l != 0) {
        load_half(opA, ring + (g & (NRING - 1)) * SLOT_FLOATS, lane, 0);
#pragma unroll
        for (int ks = 0; ks < HS; ++ks) {
          if (ks + 1 < HS) slot_step8<true>(acc, opA, blob_slots, g, nslots, ring, wave, lane, xh[ks], xl[ks]);
          else slot_step8<false>(acc, opA, blob_slots, g, nslots, ring, wave, lane, xh[ks], xl[ks]);
          ++g;
        }
      }
      if (l == 7) {
        const float* wa = sm_small + OFF_WALPHA + 4 * hi;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w4v = *reinterpret_cast<const f32x4*>(wa + ob * 32 + 8 * q);
            p0 = NM_FMA(vmax(acc[ob][4 * q + 0], 0.f), w4v[0], p0);
            p1 = NM_FMA(vmax(acc[ob][4 * q + 1], 0.f), w4v[1], p1);
            p2 = NM_FMA(vmax(acc[ob][4 * q + 2], 0.f), w4v[2], p2);
            p3 = NM_FMA(vmax(acc[ob][4 * q + 3], 0.f), w4v[3], p3);
          }
        sig_part = (p0 + p1) + (p2 + p3);
      }
      // relu (none after feature_linear) and re-pack as the next layer's B operands
      const float floor_v = (l < 8) ? 0.f : -__builtin_inff();
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float v8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = vmax(acc[ob][8 * m + i], floor_v);
          if (l == tap && need_tap) {  // tapped activations (fp32, after the relu) -> L2-resident workspace, 1 KiB per store
            tapw[(4 * ob + 2 * m) * 64] = f32x4{v8[0], v8[1], v8[2], v8[3]};
            tapw[(4 * ob + 2 * m + 1) * 64] = f32x4{v8[4], v8[5], v8[6], v8[7]};
          }
          if (NM_ABL & 1) {
            if (l == 0) split8(v8, xh[2 * ob + m], xl[2 * ob + m]);
            else asm volatile("" ::"v"(v8[0]));
          } else {
            split8(v8, xh[2 * ob + m], xl[2 * ob + m]);
          }
        }
      TRACE(3 + l);
    }
    const float sigma_raw = (sig_part + nm_shfl_xor32(sig_part)) + sm_small[OFF_MISC];

    // ---- views layer + rgb head -------------------------------------------------------------------------------------
    float c_r = 0.f, c_g = 0.f, c_b = 0.f;
    if (need_rgb) {
      const int hh = launder(lane) >> 5;
      const float* exr = sm_ex + launder(rl) * 48 + 8 * hh;  // K-slot (step e, half h, i) <-> extra input 16 e + 8 h + i
      f32x16 av[4];
      const float* bv = sm_small + OFF_BVIEWS + 4 * hh;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bv + ob * 32 + 8 * q);
          av[ob][4 * q + 0] = b[0]; av[ob][4 * q + 1] = b[1]; av[ob][4 * q + 2] = b[2]; av[ob][4 * q + 3] = b[3];
        }
      load_half(opA, ring + (g & (NRING - 1)) * SLOT_FLOATS, lane, 0);
#pragma unroll
      for (int ks = 0; ks < HS; ++ks) {
        if (ks + 1 < HS) slot_step4<true>(av, opA, blob_slots, g, nslots, ring, wave, lane, xh[ks], xl[ks]);
        else slot_step4<false>(av, opA, blob_slots, g, nslots, ring, wave, lane, xh[ks], xl[ks]);
        ++g;
      }
#pragma unroll
      for (int e = 0; e < VS; ++e) {
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(exr + 16 * e), e1 = *reinterpret_cast<const f32x4*>(exr + 16 * e + 4);
        const float v8[8] = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
        bf16x8 eh, el;
        split8(v8, eh, el);
        if (e == 0) load_half(opA, ring + (g & (NRING - 1)) * SLOT_FLOATS, lane, 0);
        if (e + 1 < VS) slot_step4<true>(av, opA, blob_slots, g, nslots, ring, wave, lane, eh, el);
        else slot_step4<false>(av, opA, blob_slots, g, nslots, ring, wave, lane, eh, el);
        ++g;
      }
      const float* wr = sm_small + OFF_WRGB + 4 * hh;
      float pr = 0.f, pg = 0.f, pb = 0.f;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wr + ob * 32 + 8 * q);
          const f32x4 wg4 = *reinterpret_cast<const f32x4*>(wr + 128 + ob * 32 + 8 * q);
          const f32x4 wb4 = *reinterpret_cast<const f32x4*>(wr + 256 + ob * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hv = vmax(av[ob][4 * q + e], 0.f);
            pr = NM_FMA(hv, wr4[e], pr);
            pg = NM_FMA(hv, wg4[e], pg);
            pb = NM_FMA(hv, wb4[e], pb);
          }
        }
      pr = (pr + nm_shfl_xor32(pr)) + sm_small[OFF_MISC + 1];
      pg = (pg + nm_shfl_xor32(pg)) + sm_small[OFF_MISC + 2];
      pb = (pb + nm_shfl_xor32(pb)) + sm_small[OFF_MISC + 3];
      c_r = 1.0f / (1.0f + expf(-pr));
      c_g = 1.0f / (1.0f + expf(-pg));
      c_b = 1.0f / (1.0f + expf(-pb));
    }
    TRACE(12);
    {
      const int jsw = launder(js);
      if ((launder(lane) >> 5) == 0) {
        sm_sigma[jsw] = sigma_raw;
        sm_rgb[jsw] = c_r; sm_rgb[TILE + jsw] = c_g; sm_rgb[2 * TILE + jsw] = c_b;
      }
    }
    __syncthreads();
    TRACE(13);

    // ---- alpha compositing (identical to nerf_fwd.hip) ---------------------------------------------------------------
    float alpha = 0.f, incl = 1.f;
    const int tid2 = launder(threadIdx.x), lane2 = tid2 & 63, wave2 = tid2 >> 6;
    if (tid2 < TILE) {
      const float sg = fmaxf(sm_sigma[tid2], 0.f);
      const float delta = (sm_t1[tid2] - sm_t0[tid2]) * sm_dn[tid2];
      alpha = 1.0f - expf(-sg * delta);
      incl = (1.0f - alpha) + 1e-10f;
      const int seg = SP < 64 ? SP : 64;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const float up = __shfl_up(incl, dlt, 64);
        if (dlt < seg && (lane2 & (seg - 1)) >= dlt) incl *= up;
      }
      if (lane2 == 63) sm_misc[wave2] = incl;
    }
    __syncthreads();
    if (tid2 < TILE) {
      const int seg = SP < 64 ? SP : 64;
      float excl = __shfl_up(incl, 1, 64);
      if ((lane2 & (seg - 1)) == 0) excl = 1.f;
      if (SP == TILE && wave2 == 1) excl *= sm_misc[0];
      excl *= carryT;
      const float wgt = alpha * excl;
      sm_w[tid2] = wgt;
      const int r2 = tid2 / SP, ray2 = bid * nr + r2;
      if (ray2 < R) {
        const int s2 = chunk * TILE + tid2 % SP;
        a.weights[(size_t)ray2 * S + s2] = wgt;
        if (a.raw && !NM_TRACE) {
          f32x4 rv = {sm_rgb[tid2], sm_rgb[TILE + tid2], sm_rgb[2 * TILE + tid2], sm_sigma[tid2]};
          *reinterpret_cast<f32x4*>(a.raw + ((size_t)ray2 * S + s2) * 4) = rv;
        }
      }
      // per-ray sums, step 1: w * {1, rgb, t_mid, mean} reduced over each 32-sample half wavefront
      float pq[8] = {wgt, wgt * sm_rgb[tid2], wgt * sm_rgb[TILE + tid2], wgt * sm_rgb[2 * TILE + tid2],
                     wgt * (0.5f * (sm_t0[tid2] + sm_t1[tid2])), wgt * sm_mean[tid2], wgt * sm_mean[TILE + tid2],
                     wgt * sm_mean[2 * TILE + tid2]};
      nm_half_sum_dpp8(pq);  // valid in lanes 16..31 / 48..63
      if ((tid2 & 31) == 16) {
        *reinterpret_cast<f32x4*>(sm_part + (tid2 >> 5) * 8) = f32x4{pq[0], pq[1], pq[2], pq[3]};
        *reinterpret_cast<f32x4*>(sm_part + (tid2 >> 5) * 8 + 4) = f32x4{pq[4], pq[5], pq[6], pq[7]};
      }
    }
    if (nchunks > 1) carryT = carryT * (sm_misc[0] * sm_misc[1]);
    __syncthreads();

    // ---- per-ray sums, step 2: combine the SP/32 half wavefronts of each ray ------------------------------------------
    if (tid2 < 8 * nr) {
      const int q = tid2 & 7, r2 = tid2 >> 3;
      const float* wv = sm_w + r2 * SP;
      if (!feat_max || q < 5) {
        float sum = 0.f;
        for (int hw = r2 * (SP / 32); hw < (r2 + 1) * (SP / 32); ++hw) sum += sm_part[hw * 8 + q];
        red_acc += sum;
      }
      if (feat_max) {
        float bw = wv[0];
        int bi = 0;
        for (int k = 1; k < SP; ++k)
          if (wv[k] > bw) { bw = wv[k]; bi = k; }
        const bool better = bw > best_w;
        if (better) best_w = bw;
        if (q == 0) sm_misc[8 + r2] = better ? __int_as_float(r2 * SP + bi) : __int_as_float(-1);
        if (q >= 5 && better) red_acc = sm_mean[(q - 5) * TILE + r2 * SP + bi];
      }
    }
    if (feat_max) __syncthreads();

    TRACE(14);
    // ---- feature output: weighted sum over the 32 samples of this wavefront straight from registers ------------------
    if (need_tap) {
      const int jl = launder(js), hl = launder(lane) >> 5;
      f32x4 tapv[2 * HS];
      {
        const f32x4* tw = reinterpret_cast<const f32x4*>(a.ws) + ((size_t)blockIdx.x * 4 + (launder(threadIdx.x) >> 6)) * 32 * 64 + (launder(threadIdx.x) & 63);
#pragma unroll
        for (int c = 0; c < 2 * HS; ++c) tapv[c] = tw[c * 64];
      }
      const float wj = sm_w[jl];
      const int rsel = jl / SP;                                   // ray slot of this lane's sample
      const int best = feat_max ? __float_as_int(sm_misc[8 + rsel]) : -2;
      float* prow = sm_feat + (jl >> 5) * 256 + 4 * hl;           // partial sums of this wavefront
#pragma unroll
      for (int ks = 0; ks < HS; ++ks) {
        const f32x4 ta = tapv[2 * ks], tb = tapv[2 * ks + 1];
        float v8[8] = {ta[0], ta[1], ta[2], ta[3], tb[0], tb[1], tb[2], tb[3]};
        if (a.sfeat && ray < R) {
          float* dsf = a.sfeat + ((size_t)ray * S + sidx) * 256 + (ks >> 1) * 32 + 16 * (ks & 1) + 4 * hl;
          *reinterpret_cast<f32x4*>(dsf) = f32x4{v8[0], v8[1], v8[2], v8[3]};
          *reinterpret_cast<f32x4*>(dsf + 8) = f32x4{v8[4], v8[5], v8[6], v8[7]};
        }
        if (a.feat) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = feat_max ? (jl == best ? v8[i] : 0.f) : wj * v8[i];
          if (!(NM_ABL & 4)) nm_half_sum_dpp8(v8);  // 32-sample sums, valid in lanes 16..31 / 48..63
          if ((jl & 31) == 16) {
            // registers 8m+i of block ob <-> neurons 32 ob + nrow(8m+i, h): i = 0..3 -> +0..3, i = 4..7 -> +8..11 (plus 16 m)
            float* d = prow + (ks >> 1) * 32 + 16 * (ks & 1);
            *reinterpret_cast<f32x4*>(d) = f32x4{v8[0], v8[1], v8[2], v8[3]};
            *reinterpret_cast<f32x4*>(d + 8) = f32x4{v8[4], v8[5], v8[6], v8[7]};
          }
        }
      }
    }
    TRACE(15);
    __syncthreads();
    TRACE(16);
    if (a.feat) {
      // combine the wavefronts of each ray: SP samples = SP/32 wavefronts
      const int wpr = SP / 32;  // wavefronts per ray slot (1, 2 or 4)
      for (int r2 = 0; r2 < nr; ++r2) {
        float f = 0.f;
        bool any = !feat_max;
        if (feat_max) {
          const int best = __float_as_int(sm_misc[8 + r2]);
          any = best >= 0;
        }
        for (int w2 = 0; w2 < wpr; ++w2) f += sm_feat[(r2 * wpr + w2) * 256 + tid2];
        const int ray2 = bid * nr + r2;
        if (nchunks > 1) {
          if (feat_max) { if (any) feat_run = f; }
          else feat_run += f;
          f = feat_run;
        }
        if (ray2 < R && chunk == nchunks - 1 && (any || nchunks > 1)) a.feat[(size_t)ray2 * 256 + tid2] = f;
      }
    }
    __syncthreads();
  }

  if (tid < 8 * nr) {
    const int q = tid & 7, r2 = tid >> 3, ray2 = bid * nr + r2;
    const float accv = __shfl(red_acc, lane & ~7, 64);
    if (ray2 < R) {
      if (q == 0) { if (a.acc) a.acc[ray2] = red_acc; }
      else if (q <= 3) { if (a.rgb && need_rgb) a.rgb[(size_t)ray2 * 3 + (q - 1)] = a.white_bg ? red_acc + (1.0f - accv) : red_acc; }
      else if (q == 4) { if (a.depth) a.depth[ray2] = red_acc; }
      else { if (a.pts) a.pts[(size_t)ray2 * 3 + (q - 5)] = red_acc; }
    }
  }
  TRACE(17);
  }  // tile loop
}

// ---- host-side packing ------------------------------------------------------------------------------------------------
inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float bf16_to_f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// one slot: element (obo, hl, lane, i) = split(W[32*obo + (lane&31)][col(lane>>5, i)]); col < 0 -> 0
template <typename ColFn>
void pack_slot(uint16_t* slot, const float* W, int ld, int nob, ColFn col) {
  for (int obo = 0; obo < nob; ++obo)
    for (int ln = 0; ln < 64; ++ln)
      for (int i = 0; i < 8; ++i) {
        const int c = col(ln >> 5, i);
        const float w = c < 0 ? 0.f : W[(size_t)(32 * obo + (ln & 31)) * ld + c];
        const uint16_t h = bf16_rne(w);
        const uint16_t l = bf16_rne(w - bf16_to_f(h));
        slot[((obo * 2 + 0) * 64 + ln) * 8 + i] = h;
        slot[((obo * 2 + 1) * 64 + ln) * 8 + i] = l;
      }
}

}  // namespace

extern "C" size_t nm_nerf_blob_bytes_bf16x3(void) { return BLOB_BYTES; }
extern "C" size_t nm_nerf_workspace_bytes_bf16x3(void) { return (size_t)WS_WORKGROUPS * TILE * 256 * sizeof(float); }

extern "C" int nm_nerf_pack_bf16x3(const nmNerfWeights* w, void* blob_v) {
  if (!w || !blob_v) return NM_ERR_ARG;
  for (int i = 0; i < 8; ++i)
    if (!w->pts_w[i] || !w->pts_b[i]) return NM_ERR_ARG;
  if (!w->alpha_w || !w->alpha_b || !w->feat_w || !w->feat_b || !w->views_w || !w->views_b || !w->rgb_w || !w->rgb_b)
    return NM_ERR_ARG;
  if (w->app_dim != 0 && w->app_dim != 16) return NM_ERR_UNSUPPORTED;
  memset(blob_v, 0, BLOB_BYTES);
  float* small = (float*)blob_v;
  for (int l = 0; l < 8; ++l)
    for (int n = 0; n < 256; ++n) small[OFF_BIAS + l * 256 + n] = w->pts_b[l][n];
  for (int n = 0; n < 256; ++n) small[OFF_BIAS + 8 * 256 + n] = w->feat_b[n];
  for (int n = 0; n < 128; ++n) small[OFF_BVIEWS + n] = w->views_b[n];
  for (int n = 0; n < 256; ++n) small[OFF_WALPHA + n] = w->alpha_w[n];
  for (int n = 0; n < 384; ++n) small[OFF_WRGB + n] = w->rgb_w[n];
  small[OFF_MISC] = w->alpha_b[0];
  for (int c = 0; c < 3; ++c) small[OFF_MISC + 1 + c] = w->rgb_b[c];

  uint16_t* slots = (uint16_t*)((char*)blob_v + (size_t)SMALL_PAD * 4);
  int g = 0;
  auto next = [&]() { return slots + (size_t)(g++) * (SLOT_BYTES / 2); };
  auto ipe_steps = [&](const float* W, int ld) {
    for (int m = 0; m < XS; ++m)
      pack_slot(next(), W, ld, 8, [&](int h, int i) { const int f = 16 * m + 8 * h + i; return f < 90 ? f : -1; });
  };
  auto hid_steps = [&](const float* W, int ld, int col0, int nob) {
    for (int ks = 0; ks < HS; ++ks)
      pack_slot(next(), W, ld, nob, [&](int h, int i) { return col0 + 32 * (ks >> 1) + nrow(8 * (ks & 1) + i, h); });
  };
  for (int l = 0; l < 8; ++l) {
    if (l == 0) ipe_steps(w->pts_w[0], 90);
    if (l == 5) ipe_steps(w->pts_w[5], 346);
    if (l != 0) hid_steps(w->pts_w[l], l == 5 ? 346 : 256, l == 5 ? 90 : 0, 8);
  }
  hid_steps(w->feat_w, 256, 0, 8);
  const int ldv = 283 + w->app_dim;
  hid_steps(w->views_w, ldv, 0, 4);
  for (int e = 0; e < VS; ++e)
    pack_slot(next(), w->views_w, ldv, 4, [&](int h, int i) {
      const int f = 16 * e + 8 * h + i;
      if (f < 27) return 256 + f;
      if (f < 43 && w->app_dim) return 283 + (f - 27);
      return -1;
    });
  return g == NSLOT_FULL ? NM_OK : NM_ERR_ARG;
}

extern "C" int nm_nerf_fwd_bf16x3(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                                  int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                                  float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                                  nmStream_t stream) {
  NM_CHECK_ARG(blob && rays && t && weights && R > 0 && S > 0);
  if (!(S == 32 || S == 64 || (S % 128) == 0)) return NM_ERR_UNSUPPORTED;
  if (tap_layer > 7) return NM_ERR_ARG;
  if ((feat || sample_feat) && !workspace) return NM_ERR_WORKSPACE;
  NerfArgs a;
  a.ws = (float*)workspace;
  a.blob = (const char*)blob; a.rays = rays; a.t = t; a.app_row = app_row;
  a.weights = weights; a.feat = feat; a.pts = pts; a.rgb = rgb; a.depth = depth; a.acc = acc; a.raw = raw; a.sfeat = sample_feat;
  a.R = R; a.S = S; a.tap = tap_layer; a.white_bg = white_bg; a.flags = flags; a.var_scale = var_scale;
  const int SP = S < TILE ? S : TILE, nr = TILE / SP;
  a.ntiles = (R + nr - 1) / nr;
  const int ncu = nm_cu_count();
  const int grid = a.ntiles < ncu ? a.ntiles : (ncu < WS_WORKGROUPS ? ncu : WS_WORKGROUPS);
  nerf_fwd_bf16x3_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}
