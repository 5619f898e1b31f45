// Second generation of the split-bf16 fused NeRF pass: TWO wavefronts per SIMD.
//
// Same computation, blob, workspace and outputs as nerf_fwd_bf16.hip (every
// product = w_hi*x_hi + w_hi*x_lo + w_lo*x_hi on v_mfma_f32_32x32x16_bf16).  What changes is the mapping:
//
//   * the first generation runs ONE wavefront per SIMD with the whole 512-register file (32 samples x 256 neurons per
//     wavefront: 128 accumulators + the finished layer in 128 VGPRs).  With a single in-order instruction stream per SIMD
//     every LDS wait, barrier, DMA issue and re-packing instruction that is not perfectly placed leaves the matrix pipe idle
//     (measured 46 cycles per MFMA against the 32-cycle floor) and the layer hand-over / heads / reductions run with the
//     pipe empty (a quarter of a tile).
//   * here a workgroup is 8 wavefronts; a PAIR of wavefronts shares 32 samples and each computes HALF of the neurons
//     (4 output blocks of 32 -> 64 accumulator registers).  Two accumulator sets (128 AGPRs) ping-pong between consecutive
//     layers, so a finished layer is never copied: its K-step "units" (bias, relu, hi/lo split of 8 values) are read straight
//     out of the previous layer's accumulators in the shadow of the current layer's MFMAs.  A wavefront owns 8 of the 16 units
//     of a layer; the other 8 come from its partner through a 2-deep LDS exchange buffer, written one K-step before use and
//     published by the K-step's one workgroup barrier (the same barrier that hands over the weight slot).
//     ~230 registers per wavefront => 2 wavefronts per SIMD: while one waits (LDS, barrier, DMA issue, VALU re-packing)
//     the other one's MFMAs keep the matrix pipe busy.
//   * weights: the SAME blob (16 KiB slot per K-step, LDS-DMA ring, 3 slots ahead); a wavefront reads only the 8 KiB of its
//     4 output blocks, in two halves that are re-loaded right behind the MFMAs that consumed them.
#include "nerf_bf16_common.h"

// Timing-only ablation switches (-DNM_ABL=<bits>; results are garbage): 1 no MFMA, 2 no unit re-packing, 4 no weight DMA,
// 8 no A-operand loads, 16 no K-step barrier
#ifndef NM_ABL
#define NM_ABL 0
#endif

namespace {
using namespace nmbf;

#if NM_ABL & 1
#undef MFMA_BF16
#define MFMA_BF16(a, b, c) (c)
#endif

constexpr int NWAVE = 8;
constexpr int THREADS = 64 * NWAVE;
constexpr int SMALL_LDS = 3104;  // floats of the small-parameter block kept in LDS (>= SMALL, 16-byte multiple)

// LDS map (floats)
constexpr int L_SMALL = 0;
constexpr int L_RING = SMALL_LDS;                    // [NRING][16 KiB]
constexpr int L_IPE = L_RING + NRING * SLOT_FLOATS;  // [4 groups][XS][2 (hi,lo)][64 lanes][4]: B operands of the IPE K-steps
constexpr int L_XCH = L_IPE + 4 * XS * 2 * 256;      // [4 groups][2 buffers][2 (hi,lo)][64 lanes][4]: unit exchange of a pair
constexpr int L_SCR = L_XCH + 4 * 2 * 2 * 256;       // per-sample scratch
constexpr int L_P2 = L_SCR + TILE * 12 + 32;         // partial sums of the second wavefront of a pair: [128] sigma, [3][128] rgb
constexpr int L_FEAT = L_P2 + 4 * TILE;              // [4 groups][256] partial feature sums
constexpr int L_EX = L_FEAT + 4 * 256;               // [4 ray slots][48] views-layer extra inputs
constexpr int L_LEFT = L_EX + 4 * 48;                // leftover list: [128] ray index, [128] transmittance
constexpr int L_TOTAL = L_LEFT + 2 * TILE;
static_assert(L_TOTAL * 4 <= 163840, "LDS budget");
static_assert((L_RING * 4) % 16 == 0 && (L_IPE * 4) % 16 == 0 && (L_XCH * 4) % 16 == 0, "16-byte alignment of the operand regions");

// Slot g: 16 pieces of 1 KiB, two per wavefront; both share ONE global address / M0 and differ in the immediate offset.
__device__ __forceinline__ void dma_slot(const char* blob_slots, int g, float* ring, int wave, int lane) {
  const unsigned voff = (unsigned)(wave * 2048 + lane * 16);
  const char* base = blob_slots + (size_t)g * SLOT_BYTES;  // uniform
  const auto* src = (const __attribute__((address_space(1))) void*)(base + voff);
  auto* dst = (__attribute__((address_space(3))) void*)(ring + (g & (NRING - 1)) * SLOT_FLOATS + wave * 512);
#if !(NM_ABL & 4)
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
#endif
}

struct Unit {
  u32x4 h, l;  // B operands (hi, lo) of one K-step: 8 bf16 each, as 4 packed pairs
};

// A operands of a wavefront for one K-step: NB output blocks x (hi, lo)
template <int NB>
struct AOps {
  bf16x8 h[NB], l[NB];
};

struct Ctx {
  const char* blob_slots;
  float* ring;
  const float* sm_small;
  float* xch;       // exchange buffer of this pair, + lane * 4: [2 buffers][2 (hi,lo)][256]
  int nslots, wave, lane, hi, half;
  int g;            // weight slot of the current K-step
  Unit xc, xn;      // B operands of the current / next K-step
  f32x4 b0, b1;     // bias of the unit this wavefront makes in the next K-step (loaded one K-step ahead)
};

// blocks [b0, b0 + n) of this wavefront (block index within the slot = blk0 + b)
template <int NB>
__device__ __forceinline__ void load_a(AOps<NB>& A, int b0, int n, const float* slot, int blk0, int lane) {
  const u32x4* s4 = reinterpret_cast<const u32x4*>(slot) + lane;
#if NM_ABL & 8
  return;
#endif
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b >= b0 && b < b0 + n) {
      A.h[b] = __builtin_bit_cast(bf16x8, s4[((blk0 + b) * 2 + 0) * 64]);
      A.l[b] = __builtin_bit_cast(bf16x8, s4[((blk0 + b) * 2 + 1) * 64]);
    }
  }
}

// Rendezvous in the middle of K-step g: this wavefront's DMA pieces of slot g+1 have landed (loads retire in order: at most
// the 4 pieces of slots g+2, g+3 may remain), its LDS traffic (exchange writes, operand reads) is complete, then the
// workgroup barrier, then the DMA of slot g+4 into the ring position slot g occupied (every wavefront has read slot g).
// Branch free: the stream simply runs 4 slots past the last one a tile uses (the blob is padded by 4 slots), so that the
// compiler sees straight-line code and can count its LDS waits instead of draining the queue at every control-flow join.
__device__ __forceinline__ void rendezvous(Ctx& cx) {
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
#if !(NM_ABL & 16)
  __builtin_amdgcn_s_barrier();
#endif
  dma_slot(cx.blob_slots, cx.g + 4, cx.ring, cx.wave, cx.lane);
}

// Where the B operands of the NEXT K-step come from
enum NextKind { NEXT_NONE = 0, NEXT_UNIT = 1, NEXT_LDS = 2 };

// Next K-step's unit: made from the previous layer's accumulators by the wavefront that owns it (and published through the
// exchange buffer), fetched from the exchange buffer by the partner.  `prev` = the accumulator set of the finished layer,
// `u` = unit index 0..15, `lo` = index of the finished layer in the bias table, floor_v = 0 (relu) or -inf (feature_linear).
struct NextUnit {
  Ctx& cx;
  const f32x16 (&prev)[4];
  int u, lo;
  float floor_v;
  bool own;        // this wavefront makes unit u (in this K-step)
  float v8[8];
  // bias of unit `unit` of layer lo (both wavefronts of a pair load it: no branch around an LDS operation)
  __device__ __forceinline__ void load_bias(int unit) {
    const float* bl = cx.sm_small + OFF_BIAS + lo * 256 + (unit >> 1) * 32 + 16 * (unit & 1) + 4 * cx.hi;
    cx.b0 = *reinterpret_cast<const f32x4*>(bl);
    cx.b1 = *reinterpret_cast<const f32x4*>(bl + 8);
  }
  // pieces 0..5, issued behind the phase-1 MFMAs
  __device__ __forceinline__ void piece(int j) {
#if NM_ABL & 2
    return;
#endif
    if (!own) return;
    const int ob = (u >> 1) & 3, m = u & 1;
    if (j < 2) {  // bias + relu of 4 elements
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = 2 * j + e;
        v8[i] = __builtin_fmaxf(prev[ob][8 * m + i] + cx.b0[i], floor_v);
        v8[4 + i] = __builtin_fmaxf(prev[ob][8 * m + 4 + i] + cx.b1[i], floor_v);
      }
    } else {      // pair p: hi halves, remainders, lo halves
      const int p = j - 2;
      const unsigned hp = pack_bf16(v8[2 * p], v8[2 * p + 1]);
      const float f0 = __uint_as_float(hp << 16), f1 = __uint_as_float(hp & 0xffff0000u);
      cx.xn.h[p] = hp;
      cx.xn.l[p] = pack_bf16(v8[2 * p] - f0, v8[2 * p + 1] - f1);
    }
  }
  // owner: publish (before the rendezvous)
  __device__ __forceinline__ void publish() {
    if (own) {
      float* d = cx.xch + (u & 1) * 512;
      *reinterpret_cast<u32x4*>(d) = cx.xn.h;
      *reinterpret_cast<u32x4*>(d + 256) = cx.xn.l;
    }
  }
  // after the rendezvous, BOTH wavefronts fetch the unit (the owner reads its own data back: no branch, no select) and
  // the bias of the unit made in the next K-step
  __device__ __forceinline__ void fetch() {
    const float* d = cx.xch + (u & 1) * 512;
    cx.xn.h = *reinterpret_cast<const u32x4*>(d);
    cx.xn.l = *reinterpret_cast<const u32x4*>(d + 256);
    if (u + 1 < HS) load_bias(u + 1);
  }
};
// Next K-step's operands are ready-made in LDS (IPE steps): both wavefronts of the pair fetch them after the rendezvous.
struct NextLds {
  Ctx& cx;
  const float* src;  // + lane * 4; hi at src, lo at src + 256
  __device__ __forceinline__ void piece(int) {}
  __device__ __forceinline__ void publish() {}
  __device__ __forceinline__ void fetch() {
    cx.xn.h = *reinterpret_cast<const u32x4*>(src);
    cx.xn.l = *reinterpret_cast<const u32x4*>(src + 256);
  }
};
struct NextNone {
  __device__ __forceinline__ void piece(int) {}
  __device__ __forceinline__ void publish() {}
  __device__ __forceinline__ void fetch() {}
};

// One K-step (weight slot cx.g) of a wavefront with NB output blocks, B operands = cx.xc:
//   phase 1: blocks [0, NB/2): 3 MFMAs each (w_hi*x_hi, w_hi*x_lo, w_lo*x_hi), the next unit's pieces behind them
//   publish the next unit | rendezvous (slot g+1 handed over, exchange visible) | re-load the phase-1 operands from slot g+1,
//   fetch the next unit
//   phase 2: blocks [NB/2, NB), then re-load their operands from slot g+1
// blk0 = first block of this wavefront inside a slot.
template <int NB, bool FIRST, class Next>
__device__ __forceinline__ void kstep(f32x16 (&acc)[4], Ctx& cx, AOps<NB>& A, int blk0, Next next) {
  constexpr int HB = NB / 2;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bf16x8 xh = __builtin_bit_cast(bf16x8, cx.xc.h), xl = __builtin_bit_cast(bf16x8, cx.xc.l);
  const int g = cx.g;
  const float* nslot = cx.ring + ((g + 1) & (NRING - 1)) * SLOT_FLOATS;
  int pc = 0;
#pragma unroll
  for (int o = 0; o < HB; ++o) {
    acc[o] = MFMA_BF16(A.h[o], xh, FIRST ? zero : acc[o]);
    next.piece(pc++);
  }
#pragma unroll
  for (int o = 0; o < HB; ++o) {
    acc[o] = MFMA_BF16(A.h[o], xl, acc[o]);
    next.piece(pc++);
  }
#pragma unroll
  for (int o = 0; o < HB; ++o) {
    acc[o] = MFMA_BF16(A.l[o], xh, acc[o]);
    next.piece(pc++);
  }
#pragma unroll
  for (; pc < 6; ++pc) next.piece(pc);
  next.publish();
  rendezvous(cx);
  load_a<NB>(A, 0, HB, nslot, blk0, cx.lane);
  next.fetch();
#pragma unroll
  for (int o = HB; o < NB; ++o) acc[o] = MFMA_BF16(A.h[o], xh, FIRST ? zero : acc[o]);
#pragma unroll
  for (int o = HB; o < NB; ++o) acc[o] = MFMA_BF16(A.h[o], xl, acc[o]);
#pragma unroll
  for (int o = HB; o < NB; ++o) acc[o] = MFMA_BF16(A.l[o], xh, acc[o]);
  load_a<NB>(A, HB, HB, nslot, blk0, cx.lane);
  cx.xc = cx.xn;
  cx.g = g + 1;
}

// Unit 0 of a finished layer (its accumulators are complete only now): made by the first wavefront of the pair, published,
// one workgroup barrier, fetched by the partner.  The only part of the re-packing that is not hidden behind MFMAs.
__device__ __forceinline__ void first_unit(const f32x16 (&prev)[4], int lo, float floor_v, Ctx& cx) {
  NextUnit nu{cx, prev, 0, lo, floor_v, cx.half == 0, {}};
  nu.load_bias(0);
#pragma unroll
  for (int j = 0; j < 6; ++j) nu.piece(j);
  nu.publish();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  nu.fetch();  // + bias of unit 1, made during the first K-step of the next layer
  cx.xc = cx.xn;
}

// 16 hidden K-steps of a layer whose input is the finished layer `lo` held in `prev`; `tail_lds` != nullptr: the K-step after
// the last hidden one takes its operands from LDS (layer 5: IPE steps follow).
template <int NB>
__device__ __forceinline__ void hidden_steps(f32x16 (&cur)[4], const f32x16 (&prev)[4], int lo, float floor_v, Ctx& cx, AOps<NB>& A,
                                             int blk0, const float* tail_lds) {
#pragma unroll
  for (int ks = 0; ks < HS; ++ks) {
    if (ks + 1 < HS) {
      NextUnit nu{cx, prev, ks + 1, lo, floor_v, ((ks + 1) >> 3) == cx.half, {}};
      if (ks == 0) kstep<NB, true>(cur, cx, A, blk0, nu);
      else kstep<NB, false>(cur, cx, A, blk0, nu);
    } else if (tail_lds) {
      kstep<NB, false>(cur, cx, A, blk0, NextLds{cx, tail_lds});
    } else {
      kstep<NB, false>(cur, cx, A, blk0, NextNone{});
    }
  }
}

// 6 IPE K-steps (operands in LDS: [m][hi/lo][256] at ipe_src, + lane * 4); FIRST: they open the layer (layer 0)
template <bool FIRST>
__device__ __forceinline__ void ipe_steps(f32x16 (&cur)[4], Ctx& cx, AOps<4>& A, int blk0, const float* ipe_src) {
#pragma unroll
  for (int m = 0; m < XS; ++m) {
    if (m + 1 < XS) {
      NextLds nl{cx, ipe_src + (m + 1) * 512};
      if (m == 0) kstep<4, FIRST>(cur, cx, A, blk0, nl);
      else kstep<4, false>(cur, cx, A, blk0, nl);
    } else {
      kstep<4, false>(cur, cx, A, blk0, NextNone{});
    }
  }
}

__global__ void __launch_bounds__(THREADS, 2) nerf_fwd_bf16x3_2w_kernel(NerfArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[L_TOTAL];
  float* const sm_small = sm + L_SMALL;
  float* const ring = sm + L_RING;
  float* const sm_ipe = sm + L_IPE;
  float* const sm_xch = sm + L_XCH;
  float* const sm_sigma = sm + L_SCR;         // [128]
  float* const sm_rgb = sm_sigma + TILE;      // [3][128] pre-activation partial of the pair's first wavefront (+ bias)
  float* const sm_t0 = sm_rgb + 3 * TILE;
  float* const sm_t1 = sm_t0 + TILE;
  float* const sm_mean = sm_t1 + TILE;        // [3][128]
  float* const sm_dn = sm_mean + 3 * TILE;
  float* const sm_w = sm_dn + TILE;
  float* const sm_misc = sm_w + TILE;         // [32]
  float* const sm_p2 = sm + L_P2;             // [128] sigma, [3][128] rgb partials of the pair's second wavefront
  float* const sm_feat = sm + L_FEAT;         // [4][256]
  float* const sm_part = sm_feat;             // [4 half wavefronts][8] partial per-ray sums (before sm_feat is written)
  float* const sm_ex = sm + L_EX;             // [nr][48]
  int* const sm_lray = reinterpret_cast<int*>(sm + L_LEFT);
  float* const sm_lT = sm + L_LEFT + TILE;

  const int tid = threadIdx.x, lane = tid & 63, s = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: branches on it are scalar
  const int grp = wave & 3, half = wave >> 2;  // pair = wavefronts (grp, grp + 4): 32 samples, neurons 128 * half .. + 127
  // the zero-tail decision is taken HERE, from the flag nm_resample_ex left on the device (no promise by the caller)
  const bool tail_ok = a.left && !(a.tail_viol && *a.tail_viol != 0);
  const int S = a.S, R = a.R;
  const int Sa = tail_ok ? a.Sa : S;
  const int left = tail_ok ? 1 : 0;
  const int ntiles = tail_ok ? a.ntiles : a.ntiles_full;
  const int SP = Sa < TILE ? Sa : TILE;
  const int nr = TILE / SP;
  const int nchunks = (Sa + TILE - 1) / TILE;
  const bool need_rgb = !(a.flags & NM_NERF_SKIP_RGB);
  const bool feat_max = (a.flags & NM_NERF_FEAT_MAX) != 0;
  const bool need_tap = (a.feat != nullptr) || (a.sfeat != nullptr);
  const int tap = (a.tap < 0 || a.tap > 7) ? 7 : a.tap;
  const int nslots = need_rgb ? NSLOT_FULL : NSLOT_NORGB;
  const char* const blob_slots = a.blob + (size_t)SMALL_PAD * 4;

  for (int i = tid; i < SMALL / 4; i += THREADS) reinterpret_cast<f32x4*>(sm_small)[i] = reinterpret_cast<const f32x4*>(a.blob)[i];

  int nleft = 0;  // queued leftovers (uniform)
  int bid = blockIdx.x;
#pragma unroll 1
  for (;;) {
  bool lo_pass = false;
  if (left && (nleft > TILE - 4 || (bid >= ntiles && nleft > 0))) lo_pass = true;
  else if (bid >= ntiles) break;
  const int nent = lo_pass ? nleft : 0;
  // extra inputs of the views layer, one value per thread (they depend on the ray only):
  // f = 0..11 sin(2^k d), 12..23 sin(2^k d + pi/2), 24..26 raw d, 27..42 appearance, 43..47 padding
  if (need_rgb && !lo_pass && tid < nr * 48) {
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
  __syncthreads();  // small parameters / sm_ex visible; every wavefront is done with the previous tile's ring and scratch

  // (the per-lane ray data is loaded inside the chunk loop and the lane indices are re-derived after the network: nothing but
  // the four running sums below stays live across the MLP, whose K-loops need ~210 of the 256 registers)
  float red_acc = 0.f;
  float carryT = 1.f;
  float best_w = -1.f;
  float feat_run = 0.f;  // thread t < 256: running feature channel t of the (single) ray when S > 128

  const int nch = lo_pass ? 1 : nchunks;
  for (int chunk = 0; chunk < nch; ++chunk) {
    // an opaque copy of the lane id per tile: without it LLVM hoists every lane-dependent LDS address of the network (bias
    // rows, exchange slots, operand offsets) out of the persistent loop and spills them
    const int lane = launder((int)(threadIdx.x & 63)), s = lane & 31, hi = lane >> 5;
    const int js = grp * 32 + s;
    const int rl = js / SP;
    // regular tile: lane's ray = slot js / SP of the tile; leftover pass: lane js owns queue entry js (idle lanes redo entry 0)
    const int ray = lo_pass ? (js < nent ? sm_lray[js] : R) : bid * nr + rl;
    const int rc = lo_pass ? sm_lray[js < nent ? js : 0] : (ray < R ? ray : R - 1);
    const float* rp = a.rays + (size_t)rc * 12;
    const float o0 = rp[0], o1 = rp[1], o2 = rp[2], d0 = rp[3], d1 = rp[4], d2 = rp[5], radius = rp[11];
    const float dsq0 = d0 * d0, dsq1 = d1 * d1, dsq2 = d2 * d2;
    const float dmag = fmaxf(1e-10f, (dsq0 + dsq1) + dsq2);
    const float dnorm = sqrtf((dsq0 + dsq1) + dsq2);
    const float nul0 = 1.0f - dsq0 / dmag, nul1 = 1.0f - dsq1 / dmag, nul2 = 1.0f - dsq2 / dmag;
    const int sidx = lo_pass ? Sa : chunk * TILE + (js % SP);
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
    if (hi == 0 && half == 0) {
      sm_t0[js] = t0; sm_t1[js] = t1;
      sm_mean[js] = mean[0]; sm_mean[TILE + js] = mean[1]; sm_mean[2 * TILE + js] = mean[2];
      sm_dn[js] = dnorm;
    }

    // start the weight stream: slots 0, 1, 2
    dma_slot(blob_slots, 0, ring, wave, lane);
    dma_slot(blob_slots, 1, ring, wave, lane);
    dma_slot(blob_slots, 2, ring, wave, lane);

    // ---- integrated positional encoding -> B operands of the 6 IPE K-steps, parked in LDS; the two wavefronts of a pair
    // make three K-steps each.  K-slot (step m, half h, i) <-> encoding index f = 16 m + 8 h + i in the reference's order
    // f = part*45 + scale*3 + axis (part 0: sin(2^scale x), part 1: sin(2^scale x + pi/2)); f >= 90 is padding.
    {
      float* dst = sm_ipe + grp * (XS * 2 * 256) + lane * 4;
#pragma unroll
      for (int mm = 0; mm < XS / 2; ++mm) {
        float v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // the four compile-time candidates (pair half hh, lane half h2) of this lane's encoding index; the parameters are
          // selected per lane, the encoding itself is evaluated once
          const int f00 = 16 * mm + i, f01 = f00 + 8, f10 = 16 * (3 + mm) + i, f11 = f10 + 8;
          const int g00 = f00 < 90 ? f00 : 0, g01 = f01 < 90 ? f01 : 0, g10 = f10 < 90 ? f10 : 0, g11 = f11 < 90 ? f11 : 0;
          const int f_lo = half ? g10 : g00, f_hi = half ? g11 : g01;  // wave-uniform selects
          const int fsel = hi ? f_hi : f_lo;
          const bool live = (half ? (hi ? f11 : f10) : (hi ? f01 : f00)) < 90;
          const int ax0 = half ? g10 % 3 : g00 % 3, ax1 = half ? g11 % 3 : g01 % 3;
          const float mu_lo = ax0 == 0 ? mean[0] : ax0 == 1 ? mean[1] : mean[2], mu_hi = ax1 == 0 ? mean[0] : ax1 == 1 ? mean[1] : mean[2];
          const float vr_lo = ax0 == 0 ? var[0] : ax0 == 1 ? var[1] : var[2], vr_hi = ax1 == 0 ? var[0] : ax1 == 1 ? var[1] : var[2];
          const float mu_s = hi ? mu_hi : mu_lo, vr_s = hi ? vr_hi : vr_lo;
          const float sc = (float)(1 << ((fsel % 45) / 3));
          const float ph = fsel >= 45 ? 1.57079637050628662109375f : 0.f;
          const float xe = mu_s * sc;
          const float v = __builtin_amdgcn_exp2f((-0.5f * (vr_s * (sc * sc))) * 1.44269504088896340736f) * sin32(xe + ph);
          v8[i] = live ? v : 0.f;
        }
        bf16x8 h8, l8;
        split8(v8, h8, l8);
        const int m = 3 * half + mm;
        *reinterpret_cast<u32x4*>(dst + (m * 2 + 0) * 256) = __builtin_bit_cast(u32x4, h8);
        *reinterpret_cast<u32x4*>(dst + (m * 2 + 1) * 256) = __builtin_bit_cast(u32x4, l8);
      }
    }

    // ---- the network ---------------------------------------------------------------------------------------------------
    Ctx cx;
    cx.blob_slots = blob_slots; cx.ring = ring; cx.sm_small = sm_small;
    cx.xch = sm_xch + grp * (2 * 2 * 256) + lane * 4;
    cx.nslots = nslots; cx.wave = wave; cx.lane = lane; cx.hi = hi; cx.half = half; cx.g = 0;
    const float* ipe_src = sm_ipe + grp * (XS * 2 * 256) + lane * 4;
    const int blk0 = 4 * half;
    // slot 0: own pieces landed (slots 1, 2 may be in flight), IPE operands of the pair written, barrier, DMA of slot 3
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dma_slot(blob_slots, 3, ring, wave, lane);
    AOps<4> A;
    load_a<4>(A, 0, 4, ring, blk0, lane);
    cx.xn.h = *reinterpret_cast<const u32x4*>(ipe_src);
    cx.xn.l = *reinterpret_cast<const u32x4*>(ipe_src + 256);
    cx.xc = cx.xn;

    f32x16 accA[4], accB[4];
    float sig_part = 0.f;
    f32x4* const tapw = reinterpret_cast<f32x4*>(a.ws) + ((size_t)blockIdx.x * NWAVE + wave) * 16 * 64 + lane;

    // tapped activations (fp32, after bias and relu) of the finished layer lo -> L2-resident workspace, 1 KiB per store
    auto dump_tap = [&](const f32x16 (&fin)[4], int lo) {
      const float* bl = sm_small + OFF_BIAS + lo * 256 + 128 * half + 4 * hi;
      f32x4* tp = tapw;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(fin[ob][4 * q + e] + b[e], 0.f);
          tp[(ob * 4 + q) * 64] = v;
        }
      }
    };
    // density head on the finished layer 7: this wavefront's share of relu(h7) . w_alpha
    auto alpha_head = [&](const f32x16 (&fin)[4]) {
      const float* bl = sm_small + OFF_BIAS + 7 * 256 + 128 * half + 4 * hi;
      const float* wa = sm_small + OFF_WALPHA + 128 * half + 4 * hi;
      float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q);
          const f32x4 w4v = *reinterpret_cast<const f32x4*>(wa + ob * 32 + 8 * q);
          p0 = NM_FMA(__builtin_fmaxf(fin[ob][4 * q + 0] + b[0], 0.f), w4v[0], p0);
          p1 = NM_FMA(__builtin_fmaxf(fin[ob][4 * q + 1] + b[1], 0.f), w4v[1], p1);
          p2 = NM_FMA(__builtin_fmaxf(fin[ob][4 * q + 2] + b[2], 0.f), w4v[2], p2);
          p3 = NM_FMA(__builtin_fmaxf(fin[ob][4 * q + 3] + b[3], 0.f), w4v[3], p3);
        }
      sig_part = (p0 + p1) + (p2 + p3);
    };
    const int tapl = need_tap ? tap : -1;

    // layer 0 (IPE -> 256): accA
    ipe_steps<true>(accA, cx, A, blk0, ipe_src);
    if (tapl == 0) dump_tap(accA, 0);
    first_unit(accA, 0, 0.f, cx);
    // layers 1..7 alternate between the two accumulator sets (odd layers: accB <- accA, even layers: accA <- accB); layer 5
    // appends the IPE K-steps of the skip connection
#pragma unroll 1
    for (int l = 1; l <= 7; l += 2) {
      hidden_steps<4>(accB, accA, l - 1, 0.f, cx, A, blk0, l == 5 ? ipe_src : nullptr);
      if (l == 5) ipe_steps<false>(accB, cx, A, blk0, ipe_src);
      if (tapl == l) dump_tap(accB, l);
      if (l == 7) break;
      first_unit(accB, l, 0.f, cx);
      hidden_steps<4>(accA, accB, l, 0.f, cx, A, blk0, nullptr);
      if (tapl == l + 1) dump_tap(accA, l + 1);
      first_unit(accA, l + 1, 0.f, cx);
    }
    alpha_head(accB);
    float pr = 0.f, pg = 0.f, pb = 0.f;
    if (need_rgb) {
      first_unit(accB, 7, 0.f, cx);
      hidden_steps<4>(accA, accB, 7, 0.f, cx, A, blk0, nullptr);                     // feature_linear (no relu on its output)
      // ---- views layer: 128 outputs = 2 blocks per wavefront; input = feature_linear output, then 3 extra K-steps ----
      first_unit(accA, 8, -__builtin_inff(), cx);
      AOps<2> V;
      const int vb0 = 2 * half;
      // the operands pre-loaded by the last K-step belong to blocks 4*half..: re-load the two blocks of the views layout
      load_a<2>(V, 0, 2, ring + (cx.g & (NRING - 1)) * SLOT_FLOATS, vb0, lane);
#pragma unroll
      for (int ks = 0; ks < HS; ++ks) {
        if (ks + 1 < HS) {
          NextUnit nu{cx, accA, ks + 1, 8, -__builtin_inff(), ((ks + 1) >> 3) == half, {}};
          if (ks == 0) kstep<2, true>(accB, cx, V, vb0, nu);
          else kstep<2, false>(accB, cx, V, vb0, nu);
        } else {
          kstep<2, false>(accB, cx, V, vb0, NextNone{});
        }
      }
      const float* exr = sm_ex + rl * 48 + 8 * hi;  // K-slot (step e, half h, i) <-> extra input 16 e + 8 h + i
#pragma unroll
      for (int e = 0; e < VS; ++e) {
        float v8[8];
        if (!lo_pass) {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(exr + 16 * e), e1 = *reinterpret_cast<const f32x4*>(exr + 16 * e + 4);
          v8[0] = e0[0]; v8[1] = e0[1]; v8[2] = e0[2]; v8[3] = e0[3]; v8[4] = e1[0]; v8[5] = e1[1]; v8[6] = e1[2]; v8[7] = e1[3];
        } else {
          // leftover pass: every lane has its own ray, so the per-slot table does not apply; same formulas, in registers
          const float* rq = a.rays + (size_t)rc * 12 + 8;
          const float vd0 = rq[0], vd1 = rq[1], vd2 = rq[2];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float v = 0.f;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              const int f = 16 * e + 8 * h2 + i;
              const int ax = f % 3;
              const float dax = ax == 0 ? vd0 : ax == 1 ? vd1 : vd2;
              float c = 0.f;
              if (f < 24) {
                const float xe = dax * (float)(1 << ((f % 12) / 3));
                c = nm_sinf(f < 12 ? xe : xe + 1.57079637050628662109375f);
              } else if (f < 27) {
                c = dax;
              } else if (f < 43) {
                c = a.app_row ? a.app_row[f - 27] : 0.f;
              }
              if (h2 == hi) v = c;
            }
            v8[i] = v;
          }
        }
        bf16x8 eh, el;
        split8(v8, eh, el);
        cx.xc.h = __builtin_bit_cast(u32x4, eh);
        cx.xc.l = __builtin_bit_cast(u32x4, el);
        cx.xn = cx.xc;
        kstep<2, false>(accB, cx, V, vb0, NextNone{});
      }
      // rgb head: this wavefront's 64 of the 128 views neurons
      const float* bv = sm_small + OFF_BVIEWS + 64 * half + 4 * hi;
      const float* wr = sm_small + OFF_WRGB + 64 * half + 4 * hi;
#pragma unroll
      for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + ob * 32 + 8 * q);
          const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wr + ob * 32 + 8 * q);
          const f32x4 wg4 = *reinterpret_cast<const f32x4*>(wr + 128 + ob * 32 + 8 * q);
          const f32x4 wb4 = *reinterpret_cast<const f32x4*>(wr + 256 + ob * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hv = __builtin_fmaxf(accB[ob][4 * q + e] + b4[e], 0.f);
            pr = NM_FMA(hv, wr4[e], pr);
            pg = NM_FMA(hv, wg4[e], pg);
            pb = NM_FMA(hv, wb4[e], pb);
          }
        }
      pr = pr + nm_shfl_xor32(pr);
      pg = pg + nm_shfl_xor32(pg);
      pb = pb + nm_shfl_xor32(pb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the weight stream runs 4 slots past the tile's last one: drain it
    const float sig_wave = sig_part + nm_shfl_xor32(sig_part);
    if (hi == 0) {
      if (half == 0) {
        sm_sigma[js] = sig_wave + sm_small[OFF_MISC];
        sm_rgb[js] = pr + sm_small[OFF_MISC + 1]; sm_rgb[TILE + js] = pg + sm_small[OFF_MISC + 2]; sm_rgb[2 * TILE + js] = pb + sm_small[OFF_MISC + 3];
      } else {
        sm_p2[js] = sig_wave;
        sm_p2[TILE + js] = pr; sm_p2[2 * TILE + js] = pg; sm_p2[3 * TILE + js] = pb;
      }
    }
    __syncthreads();
    // indices re-derived from an opaque copy of the thread id: the compiler must not keep the pre-network values alive
    const int tid2 = launder(threadIdx.x), lane2 = tid2 & 63, wave2 = __builtin_amdgcn_readfirstlane(tid2 >> 6);
    const int s_2 = lane2 & 31, hi2 = lane2 >> 5, grp2 = wave2 & 3, half2 = wave2 >> 2, js2 = grp2 * 32 + s_2;
    const f32x4* const tapr = reinterpret_cast<const f32x4*>(a.ws) + ((size_t)blockIdx.x * NWAVE + wave2) * 16 * 64 + lane2;
    // this thread's sample (threads < 128): density and colour of the pair
    float sig_s = 0.f, c_r = 0.f, c_g = 0.f, c_b = 0.f;
    if (tid2 < TILE) {
      sig_s = sm_sigma[tid2] + sm_p2[tid2];
      if (need_rgb) {
        c_r = 1.0f / (1.0f + expf(-(sm_rgb[tid2] + sm_p2[TILE + tid2])));
        c_g = 1.0f / (1.0f + expf(-(sm_rgb[TILE + tid2] + sm_p2[2 * TILE + tid2])));
        c_b = 1.0f / (1.0f + expf(-(sm_rgb[2 * TILE + tid2] + sm_p2[3 * TILE + tid2])));
      }
    }

    if (lo_pass) {
      // ---- leftover pass: thread tid2 < nent is sample Sa of ray sm_lray[tid2]; its weight is alpha * T(first Sa samples) and
      // its contributions are ADDED (atomics: they execute at L2, where this workgroup's earlier plain stores are)
      if (tid2 < nent) {
        const int ray2 = sm_lray[tid2];
        const float sg = fmaxf(sig_s, 0.f);
        const float delta = (sm_t1[tid2] - sm_t0[tid2]) * sm_dn[tid2];
        const float wgt = (1.0f - expf(-sg * delta)) * sm_lT[tid2];
        sm_w[tid2] = wgt;
        a.weights[(size_t)ray2 * S + Sa] = wgt;
        if (a.acc) atomicAdd(a.acc + ray2, wgt);
        if (a.rgb && need_rgb) {
          const float cc[3] = {c_r, c_g, c_b};
#pragma unroll
          for (int c = 0; c < 3; ++c) atomicAdd(a.rgb + (size_t)ray2 * 3 + c, a.white_bg ? wgt * cc[c] - wgt : wgt * cc[c]);
        }
        if (a.depth) atomicAdd(a.depth + ray2, wgt * (0.5f * (sm_t0[tid2] + sm_t1[tid2])));
        if (a.pts) {
#pragma unroll
          for (int c = 0; c < 3; ++c) atomicAdd(a.pts + (size_t)ray2 * 3 + c, wgt * sm_mean[c * TILE + tid2]);
        }
      }
      __syncthreads();
      if (need_tap) {
        if (js2 < nent) {
          const int ray2 = sm_lray[js2];
          const float wj = sm_w[js2];
#pragma unroll 4
          for (int u = 0; u < 8; ++u) {  // this wavefront's units 8 * half + u
            const f32x4 ta = tapr[(2 * u) * 64], tb = tapr[(2 * u + 1) * 64];
            const int n0 = 128 * half2 + (u >> 1) * 32 + 16 * (u & 1) + 4 * hi2;  // neurons n0 .. n0+3 and n0+8 .. n0+11
            if (a.sfeat) {
              float* dsf = a.sfeat + ((size_t)ray2 * S + Sa) * 256 + n0;
              *reinterpret_cast<f32x4*>(dsf) = ta;
              *reinterpret_cast<f32x4*>(dsf + 8) = tb;
            }
            if (a.feat) {
              float* df = a.feat + (size_t)ray2 * 256 + n0;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                atomicAdd(df + e, wj * ta[e]);
                atomicAdd(df + 8 + e, wj * tb[e]);
              }
            }
          }
        }
      }
      __syncthreads();
      break;  // (the chunk loop; a leftover pass has a single chunk)
    }

    // ---- alpha compositing --------------------------------------------------------------------------------------------------
    float alpha = 0.f, incl = 1.f;
    if (tid2 < TILE) {
      const float sg = fmaxf(sig_s, 0.f);
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
        if (left && chunk == nchunks - 1) {
          // the zero-width tail carries weight exactly 0; sample Sa is queued with the transmittance in front of it
          for (int k = Sa + 1 + tid2 % SP; k < S; k += SP) a.weights[(size_t)ray2 * S + k] = 0.f;
          if (tid2 % SP == SP - 1) {
            sm_lray[nleft + r2] = ray2;
            sm_lT[nleft + r2] = excl * ((1.0f - alpha) + 1e-10f);
          }
        }
        if (a.raw) {
          f32x4 rv = {c_r, c_g, c_b, sig_s};
          *reinterpret_cast<f32x4*>(a.raw + ((size_t)ray2 * S + s2) * 4) = rv;
        }
      }
      // per-ray sums, step 1: w * {1, rgb, t_mid, mean} reduced over each 32-sample half wavefront
      float pq[8] = {wgt, wgt * c_r, wgt * c_g, wgt * c_b, wgt * (0.5f * (sm_t0[tid2] + sm_t1[tid2])), wgt * sm_mean[tid2],
                     wgt * sm_mean[TILE + tid2], wgt * sm_mean[2 * TILE + tid2]};
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
        for (int hw2i = r2 * (SP / 32); hw2i < (r2 + 1) * (SP / 32); ++hw2i) sum += sm_part[hw2i * 8 + q];
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
    __syncthreads();  // sm_part has been consumed (sm_feat overlays it); feat_max: sm_misc[8..] visible

    // ---- feature output: weighted sum over the 32 samples of this pair, each wavefront its 128 channels ----------------
    if (need_tap) {
      f32x4 tapv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) tapv[c] = tapr[c * 64];
      const float wj = sm_w[js2];
      const int rsel = js2 / SP;                                   // ray slot of this lane's sample
      const int best = feat_max ? __float_as_int(sm_misc[8 + rsel]) : -2;
      float* prow = sm_feat + grp2 * 256 + 128 * half2 + 4 * hi2;    // partial sums of this pair
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 ta = tapv[2 * u], tb = tapv[2 * u + 1];
        float v8[8] = {ta[0], ta[1], ta[2], ta[3], tb[0], tb[1], tb[2], tb[3]};
        const int ray_s = bid * nr + js2 / SP, sidx_s = chunk * TILE + js2 % SP;  // (regular tiles only)
        if (a.sfeat && ray_s < R) {
          float* dsf = a.sfeat + ((size_t)ray_s * S + sidx_s) * 256 + 128 * half2 + (u >> 1) * 32 + 16 * (u & 1) + 4 * hi2;
          *reinterpret_cast<f32x4*>(dsf) = f32x4{v8[0], v8[1], v8[2], v8[3]};
          *reinterpret_cast<f32x4*>(dsf + 8) = f32x4{v8[4], v8[5], v8[6], v8[7]};
        }
        if (a.feat) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = feat_max ? (js2 == best ? v8[i] : 0.f) : wj * v8[i];
          nm_half_sum_dpp8(v8);  // 32-sample sums, valid in lanes 16..31 / 48..63
          if (s_2 == 16) {
            float* d = prow + (u >> 1) * 32 + 16 * (u & 1);
            *reinterpret_cast<f32x4*>(d) = f32x4{v8[0], v8[1], v8[2], v8[3]};
            *reinterpret_cast<f32x4*>(d + 8) = f32x4{v8[4], v8[5], v8[6], v8[7]};
          }
        }
      }
    }
    __syncthreads();
    if (a.feat && tid2 < 256) {
      // combine the pairs of each ray: SP samples = SP/32 pairs
      const int wpr = SP / 32;
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

  if (lo_pass) {
    nleft = 0;
    continue;
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
  if (left) nleft += (R - bid * nr) < nr ? (R - bid * nr) : nr;
  bid += gridDim.x;
  }  // tile loop
}

}  // namespace

namespace nmbf {
void launch_2w(const NerfArgs& a, int grid, hipStream_t stream) { nerf_fwd_bf16x3_2w_kernel<<<grid, THREADS, 0, stream>>>(a); }
}  // namespace nmbf
