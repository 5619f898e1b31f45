// Fused per-ray NeRF evaluation on the 16-bit matrix cores with operand splitting ("bf16x3" / "fp16x3"; "fp16x1").
//
// Round 3: the split exists in two number formats.  bf16 hi/lo parts carry 16 mantissa bits together (error of a product
// ~2^-16.5): enough for the smooth random-weight fixtures (features 3e-7 from fp32), NOT enough for a trained-like scene --
// densities of +-1e4 come out 0.25 off and the compositing weights 7e-4 (tests/golden/nerf_surface_r512_s128.npz).  fp16
// hi/lo parts carry 22 bits (11 + 11; below 2^-14 the lo part is a subnormal with an ABSOLUTE quantum of 2^-24, which is all
// a sum of products needs): the same three MFMAs per product (v_mfma_f32_32x32x16_f16 runs at the bf16 rate), the same
// re-packing cost, fp32-class results (density error 0.01 on that fixture, the fp32 MFMA kernel: 0.013).  fp16x3 is the
// default parity arithmetic (NerfRenderer.precision); operands beyond +-65504 saturate (v_med3), they do not overflow.
//
// Same computation and outputs as nerf_fwd.hip; what changes is
// the arithmetic of the layer products: every fp32 operand x is split into two bf16 values x = hi + lo (16 mantissa
// bits together) and each product is evaluated as  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi  with three
// v_mfma_f32_32x32x16_bf16 (fp32 accumulation).  The dropped lo*lo term is 2^-16 relative; measured against the fp32
// oracle the rendered features differ by < 1e-6 (tolerance 1e-4) -- see DESIGN.md section 3.1b.  Three bf16 MFMAs
// cost 3/16 of the fp32 MFMA they replace.
//
// Structure (persistent workgroups, one per CU; a tile = 4 wavefronts x 32 samples, one wavefront per SIMD):
//   * activations stay in registers between layers: the MFMA result layout (lane = sample + 32*half, register r <->
//     neuron (r&3)+8*(r>>2)+4*half) maps 8 consecutive registers of a lane onto the 8 K-slots of one 32x32x16 step, so
//     re-packing into (hi, lo) bf16 B operands is lane-local (bias, relu, cvt, subtract, cvt);
//   * cross-layer software pipeline: a finished layer is moved out of the accumulators (AGPRs -> 128 VGPR scalars) and
//     re-packed one K-step "unit" at a time INSIDE the K-loop of the layer that consumes it, 3-4 VALU instructions
//     behind each MFMA of the second half slot (pinned with sched_barriers and opaque asm, see UnitWork), so that only
//     the 128 accumulator reads and unit 0 are exposed per layer;
//   * weights are pre-split and pre-ordered on the host into 16 KiB "slots" = one K-step for all 8 output blocks,
//     streamed by all 4 wavefronts with global_load_lds (LDS DMA, no VGPRs) into a 4-slot LDS ring: two slots in use, two
//     in flight, ONE s_barrier + s_waitcnt vmcnt per TWO slots (ring_acquire_two; the barrier's skew was the largest single
//     overhead of a K-step in the cycle trace); every wavefront then reads its A operands with conflict-free ds_read_b128
//     (the ring layout is lane-linear, exactly what the DMA writes);
//   * the tapped activations (feature output) are parked in an L2-resident workspace (32 x 1 KiB stores per wavefront)
//     until the compositing weights are known, then reduced over the 32 samples of a wavefront with DPP adds;
//   * the integrated positional encoding is evaluated once per 128-sample chunk (each lane the 48 values of its wavefront
//     half, fp32 sine with a 4-term Cody-Waite reduction of the exact argument 2^i x) and parked in LDS as ready-made
//     B operands for layers 0 and 5.
#include "nerf_bf16_common.h"
#include <stdlib.h>

// -DNM_ABL=<bits>: TIMING-ONLY ablations for scripts/ab_nerf.py (results are garbage): upper bounds of what removing one
// ingredient of the K-loop / tile could buy.  1: no weight DMA, 2: no ring barrier, 4: no LDS operand reads, 8: no unit
// re-packing work, 16: no accumulator hand-over reads, 32: no MFMAs (everything else stays).  Never set in the library build.
#ifndef NM_ABL
#define NM_ABL 0
#endif
// the tapped activations come back from the workspace by LDS-DMA, started when the tile's last K-loop ends   [NM_TAP_PREFETCH: the losing arm is in scripts/variants/nerf_study_switches_r4.patch]
// fp16x3: hi part of an activation rounded to NEAREST (v_cvt_pk_f16_f32) instead of toward zero -- halves |lo|   [NM_HI_RNE: the losing arm is in scripts/variants/nerf_study_switches_r4.patch]
#ifndef NM_TELEMETRY
#define NM_TELEMETRY 1  // fp16x3: running maximum of the re-packed values (saturation flag, range telemetry); 0 in timing A/B builds only
#endif
#ifndef NM_IPE_EXACT
#define NM_IPE_EXACT 0  // 1: IPE by expf + fp64-reduced sine like nerf_fwd.hip (A/B of the encoding's share of the error)
#endif
// split modes: one ring barrier per TWO K-steps (0: the round-1..3 protocol, one per K-step)   [NM_RING_PAIRS: the losing arm is in scripts/variants/nerf_study_switches_r4.patch]

namespace {
using namespace nmbf;

// Arithmetic mode P of the layer products (template parameter of everything below):
//   P = 0  "bf16x3": operands split into bf16 hi / lo parts, three MFMAs per product block (16 KiB weight slots: hi and lo)
//   P = 2  "fp16x3": the same with fp16 hi / lo parts (22 instead of 16 mantissa bits; saturating at the fp16 range)
//   P = 1  "fp16x1": operands rounded once to fp16, ONE MFMA per product block (8 KiB slots) -- opt-in throughput mode of
//                    the lean render's coarse pass: DESIGN.md section 3.1d
// Operands are carried as 16-byte vectors typed bf16x8 in all modes; P = 1, 2 reinterpret them as 8 x fp16.
template <int P> constexpr bool is_bf16() { return P == 0 || P == 4; }
template <int P> constexpr bool has_gates() { return P == 4; }
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int P>
__device__ __forceinline__ f32x16 mfma_p(const bf16x8& a, const bf16x8& b, const f32x16& c) {
#if NM_ABL & 32
  f32x16 r = c;  // (keeps the operands alive, issues nothing)
  asm volatile("" : "+v"(r) : "v"(a), "v"(b));
  return r;
#endif
  if constexpr (is_bf16<P>()) return MFMA_BF16(a, b, c);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
//   P = 4  bf16x3 + the ReLU gates of every layer recorded as bits (the pointwise forward of the iNeRF refinement, see the end of the file)
template <int P> constexpr bool is_split() { return P != 1; }  // hi / lo operand pairs, three products
template <int P> constexpr int slot_bytes() { return is_split<P>() ? SLOT_BYTES : SLOT_BYTES / 2; }
template <int P> constexpr int slot_floats() { return slot_bytes<P>() / 4; }
// ring geometry: the same 64 KiB hold 4 slots of 16 KiB or 8 of 8 KiB; a slot is requested `ring_ahead` K-steps before its use
// (fp16x1: a K-step is 8 MFMAs, ~300 cycles -- two steps ahead would be less than the L2 -> LDS latency)
template <int P> constexpr int ring_slots() { return is_split<P>() ? NRING : 2 * NRING; }
template <int P> constexpr int ring_ahead() { return is_split<P>() ? 4 : 6; }
constexpr float F16_MAX = 65504.0f;
// x = hi + lo with hi, lo fp16 (round to nearest even), x clamped to the fp16 range first
__device__ __forceinline__ void split8_f16(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  f16x8 h8, l8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float c = __builtin_amdgcn_fmed3f(v[i], -F16_MAX, F16_MAX);
    const _Float16 h = (_Float16)c;
    h8[i] = h;
    l8[i] = (_Float16)(c - (float)h);
  }
  hi = __builtin_bit_cast(bf16x8, h8);
  lo = __builtin_bit_cast(bf16x8, l8);
}
template <int P>
__device__ __forceinline__ void split8_p(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  if constexpr (is_bf16<P>()) split8(v, hi, lo);
  else split8_f16(v, hi, lo);
}
template <int P>
__device__ __forceinline__ float* ring_slot(float* ring, int g) { return ring + (g & (ring_slots<P>() - 1)) * slot_floats<P>(); }
__device__ __forceinline__ unsigned pack_f16(float a, float b) {  // v_cvt_pk_f16_f32 (round to nearest even, two values)
  const f16x2 h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ bf16x8 pack8_f16(const float (&v)[8]) {
  const u32x4 r = {pack_f16(v[0], v[1]), pack_f16(v[2], v[3]), pack_f16(v[4], v[5]), pack_f16(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, r);
}

// The 4 pieces share ONE global address and ONE M0 (LDS base) and differ only in the instruction's immediate offset,
// which the hardware adds on both sides -- measured 31 instead of 58 cycles of issue per piece beside the MFMAs.
// Address = uniform slot base (SGPR pair) + one 32-bit per-lane offset: no 64-bit VGPR arithmetic per slot.
template <int P>
__device__ __forceinline__ void dma_slot(const char* blob_slots, int g, float* ring, int wave, int lane) {
  if constexpr (NM_ABL & 1) return;
  const unsigned voff = (unsigned)(wave * (slot_bytes<P>() / 4) + lane * 16);
  const char* base = blob_slots + (size_t)g * slot_bytes<P>();  // uniform
  const auto* src = (const __attribute__((address_space(1))) void*)(base + voff);
  auto* dst = (__attribute__((address_space(3))) void*)(ring_slot<P>(ring, g) + wave * (slot_floats<P>() / 4));
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
  if constexpr (is_split<P>()) {
    __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
  }
}

// Ring protocol for slot g (identical sequence in all 4 wavefronts):
//   wait until this wavefront's DMA pieces of slot g have landed (at most the 4 instructions of slot g+1 may remain
//   in flight), barrier (=> every wavefront's pieces landed AND everybody finished reading slot g-1... g-2), then
//   start the DMA of slot g+2 into the ring position that slot g-2 occupied.
template <int P>
__device__ __forceinline__ void ring_acquire(const char* blob_slots, int g, int nslots, float* ring, int wave, int lane) {
  if constexpr (is_split<P>()) {
    NM_WAIT_VMCNT(4);  // (slot g+1 is always in flight: the stream runs on into the blob's padding)
  } else {
    // Branch free: the stream simply runs on past the tile's last slot (the blob is padded by ring_ahead slots), so slots
    // g+1 .. g+5 are ALWAYS in flight here, 2 DMA instructions per wavefront each.  (A first version that counted the
    // remaining slots cost ten scalar branches per K-step -- as much as the 8 MFMAs.)
    NM_WAIT_VMCNT(10);
  }
  if constexpr (!(NM_ABL & 2)) __builtin_amdgcn_s_barrier();
  if constexpr (is_split<P>()) dma_slot<P>(blob_slots, g + ring_ahead<P>(), ring, wave, lane);
  // (fp16x1: this form only opens a tile -- slot 0 landed, slots 1..5 in flight, nothing new requested; inside the stream
  //  ring_acquire_pair does the work for two K-steps at once)
}

// Split modes, NM_RING_PAIRS: ONE barrier per two K-steps.  The s_memtime trace with the barrier compiled out
// (scripts/trace_nerf.py on -DNM_ABL=2) put the per-K-step barrier at 196 of a K-step's 1150 cycles -- more than the weight DMA
// (81), the operand reads (160) or the re-packing (143): four wavefronts on four SIMDs re-synchronised every 24 MFMAs pay the
// slowest one's stalls every time.  Called in the middle of every ODD K-step g (8-block layers; at the start of it in the views
// layer): slots g+1 and g+2 -- requested two K-steps ago, right behind the previous barrier -- must have landed (this wavefront's
// pieces: vmcnt(0); everybody's: the barrier); then slots g+3 and g+4 are requested into the ring positions of slots g-1 and g,
// whose last reads (the second-half operands of slot g, fetched in the first half of K-step g) every wavefront issued before it
// arrived here.  The 4-slot ring suffices: two slots in use, two in flight.
template <int P>
__device__ __forceinline__ void ring_acquire_two(const char* blob_slots, int g, float* ring, int wave, int lane) {
  // lgkmcnt(0): this wavefront's own reads of slot g (issued 8 MFMAs ago) have RETURNED before it signals the barrier -- the DMA
  // another wavefront issues right behind the barrier overwrites that ring position
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if constexpr (!(NM_ABL & 2)) __builtin_amdgcn_s_barrier();
  dma_slot<P>(blob_slots, g + 3, ring, wave, lane);
  dma_slot<P>(blob_slots, g + 4, ring, wave, lane);
}

// fp16x1, called in every EVEN K-step g: slots g+1 and g+2 have landed when at most the 6 DMA instructions of slots g+3..g+5
// remain in flight; one barrier for both; then slots g+6 and g+7 are requested into the ring positions of slots g-2 and g-1
// (every wavefront is past their MFMAs).  Halves the barriers / counted waits per MFMA of a stream whose K-step is 8 MFMAs.
__device__ __forceinline__ void ring_acquire_pair(const char* blob_slots, int g, float* ring, int wave, int lane) {
  NM_WAIT_VMCNT(6);
  __builtin_amdgcn_s_barrier();
  dma_slot<1>(blob_slots, g + 6, ring, wave, lane);
  dma_slot<1>(blob_slots, g + 7, ring, wave, lane);
}

// A operands of half a slot: 4 output blocks x (hi, lo) = 8 x 16 bytes per lane.
struct OpHalf {
  bf16x8 h[4], l[4];
};

template <int P>
__device__ __forceinline__ void load_half(OpHalf& d, const float* slot, int lane, int p) {
  if constexpr (NM_ABL & 4) {
    asm volatile("" : "+v"(d.h[0]), "+v"(d.h[1]), "+v"(d.h[2]), "+v"(d.h[3]), "+v"(d.l[0]), "+v"(d.l[1]), "+v"(d.l[2]), "+v"(d.l[3]));
    return;
  }
  const u32x4* s4 = reinterpret_cast<const u32x4*>(slot) + lane;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    if constexpr (is_split<P>()) {
      d.h[o] = __builtin_bit_cast(bf16x8, s4[((4 * p + o) * 2 + 0) * 64]);
      d.l[o] = __builtin_bit_cast(bf16x8, s4[((4 * p + o) * 2 + 1) * 64]);
    } else {
      d.h[o] = __builtin_bit_cast(bf16x8, s4[(4 * p + o) * 64]);
    }
  }
}

// Loop-carried state of the layer pipeline (all per wavefront)
struct Unit {
  u32x4 h, l;  // B operands (hi, lo) of one K-step: 8 bf16 each, as 4 packed pairs
};
struct Ctx {
  const char* blob_slots;
  float* ring;
  const float* sm_small;
  f32x4* tapw;      // this lane's column of the workspace
  // NM_TAP_PREFETCH: read-back of the tile's tapped activations (32 rows of 1 KiB per wavefront; the workspace of the 32 CUs of an XCD
  // is as large as their L2, so the rows come back over the fabric: 6.6 k cycles when the reduction asks for them itself)
  bool tap_pref;    // this tile reads its tap back (a feature output is wanted, regular tile)
  bool rgb;         // the pass has colour heads (the views K-loop is the tile's last; else layer 7's)
  float* tap_ring;  // landing zones of this wavefront, both free once the tile's last K-loop is over: ring slot `wave` (rows 0..15)
  float* tap_ipe;   //   and its IPE operand region (rows 16..27); rows 28..31 are loaded by the reduction itself, behind its first 14 units
  int nslots, wave, lane, hi;
  int tap;          // layer whose activations are tapped (-1: none)
  int g;            // next weight slot
  OpHalf opA;       // A operands of the next half slot, fetched one half slot ahead
  OpHalf opB;       // fp16x1: blocks 4-7 of the next slot (the whole slot is fetched one K-step ahead there)
  Unit xn;          // B operands of the next hidden K-step
  float sig_part;   // this lane's partial dot product of the density head
  float sc;         // fp16x3: s_l of the finished layer in cx.hv (OFF_SCALE), wavefront-uniform -> lives in an SGPR: the re-packing fma
                    // has two VGPR sources like the add it replaces
  float vmax;       // fp16x3: running max |re-packed value| of the layer being consumed (range telemetry / saturation flag)
  unsigned* rng;    // this thread's column of the [NRANGE][256] LDS table
  u32x4* gptr;       // P = 4: this thread's cell of the gate table of the tile in flight, [layer][256 threads] x 16 bytes
  unsigned gbits[4]; // P = 4: ReLU gates of the layer being re-packed, 8 bits per unit (bit k < 4: element 2k, bit 4 + k: element 2k + 1)
  float hv[128];    // finished layer (raw accumulators, before bias/relu), lane local: hv[16 block + register]
};

// 8 gate bits of one unit from its four packed hi words (two bf16 halves each): bit k = low half of word k non-zero, bit 4 + k = high half
__device__ __forceinline__ unsigned gate_byte(const u32x4& h) {
  // v_pk_min_u16 against (1, 1): 0 / 1 per half.  (Inline asm on scalars: the vector-typed __builtin_elementwise_min on a bit-cast
  // element of the ext-vector came out reading word 0 four times -- scripts/ubench/gate_byte.hip.)
  const unsigned w0 = h[0], w1 = h[1], w2 = h[2], w3 = h[3];
  unsigned m0, m1, m2, m3;
  const unsigned one = 0x00010001u;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m0) : "v"(w0), "v"(one));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m1) : "v"(w1), "v"(one));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m2) : "v"(w2), "v"(one));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m3) : "v"(w3), "v"(one));
  const unsigned t = m0 | (m1 << 1) | (m2 << 2) | (m3 << 3);
  return (t & 0xfu) | ((t >> 12) & 0xf0u);
}

// Unit u of the finished layer lo held in cx.hv: registers 8m .. 8m+7 (m = u & 1) of output block u >> 1, i.e. neurons
// 32 (u>>1) + 16 m + 4 half + {0..3, 8..11}  ->  + bias, relu, hi/lo split.
// Cut into pieces of <= 6 VALU instructions; slot_step8/4 issue one piece behind each MFMA of a half slot, pinned with
// sched_barriers, so the re-packing runs in the shadow of the matrix pipe.  Branch free on purpose: the pieces must stay
// inside the MFMAs' basic block.
template <int P>
struct UnitWork {
  Ctx& cx;
  Unit& out;
  int u, lo;
  float floor_v;
  f32x4 b0, b1;
  float sc;  // fp16x3: s_lo (OFF_SCALE)
  float v8[8];
  float f0, f1;
  unsigned hpk;  // fp16x3: the packed hi pair of the current pair of values
  // bias loads; issued ahead of the MFMAs that shadow the pieces (and ahead of the next A-operand fetch, so that the
  // counted LDS wait in front of piece 0 covers these two reads only)
  __device__ __forceinline__ void prefetch() {
    const int ob = u >> 1, m = u & 1;
    const float* bl = cx.sm_small + OFF_BIAS + lo * 256 + ob * 32 + 16 * m + 4 * cx.hi;
    b0 = *reinterpret_cast<const f32x4*>(bl); b1 = *reinterpret_cast<const f32x4*>(bl + 8);
  }
  __device__ __forceinline__ void operator()(int j) {
    if constexpr (NM_ABL & 8) return;
    const int ob = u >> 1, m = u & 1;
    if (j < 4) {               // elements j and 4 + j: bias, relu
      if constexpr (is_bf16<P>()) {
        v8[j] = __builtin_fmaxf(cx.hv[ob * 16 + 8 * m + j] + b0[j], floor_v);
        v8[4 + j] = __builtin_fmaxf(cx.hv[ob * 16 + 8 * m + 4 + j] + b1[j], floor_v);
      } else if constexpr (P == 1) {  // fp16 operands: the same instruction count with v_med3_f32 -- an activation beyond the fp16
                                      // range saturates instead of turning into infinity (and the pass into NaNs)
        v8[j] = __builtin_amdgcn_fmed3f(cx.hv[ob * 16 + 8 * m + j] + b0[j], floor_v, F16_MAX);
        v8[4 + j] = __builtin_amdgcn_fmed3f(cx.hv[ob * 16 + 8 * m + 4 + j] + b1[j], floor_v, F16_MAX);
      } else {  // fp16x3: the accumulator goes to the next layer's input scale inside the bias add (one v_fma instead of one v_add;
                // exact), and the running maximum of what is about to become fp16 is kept: a value AT the limit raises the
                // saturation flag at the end of the kernel (status[0]) -- never a silent clamp
        if constexpr (NM_ABL & 64) {  // (timing only: the round-3 form, plain add)
          v8[j] = __builtin_amdgcn_fmed3f(cx.hv[ob * 16 + 8 * m + j] + b0[j], floor_v, F16_MAX);
          v8[4 + j] = __builtin_amdgcn_fmed3f(cx.hv[ob * 16 + 8 * m + 4 + j] + b1[j], floor_v, F16_MAX);
        } else {
          v8[j] = __builtin_amdgcn_fmed3f(__builtin_fmaf(cx.hv[ob * 16 + 8 * m + j], sc, b0[j]), floor_v, F16_MAX);
          v8[4 + j] = __builtin_amdgcn_fmed3f(__builtin_fmaf(cx.hv[ob * 16 + 8 * m + 4 + j], sc, b1[j]), floor_v, F16_MAX);
        }
#if NM_TELEMETRY
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(cx.vmax) : "v"(v8[j]), "v"(v8[4 + j]));
#endif
      }
      pin(v8[j]); pin(v8[4 + j]);
    } else if constexpr (P == 1) {  // pieces 4..7: pair p = j - 4 rounded to fp16 and packed (pieces 8..11: nothing)
      if (j < 8) {
        unsigned hp = pack_f16(v8[2 * (j - 4)], v8[2 * (j - 4) + 1]);
        pin(hp);
        out.h[j - 4] = hp;
      }
    } else if (!(j & 1)) {     // pair p = (2p, 2p+1): hi halves and their fp32 values
      const int p = (j - 4) >> 1;
      unsigned hp;
      if constexpr (is_bf16<P>()) {
        hp = pack_bf16(v8[2 * p], v8[2 * p + 1]);
        f0 = __uint_as_float(hp << 16);
        f1 = __uint_as_float(hp & 0xffff0000u);
      } else {
        // fp16 parts: hi = the value truncated to 11 significant bits by v_cvt_pkrtz_f16_f32 (round toward zero, two values per
        // instruction); lo = v - hi comes straight from the PACKED hi register with v_fma_mix_f32 (an fp16 half as a source of
        // an fp32 FMA: hi * -1 + v, exact) in piece j+1 -- no fp32 copy of hi is made -- and needs 12 bits at most, rounded to
        // nearest by v_cvt_pk_f16_f32: 22 significant bits like the round-to-nearest split (below 2^-14, where fp16 is
        // subnormal, the absolute quantum 2^-24 bounds the error).
        hp = pack_f16(v8[2 * p], v8[2 * p + 1]);  // (round to nearest: |lo| <= 2^-12 |v|; the remainder below is exact for either rounding)
        hpk = hp;
      }
      pin(hp);
      if constexpr (is_bf16<P>()) { pin(f0); pin(f1); }
      out.h[p] = hp;
    } else {                   // lo halves = rounded remainders
      const int p = (j - 5) >> 1;
      float r0, r1;
      if constexpr (is_bf16<P>()) {
        r0 = v8[2 * p] - f0; r1 = v8[2 * p + 1] - f1;
      } else {
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hpk), "v"(v8[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hpk), "v"(v8[2 * p + 1]));
      }
      pin(r0); pin(r1);
      unsigned lp = is_bf16<P>() ? pack_bf16(r0, r1) : pack_f16(r0, r1);
      pin(lp);
      out.l[p] = lp;
      if constexpr (has_gates<P>()) {
        if (j == 11) {  // all four hi words of the unit exist: value > 0 <=> its bf16 hi half is non-zero (after the ReLU nothing is negative)
          const unsigned t = gate_byte(out.h);
          cx.gbits[u >> 2] |= t << (8 * (u & 3));
        }
      }
    }
  }
};
// fp16x3: the consumer of layer `slot`'s output has made all its units -- fold the running maximum into this thread's LDS cell
// (ds_max_u32 without return: fire and forget; the values are >= 0, so the bit patterns order like the floats)
template <int P>
__device__ __forceinline__ void fold_range(Ctx& cx, int slot) {
  if constexpr (P == 2) {
    // (inline asm: for a ds_ instruction it can see, the compiler first waits vmcnt(0) -- the weight stream's LDS-DMA "may write LDS" --
    //  i.e. for the two slots requested half a K-step ago, at the end of EVERY layer's K-loop.  An LDS atomic without return needs no wait;
    //  LDS operations complete in order, so one more in flight only makes the compiler's own lgkmcnt waits conservative.)
    const unsigned addr = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned*)(cx.rng + slot * 256);
    asm volatile("ds_max_u32 %0, %1" :: "v"(addr), "v"(__float_as_uint(cx.vmax)) : "memory");
    cx.vmax = 0.f;
  }
}
struct NoWork {
  __device__ __forceinline__ void prefetch() {}
  __device__ __forceinline__ void operator()(int) {}
};
// The work of a layer's LAST K-step (which re-packs nothing: all 16 units of the previous layer exist, cx.hv is dead): blocks 0..3 of the
// layer being finished are final once that K-step's first half is through, so their 64 accumulator reads (v_accvgpr_read, finish_layer's
// first half) go behind the MFMAs of its second half instead of in front of the next layer.
template <int P>
struct AccTake {
  const f32x16 (&acc)[8];
  Ctx& cx;
  __device__ __forceinline__ void prefetch() {}
  __device__ __forceinline__ void operator()(int j) {
    constexpr int per = is_split<P>() ? 6 : 8;  // 12 pieces of 6 (split modes) / 8 pieces of 8 (fp16x1)
#pragma unroll
    for (int k = 0; k < per; ++k) {
      const int i = per * j + k;
      if (i < 64) cx.hv[i] = acc_read(acc[i >> 4][i & 15]);
    }
  }
};
template <int P>
__device__ __forceinline__ UnitWork<P> unit_work(int u, int lo, Ctx& cx, Unit& out) {
  return UnitWork<P>{cx, out, u, lo, lo < 8 ? 0.f : (!is_bf16<P>() ? -F16_MAX : -__builtin_inff()), {}, {}, cx.sc, {}, 0.f, 0.f, 0u};
}

// End of layer l: move the accumulators out of the AGPRs (the next layer starts from C = 0 in the same registers) and
// make unit 0.  The only part of the re-packing that is not hidden behind MFMAs (128 + ~40 VALU instructions).
template <int P>
__device__ __forceinline__ void finish_layer(const f32x16 (&acc)[8], int l, Ctx& cx) {
  if constexpr (!(NM_ABL & 16)) {
#pragma unroll
    for (int ob = 4; ob < 8; ++ob)  // (blocks 0..3: AccTake, in the shadow of the layer's last K-step)
#pragma unroll
      for (int r = 0; r < 16; ++r) cx.hv[ob * 16 + r] = acc_read(acc[ob][r]);
  }
  if constexpr (P == 2)
    cx.sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cx.sm_small[OFF_SCALE + l])));
  UnitWork<P> w = unit_work<P>(0, l, cx, cx.xn);
  w.prefetch();
#pragma unroll
  for (int j = 0; j < 12; ++j) w(j);
}

// Density head on the finished layer 7: sigma partial = relu(h7) . w_alpha over this lane's 128 neurons.  Once per tile,
// not hidden behind MFMAs (~2k cycles).
// (fp16x3: bias and density vector are stored pre-scaled -- relu(fma(acc, s_7, b'_7)) = 2^c_8 relu(h_7), w'_alpha = 2^-c_8 w_alpha)
__device__ __forceinline__ void alpha_head(Ctx& cx) {
  const float* bl = cx.sm_small + OFF_BIAS + 7 * 256 + 4 * cx.hi;
  const float* wa = cx.sm_small + OFF_WALPHA + 4 * cx.hi;
  const float s7 = cx.sm_small[OFF_SCALE + 7];  // (1 in the other modes: fma(x, 1, b) == x + b)
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q);
      const f32x4 w4v = *reinterpret_cast<const f32x4*>(wa + ob * 32 + 8 * q);
      p0 = NM_FMA(__builtin_fmaxf(__builtin_fmaf(cx.hv[ob * 16 + 4 * q + 0], s7, b[0]), 0.f), w4v[0], p0);
      p1 = NM_FMA(__builtin_fmaxf(__builtin_fmaf(cx.hv[ob * 16 + 4 * q + 1], s7, b[1]), 0.f), w4v[1], p1);
      p2 = NM_FMA(__builtin_fmaxf(__builtin_fmaf(cx.hv[ob * 16 + 4 * q + 2], s7, b[2]), 0.f), w4v[2], p2);
      p3 = NM_FMA(__builtin_fmaxf(__builtin_fmaf(cx.hv[ob * 16 + 4 * q + 3], s7, b[3]), 0.f), w4v[3], p3);
    }
  cx.sig_part = (p0 + p1) + (p2 + p3);
}

// Start the read-back of this wavefront's 32 tapped rows: LDS-DMA into the weight ring and the IPE region (nobody needs them before the
// next tile), the last four rows into registers.  Called when the tile's last K-loop is over; between here and the reduction that
// consumes the rows lie the density / colour heads and the compositing -- barriers there wait for LDS only (NM_EPI_BARRIER).
// Two calls per tile (PART 0: the ring rows, right after the last K-loop; PART 1: the IPE rows, behind the head that follows): 28 KiB per
// wavefront in one burst outruns the CU's miss queue and the wavefront sits in ISSUE for most of the latency it wanted to hide (A/B on
// one box, 4 x 4800 x 64 with all heads: one burst -0.9 % of a launch, two -1.6 ... -2.2 %, three -1.1 %: profiles/r4_ab_tap_prefetch.log).
template <int PART>
__device__ __forceinline__ void tap_prefetch(Ctx& cx) {
  // (the caller has waited for this wavefront's last operand reads of the ring: NM_MLP_DONE_WAIT)
  if constexpr (PART == 0) __builtin_amdgcn_s_barrier();  // everybody else is through with the ring as well
  const char* src = reinterpret_cast<const char*>(cx.tapw);
  // (inline asm, not __builtin_amdgcn_global_load_lds: the compiler cannot tell these LDS writes from the head vectors / scratch the
  //  colour heads and the compositing read -- one __shared__ array -- and would put a vmcnt(0) in front of their first ds_read; the one
  //  wait these rows need is the explicit one in front of the reduction)
#pragma unroll
  for (int q = (PART == 0 ? 0 : 4); q < (PART == 0 ? 4 : 7); ++q) {
    const char* sp = src + q * 4096;
    const float* dp = q < 4 ? cx.tap_ring + q * 1024 : cx.tap_ipe + (q - 4) * 1024;
    const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(const __attribute__((address_space(3))) float*)dp);
    unsigned m0_saved;  // (M0 belongs to the compiler: handed back as found)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %2, off\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %2, off offset:2048\n\tglobal_load_lds_dwordx4 %2, off offset:3072\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "s"(lds), "v"(sp) : "memory");
  }
}

// Tapped activations (fp32, after bias and relu) of the finished layer lo -> L2-resident workspace, 1 KiB per store.
// Once per tile and not hidden behind MFMAs (~2k cycles).
__device__ __forceinline__ void dump_tap(int lo, Ctx& cx) {
  const float* bl = cx.sm_small + OFF_BIAS + lo * 256 + 4 * cx.hi;
  const float sl = cx.sm_small[OFF_SCALE + lo];  // (fp16x3: the workspace holds 2^c_{lo+1} x the activations; OFF_DESCALE undoes it per ray)
  auto* tp = (__attribute__((address_space(1))) f32x4*)cx.tapw;  // (global_store, not flat_store: a pending FLAT access makes every later ds_read wait for vmcnt)
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) {
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q][e] = __builtin_fmaxf(__builtin_fmaf(cx.hv[ob * 16 + 4 * q + e], sl, b[e]), 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr ((NM_ABL & 2048) != 0) { if (v[q][0] == 1.2345e-30f) tp[q * 64] = v[q]; }  // (timing only: the dump's arithmetic without its stores)
      else tp[q * 64] = v[q];  // immediate offsets 0, 1, 2, 3 KiB
    }
    tp += 256;
    pin(tp);  // one running pointer instead of 32 precomputed addresses
  }
}

// acc[4p .. 4p+3] (+)= W_half . (xh + xl)  as  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi; FIRST starts from C = 0
template <int P, bool FIRST, int NOB>
__device__ __forceinline__ void mfma_head(f32x16 (&acc)[NOB], int p, const OpHalf& a, const bf16x8& xh) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[4 * p + o] = mfma_p<P>(a.h[o], xh, FIRST ? zero : acc[4 * p + o]);
}
template <int P, int NOB>
__device__ __forceinline__ void mfma_tail(f32x16 (&acc)[NOB], int p, const OpHalf& a, const bf16x8& xh, const bf16x8& xl) {
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[4 * p + o] = mfma_p<P>(a.h[o], xl, acc[4 * p + o]);
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[4 * p + o] = mfma_p<P>(a.l[o], xh, acc[4 * p + o]);
}

// fp16x1 form of the K-step: 8 MFMAs whose A operands (all 8 blocks of slot g) were fetched during the PREVIOUS K-step, so
// no MFMA waits for LDS, and everything else a K-step has to issue -- the ring barrier of slot g+1 with the DMA of a later
// slot, the 8 operand reads of slot g+1, the bias reads and the 8 pieces of re-packing work -- sits BETWEEN the MFMAs, a
// few instructions behind each (with 8 MFMAs per K-step instead of 24 there is no second half to hide them behind; a first
// version that issued barrier and reads up front ran at 640 cycles per K-step against 256 of MFMA time).
// (past the last slot the fetched operands are stale ring contents nobody uses)
__device__ __forceinline__ bf16x8 load_op1(const float* slot, int lane, int blk) {
  return __builtin_bit_cast(bf16x8, (reinterpret_cast<const u32x4*>(slot) + lane)[blk * 64]);
}
template <bool FIRST, bool ACQ, class Work>
__device__ __forceinline__ void slot_step8_one(f32x16 (&acc)[8], Ctx& cx, const bf16x8& x, Work work) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int g = cx.g;
  const OpHalf A = cx.opA, B = cx.opB;
  const float* nxt = ring_slot<1>(cx.ring, g + 1);
#define NM_SB __builtin_amdgcn_sched_barrier(0)
  acc[0] = mfma_p<1>(A.h[0], x, FIRST ? zero : acc[0]); NM_SB;
  if constexpr (ACQ) ring_acquire_pair(cx.blob_slots, g, cx.ring, cx.wave, cx.lane);
  NM_SB;
  acc[1] = mfma_p<1>(A.h[1], x, FIRST ? zero : acc[1]); NM_SB;
  cx.opA.h[0] = load_op1(nxt, cx.lane, 0); cx.opA.h[1] = load_op1(nxt, cx.lane, 1); work.prefetch(); NM_SB;
  acc[2] = mfma_p<1>(A.h[2], x, FIRST ? zero : acc[2]); NM_SB;
  cx.opA.h[2] = load_op1(nxt, cx.lane, 2); cx.opA.h[3] = load_op1(nxt, cx.lane, 3); NM_SB;
  acc[3] = mfma_p<1>(A.h[3], x, FIRST ? zero : acc[3]); NM_SB;
  cx.opB.h[0] = load_op1(nxt, cx.lane, 4); cx.opB.h[1] = load_op1(nxt, cx.lane, 5); NM_SB;
  acc[4] = mfma_p<1>(B.h[0], x, FIRST ? zero : acc[4]); NM_SB;
  cx.opB.h[2] = load_op1(nxt, cx.lane, 6); cx.opB.h[3] = load_op1(nxt, cx.lane, 7); NM_SB;
  acc[5] = mfma_p<1>(B.h[1], x, FIRST ? zero : acc[5]); NM_SB;
  work(0); work(1); NM_SB;  // (first use of the bias reads: the LDS wait in front of it has two more MFMAs of cover)
  acc[6] = mfma_p<1>(B.h[2], x, FIRST ? zero : acc[6]); NM_SB;
  work(2); work(3); work(4); NM_SB;
  acc[7] = mfma_p<1>(B.h[3], x, FIRST ? zero : acc[7]); NM_SB;
  work(5); work(6); work(7); NM_SB;
  cx.g = g + 1;
}

// One K-step (slot cx.g) of an 8-block layer, software pipelined over half slots with a "consume first" order: every
// batch of LDS reads is issued right AFTER four MFMAs that use the previously fetched operands:
//   head(blocks 0-3, A) | fetch B = blocks 4-7 of slot g | tail(blocks 0-3, A)
//   ring barrier of slot g+1 (+ DMA of slot g+3)
//   head(blocks 4-7, B) | fetch A = blocks 0-3 of slot g+1 | tail(blocks 4-7, B)
// work(j), j = 0..11, is VALU work independent of this slot's second half (re-packing of a later K-step's B operands);
// piece j is issued right behind the j-th MFMA of the second half.
// (EVEN: the K-step's position in the weight stream is even -- every layer holds an even number of K-steps, so the callers know)
template <int P, bool FIRST, bool EVEN, class Work>
__device__ __forceinline__ void slot_step8(f32x16 (&acc)[8], Ctx& cx, const bf16x8& xh, const bf16x8& xl, Work work) {
  if constexpr (P == 1) {
    slot_step8_one<FIRST, EVEN>(acc, cx, xh, work);
    return;
  }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int g = cx.g;
  OpHalf B;
  mfma_head<P, FIRST, 8>(acc, 0, cx.opA, xh);
  __builtin_amdgcn_sched_barrier(0);
  load_half<P>(B, cx.ring + (g & (NRING - 1)) * SLOT_FLOATS, cx.lane, 1);
  work.prefetch();
  __builtin_amdgcn_sched_barrier(0);
  mfma_tail<P, 8>(acc, 0, cx.opA, xh, xl);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!EVEN) ring_acquire_two<P>(cx.blob_slots, g, cx.ring, cx.wave, cx.lane);
  // from here to the end of the K-step: ONE basic block (the work pieces must not be separated from their MFMAs)
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    acc[4 + o] = mfma_p<P>(B.h[o], xh, FIRST ? zero : acc[4 + o]);
    __builtin_amdgcn_sched_barrier(0);
    work(o);
    __builtin_amdgcn_sched_barrier(0);
  }
  // the next slot's first operands, behind four MFMAs ("consume first"); unconditional: past the last slot they are
  // stale ring contents nobody uses
  load_half<P>(cx.opA, cx.ring + ((g + 1) & (NRING - 1)) * SLOT_FLOATS, cx.lane, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    acc[4 + o] = mfma_p<P>(B.h[o], xl, acc[4 + o]);
    __builtin_amdgcn_sched_barrier(0);
    work(4 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    acc[4 + o] = mfma_p<P>(B.l[o], xh, acc[4 + o]);
    __builtin_amdgcn_sched_barrier(0);
    work(8 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
  cx.g = g + 1;
}

// Same for the 4-block views layer (a slot is a single half).
template <bool FIRST, bool ACQ, class Work>
__device__ __forceinline__ void slot_step4_one(f32x16 (&acc)[4], Ctx& cx, const bf16x8& x, Work work) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int g = cx.g;
  const OpHalf C = cx.opA;
  const float* nxt = ring_slot<1>(cx.ring, g + 1);
  acc[0] = mfma_p<1>(C.h[0], x, FIRST ? zero : acc[0]); NM_SB;
  if constexpr (ACQ) ring_acquire_pair(cx.blob_slots, g, cx.ring, cx.wave, cx.lane);
  NM_SB;
  acc[1] = mfma_p<1>(C.h[1], x, FIRST ? zero : acc[1]); NM_SB;
#pragma unroll
  for (int o = 0; o < 4; ++o) cx.opA.h[o] = load_op1(nxt, cx.lane, o);
  work.prefetch(); NM_SB;
  acc[2] = mfma_p<1>(C.h[2], x, FIRST ? zero : acc[2]); NM_SB;
  work(0); work(1); work(2); work(3); NM_SB;
  acc[3] = mfma_p<1>(C.h[3], x, FIRST ? zero : acc[3]); NM_SB;
  work(4); work(5); work(6); work(7); NM_SB;
  cx.g = g + 1;
}

template <int P, bool FIRST, bool EVEN, class Work>
__device__ __forceinline__ void slot_step4(f32x16 (&acc)[4], Ctx& cx, const bf16x8& xh, const bf16x8& xl, Work work) {
  if constexpr (P == 1) {
    slot_step4_one<FIRST, EVEN>(acc, cx, xh, work);
    return;
  }
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int g = cx.g;
  const OpHalf C = cx.opA;
  work.prefetch();
  if constexpr (!EVEN) ring_acquire_two<P>(cx.blob_slots, g, cx.ring, cx.wave, cx.lane);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    acc[o] = mfma_p<P>(C.h[o], xh, FIRST ? zero : acc[o]);
    __builtin_amdgcn_sched_barrier(0);
    work(o);
    __builtin_amdgcn_sched_barrier(0);
  }
  load_half<P>(cx.opA, cx.ring + ((g + 1) & (NRING - 1)) * SLOT_FLOATS, cx.lane, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    acc[o] = mfma_p<P>(C.h[o], xl, acc[o]);
    __builtin_amdgcn_sched_barrier(0);
    work(4 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    acc[o] = mfma_p<P>(C.l[o], xh, acc[o]);
    __builtin_amdgcn_sched_barrier(0);
    work(8 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
  cx.g = g + 1;
}

// Split modes, NM_VIEWS_PAIRS: TWO K-steps of the 4-block views layer per weight slot (half 0: the four output blocks of K-step 2s, half 1:
// those of K-step 2s + 1) -- the shape of slot_step8 with both halves accumulating into the same four blocks: one ring barrier, one DMA of a
// FULL slot and one counted wait per 24 MFMAs instead of per 12 (a single 4-block K-step runs at 67 cycles per MFMA against the 8-block
// layers' 46: its fixed cost does not hide behind 12 MFMAs).  w0 makes the unit the SECOND half consumes (u1, ready behind this slot's
// 12th MFMA), w1 the first unit of the next slot (cx.xn).
template <int P, bool FIRST, bool EVEN, class W0, class W1>
__device__ __forceinline__ void slot_step4x2(f32x16 (&av)[4], Ctx& cx, const bf16x8& x0h, const bf16x8& x0l, Unit& u1, W0 w0, W1 w1) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int g = cx.g;
  const OpHalf A = cx.opA;
  OpHalf B;
  w0.prefetch();
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    av[o] = mfma_p<P>(A.h[o], x0h, FIRST ? zero : av[o]);
    __builtin_amdgcn_sched_barrier(0);
    w0(o);
    __builtin_amdgcn_sched_barrier(0);
  }
  load_half<P>(B, cx.ring + (g & (NRING - 1)) * SLOT_FLOATS, cx.lane, 1);  // the second K-step's operands, behind four MFMAs
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    av[o] = mfma_p<P>(A.h[o], x0l, av[o]);
    __builtin_amdgcn_sched_barrier(0);
    w0(4 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    av[o] = mfma_p<P>(A.l[o], x0h, av[o]);
    __builtin_amdgcn_sched_barrier(0);
    w0(8 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
  // (all reads of this slot are issued: the barrier below may hand its ring position to slot g + 4)
  if constexpr (!EVEN) ring_acquire_two<P>(cx.blob_slots, g, cx.ring, cx.wave, cx.lane);
  const bf16x8 x1h = __builtin_bit_cast(bf16x8, u1.h), x1l = __builtin_bit_cast(bf16x8, u1.l);
  w1.prefetch();
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    av[o] = mfma_p<P>(B.h[o], x1h, av[o]);
    __builtin_amdgcn_sched_barrier(0);
    w1(o);
    __builtin_amdgcn_sched_barrier(0);
  }
  load_half<P>(cx.opA, cx.ring + ((g + 1) & (NRING - 1)) * SLOT_FLOATS, cx.lane, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    av[o] = mfma_p<P>(B.h[o], x1l, av[o]);
    __builtin_amdgcn_sched_barrier(0);
    w1(4 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    av[o] = mfma_p<P>(B.l[o], x1h, av[o]);
    __builtin_amdgcn_sched_barrier(0);
    w1(8 + o);
    __builtin_amdgcn_sched_barrier(0);
  }
  cx.g = g + 1;
}

// The hidden part of the views layer: layer 7's sixteen units (unit 0 in cx.xn) against the folded 128 x 256 matrix
template <int P>
__device__ __forceinline__ void views_hidden(f32x16 (&av)[4], Ctx& cx) {
  if constexpr (is_split<P>()) {
#pragma unroll
    for (int sl = 0; sl < HS / 2; ++sl) {
      const Unit x0 = cx.xn;
      Unit u1;
      const bf16x8 x0h = __builtin_bit_cast(bf16x8, x0.h), x0l = __builtin_bit_cast(bf16x8, x0.l);
      // (NSLOT_NORGB is even: slot sl of the views layer sits at an even stream position iff sl is even)
      if (sl == 0) slot_step4x2<P, true, true>(av, cx, x0h, x0l, u1, unit_work<P>(1, 7, cx, u1), unit_work<P>(2, 7, cx, cx.xn));
      else if (sl + 1 == HS / 2) slot_step4x2<P, false, false>(av, cx, x0h, x0l, u1, unit_work<P>(2 * sl + 1, 7, cx, u1), NoWork{});
      else if (sl & 1) slot_step4x2<P, false, false>(av, cx, x0h, x0l, u1, unit_work<P>(2 * sl + 1, 7, cx, u1), unit_work<P>(2 * sl + 2, 7, cx, cx.xn));
      else slot_step4x2<P, false, true>(av, cx, x0h, x0l, u1, unit_work<P>(2 * sl + 1, 7, cx, u1), unit_work<P>(2 * sl + 2, 7, cx, cx.xn));
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < HS; ks += 2) {
      {
        const Unit xc = cx.xn;
        if (ks == 0) slot_step4<P, true, true>(av, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), unit_work<P>(ks + 1, 7, cx, cx.xn));
        else slot_step4<P, false, true>(av, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), unit_work<P>(ks + 1, 7, cx, cx.xn));
      }
      {
        const Unit xc = cx.xn;
        if (ks + 2 < HS) slot_step4<P, false, false>(av, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), unit_work<P>(ks + 2, 7, cx, cx.xn));
        else slot_step4<P, false, false>(av, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), NoWork{});
      }
    }
  }
}
// The three extra K-steps (direction encoding, appearance row, padding): split modes with NM_VIEWS_PAIRS -- the first two share a slot
template <int P>
__device__ __forceinline__ void views_extras(f32x16 (&av)[4], Ctx& cx, const bf16x8 (&eh)[VS], const bf16x8 (&el)[VS]) {
  if constexpr (is_split<P>()) {
    Unit u1;
    u1.h = __builtin_bit_cast(u32x4, eh[1]); u1.l = __builtin_bit_cast(u32x4, el[1]);
    slot_step4x2<P, false, true>(av, cx, eh[0], el[0], u1, NoWork{}, NoWork{});   // stream position NSLOT_NORGB + 8: even
    slot_step4<P, false, false>(av, cx, eh[2], el[2], NoWork{});                  // a single half slot at an odd position
  } else {
#pragma unroll
    for (int e = 0; e < VS; ++e) {
      if (e & 1) slot_step4<P, false, false>(av, cx, eh[e], el[e], NoWork{});  // (the views layer's extra K-steps sit at positions 16, 17, 18)
      else slot_step4<P, false, true>(av, cx, eh[e], el[e], NoWork{});
    }
  }
}

// IPE K-steps of layers 0 (FIRST: they open the layer) and 5 (skip connection, after the hidden K-steps)
template <int P, bool FIRST>
__device__ __forceinline__ void ipe_steps(f32x16 (&acc)[8], Ctx& cx, const float* ipe_src) {
  auto operand = [&](int m, bf16x8& ph, bf16x8& pl) {
    // (fp16x1: one operand per K-step, at [m][64 lanes][4 floats] of the same LDS region)
    ph = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ipe_src + (is_split<P>() ? (m * 2 + 0) : m) * 256));
    pl = ph;
    if constexpr (is_split<P>()) pl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(ipe_src + (m * 2 + 1) * 256));
  };
#pragma unroll
  for (int m = 0; m < XS; m += 2) {  // (XS is even; pairs so that the position parity is a template argument)
    bf16x8 ph, pl;
    operand(m, ph, pl);
    if (m == 0) slot_step8<P, FIRST, true>(acc, cx, ph, pl, NoWork{});
    else slot_step8<P, false, true>(acc, cx, ph, pl, NoWork{});
    operand(m + 1, ph, pl);
    if (m + 2 == XS) slot_step8<P, false, false>(acc, cx, ph, pl, AccTake<P>{acc, cx});  // (the IPE steps close layers 0 and 5)
    else slot_step8<P, false, false>(acc, cx, ph, pl, NoWork{});
  }
}

// One pts layer (l = 1..7): unit ks+1 of the finished layer l-1 (in cx.hv) is made in the shadow of K-step ks.
// (feature_linear is no layer of this kernel: it has no activation, so nerf_pack_split multiplies it into the views layer.)
template <int P>
__device__ __forceinline__ void layer_pass(f32x16 (&acc)[8], int l, Ctx& cx, const float* ipe_src) {
  if (l - 1 == cx.tap) dump_tap(l - 1, cx);
#pragma unroll
  for (int ks = 0; ks < HS; ks += 2) {  // (pairs: the parity of a K-step's position in the stream is a template argument)
    {
      const Unit xc = cx.xn;
      if (ks == 0) slot_step8<P, true, true>(acc, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), unit_work<P>(ks + 1, l - 1, cx, cx.xn));
      else slot_step8<P, false, true>(acc, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), unit_work<P>(ks + 1, l - 1, cx, cx.xn));
    }
    {
      const Unit xc = cx.xn;
      if (ks + 2 < HS) slot_step8<P, false, false>(acc, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), unit_work<P>(ks + 2, l - 1, cx, cx.xn));
      // (every layer, no branch in the MFMA stream: in layer 5 the skip connection's IPE steps still follow, what is taken here is
      //  overwritten by their own AccTake)
      else slot_step8<P, false, false>(acc, cx, __builtin_bit_cast(bf16x8, xc.h), __builtin_bit_cast(bf16x8, xc.l), AccTake<P>{acc, cx});
    }
  }
  fold_range<P>(cx, l - 1);  // (all 16 units of layer l-1's output exist now)
  if constexpr (has_gates<P>()) {
    cx.gptr[(l - 1) * 256] = u32x4{cx.gbits[0], cx.gbits[1], cx.gbits[2], cx.gbits[3]};
    cx.gbits[0] = cx.gbits[1] = cx.gbits[2] = cx.gbits[3] = 0u;
  }
  if (l == 5) ipe_steps<P, false>(acc, cx, ipe_src);
  finish_layer<P>(acc, l, cx);
  if (l == 7) {  // the last pts layer: tap / density head read it from cx.hv (inside the layer loop's body: after the loop, next to the
                 // views K-loop, the register allocator spilled ~150 registers per tile)
    // (a pass without colour heads has no K-loop behind this point: it does the same after the layer loop, behind tap_prefetch)
    if (cx.rgb) {
      if (cx.tap == 7) dump_tap(7, cx);
      alpha_head(cx);
    }
  }
}

// Sums over the 32 lanes of each half wavefront of 32 rows of four values (v[4 row + e]), "reduce-scatter": every step pairs two rows,
// adds across a lane distance (16, 8, 4, 2, 1) and keeps one row of the pair on either side, so the number of live values halves each time
// -- 64 + 32 + 16 + 8 + 4 outputs at 2-3 instructions each plus nothing for the lanes that used to idle, against 5 DPP adds for each of
// the 128 values when every row is reduced on its own (nm_half_sum_dpp8).  Lane L ends up with the four sums of row L & 31.
//   distance 16: v_permlane16_swap (gfx950) exchanges odd rows of one register with even rows of the other, then one add;
//   distance 8 / 4: v_add_dpp row_mirror / row_half_mirror, two instructions per output with complementary bank masks writing one register;
//   distance 2 / 1: quad_perm adds of both rows and a select on the lane bit.
// (summation order differs from nm_half_sum_dpp8's tree: results agree to rounding)
__device__ __forceinline__ void dpp_pairs8(float (&o)[8], const float (&x)[8], const float (&y)[8], bool eight) {
  // o = lanes whose bank bit is clear: x + x[mirror partner]; set: y + y[partner]
  if (eight)
    asm("s_nop 1\n"
        "v_add_f32_dpp %0, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %1, %9, %9 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %10, %10 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %3, %11, %11 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %12, %12 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %5, %13, %13 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %14, %14 row_mirror row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %7, %15, %15 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %16, %16 row_mirror row_mask:0xf bank_mask:0xc\n\tv_add_f32_dpp %1, %17, %17 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %18, %18 row_mirror row_mask:0xf bank_mask:0xc\n\tv_add_f32_dpp %3, %19, %19 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %20, %20 row_mirror row_mask:0xf bank_mask:0xc\n\tv_add_f32_dpp %5, %21, %21 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %22, %22 row_mirror row_mask:0xf bank_mask:0xc\n\tv_add_f32_dpp %7, %23, %23 row_mirror row_mask:0xf bank_mask:0xc"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]));
  else
    asm("s_nop 1\n"
        "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %12, %12 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %5, %13, %13 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %6, %14, %14 row_half_mirror row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %7, %15, %15 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %16, %16 row_half_mirror row_mask:0xf bank_mask:0xa\n\tv_add_f32_dpp %1, %17, %17 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %18, %18 row_half_mirror row_mask:0xf bank_mask:0xa\n\tv_add_f32_dpp %3, %19, %19 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %4, %20, %20 row_half_mirror row_mask:0xf bank_mask:0xa\n\tv_add_f32_dpp %5, %21, %21 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %6, %22, %22 row_half_mirror row_mask:0xf bank_mask:0xa\n\tv_add_f32_dpp %7, %23, %23 row_half_mirror row_mask:0xf bank_mask:0xa"
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
          "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]));
}
template <int XOR>
__device__ __forceinline__ void quad_add8(float (&v)[8]) {  // v[i] += v[i] of lane ^ XOR (1 or 2), in place, all lanes
  if (XOR == 1)
    asm("s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  else
    asm("s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
__device__ __forceinline__ f32x4 reduce_scatter_32rows(float (&v)[128], int lane) {
  float s1[64];  // rows 0..15 | (lanes 16..31 of the half: rows 16..31)
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[64 + i]), false, false);
    s1[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  float s2[32];  // 8 rows
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float o[8], x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = s1[8 * b + i]; y[i] = s1[32 + 8 * b + i]; }
    dpp_pairs8(o, x, y, true);
#pragma unroll
    for (int i = 0; i < 8; ++i) s2[8 * b + i] = o[i];
  }
  float s3[16];  // 4 rows
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float o[8], x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = s2[8 * b + i]; y[i] = s2[16 + 8 * b + i]; }
    dpp_pairs8(o, x, y, false);
#pragma unroll
    for (int i = 0; i < 8; ++i) s3[8 * b + i] = o[i];
  }
  float lo8[8], hi8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { lo8[i] = s3[i]; hi8[i] = s3[8 + i]; }
  quad_add8<2>(lo8); quad_add8<2>(hi8);
  float s4[8];  // 2 rows
#pragma unroll
  for (int i = 0; i < 8; ++i) s4[i] = (lane & 2) ? hi8[i] : lo8[i];
  float t8[8] = {s4[0], s4[1], s4[2], s4[3], s4[4], s4[5], s4[6], s4[7]};
  quad_add8<1>(t8);
  return (lane & 1) ? f32x4{t8[4], t8[5], t8[6], t8[7]} : f32x4{t8[0], t8[1], t8[2], t8[3]};
}

// Barriers of the tile's epilogue between tap_prefetch and the feature reduction: they order LDS traffic only (per-sample scratch), so they
// wait for LDS only -- a __syncthreads() is also a memory fence and would sit out the read-back that is meant to overlap this phase.
// End of the tile's last K-loop, on EVERY path into the epilogue: vmcnt(0) lgkmcnt(0) through the BUILTIN -- (a) this wavefront's last
// operand reads of the ring have returned; (b) the compiler sees its own LDS-DMA of the weight stream (the run-ahead into the blob's
// padding) retired; otherwise it keeps "an LDS write may be pending" on its books and puts a vmcnt(0) of its own in front of the next
// ds_read of the epilogue, which would then wait for the rows tap_prefetch requests right behind this.
#define NM_MLP_DONE_WAIT() __builtin_amdgcn_s_waitcnt(0x0070)
#define NM_EPI_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int P>
__device__ __forceinline__ void nerf_fwd_body(const NerfArgs& a) {
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
  int* const sm_lray = reinterpret_cast<int*>(sm + LDS_LEFT);  // [128] rays whose sample Sa is still to be evaluated
  float* const sm_lT = sm + LDS_LEFT + TILE;                   // [128] their transmittance after the first Sa samples
  unsigned* const sm_rng = reinterpret_cast<unsigned*>(sm + LDS_RNG);  // [NRANGE][256] range telemetry (fp16x3)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, hi = lane >> 5;
  // the zero-tail decision is taken HERE, from the flag nm_resample_ex left on the device (no promise by the caller)
  const bool tail_ok = a.left && !(a.tail_viol && *a.tail_viol != 0);
  const int S = a.S, R = a.R;                // S: row length of t / weights
  const int Sa = tail_ok ? a.Sa : S;         // samples evaluated by the regular tiles
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

  for (int i = tid; i < SMALL / 4; i += 256) reinterpret_cast<f32x4*>(sm_small)[i] = reinterpret_cast<const f32x4*>(a.blob)[i];
  if constexpr (P == 2) {
#pragma unroll
    for (int k = 0; k < NRANGE; ++k) sm_rng[k * 256 + tid] = 0u;
  }

  // persistent workgroups: one per CU (the LDS footprint allows no more), tiles dealt round robin.
  // NM_NERF_ZERO_TAIL: a regular tile evaluates samples 0..Sa-1 of its rays and queues (ray, transmittance) for the one
  // remaining non-degenerate sample (index Sa); whenever 128 of them have piled up, and at the end, a "leftover" pass
  // runs the same network over 128 queued samples (one per lane, each of a different ray) and adds their contribution
  // to the outputs of rays this workgroup has already written.
  int nleft = 0;  // queued leftovers (uniform)
  int bid = blockIdx.x;
#pragma unroll 1
  for (;;) {
  bool lo_pass = false;
  if (left && (nleft > TILE - 4 || (bid >= ntiles && nleft > 0))) lo_pass = true;
  else if (bid >= ntiles) break;
  const int nent = lo_pass ? nleft : 0;
  TRACE(0);
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
    if constexpr (P == 2) {  // (straight from the blob: sm_small may not have landed yet in the first tile)
      v *= reinterpret_cast<const float*>(a.blob)[OFF_INSCALE + (f < 27 ? 1 : 2)];
      __hip_atomic_fetch_max(sm_rng + 9 * 256 + tid, __float_as_uint(fabsf(v)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    sm_ex[tid] = v;
  }
  __syncthreads();  // biases are read before the first ring barrier
  TRACE(1);

  const int js = wave * 32 + s;
  const int rl = js / SP;
  // regular tile: lane's ray = slot js / SP of the tile; leftover pass: lane js owns queue entry js (idle lanes redo entry 0)
  const int ray = lo_pass ? (js < nent ? sm_lray[js] : R) : bid * nr + rl;
  const int rc = lo_pass ? sm_lray[js < nent ? js : 0] : (ray < R ? ray : R - 1);
  const float* rp = a.rays + (size_t)rc * 12;
#if NM_ABL & 1024  // (timing only: no ray / fence-post loads at the start of a tile)
  const float o0 = a.var_scale, o1 = 0.1f, o2 = 0.2f, d0 = 0.3f, d1 = 0.4f, d2 = 0.5f + a.var_scale, radius = 0.001f;
#else
  const float o0 = rp[0], o1 = rp[1], o2 = rp[2], d0 = rp[3], d1 = rp[4], d2 = rp[5], radius = rp[11];
#endif
  const float dsq0 = d0 * d0, dsq1 = d1 * d1, dsq2 = d2 * d2;
  const float dmag = fmaxf(1e-10f, (dsq0 + dsq1) + dsq2);
  const float dnorm = sqrtf((dsq0 + dsq1) + dsq2);
  const float nul0 = 1.0f - dsq0 / dmag, nul1 = 1.0f - dsq1 / dmag, nul2 = 1.0f - dsq2 / dmag;

  float red_acc = 0.f;
  float carryT = 1.f;
  float best_w = -1.f;
  float feat_run = 0.f;  // thread t: running feature channel t of the (single) ray when S > 128

  const int nch = lo_pass ? 1 : nchunks;
  for (int chunk = 0; chunk < nch; ++chunk) {
    const int sidx = lo_pass ? Sa : chunk * TILE + (js % SP);
#if NM_ABL & 1024
    const float t0 = 2.0f + 0.01f * (float)sidx + a.var_scale, t1 = t0 + 0.01f;
#else
    const float t0 = a.t[(size_t)rc * (S + 1) + sidx];
    const float t1 = a.t[(size_t)rc * (S + 1) + sidx + 1];
#endif
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
#pragma unroll
    for (int g0 = 0; g0 < ring_ahead<P>(); ++g0) dma_slot<P>(blob_slots, g0, ring, wave, lane);

    // ---- integrated positional encoding -> B operands of the 6 IPE K-steps, parked in LDS -------------------------
    // K-slot (step m, half h, i) <-> encoding index f = 45 h + 8 m + i of the reference's order (8 m + i < 45; the last three slots of step 5
    // are padding), f = part*45 + scale*3 + axis  (part 0: sin(2^scale x), part 1: sin(2^scale x + pi/2)): the two halves of a wavefront
    // evaluate the SAME (scale, axis) and differ in the phase only, so scale, axis and the exponential's constant are compile-time per
    // value and nothing is selected on the lane's half (round 4: with f = 16 m + 8 h + i the compiler turned the per-half selects into
    // run-time integer arithmetic on f -- 43 instructions per value, 28 now; nerf_pack_split permutes the weight columns to match).
    // Every lane evaluates only the 48 encodings its wavefront half feeds to the MFMAs, directly in fp32: the argument
    // 2^scale * x is exact, sin32 (4-term Cody-Waite + degree-9 polynomial, |err| <= 1e-7 for |arg| < 6.5e4) replaces the
    // earlier fp64 angle-doubling recurrence (which both halves had to run over all 90 values), and the second half of
    // the encoding takes sin(fl32(arg + fl32(pi/2))) literally like the reference (x + 0.f is x).
    {
      float* dst = sm_ipe + wave * (XS * 2 * 64 * 4) + lane * 4;
      const float ipe_scale = sm_small[OFF_INSCALE];
      const float phl = hi ? 1.57079637050628662109375f : 0.f;
#pragma unroll
      for (int m = 0; m < XS; ++m) {
        float v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int idx = 8 * m + i;                     // compile time
          const bool live = idx < 45;
          const int ax = (live ? idx : 0) % 3, sb = (live ? idx : 0) / 3;
          const float mu = mean[ax], vr = var[ax];
          const float sc = (float)(1 << sb);
          const float xe = mu * sc;
#if NM_ABL & 256
          float v = xe + phl;  // (timing only: no sine / exponential in the positional encoding)
#elif NM_IPE_EXACT == 1
          float v = expf(-0.5f * (vr * (sc * sc))) * nm_sinf(xe + phl);
#elif NM_IPE_EXACT == 2  // (study: exact sine, fast exponential)
          float v = __builtin_amdgcn_exp2f((-0.5f * (vr * (sc * sc))) * 1.44269504088896340736f) * nm_sinf(xe + phl);
#elif NM_IPE_EXACT == 3  // (study: fast sine, exact exponential)
          float v = expf(-0.5f * (vr * (sc * sc))) * sin32(xe + phl);
#else
          float v = __builtin_amdgcn_exp2f((-0.5f * (vr * (sc * sc))) * 1.44269504088896340736f) * sin32(xe + phl);
#endif
          if constexpr (P == 2) v *= ipe_scale;  // 2^c_ipe (|v| <= 1: no saturation possible for c_ipe <= 15)
          v8[i] = live ? v : 0.f;
        }
        if constexpr (is_split<P>()) {
          bf16x8 h8, l8;
          split8_p<P>(v8, h8, l8);
          *reinterpret_cast<u32x4*>(dst + (m * 2 + 0) * 256) = __builtin_bit_cast(u32x4, h8);
          *reinterpret_cast<u32x4*>(dst + (m * 2 + 1) * 256) = __builtin_bit_cast(u32x4, l8);
        } else {
          *reinterpret_cast<u32x4*>(dst + m * 256) = __builtin_bit_cast(u32x4, pack8_f16(v8));
        }
      }
    }

    TRACE(2);
    // ---- 8 pts layers + views layer (feature_linear folded in at pack time), software pipelined across layers ----------
    // The finished layer is moved out of the accumulators (finish_layer: AGPRs -> cx.hv, plus unit 0) and re-packed one
    // K-step unit at a time INSIDE the K-loop of the layer that consumes it: unit u+1 (bias, relu, hi/lo split = ~40 VALU
    // instructions) is computed in the shadow of the second-half MFMAs of K-step u (UnitWork, slot_step8).
    Ctx cx;
    cx.blob_slots = blob_slots; cx.ring = ring; cx.sm_small = sm_small;
    cx.tapw = reinterpret_cast<f32x4*>(a.ws) + ((size_t)blockIdx.x * 4 + wave) * 32 * 64 + lane;
    cx.nslots = nslots; cx.wave = wave; cx.lane = lane; cx.hi = hi; cx.tap = need_tap ? tap : -1; cx.g = 0; cx.sig_part = 0.f;
    cx.vmax = 0.f; cx.rng = sm_rng + tid; cx.sc = 1.f;
    cx.tap_pref = need_tap && !lo_pass; cx.rgb = need_rgb;
    cx.tap_ring = ring + wave * SLOT_FLOATS; cx.tap_ipe = sm_ipe + wave * (XS * 2 * 64 * 4);
    if constexpr (is_split<P>()) {
      // slots 0 and 1 landed (2 and 3 may stay in flight until the barrier of K-step 1), everybody's pieces: barrier
      NM_WAIT_VMCNT(8);
      __builtin_amdgcn_s_barrier();
    } else {
      ring_acquire<P>(blob_slots, 0, nslots, ring, wave, lane);
    }
    load_half<P>(cx.opA, ring, lane, 0);
    if constexpr (P == 1) load_half<1>(cx.opB, ring, lane, 1);
    const float* ipe_src = sm_ipe + wave * (XS * 2 * 64 * 4) + lane * 4;
    f32x16 acc[8];
    ipe_steps<P, true>(acc, cx, ipe_src);  // layer 0
    finish_layer<P>(acc, 0, cx);
    TRACE(3);
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
      layer_pass<P>(acc, l, cx, ipe_src);
      TRACE(3 + l);
    }
    float c_r = 0.f, c_g = 0.f, c_b = 0.f;
    if (!need_rgb) {
      if (cx.tap == 7) dump_tap(7, cx);  // (in front of the wait: its stores are retired before the rows are asked back)
      NM_MLP_DONE_WAIT();
      if (cx.tap_pref) tap_prefetch<0>(cx);
      alpha_head(cx);
      if (cx.tap_pref) tap_prefetch<1>(cx);
    } else {
      // ---- views layer + rgb head.  Input: layer 7's activations (cx.hv, bias + relu like any pts layer) through the PRODUCT
      // views_w[:, :256] . feature_w that nerf_pack_split forms (feature_linear is linear: one 128 x 256 map instead of a 256 x 256
      // layer followed by a 128 x 256 one), then the direction / appearance columns ------------------------------------------
      const int hh = launder(lane) >> 5;
      const float* exr = sm_ex + launder(rl) * 48 + 8 * hh;  // K-slot (step e, half h, i) <-> extra input 16 e + 8 h + i
      f32x16 av[4];
      views_hidden<P>(av, cx);
      fold_range<P>(cx, 7);  // (layer 7's output is re-packed by the views layer's K-loop)
      bf16x8 exh[VS], exl[VS];
#pragma unroll
      for (int e = 0; e < VS; ++e) {
        float v8[8];
        if (!lo_pass) {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(exr + 16 * e), e1 = *reinterpret_cast<const f32x4*>(exr + 16 * e + 4);
          v8[0] = e0[0]; v8[1] = e0[1]; v8[2] = e0[2]; v8[3] = e0[3]; v8[4] = e1[0]; v8[5] = e1[1]; v8[6] = e1[2]; v8[7] = e1[3];
        } else {
          // leftover pass: every lane has its own ray, so the per-slot table does not apply; same formulas, in registers
          const float* rq = a.rays + (size_t)launder(rc) * 12 + 8;
          const float vd0 = rq[0], vd1 = rq[1], vd2 = rq[2];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int f = 16 * e + 8 * hh + i;
            const int ax = f % 3;
            const float dax = ax == 0 ? vd0 : ax == 1 ? vd1 : vd2;
            float v = 0.f;
            if (f < 24) {
              const float xe = dax * (float)(1 << ((f % 12) / 3));
              v = nm_sinf(f < 12 ? xe : xe + 1.57079637050628662109375f);
            } else if (f < 27) {
              v = dax;  // f - 24 == f % 3
            } else if (f < 43) {
              v = a.app_row ? a.app_row[f - 27] : 0.f;
            }
            if constexpr (P == 2) v *= sm_small[OFF_INSCALE + (f < 27 ? 1 : 2)];
            v8[i] = v;
          }
        }
        if constexpr (is_split<P>()) split8_p<P>(v8, exh[e], exl[e]);
        else exh[e] = exl[e] = pack8_f16(v8);
      }
      views_extras<P>(av, cx, exh, exl);
      TRACE(11);
      NM_MLP_DONE_WAIT();
      if (cx.tap_pref) tap_prefetch<0>(cx);
      const float* bv = sm_small + OFF_BVIEWS + 4 * hh;
      const float* wr = sm_small + OFF_WRGB + 4 * hh;
      float pr = 0.f, pg = 0.f, pb = 0.f;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + ob * 32 + 8 * q);
          const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wr + ob * 32 + 8 * q);
          const f32x4 wg4 = *reinterpret_cast<const f32x4*>(wr + 128 + ob * 32 + 8 * q);
          const f32x4 wb4 = *reinterpret_cast<const f32x4*>(wr + 256 + ob * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hv = __builtin_fmaxf(acc_read(av[ob][4 * q + e]) + b4[e], 0.f);
            pr = NM_FMA(hv, wr4[e], pr);
            pg = NM_FMA(hv, wg4[e], pg);
            pb = NM_FMA(hv, wb4[e], pb);
          }
        }
      }
      pr = (pr + nm_shfl_xor32(pr)) + sm_small[OFF_MISC + 1];
      pg = (pg + nm_shfl_xor32(pg)) + sm_small[OFF_MISC + 2];
      pb = (pb + nm_shfl_xor32(pb)) + sm_small[OFF_MISC + 3];
      c_r = 1.0f / (1.0f + expf(-pr));
      c_g = 1.0f / (1.0f + expf(-pg));
      c_b = 1.0f / (1.0f + expf(-pb));
      if (cx.tap_pref) tap_prefetch<1>(cx);
    }
    const float sigma_raw = (cx.sig_part + nm_shfl_xor32(cx.sig_part)) + sm_small[OFF_MISC];
    TRACE(12);
    {
      const int jsw = launder(js);
      if ((launder(lane) >> 5) == 0) {
        sm_sigma[jsw] = sigma_raw;
        sm_rgb[jsw] = c_r; sm_rgb[TILE + jsw] = c_g; sm_rgb[2 * TILE + jsw] = c_b;
      }
    }
    NM_EPI_BARRIER();
    TRACE(13);
    if constexpr ((NM_ABL & 128) != 0) continue;  // (timing only: no compositing / feature read-back / reductions / stores -- the epilogue's share of a tile)
    const int tid2 = launder(threadIdx.x), lane2 = tid2 & 63, wave2 = tid2 >> 6;

    if (lo_pass) {
      // ---- leftover pass: lane tid2 < nent is sample Sa of ray sm_lray[tid2]; its weight is alpha * T(first Sa samples) and
      // its contributions are ADDED (atomics: they execute at L2, where this workgroup's earlier plain stores are)
      if (tid2 < nent) {
        const int ray2 = sm_lray[tid2];
        const float sg = fmaxf(sm_sigma[tid2], 0.f);
        const float delta = (sm_t1[tid2] - sm_t0[tid2]) * sm_dn[tid2];
        const float wgt = (1.0f - expf(-sg * delta)) * sm_lT[tid2];
        sm_w[tid2] = wgt;
        a.weights[(size_t)ray2 * S + Sa] = wgt;
        if (a.acc) atomicAdd(a.acc + ray2, wgt);
        if (a.rgb && need_rgb) {
#pragma unroll
          for (int c = 0; c < 3; ++c) atomicAdd(a.rgb + (size_t)ray2 * 3 + c, a.white_bg ? wgt * sm_rgb[c * TILE + tid2] - wgt : wgt * sm_rgb[c * TILE + tid2]);
        }
        if (a.depth) atomicAdd(a.depth + ray2, wgt * (0.5f * (sm_t0[tid2] + sm_t1[tid2])));
        if (a.pts) {
#pragma unroll
          for (int c = 0; c < 3; ++c) atomicAdd(a.pts + (size_t)ray2 * 3 + c, wgt * sm_mean[c * TILE + tid2]);
        }
      }
      __syncthreads();
      if (need_tap) {
        const int jl = launder(js), hl = launder(lane) >> 5;
        const f32x4* tw = reinterpret_cast<const f32x4*>(a.ws) + ((size_t)blockIdx.x * 4 + wave2) * 32 * 64 + lane2;
        if (jl < nent) {
          const int ray2 = sm_lray[jl];
          const float desc = sm_small[OFF_DESCALE + tap];  // (workspace values carry the next layer's input scale)
          const float wj = sm_w[jl] * desc;
#pragma unroll 4
          for (int ks = 0; ks < HS; ++ks) {
            const f32x4 ta = tw[(2 * ks) * 64], tb = tw[(2 * ks + 1) * 64];
            const int n0 = (ks >> 1) * 32 + 16 * (ks & 1) + 4 * hl;  // neurons n0 .. n0+3 and n0+8 .. n0+11
            if (a.sfeat) {
              float* dsf = a.sfeat + ((size_t)ray2 * S + Sa) * 256 + n0;
              *reinterpret_cast<f32x4*>(dsf) = ta * desc;
              *reinterpret_cast<f32x4*>(dsf + 8) = tb * desc;
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

    // ---- alpha compositing (identical to nerf_fwd.hip) ---------------------------------------------------------------
    float alpha = 0.f, incl = 1.f;
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
    NM_EPI_BARRIER();
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
    NM_EPI_BARRIER();

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
    if (feat_max) NM_EPI_BARRIER();

    TRACE(14);
    // ---- feature output: weighted sum over the 32 samples of this wavefront straight from registers ------------------
    if (need_tap) {
      const int jl = launder(js), hl = launder(lane) >> 5;
      f32x4 tapv[2 * HS];
      {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's DMA rows have landed (nobody else reads them)
        // the four rows that did not fit: ordinary loads, issued now, used by the last two units (their latency sits behind the DPP work
        // of the first fourteen).  Not earlier: a compiler-tracked load in flight turns every wait on the way here into a wait for the DMA.
        f32x4 tail[4];
        {
          const f32x4* tw = reinterpret_cast<const f32x4*>(a.ws) + ((size_t)blockIdx.x * 4 + (launder(threadIdx.x) >> 6)) * 32 * 64 + (launder(threadIdx.x) & 63);
#pragma unroll
          for (int i = 0; i < 4; ++i) tail[i] = tw[(28 + i) * 64];
        }
        const float* lr = ring + (launder(threadIdx.x) >> 6) * SLOT_FLOATS + (launder(threadIdx.x) & 63) * 4;
        const float* li = sm_ipe + (launder(threadIdx.x) >> 6) * (XS * 2 * 64 * 4) + (launder(threadIdx.x) & 63) * 4;
#pragma unroll
        for (int c = 0; c < 2 * HS; ++c)
          tapv[c] = c < 16 ? *reinterpret_cast<const f32x4*>(lr + c * 256) : c < 28 ? *reinterpret_cast<const f32x4*>(li + (c - 16) * 256) : tail[c - 28];
      }
      const float desc = sm_small[OFF_DESCALE + tap];             // back to true units (1 unless fp16x3): folded into the weight
      const float wj = sm_w[jl] * desc;
      const int rsel = jl / SP;                                   // ray slot of this lane's sample
      const int best = feat_max ? __float_as_int(sm_misc[8 + rsel]) : -2;
      float* prow = sm_feat + (jl >> 5) * 256 + 4 * hl;           // partial sums of this wavefront
      if (a.sfeat && ray < R) {
#pragma unroll
        for (int ks = 0; ks < HS; ++ks) {
          const f32x4 ta = tapv[2 * ks], tb = tapv[2 * ks + 1];
          float* dsf = a.sfeat + ((size_t)ray * S + sidx) * 256 + (ks >> 1) * 32 + 16 * (ks & 1) + 4 * hl;
          *reinterpret_cast<f32x4*>(dsf) = ta * desc;
          *reinterpret_cast<f32x4*>(dsf + 8) = tb * desc;
        }
      }
      if (a.feat) {
        float wv[128];
        // one multiplier per lane (the sample's weight; feat_comb max: the descale for the selected sample, 0 for the others -- the tapped
        // activations are finite and >= 0, so 0 * v is the 0.f the select produced): a run-time select per VALUE cost 270 v_cndmask here
        const float mult = feat_max ? (jl == best ? desc : 0.f) : wj;
#pragma unroll
        for (int c = 0; c < 2 * HS; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[4 * c + e] = mult * tapv[c][e];
        const f32x4 sum4 = reduce_scatter_32rows(wv, jl);
        // lane (half hl, r = jl & 31) holds row r: K-step unit r >> 1, second quad if r & 1 -> neurons 32 (r >> 2) + 16 ((r >> 1) & 1) + 8 (r & 1) + 4 hl + 0..3
        const int r = jl & 31;
        *reinterpret_cast<f32x4*>(prow + (r >> 2) * 32 + 16 * ((r >> 1) & 1) + 8 * (r & 1)) = sum4;
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
  TRACE(17);
  if (left) nleft += (R - bid * nr) < nr ? (R - bid * nr) : nr;
  bid += gridDim.x;
  }  // tile loop
  if constexpr (P == 2) {
    // range telemetry / saturation flag: once per workgroup lifetime.  A re-packed value AT the fp16 limit (v_med3 clamps there)
    // means some operand of this launch was saturated: status[0] |= 1, which the caller turns into a re-run on the fp32 kernel
    // (nm_nerf_fwd_guarded reads the flag on the device) -- the fp16x3 path never returns silently clamped results.
    __syncthreads();
    if (a.status && tid < NRANGE) {
      unsigned m = 0u;
      for (int i = 0; i < 256; ++i) m = max(m, sm_rng[tid * 256 + ((i + 32 * tid) & 255)]);
      if (m) atomicMax(reinterpret_cast<unsigned*>(a.status) + 1 + tid, m);
      if (m >= __float_as_uint(F16_MAX)) atomicOr(a.status, 1);
    }
  }
}


// =====================================================================================================================
// Pointwise forward / backward of one NeRF MLP on the same K-loop machinery (round 4; the fine pass of the iNeRF refinement,
// nerfmatch/nerfmatch_evaluator.py:348-430 -- SURVEY.md section 8f rank 1).  The refinement needs d loss / d (ray origin, view
// direction) through the FINE network only, i.e. dX of every layer and no dW.  Both passes are pointwise over samples: the
// encodings (nm_inerf_encode) come in as rows, the compositing (nm_inerf_composite*) stays a kernel of its own, and what the
// backward needs from the forward is one BIT per activation (the ReLU gate) -- 9 x 16 bytes per sample lane instead of 8 KB of
// activations.  Arithmetic: the bf16 hi/lo split (three products, fp32 accumulate; gradients need the fp32 exponent range).
//
//   points_fwd (P = 4):  xi [n,96], xd [n,48]  ->  out4 [n,4] = (rgb logits, raw sigma),  gates [tiles][9][256] x 16 B
//       same blob and layer walk as the render kernel (nm_nerf_pack_bf16x3); gate table rows 0..7: layers 0..7 (8 bits per K-step
//       unit, gate_byte), row 8: the views layer (64 bits per lane: dword ob >> 1, bit 16 (ob & 1) + r)
//   points_bwd:  g4 [n,4] = d loss / d (logits, sigma),  gates  ->  g_xi0, g_xi5 [n,96] (layer 0 / skip connection parts), g_xd [n,48]
//       its own blob of TRANSPOSED weights (nm_nerf_pack_bwd_bf16x3), products in this order (K-steps x output blocks):
//       views^T -> xd (8 x 4), (views . feature_linear)^T -> h_7 (8 x 8: the folded matrix of the forward blob), pts 7^T, 6^T (16 x 8), pts 5^T -> IPE part (16 x 4),
//       pts 5^T, 4^T .. 1^T (16 x 8), pts 0^T -> IPE (16 x 4).  A finished product is copied out of the accumulators like in the forward
//       pass; re-packing a unit = AND with the sign-extended gate bit (v_bfe_i32 + v_and: two instructions per value, as bias + ReLU
//       were) + the hi/lo split, in the shadow of the consumer's MFMAs.
struct PointsArgs {
  const char* blob;
  const float* xi;   // fwd: [n,96];  bwd: unused
  const float* xd;   // fwd: [n,48]
  const float* g4;   // bwd: [n,4]
  float* out4;       // fwd: [n,4]
  float* g_xi0;      // bwd: [n,96]
  float* g_xi5;      // bwd: [n,96]
  float* g_xd;       // bwd: [n,48]
  u32x4* gates;      // [ntiles][9][256]
  int n, ntiles;
  const float* rays; // fwd, "from rays" form: [R,12]; then xi / xd are not read -- the kernel encodes its samples itself (nm_inerf_encode's formulas)
  const float* z;    //   fence posts [R, S + 1]; sample n = (ray n / Sa, interval n % Sa)
  const float* app_row;
  int S, Sa;
  float* dbg;        // debugging aid (scripts/debug_points_bwd.py): [n,256] <- cx.hv in neuron order after stage `dbg_stage` of the backward chain
  int dbg_stage;
  // the tapped layer (the matching term of the refinement, nerfmatch_evaluator.py:420-441; round 5)
  int tap;              // pts layer 0..7 whose post-ReLU activations are the rendered features; -1: none
  float* feats;         // fwd: [n,256] row-major <- those activations
  const float* tap_w;   // bwd: [n] compositing weights and
  const float* tap_g;   //      [n / Sa rays, 256] d loss / d pt_feat: d loss / d activation (n, c) += tap_w[n] * tap_g[n / Sa][c]
};
constexpr int NSLOT_BWD = 8 + 8 + 16 * 9;  // 160

// Post-ReLU activations of the finished pts layer lo (raw accumulators in cx.hv) -> row `dst_row` of a row-major [n,256] matrix:
// register 4 q + e of block ob is column 32 ob + 8 q + 4 half + e (the two half-wavefronts of a sample write adjacent 16 bytes).
// Once per tile, like dump_tap (whose workspace layout only the render kernel's own reduction reads).
__device__ __forceinline__ void dump_tap_rows(int lo, const Ctx& cx, float* dst_row, int hh, bool valid) {
  const float* bl = cx.sm_small + OFF_BIAS + lo * 256 + 4 * hh;
  auto* tp = (__attribute__((address_space(1))) f32x4*)(dst_row + 4 * hh);
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(cx.hv[ob * 16 + 4 * q + e] + b[e], 0.f);
      if (valid) tp[ob * 8 + q * 2] = v;
    }
}

template <int P, bool RAYS>
__device__ __forceinline__ void points_fwd_body(const PointsArgs& a) {
  __shared__ __attribute__((aligned(16))) float sm[LDS_SCR];  // small block, ring, IPE operands
  float* const sm_small = sm + LDS_SMALL;
  float* const ring = sm + LDS_RING;
  float* const sm_ipe = sm + LDS_IPE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, hi = lane >> 5;
  const char* const blob_slots = a.blob + (size_t)SMALL_PAD * 4;
  for (int i = tid; i < SMALL / 4; i += 256) reinterpret_cast<f32x4*>(sm_small)[i] = reinterpret_cast<const f32x4*>(a.blob)[i];
#pragma unroll 1
  for (int bid = blockIdx.x; bid < a.ntiles; bid += gridDim.x) {
    __syncthreads();  // small block landed / the previous tile is through with the LDS
    const int sample = bid * TILE + wave * 32 + s;
    const size_t sc = (size_t)(sample < a.n ? sample : a.n - 1);
#pragma unroll
    for (int g0 = 0; g0 < ring_ahead<P>(); ++g0) dma_slot<P>(blob_slots, g0, ring, wave, lane);
    float vdir[3] = {0.f, 0.f, 0.f};  // "from rays": this sample's view direction (the views layer's extra inputs are made from it below)
    {  // the 6 IPE K-steps' B operands: this lane's 8 columns per step
      float* dst = sm_ipe + wave * (XS * 2 * 64 * 4) + lane * 4;
      if constexpr (RAYS) {
        // encode here (round 4: saves nm_inerf_encode and the 144 floats per sample it writes): the formulas of nm_inerf_encode /
        // the reference's cast_rays + PositionalEncodingMIP (render_utils.py:326-402, embedding.py:66-84), exact sine and exponential
        const int r = (int)(sc / (size_t)a.Sa), si = (int)(sc % (size_t)a.Sa);
        const float* rp = a.rays + (size_t)r * 12;
        const float t0 = a.z[(size_t)r * (a.S + 1) + si], t1 = a.z[(size_t)r * (a.S + 1) + si + 1];
        const float d0 = rp[3], d1 = rp[4], d2 = rp[5], radius = rp[11];
        vdir[0] = rp[8]; vdir[1] = rp[9]; vdir[2] = rp[10];
        const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
        const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
        const float denom = fmaxf(1.1920928955078125e-07f, 3.0f * mu2 + hw2);
        const float t_mean = mu + (2.0f * mu * hw2) / denom;
        const float t_var = hw2 / 3.0f - (float)(4.0 / 15.0) * ((hw4 * (12.0f * mu2 - hw2)) / (denom * denom));
        const float r_var = (radius * radius) * ((mu2 / 4.0f + (float)(5.0 / 12.0) * hw2) - (float)(4.0 / 15.0) * hw4 / denom);
        const float dsq[3] = {d0 * d0, d1 * d1, d2 * d2};
        const float dmag = fmaxf(1e-10f, (dsq[0] + dsq[1]) + dsq[2]);
        float mean[3], var[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          mean[ax] = rp[ax] + t_mean * vdir[ax];  // (nm_inerf_encode: origin + t_mean * view direction; rays[:, 3:6] == rays[:, 8:11] there)
          var[ax] = t_var * dsq[ax] + r_var * (1.0f - dsq[ax] / dmag);
        }
        const float phl = hi ? 1.57079637050628662109375f : 0.f;
#pragma unroll
        for (int m = 0; m < XS; ++m) {
          float v8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int idx = 8 * m + i;  // K-slot (m, half, i) <-> encoding 45 half + idx (see nerf_fwd_body)
            const bool live = idx < 45;
            const int ax = (live ? idx : 0) % 3, sb = (live ? idx : 0) / 3;
            const float scl = (float)(1 << sb);
            const float xe = mean[ax] * scl;
            const float v = expf(-0.5f * (var[ax] * (scl * scl))) * nm_sinf(xe + phl);  // (x + 0.f is x)
            v8[i] = live ? v : 0.f;
          }
          bf16x8 h8, l8;
          split8_p<P>(v8, h8, l8);
          *reinterpret_cast<u32x4*>(dst + (m * 2 + 0) * 256) = __builtin_bit_cast(u32x4, h8);
          *reinterpret_cast<u32x4*>(dst + (m * 2 + 1) * 256) = __builtin_bit_cast(u32x4, l8);
        }
      } else {
        const float* row = a.xi + sc * 96 + 45 * hi;  // xi is in the reference's order: this half's part (sin | shifted sin) starts at 45 half
#pragma unroll
        for (int m = 0; m < XS; ++m) {
          float v8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = (8 * m + i) < 45 ? row[8 * m + i] : 0.f;
          bf16x8 h8, l8;
          split8_p<P>(v8, h8, l8);
          *reinterpret_cast<u32x4*>(dst + (m * 2 + 0) * 256) = __builtin_bit_cast(u32x4, h8);
          *reinterpret_cast<u32x4*>(dst + (m * 2 + 1) * 256) = __builtin_bit_cast(u32x4, l8);
        }
      }
    }
    Ctx cx;
    cx.blob_slots = blob_slots; cx.ring = ring; cx.sm_small = sm_small; cx.tapw = nullptr;
    cx.nslots = NSLOT_FULL; cx.wave = wave; cx.lane = lane; cx.hi = hi; cx.tap = -1; cx.g = 0; cx.sig_part = 0.f;
    cx.vmax = 0.f; cx.rng = nullptr; cx.sc = 1.f; cx.tap_pref = false; cx.rgb = true; cx.tap_ring = nullptr; cx.tap_ipe = nullptr;
    cx.gptr = a.gates + (size_t)bid * 9 * 256 + tid;
    cx.gbits[0] = cx.gbits[1] = cx.gbits[2] = cx.gbits[3] = 0u;
    NM_WAIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    load_half<P>(cx.opA, ring, lane, 0);
    const float* ipe_src = sm_ipe + wave * (XS * 2 * 64 * 4) + lane * 4;
    f32x16 acc[8];
    ipe_steps<P, true>(acc, cx, ipe_src);
    finish_layer<P>(acc, 0, cx);
    const bool tapped = a.feats != nullptr;
#pragma unroll 1
    for (int l = 1; l < 8; ++l) {
      if (tapped && l - 1 == a.tap) dump_tap_rows(l - 1, cx, a.feats + sc * 256, launder(lane) >> 5, sample < a.n);
      layer_pass<P>(acc, l, cx, ipe_src);
    }
    if (tapped && a.tap == 7) dump_tap_rows(7, cx, a.feats + sc * 256, launder(lane) >> 5, sample < a.n);
    // views layer: layer 7's activations through views . feature_linear (one matrix, see nerf_fwd_body) + this sample's xd row
    f32x16 av[4];
    views_hidden<P>(av, cx);
    cx.gptr[7 * 256] = u32x4{cx.gbits[0], cx.gbits[1], cx.gbits[2], cx.gbits[3]};  // layer 7's gates (collected by the K-loop above)
    const int hh = launder(lane) >> 5;
    {
      const float* row = a.xd + sc * 48 + 8 * hh;
      bf16x8 exh[VS], exl[VS];
#pragma unroll
      for (int e = 0; e < VS; ++e) {
        float v8[8];
        if constexpr (RAYS) {  // xd row of nm_inerf_encode: sin(2^k v), sin(2^k v + pi/2), v, appearance row, padding
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int f = 16 * e + 8 * hh + i;
            const int ax = f % 3;
            const float dax = ax == 0 ? vdir[0] : ax == 1 ? vdir[1] : vdir[2];
            float v = 0.f;
            if (f < 24) {
              const float xe = dax * (float)(1 << ((f % 12) / 3));
              v = nm_sinf(f < 12 ? xe : xe + 1.57079637050628662109375f);
            } else if (f < 27) {
              v = dax;
            } else if (f < 43) {
              v = a.app_row ? a.app_row[f - 27] : 0.f;
            }
            v8[i] = v;
          }
        } else {
          const f32x4 e0 = *reinterpret_cast<const f32x4*>(row + 16 * e), e1 = *reinterpret_cast<const f32x4*>(row + 16 * e + 4);
          v8[0] = e0[0]; v8[1] = e0[1]; v8[2] = e0[2]; v8[3] = e0[3]; v8[4] = e1[0]; v8[5] = e1[1]; v8[6] = e1[2]; v8[7] = e1[3];
        }
        split8_p<P>(v8, exh[e], exl[e]);
      }
      views_extras<P>(av, cx, exh, exl);
    }
    const float* bv = sm_small + OFF_BVIEWS + 4 * hh;
    const float* wr = sm_small + OFF_WRGB + 4 * hh;
    float pr = 0.f, pg = 0.f, pb = 0.f;
    unsigned gv0 = 0u, gv1 = 0u;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bv + ob * 32 + 8 * q);
        const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wr + ob * 32 + 8 * q);
        const f32x4 wg4 = *reinterpret_cast<const f32x4*>(wr + 128 + ob * 32 + 8 * q);
        const f32x4 wb4 = *reinterpret_cast<const f32x4*>(wr + 256 + ob * 32 + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float hv = __builtin_fmaxf(acc_read(av[ob][4 * q + e]) + b4[e], 0.f);
          const unsigned bit = min(__float_as_uint(hv), 1u) << (16 * (ob & 1) + 4 * q + e);  // hv >= 0: non-zero bits <=> hv > 0
          if (ob < 2) gv0 |= bit; else gv1 |= bit;
          pr = NM_FMA(hv, wr4[e], pr);
          pg = NM_FMA(hv, wg4[e], pg);
          pb = NM_FMA(hv, wb4[e], pb);
        }
      }
    cx.gptr[8 * 256] = u32x4{gv0, gv1, 0u, 0u};
    pr = (pr + nm_shfl_xor32(pr)) + sm_small[OFF_MISC + 1];
    pg = (pg + nm_shfl_xor32(pg)) + sm_small[OFF_MISC + 2];
    pb = (pb + nm_shfl_xor32(pb)) + sm_small[OFF_MISC + 3];
    const float sigma_raw = (cx.sig_part + nm_shfl_xor32(cx.sig_part)) + sm_small[OFF_MISC];
    if (hh == 0 && sample < a.n) *reinterpret_cast<f32x4*>(a.out4 + (size_t)sample * 4) = f32x4{pr, pg, pb, sigma_raw};
  }
}

// ---- backward -----------------------------------------------------------------------------------------------------------
// Unit u of the gradient held in cx.hv (registers 8m .. 8m+7 of block u >> 1), multiplied by its ReLU gate (GATED: byte u & 3 of
// gw[u >> 2], gate_byte's bit order) and split into bf16 hi / lo -- same 12 pieces as UnitWork, same placement rules.
template <bool GATED>
struct UnitWorkB {
  Ctx& cx;
  Unit& out;
  int u;
  u32x4 gw;
  float v8[8];
  float f0, f1;
  __device__ __forceinline__ void prefetch() {}
  __device__ __forceinline__ void operator()(int j) {
    const int ob = u >> 1, m = u & 1;
    if (j < 4) {  // elements j and 4 + j
      float a0 = cx.hv[ob * 16 + 8 * m + j], a1 = cx.hv[ob * 16 + 8 * m + 4 + j];
      if constexpr (GATED) {
        const unsigned w = gw[u >> 2];
        const int base = 8 * (u & 3);
        // element e: bit e / 2 (e even) or 4 + e / 2 (e odd)
        const int e0 = j, e1 = 4 + j;
        const int b0 = base + ((e0 & 1) ? 4 + (e0 >> 1) : (e0 >> 1)), b1 = base + ((e1 & 1) ? 4 + (e1 >> 1) : (e1 >> 1));
        a0 = __int_as_float(__float_as_int(a0) & __builtin_amdgcn_sbfe((int)w, b0, 1));
        a1 = __int_as_float(__float_as_int(a1) & __builtin_amdgcn_sbfe((int)w, b1, 1));
      }
      v8[j] = a0; v8[4 + j] = a1;
      pin(v8[j]); pin(v8[4 + j]);
    } else if (!(j & 1)) {
      const int p = (j - 4) >> 1;
      unsigned hp = pack_bf16(v8[2 * p], v8[2 * p + 1]);
      f0 = __uint_as_float(hp << 16);
      f1 = __uint_as_float(hp & 0xffff0000u);
      pin(hp); pin(f0); pin(f1);
      out.h[p] = hp;
    } else {
      const int p = (j - 5) >> 1;
      float r0 = v8[2 * p] - f0, r1 = v8[2 * p + 1] - f1;
      pin(r0); pin(r1);
      unsigned lp = pack_bf16(r0, r1);
      pin(lp);
      out.l[p] = lp;
    }
  }
};
template <bool GATED>
__device__ __forceinline__ UnitWorkB<GATED> unit_work_b(int u, Ctx& cx, Unit& out, const u32x4& gw) {
  return UnitWorkB<GATED>{cx, out, u, gw, {}, 0.f, 0.f};
}
template <bool GATED>
__device__ __forceinline__ void make_unit0_b(Ctx& cx, const u32x4& gw) {
  UnitWorkB<GATED> w = unit_work_b<GATED>(0, cx, cx.xn, gw);
#pragma unroll
  for (int j = 0; j < 12; ++j) w(j);
}
// one product of the backward chain: NKS K-steps x NOB output blocks on the units of cx.hv (unit 0 is in cx.xn)
template <int NOB, int NKS, bool GATED>
__device__ __forceinline__ void bwd_product(f32x16 (&acc)[NOB], Ctx& cx, const u32x4& gw) {
#pragma unroll
  for (int ks = 0; ks < NKS; ks += 2) {
    {
      const Unit xc = cx.xn;
      const bf16x8 xh = __builtin_bit_cast(bf16x8, xc.h), xl = __builtin_bit_cast(bf16x8, xc.l);
      if constexpr (NOB == 8) {
        if (ks == 0) slot_step8<0, true, true>(acc, cx, xh, xl, unit_work_b<GATED>(ks + 1, cx, cx.xn, gw));
        else slot_step8<0, false, true>(acc, cx, xh, xl, unit_work_b<GATED>(ks + 1, cx, cx.xn, gw));
      } else {
        if (ks == 0) slot_step4<0, true, true>(acc, cx, xh, xl, unit_work_b<GATED>(ks + 1, cx, cx.xn, gw));
        else slot_step4<0, false, true>(acc, cx, xh, xl, unit_work_b<GATED>(ks + 1, cx, cx.xn, gw));
      }
    }
    {
      const Unit xc = cx.xn;
      const bf16x8 xh = __builtin_bit_cast(bf16x8, xc.h), xl = __builtin_bit_cast(bf16x8, xc.l);
      if constexpr (NOB == 8) {
        if (ks + 2 < NKS) slot_step8<0, false, false>(acc, cx, xh, xl, unit_work_b<GATED>(ks + 2, cx, cx.xn, gw));
        else slot_step8<0, false, false>(acc, cx, xh, xl, NoWork{});
      } else {
        if (ks + 2 < NKS) slot_step4<0, false, false>(acc, cx, xh, xl, unit_work_b<GATED>(ks + 2, cx, cx.xn, gw));
        else slot_step4<0, false, false>(acc, cx, xh, xl, NoWork{});
      }
    }
  }
}
__device__ __forceinline__ void take_acc8(const f32x16 (&acc)[8], Ctx& cx) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) cx.hv[ob * 16 + r] = acc_read(acc[ob][r]);
}
// 4-block result (output column c = 32 ob + nrow(r, half)) -> rows of a [n, ld] matrix, columns < ncol
__device__ __forceinline__ void store_acc4(const f32x16 (&av)[4], float* dst_row, int ncol, int hh, bool valid) {
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 32 * ob + 8 * q + 4 * hh;  // columns c .. c + 3 = registers 4 q .. 4 q + 3
      const f32x4 v = {acc_read(av[ob][4 * q + 0]), acc_read(av[ob][4 * q + 1]), acc_read(av[ob][4 * q + 2]), acc_read(av[ob][4 * q + 3])};
      if (valid && c + 3 < ncol) *reinterpret_cast<f32x4*>(dst_row + c) = v;
    }
}

#ifndef NM_POINTS_DEBUG
#define NM_POINTS_DEBUG 0  // 1 (debug builds, scripts/debug_points_bwd.py): nm_nerf_points_bwd_bf16x3_dbg can dump cx.hv after a stage of the chain
#endif
__device__ __forceinline__ void dump_hv(const PointsArgs& a, const Ctx& cx, size_t sc, int hh, bool valid, int stage) {
#if NM_POINTS_DEBUG
  if (!a.dbg || a.dbg_stage != stage || !valid) return;
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) a.dbg[sc * 256 + 32 * ob + nrow(r, hh)] = cx.hv[ob * 16 + r];
#endif
}

__device__ __forceinline__ void points_bwd_body(const PointsArgs& a) {
  __shared__ __attribute__((aligned(16))) float sm[LDS_IPE];  // small block + ring
  float* const sm_small = sm + LDS_SMALL;
  float* const ring = sm + LDS_RING;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, hi = lane >> 5;
  const char* const blob_slots = a.blob + (size_t)SMALL_PAD * 4;
  for (int i = tid; i < SMALL / 4; i += 256) reinterpret_cast<f32x4*>(sm_small)[i] = reinterpret_cast<const f32x4*>(a.blob)[i];
#pragma unroll 1
  for (int bid = blockIdx.x; bid < a.ntiles; bid += gridDim.x) {
    __syncthreads();
    const int sample = bid * TILE + wave * 32 + s;
    const bool valid = sample < a.n;
    const size_t sc = (size_t)(valid ? sample : a.n - 1);
#pragma unroll
    for (int g0 = 0; g0 < ring_ahead<0>(); ++g0) dma_slot<0>(blob_slots, g0, ring, wave, lane);
    const u32x4* gt = a.gates + (size_t)bid * 9 * 256 + tid;
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(a.g4 + sc * 4);
    Ctx cx;
    cx.blob_slots = blob_slots; cx.ring = ring; cx.sm_small = sm_small; cx.tapw = nullptr;
    // (cx.g opaque: with a compile-time slot counter the fully unrolled first products had their 16 DMA source addresses precomputed
    //  at kernel entry, spilled, and reloaded -- scratch latency and a vmcnt(0) -- right behind the ring barrier of every K-step pair)
    cx.nslots = NSLOT_BWD; cx.wave = wave; cx.lane = lane; cx.hi = hi; cx.tap = -1; cx.g = launder_s(0); cx.sig_part = 0.f;
    cx.vmax = 0.f; cx.rng = nullptr; cx.sc = 1.f; cx.tap_pref = false; cx.rgb = true; cx.tap_ring = nullptr; cx.tap_ipe = nullptr; cx.gptr = nullptr;
    const int hh = launder(lane) >> 5;
    // d loss / d (views layer's post-ReLU activations) = gate . (W_rgb^T g_logit): this lane's 64 of the 128, in accumulator order
    {
      const u32x4 gv = gt[8 * 256];
      const float* wr = sm_small + OFF_WRGB + 4 * hh;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wr + ob * 32 + 8 * q);
          const f32x4 wg4 = *reinterpret_cast<const f32x4*>(wr + 128 + ob * 32 + 8 * q);
          const f32x4 wb4 = *reinterpret_cast<const f32x4*>(wr + 256 + ob * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float g = NM_FMA(wb4[e], g4[2], NM_FMA(wg4[e], g4[1], wr4[e] * g4[0]));
            const int bit = 16 * (ob & 1) + 4 * q + e;
            cx.hv[ob * 16 + 4 * q + e] = __int_as_float(__float_as_int(g) & __builtin_amdgcn_sbfe((int)gv[ob >> 1], bit, 1));
          }
        }
    }
    NM_WAIT_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    load_half<0>(cx.opA, ring, lane, 0);
    const u32x4 none = {0u, 0u, 0u, 0u};
    f32x16 acc[8];
    // views^T -> xd columns
    {
      f32x16 av[4];
      make_unit0_b<false>(cx, none);
      bwd_product<4, 8, false>(av, cx, none);
      store_acc4(av, a.g_xd + sc * 48, 48, hh, valid);
    }
    // (views_w[:, :256] . feature_w)^T -> layer 7's post-ReLU activations, + the density head's share
    make_unit0_b<false>(cx, none);
    bwd_product<8, 8, false>(acc, cx, none);
    take_acc8(acc, cx);
    {
      const float* wa = sm_small + OFF_WALPHA + 4 * hh;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 w4v = *reinterpret_cast<const f32x4*>(wa + ob * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) cx.hv[ob * 16 + 4 * q + e] = NM_FMA(w4v[e], g4[3], cx.hv[ob * 16 + 4 * q + e]);
        }
    }
    dump_hv(a, cx, sc, hh, valid, 2);
    // The rest of the chain as ONE loop body (a second inlined copy of the 16-step product made the allocator keep two accumulator sets
    // and spill): iteration l = 7 .. 0 consumes d loss / d (post-ReLU output of pts layer l) sitting in cx.hv, gates it with the bits of
    // the forward pass and multiplies by that layer's transposed weights.  Layers 5 and 0 first send their gated gradient through the IPE
    // columns (4 output blocks).
#pragma unroll 1
    for (int l = 7; l >= 0; --l) {
      const u32x4 gw = gt[l * 256];
      if (l == a.tap && a.tap_g) {
        // the matching term's gradient enters at the tapped layer's (post-ReLU) activations: pt_feat = sum_s w_s h_tap(s), so
        // d loss / d h_tap(n) += w_n . d loss / d pt_feat[ray]  (product, then sum: nm_inerf_ray_sums_bwd's g_feats + the residual of the GEMM chain)
        const float wn = valid ? a.tap_w[sc] : 0.f;
        const float* gr = a.tap_g + (sc / (size_t)a.Sa) * 256 + 4 * hh;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gr + ob * 32 + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) cx.hv[ob * 16 + 4 * q + e] = __fadd_rn(cx.hv[ob * 16 + 4 * q + e], __fmul_rn(wn, g[e]));
          }
      }
      if (l == 5 || l == 0) {
        f32x16 av[4];
        make_unit0_b<true>(cx, gw);
        bwd_product<4, 16, true>(av, cx, gw);
        store_acc4(av, (l == 5 ? a.g_xi5 : a.g_xi0) + sc * 96, 96, hh, valid);
        if (l == 0) break;
      }
      make_unit0_b<true>(cx, gw);
      bwd_product<8, 16, true>(acc, cx, gw);
      take_acc8(acc, cx);
      dump_hv(a, cx, sc, hh, valid, 10 - l);  // (debug builds: 2 after the folded views^T + density share, 3 after pts 7^T, 4 after pts 6^T)
    }
  }
}

__global__ void __launch_bounds__(256, 1) nerf_points_fwd_kernel(PointsArgs a) { points_fwd_body<4, false>(a); }
__global__ void __launch_bounds__(256, 1) nerf_points_fwd_rays_kernel(PointsArgs a) { points_fwd_body<4, true>(a); }
__global__ void __launch_bounds__(256, 1) nerf_points_bwd_kernel(PointsArgs a) { points_bwd_body(a); }

__global__ void __launch_bounds__(256, 1) nerf_fwd_bf16x3_kernel(NerfArgs a) { nerf_fwd_body<0>(a); }
__global__ void __launch_bounds__(256, 1) nerf_fwd_fp16x1_kernel(NerfArgs a) { nerf_fwd_body<1>(a); }
__global__ void __launch_bounds__(256, 1) nerf_fwd_fp16x3_kernel(NerfArgs a) { nerf_fwd_body<2>(a); }

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
// (fp16x1 blob: element (obo, lane, i) = fp16(W[...]) rounded to nearest even, 8 KiB per slot)
inline uint16_t f16_bits(float f) {  // round to nearest even (the host compiler's float -> _Float16 conversion), saturating
  f = f > 65504.0f ? 65504.0f : (f < -65504.0f ? -65504.0f : f);
  const _Float16 hf = (_Float16)f;
  uint16_t bits;
  memcpy(&bits, &hf, 2);
  return bits;
}
inline float f16_to_f(uint16_t b) {
  _Float16 hf;
  memcpy(&hf, &b, 2);
  return (float)hf;
}
// mode 0: bf16 hi / lo, 1: single fp16, 2: fp16 hi / lo
// (sc(c): power-of-two factor of input column c -- fp16x3 operand scaling, 1 otherwise; exact in fp32)
template <typename ColFn, typename ScFn>
void pack_slot(uint16_t* slot, const float* W, int ld, int nob, ColFn col, int mode, ScFn sc) {
  for (int obo = 0; obo < nob; ++obo)
    for (int ln = 0; ln < 64; ++ln)
      for (int i = 0; i < 8; ++i) {
        const int c = col(ln >> 5, i);
        const float w = c < 0 ? 0.f : W[(size_t)(32 * obo + (ln & 31)) * ld + c] * sc(c);
        if (mode == 1) {
          slot[(obo * 64 + ln) * 8 + i] = f16_bits(w);
          continue;
        }
        uint16_t h, l;
        if (mode == 2) {
          h = f16_bits(w);
          l = f16_bits(w - f16_to_f(h));
        } else {
          h = bf16_rne(w);
          l = bf16_rne(w - bf16_to_f(h));
        }
        slot[((obo * 2 + 0) * 64 + ln) * 8 + i] = h;
        slot[((obo * 2 + 1) * 64 + ln) * 8 + i] = l;
      }
}

// a slot of the paired views layer: blocks 0..3 take their columns from colA, blocks 4..7 are the SAME 128 output rows with colB
template <typename ColA, typename ColB, typename ScFn>
void pack_slot2(uint16_t* slot, const float* W, int ld, ColA colA, ColB colB, int mode, ScFn sc) {
  for (int obo = 0; obo < 8; ++obo)
    for (int ln = 0; ln < 64; ++ln)
      for (int i = 0; i < 8; ++i) {
        const int c = obo < 4 ? colA(ln >> 5, i) : colB(ln >> 5, i);
        const float w = c < 0 ? 0.f : W[(size_t)(32 * (obo & 3) + (ln & 31)) * ld + c] * sc(c);
        uint16_t h, l;
        if (mode == 2) {
          h = f16_bits(w);
          l = f16_bits(w - f16_to_f(h));
        } else {
          h = bf16_rne(w);
          l = bf16_rne(w - bf16_to_f(h));
        }
        slot[((obo * 2 + 0) * 64 + ln) * 8 + i] = h;
        slot[((obo * 2 + 1) * 64 + ln) * 8 + i] = l;
      }
}
constexpr int NSLOT_FULL_PAIRED = NSLOT_NORGB + HS / 2 + 2;  // 134: eight paired hidden slots, extras 0 | 1, extra 2

constexpr size_t BLOB_BYTES_FP16 = (size_t)SMALL_PAD * 4 + (size_t)(NSLOT_FULL + 8) * (SLOT_BYTES / 2);  // 8 >= ring_ahead<1>() + 1 slots of padding

}  // namespace

extern "C" size_t nm_nerf_blob_bytes_bf16x3(void) { return BLOB_BYTES; }
extern "C" size_t nm_nerf_workspace_bytes_bf16x3(void) { return (size_t)WS_WORKGROUPS * TILE * 256 * sizeof(float); }

// fp16x3 operand scaling (round 4).  An fp16 hi/lo pair carries 22 significant bits only while the lo part is a NORMAL fp16
// number, i.e. for |x| >~ 2^-3; below that the lo part is a subnormal with an absolute quantum of 2^-24 (U(+-1/16) weights: ~20
// bits, 2^-25 absolute each -- as much noise as the fp32 accumulation itself, scripts/fp16x3_scaling_study.py).  Powers of two
// commute with every rounding, so operands are moved into the middle of the fp16 range and the result is moved back exactly:
//   weights of layer l, input group g (hidden columns | IPE columns | direction PE | appearance):  W * 2^a(l,g), chosen HERE
//     from max|W| (-> [2^13, 2^14): constants cannot saturate);
//   inputs of layer l:  x * 2^c_l -- c_0 (IPE, |x| <= 1) and the direction PE are static (2^12); the hidden activations' c_l come from
//     the caller (act_log2: measured ranges, nm_nerf_fwd_fp16x3_ex status[]; NULL = 0, the unscaled activations of round 3);
//   accumulator of layer l:  2^A_l x the true pre-activation, A_l = a(l,g) + c(g) for every group g (the a's are tied by that);
//   re-packing:  fma(acc, 2^(c_{l+1} - A_l), bias * 2^c_{l+1})  (OFF_SCALE, OFF_BIAS), density head vector * 2^-c_8, rgb head: bias
//     * 2^A_9, vectors * 2^-A_9; tapped activations leave the kernel through OFF_DESCALE.
struct Fp16Scales {
  int a0 = 0, ah[10] = {0}, ax5 = 0, avd = 0, ava = 0;  // weight exponents: layer 0; hidden groups of layers 1..8 and views (9); layer 5's IPE columns; views' direction / appearance columns
  int c[12] = {0};                                      // input exponents: [0] IPE, [1..9] hidden input of layers 1..8 / views, [10] direction PE, [11] appearance
  int A[10] = {0};                                      // accumulator exponents of layers 0..8, views (9)
};
static float absmax_cols(const float* W, int rows, int ld, int c0, int c1) {
  float m = 0.f;
  for (int r = 0; r < rows; ++r)
    for (int c = c0; c < c1; ++c) m = fmaxf(m, fabsf(W[(size_t)r * ld + c]));
  return m;
}
static int weight_exp(float m) {  // a with m * 2^a in [2^13, 2^14)
#ifdef NM_NO_WSCALE
  return 0;  // (A/B builds only: the unscaled weights of round 3)
#endif
  if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
  int e;
  frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
  return 14 - e;
}
// feature_linear has no activation, so  views(cat[feature_linear(h), dir, app]) = (V_h F) h + V_d dir + V_a app + (V_h f_b + v_b):
// the 128 x 256 product V_h F and the folded bias are formed here in double precision and rounded to fp32 ONCE; the kernels never
// run feature_linear as a layer (65,536 of the 607,232 multiply-adds per sample).  Layout of the result: views_w's own
// [128][283 + app] with columns 0..255 replaced, so the packing code below reads it like views_w.
struct FoldedViews {
  float* W = nullptr;  // [128][ldv]
  float b[128];
  ~FoldedViews() { free(W); }
};
static int fold_views(const nmNerfWeights* w, FoldedViews& fv) {
  const int ldv = 283 + w->app_dim;
  fv.W = (float*)malloc((size_t)128 * ldv * sizeof(float));
  if (!fv.W) return NM_ERR_ARG;
  for (int n = 0; n < 128; ++n) {
    const float* vr = w->views_w + (size_t)n * ldv;
    for (int k = 0; k < 256; ++k) {
      double acc = 0.0;
      for (int j = 0; j < 256; ++j) acc += (double)vr[j] * (double)w->feat_w[(size_t)j * 256 + k];
      fv.W[(size_t)n * ldv + k] = (float)acc;
    }
    for (int c = 256; c < ldv; ++c) fv.W[(size_t)n * ldv + c] = vr[c];
    double bb = (double)w->views_b[n];
    for (int j = 0; j < 256; ++j) bb += (double)vr[j] * (double)w->feat_b[j];
    fv.b[n] = (float)bb;
  }
  return NM_OK;
}

static int choose_fp16_scales(const nmNerfWeights* w, const float* views_folded, const int* act_log2, Fp16Scales& sc) {
  sc.c[0] = 12; sc.c[10] = 12; sc.c[11] = 0;
#ifdef NM_NO_WSCALE
  sc.c[0] = sc.c[10] = 0;  // (A/B builds only: nothing scaled at all = the operands of round 3)
#endif
  if (act_log2) {
    for (int i = 0; i < 12; ++i) {
      if (act_log2[i] < -24 || act_log2[i] > 15) return NM_ERR_ARG;
      sc.c[i] = act_log2[i];
    }
    if (sc.c[0] > 15 || sc.c[10] > 15) return NM_ERR_ARG;  // |IPE|, |direction PE| <= 1 must stay below 65504
  }
  const int ldv = 283 + w->app_dim;
  sc.a0 = weight_exp(absmax_cols(w->pts_w[0], 256, 90, 0, 90));
  sc.A[0] = sc.a0 + sc.c[0];
  for (int l = 1; l < 8; ++l) {
    const float* W = w->pts_w[l];
    const int ld = l == 5 ? 346 : 256, col0 = l == 5 ? 90 : 0;
    sc.ah[l] = weight_exp(absmax_cols(W, 256, ld, col0, col0 + 256));
    sc.A[l] = sc.ah[l] + sc.c[l];
  }
  {  // layer 5: the IPE columns share the accumulator
    const int ideal = weight_exp(absmax_cols(w->pts_w[5], 256, 346, 0, 90));
    sc.ax5 = sc.A[5] - sc.c[0];
    if (sc.ax5 > ideal + 1) {  // would push the IPE columns beyond 2^15: lower the whole layer
      const int d = sc.ax5 - (ideal + 1);
      sc.ax5 -= d; sc.ah[5] -= d; sc.A[5] -= d;
    }
  }
  {  // views layer (folded): hidden = layer 7's activations, carried at 2^c[8] | direction PE | appearance
    sc.c[9] = sc.c[8];
    sc.ah[9] = weight_exp(absmax_cols(views_folded, 128, ldv, 0, 256));
    sc.A[9] = sc.ah[9] + sc.c[9];
    const int ideal_d = weight_exp(absmax_cols(views_folded, 128, ldv, 256, 283));
    const int ideal_a = w->app_dim ? weight_exp(absmax_cols(views_folded, 128, ldv, 283, ldv)) : 1 << 20;
    int d = 0;
    if (sc.A[9] - sc.c[10] > ideal_d + 1) d = sc.A[9] - sc.c[10] - (ideal_d + 1);
    if (sc.A[9] - sc.c[11] - d > ideal_a + 1) d = sc.A[9] - sc.c[11] - (ideal_a + 1);
    sc.ah[9] -= d; sc.A[9] -= d;
    sc.avd = sc.A[9] - sc.c[10];
    sc.ava = sc.A[9] - sc.c[11];
  }
  for (int i = 0; i < 10; ++i)
    if (sc.A[i] < -100 || sc.A[i] > 100) return NM_ERR_ARG;
  return NM_OK;
}

static int nerf_pack_split(const nmNerfWeights* w, void* blob_v, int fp16, const int* act_log2 = nullptr) {  // 0: bf16x3, 1: fp16x1, 2: fp16x3
  if (!w || !blob_v) return NM_ERR_ARG;
  for (int i = 0; i < 8; ++i)
    if (!w->pts_w[i] || !w->pts_b[i]) return NM_ERR_ARG;
  if (!w->alpha_w || !w->alpha_b || !w->feat_w || !w->feat_b || !w->views_w || !w->views_b || !w->rgb_w || !w->rgb_b)
    return NM_ERR_ARG;
  if (w->app_dim != 0 && w->app_dim != 16) return NM_ERR_UNSUPPORTED;
  FoldedViews fv;
  if (fold_views(w, fv) != NM_OK) return NM_ERR_ARG;
  Fp16Scales sc;  // all zero: the unscaled modes
  if (fp16 == 2) {
    const int rc = choose_fp16_scales(w, fv.W, act_log2, sc);
    if (rc != NM_OK) return rc;
  }
  auto p2 = [](int e) { return ldexpf(1.0f, e); };
  memset(blob_v, 0, fp16 == 1 ? BLOB_BYTES_FP16 : BLOB_BYTES);
  float* small = (float*)blob_v;
  // bias of layer l at the input scale of its consumer (c[l + 1]; layer 7 feeds the density head and the folded views layer);
  // row 8 of the bias table (feature_linear, before the fold) stays zero
  for (int l = 0; l < 8; ++l)
    for (int n = 0; n < 256; ++n) small[OFF_BIAS + l * 256 + n] = w->pts_b[l][n] * p2(sc.c[l + 1]);
  for (int n = 0; n < 128; ++n) small[OFF_BVIEWS + n] = fv.b[n] * p2(sc.A[9]);
  for (int n = 0; n < 256; ++n) small[OFF_WALPHA + n] = w->alpha_w[n] * p2(-sc.c[8]);
  for (int n = 0; n < 384; ++n) small[OFF_WRGB + n] = w->rgb_w[n] * p2(-sc.A[9]);
  small[OFF_MISC] = w->alpha_b[0];
  for (int c = 0; c < 3; ++c) small[OFF_MISC + 1 + c] = w->rgb_b[c];
  for (int l = 0; l < 16; ++l) small[OFF_SCALE + l] = l < 8 ? p2(sc.c[l + 1] - sc.A[l]) : 1.0f;
  for (int l = 0; l < 8; ++l) small[OFF_DESCALE + l] = p2(-sc.c[l + 1]);
  small[OFF_INSCALE + 0] = p2(sc.c[0]); small[OFF_INSCALE + 1] = p2(sc.c[10]); small[OFF_INSCALE + 2] = p2(sc.c[11]); small[OFF_INSCALE + 3] = 1.0f;

  uint16_t* slots = (uint16_t*)((char*)blob_v + (size_t)SMALL_PAD * 4);
  int g = 0;
  auto next = [&]() { return slots + (size_t)(g++) * ((fp16 == 1 ? SLOT_BYTES / 2 : SLOT_BYTES) / 2); };
  auto ipe_steps = [&](const float* W, int ld, int aexp) {
    const float f = p2(aexp);
    for (int m = 0; m < XS; ++m)
      pack_slot(next(), W, ld, 8, [&](int h, int i) { const int idx = 8 * m + i; return idx < 45 ? 45 * h + idx : -1; }, fp16, [&](int) { return f; });
  };
  auto hid_steps = [&](const float* W, int ld, int col0, int nob, int aexp) {
    const float f = p2(aexp);
    for (int ks = 0; ks < HS; ++ks)
      pack_slot(next(), W, ld, nob, [&](int h, int i) { return col0 + 32 * (ks >> 1) + nrow(8 * (ks & 1) + i, h); }, fp16, [&](int) { return f; });
  };
  for (int l = 0; l < 8; ++l) {  // kernel order: layer 0 = IPE steps; layer 5 = hidden steps, then the skip connection's IPE steps
    if (l == 0) ipe_steps(w->pts_w[0], 90, sc.a0);
    if (l != 0) hid_steps(w->pts_w[l], l == 5 ? 346 : 256, l == 5 ? 90 : 0, 8, sc.ah[l]);
    if (l == 5) ipe_steps(w->pts_w[5], 346, sc.ax5);
  }
  const int ldv = 283 + w->app_dim;
  const float fvh = p2(sc.ah[9]), fvd = p2(sc.avd), fva = p2(sc.ava);
  auto hid_col = [&](int ks, int h, int i) { return 32 * (ks >> 1) + nrow(8 * (ks & 1) + i, h); };
  auto ext_col = [&](int e, int h, int i) {
    const int f = 16 * e + 8 * h + i;
    if (f < 27) return 256 + f;
    if (f < 43 && w->app_dim) return 283 + (f - 27);
    return -1;
  };
  auto vsc = [&](int c) { return c < 256 ? fvh : c < 283 ? fvd : fva; };
  if (fp16 != 1) {
    // split modes: two K-steps of the 4-block layer per slot (slot_step4x2) -- blocks 0..3 = the layer's four output blocks for the first
    // K-step, blocks 4..7 = the same four for the second; the last extra K-step has a slot of its own (first half)
    for (int sl = 0; sl < HS / 2; ++sl)
      pack_slot2(next(), fv.W, ldv, [&](int h, int i) { return hid_col(2 * sl, h, i); }, [&](int h, int i) { return hid_col(2 * sl + 1, h, i); }, fp16, vsc);
    pack_slot2(next(), fv.W, ldv, [&](int h, int i) { return ext_col(0, h, i); }, [&](int h, int i) { return ext_col(1, h, i); }, fp16, vsc);
    pack_slot(next(), fv.W, ldv, 4, [&](int h, int i) { return ext_col(2, h, i); }, fp16, vsc);
    return g == NSLOT_FULL_PAIRED ? NM_OK : NM_ERR_ARG;
  }
  hid_steps(fv.W, ldv, 0, 4, sc.ah[9]);
  for (int e = 0; e < VS; ++e) pack_slot(next(), fv.W, ldv, 4, [&](int h, int i) { return ext_col(e, h, i); }, fp16, vsc);
  return g == NSLOT_FULL ? NM_OK : NM_ERR_ARG;
}

extern "C" int nm_nerf_pack_bf16x3(const nmNerfWeights* w, void* blob_v) { return nerf_pack_split(w, blob_v, 0); }
extern "C" int nm_nerf_pack_fp16x3(const nmNerfWeights* w, void* blob_v) { return nerf_pack_split(w, blob_v, 2); }
extern "C" int nm_nerf_pack_fp16x3_scaled(const nmNerfWeights* w, const int* act_log2, void* blob_v) { return nerf_pack_split(w, blob_v, 2, act_log2); }
extern "C" size_t nm_nerf_blob_bytes_fp16x1(void) { return BLOB_BYTES_FP16; }
extern "C" int nm_nerf_pack_fp16x1(const nmNerfWeights* w, void* blob_v) { return nerf_pack_split(w, blob_v, 1); }

extern "C" int nm_nerf_fwd_bf16x3(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                                  int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                                  float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                                  nmStream_t stream) {
  return nm_nerf_fwd_bf16x3_ex(blob, rays, t, app_row, R, S, tap_layer, white_bg, var_scale, flags, weights, feat, pts, rgb, depth, acc, raw,
                               sample_feat, workspace, nullptr, stream);
}

static int nerf_fwd_split(int mode, const void* blob, const float* rays, const float* t, const float* app_row, int R, int S, int tap_layer,
                          int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts, float* rgb, float* depth,
                          float* acc, float* raw, float* sample_feat, void* workspace, const int* zero_tail_violation, nmStream_t stream,
                          int* status = nullptr) {
  NM_CHECK_ARG(blob && rays && t && weights && R > 0 && S > 0);
  if (!(S == 32 || S == 64 || (S % 128) == 0)) return NM_ERR_UNSUPPORTED;
  if (tap_layer > 7) return NM_ERR_ARG;
  if ((feat || sample_feat) && !workspace) return NM_ERR_WORKSPACE;
  NerfArgs a;
  a.ws = (float*)workspace;
  a.blob = (const char*)blob; a.rays = rays; a.t = t; a.app_row = app_row;
  a.weights = weights; a.feat = feat; a.pts = pts; a.rgb = rgb; a.depth = depth; a.acc = acc; a.raw = raw; a.sfeat = sample_feat;
  a.R = R; a.S = S; a.tap = tap_layer; a.white_bg = white_bg; a.flags = flags; a.var_scale = var_scale;
  a.status = status;
  // NM_NERF_ZERO_TAIL: samples 0 .. S/2 are evaluated (S/2 by the regular tiles, sample S/2 by leftover passes)
  // (S/2 must itself be a supported row length: 32, 64, 128 or a multiple of 128)
  const bool zero_tail = (flags & NM_NERF_ZERO_TAIL) && (S == 64 || S == 128 || (S >= 256 && S % 256 == 0)) && !raw && !sample_feat &&
                         !(flags & NM_NERF_FEAT_MAX);
  a.Sa = zero_tail ? S / 2 : S;
  a.left = zero_tail ? 1 : 0;
  const int SP = a.Sa < TILE ? a.Sa : TILE, nr = TILE / SP;
  a.ntiles = (R + nr - 1) / nr;
  const int SPf = S < TILE ? S : TILE, nrf = TILE / SPf;
  a.ntiles_full = (R + nrf - 1) / nrf;
  a.tail_viol = zero_tail ? zero_tail_violation : nullptr;
  const int ncu = nm_stream_cus(stream);  // (a CU-partitioned stream runs one workgroup per CU of its partition)
  const int grid = a.ntiles < ncu ? a.ntiles : (ncu < WS_WORKGROUPS ? ncu : WS_WORKGROUPS);
  // (the two-wavefronts-per-SIMD experiment of round 2, 12 % slower, lives in scripts/variants/ now: DESIGN.md section 3.1c)
  if (mode == 1) nerf_fwd_fp16x1_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else if (mode == 2) nerf_fwd_fp16x3_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else nerf_fwd_bf16x3_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}

extern "C" int nm_nerf_fwd_bf16x3_ex(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                                     int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                                     float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                                     const int* zero_tail_violation, nmStream_t stream) {
  return nerf_fwd_split(0, blob, rays, t, app_row, R, S, tap_layer, white_bg, var_scale, flags, weights, feat, pts, rgb, depth, acc, raw,
                        sample_feat, workspace, zero_tail_violation, stream);
}

extern "C" int nm_nerf_fwd_fp16x1(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                                  int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                                  float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                                  const int* zero_tail_violation, nmStream_t stream) {
  return nerf_fwd_split(1, blob, rays, t, app_row, R, S, tap_layer, white_bg, var_scale, flags, weights, feat, pts, rgb, depth, acc, raw,
                        sample_feat, workspace, zero_tail_violation, stream);
}

extern "C" int nm_nerf_fwd_fp16x3(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                                  int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                                  float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                                  const int* zero_tail_violation, nmStream_t stream) {
  return nerf_fwd_split(2, blob, rays, t, app_row, R, S, tap_layer, white_bg, var_scale, flags, weights, feat, pts, rgb, depth, acc, raw,
                        sample_feat, workspace, zero_tail_violation, stream);
}

extern "C" int nm_nerf_fwd_fp16x3_ex(const void* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                                     int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                                     float* rgb, float* depth, float* acc, float* raw, float* sample_feat, void* workspace,
                                     const int* zero_tail_violation, int* status, nmStream_t stream) {
  return nerf_fwd_split(2, blob, rays, t, app_row, R, S, tap_layer, white_bg, var_scale, flags, weights, feat, pts, rgb, depth, acc, raw,
                        sample_feat, workspace, zero_tail_violation, stream, status);
}

// ---- pointwise forward / backward: host side -------------------------------------------------------------------------------
constexpr size_t BLOB_BYTES_BWD = (size_t)SMALL_PAD * 4 + (size_t)(NSLOT_BWD + NSLOT_PAD) * SLOT_BYTES;

extern "C" size_t nm_nerf_blob_bytes_bwd_bf16x3(void) { return BLOB_BYTES_BWD; }
extern "C" size_t nm_nerf_points_gate_bytes(int n) { return (size_t)((n + TILE - 1) / TILE) * 9 * 256 * 16; }

// Transposed weights of one MLP in the order points_bwd_body consumes them (see the comment above PointsArgs); small block as in
// nm_nerf_pack_bf16x3 (rgb / density head vectors).
extern "C" int nm_nerf_pack_bwd_bf16x3(const nmNerfWeights* w, void* blob_v) {
  if (!w || !blob_v) return NM_ERR_ARG;
  void* tmp = malloc(BLOB_BYTES);
  if (!tmp) return NM_ERR_ARG;
  const int rc = nerf_pack_split(w, tmp, 0);
  if (rc != NM_OK) { free(tmp); return rc; }
  memset(blob_v, 0, BLOB_BYTES_BWD);
  memcpy(blob_v, tmp, (size_t)SMALL_PAD * 4);
  free(tmp);
  uint16_t* slots = (uint16_t*)((char*)blob_v + (size_t)SMALL_PAD * 4);
  int g = 0;
  auto next = [&]() { return slots + (size_t)(g++) * (SLOT_BYTES / 2); };
  const int ldv = 283 + w->app_dim;
  // product: out[o] = sum_k in[k] * Wt(o, k); rows beyond n_out are zero; nks K-steps of 16 inputs, nob blocks of 32 outputs
  auto product = [&](int n_out, int n_in, int nob, auto wt) {
    float* T = (float*)calloc((size_t)32 * nob * n_in, sizeof(float));
    for (int o = 0; o < n_out; ++o)
      for (int k = 0; k < n_in; ++k) T[(size_t)o * n_in + k] = wt(o, k);
    for (int ks = 0; ks < n_in / 16; ++ks)
      pack_slot(next(), T, n_in, nob, [&](int h, int i) { return 32 * (ks >> 1) + nrow(8 * (ks & 1) + i, h); }, 0, [](int) { return 1.0f; });
    free(T);
  };
  const int nxd = 27 + w->app_dim;
  FoldedViews fv;
  if (fold_views(w, fv) != NM_OK) return NM_ERR_ARG;
  product(nxd, 128, 4, [&](int c, int n) { return w->views_w[(size_t)n * ldv + 256 + c]; });       // views^T -> xd
  product(256, 128, 8, [&](int k, int n) { return fv.W[(size_t)n * ldv + k]; });                   // (views . feature_linear)^T -> h_7
  for (int l = 7; l >= 6; --l) product(256, 256, 8, [&](int k, int n) { return w->pts_w[l][(size_t)n * 256 + k]; });
  product(90, 256, 4, [&](int f, int n) { return w->pts_w[5][(size_t)n * 346 + f]; });            // pts 5^T -> IPE columns
  product(256, 256, 8, [&](int k, int n) { return w->pts_w[5][(size_t)n * 346 + 90 + k]; });      // pts 5^T -> hidden columns
  for (int l = 4; l >= 1; --l) product(256, 256, 8, [&](int k, int n) { return w->pts_w[l][(size_t)n * 256 + k]; });
  product(90, 256, 4, [&](int f, int n) { return w->pts_w[0][(size_t)n * 90 + f]; });             // pts 0^T -> IPE
  return g == NSLOT_BWD ? NM_OK : NM_ERR_ARG;
}

extern "C" int nm_nerf_points_bwd_bf16x3_dbg(const void* blob_bwd, const float* g4, const void* gates, int n, float* g_xi0, float* g_xi5, float* g_xd,
                                             float* dbg, int dbg_stage, nmStream_t stream);
static int points_grid(int ntiles, nmStream_t stream) {
  const int ncu = nm_stream_cus(stream);
  return ntiles < ncu ? ntiles : ncu;
}

extern "C" int nm_nerf_points_fwd_bf16x3(const void* blob, const float* xi, const float* xd, int n, float* out4, void* gates, nmStream_t stream) {
  NM_CHECK_ARG(blob && xi && xd && out4 && gates && n > 0);
  PointsArgs a = {};
  a.blob = (const char*)blob; a.xi = xi; a.xd = xd; a.out4 = out4; a.gates = (u32x4*)gates; a.n = n; a.ntiles = (n + TILE - 1) / TILE;
  a.tap = -1;
  nerf_points_fwd_kernel<<<points_grid(a.ntiles, stream), 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}

extern "C" int nm_nerf_points_bwd_bf16x3(const void* blob_bwd, const float* g4, const void* gates, int n, float* g_xi0, float* g_xi5, float* g_xd,
                                         nmStream_t stream) {
  return nm_nerf_points_bwd_bf16x3_dbg(blob_bwd, g4, gates, n, g_xi0, g_xi5, g_xd, nullptr, 0, stream);
}

extern "C" int nm_nerf_points_bwd_bf16x3_dbg(const void* blob_bwd, const float* g4, const void* gates, int n, float* g_xi0, float* g_xi5, float* g_xd,
                                             float* dbg, int dbg_stage, nmStream_t stream) {
  NM_CHECK_ARG(blob_bwd && g4 && gates && g_xi0 && g_xi5 && g_xd && n > 0);
  PointsArgs a = {};
  a.blob = (const char*)blob_bwd; a.g4 = g4; a.gates = (u32x4*)const_cast<void*>(gates); a.g_xi0 = g_xi0; a.g_xi5 = g_xi5; a.g_xd = g_xd;
  a.n = n; a.ntiles = (n + TILE - 1) / TILE; a.dbg = dbg; a.dbg_stage = dbg_stage; a.tap = -1;
  nerf_points_bwd_kernel<<<points_grid(a.ntiles, stream), 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}

extern "C" int nm_nerf_points_fwd_rays_tap_bf16x3(const void* blob, const float* rays, const float* z, int R, int S, int S_act, const float* app_row,
                                                  int tap_layer, float* out4, void* gates, float* feats, nmStream_t stream) {
  NM_CHECK_ARG(blob && rays && z && out4 && gates && R > 0 && S > 0 && S_act > 0 && S_act <= S);
  NM_CHECK_ARG(feats ? (tap_layer >= 0 && tap_layer <= 7) : tap_layer == -1);
  PointsArgs a = {};
  a.blob = (const char*)blob; a.rays = rays; a.z = z; a.app_row = app_row; a.S = S; a.Sa = S_act; a.out4 = out4; a.gates = (u32x4*)gates;
  a.n = R * S_act; a.ntiles = (a.n + TILE - 1) / TILE; a.tap = tap_layer; a.feats = feats;
  nerf_points_fwd_rays_kernel<<<points_grid(a.ntiles, stream), 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}

extern "C" int nm_nerf_points_fwd_rays_bf16x3(const void* blob, const float* rays, const float* z, int R, int S, int S_act, const float* app_row,
                                              float* out4, void* gates, nmStream_t stream) {
  return nm_nerf_points_fwd_rays_tap_bf16x3(blob, rays, z, R, S, S_act, app_row, -1, out4, gates, nullptr, stream);
}

extern "C" int nm_nerf_points_bwd_tap_bf16x3(const void* blob_bwd, const float* g4, const void* gates, int R, int S_act, int tap_layer,
                                             const float* tap_weights, const float* g_pt_feat, float* g_xi0, float* g_xi5, float* g_xd,
                                             nmStream_t stream) {
  NM_CHECK_ARG(blob_bwd && g4 && gates && g_xi0 && g_xi5 && g_xd && R > 0 && S_act > 0 && tap_layer >= 0 && tap_layer <= 7 && tap_weights && g_pt_feat);
  PointsArgs a = {};
  a.blob = (const char*)blob_bwd; a.g4 = g4; a.gates = (u32x4*)const_cast<void*>(gates); a.g_xi0 = g_xi0; a.g_xi5 = g_xi5; a.g_xd = g_xd;
  a.n = R * S_act; a.ntiles = (a.n + TILE - 1) / TILE; a.Sa = S_act; a.tap = tap_layer; a.tap_w = tap_weights; a.tap_g = g_pt_feat;
  nerf_points_bwd_kernel<<<points_grid(a.ntiles, stream), 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}
