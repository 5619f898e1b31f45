// Backward of the encoder tail for FROZEN parameters, as ONE kernel (round 6) -- the mirror image of encoder_tail.hip:
//
//     forward:  a = xh + att . Wo^T;   u = W1 . LN2(a) + b1;   y = xh + W2 . gelu(u) + b2
//     backward: g1 = dy . W2;  du = g1 o gelu'(u);  g2 = du . W1;  d_a = LN2'(a; g2);  d_xh = dy + d_a;  d_att = d_a . Wo
//
// (reference: torch autograd through GenericEncoderLayer.forward_pre_norm, nerfmatch/modules/attention.py:229-241, as the iNeRF refinement's
// matching term runs it with the matcher's parameters frozen, nerfmatch_evaluator.py:429-441: input gradients only, no dW.)  Unfused this is
// three GEMM launches, nm_gelu_bwd, nm_layernorm_bwd and two elementwise adds per layer.  Same skeleton as the forward kernel: workgroup = 4
// wavefronts x 32 rows, every wavefront owns its rows through the chain of three 256 x 256 split-bf16 products, the accumulator layout of one
// product is the K order of the next (weights of products 1 and 2 packed with nm_linear_pack_perm_bf16x3 from the TRANSPOSED matrices), one
// 48-K-step weight stream through the 4-deep LDS ring.  What is new: two row tiles arrive in the accumulator layout during the second halves
// of products 0 and 1 (u for GELU', a for the LayerNorm backward: its statistics are recomputed, four xor-32 row reductions), d_xh leaves in
// the MIDDLE of the chain through an LDS transposition buffer of its own (the ring keeps streaming product 2's weights) with dy re-read in the
// coalesced store layout, d_att leaves at the end through the same buffer.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int ET_ROWS = 128;
constexpr int ET_D = 256;                 // model dim = inner dim = FFN hidden dim
constexpr int ET_NKS = ET_D / 16;         // K-steps per product
constexpr int ET_SLOT_BYTES = 8192;       // one K-step of one 128-column chunk (linear blob format)
constexpr int ET_STEP_FLOATS = 2 * ET_SLOT_BYTES / 4;  // both chunks of a K-step: 16 KiB
// K-steps of the weight stream in flight ahead of the matrix work.  Round 5 measured 6 (ring of 8, 128 KiB) against 3: 40.6 vs 38.2 us at 4800 rows,
// 276 vs 255 us at 153,600 -- the kernel already uses all 512 registers and the deeper bookkeeping spills 49 of them to scratch; the stream is not what a
// K-step waits for (profiles/r5_ab_encoder_tail_ahead.log).  The wait counts below are derived from ET_AHEAD for any depth.
constexpr int ET_AHEAD = 3;
constexpr int ET_RING = ET_AHEAD > 3 ? 8 : 4;    // ring positions (a power of two > ET_AHEAD)

struct TailArgs {
  const float* dy;      // gradient of the tail's output y [R, 256]
  const float* a_pre;   // a = xh + att . Wo^T, the input of LayerNorm 2 (saved by the forward pass)
  const float* u_pre;   // u = W1 . LN2(a) + b1, the input of the GELU (saved by the forward pass)
  const char* blob[3];  // W2^T (standard K order), W1^T, Wo^T (accumulator K order): [chunk][ks] slots of 8 KiB
  const float* gamma;
  float* d_att;
  float* d_xh;
  int R;
  float eps;
  // the forward kernel that keeps its intermediates (encoder_tail_save_kernel): blob = Wo (standard), W1, W2 (accumulator K order)
  const float *att, *xh, *beta, *b1, *b2;
  float *y, *a_out, *u_out;
};

__host__ __device__ __forceinline__ constexpr int nrow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// d/dv of the exact-erf GELU: Phi(v) + v phi(v), erf by Abramowitz & Stegun 7.1.26 as in the forward kernel (the exponential it needs IS
// exp(-v^2 / 2), the Gaussian of the second term)
__device__ __forceinline__ float gelu_grad(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(NM_FMA(0.3275911f, x, 1.0f));
  float p = NM_FMA(1.061405429f, t, -1.453152027f);
  p = NM_FMA(p, t, 1.421413741f);
  p = NM_FMA(p, t, -0.284496736f);
  p = NM_FMA(p, t, 0.254829592f);
  const float ex = __builtin_amdgcn_exp2f(-(x * x) * 1.44269504088896340736f);  // exp(-v^2 / 2)
  const float e = 1.0f - (p * t) * ex;
  return NM_FMA(v * 0.39894228040143267794f, ex, 0.5f * (1.0f + copysignf(e, v)));
}

// exact-erf GELU, erf by Abramowitz & Stegun 7.1.26: the forward kernel's (encoder_tail.hip), verbatim
__device__ __forceinline__ float gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(NM_FMA(0.3275911f, x, 1.0f));
  float p = NM_FMA(1.061405429f, t, -1.453152027f);
  p = NM_FMA(p, t, 1.421413741f);
  p = NM_FMA(p, t, -0.284496736f);
  p = NM_FMA(p, t, 0.254829592f);
  const float e = 1.0f - (p * t) * __builtin_amdgcn_exp2f(-(x * x) * 1.44269504088896340736f);
  return 0.5f * v * (1.0f + copysignf(e, v));
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return __builtin_bit_cast(unsigned, bf16x2{(__bf16)a, (__bf16)b}); }

// K-step g of the 48-step weight stream (product g / 16): both chunks, 4 x 1 KiB pieces per wavefront
__device__ __forceinline__ void dma_step(const TailArgs& a, int g, float* ring, int wave, int lane) {
  const char* blob = a.blob[g >> 4];
  const int ks = g & 15;
  float* dst0 = ring + (g & (ET_RING - 1)) * ET_STEP_FLOATS + wave * 512;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const char* base = blob + ((size_t)c * ET_NKS + ks) * ET_SLOT_BYTES + wave * 2048 + lane * 16;
    const auto* src = (const __attribute__((address_space(1))) void*)base;
    auto* dst = (__attribute__((address_space(3))) void*)(dst0 + c * (ET_SLOT_BYTES / 4));
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
  }
}

struct Unit {
  u32x4 h, l;
};

// A operands of one 128-column chunk of a K-step: 4 blocks x (hi, lo) = 8 x 16 bytes per lane
struct OpsC {
  u32x4 h[4], l[4];
};
__device__ __forceinline__ void read_chunk(OpsC& o, const float* step, int lane, int c) {
  const u32x4* s4 = reinterpret_cast<const u32x4*>(step) + lane;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    o.h[b] = s4[(c * 8 + b * 2 + 0) * 64];
    o.l[b] = s4[(c * 8 + b * 2 + 1) * 64];
  }
}
// 12 MFMAs of one chunk (accumulators acc[4c .. 4c+3]; an accumulator is touched again after three others), one `item(j)` of
// other traffic issued right behind MFMA j: with ONE wavefront per SIMD and in-order issue nothing overlaps the matrix pipe
// unless it is interleaved with it (a first version that issued a K-step's DMA pieces and operand reads in front of its 24 MFMAs
// ran at ~1300 cycles per K-step against 768 of matrix time).
template <class Items>
__device__ __forceinline__ void half_step(f32x16 (&acc)[8], int c, const OpsC& o, const bf16x8& xh, const bf16x8& xl, Items items) {
#define NM_SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    acc[4 * c + b] = MFMA_BF16(__builtin_bit_cast(bf16x8, o.h[b]), xh, acc[4 * c + b]); NM_SB;
    items(b); NM_SB;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    acc[4 * c + b] = MFMA_BF16(__builtin_bit_cast(bf16x8, o.h[b]), xl, acc[4 * c + b]); NM_SB;
    items(4 + b); NM_SB;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    acc[4 * c + b] = MFMA_BF16(__builtin_bit_cast(bf16x8, o.l[b]), xh, acc[4 * c + b]); NM_SB;
    items(8 + b); NM_SB;
  }
#undef NM_SB
}

// values v[ob][r] (accumulator layout) -> the 16 K-step operands of the next product: unit 2 ob + m = registers 8m .. 8m+7 of block ob
__device__ __forceinline__ void repack(const f32x16 (&v)[8], Unit (&u)[16]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      unsigned h4[4], l4[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float x0 = v[ob][8 * m + 2 * p], x1 = v[ob][8 * m + 2 * p + 1];
        const unsigned hp = pack_bf16(x0, x1);
        h4[p] = hp;
        l4[p] = pack_bf16(x0 - __uint_as_float(hp << 16), x1 - __uint_as_float(hp & 0xffff0000u));
      }
      u[2 * ob + m].h = u32x4{h4[0], h4[1], h4[2], h4[3]};
      u[2 * ob + m].l = u32x4{l4[0], l4[1], l4[2], l4[3]};
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[8]) {
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ob][i] = 0.f;
}

// Ring protocol, one K-step AHEAD of the matrix work: while the MFMAs of step g run on operands already in registers, step
// g + 1's pieces have landed (at most allow(g) younger VMEM operations remain in flight), one barrier, step g + 3 is requested
// into the position of step g - 1 (everybody consumed it an iteration ago), step g + 1's operands go into the other buffer.
// Issue order per iteration: [wait] [barrier] [DMA g+3: 4 ops] [operand reads g+1] [att row pieces of step g+2: 2 ops, product 0
// only]; prologue: rows 0, 1, DMA 0, 1, 2.  Counting the operations issued after DMA g+1 gives:
// VMEM operations a K-step issues behind its mid-step wait, in this order: DMA of step g + 3 (4 pieces), att row pieces of step
// g + 4 (2, product 0), residual pieces (4 per step during the second half of product 0: the 32 pieces of xh in accumulator layout)
__device__ __forceinline__ constexpr int n_dma(int g) { return g + ET_AHEAD < 3 * ET_NKS ? 4 : 0; }
__device__ __forceinline__ constexpr int n_row(int g) { return g + 4 < ET_NKS ? 2 : 0; }
// (backward: the pre-GELU tile during the second half of product 0, the pre-LayerNorm tile during the second half of product 1)
// SCHED 1 = the backward kernel (two tiles); SCHED 0 = the forward kernel that keeps its intermediates (one tile: the residual, as in encoder_tail.hip)
template <int SCHED>
__device__ __forceinline__ constexpr int n_res(int g) { return ((g >= 8 && g < ET_NKS) || (SCHED == 1 && g >= ET_NKS + 8 && g < 2 * ET_NKS)) ? 4 : 0; }
// operations younger than the DMA of step g + 1 at the mid-step wait of step g (prologue: rows 0..3, DMA 0, 1, 2)
template <int SCHED>
__device__ __forceinline__ constexpr int allow_of(int g) {
  // the DMA of step g + 1 was issued in the prologue (g + 1 < ET_AHEAD: younger = the prologue's later DMAs + everything the loop issued so
  // far) or behind the wait of step g + 1 - ET_AHEAD (younger = that step's row / residual pieces + everything of the steps since)
  int n = 0;
  if (g + 1 < ET_AHEAD) {
    n = (ET_AHEAD - 1 - (g + 1)) * 4;
    for (int t = 0; t < g; ++t) n += n_dma(t) + n_row(t) + n_res<SCHED>(t);
  } else {
    const int t0 = g + 1 - ET_AHEAD;
    n = n_row(t0) + n_res<SCHED>(t0);
    for (int t = t0 + 1; t < g; ++t) n += n_dma(t) + n_row(t) + n_res<SCHED>(t);
  }
  return n;
}
template <int N>
__device__ __forceinline__ void wait_vm_n() {
#ifdef NM_SAFE_WAIT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (checker build: every counted wait becomes a full wait, see common.h)
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wait_vm(int allow) {
  switch (allow) {  // (g is a compile-time constant wherever this is called: the switch folds; every count is even)
#define NM_CASE(n) case n: wait_vm_n<n>(); break;
    NM_CASE(0) NM_CASE(2) NM_CASE(4) NM_CASE(6) NM_CASE(8) NM_CASE(10) NM_CASE(12) NM_CASE(14) NM_CASE(16) NM_CASE(18) NM_CASE(20) NM_CASE(22)
    NM_CASE(24) NM_CASE(26) NM_CASE(28) NM_CASE(30) NM_CASE(32) NM_CASE(34) NM_CASE(36) NM_CASE(38) NM_CASE(40) NM_CASE(42) NM_CASE(44) NM_CASE(46)
    NM_CASE(48) NM_CASE(50) NM_CASE(52) NM_CASE(54) NM_CASE(56) NM_CASE(58) NM_CASE(60) NM_CASE(62)
#undef NM_CASE
    default: wait_vm_n<0>(); break;  // (conservative)
  }
}

// K-step g of the 48-step stream, software pipelined over its two chunks ("consume first", as in nerf_fwd_bf16.hip):
//   chunk 0 MFMAs (operands c0, fetched during the previous step)  |  behind them: the 8 operand reads of chunk 1 of THIS step
//   wait: step g + 1 landed; barrier (=> for everybody; and everybody is past step g - 1)
//   chunk 1 MFMAs  |  behind them: the 4 DMA pieces of step g + 3 (ring position of step g - 1), the 8 operand reads of chunk 0
//   of step g + 1, and `tail()` (product 0: the att row pieces of step g + 2 -- issued after the DMA pieces, the order allow_of counts)
template <int SCHED, class Tail>
__device__ __forceinline__ void kstep(const TailArgs& a, int g, float* ring, int wave, int lane, f32x16 (&acc)[8], OpsC& c0, OpsC& c1,
                                      const bf16x8& xh, const bf16x8& xl, Tail tail) {
  const u32x4* cur = reinterpret_cast<const u32x4*>(ring + (g & (ET_RING - 1)) * ET_STEP_FLOATS) + lane;
  half_step(acc, 0, c0, xh, xl, [&](int j) {
    if (j < 8) {
      const int b = j >> 1;
      if (j & 1) c1.l[b] = cur[(8 + b * 2 + 1) * 64];
      else c1.h[b] = cur[(8 + b * 2 + 0) * 64];
    }
  });
  const bool more = g + 1 < 3 * ET_NKS;
  if (more) {
    wait_vm(allow_of<SCHED>(g));
    __builtin_amdgcn_s_barrier();
  }
  const int q = g + ET_AHEAD;
  const bool dma = q < 3 * ET_NKS;
  const char* src0 = nullptr;
  float* dst = nullptr;
  if (dma) {
    src0 = a.blob[q >> 4] + (size_t)(q & 15) * ET_SLOT_BYTES + wave * 2048 + lane * 16;
    dst = ring + (q & (ET_RING - 1)) * ET_STEP_FLOATS + wave * 512;
  }
  const u32x4* nxt = reinterpret_cast<const u32x4*>(ring + ((g + 1) & (ET_RING - 1)) * ET_STEP_FLOATS) + lane;
  half_step(acc, 1, c1, xh, xl, [&](int j) {
    if (j < 4) {
      if (dma) {
        const auto* src = (const __attribute__((address_space(1))) void*)(src0 + (size_t)(j >> 1) * ET_NKS * ET_SLOT_BYTES);
        auto* d = (__attribute__((address_space(3))) void*)(dst + (j >> 1) * (ET_SLOT_BYTES / 4));
        if (j & 1) __builtin_amdgcn_global_load_lds(src, d, 16, 1024, 0);
        else __builtin_amdgcn_global_load_lds(src, d, 16, 0, 0);
      }
    } else if (more) {
      const int b = (j - 4) >> 1;
      if ((j - 4) & 1) c0.l[b] = nxt[(b * 2 + 1) * 64];
      else c0.h[b] = nxt[(b * 2 + 0) * 64];
    }
  });
  tail();
  __builtin_amdgcn_sched_barrier(0);
}


__global__ void __launch_bounds__(256, 1) encoder_tail_bwd_kernel(TailArgs a) {
  // (ONE __shared__ object: see encoder_tail.hip)  ring 64 KiB | gamma 1 KiB | transposition buffer 4 x 16 KiB
  __shared__ __attribute__((aligned(16))) float lds[ET_RING * ET_STEP_FLOATS + ET_D + 4 * 4096];
  float* const ring = lds;
  float* const sm_gamma = lds + ET_RING * ET_STEP_FLOATS;
  float* const tbuf = sm_gamma + ET_D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * ET_ROWS + wave * 32;
  const int m = m0 + r, mc = m < a.R ? m : a.R - 1;
  sm_gamma[tid] = a.gamma[tid];
  // ---- product 0: g1 = dy . W2; rows of dy go global -> registers two K-steps ahead, split on the fly
  const float* xp = a.dy + (size_t)mc * ET_D + 8 * hi;
  struct XRow {
    f32x4 p, q;
  };
  auto xload = [&](int ks) {
    XRow v;
    v.p = *reinterpret_cast<const f32x4*>(xp + 16 * ks);
    v.q = *reinterpret_cast<const f32x4*>(xp + 16 * ks + 4);
    return v;
  };
  XRow xr[ET_NKS];
  xr[0] = xload(0);
  xr[1] = xload(1);
  xr[2] = xload(2);
  xr[3] = xload(3);
  // a row tile in the accumulator layout: 32 pieces of 16 bytes per lane (first u, then a)
  f32x4 tile[32];
  const float* up = a.u_pre + (size_t)mc * ET_D + 4 * hi;
  const float* ap = a.a_pre + (size_t)mc * ET_D + 4 * hi;
#pragma unroll
  for (int g0 = 0; g0 < ET_AHEAD; ++g0) dma_step(a, g0, ring, wave, lane);
  f32x16 acc[8];
  zero_acc(acc);
  OpsC c0, c1;
  wait_vm(4 * (ET_AHEAD - 1));
  __builtin_amdgcn_s_barrier();
  read_chunk(c0, ring, lane, 0);
#pragma unroll
  for (int ks = 0; ks < ET_NKS; ++ks) {
    bf16x8 xh, xl;
    {
      const XRow& x0 = xr[ks];
      const float v8[8] = {x0.p[0], x0.p[1], x0.p[2], x0.p[3], x0.q[0], x0.q[1], x0.q[2], x0.q[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v8[i];
        xh[i] = h;
        xl[i] = (__bf16)(v8[i] - (float)h);
      }
    }
    kstep<1>(a, ks, ring, wave, lane, acc, c0, c1, xh, xl, [&]() {
      if (ks + 4 < ET_NKS) xr[ks + 4] = xload(ks + 4);
      if (ks >= 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * (ks - 8) + j;
          tile[i] = *reinterpret_cast<const f32x4*>(up + 32 * (i >> 2) + 8 * (i & 3));
        }
      }
    });
  }
  // du = g1 o gelu'(u), re-pack
  Unit un[16];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 u4 = tile[4 * ob + q];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] *= gelu_grad(u4[e]);
    }
  repack(acc, un);
  // ---- product 1: g2 = du . W1; the pre-LayerNorm tile a arrives during its second half
  zero_acc(acc);
#pragma unroll
  for (int ks = 0; ks < ET_NKS; ++ks)
    kstep<1>(a, ET_NKS + ks, ring, wave, lane, acc, c0, c1, __builtin_bit_cast(bf16x8, un[ks].h), __builtin_bit_cast(bf16x8, un[ks].l), [&]() {
      if (ks >= 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * (ks - 8) + j;
          tile[i] = *reinterpret_cast<const f32x4*>(ap + 32 * (i >> 2) + 8 * (i & 3));
        }
      }
    });
  // LayerNorm backward over the row (the lane pair r, r + 32 owns it): statistics of a as in the forward kernel, then
  //   d_a = rstd (g - mean(g) - xhat mean(g xhat)),   g = g2 gamma,   xhat = (a - mean) rstd
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += tile[i][e];
    s += nm_shfl_xor32(s);
    const float mean = s * (1.0f / ET_D);
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = tile[i][e] - mean;
        tile[i][e] = d;
        vs = NM_FMA(d, d, vs);
      }
    vs += nm_shfl_xor32(vs);
    const float rstd = 1.0f / sqrtf(vs * (1.0f / ET_D) + a.eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(sm_gamma + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh_ = tile[4 * ob + q][e] * rstd;
          const float g = acc[ob][4 * q + e] * g4[e];
          tile[4 * ob + q][e] = xh_;
          acc[ob][4 * q + e] = g;
          sg += g;
          sgx = NM_FMA(g, xh_, sgx);
        }
      }
    sg += nm_shfl_xor32(sg);
    sgx += nm_shfl_xor32(sgx);
    const float mg = sg * (1.0f / ET_D), mgx = sgx * (1.0f / ET_D);
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] = rstd * ((acc[ob][4 * q + e] - mg) - tile[4 * ob + q][e] * mgx);
  }
  repack(acc, un);
  // coalesced store through this wavefront's 16 KiB of the transposition buffer (cf. the forward kernel's epilogue); `add`: a [R, 256] tensor
  // added in the store layout (dy for d_xh) or nullptr
  float* tb = tbuf + wave * 4096;
  const int rrow = lane >> 4, rpiece = lane & 15;
  auto store_rows = [&](float* __restrict__ dst, const float* __restrict__ add) {
#pragma unroll
    for (int hq = 0; hq < 4; ++hq) {
#pragma unroll
      for (int obl = 0; obl < 2; ++obl)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ob = 2 * hq + obl;
          const f32x4 v = {acc[ob][4 * q], acc[ob][4 * q + 1], acc[ob][4 * q + 2], acc[ob][4 * q + 3]};
          const int p = obl * 8 + 2 * q + hi;
          *reinterpret_cast<f32x4*>(tb + (hq & 1) * 2048 + r * 64 + ((p ^ (r & 15)) << 2)) = v;
        }
      const int n0 = 64 * hq + 4 * rpiece;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 4 * (4 * ib + j) + rrow;
          v[j] = *reinterpret_cast<const f32x4*>(tb + (hq & 1) * 2048 + row * 64 + ((rpiece ^ (row & 15)) << 2));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int mm = m0 + 4 * (4 * ib + j) + rrow;
          if (mm < a.R) {
            if (add) {
              const f32x4 d4 = *reinterpret_cast<const f32x4*>(add + (size_t)mm * ET_D + n0);
              v[j] = f32x4{v[j][0] + d4[0], v[j][1] + d4[1], v[j][2] + d4[2], v[j][3] + d4[3]};
            }
            *reinterpret_cast<f32x4*>(dst + (size_t)mm * ET_D + n0) = v[j];
          }
        }
      }
    }
  };
  store_rows(a.d_xh, a.dy);  // d_xh = dy + d_a (the wavefront's own 16 KiB: no barrier; LDS operations of a wavefront complete in order)
  // Everything this wavefront has in flight retires here -- the loads and stores of the epilogue above and the weight stream's requests for the
  // first K-steps of product 2 -- so the counted waits of product 2, which assume more in flight than there is, stay on the safe side.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // ---- product 2: d_att = d_a . Wo
  zero_acc(acc);
#pragma unroll
  for (int ks = 0; ks < ET_NKS; ++ks)
    kstep<1>(a, 2 * ET_NKS + ks, ring, wave, lane, acc, c0, c1, __builtin_bit_cast(bf16x8, un[ks].h), __builtin_bit_cast(bf16x8, un[ks].l), [] {});
  store_rows(a.d_att, nullptr);
}

// The FORWARD of the tail for the same frozen-parameter passes, as one launch that also KEEPS what the backward kernel above needs: the forward
// kernel of encoder_tail.hip (same arithmetic, same order) with a (the LayerNorm's input) and u (the GELU's input) leaving mid-chain through the
// transposition buffer.  Five launches before (GEMM + residual, LayerNorm, GEMM, GELU, GEMM + residual).
__global__ void __launch_bounds__(256, 1) encoder_tail_save_kernel(TailArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[ET_RING * ET_STEP_FLOATS + 4 * ET_D + 4 * 4096];
  float* const ring = lds;
  float* const sm_vec = lds + ET_RING * ET_STEP_FLOATS;  // gamma, beta, b1, b2
  float* const tbuf = sm_vec + 4 * ET_D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * ET_ROWS + wave * 32;
  const int m = m0 + r, mc = m < a.R ? m : a.R - 1;
  sm_vec[tid] = a.gamma[tid];
  sm_vec[ET_D + tid] = a.beta[tid];
  sm_vec[2 * ET_D + tid] = a.b1[tid];
  sm_vec[3 * ET_D + tid] = a.b2[tid];
  const float* xp = a.att + (size_t)mc * ET_D + 8 * hi;
  struct XRow {
    f32x4 p, q;
  };
  auto xload = [&](int ks) {
    XRow v;
    v.p = *reinterpret_cast<const f32x4*>(xp + 16 * ks);
    v.q = *reinterpret_cast<const f32x4*>(xp + 16 * ks + 4);
    return v;
  };
  XRow xr[ET_NKS];
  xr[0] = xload(0);
  xr[1] = xload(1);
  xr[2] = xload(2);
  xr[3] = xload(3);
  f32x4 res[32];
  const float* rp = a.xh + (size_t)mc * ET_D + 4 * hi;
#pragma unroll
  for (int g0 = 0; g0 < ET_AHEAD; ++g0) dma_step(a, g0, ring, wave, lane);
  f32x16 acc[8];
  zero_acc(acc);
  OpsC c0, c1;
  wait_vm(4 * (ET_AHEAD - 1));
  __builtin_amdgcn_s_barrier();
  read_chunk(c0, ring, lane, 0);
#pragma unroll
  for (int ks = 0; ks < ET_NKS; ++ks) {
    bf16x8 xh, xl;
    {
      const XRow& x0 = xr[ks];
      const float v8[8] = {x0.p[0], x0.p[1], x0.p[2], x0.p[3], x0.q[0], x0.q[1], x0.q[2], x0.q[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v8[i];
        xh[i] = h;
        xl[i] = (__bf16)(v8[i] - (float)h);
      }
    }
    kstep<1>(a, ks, ring, wave, lane, acc, c0, c1, xh, xl, [&]() {
      if (ks + 4 < ET_NKS) xr[ks + 4] = xload(ks + 4);
      if (ks >= 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * (ks - 8) + j;
          res[i] = *reinterpret_cast<const f32x4*>(rp + 32 * (i >> 2) + 8 * (i & 3));
        }
      }
    });
  }
  float* tb = tbuf + wave * 4096;
  const int rrow = lane >> 4, rpiece = lane & 15;
  auto store_rows = [&](float* __restrict__ dst) {
#pragma unroll
    for (int hq = 0; hq < 4; ++hq) {
#pragma unroll
      for (int obl = 0; obl < 2; ++obl)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ob = 2 * hq + obl;
          const f32x4 v = {acc[ob][4 * q], acc[ob][4 * q + 1], acc[ob][4 * q + 2], acc[ob][4 * q + 3]};
          const int p = obl * 8 + 2 * q + hi;
          *reinterpret_cast<f32x4*>(tb + (hq & 1) * 2048 + r * 64 + ((p ^ (r & 15)) << 2)) = v;
        }
      const int n0 = 64 * hq + 4 * rpiece;
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 4 * (4 * ib + j) + rrow;
          v[j] = *reinterpret_cast<const f32x4*>(tb + (hq & 1) * 2048 + row * 64 + ((rpiece ^ (row & 15)) << 2));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int mm = m0 + 4 * (4 * ib + j) + rrow;
          if (mm < a.R) *reinterpret_cast<f32x4*>(dst + (size_t)mm * ET_D + n0) = v[j];
        }
      }
    }
  };
  Unit un[16];
  {
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 rr = res[4 * ob + q];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] += rr[e];
      }
    store_rows(a.a_out);  // a = xh + att . Wo^T leaves here
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see the backward kernel: the counted waits that follow stay on the safe side)
    float s = 0.f;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[ob][i];
    s += nm_shfl_xor32(s);
    const float mean = s * (1.0f / ET_D);
    float vs = 0.f;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = acc[ob][i] - mean;
        acc[ob][i] = d;
        vs = NM_FMA(d, d, vs);
      }
    vs += nm_shfl_xor32(vs);
    const float rstd = 1.0f / sqrtf(vs * (1.0f / ET_D) + a.eps);
#pragma unroll
    for (int ob = 0; ob < 8; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(sm_vec + 32 * ob + 8 * q + 4 * hi);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sm_vec + ET_D + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] = (acc[ob][4 * q + e] * rstd) * g4[e] + b4[e];
      }
    repack(acc, un);
  }
  zero_acc(acc);
  // (the residual row is read a SECOND time during the second half of this product, for product 2's starting value: kept in registers across
  // the two mid-chain stores it spilled 58 of them to scratch -- the forward kernel of encoder_tail.hip has no store before its end)
#pragma unroll
  for (int ks = 0; ks < ET_NKS; ++ks)
    kstep<1>(a, ET_NKS + ks, ring, wave, lane, acc, c0, c1, __builtin_bit_cast(bf16x8, un[ks].h), __builtin_bit_cast(bf16x8, un[ks].l), [&]() {
      if (ks >= 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * (ks - 8) + j;
          res[i] = *reinterpret_cast<const f32x4*>(rp + 32 * (i >> 2) + 8 * (i & 3));
        }
      }
    });
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sm_vec + 2 * ET_D + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] += b4[e];
    }
  store_rows(a.u_out);  // u = W1 . LN2(a) + b1 leaves here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ob][i] = gelu_erf(acc[ob][i]);
  repack(acc, un);
#pragma unroll
  for (int ob = 0; ob < 8; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sm_vec + 3 * ET_D + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] = b4[e] + res[4 * ob + q][e];
    }
#pragma unroll
  for (int ks = 0; ks < ET_NKS; ++ks)
    kstep<1>(a, 2 * ET_NKS + ks, ring, wave, lane, acc, c0, c1, __builtin_bit_cast(bf16x8, un[ks].h), __builtin_bit_cast(bf16x8, un[ks].l), [] {});
  store_rows(a.y);
}

}  // namespace

extern "C" int nm_encoder_tail_bwd_bf16x3(const float* dy, const float* a_pre, const float* u_pre, const void* w2t_blob, const void* w1t_perm_blob,
                                          const void* wot_perm_blob, const float* gamma2, int rows, int dim, float eps, float* d_att, float* d_xh,
                                          nmStream_t stream) {
  NM_CHECK_ARG(dy && a_pre && u_pre && w2t_blob && w1t_perm_blob && wot_perm_blob && gamma2 && d_att && d_xh && rows > 0);
  if (dim != ET_D) return NM_ERR_UNSUPPORTED;
  TailArgs a{};
  a.dy = dy; a.a_pre = a_pre; a.u_pre = u_pre; a.blob[0] = (const char*)w2t_blob; a.blob[1] = (const char*)w1t_perm_blob;
  a.blob[2] = (const char*)wot_perm_blob; a.gamma = gamma2; a.d_att = d_att; a.d_xh = d_xh; a.R = rows; a.eps = eps;
  encoder_tail_bwd_kernel<<<(rows + ET_ROWS - 1) / ET_ROWS, 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}

extern "C" int nm_encoder_tail_save_bf16x3(const float* att, const float* xh, const void* wo_blob, const void* w1_perm_blob, const void* w2_perm_blob,
                                           const float* gamma2, const float* beta2, const float* b1, const float* b2, int rows, int dim, float eps,
                                           float* y, float* a_out, float* u_out, nmStream_t stream) {
  NM_CHECK_ARG(att && xh && wo_blob && w1_perm_blob && w2_perm_blob && gamma2 && beta2 && b1 && b2 && y && a_out && u_out && rows > 0);
  if (dim != ET_D) return NM_ERR_UNSUPPORTED;
  TailArgs a{};
  a.att = att; a.xh = xh; a.blob[0] = (const char*)wo_blob; a.blob[1] = (const char*)w1_perm_blob; a.blob[2] = (const char*)w2_perm_blob;
  a.gamma = gamma2; a.beta = beta2; a.b1 = b1; a.b2 = b2; a.y = y; a.a_out = a_out; a.u_out = u_out; a.R = rows; a.eps = eps;
  encoder_tail_save_kernel<<<(rows + ET_ROWS - 1) / ET_ROWS, 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}
