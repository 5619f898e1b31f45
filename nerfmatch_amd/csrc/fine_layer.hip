// The image side of the fine stage as ONE kernel on the matrix cores (round 5): window gather (FinePreprocess,
// third_party/loftr/fine_matching.py:58-71) + the self-attention encoder layer `fine_sa` on the 25 window tokens of every match
// (GenericEncoderLayer.forward_pre_norm, nerfmatch/modules/attention.py:229-241; width 128, 8 heads of 16):
//     xh = LN1(x);  q, k, v = xh Wq^T, xh Wk^T, xh Wv^T;  att = softmax(q k^T scale) v per head;  a = xh + att Wo^T;
//     y = xh + W2 gelu(W1 LN2(a) + b1) + b2
// As separate launches this was window gather + LayerNorm + GEMM + attention + GEMM + LayerNorm + 2 GEMMs: eight launches, 110 us of launch
// and pipeline latency for the ~200 matches of one query, 365 us for the ~3200 of sixteen.  An fp32-VALU fusion (one workgroup per match) was
// faster for one query and slower for sixteen (profiles/r5_ab_fine_window_layer.log); this one serves both:
//   * one wavefront = one match: its 25 tokens are rows 0..24 of a 32-row MFMA tile (rows 25..31 are padding and never leave the kernel),
//     four matches per workgroup share every weight fetch;
//   * arithmetic = the split-bf16 products of gemm_bf16.hip (w_hi x_hi + w_hi x_lo + w_lo x_hi, fp32 accumulate); a product's 128 outputs
//     sit in 4 accumulator blocks (lane = row, register r of block ob <-> feature 32 ob + (r & 3) + 8 (r >> 2) + 4 half) and 8 consecutive
//     registers are one K-step operand of the NEXT product when its weights are packed in that K order (nm_linear_pack_perm_bf16x3, as in
//     encoder_tail.hip): LayerNorm, GELU, bias, residual and the hi / lo re-packing are lane local (two xor-32 reductions per LayerNorm);
//     the window is gathered straight into that layout, so all six weight matrices use the permuted pack;
//   * a product's 64 KiB of pre-split weights are copied to LDS once per workgroup (four matches read them from there);
//   * attention: per head a lane (token t, half) owns 8 of the 16 dims.  Round 6: scores and weighted sums are two small products on the
//     matrix cores (S^T = K . Q^T straight from the registers, O^T = V^T . P^T with V^T read from a 4 KiB LDS scratch), soft-max lane local
//     plus one xor-32 exchange, the result again in the accumulator layout (round 5 did both on the VALU with keys / values as broadcast LDS
//     reads: scripts/variants/fine_layer_attention_valu_r6.patch).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int FL_T = 25, FL_D = 128, FL_NKS = 8, FL_SLOT_FLOATS = 2048, FL_BLOB_FLOATS = FL_NKS * FL_SLOT_FLOATS;  // 64 KiB per matrix

struct FLArgs {
  const float* ffeat; int Hf, Wf; const int64_t* map_ids; const int64_t* i_ids; const int* count; int max_k; int stride;
  const float *ln1_g, *ln1_b; float eps1;
  const char* blob[6];  // Wq, Wk, Wv, Wo, W1, W2: permuted-K pack, one 128-column chunk x 8 K-steps of 8 KiB
  const float *ln2_g, *ln2_b; float eps2;
  const float *b1, *b2;
  float scale;
  float* out;         // [K][25][128], or NULL when only the expectation is wanted
  const float* pt_f;  // [K][128] point-side fine features (nm_fine_pt_proj), or NULL
  float* expec;       // [K][3] <- FineMatching's expectation of the match (nm_fine_expectation's arithmetic), when pt_f / pt_src is given
  // the point side computed HERE (nm_fine_pt_proj's arithmetic: pt_f[k] = W1 (W0 pt_src[pt_ids[k]] + b0) + b1, fp32 FMAs in K order) when pt_src is given
  const float* pt_src; const int64_t* pt_ids; int pt_c0;  // [rows][pt_c0], pt_c0 a multiple of 4, at most 512
  const float *pw0t, *pb0, *pw1t, *pb1;                   // transposed weights [pt_c0][128], [128][128]; biases may be NULL
};

__host__ __device__ __forceinline__ constexpr int nrow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return __builtin_bit_cast(unsigned, bf16x2{(__bf16)a, (__bf16)b}); }

// x = h + m + l with three bf16 terms (24 bits: exact for fp32 inputs up to the last term's rounding)
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    h[i] = a;
    m[i] = b;
    l[i] = (__bf16)(r1 - (float)b);
  }
}

// x = h + m with two bf16 terms (16 bits: the split of every 128-wide product of this kernel)
__device__ __forceinline__ void split2(const float (&x)[8], bf16x8& h, bf16x8& m) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    h[i] = a;
    m[i] = (__bf16)(x[i] - (float)a);
  }
}

struct Unit {
  u32x4 h, l;
};
// values v[ob][r] (accumulator layout) -> the 8 K-step operands of the next product: unit 2 ob + m = registers 8 m .. 8 m + 7 of block ob
__device__ __forceinline__ void repack(const f32x16 (&v)[4], Unit (&u)[8]) {
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
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

// one matrix: global -> LDS, 256 threads x 16 pieces of 16 bytes
__device__ __forceinline__ void load_blob(const char* blob, float* wlds, int tid) {
  const u32x4* src = reinterpret_cast<const u32x4*>(blob) + tid;
  u32x4* dst = reinterpret_cast<u32x4*>(wlds) + tid;
#pragma unroll
  for (int b = 0; b < 4; ++b) {  // (four pieces in flight per thread: 16 staging registers, not 64)
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = src[(4 * b + i) * 256];
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[(4 * b + i) * 256] = v[i];
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc[ob] = sum over the 8 K-steps of W_slot(ks, ob) . unit ks   (three products per block and K-step)
__device__ __forceinline__ void product(const float* wlds, const Unit (&u)[8], f32x16 (&acc)[4], int lane) {
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ob][i] = 0.f;
#pragma unroll
  for (int ks = 0; ks < FL_NKS; ++ks) {
    const u32x4* s4 = reinterpret_cast<const u32x4*>(wlds + ks * FL_SLOT_FLOATS) + lane;
    const bf16x8 xh = __builtin_bit_cast(bf16x8, u[ks].h), xl = __builtin_bit_cast(bf16x8, u[ks].l);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const bf16x8 wh = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 0) * 64]), wl = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 1) * 64]);
      acc[ob] = MFMA_BF16(wh, xh, acc[ob]);
      acc[ob] = MFMA_BF16(wh, xl, acc[ob]);
      acc[ob] = MFMA_BF16(wl, xh, acc[ob]);
    }
  }
}

// LayerNorm of the rows held in the accumulator layout (the lane pair r, r + 32 owns a row): mean, centred variance, affine
__device__ __forceinline__ void layernorm_rows(f32x16 (&v)[4], const float* g, const float* b, float eps, int hi) {
  float s = 0.f;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[ob][i];
  s += nm_shfl_xor32(s);
  const float mean = s * (1.0f / FL_D);
  float vs = 0.f;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float d = v[ob][i] - mean;
      v[ob][i] = d;
      vs = NM_FMA(d, d, vs);
    }
  vs += nm_shfl_xor32(vs);
  const float rstd = 1.0f / sqrtf(vs * (1.0f / FL_D) + eps);
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + 32 * ob + 8 * q + 4 * hi);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(b + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[ob][4 * q + e] = (v[ob][4 * q + e] * rstd) * g4[e] + b4[e];
    }
}

// exact-erf GELU with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, as in encoder_tail.hip: one v_rcp + one v_exp + 7 FMA-class
// instructions instead of the ~50 of erff -- 64 activations per lane sit between two products with nothing to overlap them)
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

__global__ void __launch_bounds__(256, 1) fine_layer_kernel(FLArgs a) {
  __shared__ __attribute__((aligned(16))) float wlds[FL_BLOB_FLOATS];       // 64 KiB: the current product's weights
  __shared__ __attribute__((aligned(16))) float scr[4][2][32][8];           // [wavefront][half][token][8 dims]: the values of the head at hand, 8 KiB
  __shared__ __attribute__((aligned(16))) float vec[6][FL_D];               // ln1 g, b, ln2 g, b, b1, b2
  __shared__ float xh_lds[4][64][64];                                        // [wavefront][register][lane]: the normalised input, parked: 64 KiB
  __shared__ __attribute__((aligned(16))) float pfl[4][FL_D];               // the four matches' point-side fine features
  const int n = min(*a.count, a.max_k);
  // Round 6: which group of four matches a workgroup takes.  Workgroup b runs on XCD b % 8, and the matches are sorted by image cell: with
  // group = b, eight neighbouring groups -- whose 5 x 5 windows share the 128-byte lines of the NCHW map (a line spans eight cells' width) -- sit
  // on eight different XCDs, each L2 fetching the same lines.  XCD x takes the x-th contiguous eighth of the VALID groups instead (a permutation
  // of [0, 8 per); the groups behind it -- slots past the count, which only write zeros -- keep group = b): -2.6 % at 4000 matches, -2.9 % at
  // 64000, neutral at 192 (profiles/r6_ab_fine_stage_phases.log) -- the gather's cost is mostly the 20-of-128 bytes used per line, not this.
  int grp;
  {
    const int G = (int)gridDim.x, Gv = (n + 3) / 4, b = (int)blockIdx.x;
    int per = (Gv + 7) / 8;
    if (8 * per > G) per = G / 8;
    grp = b < 8 * per ? (b % 8) * per + b / 8 : b;
  }
  if (grp * 4 >= n) {  // (whole workgroup: no barrier is left waiting)
    // slots behind the count hold zeros, like nm_fine_pt_proj's (the speculative single-pair path hands all `cap` slots on to
    // nm_assemble_matches, which must not read uninitialised floats)
    if (a.expec && threadIdx.x < 12 && grp * 4 + (int)threadIdx.x / 3 < a.max_k) a.expec[(size_t)grp * 12 + threadIdx.x] = 0.f;
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int k = grp * 4 + wave;
  const bool valid = k < n;
  const int kc = valid ? k : n - 1;  // (a wavefront without a match computes on the last one and stores nothing)
  if (tid < FL_D) {
    vec[0][tid] = a.ln1_g[tid]; vec[1][tid] = a.ln1_b[tid]; vec[2][tid] = a.ln2_g[tid]; vec[3][tid] = a.ln2_b[tid];
    vec[4][tid] = a.b1[tid]; vec[5][tid] = a.b2[tid];
  }
  // the window, straight into the accumulator layout: lane (token r, half) holds features 32 ob + nrow(q, half)
  f32x16 xh[4];
  {
    const float* fm = a.ffeat + (size_t)a.map_ids[kc] * FL_D * a.Hf * a.Wf;
    const int cells_w = (a.Wf + 2 * 2 - 5) / a.stride + 1;
    const int cell = (int)a.i_ids[kc];
    const int y = (cell / cells_w) * a.stride - 2 + r / 5, x = (cell % cells_w) * a.stride - 2 + r % 5;
    const bool in = r < FL_T && y >= 0 && y < a.Hf && x >= 0 && x < a.Wf;
    const size_t plane = (size_t)a.Hf * a.Wf;
    const float* px = fm + (in ? (size_t)y * a.Wf + x : 0);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int q = 0; q < 16; ++q) xh[ob][q] = in ? px[(size_t)(32 * ob + nrow(q, hi)) * plane] : 0.f;
  }
  if (a.pt_src) {
    // point side of the four matches (fine_pt_proj_kernel's expressions): source rows through the (still unused) weight buffer, thread (column c,
    // half) owns output column c of two matches
    float* xs = wlds;                 // [4][pt_c0]
    float* hm = wlds + 4 * 512;       // [4][128]
    const int C0 = a.pt_c0;
    {
      const float* row = a.pt_src + (size_t)a.pt_ids[kc] * C0;
      for (int i = lane; i < C0; i += 64) xs[wave * C0 + i] = row[i];
    }
    __syncthreads();
    const int c = tid & (FL_D - 1), half = tid >> 7;
    float acc0, acc1;
    acc0 = acc1 = a.pb0 ? a.pb0[c] : 0.f;
#pragma unroll 8  // (eight steps' weight loads in flight: 1 -> +27 %, 4 -> +3 %, 16 -> +28 % of the launch at 4 k matches, profiles/r6_ab_fine_stage_phases.log)
    for (int kk2 = 0; kk2 < C0; kk2 += 4) {
      const float w0 = a.pw0t[(size_t)kk2 * FL_D + c], w1 = a.pw0t[(size_t)(kk2 + 1) * FL_D + c], w2 = a.pw0t[(size_t)(kk2 + 2) * FL_D + c],
                  w3 = a.pw0t[(size_t)(kk2 + 3) * FL_D + c];
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + (2 * half) * C0 + kk2), x1 = *reinterpret_cast<const f32x4*>(xs + (2 * half + 1) * C0 + kk2);
      acc0 = NM_FMA(x0[3], w3, NM_FMA(x0[2], w2, NM_FMA(x0[1], w1, NM_FMA(x0[0], w0, acc0))));
      acc1 = NM_FMA(x1[3], w3, NM_FMA(x1[2], w2, NM_FMA(x1[1], w1, NM_FMA(x1[0], w0, acc1))));
    }
    hm[(2 * half) * FL_D + c] = acc0;
    hm[(2 * half + 1) * FL_D + c] = acc1;
    __syncthreads();
    acc0 = acc1 = a.pb1 ? a.pb1[c] : 0.f;
#pragma unroll 8
    for (int kk2 = 0; kk2 < FL_D; kk2 += 4) {
      const float w0 = a.pw1t[(size_t)kk2 * FL_D + c], w1 = a.pw1t[(size_t)(kk2 + 1) * FL_D + c], w2 = a.pw1t[(size_t)(kk2 + 2) * FL_D + c],
                  w3 = a.pw1t[(size_t)(kk2 + 3) * FL_D + c];
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(hm + (2 * half) * FL_D + kk2), h1 = *reinterpret_cast<const f32x4*>(hm + (2 * half + 1) * FL_D + kk2);
      acc0 = NM_FMA(h0[3], w3, NM_FMA(h0[2], w2, NM_FMA(h0[1], w1, NM_FMA(h0[0], w0, acc0))));
      acc1 = NM_FMA(h1[3], w3, NM_FMA(h1[2], w2, NM_FMA(h1[1], w1, NM_FMA(h1[0], w0, acc1))));
    }
    pfl[2 * half][c] = acc0;
    pfl[2 * half + 1][c] = acc1;
  }
  __syncthreads();  // the vectors (and the point features) are in LDS; nobody reads the weight buffer's staging area any more
  layernorm_rows(xh, vec[0], vec[1], a.eps1, hi);
  Unit un[8];
  repack(xh, un);
  // xh is needed twice more (the two residuals): parked in LDS, lane-strided, instead of 64 registers that are live through everything
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) xh_lds[wave][16 * ob + i][lane] = xh[ob][i];
  // q, k, v and the attention one 32-column block (= two heads) at a time: the block's slices of Wq, Wk, Wv (3 x 16 KiB) are what is in LDS,
  // and only 48 product registers + the finished blocks of the attention output are live (all 128 columns of q, k and v at once spilled
  // ~300 registers per lane)
  f32x16 att[4];
  {
    float* vs_ = &scr[wave][hi][0][0];
    // (NOT unrolled: with the four block iterations unrolled the register allocator kept ~340 values too many alive and spilled them)
#pragma unroll 1
    for (int ob = 0; ob < 4; ++ob) {
      if (ob > 0) __syncthreads();  // everybody is through with the previous block's slices
      f32x16 blk;
      {  // slice (matrix mat, K-step ks) = 2 KiB at blob[mat] + ks * 8 KiB + ob * 2 KiB -> wlds[(mat * 8 + ks) * 512 floats]: 3072 pieces of 16 bytes
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          u32x4 v4[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int p = (4 * b + i) * 256 + tid;  // piece 0 .. 3071
            const int mat = p >> 10, rest = p & 1023, ksl = rest >> 7, w = rest & 127;
            v4[i] = reinterpret_cast<const u32x4*>(a.blob[mat] + (size_t)ksl * 8192 + (size_t)ob * 2048)[w];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) reinterpret_cast<u32x4*>(wlds)[(4 * b + i) * 256 + tid] = v4[i];
        }
      }
      __syncthreads();
      f32x16 qkv[3];
#pragma unroll
      for (int mat = 0; mat < 3; ++mat) {
#pragma unroll
        for (int i = 0; i < 16; ++i) qkv[mat][i] = 0.f;
#pragma unroll
        for (int ksl = 0; ksl < FL_NKS; ++ksl) {
          const u32x4* s4 = reinterpret_cast<const u32x4*>(wlds + (mat * 8 + ksl) * 512) + lane;
          const bf16x8 wh = __builtin_bit_cast(bf16x8, s4[0]), wl = __builtin_bit_cast(bf16x8, s4[64]);
          const bf16x8 xh8 = __builtin_bit_cast(bf16x8, un[ksl].h), xl8 = __builtin_bit_cast(bf16x8, un[ksl].l);
          qkv[mat] = MFMA_BF16(wh, xh8, qkv[mat]);
          qkv[mat] = MFMA_BF16(wh, xl8, qkv[mat]);
          qkv[mat] = MFMA_BF16(wl, xh8, qkv[mat]);
        }
      }
      // the block's two heads: head m = registers 8 m .. 8 m + 7 in both halves (16 dims)
      // Round 6: scores and weighted sums on the matrix cores.  K as the A operand and Q as the B operand of ONE 32 x 32 x 16 product are
      // the registers as they stand (lane (token, half) holds its 8 of the head's 16 dims for both): S^T[key][query] comes out with lane =
      // query and registers = keys (register v of half h: key 8 (v >> 2) + 4 h + (v & 3)), so soft-max is lane local plus one xor-32 exchange.
      // Those registers are the B operand of O^T = V^T . P^T with the contraction slots (step s, half h, t) <-> key of register 8 s + t -- V^T,
      // lane = dim, is read from the LDS scratch in that key order (16 scalar reads) --, and O^T[dim row][query] lands in lane = query,
      // register v < 8 of half h = the head's dim slot (h, v): the accumulator layout of the chain, no movement.  The SCORES' operands are split in
      // THREE bf16 terms (24 bits, six products: what enters the exponential is fp32-exact like the VALU form this replaces -- a two-term score of
      // magnitude 50 would carry 5e-4 into the soft-max), P and V in two (16 bits, three products: the split of every other product of this
      // kernel; measured error of the layer against fp64 unchanged, 2e-6 ... 3.6e-6).  The VALU form -- 25 x 25 scores and sums per head and lane
      // with keys / values as broadcast LDS reads -- was 29 % of the launch at 64 k matches, LDS-bound (profiles/r6_ab_fine_stage_phases.log).
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float q8[8], k8[8], v8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { q8[i] = qkv[0][8 * m + i]; k8[i] = qkv[1][8 * m + i]; v8[i] = qkv[2][8 * m + i]; }
        *reinterpret_cast<f32x4*>(vs_ + r * 8) = f32x4{v8[0], v8[1], v8[2], v8[3]};
        *reinterpret_cast<f32x4*>(vs_ + r * 8 + 4) = f32x4{v8[4], v8[5], v8[6], v8[7]};
        bf16x8 qh, qm, ql, kh, km, kl;
        split3(q8, qh, qm, ql);
        split3(k8, kh, km, kl);
        f32x16 st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = 0.f;
        st = MFMA_BF16(kl, qh, st);  // (small terms first)
        st = MFMA_BF16(kh, ql, st);
        st = MFMA_BF16(km, qm, st);
        st = MFMA_BF16(km, qh, st);
        st = MFMA_BF16(kh, qm, st);
        st = MFMA_BF16(kh, qh, st);
        float p[16], mx = -__builtin_inff();
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const bool ok = 8 * (v >> 2) + 4 * hi + (v & 3) < FL_T;  // (rows 25..31 of the tile are padding tokens)
          p[v] = ok ? st[v] * a.scale : -__builtin_inff();
          mx = fmaxf(mx, p[v]);
        }
        mx = fmaxf(mx, nm_shfl_xor32(mx));
        float den = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          p[v] = expf(p[v] - mx);  // (exp(-inf) = 0 for the padding keys)
          den += p[v];
        }
        den += nm_shfl_xor32(den);
        // V^T rows: lane rho = r < 16 stands for dim slot (half (rho >> 2) & 1, e = (rho & 3) + 4 (rho >> 3)); its 16 keys in register order of half hi
        f32x16 ot;
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[i] = 0.f;
        const float* vsrc = &scr[wave][(r >> 2) & 1][0][0] + ((r & 3) + 4 * ((r >> 3) & 1));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float vt[8], pp[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int key = 8 * ((8 * s2 + t) >> 2) + 4 * hi + (t & 3);
            vt[t] = r < 16 ? vsrc[key * 8] : 0.f;
            pp[t] = p[8 * s2 + t];
          }
          bf16x8 vh, vm, ph, pm;
          split2(vt, vh, vm);
          split2(pp, ph, pm);
          ot = MFMA_BF16(vm, ph, ot);
          ot = MFMA_BF16(vh, pm, ot);
          ot = MFMA_BF16(vh, ph, ot);
        }
        const float inv = 1.0f / den;
#pragma unroll
        for (int i = 0; i < 8; ++i) blk[8 * m + i] = ot[i] * inv;
        __builtin_amdgcn_sched_barrier(0);  // (heads one after the other)
      }
      // (att[ob] = blk with the loop counter as the index made `att` a scratch array: 256 bytes per lane out and back in, 1.05 GB written at 64 k
      //  matches -- profiles/r6_pmc_fine_layer_64k_before.json; four predicated copies keep the blocks in registers)
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4)
        if (ob == b4) att[b4] = blk;
    }
  }
  __syncthreads();  // everybody is through with the last block's slices
  load_blob(a.blob[3], wlds, tid);
  __syncthreads();  // Wo in LDS
  repack(att, un);
  f32x16 acc[4];
  product(wlds, un, acc, lane);
  __syncthreads();
  load_blob(a.blob[4], wlds, tid);
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ob][i] += xh_lds[wave][16 * ob + i][lane];  // a = att Wo^T + xh
  layernorm_rows(acc, vec[2], vec[3], a.eps2, hi);
  repack(acc, un);
  __syncthreads();  // W1 in LDS
  product(wlds, un, acc, lane);
  __syncthreads();
  load_blob(a.blob[5], wlds, tid);
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(vec[4] + 32 * ob + 8 * qq + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ob][4 * qq + e] = gelu_erf(acc[ob][4 * qq + e] + b4[e]);
    }
  repack(acc, un);
  __syncthreads();  // W2 in LDS
  product(wlds, un, acc, lane);
  float dot = 0.f;  // <pt_f[k], y[r]> over this lane's 64 columns
  const float* pf = a.pt_src ? pfl[wave] : a.pt_f ? a.pt_f + (size_t)kc * FL_D : nullptr;
  float* y = (a.out && valid && r < FL_T) ? a.out + ((size_t)k * FL_T + r) * FL_D : nullptr;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(vec[5] + 32 * ob + 8 * qq + 4 * hi);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (acc[ob][4 * qq + e] + b4[e]) + xh_lds[wave][16 * ob + 4 * qq + e][lane];
      if (y) *reinterpret_cast<f32x4*>(y + 32 * ob + 8 * qq + 4 * hi) = o;
      if (pf) {
        const f32x4 p4 = *reinterpret_cast<const f32x4*>(pf + 32 * ob + 8 * qq + 4 * hi);
        dot = NM_FMA(p4[3], o[3], NM_FMA(p4[2], o[2], NM_FMA(p4[1], o[1], NM_FMA(p4[0], o[0], dot))));
      }
    }
  if (pf) {
    // FineMatching (third_party/loftr/fine_matching.py:88-121): soft-max of the 25 correlations, expectation and spread over the normalised grid
    // -- nm_fine_expectation's expressions with the wavefront's lanes 0 .. 24 as the window positions
    dot += nm_shfl_xor32(dot);
    const bool pos = hi == 0 && r < FL_T;
    const float sim = pos ? dot * (1.0f / sqrtf((float)FL_D)) : -__builtin_inff();
    float mx = sim;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float p = pos ? expf(sim - mx) : 0.f;
    float sp = p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sp += __shfl_xor(sp, o, 64);
    p = p / sp;
    const int gy_i = r / 5, gx_i = r % 5;
    const float stepg = 2.0f / 4.0f;
    auto lin = [&](int i) -> float { return (i < 2) ? -1.0f + stepg * (float)i : 1.0f - stepg * (float)(4 - i); };
    const float gx = pos ? lin(gx_i) : 0.f, gy = pos ? lin(gy_i) : 0.f;
    float ex = gx * p, ey = gy * p, exx = gx * gx * p, eyy = gy * gy * p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ex += __shfl_xor(ex, o, 64);
      ey += __shfl_xor(ey, o, 64);
      exx += __shfl_xor(exx, o, 64);
      eyy += __shfl_xor(eyy, o, 64);
    }
    if (lane == 0 && valid) {
      const float vx = fmaxf(exx - ex * ex, 1e-10f), vy = fmaxf(eyy - ey * ey, 1e-10f);
      a.expec[(size_t)k * 3 + 0] = ex;
      a.expec[(size_t)k * 3 + 1] = ey;
      a.expec[(size_t)k * 3 + 2] = sqrtf(vx) + sqrtf(vy);
    } else if (lane == 0 && k < a.max_k) {  // a slot behind the count inside the last working workgroup: zeros
      a.expec[(size_t)k * 3 + 0] = 0.f;
      a.expec[(size_t)k * 3 + 1] = 0.f;
      a.expec[(size_t)k * 3 + 2] = 0.f;
    }
  }
}

}  // namespace

extern "C" int nm_fine_stage(const float*, int, int, int, int, const int64_t*, const int64_t*, const int*, int, int, int, int, const float*, const float*,
                             float, const void*, const void*, const void*, const void*, const float*, const float*, float, const void*, const float*,
                             const void*, const float*, float, float*, const float*, const float*, const int64_t*, int, const float*, const float*,
                             const float*, const float*, float*, nmStream_t);

extern "C" int nm_fine_window_layer(const float* ffeat, int B, int C, int Hf, int Wf, const int64_t* map_ids, const int64_t* i_ids, const int* count,
                                    int max_k, int win, int stride, int heads, const float* ln1_gamma, const float* ln1_beta, float ln1_eps,
                                    const void* wq_perm, const void* wk_perm, const void* wv_perm, const void* wo_perm, const float* ln2_gamma,
                                    const float* ln2_beta, float ln2_eps, const void* w1_perm, const float* b1, const void* w2_perm, const float* b2,
                                    float scale, float* out, const float* pt_f, float* expec_f, nmStream_t stream) {
  return nm_fine_stage(ffeat, B, C, Hf, Wf, map_ids, i_ids, count, max_k, win, stride, heads, ln1_gamma, ln1_beta, ln1_eps, wq_perm, wk_perm, wv_perm,
                       wo_perm, ln2_gamma, ln2_beta, ln2_eps, w1_perm, b1, w2_perm, b2, scale, out, pt_f, nullptr, nullptr, 0, nullptr, nullptr, nullptr,
                       nullptr, expec_f, stream);
}

extern "C" int nm_fine_stage(const float* ffeat, int B, int C, int Hf, int Wf, const int64_t* map_ids, const int64_t* i_ids, const int* count, int max_k,
                             int win, int stride, int heads, const float* ln1_gamma, const float* ln1_beta, float ln1_eps, const void* wq_perm,
                             const void* wk_perm, const void* wv_perm, const void* wo_perm, const float* ln2_gamma, const float* ln2_beta, float ln2_eps,
                             const void* w1_perm, const float* b1, const void* w2_perm, const float* b2, float scale, float* out, const float* pt_f,
                             const float* pt_src, const int64_t* pt_ids, int pt_c0, const float* pt_w0t, const float* pt_b0, const float* pt_w1t,
                             const float* pt_b1, float* expec_f, nmStream_t stream) {
  NM_CHECK_ARG((out || expec_f) && !(pt_f && pt_src) && (!expec_f == !(pt_f || pt_src)));
  if (pt_src) {
    NM_CHECK_ARG(pt_ids && pt_w0t && pt_w1t && pt_c0 > 0);
    if (pt_c0 % 4 || pt_c0 > 512) return NM_ERR_UNSUPPORTED;
  }
  NM_CHECK_ARG(ffeat && map_ids && i_ids && count && ln1_gamma && ln1_beta && wq_perm && wk_perm && wv_perm && wo_perm && ln2_gamma && ln2_beta &&
               w1_perm && b1 && w2_perm && b2 && B > 0 && Hf > 0 && Wf > 0 && stride > 0);
  if (C != FL_D || win != 5 || heads != 8) return NM_ERR_UNSUPPORTED;
  if (max_k <= 0) return NM_OK;
  FLArgs a;
  a.ffeat = ffeat; a.Hf = Hf; a.Wf = Wf; a.map_ids = map_ids; a.i_ids = i_ids; a.count = count; a.max_k = max_k; a.stride = stride;
  a.ln1_g = ln1_gamma; a.ln1_b = ln1_beta; a.eps1 = ln1_eps;
  a.blob[0] = (const char*)wq_perm; a.blob[1] = (const char*)wk_perm; a.blob[2] = (const char*)wv_perm; a.blob[3] = (const char*)wo_perm;
  a.blob[4] = (const char*)w1_perm; a.blob[5] = (const char*)w2_perm;
  a.ln2_g = ln2_gamma; a.ln2_b = ln2_beta; a.eps2 = ln2_eps; a.b1 = b1; a.b2 = b2; a.scale = scale; a.out = out; a.pt_f = pt_f; a.expec = expec_f;
  a.pt_src = pt_src; a.pt_ids = pt_ids; a.pt_c0 = pt_c0; a.pw0t = pt_w0t; a.pb0 = pt_b0; a.pw1t = pt_w1t; a.pb1 = pt_b1;
  fine_layer_kernel<<<(max_k + 3) / 4, 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}
