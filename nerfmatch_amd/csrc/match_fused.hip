// Dual-softmax matching + (mutual) nearest neighbour for a BATCH of image / point-set pairs WITHOUT the similarity matrix in
// HBM (inference, conf not requested).  Replaces, like match.hip, coarse_matching (nerfmatch/nerfmatch_c2f_trainer.py:289-300)
// + the inference branch of extract_mutual_matches (nerfmatch/modules/extract_matches.py:21-36); SURVEY.md 8a rows M4, M5.
//
// match.hip writes sim (92 MB per 4800 x 4800 pair) and sweeps it four times, nine launches per PAIR; at the evaluator's
// batch of 16 queries half of the wall time of the coarse-only model was launch gaps.  Here the whole
// batch is nine launches and sim / conf exist only in accumulator registers:
//   norm_pack32 (x 2)  f / (|f| + 1e-6), split into bf16 hi / lo ONCE and written as the tile kernel's ready-made MFMA
//              operands (image rows: B operands, point rows: A-operand slots): no conversion work in the tile's K-loop
//   tile<1>    128 x 128 similarity tile on the bf16 matrix cores (hi/lo split, K-loop of gemm_bf16.hip); e = exp(sim - |scale|)
//              -- the cosine similarity is bounded by |scale|, so ONE fixed shift serves the row and the column soft-max and no
//              running maxima are needed -- per-tile partial row sums and column sums (fixed summation order: deterministic)
//   merge      reciprocal row / column sums (-1 marks an all-masked row / column: the reference's soft-max is uniform there)
//   tile<2>    the tile again (11.8 GFLOP x 3 per pair is cheaper than writing and re-reading it), conf = (e ics)(e irs) in
//              registers, per-tile row maximum + first column + "several entries attain it", column maxima (atomicMax on bits)
//   select     per row: maximum over the tiles, first column; the reference's mask conf > thr && conf == rowmax [&& == colmax]
//   tie        only for rows whose maximum is attained more than once (exact ties: masked / degenerate inputs): the tiles of
//              that row block once more, every entry tested against the full mask, first passing column by atomicMin
//   compact    ordered compaction per pair
// Every conf value comes out of the same inlined expression on bit-identical accumulators (the tile code is one function), so
// the equality tests compare equal bits exactly like the reference's `conf == conf.max()`.
#include "common.h"
#include <limits.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int FT = 128;               // tile edge (rows and columns)
constexpr int F_SLOT_BYTES = 8192;    // 4 blocks x (hi, lo) x 64 lanes x 16 bytes: one 16-wide K-step of 128 point rows
constexpr int F_SLOT_FLOATS = F_SLOT_BYTES / 4;
constexpr int F_RING = 4;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float F_MAX_SHIFT = 60.0f;  // |scale| log2 e above this: exp2(-2 shift) would leave the normal fp32 range -> match.hip

struct FArgs {
  const char* imb;       // [P][ceil(M/32)][nks] pieces of 2 KiB: normalised image tokens, split and laid out as B operands
  const char* blob;      // [P][tiles_n][nks] slots: normalised point tokens, split and laid out
  const uint8_t* im_mask;  // [P][M] or NULL
  const uint8_t* pt_mask;  // [P][N] or NULL
  int M, N, C, nks, tiles_m, tiles_n;
  float s2, shift;       // scale * log2 e, |scale| * log2 e
  float* rpart;          // [P][tiles_n][M] partial row sums of e
  float* cpart;          // [P][tiles_m][N] partial column sums of e
  float* irs;            // [P][M] reciprocal row sums (-1: no unmasked entry in the row)
  float* ics;            // [P][N]
  float inv_mn;          // conf of an entry whose row AND column are entirely masked: (1/M)(1/N)
  float* rbest;          // [P][tiles_n][M] per-tile row maximum of conf
  int* ridx;             // [P][tiles_n][M] its first column | bit 31: attained more than once inside the tile
  unsigned* colmax;      // [P][N] bits of the column maxima of conf (conf >= 0: ordered like the floats)
  float thr;
  int mutual;
  int* sel_j;            // [P][M] selected column, -1 none, INT_MAX = tie still to be resolved
  float* sel_v;          // [P][M] row maximum of conf
  int* tie;              // [P][M] 1: resolve in the tie pass
};

__device__ __forceinline__ void dma_slot(const char* slots, int g, float* ring, int wave, int lane) {
  const unsigned voff = (unsigned)(wave * 2048 + lane * 16);
  const char* base = slots + (size_t)g * F_SLOT_BYTES;
  const auto* src = (const __attribute__((address_space(1))) void*)(base + voff);
  auto* dst = (__attribute__((address_space(3))) void*)(ring + (g & (F_RING - 1)) * F_SLOT_FLOATS + wave * 512);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
}

// acc[ob][reg] = dot(imn[row], ptn[col]) for row = 128 row_tile + 32 wave + (lane & 31), col = 128 chunk + 32 ob + (reg & 3) +
// 8 (reg >> 2) + 4 (lane >> 5).  The K-loop of gemm_bf16x3_kernel (gemm_bf16.hip) with both sides pre-split: point slots by LDS
// DMA three K-steps ahead, image operands global -> registers three K-steps ahead.  All 4 wavefronts call it together (ring
// barriers).
__device__ __forceinline__ void sim_tile(const FArgs& a, int p, int row_tile, int chunk, float* ring, f32x16 (&acc)[4]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int nks = a.nks;
  const char* slots = a.blob + ((size_t)p * a.tiles_n + chunk) * nks * F_SLOT_BYTES;
  // this wavefront's 32 image rows: pre-split B operands, [K-step][hi, lo][64 lanes][16 bytes] (norm_pack32_kernel): no
  // conversion work in the loop (the fp32 rows cost ~30 VALU per K-step to split, a quarter of these kernels' VALU work)
  const int groups = (a.M + 31) >> 5;
  const u32x4* xp = reinterpret_cast<const u32x4*>(a.imb + (((size_t)p * groups + row_tile * 4 + wave) * nks) * 2048) + lane;
  const bool live = row_tile * 4 + wave < groups;  // (a wavefront entirely outside the matrix re-reads group 0: results unused)
  struct XOp {
    u32x4 h, l;
  };
  auto xload = [&](int ks) {
    XOp v;
    const u32x4* q = live ? xp + (size_t)ks * 128 : reinterpret_cast<const u32x4*>(a.imb) + lane;
    v.h = q[0];
    v.l = q[64];
    return v;
  };
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ob][i] = 0.f;
  // three K-steps ahead (the tiles are only 16 K-steps long: profiles/r3_pmc_match_tile*.json);
  // K-step g's slot goes to ring position g % 4, requested at step g - 3, when everybody is past step g - 4 (barrier of g - 3)
  XOp xq[4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
    if (g < nks) {
      dma_slot(slots, g, ring, wave, lane);
      xq[g] = xload(g);
    }
  for (int ks0 = 0; ks0 < nks; ks0 += 4)  // (nks is a multiple of 4: C in {64, 128, 256, 512}; j is the compile-time index of xq)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ks = ks0 + j;
    // slot ks and row piece ks have landed when at most the 4 VMEM operations of each of the steps ks+1, ks+2 remain in flight
    if (ks + 2 < nks) NM_WAIT_VMCNT(8);
    else if (ks + 1 < nks) NM_WAIT_VMCNT(4);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (ks + 3 < nks) {
      dma_slot(slots, ks + 3, ring, wave, lane);
      xq[(j + 3) & 3] = xload(ks + 3);
    }
    const bf16x8 xh = __builtin_bit_cast(bf16x8, xq[j].h), xl = __builtin_bit_cast(bf16x8, xq[j].l);
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ring + (ks & (F_RING - 1)) * F_SLOT_FLOATS) + lane;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const bf16x8 wh = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 0) * 64]);
      const bf16x8 wl = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 1) * 64]);
      acc[ob] = MFMA_BF16(wh, xh, acc[ob]);
      acc[ob] = MFMA_BF16(wh, xl, acc[ob]);
      acc[ob] = MFMA_BF16(wl, xh, acc[ob]);
    }
  }
  __builtin_amdgcn_s_barrier();  // every wavefront is done with the ring: the epilogues use it as scratch
}

// column validity of the tile as two 64-bit masks in LDS (bit c of word c / 64 <-> local column c), shifted per lane so that
// the bit of accumulator register `reg` of block `ob` sits at the compile-time position 32 ob + (reg & 3) + 8 (reg >> 2)
__device__ __forceinline__ void column_masks(const FArgs& a, int p, int chunk, unsigned long long* sm_mask, unsigned long long (&mk)[2]) {
  const int tid = threadIdx.x;
  if (tid < 128) {
    const int col = chunk * FT + tid;
    const bool ok = col < a.N && (a.pt_mask ? a.pt_mask[(size_t)p * a.N + col] != 0 : true);
    const unsigned long long b = __builtin_amdgcn_ballot_w64(ok);
    if ((tid & 63) == 0) sm_mask[tid >> 6] = b;
  }
  __syncthreads();
  const int sh = 4 * ((tid & 63) >> 5);
  mk[0] = sm_mask[0] >> sh;
  mk[1] = sm_mask[1] >> sh;
}
__device__ __forceinline__ bool col_ok(const unsigned long long (&mk)[2], int ob, int reg) {
  const int c = 32 * ob + (reg & 3) + 8 * (reg >> 2);  // + 4 hi is in the shift
  return (mk[c >> 6] >> (c & 63)) & 1ull;
}

// e = exp(sim - |scale|) of a valid entry, 0 of a masked one
__device__ __forceinline__ float e_value(float dot, float s2, float shift, bool ok) {
  return ok ? __builtin_amdgcn_exp2f(NM_FMA(dot, s2, -shift)) : 0.f;
}
// softmax(sim, dim=1) * softmax(sim, dim=2); an all-masked row (column) has the uniform soft-max 1/N (1/M) in the reference,
// which survives in the product only where the column (row) is all-masked too
__device__ __forceinline__ float conf_value(float e, float ics, float irs, float inv_mn) {
  const float c = (e * ics) * (e * irs);
  return (ics < 0.f || irs < 0.f) ? ((ics < 0.f && irs < 0.f) ? inv_mn : 0.f) : c;
}

// Column reduction (sum or maximum over the 32 rows = lanes of a half) of the 64 values a lane holds, by recursive halving: a
// step pairs two registers and two lanes; each lane keeps one register of the pair (by one bit of its number), receives the
// partner's copy of the SAME register through DPP and combines: half as many registers after every step.  184 VALU
// instructions for 64 columns (2 v_cndmask + 1 DPP op per pair; the 16 <-> 16 step is gfx950's v_permlane16_swap + 1 op) against
// 320 for five DPP steps on every register.  Steps: lane ^ 1, lane ^ 2 (quad_perm), rotate by 4 and by 8 within the row of 16
// (a rotation is not an involution, but bits 0..1 [0..2] of sender and receiver agree, which is all the pairing needs, and the
// two receivers of a register cover its four [two] sources once), rows 0 <-> 1 / 2 <-> 3.
// Result: out[j] = the reduction of value index i = 32 j + (lane & 31), i.e. of acc[i >> 4][i & 15].
template <int CTRL>
__device__ __forceinline__ float dpp_read(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <bool MAX>
__device__ __forceinline__ float red2(float x, float y) {
  // the maximum is taken on the bit patterns: the values are confidences >= +0 (or NaN in columns outside the matrix, which are
  // dropped whole), for which the integer order is the float order, and v_max_i32 folds into the DPP instruction where v_max_f32 on
  // a value of unknown origin would first be canonicalised (one more instruction per value)
  if (MAX) return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), __builtin_bit_cast(int, y)));
  return x + y;
}
template <bool MAX, int CTRL, int NOUT>
__device__ __forceinline__ void halve(const float* in, float* out, bool bit) {
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    const float x = in[2 * j], y = in[2 * j + 1];
    const float keep = bit ? y : x, send = bit ? x : y;
    out[j] = red2<MAX>(keep, dpp_read<CTRL>(send));
  }
}
template <bool MAX>
__device__ __forceinline__ void column_reduce(const f32x16 (&acc)[4], int lane, float (&out)[2]) {
  float v[64], w1[32], w2[16], w3[8], w4[4];
#pragma unroll
  for (int i = 0; i < 64; ++i) v[i] = acc[i >> 4][i & 15];
  halve<MAX, 0xB1, 32>(v, w1, lane & 1);    // quad_perm:[1,0,3,2]
  halve<MAX, 0x4E, 16>(w1, w2, lane & 2);   // quad_perm:[2,3,0,1]
  halve<MAX, 0x124, 8>(w2, w3, lane & 4);   // row_ror:4
  halve<MAX, 0x128, 4>(w3, w4, lane & 8);   // row_ror:8
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float x = w4[2 * j], y = w4[2 * j + 1];
    // odd rows of x <-> even rows of y: x = (x.r0, y.r0, x.r2, y.r2), y = (x.r1, y.r1, x.r3, y.r3)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    out[j] = red2<MAX>(x, y);
  }
}
// local column of out[j] of column_reduce in the lane
__device__ __forceinline__ int reduced_column(int lane, int j) {
  const int i = 32 * j + (lane & 31);
  return 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * (lane >> 5);
}

// PASS 1: partial sums of e.  PASS 2: conf, per-tile row maximum / first column, column maxima.
// Grid: y = pair, x = XCD-aware tile id: consecutive workgroup ids go round robin to the 8 XCDs (one 4 MiB L2 each), so
//   id -> xcd = id % 8, j = id / 8, row tile = xcd * rpx + j % rpx, chunk = j / rpx      (rpx = ceil(tiles_m / 8))
// gives every XCD a band of rpx row tiles (640 KiB of image rows at 4800 tokens: resident in its L2 for the whole launch) and
// walks the point chunks once, each 128 KiB chunk serving the band's rpx row tiles back to back.  With the GEMM's mapping (all
// chunks of one row tile in a row) every XCD streamed the whole 4.9 MB point blob once per ROW TILE: 110 MB of L2 <-> fabric
// traffic per pair, L2 hit rate 0.79, wavefronts waiting half of their cycles (profiles/r3_pmc_match_tile*_v1.json).
template <int PASS>
__global__ void __launch_bounds__(256, 4) match_tile_kernel(FArgs a) {
  __shared__ __attribute__((aligned(16))) float ring[F_RING * F_SLOT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int p = blockIdx.y;
  const int rpx = (a.tiles_m + 7) >> 3, j = blockIdx.x >> 3;
  const int chunk = j / rpx, row_tile = (blockIdx.x & 7) * rpx + j % rpx;
  if (row_tile >= a.tiles_m) return;
  f32x16 acc[4];
  sim_tile(a, p, row_tile, chunk, ring, acc);
  const int m = row_tile * FT + wave * 32 + r;
  float* scr = ring;  // scratch: [4 wavefronts][128 columns] partial column results, then per-column vectors of the tile
  // The epilogues are VALU work on 64 values per lane while the matrix pipe idles (ablation, DESIGN.md 3.5: no epilogue -45 %),
  // so everything per-column is folded into vectors in LDS (one 16-byte read per 4 values) instead of per-value selects:
  //   nsh[c] = -|scale| log2e of a valid column, -inf of a masked one or one outside the matrix: e = exp2(dot s2 + nsh) is 0 there
  if (tid < FT) {
    const int col = chunk * FT + tid;
    const bool ok = col < a.N && (a.pt_mask ? a.pt_mask[(size_t)p * a.N + col] != 0 : true);
    scr[4 * FT + tid] = ok ? -a.shift : -__builtin_inff();
    if constexpr (PASS == 2) {
      //   ic[c] = reciprocal column sum; 0 for a column without an unmasked entry; NaN outside the matrix: conf = NaN never wins a
      //           maximum (v_max returns the other operand, == is false)
      //   dc[c] = 1 for a column without an unmasked entry (conf there is 1 / (M N) in rows without one either, cf. conf_value)
      const float ics = col < a.N ? a.ics[(size_t)p * a.N + col] : 0.f;
      scr[5 * FT + tid] = col < a.N ? (ics > 0.f ? ics : 0.f) : __builtin_nanf("");
      scr[6 * FT + tid] = (col < a.N && ics < 0.f) ? 1.f : 0.f;
    }
  }
  __syncthreads();
  if constexpr (PASS == 1) {
    const bool row_ok = m < a.M && (a.im_mask ? a.im_mask[(size_t)p * a.M + m] != 0 : true);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ns = *reinterpret_cast<const f32x4*>(scr + 4 * FT + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) acc[ob][4 * q + e4] = __builtin_amdgcn_exp2f(NM_FMA(acc[ob][4 * q + e4], a.s2, ns[e4]));
      }
    if (__builtin_amdgcn_ballot_w64(row_ok) != ~0ull) {  // (wavefront-uniform: only the last row tile, or with an image mask)
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ob][i] = row_ok ? acc[ob][i] : 0.f;
    }
    float rs = 0.f;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int i = 0; i < 16; ++i) rs += acc[ob][i];
    rs += nm_shfl_xor32(rs);
    if (hi == 0 && m < a.M) a.rpart[((size_t)p * a.tiles_n + chunk) * a.M + m] = rs;
    {
      float cs[2];
      column_reduce<false>(acc, lane, cs);
      scr[wave * FT + reduced_column(lane, 0)] = cs[0];
      scr[wave * FT + reduced_column(lane, 1)] = cs[1];
    }
    __syncthreads();
    if (tid < FT) {
      const int col = chunk * FT + tid;
      if (col < a.N) a.cpart[((size_t)p * a.tiles_m + row_tile) * a.N + col] = ((scr[tid] + scr[FT + tid]) + scr[2 * FT + tid]) + scr[3 * FT + tid];
    }
  } else {
    // a row without an unmasked entry (or outside the matrix) has irs = 0: conf = 0 whatever e is (e <= 1 is finite), so the
    // image mask is not needed here; dr = 1 / (M N) marks such a row of the matrix for the dc term
    const float irs_g = m < a.M ? a.irs[(size_t)p * a.M + m] : 0.f;
    const float irs = irs_g > 0.f ? irs_g : 0.f, dr = irs_g < 0.f ? a.inv_mn : 0.f;
    float best = -1.f;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 ns = *reinterpret_cast<const f32x4*>(scr + 4 * FT + 32 * ob + 8 * q + 4 * hi);
        const f32x4 ic = *reinterpret_cast<const f32x4*>(scr + 5 * FT + 32 * ob + 8 * q + 4 * hi);
        const f32x4 dc = *reinterpret_cast<const f32x4*>(scr + 6 * FT + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const float e = __builtin_amdgcn_exp2f(NM_FMA(acc[ob][4 * q + e4], a.s2, ns[e4]));
          acc[ob][4 * q + e4] = NM_FMA(dr, dc[e4], (e * ic[e4]) * (e * irs));  // == conf_value(): fma(0, 0, x) = x
        }
        best = __builtin_fmaxf(__builtin_fmaxf(best, __builtin_fmaxf(acc[ob][4 * q], acc[ob][4 * q + 1])),
                               __builtin_fmaxf(acc[ob][4 * q + 2], acc[ob][4 * q + 3]));
      }
    // first column holding the maximum (descending, so that the smallest is written last) and whether several hold it: the
    // comparison masks are wavefront-wide scalars, so "seen twice" is two SALU instructions per value beside the VALU stream
    int bidx = 0;
    unsigned long long once = 0ull, twice = 0ull;
#pragma unroll
    for (int ob = 3; ob >= 0; --ob)
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        // four compares, then four selects: written out because the compiler turns select(c == best, k, bidx) into a second,
        // inverted compare per value and puts an s_nop between each compare and its select (SGPR written by VALU -> VALU mask)
        unsigned long long m0, m1, m2, m3;
        asm("v_cmp_eq_f32_e64 %[m3], %[c3], %[b]\n\t"
            "v_cmp_eq_f32_e64 %[m2], %[c2], %[b]\n\t"
            "v_cmp_eq_f32_e64 %[m1], %[c1], %[b]\n\t"
            "v_cmp_eq_f32_e64 %[m0], %[c0], %[b]\n\t"
            "v_cndmask_b32_e64 %[x], %[x], %[k3], %[m3]\n\t"
            "v_cndmask_b32_e64 %[x], %[x], %[k2], %[m2]\n\t"
            "v_cndmask_b32_e64 %[x], %[x], %[k1], %[m1]\n\t"
            "v_cndmask_b32_e64 %[x], %[x], %[k0], %[m0]"
            : [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [x] "+v"(bidx)
            : [c0] "v"(acc[ob][4 * q]), [c1] "v"(acc[ob][4 * q + 1]), [c2] "v"(acc[ob][4 * q + 2]), [c3] "v"(acc[ob][4 * q + 3]), [b] "v"(best),
              [k0] "n"(16 * ob + 4 * q), [k1] "n"(16 * ob + 4 * q + 1), [k2] "n"(16 * ob + 4 * q + 2), [k3] "n"(16 * ob + 4 * q + 3));
        twice |= (once & m3); once |= m3;
        twice |= (once & m2); once |= m2;
        twice |= (once & m1); once |= m1;
        twice |= (once & m0); once |= m0;
      }
    const int cnt = ((twice >> lane) & 1ull) ? 2 : 1;
    bidx = chunk * FT + 32 * (bidx >> 4) + 8 * ((bidx & 15) >> 2) + (bidx & 3) + 4 * hi;
    int mult = cnt > 1;
    {
      const float ob_ = nm_shfl_xor32(best);
      const int oi = __shfl_xor(bidx, 32, 64), om = __shfl_xor(mult, 32, 64);
      if (ob_ > best) { best = ob_; bidx = oi; mult = om; }
      else if (ob_ == best) { mult = 1; bidx = oi < bidx ? oi : bidx; }
    }
    if (hi == 0 && m < a.M) {
      a.rbest[((size_t)p * a.tiles_n + chunk) * a.M + m] = best;
      a.ridx[((size_t)p * a.tiles_n + chunk) * a.M + m] = bidx | (mult ? (int)0x80000000 : 0);
    }
    // column maxima: rows outside the matrix hold conf = 0 (irs = 0) or NaN, neither of which changes a maximum of values >= 0
    // (the partial maxima go to scr[0 .. 4 FT): no overlap with the vectors at 4 FT .. 7 FT other wavefronts may still be reading)
    {
      float cm[2];
      column_reduce<true>(acc, lane, cm);
      scr[wave * FT + reduced_column(lane, 0)] = cm[0];
      scr[wave * FT + reduced_column(lane, 1)] = cm[1];
    }
    __syncthreads();
    if (tid < FT) {
      const int col = chunk * FT + tid;
      if (col < a.N) {
        const float mx = fmaxf(fmaxf(scr[tid], scr[FT + tid]), fmaxf(scr[2 * FT + tid], scr[3 * FT + tid]));
        atomicMax(a.colmax + (size_t)p * a.N + col, __float_as_uint(mx));
      }
    }
  }
}

// reciprocal sums; grid (ceil(max(M, N) / 256), P, 2): z = 0 rows, 1 columns
__global__ void __launch_bounds__(256) match_merge_kernel(FArgs a) {
  const int p = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.z == 0) {
    if (i >= a.M) return;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < a.tiles_n; ++c) s += a.rpart[((size_t)p * a.tiles_n + c) * a.M + i];
    a.irs[(size_t)p * a.M + i] = s > 0.f ? 1.0f / s : -1.f;
  } else {
    if (i >= a.N) return;
    float s = 0.f;
#pragma unroll 8
    for (int t = 0; t < a.tiles_m; ++t) s += a.cpart[((size_t)p * a.tiles_m + t) * a.N + i];
    a.ics[(size_t)p * a.N + i] = s > 0.f ? 1.0f / s : -1.f;
    a.colmax[(size_t)p * a.N + i] = 0u;  // pass 2's column maxima start from +0 (round 5: was a memset launch of its own)
  }
}

// grid (ceil(M / 64), P), block 256: lane = row, the four wavefronts take every fourth tile and wavefront 0 combines (round 5: one thread per
// row walked the tiles_n partial results alone -- 24 us for 4800 rows, a tenth of a one-query matching call).  The combination is exact
// whatever the split: the maximum is a maximum, and whenever it is attained in more than one tile (or more than once inside one) the row
// goes to the tie pass, which does not look at `idx`.
__global__ void __launch_bounds__(256) match_select_kernel(FArgs a) {
  __shared__ float s_v[4][64];
  __shared__ int s_i[4][64];
  const int p = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = blockIdx.x * 64 + lane;
  float v = -1.f;
  int idx = 0, several = 0;
  if (i < a.M) {
    for (int c = wave; c < a.tiles_n; c += 4) {
      const float b = a.rbest[((size_t)p * a.tiles_n + c) * a.M + i];
      const int ri = a.ridx[((size_t)p * a.tiles_n + c) * a.M + i];
      if (b > v) { v = b; idx = ri & 0x7fffffff; several = ri < 0; }
      else if (b == v) several = 1;
    }
  }
  s_v[wave][lane] = v;
  s_i[wave][lane] = idx | (several ? (int)0x80000000 : 0);
  __syncthreads();
  if (wave != 0 || i >= a.M) return;
  v = -1.f; idx = 0; several = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float b = s_v[w][lane];
    const int ri = s_i[w][lane];
    if (b > v) { v = b; idx = ri & 0x7fffffff; several = ri < 0; }
    else if (b == v) several = 1;
  }
  int sel = -1, tie = 0;
  if (v > a.thr) {
    if (several) { tie = 1; sel = INT_MAX; }
    else if (!a.mutual || a.colmax[(size_t)p * a.N + idx] == __float_as_uint(v)) sel = idx;
  }
  a.sel_j[(size_t)p * a.M + i] = sel;
  a.sel_v[(size_t)p * a.M + i] = v;
  a.tie[(size_t)p * a.M + i] = tie;
}

// grid (tiles_m, P): nothing to do unless a row of the block is flagged
__global__ void __launch_bounds__(256, 3) match_tie_kernel(FArgs a) {
  __shared__ __attribute__((aligned(16))) float ring[F_RING * F_SLOT_FLOATS];
  __shared__ unsigned long long sm_mask[2];
  __shared__ int sm_any;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int p = blockIdx.y, row_tile = blockIdx.x;
  const int m = row_tile * FT + wave * 32 + r;
  const bool mine = m < a.M && a.tie[(size_t)p * a.M + m] != 0;
  if (tid == 0) sm_any = 0;
  __syncthreads();
  if (mine) sm_any = 1;
  __syncthreads();
  if (!sm_any) return;
  const bool row_ok = m < a.M && (a.im_mask ? a.im_mask[(size_t)p * a.M + m] != 0 : true);
  const float irs = m < a.M ? a.irs[(size_t)p * a.M + m] : 0.f;
  const float v = m < a.M ? a.sel_v[(size_t)p * a.M + m] : 0.f;
  int first = INT_MAX;
  float* scr = ring;
  for (int chunk = 0; chunk < a.tiles_n; ++chunk) {
    f32x16 acc[4];
    sim_tile(a, p, row_tile, chunk, ring, acc);
    unsigned long long mk[2];
    column_masks(a, p, chunk, sm_mask, mk);
    if (tid < FT) {
      const int col = chunk * FT + tid;
      scr[4 * FT + tid] = col < a.N ? a.ics[(size_t)p * a.N + col] : 0.f;
      scr[5 * FT + tid] = col < a.N ? __uint_as_float(a.colmax[(size_t)p * a.N + col]) : -2.f;
    }
    __syncthreads();
    if (mine) {
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 ic = *reinterpret_cast<const f32x4*>(scr + 4 * FT + 32 * ob + 8 * q + 4 * hi);
          const f32x4 cm = *reinterpret_cast<const f32x4*>(scr + 5 * FT + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const int i = 4 * q + e4, col = chunk * FT + 32 * ob + 8 * q + 4 * hi + e4;
            const float e = e_value(acc[ob][i], a.s2, a.shift, row_ok && col_ok(mk, ob, i));
            const float c = col < a.N ? conf_value(e, ic[e4], irs, a.inv_mn) : -1.f;
            const bool pass = c > a.thr && c == v && (!a.mutual || c == cm[e4]);
            if (pass && col < first) first = col;
          }
        }
    }
    __syncthreads();  // scratch is the ring: the next tile's DMA must not start before everybody has read it
  }
  if (mine && first != INT_MAX) atomicMin(a.sel_j + (size_t)p * a.M + m, first);
}

// Rows -> ready-made MFMA operands: f / (|f| + 1e-6) (the summation order of match.hip's l2norm_kernel: one wavefront per row,
// lane l sums channels l, l + 64, ..., then the xor tree), split into bf16 hi / lo, written as [K-step][hi, lo][lane' = (row & 31) +
// 32 half][8 values k = 16 ks + 8 half + i].  One workgroup per group of 32 rows, so that every 1 KiB (K-step, hi / lo) piece
// leaves as ONE coalesced store of 64 x 16 bytes (a first version, one wavefront per row writing 2-byte elements, ran at a third
// of the memory rate: 48 us per side and 16 pairs).  Rows beyond the matrix are zero.
//   image side (POINTS = false): piece (group, ks, hl) at ((p groups + group) nks + ks) 2 KiB + hl KiB          -- B operands
//   point side (POINTS = true):  slot (chunk = group / 4, ks) of 8 KiB, piece ((group & 3) 2 + hl) KiB          -- A-operand slots
// grid (groups, P), C = 64 PER
template <int PER, bool POINTS>
__device__ __forceinline__ void norm_pack32_body(const float* __restrict__ x, int rows, int groups, int nks, char* __restrict__ blob, int g, float* s_inv) {
  const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xg = x + ((size_t)p * rows + (size_t)g * 32) * 64 * PER;
#pragma unroll 1
  for (int rr = 0; rr < 8; ++rr) {
    const int r = wave * 8 + rr;
    float q = 0.f;
    if (g * 32 + r < rows) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const float v = xg[(size_t)r * 64 * PER + lane + 64 * i];
        q = NM_FMA(v, v, q);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (lane == 0) s_inv[r] = sqrtf(q) + 1e-6f;
  }
  __syncthreads();
  const int r = lane & 31, half = lane >> 5;
  const bool live = g * 32 + r < rows;
  const float den = s_inv[r];
  char* base = POINTS ? blob + (((size_t)p * (groups >> 2) + (g >> 2)) * nks) * F_SLOT_BYTES + (size_t)(g & 3) * 2048
                      : blob + (((size_t)p * groups + g) * nks) * 2048;
  constexpr size_t KS_STRIDE = POINTS ? F_SLOT_BYTES : 2048;
  for (int ks = wave; ks < nks; ks += 4) {
    float v8[8];
    if (live) {
      const float* src = xg + (size_t)r * 64 * PER + 16 * ks + 8 * half;
      const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
      v8[0] = a[0]; v8[1] = a[1]; v8[2] = a[2]; v8[3] = a[3]; v8[4] = b[0]; v8[5] = b[1]; v8[6] = b[2]; v8[7] = b[3];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v8[i] = 0.f;
    }
    bf16x8 h8, l8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float f = v8[i] / den;
      const __bf16 h = (__bf16)f;
      h8[i] = h;
      l8[i] = (__bf16)(f - (float)h);
    }
    u32x4* d = reinterpret_cast<u32x4*>(base + (size_t)ks * KS_STRIDE) + lane;
    d[0] = __builtin_bit_cast(u32x4, h8);
    d[64] = __builtin_bit_cast(u32x4, l8);
  }
}
// both sides of a pair in ONE launch (round 5; two launches of ~150 workgroups each before): workgroups [0, gi) take the image rows,
// [gi, gi + gp) the point rows
template <int PER>
__global__ void __launch_bounds__(256) norm_pack32_kernel(const float* __restrict__ im, int M, int gi, const float* __restrict__ pt, int N, int gp, int nks,
                                                          char* __restrict__ imb, char* __restrict__ blob) {
  __shared__ float s_inv[32];
  if ((int)blockIdx.x < gi) norm_pack32_body<PER, false>(im, M, gi, nks, imb, blockIdx.x, s_inv);
  else norm_pack32_body<PER, true>(pt, N, gp, nks, blob, blockIdx.x - gi, s_inv);
}

// ordered compaction, one workgroup per pair (cf. compact_kernel in match.hip)
__global__ void __launch_bounds__(1024) match_compact_kernel(const int* __restrict__ sel_j, const float* __restrict__ sel_v, int M,
                                                              int64_t* __restrict__ out_i, int64_t* __restrict__ out_j,
                                                              float* __restrict__ out_conf, int* __restrict__ count) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int p = blockIdx.x;
  sel_j += (size_t)p * M; sel_v += (size_t)p * M; out_i += (size_t)p * M; out_j += (size_t)p * M; out_conf += (size_t)p * M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int start = 0; start < M; start += 1024) {
    const int i = start + tid;
    const int sj = i < M ? sel_j[i] : -1;
    const int has = (sj >= 0 && sj != INT_MAX) ? 1 : 0;
    const unsigned long long bal = __ballot(has);
    const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    int total = 0;
    for (int w = 0; w < 16; ++w) total += s_wave[w];
    const int base = s_base;
    if (has) {
      const int o = base + woff + prefix;
      out_i[o] = i;
      out_j[o] = sj;
      out_conf[o] = sel_v[i];
    }
    __syncthreads();
    if (tid == 0) s_base = base + total;
    __syncthreads();
  }
  // the slots behind the count: index 0 / confidence 0 (round 5: the single-pair path's speculative fine stage reads the first `cap` slots as
  // indices before the count is known -- this used to be a fill launch of the caller's)
  for (int i = s_base + tid; i < M; i += 1024) { out_i[i] = 0; out_j[i] = 0; out_conf[i] = 0.f; }
  if (tid == 0) count[p] = s_base;
}

size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

struct FWork {
  float *rpart, *cpart, *irs, *ics, *rbest, *sel_v;
  char* imb;
  int *ridx, *sel_j, *tie;
  unsigned* colmax;
  char* blob;
  size_t bytes;
};
FWork carve(void* base, int P, int M, int N, int C) {
  FWork w{};
  size_t off = 0;
  auto take = [&](size_t n) {
    void* q = base ? (char*)base + off : nullptr;
    off += al256(n);
    return q;
  };
  const int tm = (M + FT - 1) / FT, tn = (N + FT - 1) / FT, nks = C / 16;
  w.imb = (char*)take((size_t)P * ((M + 31) / 32) * nks * 2048);
  w.blob = (char*)take((size_t)P * tn * nks * F_SLOT_BYTES);
  w.rpart = (float*)take((size_t)P * tn * M * 4);
  w.cpart = (float*)take((size_t)P * tm * N * 4);
  w.rbest = (float*)take((size_t)P * tn * M * 4);
  w.ridx = (int*)take((size_t)P * tn * M * 4);
  w.irs = (float*)take((size_t)P * M * 4);
  w.ics = (float*)take((size_t)P * N * 4);
  w.colmax = (unsigned*)take((size_t)P * N * 4);
  w.sel_j = (int*)take((size_t)P * M * 4);
  w.sel_v = (float*)take((size_t)P * M * 4);
  w.tie = (int*)take((size_t)P * M * 4);
  w.bytes = off;
  return w;
}

}  // namespace

extern "C" size_t nm_match_fused_workspace_bytes(int P, int M, int N, int C) {
  if (P <= 0 || M <= 0 || N <= 0 || C <= 0) return 0;
  return carve(nullptr, P, M, N, C).bytes;
}

extern "C" int nm_dual_softmax_match_fused(const float* im, const float* pt, int P, int M, int N, int C, float scale,
                                           const uint8_t* im_mask, const uint8_t* pt_mask, float threshold, int mutual, int64_t* out_i,
                                           int64_t* out_j, float* out_conf, int* counts, void* workspace, size_t workspace_bytes,
                                           nmStream_t stream) {
  NM_CHECK_ARG(im && pt && out_i && out_j && out_conf && counts && workspace && P > 0 && M > 0 && N > 0);
  if (C != 64 && C != 128 && C != 256 && C != 512) return NM_ERR_UNSUPPORTED;
  if (!(fabsf(scale) * LOG2E <= F_MAX_SHIFT) || P > 65535) return NM_ERR_UNSUPPORTED;  // (also refuses NaN)
  FWork w = carve(workspace, P, M, N, C);
  if (workspace_bytes < w.bytes) return NM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  FArgs a{};
  a.imb = w.imb; a.blob = w.blob; a.im_mask = im_mask; a.pt_mask = pt_mask;
  a.M = M; a.N = N; a.C = C; a.nks = C / 16; a.tiles_m = (M + FT - 1) / FT; a.tiles_n = (N + FT - 1) / FT;
  a.s2 = scale * LOG2E; a.shift = fabsf(scale) * LOG2E;
  a.rpart = w.rpart; a.cpart = w.cpart; a.irs = w.irs; a.ics = w.ics; a.inv_mn = (1.0f / (float)M) * (1.0f / (float)N);
  a.rbest = w.rbest; a.ridx = w.ridx; a.colmax = w.colmax; a.thr = threshold; a.mutual = mutual;
  a.sel_j = w.sel_j; a.sel_v = w.sel_v; a.tie = w.tie;
  const int gi = (M + 31) / 32, gp = a.tiles_n * 4;  // groups of 32 rows (the point side is padded to whole 128-row chunks)
  const dim3 gn(gi + gp, P);
  switch (C) {
    case 64: norm_pack32_kernel<1><<<gn, 256, 0, s>>>(im, M, gi, pt, N, gp, a.nks, w.imb, w.blob); break;
    case 128: norm_pack32_kernel<2><<<gn, 256, 0, s>>>(im, M, gi, pt, N, gp, a.nks, w.imb, w.blob); break;
    case 256: norm_pack32_kernel<4><<<gn, 256, 0, s>>>(im, M, gi, pt, N, gp, a.nks, w.imb, w.blob); break;
    default: norm_pack32_kernel<8><<<gn, 256, 0, s>>>(im, M, gi, pt, N, gp, a.nks, w.imb, w.blob); break;
  }
  const dim3 gt((unsigned)(((a.tiles_m + 7) / 8) * 8 * a.tiles_n), P);
  match_tile_kernel<1><<<gt, 256, 0, s>>>(a);
  const int mx = M > N ? M : N;
  match_merge_kernel<<<dim3((mx + 255) / 256, P, 2), 256, 0, s>>>(a);
  match_tile_kernel<2><<<gt, 256, 0, s>>>(a);
  match_select_kernel<<<dim3((M + 63) / 64, P), 256, 0, s>>>(a);
  match_tie_kernel<<<dim3(a.tiles_m, P), 256, 0, s>>>(a);
  match_compact_kernel<<<P, 1024, 0, s>>>(w.sel_j, w.sel_v, M, out_i, out_j, out_conf, counts);
  return nm_launch_status();
}
