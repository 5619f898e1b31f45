// y[M,N] = act(x[M,K] . w[N,K]^T + bias) + residual  on the bf16 matrix cores with fp32-accurate operand splitting.
//
// Same contract as nm_linear (gemm.hip); the arithmetic is that of nerf_fwd_bf16.hip: every fp32 operand is split into
// two bf16 values (x = hi + lo) and each product is w_hi*x_hi + w_hi*x_lo + w_lo*x_hi on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation (error ~1e-6 relative; 3/16 of the fp32-MFMA time).
//
//   * the weights are constant per module, so they are split and laid out once (nm_linear_pack_bf16x3, on the device)
//     into 8 KiB "slots" = one 16-wide K-step for a chunk of 128 output features, in A-operand order;
//   * a workgroup = 4 wavefronts x 32 rows computes a 128-row x 128-column tile: the slots are streamed through a
//     4-slot LDS ring with global_load_lds two K-steps ahead (one counted s_waitcnt + one s_barrier per K-step), the
//     rows of x go global -> registers two K-steps ahead and are split on the fly;
//   * 64 accumulator registers per wavefront: 3 waves/SIMD, so neighbouring workgroups hide each other's LDS and
//     memory latency -- no hand scheduling here (contrast nerf_fwd_bf16.hip, which runs one wave per SIMD);
//   * result layout lane = row, register = feature.  Stored like that (every lane 16-byte pieces of its own row, 32 bytes
//     per row and instruction) the kernel ran at 2-3 TB/s: that store pattern, harmless on its own, halves the throughput
//     as soon as it is MIXED with loads (scripts/ubench/access_pattern.hip: 3.7 TB/s for read + lane=row write against 6.6
//     with coalesced writes).  The epilogue therefore transposes the tile through the (by then idle) LDS ring, half a
//     tile at a time, and writes 256-byte row segments; `pre` / residual / `gate` are read in the same coalesced layout,
//     the bias is the accumulators' starting value.
#include "common.h"
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int GB_ROWS = 128;         // rows per workgroup
constexpr int GB_COLS = 128;         // output features per workgroup (4 blocks of 32)
constexpr int GB_SLOT_BYTES = 8192;  // 4 blocks x (hi, lo) x 64 lanes x 16 bytes
constexpr int GB_SLOT_FLOATS = GB_SLOT_BYTES / 4;
constexpr int GB_RING = 4;

struct GemmBArgs {
  const float* x;
  const char* blob;
  const float* bias;
  const float* res;
  const float* pre;   // added before the activation (or NULL)
  const float* gate;  // output multiplied by [gate > 0] (or NULL)
  float* y;
  int M, N, K, act, nks;
  // similarity mode (dual-softmax matching): y = mask_fill(scale * x . w^T)
  int sim;
  float scale;
  const uint8_t* row_mask;
  const uint8_t* col_mask;
  int fast_epi;  // 1: transposed (coalesced) epilogue; 0: register-layout epilogue (pre / gate present)
  // fused q|k|v (or k|v) projection for attn32_v2_kernel (nm_linear_qkv_bf16x3): columns [0, n_q) go to y (row stride n_q),
  // columns [n_q, n_q + 32 H) are the keys and [n_q + 32 H, n_q + 64 H) the values, written split and laid out as that
  // kernel's MFMA operands (attention_v2.hip: 8 KiB slot per (batch, head, 32-key tile)) instead of as fp32 rows
  char* kv_blob;
  int n_q, S, H, chunk0, nchunks;
};

__device__ __forceinline__ float gelu_erf_b(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// two 1 KiB pieces per wavefront: one address / one M0, told apart by the immediate offset
__device__ __forceinline__ void dma_slot(const char* slots, int g, float* ring, int wave, int lane) {
  const unsigned voff = (unsigned)(wave * 2048 + lane * 16);
  const char* base = slots + (size_t)g * GB_SLOT_BYTES;
  const auto* src = (const __attribute__((address_space(1))) void*)(base + voff);
  auto* dst = (__attribute__((address_space(3))) void*)(ring + (g & (GB_RING - 1)) * GB_SLOT_FLOATS + wave * 512);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
}

// epilogue: register 4q+e of block ob <-> feature 32 ob + 8 q + 4 half + e of row m (n_base = first column of the chunk + 4 half)
__device__ __forceinline__ void epilogue(const GemmBArgs& a, const f32x16 (&acc)[4], int m, int n_base) {
  if (m >= a.M) return;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n0 = n_base + 32 * ob + 8 * q;
      if (n0 < a.N) {  // N is a multiple of 8: a 4-wide piece is inside or outside
        f32x4 v = {acc[ob][4 * q], acc[ob][4 * q + 1], acc[ob][4 * q + 2], acc[ob][4 * q + 3]};
        if (a.sim) {
          const bool rk = a.row_mask ? a.row_mask[m] != 0 : true;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool keep = rk && (a.col_mask ? a.col_mask[n0 + e] != 0 : true);
            v[e] = keep ? v[e] * a.scale : -1e9f;
          }
        }
        if (a.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n0);
          v = {v[0] + b[0], v[1] + b[1], v[2] + b[2], v[3] + b[3]};
        }
        if (a.pre) {
          const f32x4 pp = *reinterpret_cast<const f32x4*>(a.pre + (size_t)m * a.N + n0);
          v = {v[0] + pp[0], v[1] + pp[1], v[2] + pp[2], v[3] + pp[3]};
        }
        if (a.act == NM_ACT_RELU) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        else if (a.act == NM_ACT_GELU) v = {gelu_erf_b(v[0]), gelu_erf_b(v[1]), gelu_erf_b(v[2]), gelu_erf_b(v[3])};
        if (a.res) {
          const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + (size_t)m * a.N + n0);
          v = {v[0] + rr[0], v[1] + rr[1], v[2] + rr[2], v[3] + rr[3]};
        }
        if (a.gate) {
          const f32x4 gg = *reinterpret_cast<const f32x4*>(a.gate + (size_t)m * a.N + n0);
          v = {gg[0] > 0.f ? v[0] : 0.f, gg[1] > 0.f ? v[1] : 0.f, gg[2] > 0.f ? v[2] : 0.f, gg[3] > 0.f ? v[3] : 0.f};
        }
        *reinterpret_cast<f32x4*>(a.y + (size_t)m * a.N + n0) = v;
      }
    }
}

__device__ __forceinline__ f32x4 activate(f32x4 v, int act) {
  if (act == NM_ACT_RELU) return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
  if (act == NM_ACT_GELU) return f32x4{gelu_erf_b(v[0]), gelu_erf_b(v[1]), gelu_erf_b(v[2]), gelu_erf_b(v[3])};
  return v;
}

// Coalesced epilogue.  `tb` = this wavefront's 8 KiB of LDS (32 rows x 16 pieces of 16 bytes, piece index XOR-swizzled with
// the row so that both the row-wise writes and the piece-wise reads are conflict free).  The accumulators already hold the
// bias (bias_init).  m0 = first row of the wavefront's 32, n_chunk = first column of the 128-column chunk.
__device__ __forceinline__ void epilogue_coalesced(const GemmBArgs& a, const f32x16 (&acc)[4], float* tb, int lane, int m0, int n_chunk) {
  const int r = lane & 31, hi = lane >> 5;
  const int rrow = lane >> 4, rpiece = lane & 15;  // read side: row 4 i + rrow, piece rpiece
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int obl = 0; obl < 2; ++obl)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ob = 2 * h + obl;
        f32x4 v = {acc[ob][4 * q], acc[ob][4 * q + 1], acc[ob][4 * q + 2], acc[ob][4 * q + 3]};
        if (!a.pre) v = activate(v, a.act);  // (with a `pre` addend the activation follows it, in the store layout)
        const int p = obl * 8 + 2 * q + hi;
        *reinterpret_cast<f32x4*>(tb + r * 64 + ((p ^ (r & 15)) << 2)) = v;
      }
    const int n0 = n_chunk + 64 * h + 4 * rpiece;
    const int ncols = a.kv_blob ? a.n_q : a.N;  // columns (and row stride) of y
    if (n0 < ncols) {  // a multiple of 8, n0 of 4: a piece is inside or outside
      bool ck[4] = {true, true, true, true};
      if (a.sim && a.col_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ck[e] = a.col_mask[n0 + e] != 0;
      }
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {  // 16 rows at a time (register budget: 4 waves / SIMD)
        f32x4 v[4], rr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 4 * (4 * ib + j) + rrow;
          v[j] = *reinterpret_cast<const f32x4*>(tb + row * 64 + ((rpiece ^ (row & 15)) << 2));
        }
        auto fetch = [&](const float* src) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int m = m0 + 4 * (4 * ib + j) + rrow;
            rr[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(m < a.M ? m : a.M - 1) * ncols + n0);
          }
        };
        if (a.pre) {
          fetch(a.pre);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = activate(v[j] + rr[j], a.act);
        }
        if (a.res) {
          fetch(a.res);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += rr[j];
        }
        if (a.gate) {
          fetch(a.gate);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = rr[j][e] > 0.f ? v[j][e] : 0.f;
        }
        if (a.sim) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int m = m0 + 4 * (4 * ib + j) + rrow;
            const bool rk = a.row_mask ? a.row_mask[m < a.M ? m : a.M - 1] != 0 : true;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = (rk && ck[e]) ? v[j][e] * a.scale : -1e9f;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = m0 + 4 * (4 * ib + j) + rrow;
          if (m < a.M) *reinterpret_cast<f32x4*>(a.y + (size_t)m * ncols + n0) = v[j];
        }
      }
    }
  }
}

// Keys of the fused projection: accumulator layout lane = (key r of the wavefront's 32-key tile, half hi), register 4q+e of
// block ob = dim 8q + 4hi + e of head (head0 + ob).  attn32_v2's K operand of K-step ks = dim / 16 wants lane' = (key,
// (dim % 16) / 8) to hold dims 8 (dim / 8) .. +7: this lane owns 8 bytes of two neighbouring lanes' 16-byte operands, and the
// 32 keys x 2 halves of one (ob, q) make 512 contiguous bytes.
__device__ __forceinline__ void epilogue_keys(const f32x16 (&acc)[4], char* slot0, size_t head_stride, int lane) {
  const int r = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __bf16 h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[ob][4 * q + e];
        h[e] = (__bf16)v;
        l[e] = (__bf16)(v - (float)h[e]);
      }
      typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
      const u16x4 hv = {__builtin_bit_cast(unsigned short, h[0]), __builtin_bit_cast(unsigned short, h[1]),
                        __builtin_bit_cast(unsigned short, h[2]), __builtin_bit_cast(unsigned short, h[3])};
      const u16x4 lv = {__builtin_bit_cast(unsigned short, l[0]), __builtin_bit_cast(unsigned short, l[1]),
                        __builtin_bit_cast(unsigned short, l[2]), __builtin_bit_cast(unsigned short, l[3])};
      char* base = slot0 + ob * head_stride + (size_t)(((q >> 1) * 2) * 64 + (q & 1) * 32 + r) * 16 + 8 * hi;
      *reinterpret_cast<u16x4*>(base) = hv;
      *reinterpret_cast<u16x4*>(base + 64 * 16) = lv;
    }
}

// Values: computed with the MFMA operands swapped, so lane = (dim r of head head0 + ob, half hi) and register 4q+e = key
// 8q + 4hi + e of the tile -- exactly attn32_v2's V^T operand (k-slot i of step ks <-> key (i & 3) + 16 ks + 8 (i >> 2) + 4 half):
// one 16-byte store per (block, K-step, hi / lo part), 1 KiB contiguous per instruction.
__device__ __forceinline__ void epilogue_values(const f32x16 (&acc)[4], char* slot0, size_t head_stride, int lane) {
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 h8, l8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = acc[ob][4 * (2 * ks + (i >> 2)) + (i & 3)];
        const __bf16 h = (__bf16)v;
        h8[i] = h;
        l8[i] = (__bf16)(v - (float)h);
      }
      u32x4* dst = reinterpret_cast<u32x4*>(slot0 + ob * head_stride) + (4 + ks * 2) * 64 + lane;
      dst[0] = __builtin_bit_cast(u32x4, h8);
      dst[64] = __builtin_bit_cast(u32x4, l8);
    }
}

// accumulator start: register 4q+e of block ob <- bias[n_base + 32 ob + 8 q + e] on the coalesced path (the register-layout
// epilogue adds the bias itself), zero otherwise
__device__ __forceinline__ void bias_init(const GemmBArgs& a, f32x16 (&acc)[4], int n_base) {
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n0 = n_base + 32 * ob + 8 * q;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (a.fast_epi && a.bias && n0 < a.N) b = *reinterpret_cast<const f32x4*>(a.bias + n0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] = b[e];
    }
}

struct XRow {
  f32x4 a, b;  // x[m][16 ks + 8 half .. + 7]
};

// FUSED: 0 plain GEMM; 1 the q and key chunks of a fused projection; 2 its value chunks (transposed product)
template <int FUSED>
__global__ void __launch_bounds__(256, 4) gemm_bf16x3_kernel(GemmBArgs a) {
  __shared__ __attribute__((aligned(16))) float ring[GB_RING * GB_SLOT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  // XCD-aware 1-D grid: consecutive workgroup ids go round robin to the 8 XCDs (one L2 each), so
  //   id -> xcd = id % 8, g = id / 8, column chunk = g % chunks, row tile = 8 (g / chunks) + xcd:
  // the `chunks` workgroups that read the SAME 128 rows of x run back to back on ONE XCD and all but the first find the rows
  // in its L2 (with a (row tile, chunk) grid the chunks of a row tile were dispatched a whole pass apart and x was fetched
  // `chunks` times over the fabric -- 2x at N = 256, 6x for the fused q|k|v projection).  Measured effect: small (19.8 ->
  // 18.9 us at 19200 x 256 x 256): the re-reads were served by the Infinity Cache, the kernel stays latency-bound.
  // (fused projection: this launch covers the column chunks [chunk0, chunk0 + nchunks) of the stacked weight)
  const int chunks = FUSED ? a.nchunks : (a.N + GB_COLS - 1) / GB_COLS;
  const int g = blockIdx.x >> 3, chunk = (FUSED ? a.chunk0 : 0) + g % chunks, row_tile = 8 * (g / chunks) + (blockIdx.x & 7);
  if (row_tile * GB_ROWS >= a.M) return;  // (whole workgroup: the row tiles are padded to a multiple of 8)
  const int m = row_tile * GB_ROWS + wave * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;
  const int nks = a.nks;
  const char* slots = a.blob + (size_t)chunk * nks * GB_SLOT_BYTES;
  // fused projection: 0 = columns of y, 1 = keys, 2 = values (whole chunks: n_q and 32 H are multiples of 128)
  const int kind = FUSED == 0 ? 0 : FUSED == 2 ? 2 : (chunk * GB_COLS < a.n_q ? 0 : 1);
  const float* xp = a.x + (size_t)mc * a.K + 8 * hi;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto xload = [&](int ks) {
    XRow v;
    const int k = 16 * ks + 8 * hi;
    if (k < a.K) {  // K is a multiple of 8: an 8-wide piece is inside the row or entirely outside (K % 16 == 8 tail)
      v.a = *reinterpret_cast<const f32x4*>(xp + 16 * ks);
      v.b = *reinterpret_cast<const f32x4*>(xp + 16 * ks + 4);
    } else {
      v.a = zero4;
      v.b = zero4;
    }
    return v;
  };
  // (the bias loads only ADD younger operations to the loop's counted waits, which stay conservative wherever the compiler
  // places them: slot ks is older than at least the 4 operations of K-step ks+1)
  f32x16 acc[4];
  bias_init(a, acc, chunk * GB_COLS + 4 * hi);

  // prologue: slots / x pieces of K-steps 0 and 1 (same issue order as the loop: DMA, then x)
  dma_slot(slots, 0, ring, wave, lane);
  XRow x0 = xload(0);
  XRow x1 = x0;
  if (nks > 1) {
    dma_slot(slots, 1, ring, wave, lane);
    x1 = xload(1);
  }
  for (int ks = 0; ks < nks; ++ks) {
    // slot ks and x piece ks have landed when at most the 4 VMEM operations of K-step ks+1 remain in flight
    if (ks + 1 < nks) NM_WAIT_VMCNT(4);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody's pieces of slot ks landed; nobody reads slot ks-2 any more
    XRow x2 = x1;
    if (ks + 2 < nks) {
      dma_slot(slots, ks + 2, ring, wave, lane);
      x2 = xload(ks + 2);
    }
    bf16x8 xh, xl;
    {
      const float v8[8] = {x0.a[0], x0.a[1], x0.a[2], x0.a[3], x0.b[0], x0.b[1], x0.b[2], x0.b[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v8[i];
        xh[i] = h;
        xl[i] = (__bf16)(v8[i] - (float)h);
      }
    }
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ring + (ks & (GB_RING - 1)) * GB_SLOT_FLOATS) + lane;
    if (kind == 2) {  // (workgroup-uniform) transposed product: rows of x as the A operand
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 0) * 64]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 1) * 64]);
        acc[ob] = MFMA_BF16(xh, wh, acc[ob]);
        acc[ob] = MFMA_BF16(xl, wh, acc[ob]);
        acc[ob] = MFMA_BF16(xh, wl, acc[ob]);
      }
    } else {
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 0) * 64]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, s4[(ob * 2 + 1) * 64]);
        acc[ob] = MFMA_BF16(wh, xh, acc[ob]);
        acc[ob] = MFMA_BF16(wh, xl, acc[ob]);
        acc[ob] = MFMA_BF16(wl, xh, acc[ob]);
      }
    }
    x0 = x1;
    x1 = x2;
  }

  if (kind) {
    // this wavefront's 32 rows are the keys 32 t .. 32 t + 31 of batch element b (S is a multiple of 32)
    const int m0 = row_tile * GB_ROWS + wave * 32;
    if (m0 >= a.M) return;
    const int b = m0 / a.S, t = (m0 % a.S) >> 5, nt = a.S >> 5;
    const int head0 = (chunk * GB_COLS - a.n_q - (kind == 2 ? 32 * a.H : 0)) >> 5;
    const size_t head_stride = (size_t)nt * 8192;
    char* slot0 = a.kv_blob + (((size_t)b * a.H + head0) * nt + t) * 8192;
    if (kind == 1) epilogue_keys(acc, slot0, head_stride, lane);
    else epilogue_values(acc, slot0, head_stride, lane);
    return;
  }
  if (a.fast_epi) {
    __builtin_amdgcn_s_barrier();  // every wavefront is done with the ring: it becomes the transposition buffer
    epilogue_coalesced(a, acc, ring + wave * 2048, lane, row_tile * GB_ROWS + wave * 32, chunk * GB_COLS);
  } else {
    epilogue(a, acc, m, chunk * GB_COLS + 4 * hi);
  }
}

// Small-grid form (round 5: the reference's operating point is ONE query per step -- 4800 rows are 38 row tiles on 256 CUs).  A launch
// whose workgroups all run at once is bound by ONE workgroup's latency, and the ring loop above exposes a memory round trip every two
// K-steps (17-26 us for a 3 us product at K = 256).  Here the workgroup's requests are issued up front -- the weight slots by LDS DMA
// into NKS x 8 KiB of LDS, the row pieces into registers (8 NKS VGPRs); the hardware's request counter holds 63, so the last K-steps'
// requests follow as soon as the bias (oldest) and K-step 0 have retired -- and the K-steps follow the data with counted waits
// (requests retire in issue order).  The operands of K-step ks + 1 are read from LDS behind the MFMAs of K-step ks (one wavefront per
// SIMD: nobody else hides that latency).  Same products in the same order as the ring kernel, the bias as the accumulators' starting
// value there and here: results are bit-identical, whichever form a launch takes.
template <int N>
__device__ __forceinline__ void gemm_wait_vm() {
  // (the BUILTIN, not inline asm: the compiler's own request counting -- it guards the 6-bit counter against overflow -- sees it)
#ifdef NM_SAFE_WAIT
  __builtin_amdgcn_s_waitcnt(0x0F70);
#else
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);  // vmcnt(N); expcnt / lgkmcnt: no wait
#endif
}

template <int I, int N, class F>
__device__ __forceinline__ void gemm_static_for(F& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    gemm_static_for<I + 1, N>(f);
  }
}

struct GemmOps {
  u32x4 h[4], l[4];
};

template <int FUSED, int NKS>
__global__ void __launch_bounds__(256, 1) gemm_bf16x3_small_kernel(GemmBArgs a) {
  // requests: ONE bias float per thread (it travels through LDS: 16 pieces per lane in the accumulator layout would take 16 of the 63
  // request slots), then 4 per K-step; UP = K-steps requested up front (1 + 4 UP <= 61), the rest follows K-step 0's retirement
  constexpr int UP = NKS < 15 ? NKS : 15;
  __shared__ __attribute__((aligned(16))) float slots_lds[NKS * GB_SLOT_FLOATS + GB_COLS];
  float* const sm_bias = slots_lds + NKS * GB_SLOT_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int chunks = FUSED ? a.nchunks : (a.N + GB_COLS - 1) / GB_COLS;
  const int g = blockIdx.x >> 3, chunk = (FUSED ? a.chunk0 : 0) + g % chunks, row_tile = 8 * (g / chunks) + (blockIdx.x & 7);
  if (row_tile * GB_ROWS >= a.M) return;
  const int m = row_tile * GB_ROWS + wave * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;
  const char* slots = a.blob + (size_t)chunk * NKS * GB_SLOT_BYTES;
  // FUSED == 3 (small-grid form only): the q, key and value chunks of a fused projection in ONE launch -- the transposed product of the
  // value chunks is a workgroup-uniform branch here, and at one query per step a launch costs more than it computes
  const int kind = FUSED == 0 ? 0 : FUSED == 2 ? 2 : (chunk * GB_COLS < a.n_q ? 0 : (FUSED == 3 && chunk * GB_COLS >= a.n_q + 32 * a.H) ? 2 : 1);
  const float* xp = a.x + (size_t)mc * a.K + 8 * hi;
  float bias_v = 0.f;
  {
    const int n = chunk * GB_COLS + (tid & (GB_COLS - 1));
    if (a.fast_epi && a.bias && n < a.N) bias_v = a.bias[n];  // (threads 128..255 load the same values again: one request either way)
  }
  XRow xr[NKS];
  auto request = [&](auto KS) {  // 2 DMA pieces, 2 row pieces (K = 16 NKS exactly: no tail)
    constexpr int ks = decltype(KS)::value;
    const unsigned voff = (unsigned)(wave * 2048 + lane * 16);
    const auto* src = (const __attribute__((address_space(1))) void*)(slots + (size_t)ks * GB_SLOT_BYTES + voff);
    auto* dst = (__attribute__((address_space(3))) void*)(slots_lds + ks * GB_SLOT_FLOATS + wave * 512);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
    xr[ks].a = *reinterpret_cast<const f32x4*>(xp + 16 * ks);
    xr[ks].b = *reinterpret_cast<const f32x4*>(xp + 16 * ks + 4);
  };
  gemm_static_for<0, UP>(request);
  __builtin_amdgcn_sched_barrier(0);
  gemm_wait_vm<4 * UP>();  // the bias (oldest request) has retired
  if (tid < GB_COLS) sm_bias[tid] = bias_v;
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the bias is in LDS before this wavefront passes the barrier below
  auto read_ops = [&](GemmOps& o, int ks) {
    const u32x4* s4 = reinterpret_cast<const u32x4*>(slots_lds + ks * GB_SLOT_FLOATS) + lane;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      o.h[ob] = s4[(ob * 2 + 0) * 64];
      o.l[ob] = s4[(ob * 2 + 1) * 64];
    }
  };
  GemmOps cur, nxt;
  gemm_wait_vm<4 * (UP - 1)>();  // K-step 0
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  gemm_static_for<UP, NKS>(request);
  __builtin_amdgcn_sched_barrier(0);
  f32x16 acc[4];
#pragma unroll
  for (int ob = 0; ob < 4; ++ob)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sm_bias + 32 * ob + 8 * q + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ob][4 * q + e] = b4[e];
    }
  read_ops(cur, 0);
  __builtin_amdgcn_sched_barrier(0);
  auto kstep = [&](auto KS) {
    constexpr int ks = decltype(KS)::value;
    if constexpr (ks + 1 < NKS) {
      gemm_wait_vm<4 * (NKS - 2 - ks)>();  // K-step ks + 1 has landed ...
      __builtin_amdgcn_s_barrier();        // ... everybody's pieces of it
      read_ops(nxt, ks + 1);               // its operands travel LDS -> registers behind this K-step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16x8 xh, xl;
    {
      const float v8[8] = {xr[ks].a[0], xr[ks].a[1], xr[ks].a[2], xr[ks].a[3], xr[ks].b[0], xr[ks].b[1], xr[ks].b[2], xr[ks].b[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v8[i];
        xh[i] = h;
        xl[i] = (__bf16)(v8[i] - (float)h);
      }
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const bf16x8 wh = __builtin_bit_cast(bf16x8, cur.h[ob]), wl = __builtin_bit_cast(bf16x8, cur.l[ob]);
      if (kind == 2) {  // (workgroup-uniform) transposed product
        acc[ob] = MFMA_BF16(xh, wh, acc[ob]);
        acc[ob] = MFMA_BF16(xl, wh, acc[ob]);
        acc[ob] = MFMA_BF16(xh, wl, acc[ob]);
      } else {
        acc[ob] = MFMA_BF16(wh, xh, acc[ob]);
        acc[ob] = MFMA_BF16(wh, xl, acc[ob]);
        acc[ob] = MFMA_BF16(wl, xh, acc[ob]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ks + 1 < NKS) cur = nxt;
  };
  gemm_static_for<0, NKS>(kstep);

  if (kind) {
    const int m0 = row_tile * GB_ROWS + wave * 32;
    if (m0 >= a.M) return;
    const int b = m0 / a.S, t = (m0 % a.S) >> 5, nt = a.S >> 5;
    const int head0 = (chunk * GB_COLS - a.n_q - (kind == 2 ? 32 * a.H : 0)) >> 5;
    const size_t head_stride = (size_t)nt * 8192;
    char* slot0 = a.kv_blob + (((size_t)b * a.H + head0) * nt + t) * 8192;
    if (kind == 1) epilogue_keys(acc, slot0, head_stride, lane);
    else epilogue_values(acc, slot0, head_stride, lane);
    return;
  }
  if (a.fast_epi) {
    __builtin_amdgcn_s_barrier();  // every wavefront is done with the slots: their LDS becomes the transposition buffer
    epilogue_coalesced(a, acc, slots_lds + wave * 2048, lane, row_tile * GB_ROWS + wave * 32, chunk * GB_COLS);
  } else {
    epilogue(a, acc, m, chunk * GB_COLS + 4 * hi);
  }
}

// The small-grid form is taken when every workgroup of the launch is resident at once with one workgroup per CU (<= CU count) or the
// launch is at most two such rounds, and K is exactly 128 or 256 (the matcher's widths).  NM_GEMM_SMALL=0 disables it (A/B runs).
template <int FUSED>
bool launch_small(const GemmBArgs& a, unsigned grid, hipStream_t s) {
  static const bool off = getenv("NM_GEMM_SMALL") && atoi(getenv("NM_GEMM_SMALL")) == 0;
  if (off || grid > 2u * (unsigned)nm_cu_count()) return false;
  if (a.K == 256) gemm_bf16x3_small_kernel<FUSED, 16><<<grid, 256, 0, s>>>(a);
  else if (a.K == 128) gemm_bf16x3_small_kernel<FUSED, 8><<<grid, 256, 0, s>>>(a);
  else return false;
  return true;
}

// NM_GEMM_COALESCED=0 forces the register-layout epilogue (A/B runs)
int gemm_fast_epilogue(const GemmBArgs& a) {
  static const bool off = getenv("NM_GEMM_COALESCED") && atoi(getenv("NM_GEMM_COALESCED")) == 0;
  return off ? 0 : 1;
}

// 1-D grid of the kernel above: row tiles padded to a multiple of 8, times the column chunks
unsigned gemm_grid(int M, int N) {
  const int row_tiles = (M + GB_ROWS - 1) / GB_ROWS, chunks = (N + GB_COLS - 1) / GB_COLS;
  return (unsigned)(((row_tiles + 7) / 8) * 8 * chunks);
}

// blob element (chunk, ks, ob, hl, lane, i) = split(w[128 chunk + 32 ob + (lane & 31)][16 ks + 8 (lane >> 5) + i])
// (ldw, trans): element (n, k) of the packed matrix is w[n * ldw + k], or -- trans: the packed matrix is the TRANSPOSE of the stored one, the
// weight of a layer's input-gradient product dy . W, round 6 -- w[k * ldw + n]
__global__ void linear_pack_bf16x3_kernel(const float* __restrict__ w, int N, int K, int nks, unsigned short* __restrict__ blob,
                                          size_t total, int ldw = 0, int trans = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (hi, lo) pair per thread
  if (idx >= total) return;
  const int i = idx & 7, lane = (idx >> 3) & 63, ob = (idx >> 9) & 3;
  const size_t slot = idx >> 11;
  const int ks = slot % nks, chunk = slot / nks;
  const int n = 128 * chunk + 32 * ob + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + i;
  const float v = (n < N && k < K) ? (trans ? w[(size_t)k * ldw + n] : w[(size_t)n * (ldw ? ldw : K) + k]) : 0.f;
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  unsigned short* s = blob + slot * (GB_SLOT_BYTES / 2);
  s[((ob * 2 + 0) * 64 + lane) * 8 + i] = __builtin_bit_cast(unsigned short, h);
  s[((ob * 2 + 1) * 64 + lane) * 8 + i] = __builtin_bit_cast(unsigned short, l);
}

}  // namespace

extern "C" size_t nm_linear_blob_bytes_bf16x3(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (size_t)((N + GB_COLS - 1) / GB_COLS) * ((K + 15) / 16) * GB_SLOT_BYTES;
}

extern "C" int nm_linear_pack_bf16x3(const float* w, int N, int K, void* blob, nmStream_t stream) {
  NM_CHECK_ARG(w && blob && N > 0 && K > 0);
  const int nks = (K + 15) / 16;
  const size_t total = (size_t)((N + GB_COLS - 1) / GB_COLS) * nks * (GB_SLOT_BYTES / 4);
  linear_pack_bf16x3_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, N, K, nks, (unsigned short*)blob, total);
  return nm_launch_status();
}

extern "C" int nm_linear_pack_t_bf16x3(const float* w, int N, int K, void* blob, nmStream_t stream) {
  // blob of the (N, K) matrix W^T for a stored w of shape (K, N): what nm_linear_bf16x3 needs to compute dy . W (a linear layer's input gradient)
  NM_CHECK_ARG(w && blob && N > 0 && K > 0);
  const int nks = (K + 15) / 16;
  const size_t total = (size_t)((N + GB_COLS - 1) / GB_COLS) * nks * (GB_SLOT_BYTES / 4);
  linear_pack_bf16x3_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, N, K, nks, (unsigned short*)blob, total, N, 1);
  return nm_launch_status();
}

// internal (used by match.hip): sim[M,N] = mask_fill(scale * im[M,C] . pt[N,C]^T) on the split-bf16 path; pt is packed
// into `blob` (nm_linear_blob_bytes_bf16x3(N, C) bytes of workspace) first.  N % 8 == 0, C % 8 == 0.
int nm_internal_sim_bf16x3(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                           const uint8_t* pt_mask, float* sim, void* blob, hipStream_t s) {
  if (N % 8 != 0 || C % 8 != 0) return NM_ERR_UNSUPPORTED;
  int rc = nm_linear_pack_bf16x3(pt, N, C, blob, (nmStream_t)s);
  if (rc != NM_OK) return rc;
  GemmBArgs a{};
  a.x = im; a.blob = (const char*)blob; a.y = sim;
  a.M = M; a.N = N; a.K = C; a.act = NM_ACT_NONE; a.nks = (C + 15) / 16;
  a.sim = 1; a.scale = scale; a.row_mask = im_mask; a.col_mask = pt_mask;
  a.fast_epi = gemm_fast_epilogue(a);
  if (!launch_small<0>(a, gemm_grid(M, N), s)) gemm_bf16x3_kernel<0><<<gemm_grid(M, N), 256, 0, s>>>(a);
  return nm_launch_status();
}

extern "C" int nm_linear_qkv_bf16x3(const float* x, const void* blob, int M, int K, int n_q, int heads, int S, float* q_out, void* kv_slots,
                                    nmStream_t stream) {
  NM_CHECK_ARG(x && blob && kv_slots && M > 0 && K > 0 && n_q >= 0 && heads > 0 && S > 0 && (q_out || n_q == 0));
  if (K % 8 != 0 || n_q % GB_COLS != 0 || (32 * heads) % GB_COLS != 0 || S % 32 != 0 || M % S != 0) return NM_ERR_UNSUPPORTED;
  GemmBArgs a{};
  a.x = x; a.blob = (const char*)blob; a.y = q_out;
  a.M = M; a.N = n_q + 64 * heads; a.K = K; a.act = NM_ACT_NONE; a.nks = (K + 15) / 16;
  a.fast_epi = 1;
  a.kv_blob = (char*)kv_slots; a.n_q = n_q; a.S = S; a.H = heads;
  // q and key chunks, then the value chunks (transposed product: a different instruction stream, hence a second launch)
  const int cq = n_q / GB_COLS, ch = 32 * heads / GB_COLS;
  a.chunk0 = 0; a.nchunks = cq + 2 * ch;
  if (launch_small<3>(a, gemm_grid(M, a.nchunks * GB_COLS), (hipStream_t)stream)) return nm_launch_status();
  a.chunk0 = 0; a.nchunks = cq + ch;
  if (!launch_small<1>(a, gemm_grid(M, a.nchunks * GB_COLS), (hipStream_t)stream))
    gemm_bf16x3_kernel<1><<<gemm_grid(M, a.nchunks * GB_COLS), 256, 0, (hipStream_t)stream>>>(a);
  a.chunk0 = cq + ch; a.nchunks = ch;
  if (!launch_small<2>(a, gemm_grid(M, a.nchunks * GB_COLS), (hipStream_t)stream))
    gemm_bf16x3_kernel<2><<<gemm_grid(M, a.nchunks * GB_COLS), 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}

extern "C" int nm_linear_bf16x3(const float* x, const void* blob, const float* bias, const float* residual, int M, int N, int K,
                                int act, float* y, nmStream_t stream) {
  return nm_linear_ex_bf16x3(x, blob, bias, nullptr, residual, nullptr, M, N, K, act, y, stream);
}

extern "C" int nm_linear_ex_bf16x3(const float* x, const void* blob, const float* bias, const float* pre, const float* residual,
                                   const float* gate, int M, int N, int K, int act, float* y, nmStream_t stream) {
  NM_CHECK_ARG(x && blob && y && M > 0 && N > 0 && K > 0);
  if (act < NM_ACT_NONE || act > NM_ACT_GELU) return NM_ERR_ARG;
  if (K % 8 != 0 || N % 8 != 0) return NM_ERR_UNSUPPORTED;  // 16-byte row pieces on both sides
  GemmBArgs a{};
  a.x = x; a.blob = (const char*)blob; a.bias = bias; a.res = residual; a.pre = pre; a.gate = gate; a.y = y;
  a.M = M; a.N = N; a.K = K; a.act = act; a.nks = (K + 15) / 16;
  a.fast_epi = gemm_fast_epilogue(a);
  if (!launch_small<0>(a, gemm_grid(M, N), (hipStream_t)stream)) gemm_bf16x3_kernel<0><<<gemm_grid(M, N), 256, 0, (hipStream_t)stream>>>(a);
  return nm_launch_status();
}
