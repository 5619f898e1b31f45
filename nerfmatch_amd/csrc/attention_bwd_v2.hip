// Backward of softmax attention (head dim 32) on the bf16 matrix cores with hi/lo operand splitting -- the training-side
// counterpart of attention_v2.hip, same contractions as attention_bwd.hip (fp32 MFMA) at 3/16 of their matrix time:
//     S = scale Q K^T,  P = exp(S - lse),  dP = dO V^T,  dS = P o (dP - D),  dQ = scale dS K,  dK = scale dS^T Q,  dV = P^T dO
// Every operand that is an MFMA "A" matrix is split into bf16 hi/lo ONCE per call (bwd_presplit_kernel) and laid out in
// MFMA-operand order, one slot per 32-row tile, streamed through a 3-slot LDS ring with global_load_lds two tiles ahead:
//   key-tile slot (12 KiB)   : K rows, V rows, K^T (key-permuted)                    -> attn32_bwd_dq_v2_kernel (lane = query)
//   query-tile slot (17 KiB) : Q rows, dO rows, Q^T, dO^T (query-permuted), -lse/-D  -> attn32_bwd_dkv_v2_kernel (lane = key)
// "rows" = A operand of a (rows x dims) . (dims x lanes) product; "^T permuted" = A operand of a (dims x rows) . (rows x
// lanes) product whose B operand is what the first product left in the accumulator registers (k-slot i of step ks <->
// row (i&3) + 16 ks + 8 (i>>2) + 4 half, cf. attention_v2.hip).  The soft-max shift and the D term enter as the C operand
// of the first MFMA (-lse, -D), so the scores come out ready for exp2 and dP - D needs no subtraction.
// The dq kernel first rebuilds the log-sum-exp of its queries (scores only, lazy running maximum) and publishes -lse, -D.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int KSLOT_BYTES = 12288, KSLOT_FLOATS = KSLOT_BYTES / 4;  // 12 operand pieces of 1 KiB
constexpr int QSLOT_BYTES = 17408, QSLOT_FLOATS = QSLOT_BYTES / 4;  // 16 operand pieces + 1 KiB of per-query scalars
constexpr int RING = 3;
constexpr float LOG2E = 1.44269504088896340736f;
constexpr float RAISE = 8.0f;

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// grid (tiles, H, B), block 256.  Piece p of a slot (64 lanes x 16 bytes): group g = p / 4 in {x rows, y rows, x^T, y^T},
// k-step ks = (p / 2) & 1, hi / lo = p & 1.  groups: 3 (key tiles: x = K, y = V) or 4 (query tiles: x = Q, y = dO, plus the
// scalar piece: floats [0,32) = -lse, [32,64) = -D of the tile's queries; -lse = -inf past the end => probability 0).
__global__ void __launch_bounds__(256) bwd_presplit_kernel(const float* __restrict__ x, const float* __restrict__ y, int ldx, int ldy,
                                                            int n, int H, int groups, int slot_bytes, const float* __restrict__ nlse,
                                                            const float* __restrict__ nd, char* __restrict__ blob) {
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nt = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, half = lane >> 5;
  char* slot = blob + (((size_t)b * H + h) * nt + t) * slot_bytes;
  for (int job = tid >> 6; job < 2 * groups; job += 4) {  // job = (group, k-step)
    const int g = job >> 1, ks = job & 1;
    const float* src = (g & 1) ? y : x;
    const int ld = (g & 1) ? ldy : ldx;
    float v8[8];
    if (g < 2) {  // rows: row = r, k-slots = dims 16 ks + 8 half + i
      const int row = t * 32 + r;
      if (row < n) {
        const float* p = src + ((size_t)b * n + row) * ld + h * 32 + 16 * ks + 8 * half;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), c = *reinterpret_cast<const f32x4*>(p + 4);
        v8[0] = a[0]; v8[1] = a[1]; v8[2] = a[2]; v8[3] = a[3]; v8[4] = c[0]; v8[5] = c[1]; v8[6] = c[2]; v8[7] = c[3];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = 0.f;
      }
    } else {  // transposed: row = dim r, k-slot i of step ks <-> tile row (i&3) + 16 ks + 8 (i>>2) + 4 half
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = t * 32 + (i & 3) + 16 * ks + 8 * (i >> 2) + 4 * half;
        v8[i] = row < n ? src[((size_t)b * n + row) * ld + h * 32 + r] : 0.f;
      }
    }
    bf16x8 hi8, lo8;
    split8(v8, hi8, lo8);
    u32x4* o = reinterpret_cast<u32x4*>(slot) + (size_t)(4 * g + 2 * ks) * 64 + lane;
    o[0] = __builtin_bit_cast(u32x4, hi8);
    o[64] = __builtin_bit_cast(u32x4, lo8);
  }
  if (groups == 4 && tid < 64) {
    const int q = t * 32 + (tid & 31);
    const size_t base = ((size_t)b * H + h) * n;
    float v;
    if (tid < 32) v = q < n ? nlse[base + q] : -__builtin_inff();
    else v = q < n ? nd[base + q] : 0.f;
    reinterpret_cast<float*>(slot + 16384)[tid] = v;
  }
}

// `pieces` 1 KiB pieces per wavefront (3 or 4): ONE address / M0 per tile, the pieces differ in the immediate offset
template <int PIECES, int SLOT_BYTES>
__device__ __forceinline__ void dma_tile(const char* slots, int t, float* ring, int wave, int lane) {
  const unsigned voff = (unsigned)(wave * PIECES * 1024 + lane * 16);
  const char* base = slots + (size_t)t * SLOT_BYTES;
  const auto* src = (const __attribute__((address_space(1))) void*)(base + voff);
  auto* dst = (__attribute__((address_space(3))) void*)(ring + (t % RING) * (SLOT_BYTES / 4) + wave * PIECES * 256);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
  if (PIECES == 4) __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
}

struct BwdArgs2 {
  const float *q, *o, *d_o;
  int ldq, ldo, lddo;
  const float *k, *v;
  int ldk, ldv;
  float *dq, *dk, *dv;
  int lddq, lddk, lddv;
  int L, S, H, B;
  float scale;
  const char* blob;  // key-tile slots (dq kernel) or query-tile slots (dkv kernel)
  float *nlse, *nd;  // [B][H][L]: -lse (log2 domain), -D
};

// acc (+)= A(rows from LDS pieces p0 .. p0+3) . B(registers bh/bl): 6 MFMAs
__device__ __forceinline__ f32x16 mma_rows(const u32x4* s4, int p0, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2], f32x16 acc) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, s4[(p0 + 2 * m + 0) * 64]);
    const bf16x8 al = __builtin_bit_cast(bf16x8, s4[(p0 + 2 * m + 1) * 64]);
    acc = MFMA_BF16(ah, bh[m], acc);
    acc = MFMA_BF16(ah, bl[m], acc);
    acc = MFMA_BF16(al, bh[m], acc);
  }
  return acc;
}

__device__ __forceinline__ void load_split(const float* p, float s, bf16x8 (&h)[2], bf16x8 (&l)[2]) {  // dims 16 m + 8 half + i
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(p + 16 * m), b4 = *reinterpret_cast<const f32x4*>(p + 16 * m + 4);
    const float v8[8] = {a4[0] * s, a4[1] * s, a4[2] * s, a4[3] * s, b4[0] * s, b4[1] * s, b4[2] * s, b4[3] * s};
    split8(v8, h[m], l[m]);
  }
}

__device__ __forceinline__ void split16(const f32x16& v, bf16x8 (&h)[2], bf16x8 (&l)[2]) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const float v8[8] = {v[8 * m], v[8 * m + 1], v[8 * m + 2], v[8 * m + 3], v[8 * m + 4], v[8 * m + 5], v[8 * m + 6], v[8 * m + 7]};
    split8(v8, h[m], l[m]);
  }
}

__device__ __forceinline__ void store16(float* p, const f32x16& a, float s) {  // register 4g+e <-> dim 8g + 4hi + e (p offset by 4hi)
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(p + 8 * g) = f32x4{a[4 * g] * s, a[4 * g + 1] * s, a[4 * g + 2] * s, a[4 * g + 3] * s};
}

// XCD-aware 1-D grid as in attention_v2.hip: id -> (xcd = id % 8, row block = (id / 8) % nblk, (batch, head) = 8 (id / (8 nblk)) + xcd)
__device__ __forceinline__ bool map_work(int nblk, int BH, int& bh, int& blk) {
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  bh = 8 * (jj / nblk) + xcd;
  blk = jj % nblk;
  return bh < BH;
}

template <bool HAVE_LSE>
__global__ void __launch_bounds__(256, 3) attn32_bwd_dq_v2_kernel(BwdArgs2 a) {
  __shared__ __attribute__((aligned(16))) float ring[RING * KSLOT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int nqb = ((a.L + 31) / 32 + 3) / 4;
  int bh, blk;
  if (!map_work(nqb, a.B * a.H, bh, blk)) return;
  const int h = bh % a.H, b = bh / a.H;
  const int qrow = (blk * 4 + wave) * 32 + j;
  const int qc = qrow < a.L ? qrow : a.L - 1;
  const int nt = (a.S + 31) / 32;
  const char* slots = a.blob + ((size_t)b * a.H + h) * nt * KSLOT_BYTES;
  dma_tile<3, KSLOT_BYTES>(slots, 0, ring, wave, lane);
  if (nt > 1) dma_tile<3, KSLOT_BYTES>(slots, 1, ring, wave, lane);
  bf16x8 qh[2], ql[2], doh[2], dol[2];
  load_split(a.q + ((size_t)b * a.L + qc) * a.ldq + h * 32 + 8 * hi, a.scale * LOG2E, qh, ql);
  load_split(a.d_o + ((size_t)b * a.L + qc) * a.lddo + h * 32 + 8 * hi, 1.0f, doh, dol);
  float dsum = 0.f;
  {
    const float* op = a.o + ((size_t)b * a.L + qc) * a.ldo + h * 32 + 8 * hi;
    const float* dp = a.d_o + ((size_t)b * a.L + qc) * a.lddo + h * 32 + 8 * hi;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 o4 = *reinterpret_cast<const f32x4*>(op + 16 * m + 4 * c), d4 = *reinterpret_cast<const f32x4*>(dp + 16 * m + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) dsum = NM_FMA(o4[e], d4[e], dsum);
      }
    dsum += nm_shfl_xor32(dsum);
  }
  auto acquire = [&](int t, int ntiles) {
    // tile t has landed when at most the 3 DMA instructions of tile t+1 remain in flight (q / o / dO loads are older)
    if (t + 1 < ntiles) NM_WAIT_VMCNT(3);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody's pieces of tile t landed; nobody reads tile t-1 any more
    if (t + 2 < ntiles) dma_tile<3, KSLOT_BYTES>(slots, t + 2, ring, wave, lane);
  };
  auto mask_tail = [&](int t, f32x16& sc) {
    if (t == nt - 1 && (a.S & 31)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= a.S) sc[r] = -__builtin_inff();
    }
  };
  // ---- pass 1: log-sum-exp of each query (log2 domain), lazy running maximum as in the forward kernel -- unless the forward pass
  // kept it (HAVE_LSE, round 6: nm_attention_ws_lse), in which case the kernel is ONE pass over the keys: 3 products instead of 4
  float nlse;
  if constexpr (HAVE_LSE) {
    nlse = a.nlse[((size_t)b * a.H + h) * a.L + qc];
    if (qrow < a.L && hi == 0) a.nd[((size_t)b * a.H + h) * a.L + qrow] = -dsum;
  } else {
  f32x16 negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  float lrun = 0.f;
  bool first = true;
  for (int t = 0; t < nt; ++t) {
    acquire(t, nt);
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ring + (t % RING) * KSLOT_FLOATS) + lane;
    f32x16 sc = mma_rows(s4, 0, qh, ql, negm);
    mask_tail(t, sc);
    float mx = fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3]));
#pragma unroll
    for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(sc[r], sc[r + 1]), fmaxf(sc[r + 2], sc[r + 3])));
    mx = fmaxf(mx, nm_shfl_xor32(mx));
    const bool raise = first || mx > RAISE;
    if (__builtin_amdgcn_ballot_w64(raise) != 0) {
      const float delta = raise ? mx : 0.f;
      lrun *= __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sc[i] -= delta;
        negm[i] -= delta;
      }
    }
    first = false;
    float ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) ps += __builtin_amdgcn_exp2f(sc[i]);
    lrun += ps;
  }
  const float ltot = lrun + nm_shfl_xor32(lrun);
  nlse = negm[0] - __builtin_amdgcn_logf(ltot);  // -(m + log2 l)
  if (qrow < a.L && hi == 0) {
    a.nlse[((size_t)b * a.H + h) * a.L + qrow] = nlse;
    a.nd[((size_t)b * a.H + h) * a.L + qrow] = -dsum;
  }
  // ---- pass 2: dQ
  __builtin_amdgcn_s_barrier();  // the last tiles of pass 1 are no longer read before the ring is refilled
  dma_tile<3, KSLOT_BYTES>(slots, 0, ring, wave, lane);
  if (nt > 1) dma_tile<3, KSLOT_BYTES>(slots, 1, ring, wave, lane);
  }  // !HAVE_LSE
  f32x16 c_lse, c_d, dq;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c_lse[i] = nlse;
    c_d[i] = -dsum;
    dq[i] = 0.f;
  }
  for (int t = 0; t < nt; ++t) {
    acquire(t, nt);
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ring + (t % RING) * KSLOT_FLOATS) + lane;
    f32x16 sc = mma_rows(s4, 0, qh, ql, c_lse);          // log2 P
    mask_tail(t, sc);
    const f32x16 dp = mma_rows(s4, 4, doh, dol, c_d);    // dP - D
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[i] = __builtin_amdgcn_exp2f(sc[i]) * dp[i];  // dS
    bf16x8 sh[2], sl[2];
    split16(sc, sh, sl);
    dq = mma_rows(s4, 8, sh, sl, dq);                     // dQ^T += K^T . dS
  }
  if (qrow < a.L) store16(a.dq + ((size_t)b * a.L + qrow) * a.lddq + h * 32 + 4 * hi, dq, a.scale);
}

__global__ void __launch_bounds__(256, 3) attn32_bwd_dkv_v2_kernel(BwdArgs2 a) {
  __shared__ __attribute__((aligned(16))) float ring[RING * QSLOT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int nkb = ((a.S + 31) / 32 + 3) / 4;
  int bh, blk;
  if (!map_work(nkb, a.B * a.H, bh, blk)) return;
  const int h = bh % a.H, b = bh / a.H;
  const int krow = (blk * 4 + wave) * 32 + j;
  const int kc = krow < a.S ? krow : a.S - 1;
  const int nt = (a.L + 31) / 32;
  const char* slots = a.blob + ((size_t)b * a.H + h) * nt * QSLOT_BYTES;
  auto dma = [&](int t) {
    dma_tile<4, QSLOT_BYTES>(slots, t, ring, wave, lane);
    if (wave == 0) {  // the scalar piece (only 256 bytes are meaningful; one more 1 KiB instruction of this wavefront)
      const char* base = slots + (size_t)t * QSLOT_BYTES + 16384 + lane * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)base,
                                       (__attribute__((address_space(3))) void*)(ring + (t % RING) * QSLOT_FLOATS + 4096), 16, 0, 0);
    }
  };
  dma(0);
  if (nt > 1) dma(1);
  bf16x8 kh[2], kl[2], vh[2], vl[2];
  load_split(a.k + ((size_t)b * a.S + kc) * a.ldk + h * 32 + 8 * hi, a.scale * LOG2E, kh, kl);
  load_split(a.v + ((size_t)b * a.S + kc) * a.ldv + h * 32 + 8 * hi, 1.0f, vh, vl);
  f32x16 dk, dv;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    dk[i] = 0.f;
    dv[i] = 0.f;
  }
  for (int t = 0; t < nt; ++t) {
    // wavefront 0 issues 5 DMA instructions per tile, the others 4
    if (t + 1 < nt) {
      if (wave == 0) NM_WAIT_VMCNT(5);
      else NM_WAIT_VMCNT(4);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) dma(t + 2);
    const float* slot = ring + (t % RING) * QSLOT_FLOATS;
    const u32x4* s4 = reinterpret_cast<const u32x4*>(slot) + lane;
    f32x16 c_lse, c_d;  // register 4c+e <-> query 8c + 4hi + e of the tile
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(slot + 4096 + 8 * c + 4 * hi);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(slot + 4096 + 32 + 8 * c + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        c_lse[4 * c + e] = l4[e];
        c_d[4 * c + e] = d4[e];
      }
    }
    f32x16 p = mma_rows(s4, 0, kh, kl, c_lse);          // log2 P: lane = key, register <-> query
    const f32x16 dp = mma_rows(s4, 4, vh, vl, c_d);     // dP - D
    f32x16 ds;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      p[i] = __builtin_amdgcn_exp2f(p[i]);
      ds[i] = p[i] * dp[i];
    }
    bf16x8 ph[2], pl[2], sh[2], sl[2];
    split16(p, ph, pl);
    split16(ds, sh, sl);
    dv = mma_rows(s4, 12, ph, pl, dv);                   // dV^T += dO^T . P
    dk = mma_rows(s4, 8, sh, sl, dk);                    // dK^T += Q^T . dS
  }
  if (krow < a.S) {
    store16(a.dk + ((size_t)b * a.S + krow) * a.lddk + h * 32 + 4 * hi, dk, a.scale);
    store16(a.dv + ((size_t)b * a.S + krow) * a.lddv + h * 32 + 4 * hi, dv, 1.0f);
  }
}

}  // namespace

size_t nm_internal_attn_bwd_v2_workspace(int B, int L, int S, int heads) {
  const size_t kt = (size_t)B * heads * ((S + 31) / 32) * KSLOT_BYTES, qt = (size_t)B * heads * ((L + 31) / 32) * QSLOT_BYTES;
  return kt + qt + (size_t)2 * B * heads * L * sizeof(float) + 512;
}

int nm_internal_attn_bwd_v2(const float* q, const float* k, const float* v, const float* o, const float* d_o, int ldq, int ldk, int ldv,
                            int ldo, int lddo, int B, int L, int S, int heads, float scale, float* dq, float* dk, float* dv, int lddq,
                            int lddk, int lddv, void* workspace, hipStream_t s, const float* nlse_fwd) {
  const int ntk = (S + 31) / 32, ntq = (L + 31) / 32;
  char* kblob = (char*)workspace;
  char* qblob = kblob + (size_t)B * heads * ntk * KSLOT_BYTES;
  float* nlse = (float*)(((uintptr_t)(qblob + (size_t)B * heads * ntq * QSLOT_BYTES) + 255) & ~(uintptr_t)255);
  float* nd = nlse + (size_t)B * heads * L;
  const long long gq = (long long)((B * heads + 7) / 8) * 8 * ((ntq + 3) / 4), gk = (long long)((B * heads + 7) / 8) * 8 * ((ntk + 3) / 4);
  if (gq > 0x7fffffffLL || gk > 0x7fffffffLL) return NM_ERR_UNSUPPORTED;
  if (nlse_fwd) nlse = const_cast<float*>(nlse_fwd);  // (read only on this path: the dq kernel loads it, the query-tile pre-split copies it)
  BwdArgs2 a{q, o, d_o, ldq, ldo, lddo, k, v, ldk, ldv, dq, dk, dv, lddq, lddk, lddv, L, S, heads, B, scale, kblob, nlse, nd};
  bwd_presplit_kernel<<<dim3(ntk, heads, B), 256, 0, s>>>(k, v, ldk, ldv, S, heads, 3, KSLOT_BYTES, nullptr, nullptr, kblob);
  if (nlse_fwd) attn32_bwd_dq_v2_kernel<true><<<(unsigned)gq, 256, 0, s>>>(a);
  else attn32_bwd_dq_v2_kernel<false><<<(unsigned)gq, 256, 0, s>>>(a);
  bwd_presplit_kernel<<<dim3(ntq, heads, B), 256, 0, s>>>(q, d_o, ldq, lddo, L, heads, 4, QSLOT_BYTES, nlse, nd, qblob);
  a.blob = qblob;
  attn32_bwd_dkv_v2_kernel<<<(unsigned)gk, 256, 0, s>>>(a);
  return nm_launch_status();
}
