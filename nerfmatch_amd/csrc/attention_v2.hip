// Flash-style softmax attention for head_dim 32 on the bf16 matrix cores (hi/lo operand splitting), second generation.
//
// attn32_bf16x3_kernel (attention.hip) re-splits every K/V tile in every workgroup (38 times for 4800 queries) and
// spends more VALU cycles on staging and on the online-softmax bookkeeping than the matrix cores need for the two
// contractions.  Here
//   * K and V are split ONCE per call (kv_presplit_kernel) into ready-made MFMA A operands: per (batch, head, 32-key tile)
//     one 8 KiB slot = {K, V^T} x {dims/keys 0-15, 16-31} x {hi, lo} x 64 lanes x 8 bf16, V^T key-permuted so that a
//     lane's 8 consecutive bf16 are exactly the 8 K-slots the probabilities occupy after the first MFMA;
//   * the slots are streamed into a 4-slot LDS ring with global_load_lds two tiles ahead (no VGPRs, no VALU), one
//     counted s_waitcnt + one s_barrier per tile;
//   * "lazy" running maximum: the scores come out of the MFMA already shifted (C operand = -m), m is only raised when a
//     score exceeds it by more than 2^8 (wave-uniform branch), so the common tile costs max + exp2 + sum + split only.
// Arithmetic and accuracy are those of the first generation (w_hi*x_hi + w_hi*x_lo + w_lo*x_hi, fp32 accumulate).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int AT_SLOT_BYTES = 8192;
constexpr int AT_SLOT_FLOATS = AT_SLOT_BYTES / 4;
constexpr int AT_RING = 4;
constexpr float AT_RAISE = 8.0f;  // log2 units: probabilities stay below 2^8 between two raises of the running maximum

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// grid (tiles, H, B), block 256: thread = (which in {K, V}, k-step, lane) -> one hi and one lo operand of 16 bytes
__global__ void __launch_bounds__(256) kv_presplit_kernel(const float* __restrict__ k, const float* __restrict__ v, int ldk, int ldv,
                                                           int S, int H, char* __restrict__ blob) {
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nt = gridDim.x;
  const int tid = threadIdx.x, which = tid >> 7, ks = (tid >> 6) & 1, lane = tid & 63, r = lane & 31, half = lane >> 5;
  float v8[8];
  if (which == 0) {
    // K tile as A operand of S^T = K . Q^T: row = key r, k-slots = dims 16 ks + 8 half + i
    const int key = t * 32 + r;
    if (key < S) {
      const float* p = k + ((size_t)b * S + key) * ldk + h * 32 + 16 * ks + 8 * half;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), c = *reinterpret_cast<const f32x4*>(p + 4);
      v8[0] = a[0]; v8[1] = a[1]; v8[2] = a[2]; v8[3] = a[3]; v8[4] = c[0]; v8[5] = c[1]; v8[6] = c[2]; v8[7] = c[3];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v8[i] = 0.f;
    }
  } else {
    // V^T tile as A operand of O^T += V^T . P^T: row = dim r, k-slot i of step ks <-> key (i&3) + 16 ks + 8 (i>>2) + 4 half
    // (= the key held by accumulator register 8 ks + i of a lane in half `half` after the first MFMA)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int key = t * 32 + (i & 3) + 16 * ks + 8 * (i >> 2) + 4 * half;
      v8[i] = key < S ? v[((size_t)b * S + key) * ldv + h * 32 + r] : 0.f;
    }
  }
  bf16x8 hi8, lo8;
  split8(v8, hi8, lo8);
  u32x4* slot = reinterpret_cast<u32x4*>(blob + (((size_t)b * H + h) * nt + t) * AT_SLOT_BYTES);
  const int op = which * 4 + ks * 2;
  slot[(op + 0) * 64 + lane] = __builtin_bit_cast(u32x4, hi8);
  slot[(op + 1) * 64 + lane] = __builtin_bit_cast(u32x4, lo8);
}

// two 1 KiB pieces per wavefront: one address / one M0, told apart by the immediate offset
__device__ __forceinline__ void dma_tile(const char* slots, int t, float* ring, int wave, int lane) {
  const unsigned voff = (unsigned)(wave * 2048 + lane * 16);
  const char* base = slots + (size_t)t * AT_SLOT_BYTES;
  const auto* src = (const __attribute__((address_space(1))) void*)(base + voff);
  auto* dst = (__attribute__((address_space(3))) void*)(ring + (t & (AT_RING - 1)) * AT_SLOT_FLOATS + wave * 512);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
}

// 1-D grid of ceil(B*H / 8) * 8 * ceil(L/128) workgroups, block 256 = 4 wavefronts x 32 queries.
// XCD-aware work mapping: consecutive workgroup ids go round robin to the 8 XCDs (each with its own 4 MiB L2), so
// id -> (xcd = id % 8, query block = (id / 8) % nqb, (batch, head) = 8 * (id / (8 nqb)) + xcd): all query blocks of one
// (batch, head) run on ONE XCD, whose L2 then holds the 1.2 MB of K/V slots they all stream (with the natural
// blockIdx.{x,y,z} order ~20 different (batch, head) pairs were live per XCD and the slots came back from the fabric
// nine times over: 358 MB of FETCH_SIZE per 4-query launch).
// Third generation: the same arithmetic, software-pipelined INSIDE the wavefront.  In the second generation a tile was a strict
// chain -- 6 dependent QK^T MFMAs -> ~95 VALU instructions of softmax -> 6 dependent PV MFMAs -- so the matrix pipe only
// worked while some OTHER wavefront of the SIMD happened to be in its VALU phase (measured: one tile per 754 cycles and SIMD
// against 384 MFMA and ~430 VALU cycles).  Here iteration t issues the QK^T MFMAs of tile t+1 before the softmax of tile t,
// so they run underneath that VALU work, and the running maximum is not tracked per tile any more: the scores come out of the
// MFMA shifted by the current maximum as before, but it is only raised when a tile's probabilities get near the fp32 range
// (detected on the row sum the loop needs anyway; the first tile always sets it) -- the per-tile max / swap / compare /
// branch of the second generation is gone from the common path.  Ring: tile t+3 is requested while tile t+1's keys and
// tile t's values are read (4 slots).
constexpr float AT_BIG = 1.2676506e30f;  // 2^100: a tile whose per-lane probability sum reaches this raises the maximum

__global__ void __launch_bounds__(256, 4) attn32_v3_kernel(const float* __restrict__ q, int ldq, const char* __restrict__ blob, int L,
                                                         int S, int H, int B, float scale, float* __restrict__ out, float* __restrict__ nlse_out) {
  __shared__ __attribute__((aligned(16))) float ring[AT_RING * AT_SLOT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int nqb = ((L + 31) / 32 + 3) / 4;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int bh = 8 * (jj / nqb) + xcd;
  if (bh >= B * H) return;  // (whole workgroup: B*H is padded to a multiple of 8)
  const int h = bh % H, b = bh / H;
  const int qt = (jj % nqb) * 4 + wave;
  const int C = H * 32;
  const int qrow = qt * 32 + j;
  const int qc = qrow < L ? qrow : L - 1;
  const int nt = (S + 31) / 32;
  const char* slots = blob + ((size_t)b * H + h) * nt * AT_SLOT_BYTES;
  dma_tile(slots, 0, ring, wave, lane);
  if (nt > 1) dma_tile(slots, 1, ring, wave, lane);
  if (nt > 2) dma_tile(slots, 2, ring, wave, lane);
  bf16x8 qh[2], ql[2];
  {
    const float qs = scale * 1.44269504088896340736f;
    const float* qp = q + ((size_t)b * L + qc) * ldq + h * 32 + 8 * hi;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(qp + 16 * m), b4 = *reinterpret_cast<const f32x4*>(qp + 16 * m + 4);
      const float v8[8] = {a4[0] * qs, a4[1] * qs, a4[2] * qs, a4[3] * qs, b4[0] * qs, b4[1] * qs, b4[2] * qs, b4[3] * qs};
      split8(v8, qh[m], ql[m]);
    }
  }
  f32x16 o, negm, sc;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[i] = 0.f; negm[i] = 0.f; }
  float lrun = 0.f;
  auto scores = [&](int t, const f32x16& c) {
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ring + (t & (AT_RING - 1)) * AT_SLOT_FLOATS) + lane;
    f32x16 r = c;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const bf16x8 kh = __builtin_bit_cast(bf16x8, s4[(2 * m + 0) * 64]);
      const bf16x8 kl = __builtin_bit_cast(bf16x8, s4[(2 * m + 1) * 64]);
      r = MFMA_BF16(kh, qh[m], r);
      r = MFMA_BF16(kh, ql[m], r);
      r = MFMA_BF16(kl, qh[m], r);
    }
    return r;
  };
  // everything requested so far (tiles 0..2, q) has landed; the loop's counted waits start from a clean slate
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  sc = scores(0, negm);
  // one tile: softmax / PV of the scores in `sc`, QK^T of the next tile into `scn`.  Called twice per loop iteration with the
  // two score registers swapping roles, so no tile pays a 16-register copy.
  auto step = [&](int t, f32x16& sc, f32x16& scn) __attribute__((always_inline)) {
    // tile t+1 (its keys are read below) has landed when at most the 2 DMA instructions of tile t+2 remain in flight
    if (t + 2 < nt) NM_WAIT_VMCNT(2);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody's pieces of tile t+1 landed; nobody reads tile t-1 any more
    if (t + 3 < nt) dma_tile(slots, t + 3, ring, wave, lane);
    if (t == nt - 1 && (S & 31)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= S) sc[r] = -__builtin_inff();
    }
    // ONE basic block: the 6 (dependent) QK^T MFMAs of the next tile, each followed by a share of this tile's exp2 / row-sum
    // VALU work -- the in-order front end reaches the next MFMA just as the previous one leaves the pipe.  (Last iteration:
    // the scores of tile nt-1 once more, unused.)
    scn = scores(t + 1 < nt ? t + 1 : t, negm);
    float p[16], ps = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      p[i] = __builtin_amdgcn_exp2f(sc[i]);
      ps += p[i];
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
    }
    if (__builtin_expect(t == 0 || __builtin_amdgcn_ballot_w64(!(ps < AT_BIG)) != 0, 0)) {
      // set (first tile) or raise the running maximum of the queries that need it, rescale their partial results, redo the tile.
      // A REAL branch (round 5): the compiler had if-converted this block -- its 16 extra exponentials, 48 subtractions, the row
      // maximum and the rescaling of o ran on EVERY tile behind selects (32 v_exp_f32 per tile in the loop body instead of 16; the
      // kernel is VALU-port bound).  A volatile asm statement cannot be speculated, so the block stays behind its branch.
      asm volatile("; raise the running maximum (rare path)");
      float mlo, mhi;
      nm_swap32(nm_max16(sc), mlo, mhi);
      const float mx = nm_max3(mlo, mhi, mhi);  // row maximum over the tile's 32 keys (both wavefront halves)
      const float delta = (t == 0 || mx > 64.0f) ? mx : 0.f;
      const float alpha = __builtin_amdgcn_exp2f(-delta);
      lrun *= alpha;
      ps = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        o[i] *= alpha;
        negm[i] -= delta;
        scn[i] -= delta;
        p[i] = __builtin_amdgcn_exp2f(sc[i] - delta);
        ps += p[i];
      }
    }
    lrun += ps;
    bf16x8 ph[2], pl[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float p8[8] = {p[8 * m], p[8 * m + 1], p[8 * m + 2], p[8 * m + 3], p[8 * m + 4], p[8 * m + 5], p[8 * m + 6], p[8 * m + 7]};
      split8(p8, ph[m], pl[m]);
    }
    const u32x4* s4 = reinterpret_cast<const u32x4*>(ring + (t & (AT_RING - 1)) * AT_SLOT_FLOATS) + lane;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const bf16x8 vh = __builtin_bit_cast(bf16x8, s4[(4 + 2 * m + 0) * 64]);
      const bf16x8 vl = __builtin_bit_cast(bf16x8, s4[(4 + 2 * m + 1) * 64]);
      o = MFMA_BF16(vh, ph[m], o);
      o = MFMA_BF16(vh, pl[m], o);
      o = MFMA_BF16(vl, ph[m], o);
    }
  };
  f32x16 sc2;
  for (int t = 0; t < nt; t += 2) {
    step(t, sc, sc2);
    if (t + 1 < nt) step(t + 1, sc2, sc);
  }
  const float ltot = lrun + nm_shfl_xor32(lrun);
  // training forward (round 6): -(log-sum-exp) of the query's scores, log2 domain, [B][H][L] -- what the backward kernels shift their
  // recomputed scores by; they used to rebuild it with a pass of their own over all keys (one of the dq kernel's four products)
  if (nlse_out && qrow < L && hi == 0) nlse_out[((size_t)b * H + h) * L + qrow] = negm[0] - __builtin_amdgcn_logf(ltot);
  if (qrow < L) {
    const float inv = 1.0f / ltot;
    float* op = out + ((size_t)b * L + qrow) * C + h * 32 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w4 = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(op + 8 * g) = w4;
    }
  }
}

void attn32_launch(unsigned grid, hipStream_t s, const float* q, int ldq, const char* blob, int L, int S, int H, int B, float scale, float* out,
                   float* nlse_out = nullptr) {
  attn32_v3_kernel<<<grid, 256, 0, s>>>(q, ldq, blob, L, S, H, B, scale, out, nlse_out);
}

}  // namespace

size_t nm_internal_attn_v2_workspace(int B, int S, int heads) {
  return (size_t)B * heads * ((S + 31) / 32) * AT_SLOT_BYTES;
}

// attention over K / V slots that are already in place (written by nm_linear_qkv_bf16x3)
extern "C" int nm_attention_presplit(const float* q, int ldq, const void* kv_slots, int B, int L, int S, int heads, float scale, float* out,
                                     nmStream_t stream) {
  NM_CHECK_ARG(q && kv_slots && out && B > 0 && L > 0 && S > 0 && heads > 0);
  if (ldq < 32 * heads || ldq % 4 || B > 65535 || heads > 65535) return NM_ERR_ARG;
  const int nqb = ((L + 31) / 32 + 3) / 4;
  const long long grid = (long long)((B * heads + 7) / 8) * 8 * nqb;
  if (grid > 0x7fffffffLL) return NM_ERR_UNSUPPORTED;
  attn32_launch((unsigned)grid, (hipStream_t)stream, q, ldq, (const char*)kv_slots, L, S, heads, B, scale, out);
  return nm_launch_status();
}

int nm_internal_attn_v2(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S, int heads,
                        float scale, void* workspace, float* out, hipStream_t s, float* nlse_out) {
  const int nt = (S + 31) / 32;
  kv_presplit_kernel<<<dim3(nt, heads, B), 256, 0, s>>>(k, v, ldk, ldv, S, heads, (char*)workspace);
  const int nqb = ((L + 31) / 32 + 3) / 4;
  const long long grid = (long long)((B * heads + 7) / 8) * 8 * nqb;
  if (grid > 0x7fffffffLL) return NM_ERR_UNSUPPORTED;
  attn32_launch((unsigned)grid, s, q, ldq, (const char*)workspace, L, S, heads, B, scale, out, nlse_out);
  return nm_launch_status();
}
