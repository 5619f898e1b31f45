// Flash-style softmax attention for head_dim 32 with BOTH contractions on fp8 (OCP e4m3) matrix-core operands -- the
// throughput configuration BASELINE.json config 5 names ("fp8 MFMA attention").  NOT a parity arithmetic: e4m3 carries 3
// mantissa bits; the error against the fp32 result is reported by bench.py (`variants.attention_fp8`) and bounded by
// tests/test_matcher_gpu.py::test_attention_fp8_error_bound, never asserted at 1e-4.
// Reference arithmetic it approximates: FullAttention, nerfmatch/modules/attention.py:44-57 (fp32 einsum, softmax, einsum).
//
// What changes against the split-bf16 kernel (attention_v2.hip, three bf16 MFMAs per product block):
//   * ONE v_mfma_f32_32x32x16_fp8_fp8 per product block (a third of the matrix work; non-scaled fp8 runs at the bf16 rate) and
//     no hi/lo splitting of the probabilities (the VALU work that bounds the bf16x3 kernel): 8 v_cvt_pk_fp8_f32 per 16 scores;
//   * scaling, all powers of two (exact): keys and values per (batch, head) from their absolute maximum (absmax_kernel ->
//     kv_prepack_fp8_kernel), queries per QUERY (a lane pair owns a query; the pre-scaled q * scale * log2 e), probabilities
//     by 2^8 after the shift by the running maximum -- which therefore is a true running maximum here (a probability must
//     stay <= 1 to fit e4m3's 448), not the lazy one of the bf16x3 kernel;
//   * K / V^T operands are packed once per call into 4 KiB slots of 64 keys = {K, V^T} x {sub-tile of 32 keys} x {k-step} x
//     64 lanes x 8 bytes, streamed by LDS DMA (one 1 KiB instruction per wavefront and slot) into a 4-slot ring three slots
//     ahead; the row sum of the probabilities is taken in fp32 before quantisation.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int F8_SLOT_BYTES = 4096;
constexpr int F8_SLOT_FLOATS = F8_SLOT_BYTES / 4;
constexpr int F8_RING = 4;
constexpr float F8_TARGET = 224.0f;  // scaled absolute maximum lands in [112, 224]: inside e4m3's +-448 with a binade to spare
constexpr float F8_PSCALE = 256.0f;  // probabilities (<= 1) are quantised as p * 2^8

#define MFMA_FP8(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float pow2_scale(float amax) {  // largest power of two s with amax * s <= F8_TARGET (1 for amax = 0)
  if (!(amax > 0.f)) return 1.f;
  return __builtin_ldexpf(1.0f, (int)__builtin_floorf(__builtin_log2f(F8_TARGET / amax)));
}
__device__ __forceinline__ long pack8_fp8(const float (&v)[8]) {  // 8 x e4m3 (round to nearest even), byte i = element i
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

// amax[(b * H + h) * 2 + {0: K, 1: V}] = max |x| over the S rows and the head's 32 columns; grid (ceil(S/64), B), block 32*H <= 1024
__global__ void absmax_kernel(const float* __restrict__ k, const float* __restrict__ v, int ldk, int ldv, int S, int H,
                              unsigned* __restrict__ amax) {
  const int c = threadIdx.x, b = blockIdx.y, s0 = blockIdx.x * 64;
  float mk = 0.f, mv = 0.f;
  for (int s = s0; s < s0 + 64 && s < S; ++s) {
    mk = fmaxf(mk, fabsf(k[((size_t)b * S + s) * ldk + c]));
    mv = fmaxf(mv, fabsf(v[((size_t)b * S + s) * ldv + c]));
  }
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {
    mk = fmaxf(mk, __shfl_xor(mk, d, 64));
    mv = fmaxf(mv, __shfl_xor(mv, d, 64));
  }
  if ((c & 31) == 0) {  // non-negative floats order like their bit patterns
    atomicMax(amax + ((size_t)b * H + (c >> 5)) * 2 + 0, __float_as_uint(mk));
    atomicMax(amax + ((size_t)b * H + (c >> 5)) * 2 + 1, __float_as_uint(mv));
  }
}

// grid (ceil(S/64), H, B), block 256: thread = (op in 0..3 = {K sub 0, K sub 1, V^T sub 0, V^T sub 1}, lane); both k-steps
__global__ void __launch_bounds__(256) kv_prepack_fp8_kernel(const float* __restrict__ k, const float* __restrict__ v, int ldk, int ldv,
                                                              int S, int H, const unsigned* __restrict__ amax, char* __restrict__ slots) {
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z, nt = gridDim.x;
  const int tid = threadIdx.x, op = tid >> 6, lane = tid & 63, r = lane & 31, half = lane >> 5;
  const int which = op >> 1, sub = op & 1;
  const float sc = pow2_scale(__uint_as_float(amax[((size_t)b * H + h) * 2 + which]));
  long* slot = reinterpret_cast<long*>(slots + (((size_t)b * H + h) * nt + t) * F8_SLOT_BYTES);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    float v8[8];
    if (which == 0) {
      // K sub-tile as A operand of S^T = K . Q^T: row = key r, k-slots = dims 16 ks + 8 half + i
      const int key = t * 64 + sub * 32 + r;
      if (key < S) {
        const float* p = k + ((size_t)b * S + key) * ldk + h * 32 + 16 * ks + 8 * half;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), c = *reinterpret_cast<const f32x4*>(p + 4);
        v8[0] = a[0]; v8[1] = a[1]; v8[2] = a[2]; v8[3] = a[3]; v8[4] = c[0]; v8[5] = c[1]; v8[6] = c[2]; v8[7] = c[3];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v8[i] = 0.f;
      }
    } else {
      // V^T sub-tile as A operand of O^T += V^T . P^T: row = dim r, k-slot i of step ks <-> key (i&3) + 16 ks + 8 (i>>2) + 4 half
      // (= the key held by accumulator register 8 ks + i of a lane in half `half` after the first MFMA)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int key = t * 64 + sub * 32 + (i & 3) + 16 * ks + 8 * (i >> 2) + 4 * half;
        v8[i] = key < S ? v[((size_t)b * S + key) * ldv + h * 32 + r] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v8[i] *= sc;
    slot[((which * 2 + sub) * 2 + ks) * 64 + lane] = pack8_fp8(v8);
  }
}

// one 1 KiB piece per wavefront and slot
__device__ __forceinline__ void dma_slot(const char* slots, int t, float* ring, int wave, int lane) {
  const char* base = slots + (size_t)t * F8_SLOT_BYTES + wave * 1024 + lane * 16;
  const auto* src = (const __attribute__((address_space(1))) void*)base;
  auto* dst = (__attribute__((address_space(3))) void*)(ring + (t & (F8_RING - 1)) * F8_SLOT_FLOATS + wave * 256);
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
}

// Grid / work mapping as attn32_v3_kernel (XCD-aware: all query blocks of one (batch, head) on one XCD).
__global__ void __launch_bounds__(256, 4) attn32_fp8_kernel(const float* __restrict__ q, int ldq, const unsigned* __restrict__ amax,
                                                          const char* __restrict__ blob, int L, int S, int H, int B, float scale,
                                                          float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float ring[F8_RING * F8_SLOT_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int nqb = ((L + 31) / 32 + 3) / 4;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int bh = 8 * (jj / nqb) + xcd;
  if (bh >= B * H) return;
  const int h = bh % H, b = bh / H;
  const int qt = (jj % nqb) * 4 + wave;
  const int C = H * 32;
  const int qrow = qt * 32 + j;
  const int qc = qrow < L ? qrow : L - 1;
  const int nt = (S + 63) / 64;
  const char* slots = blob + ((size_t)b * H + h) * nt * F8_SLOT_BYTES;
  dma_slot(slots, 0, ring, wave, lane);
  if (nt > 1) dma_slot(slots, 1, ring, wave, lane);
  if (nt > 2) dma_slot(slots, 2, ring, wave, lane);
  const float sk = pow2_scale(__uint_as_float(amax[((size_t)b * H + h) * 2 + 0]));
  const float sv = pow2_scale(__uint_as_float(amax[((size_t)b * H + h) * 2 + 1]));
  long qf[2];
  float dsc;
  {
    const float qs = scale * 1.44269504088896340736f;
    const float* qp = q + ((size_t)b * L + qc) * ldq + h * 32 + 8 * hi;
    float v8[2][8];
    float am = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(qp + 16 * m), b4 = *reinterpret_cast<const f32x4*>(qp + 16 * m + 4);
      const float w8[8] = {a4[0] * qs, a4[1] * qs, a4[2] * qs, a4[3] * qs, b4[0] * qs, b4[1] * qs, b4[2] * qs, b4[3] * qs};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v8[m][i] = w8[i];
        am = fmaxf(am, fabsf(w8[i]));
      }
    }
    am = fmaxf(am, nm_shfl_xor32(am));  // the query's other 16 dims live in the other wavefront half
    const float sq = pow2_scale(am);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v8[m][i] *= sq;
      qf[m] = pack8_fp8(v8[m]);
    }
    dsc = 1.0f / (sq * sk);  // exact: powers of two
  }
  f32x16 o;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
  float mrun = -1e30f, lrun = 0.f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int t = 0; t < nt; ++t) {
    if (t > 0) {
      // slot t has landed when at most the DMA instructions of slots t+1, t+2 remain in flight
      if (t + 2 < nt) NM_WAIT_VMCNT(2);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everybody's piece of slot t landed; nobody reads slot t-1 any more
    }
    if (t + 3 < nt) dma_slot(slots, t + 3, ring, wave, lane);
    const long* s8 = reinterpret_cast<const long*>(ring + (t & (F8_RING - 1)) * F8_SLOT_FLOATS) + lane;
    f32x16 r[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      r[sub] = MFMA_FP8(s8[((0 * 2 + sub) * 2 + 0) * 64], qf[0], zero);
      r[sub] = MFMA_FP8(s8[((0 * 2 + sub) * 2 + 1) * 64], qf[1], r[sub]);
    }
    if (t == nt - 1 && (S & 63)) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (t * 64 + sub * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi >= S) r[sub][i] = -__builtin_inff();
    }
    // true running maximum (log2 units), per query = lane pair
    float mlo, mhi;
    nm_swap32(fmaxf(nm_max16(r[0]), nm_max16(r[1])), mlo, mhi);
    const float mx = fmaxf(mlo, mhi) * dsc;
    if (__builtin_amdgcn_ballot_w64(mx > mrun) != 0) {
      const float mnew = fmaxf(mrun, mx);
      const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
      lrun *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] *= alpha;
      mrun = mnew;
    }
    float ps = 0.f;
    long pf[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      float p[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        p[i] = __builtin_amdgcn_exp2f(NM_FMA(r[sub][i], dsc, -mrun));
        ps += p[i];
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const float p8[8] = {p[8 * m] * F8_PSCALE, p[8 * m + 1] * F8_PSCALE, p[8 * m + 2] * F8_PSCALE, p[8 * m + 3] * F8_PSCALE,
                             p[8 * m + 4] * F8_PSCALE, p[8 * m + 5] * F8_PSCALE, p[8 * m + 6] * F8_PSCALE, p[8 * m + 7] * F8_PSCALE};
        pf[sub][m] = pack8_fp8(p8);
      }
    }
    lrun += ps;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      o = MFMA_FP8(s8[((1 * 2 + sub) * 2 + 0) * 64], pf[sub][0], o);
      o = MFMA_FP8(s8[((1 * 2 + sub) * 2 + 1) * 64], pf[sub][1], o);
    }
  }
  const float ltot = lrun + nm_shfl_xor32(lrun);
  if (qrow < L) {
    const float inv = 1.0f / (ltot * (F8_PSCALE * sv));
    float* op = out + ((size_t)b * L + qrow) * C + h * 32 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w4 = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(op + 8 * g) = w4;
    }
  }
}

constexpr size_t F8_HEADER = 4096;  // absolute maxima (2 x 4 bytes per (batch, head)) in front of the slots, padded

}  // namespace

extern "C" size_t nm_attention_fp8_workspace_bytes(int B, int S, int heads) {
  if (B <= 0 || S <= 0 || heads <= 0) return 0;
  const size_t head = ((size_t)B * heads * 8 + F8_HEADER - 1) / F8_HEADER * F8_HEADER;
  return head + (size_t)B * heads * ((S + 63) / 64) * F8_SLOT_BYTES;
}

extern "C" int nm_attention_fp8(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S, int heads,
                                float scale, void* workspace, float* out, nmStream_t stream) {
  NM_CHECK_ARG(q && k && v && out && workspace && B > 0 && L > 0 && S > 0 && heads > 0);
  const int C = heads * 32;
  if (ldq < C || ldk < C || ldv < C || (ldq | ldk | ldv) % 4) return NM_ERR_ARG;
  if (B > 65535 || heads > 32) return NM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const size_t head = ((size_t)B * heads * 8 + F8_HEADER - 1) / F8_HEADER * F8_HEADER;
  unsigned* amax = (unsigned*)workspace;
  char* slots = (char*)workspace + head;
  if (hipMemsetAsync(amax, 0, (size_t)B * heads * 8, s) != hipSuccess) return NM_ERR_LAUNCH;
  const int nt = (S + 63) / 64;
  absmax_kernel<<<dim3(nt, B), 32 * heads, 0, s>>>(k, v, ldk, ldv, S, heads, amax);
  kv_prepack_fp8_kernel<<<dim3(nt, heads, B), 256, 0, s>>>(k, v, ldk, ldv, S, heads, amax, slots);
  const int nqb = ((L + 31) / 32 + 3) / 4;
  const long long grid = (long long)((B * heads + 7) / 8) * 8 * nqb;
  if (grid > 0x7fffffffLL) return NM_ERR_UNSUPPORTED;
  attn32_fp8_kernel<<<(unsigned)grid, 256, 0, s>>>(q, ldq, amax, slots, L, S, heads, B, scale, out);
  return nm_launch_status();
}
