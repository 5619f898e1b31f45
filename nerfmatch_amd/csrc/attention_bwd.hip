// Backward of softmax multi-head attention (head dim 32) on the fp32 matrix cores -- the training-side counterpart of
// attn32_kernel (attention.hip); reference: autograd through FullAttention.forward, modules/attention.py:44-57.
//
// With s = scale q.k, P = softmax_keys(s), O = P V and an incoming dO:
//     D_l = sum_c dO[l,c] O[l,c]      dP = dO V^T      dS = P o (dP - D)      dQ = scale dS K      dK = scale dS^T Q      dV = P^T dO
// The (L,S) matrices never exist: both kernels recompute 32x32 score tiles exactly like the forward pass (transposed
// scores: one lane = one query (dq kernel) or one key (dkv kernel), 16 partners in registers).
//   attn32_bwd_dq_kernel   one wavefront = 32 queries; pass 1 over the keys rebuilds the soft-max statistics
//                          (log2 domain), pass 2 accumulates dQ; writes lse / D for the second kernel.  No atomics.
//   attn32_bwd_dkv_kernel  one wavefront = 32 keys; loops over the query tiles, accumulates dK and dV.  No atomics.
// Seven 32x32x32 contractions per tile pair instead of the minimal five (S and dP are computed by both kernels) buy
// deterministic, atomic-free gradients.
#include "common.h"

namespace {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int LD = 36;                 // floats per LDS row (32 + 4 pad: conflict-free 16-byte row reads)
constexpr int TILE = 32 * LD;          // one 32 x 32 tile
constexpr int SLOT = 2 * TILE + 64;    // two tiles + 2 x 32 per-row scalars
constexpr float LOG2E = 1.44269504088896340736f;

__device__ __forceinline__ int nrow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// rows[32 x 32] . regs: out[row i][col j] = sum_d tile[i][d] * breg(j)[d]   (A = LDS rows, B = registers of lane j)
__device__ __forceinline__ f32x16 rows_times_regs(const float* tile, int j, int hi, const float (&breg)[16]) {
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float* a = tile + j * LD + 4 * hi;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(a + 8 * c);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = MFMA32(a4[t], breg[4 * c + t], acc);
  }
  return acc;
}

// acc[d][col j] += sum_rows tile[row][d] * p(j)[row]   (A = LDS tile read transposed, B = the registers a
// rows_times_regs result left in place: register r <-> row nrow(r, hi))
__device__ __forceinline__ void cols_times_regs(const float* tile, int j, int hi, const f32x16& p, f32x16& acc) {
  const float* a = tile + 4 * hi * LD + j;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc = MFMA32(a[((r & 3) + 8 * (r >> 2)) * LD], p[r], acc);
}

__device__ __forceinline__ void load16(const float* p, float s, float (&reg)[16]) {  // dims 8c + 4hi + t (p already offset by 4hi)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 t4 = *reinterpret_cast<const f32x4*>(p + 8 * c);
#pragma unroll
    for (int t = 0; t < 4; ++t) reg[4 * c + t] = t4[t] * s;
  }
}

__device__ __forceinline__ void store16(float* p, const f32x16& a, float s) {
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(p + 8 * g) = f32x4{a[4 * g] * s, a[4 * g + 1] * s, a[4 * g + 2] * s, a[4 * g + 3] * s};
}

struct BwdArgs {
  const float *q, *k, *v, *o, *d_o;
  int ldq, ldk, ldv, ldo, lddo;
  float *dq, *dk, *dv;
  int lddq, lddk, lddv;
  int L, S, H;
  float scale;
  float *lse, *dsum;  // [B][H][L]
};

__global__ void __launch_bounds__(256) attn32_bwd_dq_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qrow = (blockIdx.x * 4 + wave) * 32 + j;
  const int qc = qrow < a.L ? qrow : a.L - 1;
  float qreg[16], doreg[16];
  load16(a.q + ((size_t)b * a.L + qc) * a.ldq + h * 32 + 4 * hi, a.scale * LOG2E, qreg);
  load16(a.d_o + ((size_t)b * a.L + qc) * a.lddo + h * 32 + 4 * hi, 1.0f, doreg);
  float dsum = 0.f;
  {
    float oreg[16];
    load16(a.o + ((size_t)b * a.L + qc) * a.ldo + h * 32 + 4 * hi, 1.0f, oreg);
#pragma unroll
    for (int i = 0; i < 16; ++i) dsum = NM_FMA(doreg[i], oreg[i], dsum);
    dsum += nm_shfl_xor32(dsum);
  }
  const float* kbase = a.k + (size_t)b * a.S * a.ldk + h * 32;
  const float* vbase = a.v + (size_t)b * a.S * a.ldv + h * 32;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4, st_off = lrow * LD + lcol;
  const int nt = (a.S + 31) / 32;
  auto gload = [&](int t, f32x4& kk, f32x4& vv) {
    const int key = t * 32 + lrow;
    const size_t row = (size_t)(key < a.S ? key : a.S - 1);
    kk = *reinterpret_cast<const f32x4*>(kbase + row * a.ldk + lcol);
    vv = *reinterpret_cast<const f32x4*>(vbase + row * a.ldv + lcol);
  };
  auto sstore = [&](float* slot, const f32x4& kk, const f32x4& vv) {
    *reinterpret_cast<f32x4*>(slot + st_off) = kk;
    *reinterpret_cast<f32x4*>(slot + TILE + st_off) = vv;
  };
  auto mask_tail = [&](int t, f32x16& sc) {
    if (t == nt - 1 && (a.S & 31)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * 32 + nrow(r, hi) >= a.S) sc[r] = -__builtin_inff();
    }
  };
  f32x4 kst, vst;
  // ---- pass 1: soft-max statistics of each query (running max / sum in the log2 domain)
  float mrun = -__builtin_inff(), lrun = 0.f;
  gload(0, kst, vst);
  sstore(sm, kst, vst);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const float* slot = sm + (t & 1) * SLOT;
    if (t + 1 < nt) gload(t + 1, kst, vst);
    f32x16 sc = rows_times_regs(slot, j, hi, qreg);
    mask_tail(t, sc);
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    mx = fmaxf(mx, nm_shfl_xor32(mx));
    const float mnew = fmaxf(mrun, mx);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(sc[r] - mnew);
    ps += nm_shfl_xor32(ps);
    lrun = lrun * __builtin_amdgcn_exp2f(mrun - mnew) + ps;
    mrun = mnew;
    if (t + 1 < nt) sstore(sm + ((t + 1) & 1) * SLOT, kst, vst);
    __syncthreads();
  }
  const float lse = mrun + __builtin_amdgcn_logf(lrun);  // v_log_f32 is log2
  if (qrow < a.L && hi == 0) {
    a.lse[((size_t)b * a.H + h) * a.L + qrow] = lse;
    a.dsum[((size_t)b * a.H + h) * a.L + qrow] = dsum;
  }
  // ---- pass 2: dQ
  f32x16 dq;
#pragma unroll
  for (int i = 0; i < 16; ++i) dq[i] = 0.f;
  gload(0, kst, vst);
  sstore(sm, kst, vst);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const float* slot = sm + (t & 1) * SLOT;
    if (t + 1 < nt) gload(t + 1, kst, vst);
    f32x16 sc = rows_times_regs(slot, j, hi, qreg);
    mask_tail(t, sc);
    const f32x16 dp = rows_times_regs(slot + TILE, j, hi, doreg);
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r] - lse) * (dp[r] - dsum);  // dS
    cols_times_regs(slot, j, hi, sc, dq);
    if (t + 1 < nt) sstore(sm + ((t + 1) & 1) * SLOT, kst, vst);
    __syncthreads();
  }
  if (qrow < a.L) store16(a.dq + ((size_t)b * a.L + qrow) * a.lddq + h * 32 + 4 * hi, dq, a.scale);
}

__global__ void __launch_bounds__(256) attn32_bwd_dkv_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int krow = (blockIdx.x * 4 + wave) * 32 + j;
  const int kc = krow < a.S ? krow : a.S - 1;
  float kreg[16], vreg[16];
  load16(a.k + ((size_t)b * a.S + kc) * a.ldk + h * 32 + 4 * hi, a.scale * LOG2E, kreg);
  load16(a.v + ((size_t)b * a.S + kc) * a.ldv + h * 32 + 4 * hi, 1.0f, vreg);
  const float* qbase = a.q + (size_t)b * a.L * a.ldq + h * 32;
  const float* dobase = a.d_o + (size_t)b * a.L * a.lddo + h * 32;
  const float* lsebase = a.lse + ((size_t)b * a.H + h) * a.L;
  const float* dsbase = a.dsum + ((size_t)b * a.H + h) * a.L;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4, st_off = lrow * LD + lcol;
  const int nt = (a.L + 31) / 32;
  struct Stage {
    f32x4 qq, dd;
    float sc;
  };
  auto gload = [&](int t, Stage& s) {
    const int qr = t * 32 + lrow;
    const size_t row = (size_t)(qr < a.L ? qr : a.L - 1);
    s.qq = *reinterpret_cast<const f32x4*>(qbase + row * a.ldq + lcol);
    s.dd = *reinterpret_cast<const f32x4*>(dobase + row * a.lddo + lcol);
    if (tid < 64) {  // per-query scalars: lse (rows past L: +inf -> probability 0) then D
      const int q2 = t * 32 + (tid & 31);
      if (tid < 32) s.sc = q2 < a.L ? lsebase[q2] : __builtin_inff();
      else s.sc = q2 < a.L ? dsbase[q2] : 0.f;
    }
  };
  auto sstore = [&](float* slot, const Stage& s) {
    *reinterpret_cast<f32x4*>(slot + st_off) = s.qq;
    *reinterpret_cast<f32x4*>(slot + TILE + st_off) = s.dd;
    if (tid < 64) slot[2 * TILE + tid] = s.sc;
  };
  f32x16 dk, dv;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    dk[i] = 0.f;
    dv[i] = 0.f;
  }
  Stage st;
  gload(0, st);
  sstore(sm, st);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const float* slot = sm + (t & 1) * SLOT;
    if (t + 1 < nt) gload(t + 1, st);
    f32x16 p = rows_times_regs(slot, j, hi, kreg);           // scores (log2 domain): lane = key, register <-> query
    const f32x16 dp = rows_times_regs(slot + TILE, j, hi, vreg);
    f32x16 ds;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(slot + 2 * TILE + 8 * c + 4 * hi);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(slot + 2 * TILE + 32 + 8 * c + 4 * hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * c + e;
        p[r] = __builtin_amdgcn_exp2f(p[r] - l4[e]);
        ds[r] = p[r] * (dp[r] - d4[e]);
      }
    }
    cols_times_regs(slot + TILE, j, hi, p, dv);  // dV^T += dO^T . P
    cols_times_regs(slot, j, hi, ds, dk);        // dK^T += Q^T . dS
    if (t + 1 < nt) sstore(sm + ((t + 1) & 1) * SLOT, st);
    __syncthreads();
  }
  if (krow < a.S) {
    store16(a.dk + ((size_t)b * a.S + krow) * a.lddk + h * 32 + 4 * hi, dk, a.scale);
    store16(a.dv + ((size_t)b * a.S + krow) * a.lddv + h * 32 + 4 * hi, dv, 1.0f);
  }
}

// ---- small sequences (the 5x5 fine windows: 25 tokens, head dim 16): one thread per (query or key), plain FMAs ------
// grid (H, B); block 64; L, S <= 64, head dim D = 16.  Thread i < L owns query i (dq), thread i < S owns key i (dk, dv);
// P and dS are exchanged through LDS.
template <int D>
__global__ void __launch_bounds__(64) attn_small_bwd_kernel(BwdArgs a) {
  __shared__ float sq[64][D + 1], sk[64][D + 1], sv[64][D + 1], sdo[64][D + 1], sp[64][65], sds[64][65];
  const int i = threadIdx.x, h = blockIdx.x, b = blockIdx.y, L = a.L, S = a.S;
  if (i < L) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      sq[i][c] = a.q[((size_t)b * L + i) * a.ldq + h * D + c];
      sdo[i][c] = a.d_o[((size_t)b * L + i) * a.lddo + h * D + c];
    }
  }
  if (i < S) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      sk[i][c] = a.k[((size_t)b * S + i) * a.ldk + h * D + c];
      sv[i][c] = a.v[((size_t)b * S + i) * a.ldv + h * D + c];
    }
  }
  __syncthreads();
  if (i < L) {
    float dsum = 0.f;
#pragma unroll
    for (int c = 0; c < D; ++c) dsum = NM_FMA(sdo[i][c], a.o[((size_t)b * L + i) * a.ldo + h * D + c], dsum);
    float mx = -__builtin_inff();
    for (int s = 0; s < S; ++s) {
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) d = NM_FMA(sq[i][c], sk[s][c], d);
      d *= a.scale;
      sp[i][s] = d;
      mx = fmaxf(mx, d);
    }
    float sum = 0.f;
    for (int s = 0; s < S; ++s) {
      const float e = __expf(sp[i][s] - mx);
      sp[i][s] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    float dq[D];
#pragma unroll
    for (int c = 0; c < D; ++c) dq[c] = 0.f;
    for (int s = 0; s < S; ++s) {
      const float p = sp[i][s] * inv;
      float dp = 0.f;
#pragma unroll
      for (int c = 0; c < D; ++c) dp = NM_FMA(sdo[i][c], sv[s][c], dp);
      const float ds = p * (dp - dsum);
      sp[i][s] = p;
      sds[i][s] = ds;
#pragma unroll
      for (int c = 0; c < D; ++c) dq[c] = NM_FMA(ds, sk[s][c], dq[c]);
    }
#pragma unroll
    for (int c = 0; c < D; ++c) a.dq[((size_t)b * L + i) * a.lddq + h * D + c] = dq[c] * a.scale;
  }
  __syncthreads();
  if (i < S) {
    float dk[D], dv[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      dk[c] = 0.f;
      dv[c] = 0.f;
    }
    for (int l = 0; l < L; ++l) {
      const float p = sp[l][i], ds = sds[l][i];
#pragma unroll
      for (int c = 0; c < D; ++c) {
        dv[c] = NM_FMA(p, sdo[l][c], dv[c]);
        dk[c] = NM_FMA(ds, sq[l][c], dk[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
      a.dk[((size_t)b * S + i) * a.lddk + h * D + c] = dk[c] * a.scale;
      a.dv[((size_t)b * S + i) * a.lddv + h * D + c] = dv[c];
    }
  }
}

}  // namespace

// attention_bwd_v2.hip
size_t nm_internal_attn_bwd_v2_workspace(int B, int L, int S, int heads);
int nm_internal_attn_bwd_v2(const float* q, const float* k, const float* v, const float* o, const float* d_o, int ldq, int ldk, int ldv,
                            int ldo, int lddo, int B, int L, int S, int heads, float scale, float* dq, float* dk, float* dv, int lddq,
                            int lddk, int lddv, void* workspace, hipStream_t s, const float* nlse_fwd);

extern "C" size_t nm_attention_bwd_workspace_bytes(int B, int L, int S, int heads, int flags) {
  if (B <= 0 || L <= 0 || S <= 0 || heads <= 0) return 0;
  if (flags & NM_ATTN_BF16X3) return nm_internal_attn_bwd_v2_workspace(B, L, S, heads);
  return (size_t)2 * B * heads * L * sizeof(float);
}

extern "C" int nm_attention_bwd_lse(const float* q, const float* k, const float* v, const float* o, const float* d_o, int ldq, int ldk,
                                    int ldv, int ldo, int lddo, int B, int L, int S, int heads, int head_dim, float scale, float* dq,
                                    float* dk, float* dv, int lddq, int lddk, int lddv, int flags, const float* nlse, void* workspace,
                                    size_t workspace_bytes, nmStream_t stream);

extern "C" int nm_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, int ldq, int ldk,
                                int ldv, int ldo, int lddo, int B, int L, int S, int heads, int head_dim, float scale, float* dq,
                                float* dk, float* dv, int lddq, int lddk, int lddv, int flags, void* workspace, size_t workspace_bytes,
                                nmStream_t stream) {
  return nm_attention_bwd_lse(q, k, v, o, d_o, ldq, ldk, ldv, ldo, lddo, B, L, S, heads, head_dim, scale, dq, dk, dv, lddq, lddk, lddv, flags,
                              nullptr, workspace, workspace_bytes, stream);
}

extern "C" int nm_attention_bwd_lse(const float* q, const float* k, const float* v, const float* o, const float* d_o, int ldq, int ldk,
                                    int ldv, int ldo, int lddo, int B, int L, int S, int heads, int head_dim, float scale, float* dq,
                                    float* dk, float* dv, int lddq, int lddk, int lddv, int flags, const float* nlse, void* workspace,
                                    size_t workspace_bytes, nmStream_t stream) {
  NM_CHECK_ARG(q && k && v && o && d_o && dq && dk && dv && B > 0 && L > 0 && S > 0 && heads > 0);
  const int C = heads * head_dim;
  NM_CHECK_ARG(ldq >= C && ldk >= C && ldv >= C && ldo >= C && lddo >= C && lddq >= C && lddk >= C && lddv >= C);
  hipStream_t s = (hipStream_t)stream;
  BwdArgs a{q, k, v, o, d_o, ldq, ldk, ldv, ldo, lddo, dq, dk, dv, lddq, lddk, lddv, L, S, heads, scale, nullptr, nullptr};
  if (L <= 64 && S <= 64 && head_dim == 16) {
    attn_small_bwd_kernel<16><<<dim3(heads, B), 64, 0, s>>>(a);
    return nm_launch_status();
  }
  if (head_dim != 32) return NM_ERR_UNSUPPORTED;
  if ((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) % 4 != 0) return NM_ERR_UNSUPPORTED;  // 16-byte row pieces
  if (!workspace || workspace_bytes < nm_attention_bwd_workspace_bytes(B, L, S, heads, flags)) return NM_ERR_WORKSPACE;
  if (flags & NM_ATTN_BF16X3)
    return nm_internal_attn_bwd_v2(q, k, v, o, d_o, ldq, ldk, ldv, ldo, lddo, B, L, S, heads, scale, dq, dk, dv, lddq, lddk, lddv, workspace, s, nlse);
  if (nlse) return NM_ERR_UNSUPPORTED;  // (only the split-bf16 kernels take the forward pass's log-sum-exp)
  a.lse = (float*)workspace;
  a.dsum = a.lse + (size_t)B * heads * L;
  attn32_bwd_dq_kernel<<<dim3((L + 127) / 128, heads, B), 256, 0, s>>>(a);
  attn32_bwd_dkv_kernel<<<dim3((S + 127) / 128, heads, B), 256, 0, s>>>(a);
  return nm_launch_status();
}
