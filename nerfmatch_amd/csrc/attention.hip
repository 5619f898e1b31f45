// LayerNorm and softmax multi-head attention.
//
// attn32_kernel: flash-style attention for head dim 32 on the fp32 matrix cores; the (L,S,H) score tensor the
// reference materialises (415 MB at 3600^2 x 8 heads) never exists.  One wavefront owns 32 queries of one head.
//   S^T tile  = K_tile[32 keys x 32] . Q^T[32 x 32 queries]      (A = keys, B = queries; computing the TRANSPOSED
//               scores puts all keys of a query in one lane pair: row max / sum need one cross-half shuffle)
//   O^T      += V_tile^T[32 d x 32 keys] . P^T[32 keys x 32 queries]   (B = the probabilities exactly as the
//               first MFMA left them in registers: keys nrow(r, half) <-> k-step r; no LDS, no re-layout)
// K / V tiles are staged once per workgroup in LDS (see attn32_kernel) and shared by its 4 query tiles.
// attn_small_kernel: sequences <= 64 tokens (the 5x5 fine windows: 25 tokens, head dim 16): one thread per query,
// K/V of the (batch, head) in LDS, plain fp32 FMAs.
#include "common.h"

namespace {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wavefront per row; dim = 64 * PER
template <int PER>
__device__ __forceinline__ void layernorm_row(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, int rows, float eps,
                                              float* __restrict__ y, int row) {
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int dim = 64 * PER;
  const float* xr = x + (size_t)row * dim;
  float v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = xr[lane + 64 * i];
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float d = v[i] - mean;
    q = NM_FMA(d, d, q);
  }
  const float var = wave_sum(q) / (float)dim;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    y[(size_t)row * dim + c] = (v[i] - mean) * rstd * g[c] + b[c];
  }
}
template <int PER>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, int rows, float eps, float* __restrict__ y) {
  layernorm_row<PER>(x, g, b, rows, eps, y, blockIdx.x * 4 + (threadIdx.x >> 6));
}
// two tensors with their own affine parameters in one launch (round 5: the two pre-norms of a cross-attention layer): workgroups [0, g0) take
// the first, the rest the second; a row's arithmetic is layernorm_kernel's
template <int PER>
__global__ void __launch_bounds__(256) layernorm2_kernel(const float* __restrict__ x0, const float* __restrict__ g0, const float* __restrict__ b0, int rows0,
                                                         float eps0, float* __restrict__ y0, int wg0, const float* __restrict__ x1,
                                                         const float* __restrict__ g1, const float* __restrict__ b1, int rows1, float eps1,
                                                         float* __restrict__ y1) {
  if ((int)blockIdx.x < wg0) layernorm_row<PER>(x0, g0, b0, rows0, eps0, y0, blockIdx.x * 4 + (threadIdx.x >> 6));
  else layernorm_row<PER>(x1, g1, b1, rows1, eps1, y1, (blockIdx.x - wg0) * 4 + (threadIdx.x >> 6));
}

// Workgroup = 4 wavefronts = 128 queries of one (batch, head); every 32-key K/V tile is fetched ONCE per workgroup
// with coalesced 16-byte loads (thread t: key t/8, 16 bytes at column 4*(t%8)) and staged in a 3-slot LDS ring (rows
// padded to 36 floats so the 16-byte A-operand reads of 16 consecutive keys fall on distinct banks).
// Software pipeline per iteration t (one barrier):
//     global loads of tile t+2  ->  QK^T MFMAs of tile t+1  ||  softmax VALU of tile t  ->  PV MFMAs of tile t
// so the 16 score MFMAs of the next tile cover the softmax of the current one (one wavefront per SIMD has nothing else
// to overlap with).  Softmax runs in the exp2 domain: log2(e) is folded into the query scale and the exponentials are
// single v_exp_f32 instructions.
constexpr int KV_LD = 36;
constexpr int KV_SLOT = 2 * 32 * KV_LD;  // floats per ring slot: K tile then V tile

__global__ void __launch_bounds__(256) attn32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int ldq, int ldk, int ldv, int L, int S,
                                                      int H, float scale, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sm[3 * KV_SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int qt = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  const int C = H * 32;
  const int qrow = qt * 32 + j;
  const int qc = qrow < L ? qrow : L - 1;
  float qreg[16];
  {
    const float qs = scale * 1.44269504088896340736f;  // scores are kept in the log2 domain
    const float* qp = q + ((size_t)b * L + qc) * ldq + h * 32 + 4 * hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(qp + 8 * c);
#pragma unroll
      for (int t = 0; t < 4; ++t) qreg[4 * c + t] = t4[t] * qs;
    }
  }
  f32x16 o;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
  float mrun = -__builtin_inff(), lrun = 0.f;
  const float* kbase = k + (size_t)b * S * ldk + h * 32;
  const float* vbase = v + (size_t)b * S * ldv + h * 32;
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;  // staging role of this thread
  const int nt = (S + 31) / 32;
  const int st_off = lrow * KV_LD + lcol;            // where this thread stages its 16 bytes (K; V at +32*KV_LD)
  const int ka_off = j * KV_LD + 4 * hi;             // A operand of QK^T: key row j
  const int va_off = 32 * KV_LD + 4 * hi * KV_LD + j;  // A operand of PV: V^T row j (feature), keys nrow(r, hi)
  auto gload = [&](int t, f32x4& kk, f32x4& vv) {
    const int key = t * 32 + lrow;
    const size_t row = (size_t)(key < S ? key : S - 1);
    kk = *reinterpret_cast<const f32x4*>(kbase + row * ldk + lcol);
    vv = *reinterpret_cast<const f32x4*>(vbase + row * ldv + lcol);
  };
  auto sstore = [&](float* slot, const f32x4& kk, const f32x4& vv) {
    *reinterpret_cast<f32x4*>(slot + st_off) = kk;
    *reinterpret_cast<f32x4*>(slot + st_off + 32 * KV_LD) = vv;
  };
  auto scores = [&](const float* slot) {
    f32x16 sc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 ka = *reinterpret_cast<const f32x4*>(slot + ka_off + 8 * c);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) sc = MFMA32(ka[tt], qreg[4 * c + tt], sc);
    }
    return sc;
  };
  f32x4 kst, vst;
  gload(0, kst, vst);
  sstore(sm, kst, vst);
  if (nt > 1) {
    gload(1, kst, vst);
    sstore(sm + KV_SLOT, kst, vst);
  }
  __syncthreads();
  f32x16 sc_next = scores(sm);
  int cur = 0;  // ring slot of tile t
  for (int t = 0; t < nt; ++t) {
    const int nxt = cur == 2 ? 0 : cur + 1, nx2 = nxt == 2 ? 0 : nxt + 1;
    f32x16 sc = sc_next;
    if (t + 2 < nt) gload(t + 2, kst, vst);
    if (t + 1 < nt) sc_next = scores(sm + nxt * KV_SLOT);
    if (t == nt - 1 && (S & 31)) {  // ragged last tile: keys >= S get zero probability
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= S) sc[r] = -__builtin_inff();
    }
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    mx = fmaxf(mx, nm_shfl_xor32(mx));
    const float mnew = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[r] = __builtin_amdgcn_exp2f(sc[r] - mnew);
      ps += sc[r];
    }
    ps += nm_shfl_xor32(ps);
    lrun = lrun * alpha + ps;
    mrun = mnew;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] *= alpha;
    const float* vs = sm + cur * KV_SLOT + va_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) o = MFMA32(vs[((r & 3) + 8 * (r >> 2)) * KV_LD], sc[r], o);
    if (t + 2 < nt) sstore(sm + nx2 * KV_SLOT, kst, vst);
    __syncthreads();
    cur = nxt;
  }
  if (qrow < L) {
    const float inv = 1.0f / lrun;
    float* op = out + ((size_t)b * L + qrow) * C + h * 32 + 4 * hi;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 w4 = {o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(op + 8 * g) = w4;
    }
  }
}

// grid (H, B); block = 64 threads; thread i < L owns query i.  S <= 64, D <= 32.
template <int D>
__global__ void __launch_bounds__(64) attn_small_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, int ldq, int ldk, int ldv, int L, int S,
                                                         int H, float scale, float* __restrict__ out) {
  __shared__ float sk[64 * D], sv[64 * D];
  const int h = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
  const int C = H * D;
  for (int e = i; e < S * D; e += 64) {
    const int row = e / D, d = e % D;
    sk[e] = k[((size_t)b * S + row) * ldk + h * D + d];
    sv[e] = v[((size_t)b * S + row) * ldv + h * D + d];
  }
  __syncthreads();
  if (i >= L) return;
  float qr[D], o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    qr[d] = q[((size_t)b * L + i) * ldq + h * D + d] * scale;
    o[d] = 0.f;
  }
  // two passes like the reference (scores -> max -> exp/sum); S is tiny
  float mx = -__builtin_inff();
  for (int s = 0; s < S; ++s) {
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dot = NM_FMA(qr[d], sk[s * D + d], dot);
    mx = fmaxf(mx, dot);
  }
  float l = 0.f;
  for (int s = 0; s < S; ++s) {
    float dot = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) dot = NM_FMA(qr[d], sk[s * D + d], dot);
    const float p = expf(dot - mx);
    l += p;
#pragma unroll
    for (int d = 0; d < D; ++d) o[d] = NM_FMA(p, sv[s * D + d], o[d]);
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < D; ++d) out[((size_t)b * L + i) * C + h * D + d] = o[d] * inv;
}

}  // namespace

extern "C" int nm_layernorm(const float* x, const float* gamma, const float* beta, int rows, int dim, float eps, float* y,
                            nmStream_t stream) {
  NM_CHECK_ARG(x && gamma && beta && y && rows > 0 && dim > 0);
  hipStream_t s = (hipStream_t)stream;
  const int grid = (rows + 3) / 4;
  switch (dim) {
    case 64: layernorm_kernel<1><<<grid, 256, 0, s>>>(x, gamma, beta, rows, eps, y); break;
    case 128: layernorm_kernel<2><<<grid, 256, 0, s>>>(x, gamma, beta, rows, eps, y); break;
    case 256: layernorm_kernel<4><<<grid, 256, 0, s>>>(x, gamma, beta, rows, eps, y); break;
    case 512: layernorm_kernel<8><<<grid, 256, 0, s>>>(x, gamma, beta, rows, eps, y); break;
    default: return NM_ERR_UNSUPPORTED;
  }
  return nm_launch_status();
}

extern "C" int nm_layernorm2(const float* x0, const float* gamma0, const float* beta0, int rows0, float eps0, float* y0, const float* x1,
                             const float* gamma1, const float* beta1, int rows1, float eps1, float* y1, int dim, nmStream_t stream) {
  NM_CHECK_ARG(x0 && gamma0 && beta0 && y0 && x1 && gamma1 && beta1 && y1 && rows0 > 0 && rows1 > 0 && dim > 0);
  hipStream_t s = (hipStream_t)stream;
  const int wg0 = (rows0 + 3) / 4, grid = wg0 + (rows1 + 3) / 4;
  switch (dim) {
    case 64: layernorm2_kernel<1><<<grid, 256, 0, s>>>(x0, gamma0, beta0, rows0, eps0, y0, wg0, x1, gamma1, beta1, rows1, eps1, y1); break;
    case 128: layernorm2_kernel<2><<<grid, 256, 0, s>>>(x0, gamma0, beta0, rows0, eps0, y0, wg0, x1, gamma1, beta1, rows1, eps1, y1); break;
    case 256: layernorm2_kernel<4><<<grid, 256, 0, s>>>(x0, gamma0, beta0, rows0, eps0, y0, wg0, x1, gamma1, beta1, rows1, eps1, y1); break;
    case 512: layernorm2_kernel<8><<<grid, 256, 0, s>>>(x0, gamma0, beta0, rows0, eps0, y0, wg0, x1, gamma1, beta1, rows1, eps1, y1); break;
    default: return NM_ERR_UNSUPPORTED;
  }
  return nm_launch_status();
}

extern "C" int nm_attention_ex(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S,
                               int heads, int head_dim, float scale, int flags, float* out, nmStream_t stream) {
  NM_CHECK_ARG(q && k && v && out && B > 0 && L > 0 && S > 0 && heads > 0);
  const int C = heads * head_dim;
  if (ldq < C || ldk < C || ldv < C || (ldq | ldk | ldv) % 4) return NM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (S <= 64 && L <= 64 && (head_dim == 16 || head_dim == 32)) {
    dim3 grid(heads, B);
    if (head_dim == 16) attn_small_kernel<16><<<grid, 64, 0, s>>>(q, k, v, ldq, ldk, ldv, L, S, heads, scale, out);
    else attn_small_kernel<32><<<grid, 64, 0, s>>>(q, k, v, ldq, ldk, ldv, L, S, heads, scale, out);
    return nm_launch_status();
  }
  if (head_dim != 32) return NM_ERR_UNSUPPORTED;
  if (B > 65535 || heads > 65535) return NM_ERR_UNSUPPORTED;
  // the split-bf16 kernel streams pre-split operands from a workspace: nm_attention_ws (its first generation, which split
  // while staging and needed none, is gone from the library: DESIGN.md section 3.4)
  if (flags & NM_ATTN_BF16X3) return NM_ERR_WORKSPACE;
  dim3 grid(((L + 31) / 32 + 3) / 4, heads, B);
  attn32_kernel<<<grid, 256, 0, s>>>(q, k, v, ldq, ldk, ldv, L, S, heads, scale, out);
  return nm_launch_status();
}

size_t nm_internal_attn_v2_workspace(int B, int S, int heads);
int nm_internal_attn_v2(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S, int heads,
                        float scale, void* workspace, float* out, hipStream_t s, float* nlse_out);

extern "C" size_t nm_attention_workspace_bytes(int B, int S, int heads) {
  if (B <= 0 || S <= 0 || heads <= 0) return 0;
  return nm_internal_attn_v2_workspace(B, S, heads);
}

extern "C" int nm_attention_ws(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S,
                               int heads, int head_dim, float scale, int flags, void* workspace, float* out, nmStream_t stream) {
  const bool small = S <= 64 && L <= 64 && (head_dim == 16 || head_dim == 32);
  if (!(flags & NM_ATTN_BF16X3) || head_dim != 32 || small)
    return nm_attention_ex(q, k, v, ldq, ldk, ldv, B, L, S, heads, head_dim, scale, flags & ~NM_ATTN_BF16X3, out, stream);
  if (!workspace) return NM_ERR_WORKSPACE;
  NM_CHECK_ARG(q && k && v && out && B > 0 && L > 0 && S > 0 && heads > 0);
  const int C = heads * head_dim;
  if (ldq < C || ldk < C || ldv < C || (ldq | ldk | ldv) % 4) return NM_ERR_ARG;
  if (B > 65535 || heads > 65535) return NM_ERR_UNSUPPORTED;
  return nm_internal_attn_v2(q, k, v, ldq, ldk, ldv, B, L, S, heads, scale, workspace, out, (hipStream_t)stream, nullptr);
}

extern "C" int nm_attention_ws_lse(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S,
                                   int heads, int head_dim, float scale, int flags, void* workspace, float* out, float* nlse_out,
                                   nmStream_t stream) {
  const bool small = S <= 64 && L <= 64 && (head_dim == 16 || head_dim == 32);
  if (!(flags & NM_ATTN_BF16X3) || head_dim != 32 || small) return NM_ERR_UNSUPPORTED;  // only the split-bf16 kernel keeps the log-sum-exp
  if (!workspace) return NM_ERR_WORKSPACE;
  NM_CHECK_ARG(q && k && v && out && nlse_out && B > 0 && L > 0 && S > 0 && heads > 0);
  const int C = heads * head_dim;
  if (ldq < C || ldk < C || ldv < C || (ldq | ldk | ldv) % 4) return NM_ERR_ARG;
  if (B > 65535 || heads > 65535) return NM_ERR_UNSUPPORTED;
  return nm_internal_attn_v2(q, k, v, ldq, ldk, ldv, B, L, S, heads, scale, workspace, out, (hipStream_t)stream, nlse_out);
}

extern "C" int nm_attention_ld(const float* q, const float* k, const float* v, int ldq, int ldk, int ldv, int B, int L, int S,
                               int heads, int head_dim, float scale, float* out, nmStream_t stream) {
  return nm_attention_ex(q, k, v, ldq, ldk, ldv, B, L, S, heads, head_dim, scale, 0, out, stream);
}

extern "C" int nm_attention(const float* q, const float* k, const float* v, int B, int L, int S, int heads, int head_dim,
                            float scale, float* out, nmStream_t stream) {
  const int C = heads * head_dim;
  return nm_attention_ex(q, k, v, C, C, C, B, L, S, heads, head_dim, scale, 0, out, stream);
}
