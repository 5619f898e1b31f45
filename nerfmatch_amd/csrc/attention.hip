// LayerNorm and softmax multi-head attention.
//
// attn32_kernel: flash-style attention for head dim 32 on the fp32 matrix cores; the (L,S,H) score tensor the
// reference materialises (415 MB at 3600^2 x 8 heads) never exists.  One wavefront owns 32 queries of one head.
//   S^T tile  = K_tile[32 keys x 32] . Q^T[32 x 32 queries]      (A = keys, B = queries; computing the TRANSPOSED
//               scores puts all keys of a query in one lane pair: row max / sum need one cross-half shuffle)
//   O^T      += V_tile^T[32 d x 32 keys] . P^T[32 keys x 32 queries]   (B = the probabilities exactly as the
//               first MFMA left them in registers: keys nrow(r, half) <-> k-step r; no LDS, no re-layout)
// K / V tiles are read straight from L2 (a head's K and V are 2 x 614 KB at 4800 tokens).
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
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, int rows, float eps, float* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
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

__global__ void __launch_bounds__(256) attn32_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                      const float* __restrict__ v, int L, int S, int H, float scale,
                                                      float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hi = lane >> 5;
  const int qt = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
  if (qt * 32 >= L) return;
  const int C = H * 32;
  const int qrow = qt * 32 + j;
  const int qc = qrow < L ? qrow : L - 1;
  float qreg[16];
  {
    const float* qp = q + ((size_t)b * L + qc) * C + h * 32 + 4 * hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(qp + 8 * c);
#pragma unroll
      for (int t = 0; t < 4; ++t) qreg[4 * c + t] = t4[t] * scale;
    }
  }
  f32x16 o;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = 0.f;
  float mrun = -__builtin_inff(), lrun = 0.f;
  const float* kbase = k + (size_t)b * S * C + h * 32;
  const float* vbase = v + (size_t)b * S * C + h * 32;
  for (int s0 = 0; s0 < S; s0 += 32) {
    const int krow = s0 + j < S ? s0 + j : S - 1;
    const float* kp = kbase + (size_t)krow * C + 4 * hi;
    f32x4 ka[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) ka[c] = *reinterpret_cast<const f32x4*>(kp + 8 * c);
    // V operands of this tile (issued early; consumed after the softmax)
    float va[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = s0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      va[r] = vbase[(size_t)(key < S ? key : S - 1) * C + j];
    }
    f32x16 sc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) sc = MFMA32(ka[c][t], qreg[4 * c + t], sc);
    float mx = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = s0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= S) sc[r] = -__builtin_inff();
      mx = fmaxf(mx, sc[r]);
    }
    mx = fmaxf(mx, nm_shfl_xor32(mx));
    const float mnew = fmaxf(mrun, mx);
    const float alpha = expf(mrun - mnew);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[r] = expf(sc[r] - mnew);
      ps += sc[r];
    }
    ps += nm_shfl_xor32(ps);
    lrun = lrun * alpha + ps;
    mrun = mnew;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) o = MFMA32(va[r], sc[r], o);
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
                                                         const float* __restrict__ v, int L, int S, int H, float scale,
                                                         float* __restrict__ out) {
  __shared__ float sk[64 * D], sv[64 * D];
  const int h = blockIdx.x, b = blockIdx.y, i = threadIdx.x;
  const int C = H * D;
  for (int e = i; e < S * D; e += 64) {
    const int row = e / D, d = e % D;
    sk[e] = k[((size_t)b * S + row) * C + h * D + d];
    sv[e] = v[((size_t)b * S + row) * C + h * D + d];
  }
  __syncthreads();
  if (i >= L) return;
  float qr[D], o[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    qr[d] = q[((size_t)b * L + i) * C + h * D + d] * scale;
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

extern "C" int nm_attention(const float* q, const float* k, const float* v, int B, int L, int S, int heads, int head_dim,
                            float scale, float* out, nmStream_t stream) {
  NM_CHECK_ARG(q && k && v && out && B > 0 && L > 0 && S > 0 && heads > 0);
  hipStream_t s = (hipStream_t)stream;
  if (S <= 64 && L <= 64 && (head_dim == 16 || head_dim == 32)) {
    dim3 grid(heads, B);
    if (head_dim == 16) attn_small_kernel<16><<<grid, 64, 0, s>>>(q, k, v, L, S, heads, scale, out);
    else attn_small_kernel<32><<<grid, 64, 0, s>>>(q, k, v, L, S, heads, scale, out);
    return nm_launch_status();
  }
  if (head_dim != 32) return NM_ERR_UNSUPPORTED;
  if (B > 65535 || heads > 65535) return NM_ERR_UNSUPPORTED;
  dim3 grid(((L + 31) / 32 + 3) / 4, heads, B);
  attn32_kernel<<<grid, 256, 0, s>>>(q, k, v, L, S, heads, scale, out);
  return nm_launch_status();
}
