// y[M,N] = act(x[M,K] . w[N,K]^T + bias) + residual  on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Both operands are row-major with K innermost, which is exactly what the MFMA wants when the two K indices of a
// k-step are taken 4 apart: lane (row r, half h) loads ONE float4 = x[r][8c + 4h .. 8c + 4h + 3] and uses component t
// as its operand of k-step (c, t) -- lanes 0-31 then hold k = 8c + t, lanes 32-63 hold k = 8c + 4 + t.  A wavefront
// owns a 32-row x (32*NB)-column output tile: tokens on the A side (rows of D), output features on the B side
// (columns of D = lanes), so every store instruction writes 128 contiguous bytes of one output row.
// The same kernel (MODE_SIM) produces the dual-softmax similarity matrix with its scale / mask epilogue.
#include "common.h"

namespace {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

enum { MODE_LINEAR = 0, MODE_SIM = 1 };

struct GemmArgs {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  const float* pre;   // added before the activation (or NULL)
  const float* gate;  // output multiplied by [gate > 0] (or NULL): the ReLU derivative of a saved activation
  float* y;
  int M, N, K, act;
  // MODE_SIM
  float scale;
  const uint8_t* row_mask;
  const uint8_t* col_mask;
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int NB, int MODE>
__global__ void __launch_bounds__(256) gemm_kernel(GemmArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hi = lane >> 5;
  const int rt = blockIdx.x * 4 + wave;
  if (rt * 32 >= a.M) return;
  const int col0 = blockIdx.y * (32 * NB);
  const int m = rt * 32 + r;
  const int mc = m < a.M ? m : a.M - 1;
  const int K = a.K;
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int cb = col0 + 32 * nb + r;
    const float b = (MODE == MODE_LINEAR && a.bias && cb < a.N) ? a.bias[cb] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nb][i] = b;
  }
  const float* xp = a.x + (size_t)mc * K + 4 * hi;
  const float* wp[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int cw = col0 + 32 * nb + r;  // column tail: clamp the weight row, the store is masked
    wp[nb] = a.w + (size_t)(cw < a.N ? cw : a.N - 1) * K + 4 * hi;
  }
  const int nc = K / 8;
  f32x4 xa = *reinterpret_cast<const f32x4*>(xp);
  f32x4 wb[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) wb[nb] = *reinterpret_cast<const f32x4*>(wp[nb]);
  for (int c = 0; c < nc; ++c) {
    const int cn = (c + 1 < nc) ? c + 1 : c;  // software prefetch of the next 8-wide K slice
    const f32x4 xn = *reinterpret_cast<const f32x4*>(xp + 8 * cn);
    f32x4 wn[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wn[nb] = *reinterpret_cast<const f32x4*>(wp[nb] + 8 * cn);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[nb] = MFMA32(xa[t], wb[nb][t], acc[nb]);
    xa = xn;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wb[nb] = wn[nb];
  }
  // epilogue: register i of lane (col r, half hi) is output row rt*32 + (i&3) + 8*(i>>2) + 4*hi
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = col0 + 32 * nb + r;
    if (col >= a.N) continue;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
      if (row < a.M) {
        float v = acc[nb][i];
        if (MODE == MODE_LINEAR) {
          if (a.pre) v += a.pre[(size_t)row * a.N + col];
          if (a.act == NM_ACT_RELU) v = fmaxf(v, 0.f);
          else if (a.act == NM_ACT_GELU) v = gelu_erf(v);
          if (a.res) v += a.res[(size_t)row * a.N + col];
          if (a.gate && !(a.gate[(size_t)row * a.N + col] > 0.f)) v = 0.f;
        } else {
          v = v * a.scale;
          const bool keep = (a.row_mask ? a.row_mask[row] != 0 : true) && (a.col_mask ? a.col_mask[col] != 0 : true);
          if (!keep) v = -1e9f;
        }
        a.y[(size_t)row * a.N + col] = v;
      }
    }
  }
}

template <int MODE>
int launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.K % 8 != 0) return NM_ERR_UNSUPPORTED;  // rows must be 32-byte sliceable (callers pad K, e.g. 349 -> 352)
  const int row_tiles = (a.M + 31) / 32;
  dim3 block(256);
  if (a.N > 32) {
    dim3 grid((row_tiles + 3) / 4, (a.N + 63) / 64);
    gemm_kernel<2, MODE><<<grid, block, 0, s>>>(a);
  } else {
    dim3 grid((row_tiles + 3) / 4, 1);
    gemm_kernel<1, MODE><<<grid, block, 0, s>>>(a);
  }
  return nm_launch_status();
}

}  // namespace

extern "C" int nm_linear_ex(const float* x, const float* w, const float* bias, const float* pre, const float* residual,
                            const float* gate, int M, int N, int K, int act, float* y, nmStream_t stream) {
  NM_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0);
  if (act < NM_ACT_NONE || act > NM_ACT_GELU) return NM_ERR_ARG;
  GemmArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.res = residual; a.pre = pre; a.gate = gate; a.y = y;
  a.M = M; a.N = N; a.K = K; a.act = act;
  return launch_gemm<MODE_LINEAR>(a, (hipStream_t)stream);
}

extern "C" int nm_linear(const float* x, const float* w, const float* bias, const float* residual, int M, int N, int K,
                         int act, float* y, nmStream_t stream) {
  return nm_linear_ex(x, w, bias, nullptr, residual, nullptr, M, N, K, act, y, stream);
}

// internal (used by match.hip): sim[M,N] = mask_fill(scale * im[M,C] . pt[N,C]^T)
int nm_internal_sim(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                    const uint8_t* pt_mask, float* sim, hipStream_t s) {
  GemmArgs a{};
  a.x = im; a.w = pt; a.y = sim;
  a.M = M; a.N = N; a.K = C;
  a.scale = scale; a.row_mask = im_mask; a.col_mask = pt_mask;
  return launch_gemm<MODE_SIM>(a, s);
}
