// Small data-movement / encoding kernels of the matcher.
// All are HBM- or latency-bound; they exist so that one localisation step stays on the device and to avoid the
// reference's 46 MB F.unfold of the fine feature map (only the K matched 5x5 windows are gathered).
#include "common.h"

namespace {

// tokens[b][iy*w+ix][c] = cfeat[b][c][iy][ix] (+ pe[c][iy][ix]);  32x32 LDS transpose tiles, both sides coalesced.
__global__ void __launch_bounds__(256) nchw_to_tokens_kernel(const float* __restrict__ x, const float* __restrict__ pe, int C, int h,
                                                              int w, int table_h, int table_w, float* __restrict__ y) {
  __shared__ float tile[32][33];
  const int M = h * w;
  const int b = blockIdx.z, m0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, m = m0 + tx;
    float v = 0.f;
    if (c < C && m < M) {
      v = x[((size_t)b * C + c) * M + m];
      if (pe) v += pe[((size_t)c * table_h + m / w) * table_w + m % w];
    }
    tile[k][tx] = v;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int m = m0 + k, c = c0 + tx;
    if (c < C && m < M) y[((size_t)b * M + m) * C + c] = tile[tx][k];
  }
}

// out[n][0:C] = feat, out[n][C:C+3] = x, then per frequency f: sin(2^f x) (3), cos(2^f x) (3); zero padded to ld.
__global__ void __launch_bounds__(256) cat_fourier_kernel(const float* __restrict__ feat, const float* __restrict__ pt3d, int n, int C,
                                                           int num_freqs, int ld, float* __restrict__ out) {
  const int row = blockIdx.x;
  const int emb = 3 + 6 * num_freqs;
  for (int c = threadIdx.x; c < ld; c += blockDim.x) {
    float v = 0.f;
    if (c < C) v = feat[(size_t)row * C + c];
    else if (c < C + 3) v = pt3d[(size_t)row * 3 + (c - C)];
    else if (c < C + emb) {
      const int e = c - C - 3, f = e / 6, which = (e % 6) / 3, ax = e % 3;
      const float arg = (float)(1 << f) * pt3d[(size_t)row * 3 + ax] * 1.0f;
      v = which ? nm_cosf(arg) : nm_sinf(arg);
    }
    out[(size_t)row * ld + c] = v;
  }
}

// backward of the Fourier columns w.r.t. the point: thread per (row, axis)
//   g_x = dy[C + ax] + sum_f 2^f (dy[sin_f, ax] cos(2^f x) - dy[cos_f, ax] sin(2^f x))
__global__ void cat_fourier_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pt3d, int n, int C, int num_freqs, int ld,
                                       float* __restrict__ g_pt3d) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * 3) return;
  const int row = idx / 3, ax = idx % 3;
  const float x = pt3d[idx];
  const float* d = dy + (size_t)row * ld + C;
  float g = d[ax];
  for (int f = 0; f < num_freqs; ++f) {
    const float sc = (float)(1 << f);
    const float arg = sc * x * 1.0f;
    g += sc * (d[3 + f * 6 + ax] * nm_cosf(arg) - d[3 + f * 6 + 3 + ax] * nm_sinf(arg));
  }
  g_pt3d[idx] = g;
}

// feature_normalization of the coarse model's `pt_feat_norm` option (nerfmatch_coarse_trainer.py:42-47), one workgroup of 1024 threads per
// set b: centroid = mean over the N rows (summed in fp64, rounded once), x -= centroid IN PLACE (the reference's `x -= ...` changes the caller's
// tensor too), y = x / max_r ||x_r||.  An unshipped option: clarity over speed (three passes over the set by one workgroup).
__global__ void __launch_bounds__(1024) feature_normalize_kernel(float* __restrict__ x, int N, int D, float* __restrict__ y) {
  __shared__ double part[1024];
  __shared__ float cen[1024];
  __shared__ float wmax[16];
  float* xb = x + (size_t)blockIdx.x * N * D;
  float* yb = y + (size_t)blockIdx.x * N * D;
  const int tid = threadIdx.x, G = 1024 / D, g = tid / D, c = tid % D;
  double acc = 0.0;
  if (g < G)
    for (int r = g; r < N; r += G) acc += (double)xb[(size_t)r * D + c];
  part[tid] = acc;
  __syncthreads();
  if (tid < D) {
    double s = 0.0;
    for (int k = 0; k < G; ++k) s += part[k * D + tid];
    cen[tid] = (float)(s / (double)N);
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  float m = 0.f;
  for (int r = wave; r < N; r += 16) {
    float sq = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float v = xb[(size_t)r * D + k] - cen[k];
      xb[(size_t)r * D + k] = v;
      sq = NM_FMA(v, v, sq);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    m = fmaxf(m, sqrtf(sq));
  }
  if (lane == 0) wmax[wave] = m;
  __syncthreads();
  float mx = wmax[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) mx = fmaxf(mx, wmax[k]);
  const size_t total = (size_t)N * D;
  for (size_t i = tid; i < total; i += 1024) yb[i] = xb[i] / mx;
}

// block per match k (< *count); thread = channel
__global__ void fine_windows_kernel(const float* __restrict__ ffeat, int C, int Hf, int Wf, const int64_t* __restrict__ i_ids,
                                    const int* __restrict__ count, int win, int stride, float* __restrict__ out) {
  const int k = blockIdx.x;
  if (k >= *count) return;
  const int cells_w = (Wf + 2 * (win / 2) - win) / stride + 1;
  const int cell = (int)i_ids[k];
  const int cy = cell / cells_w, cx = cell % cells_w;
  const int y0 = cy * stride - win / 2, x0 = cx * stride - win / 2;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    for (int wy = 0; wy < win; ++wy)
      for (int wx = 0; wx < win; ++wx) {
        const int y = y0 + wy, x = x0 + wx;
        float v = 0.f;
        if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = ffeat[((size_t)c * Hf + y) * Wf + x];
        out[((size_t)k * win * win + wy * win + wx) * C + c] = v;
      }
}

// batched form: match k takes its window from map map_ids[k] of ffeat[B][C][Hf][Wf]
__global__ void fine_windows_batch_kernel(const float* __restrict__ ffeat, int C, int Hf, int Wf, const int64_t* __restrict__ map_ids,
                                          const int64_t* __restrict__ i_ids, const int* __restrict__ count, int win, int stride,
                                          float* __restrict__ out) {
  const int k = blockIdx.x;
  if (k >= *count) return;
  const float* fm = ffeat + (size_t)map_ids[k] * C * Hf * Wf;
  const int cells_w = (Wf + 2 * (win / 2) - win) / stride + 1;
  const int cell = (int)i_ids[k];
  const int cy = cell / cells_w, cx = cell % cells_w;
  const int y0 = cy * stride - win / 2, x0 = cx * stride - win / 2;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    for (int wy = 0; wy < win; ++wy)
      for (int wx = 0; wx < win; ++wx) {
        const int y = y0 + wy, x = x0 + wx;
        float v = 0.f;
        if (y >= 0 && y < Hf && x >= 0 && x < Wf) v = fm[((size_t)c * Hf + y) * Wf + x];
        out[((size_t)k * win * win + wy * win + wx) * C + c] = v;
      }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ ids, const int* __restrict__ count, int dim,
                                   float* __restrict__ out) {
  const int k = blockIdx.x;
  if (k >= *count) return;
  const int64_t r = ids[k];
  for (int c = threadIdx.x; c < dim; c += blockDim.x) out[(size_t)k * dim + c] = src[(size_t)r * dim + c];
}

// Point side of the fine stage as ONE kernel (round 5): out[k] = W1 (W0 src[ids[k]] + b0) + b1 -- `pt_ffeat_proj`, two Linear layers without an
// activation between them, applied to the matched points' coarse tokens (nerfmatch_c2f_trainer.py:344-346).  A few hundred rows per query: as
// gather + two GEMM launches this was 43 us of launch and pipeline latency for 20 MFLOP.  Here a workgroup takes FPP_MP matches: their source rows
// sit in LDS, thread (c, half) owns output column c of FPP_MP / 2 matches and walks K with fp32 FMAs (weights TRANSPOSED, [K][C1]: the 128 threads
// of a half read one row of 512 B per step, from L2 -- every workgroup reads the same 192 KiB).  fp32 throughout, summation in K order.
// Slots k >= *count are written as zeros.
constexpr int FPP_MP = 4, FPP_C1 = 128, FPP_C0_MAX = 512;
__global__ void __launch_bounds__(256) fine_pt_proj_kernel(const float* __restrict__ src, const int64_t* __restrict__ ids, const int* __restrict__ count,
                                                            int max_k, int C0, const float* __restrict__ w0t, const float* __restrict__ b0,
                                                            const float* __restrict__ w1t, const float* __restrict__ b1, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float x[FPP_MP * FPP_C0_MAX];
  __shared__ __attribute__((aligned(16))) float h[FPP_MP * FPP_C1];
  const int tid = threadIdx.x, c = tid & (FPP_C1 - 1), half = tid >> 7;
  const int k0 = blockIdx.x * FPP_MP;
  const int n = min(*count, max_k);
  if (k0 >= n) {  // nothing valid in this group
    for (int i = tid; i < FPP_MP * FPP_C1; i += 256)
      if (k0 + i / FPP_C1 < max_k) out[(size_t)k0 * FPP_C1 + i] = 0.f;
    return;
  }
  for (int m = 0; m < FPP_MP; ++m) {
    const int k = k0 + m;
    const float* row = src + (size_t)(k < n ? ids[k] : ids[k0]) * C0;  // (a slot behind the count computes on the group's first row; zeroed below)
    for (int i = tid; i < C0; i += 256) x[m * C0 + i] = row[i];
  }
  __syncthreads();
  constexpr int MH = FPP_MP / 2;
  float acc[MH];
  {
    const float b = b0 ? b0[c] : 0.f;
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = b;
    const float* xs = x + half * MH * C0;
    for (int k = 0; k < C0; k += 4) {
      const float w0 = w0t[(size_t)k * FPP_C1 + c], w1 = w0t[(size_t)(k + 1) * FPP_C1 + c], w2 = w0t[(size_t)(k + 2) * FPP_C1 + c],
                  w3 = w0t[(size_t)(k + 3) * FPP_C1 + c];
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + m * C0 + k);
        acc[m] = NM_FMA(xv[3], w3, NM_FMA(xv[2], w2, NM_FMA(xv[1], w1, NM_FMA(xv[0], w0, acc[m]))));
      }
    }
#pragma unroll
    for (int m = 0; m < MH; ++m) h[(half * MH + m) * FPP_C1 + c] = acc[m];
  }
  __syncthreads();
  {
    const float b = b1 ? b1[c] : 0.f;
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = b;
    const float* hs = h + half * MH * FPP_C1;
    for (int k = 0; k < FPP_C1; k += 4) {
      const float w0 = w1t[(size_t)k * FPP_C1 + c], w1 = w1t[(size_t)(k + 1) * FPP_C1 + c], w2 = w1t[(size_t)(k + 2) * FPP_C1 + c],
                  w3 = w1t[(size_t)(k + 3) * FPP_C1 + c];
#pragma unroll
      for (int m = 0; m < MH; ++m) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hs + m * FPP_C1 + k);
        acc[m] = NM_FMA(hv[3], w3, NM_FMA(hv[2], w2, NM_FMA(hv[1], w1, NM_FMA(hv[0], w0, acc[m]))));
      }
    }
#pragma unroll
    for (int m = 0; m < MH; ++m) {
      const int k = k0 + half * MH + m;
      if (k < max_k) out[(size_t)k * FPP_C1 + c] = k < n ? acc[m] : 0.f;
    }
  }
}

// Match assembly of the c2f forward (nerfmatch_c2f_trainer.py:457-483) for one image / point-set pair: one thread per match slot k
//   mpt2d_c = pt2d[i_ids[k]],  mpt3d = pt3d[j_ids[k]],  mpt2d_f = mpt2d_c + expec_f[k, :2] * win / 2 * fine_ds,  pred_mask = mconf[k] != 0
// (slots k >= *count hold index 0 -- the lists are zero-initialised -- and are computed like the others: the caller slices)
__global__ void __launch_bounds__(256) assemble_matches_kernel(const float* __restrict__ pt2d, const float* __restrict__ pt3d,
                                                                const int64_t* __restrict__ i_ids, const int64_t* __restrict__ j_ids,
                                                                const float* __restrict__ expec, const float* __restrict__ mconf, int K, float win, float fine_ds,
                                                                float* __restrict__ mpt2d_c, float* __restrict__ mpt2d_f, float* __restrict__ mpt3d,
                                                                uint8_t* __restrict__ pred_mask) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  const int64_t i = i_ids[k], j = j_ids[k];
  const float cx = pt2d[2 * i], cy = pt2d[2 * i + 1];
  mpt2d_c[2 * k] = cx; mpt2d_c[2 * k + 1] = cy;
  // (the reference's expression op for op: expec_f[:, :2] * W / 2 * fine_ds, every intermediate rounded to fp32; -ffp-contract=off)
  const float ex = ((expec[3 * k] * win) / 2.0f) * fine_ds, ey = ((expec[3 * k + 1] * win) / 2.0f) * fine_ds;
  mpt2d_f[2 * k] = cx + ex;
  mpt2d_f[2 * k + 1] = cy + ey;
  mpt3d[3 * k] = pt3d[3 * j]; mpt3d[3 * k + 1] = pt3d[3 * j + 1]; mpt3d[3 * k + 2] = pt3d[3 * j + 2];
  pred_mask[k] = mconf[k] != 0.f ? 1 : 0;
}

// one wavefront per match: lanes r < win*win hold the correlation with window position r
__global__ void __launch_bounds__(256) fine_expectation_kernel(const float* __restrict__ pt_f, const float* __restrict__ win_f,
                                                                const int* __restrict__ count, int max_k, int win, int C, float* __restrict__ expec) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  // (k < max_k as well: the speculative single-pair path launches on FEWER slots than *count may turn out to be, and the grid is rounded up to 4)
  if (k >= max_k || k >= *count) return;
  const int ww = win * win;
  float sim = -__builtin_inff();
  if (lane < ww) {
    const float* a = pt_f + (size_t)k * C;
    const float* b = win_f + ((size_t)k * ww + lane) * C;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot = NM_FMA(a[c], b[c], dot);
    sim = dot * (1.0f / sqrtf((float)C));
  }
  float mx = sim;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float p = lane < ww ? expf(sim - mx) : 0.f;
  float s = p;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  p = p / s;
  // normalised grid: linspace(-1, 1, win) along x (fast) and y
  const int gy_i = lane / win, gx_i = lane % win;
  const float stepg = 2.0f / (float)(win - 1);
  auto lin = [&](int i) -> float { return (i < win / 2) ? -1.0f + stepg * (float)i : 1.0f - stepg * (float)(win - 1 - i); };
  const float gx = lane < ww ? lin(gx_i) : 0.f, gy = lane < ww ? lin(gy_i) : 0.f;
  float ex = gx * p, ey = gy * p, exx = gx * gx * p, eyy = gy * gy * p;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ex += __shfl_xor(ex, o, 64);
    ey += __shfl_xor(ey, o, 64);
    exx += __shfl_xor(exx, o, 64);
    eyy += __shfl_xor(eyy, o, 64);
  }
  if (lane == 0) {
    const float vx = fmaxf(exx - ex * ex, 1e-10f), vy = fmaxf(eyy - ey * ey, 1e-10f);
    expec[(size_t)k * 3 + 0] = ex;
    expec[(size_t)k * 3 + 1] = ey;
    expec[(size_t)k * 3 + 2] = sqrtf(vx) + sqrtf(vy);
  }
}

// ---- training: backward of the two fine-stage kernels above ---------------------------------------------------------
// scatter-add of window gradients into the fine feature map (zeroed / accumulated onto by the caller); float atomics
__global__ void fine_windows_bwd_kernel(const float* __restrict__ dwin, int C, int Hf, int Wf, const int64_t* __restrict__ i_ids,
                                        const int* __restrict__ count, int win, int stride, float* __restrict__ dffeat) {
  const int k = blockIdx.x;
  if (k >= *count) return;
  const int cells_w = (Wf + 2 * (win / 2) - win) / stride + 1;
  const int cell = (int)i_ids[k];
  const int cy = cell / cells_w, cx = cell % cells_w;
  const int y0 = cy * stride - win / 2, x0 = cx * stride - win / 2;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    for (int wy = 0; wy < win; ++wy)
      for (int wx = 0; wx < win; ++wx) {
        const int y = y0 + wy, x = x0 + wx;
        if (y >= 0 && y < Hf && x >= 0 && x < Wf)
          atomicAdd(dffeat + ((size_t)c * Hf + y) * Wf + x, dwin[((size_t)k * win * win + wy * win + wx) * C + c]);
      }
}

// one wavefront per match; d_expec[k] = gradients of (E[x], E[y], std)
__global__ void __launch_bounds__(256) fine_expectation_bwd_kernel(const float* __restrict__ pt_f, const float* __restrict__ win_f,
                                                                    const float* __restrict__ d_expec, const int* __restrict__ count, int max_k,
                                                                    int win, int C, float* __restrict__ d_pt, float* __restrict__ d_win) {
  __shared__ float sds[4][64];
  const int w = threadIdx.x >> 6, k = blockIdx.x * 4 + w, lane = threadIdx.x & 63;
  if (k >= max_k || k >= *count) return;  // whole wavefronts leave together; no block-wide barrier below
  const int ww = win * win;
  const float inv = 1.0f / sqrtf((float)C);
  const float* a = pt_f + (size_t)k * C;
  float sim = -__builtin_inff();
  if (lane < ww) {
    const float* b = win_f + ((size_t)k * ww + lane) * C;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot = NM_FMA(a[c], b[c], dot);
    sim = dot * inv;
  }
  float mx = sim;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float p = lane < ww ? expf(sim - mx) : 0.f;
  float s = p;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  p = p / s;
  const int gy_i = lane / win, gx_i = lane % win;
  const float stepg = 2.0f / (float)(win - 1);
  auto lin = [&](int i) -> float { return (i < win / 2) ? -1.0f + stepg * (float)i : 1.0f - stepg * (float)(win - 1 - i); };
  const float gx = lane < ww ? lin(gx_i) : 0.f, gy = lane < ww ? lin(gy_i) : 0.f;
  float ex = gx * p, ey = gy * p, exx = gx * gx * p, eyy = gy * gy * p;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ex += __shfl_xor(ex, o, 64);
    ey += __shfl_xor(ey, o, 64);
    exx += __shfl_xor(exx, o, 64);
    eyy += __shfl_xor(eyy, o, 64);
  }
  const float vx = exx - ex * ex, vy = eyy - ey * ey;
  const float gex = d_expec[(size_t)k * 3 + 0], gey = d_expec[(size_t)k * 3 + 1], gsd = d_expec[(size_t)k * 3 + 2];
  const float dvx = vx >= 1e-10f ? gsd * 0.5f / sqrtf(vx) : 0.f, dvy = vy >= 1e-10f ? gsd * 0.5f / sqrtf(vy) : 0.f;
  // d loss / d heat[r]
  const float dh = gex * gx + gey * gy + dvx * (gx * gx - 2.0f * ex * gx) + dvy * (gy * gy - 2.0f * ey * gy);
  float pd = p * dh;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pd += __shfl_xor(pd, o, 64);
  sds[w][lane] = lane < ww ? p * (dh - pd) * inv : 0.f;  // d loss / d <pt, win_r>
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < C; c += 64) {
    const float pc = a[c];
    float acc = 0.f;
    for (int r = 0; r < ww; ++r) {
      const float d = sds[w][r];
      acc = NM_FMA(d, win_f[((size_t)k * ww + r) * C + c], acc);
      d_win[((size_t)k * ww + r) * C + c] = d * pc;
    }
    d_pt[(size_t)k * C + c] = acc;
  }
}

}  // namespace

extern "C" int nm_add_sine_pe(const float* x, const float* pe_table, int B, int h, int w, int C, int table_h, int table_w, float* y,
                              nmStream_t stream) {
  NM_CHECK_ARG(x && y && B > 0 && h > 0 && w > 0 && C > 0);
  if (pe_table && (h > table_h || w > table_w)) return NM_ERR_ARG;
  dim3 grid((h * w + 31) / 32, (C + 31) / 32, B);
  nchw_to_tokens_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, pe_table, C, h, w, table_h, table_w, y);
  return nm_launch_status();
}

extern "C" int nm_cat_fourier_bwd(const float* dy, const float* pt3d, int n, int C, int num_freqs, float* g_pt3d, nmStream_t stream) {
  NM_CHECK_ARG(dy && pt3d && g_pt3d && n > 0 && C >= 0 && num_freqs >= 0 && num_freqs < 31);
  const int ld = ((C + 3 + 6 * num_freqs + 7) / 8) * 8;
  cat_fourier_bwd_kernel<<<(n * 3 + 255) / 256, 256, 0, (hipStream_t)stream>>>(dy, pt3d, n, C, num_freqs, ld, g_pt3d);
  return nm_launch_status();
}

extern "C" int nm_cat_fourier(const float* feat, const float* pt3d, int n, int C, int num_freqs, float* out, nmStream_t stream) {
  NM_CHECK_ARG(feat && pt3d && out && n > 0 && C > 0 && num_freqs > 0 && num_freqs <= 30);
  const int ld = ((C + 3 + 6 * num_freqs + 7) / 8) * 8;  // row length padded to a multiple of 8 floats for nm_linear
  cat_fourier_kernel<<<n, 256, 0, (hipStream_t)stream>>>(feat, pt3d, n, C, num_freqs, ld, out);
  return nm_launch_status();
}

extern "C" int nm_feature_normalize(float* x, int B, int N, int D, float* y, nmStream_t stream) {
  NM_CHECK_ARG(x && y && B > 0 && N > 0 && D > 0);
  if (D > 1024) return NM_ERR_UNSUPPORTED;
  feature_normalize_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(x, N, D, y);
  return nm_launch_status();
}

extern "C" int nm_fine_windows(const float* ffeat, int C, int Hf, int Wf, const int64_t* i_ids, const int* count, int max_k, int win,
                               int stride, float* out, nmStream_t stream) {
  NM_CHECK_ARG(ffeat && i_ids && count && out && C > 0 && Hf > 0 && Wf > 0 && win > 0 && stride > 0);
  if (max_k <= 0) return NM_OK;
  fine_windows_kernel<<<max_k, 128, 0, (hipStream_t)stream>>>(ffeat, C, Hf, Wf, i_ids, count, win, stride, out);
  return nm_launch_status();
}

extern "C" int nm_fine_windows_batch(const float* ffeat, int B, int C, int Hf, int Wf, const int64_t* map_ids, const int64_t* i_ids,
                                     const int* count, int max_k, int win, int stride, float* out, nmStream_t stream) {
  NM_CHECK_ARG(ffeat && map_ids && i_ids && count && out && B > 0 && C > 0 && Hf > 0 && Wf > 0 && win > 0 && stride > 0);
  if (max_k <= 0) return NM_OK;
  fine_windows_batch_kernel<<<max_k, 128, 0, (hipStream_t)stream>>>(ffeat, C, Hf, Wf, map_ids, i_ids, count, win, stride, out);
  return nm_launch_status();
}

extern "C" int nm_gather_rows(const float* src, const int64_t* ids, const int* count, int max_k, int dim, float* out, nmStream_t stream) {
  NM_CHECK_ARG(src && ids && count && out && dim > 0);
  if (max_k <= 0) return NM_OK;
  gather_rows_kernel<<<max_k, 128, 0, (hipStream_t)stream>>>(src, ids, count, dim, out);
  return nm_launch_status();
}

extern "C" int nm_fine_pt_proj(const float* src, const int64_t* ids, const int* count, int max_k, int C0, int C1, const float* w0t, const float* b0,
                               const float* w1t, const float* b1, float* out, nmStream_t stream) {
  NM_CHECK_ARG(src && ids && count && w0t && w1t && out && C0 > 0);
  if (C1 != FPP_C1 || C0 % 4 || C0 > FPP_C0_MAX) return NM_ERR_UNSUPPORTED;
  if (max_k <= 0) return NM_OK;
  fine_pt_proj_kernel<<<(max_k + FPP_MP - 1) / FPP_MP, 256, 0, (hipStream_t)stream>>>(src, ids, count, max_k, C0, w0t, b0, w1t, b1, out);
  return nm_launch_status();
}

extern "C" int nm_assemble_matches(const float* pt2d, const float* pt3d, const int64_t* i_ids, const int64_t* j_ids, const float* expec_f,
                                   const float* mconf, int K, float win, float fine_ds, float* mpt2d_c, float* mpt2d_f, float* mpt3d,
                                   uint8_t* pred_mask, nmStream_t stream) {
  NM_CHECK_ARG(pt2d && pt3d && i_ids && j_ids && expec_f && mconf && mpt2d_c && mpt2d_f && mpt3d && pred_mask);
  if (K <= 0) return NM_OK;
  assemble_matches_kernel<<<(K + 255) / 256, 256, 0, (hipStream_t)stream>>>(pt2d, pt3d, i_ids, j_ids, expec_f, mconf, K, win, fine_ds, mpt2d_c, mpt2d_f,
                                                                            mpt3d, pred_mask);
  return nm_launch_status();
}

extern "C" int nm_fine_expectation(const float* pt_f, const float* win_f, const int* count, int max_k, int win, int C, float* expec_f,
                                   nmStream_t stream) {
  NM_CHECK_ARG(pt_f && win_f && count && expec_f && win > 1 && win * win <= 64 && C > 0);
  if (max_k <= 0) return NM_OK;
  fine_expectation_kernel<<<(max_k + 3) / 4, 256, 0, (hipStream_t)stream>>>(pt_f, win_f, count, max_k, win, C, expec_f);
  return nm_launch_status();
}

extern "C" int nm_fine_windows_bwd(const float* dwin, int C, int Hf, int Wf, const int64_t* i_ids, const int* count, int max_k, int win,
                                   int stride, float* dffeat, nmStream_t stream) {
  NM_CHECK_ARG(dwin && i_ids && count && dffeat && C > 0 && Hf > 0 && Wf > 0 && win > 0 && stride > 0);
  if (max_k <= 0) return NM_OK;
  fine_windows_bwd_kernel<<<max_k, 128, 0, (hipStream_t)stream>>>(dwin, C, Hf, Wf, i_ids, count, win, stride, dffeat);
  return nm_launch_status();
}

extern "C" int nm_fine_expectation_bwd(const float* pt_f, const float* win_f, const float* d_expec, const int* count, int max_k, int win,
                                       int C, float* d_pt, float* d_win, nmStream_t stream) {
  NM_CHECK_ARG(pt_f && win_f && d_expec && count && d_pt && d_win && win > 1 && win * win <= 64 && C > 0);
  if (max_k <= 0) return NM_OK;
  fine_expectation_bwd_kernel<<<(max_k + 3) / 4, 256, 0, (hipStream_t)stream>>>(pt_f, win_f, d_expec, count, max_k, win, C, d_pt, d_win);
  return nm_launch_status();
}
