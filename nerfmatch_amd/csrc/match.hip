// Dual-softmax matching score + (mutual) nearest-neighbour selection.
//
// Round-1 structure: the similarity matrix is produced by the fp32-MFMA GEMM (gemm.hip, MODE_SIM) into workspace
// (M x N fp32, 92 MB at 4800^2) and then swept by HBM-bound kernels:
//   row_stats   : per image token i   rmax_i, rsum_i = sum_j exp(sim_ij - rmax_i)            (softmax over dim 2)
//   col_stats   : per point j         cmax_j, csum_j                                           (softmax over dim 1)
//   col_confmax : per point j         max_i conf_ij   (needed by the mutual test)
//   row_select  : per image token i   conf row (optionally written out), row max, and the FIRST column with
//                 conf > thr  &&  conf == row max  [&& conf == column max]   -- the reference's mask.max(dim=2)
//   compact     : ordered compaction of the per-row results into (i_ids, j_ids, mconf), count
// conf_ij is always evaluated by the same inlined expression (one v_exp_f32 of the combined exponent times the two
// reciprocal sums), so the equality tests compare bit-identical values exactly like the reference's
// `conf == conf.max()`; against the reference's softmax * softmax the values differ by rounding only (<= 3e-7 relative).
// Algorithmic HBM bytes: 8*M*N when conf is returned (1 write + 1 read); this version moves 5 passes over
// sim, read in 16-byte pieces.
#include "common.h"

int nm_internal_sim(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                    const uint8_t* pt_mask, float* sim, hipStream_t s);
int nm_internal_sim_bf16x3(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                           const uint8_t* pt_mask, float* sim, void* blob, hipStream_t s);

namespace {

constexpr int COL_CHUNKS = 64;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// f / (|f| + 1e-6), one wavefront per row (C = 64 * PER)
template <int PER>
__global__ void __launch_bounds__(256) l2norm_kernel(const float* __restrict__ x, int rows, float* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[PER], q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = x[(size_t)row * 64 * PER + lane + 64 * i];
    q = NM_FMA(v[i], v[i], q);
  }
  const float den = sqrtf(wave_sum(q)) + 1e-6f;
#pragma unroll
  for (int i = 0; i < PER; ++i) y[(size_t)row * 64 * PER + lane + 64 * i] = v[i] / den;
}

// All sweeps evaluate exponentials as v_exp_f32 of a log2(e)-scaled argument (1 ulp) and read the matrix in 16-byte
// pieces when the row pitch allows (VEC = 4: N % 4 == 0), else element-wise (VEC = 1).
constexpr float LOG2E = 1.44269504088896340736f;
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

template <int VEC>
struct Piece {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ Piece<VEC> load_piece(const float* p) {
  Piece<VEC> r;
  if (VEC == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    r.v[0] = t[0]; r.v[1 % VEC] = t[1]; r.v[2 % VEC] = t[2]; r.v[3 % VEC] = t[3];
  } else {
    r.v[0] = *p;
  }
  return r;
}

// one wavefront per row.  NREG > 0: the row (N <= 64 VEC NREG) is held in registers between the two passes -- the second
// read would miss L2 (tens of MB stream through each XCD between a wavefront's two passes); NREG = 0: two reads.
constexpr int ROW_NREG = 20;  // 64 lanes x 4 floats x 20 = 5120 columns
template <int VEC, int NREG>
__global__ void __launch_bounds__(256) row_stats_kernel(const float* __restrict__ sim, int M, int N, float* __restrict__ rmax,
                                                         float* __restrict__ rsum) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* r = sim + (size_t)row * N;
  float mx = -__builtin_inff(), s = 0.f;
  if (NREG > 0) {
    Piece<VEC> c[NREG > 0 ? NREG : 1];
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int j = (lane + 64 * k) * VEC;
      if (j < N) {
        c[k] = load_piece<VEC>(r + j);
#pragma unroll
        for (int e = 0; e < VEC; ++e) mx = fmaxf(mx, c[k].v[e]);
      }
    }
    mx = wave_max(mx);
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int j = (lane + 64 * k) * VEC;
      if (j < N) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += exp_fast(c[k].v[e] - mx);
      }
    }
  } else {
    for (int j = lane * VEC; j < N; j += 64 * VEC) {
      const Piece<VEC> p = load_piece<VEC>(r + j);
#pragma unroll
      for (int e = 0; e < VEC; ++e) mx = fmaxf(mx, p.v[e]);
    }
    mx = wave_max(mx);
    for (int j = lane * VEC; j < N; j += 64 * VEC) {
      const Piece<VEC> p = load_piece<VEC>(r + j);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += exp_fast(p.v[e] - mx);
    }
  }
  s = wave_sum(s);
  if (lane == 0) {
    rmax[row] = mx;
    rsum[row] = 1.0f / s;  // the sweeps multiply by the reciprocal
  }
}

// grid (ceil(N / (64 VEC)), COL_CHUNKS); block 256 = 64 column groups x 4 row lanes; online (max, sum) per column
template <int VEC>
__global__ void __launch_bounds__(256) col_stats_partial_kernel(const float* __restrict__ sim, int M, int N,
                                                                 float* __restrict__ pmax, float* __restrict__ psum) {
  __shared__ float smx[4][64 * VEC], ssm[4][64 * VEC];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + tx) * VEC;
  const int rows_per = (M + COL_CHUNKS - 1) / COL_CHUNKS;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float mx[VEC], s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { mx[e] = -__builtin_inff(); s[e] = 0.f; }
  if (col < N)
    for (int i = r0 + ty; i < r1; i += 4) {
      const Piece<VEC> p = load_piece<VEC>(sim + (size_t)i * N + col);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float v = p.v[e];
        const float m2 = fmaxf(mx[e], v);
        s[e] = s[e] * exp_fast(mx[e] - m2) + exp_fast(v - m2);  // (first element: 0 * exp(-inf) + 1)
        mx[e] = m2;
      }
    }
#pragma unroll
  for (int e = 0; e < VEC; ++e) { smx[ty][tx * VEC + e] = mx[e]; ssm[ty][tx * VEC + e] = s[e]; }
  __syncthreads();
  if (ty == 0 && col < N) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = tx * VEC + e;
      const float m2 = fmaxf(fmaxf(smx[0][c], smx[1][c]), fmaxf(smx[2][c], smx[3][c]));
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (ssm[t][c] > 0.f) s2 += ssm[t][c] * exp_fast(smx[t][c] - m2);
      pmax[blockIdx.y * N + col + e] = m2;
      psum[blockIdx.y * N + col + e] = s2;
    }
  }
}

// grid ceil(N/64), block 256 = 64 columns x 4 chunk lanes
__global__ void __launch_bounds__(256) col_stats_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum, int N,
                                                               float* __restrict__ cmax, float* __restrict__ csum) {
  __shared__ float smx[4][64], ssm[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  float m = -__builtin_inff(), s = 0.f;
  if (col < N) {
    float pm[COL_CHUNKS / 4], ps[COL_CHUNKS / 4];
#pragma unroll
    for (int c = 0; c < COL_CHUNKS / 4; ++c) {
      pm[c] = pmax[(ty + 4 * c) * N + col];
      ps[c] = psum[(ty + 4 * c) * N + col];
      m = fmaxf(m, pm[c]);
    }
#pragma unroll
    for (int c = 0; c < COL_CHUNKS / 4; ++c)
      if (ps[c] > 0.f) s += ps[c] * exp_fast(pm[c] - m);
  }
  smx[ty][tx] = m;
  ssm[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < N) {
    const float m2 = fmaxf(fmaxf(smx[0][tx], smx[1][tx]), fmaxf(smx[2][tx], smx[3][tx]));
    float s2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (ssm[t][tx] > 0.f) s2 += ssm[t][tx] * exp_fast(smx[t][tx] - m2);
    cmax[col] = m2;
    csum[col] = 1.0f / s2;  // reciprocal, see conf_value
  }
}

// softmax(sim, dim=1)[i][j] * softmax(sim, dim=2)[i][j] = exp((v - cmax_j) + (v - rmax_i)) / (csum_j rsum_i); icsm / irsm are
// the reciprocal sums.  ONE inlined expression for every sweep, so the equality tests compare bit-identical values.
__device__ __forceinline__ float conf_value(float v, float cmx, float icsm, float rmx, float irsm) {
  return (exp_fast((v - cmx) + (v - rmx)) * icsm) * irsm;
}

// column max of conf: same blocking as col_stats_partial, merged with atomicMax on the (non-negative) float bits
template <int VEC>
__global__ void __launch_bounds__(256) col_confmax_kernel(const float* __restrict__ sim, int M, int N, const float* __restrict__ rmax,
                                                           const float* __restrict__ rsum, const float* __restrict__ cmax,
                                                           const float* __restrict__ csum, unsigned int* __restrict__ colmax_bits) {
  __shared__ float smx[4][64 * VEC];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + tx) * VEC;
  const int rows_per = (M + COL_CHUNKS - 1) / COL_CHUNKS;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float mx[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) mx[e] = 0.f;
  if (col < N) {
    const Piece<VEC> cm = load_piece<VEC>(cmax + col), cs = load_piece<VEC>(csum + col);
    for (int i = r0 + ty; i < r1; i += 4) {
      const Piece<VEC> p = load_piece<VEC>(sim + (size_t)i * N + col);
      const float rm = rmax[i], rs = rsum[i];
#pragma unroll
      for (int e = 0; e < VEC; ++e) mx[e] = fmaxf(mx[e], conf_value(p.v[e], cm.v[e], cs.v[e], rm, rs));
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) smx[ty][tx * VEC + e] = mx[e];
  __syncthreads();
  if (ty == 0 && col < N) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = tx * VEC + e;
      const float m2 = fmaxf(fmaxf(smx[0][c], smx[1][c]), fmaxf(smx[2][c], smx[3][c]));
      atomicMax(colmax_bits + col + e, __float_as_uint(m2));
    }
  }
}

// one wavefront per row: writes conf (optional), finds the first column passing the reference's mask.  NREG as above: the
// conf values of the row stay in registers between the "row maximum" and the "first column" passes.
template <int VEC, int NREG>
__global__ void __launch_bounds__(256) row_select_kernel(const float* __restrict__ sim, int M, int N, const float* __restrict__ rmax,
                                                          const float* __restrict__ rsum, const float* __restrict__ cmax,
                                                          const float* __restrict__ csum, const unsigned int* __restrict__ colmax_bits,
                                                          float thr, int mutual, float* __restrict__ conf_out,
                                                          int* __restrict__ sel_j, float* __restrict__ sel_conf) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* r = sim + (size_t)row * N;
  const float rm = rmax[row], rs = rsum[row];
  float best = 0.f;
  int first = 0x7fffffff;
  auto conf_piece = [&](int j, float (&c)[VEC]) {
    const Piece<VEC> p = load_piece<VEC>(r + j), cm = load_piece<VEC>(cmax + j), cs = load_piece<VEC>(csum + j);
#pragma unroll
    for (int e = 0; e < VEC; ++e) c[e] = conf_value(p.v[e], cm.v[e], cs.v[e], rm, rs);
  };
  auto emit = [&](int j, const float (&c)[VEC]) {
    if (conf_out) {
      if (VEC == 4) *reinterpret_cast<f32x4*>(conf_out + (size_t)row * N + j) = f32x4{c[0], c[1 % VEC], c[2 % VEC], c[3 % VEC]};
      else conf_out[(size_t)row * N + j] = c[0];
    }
  };
  auto pick = [&](int j, const float (&c)[VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      bool ok = (c[e] > thr) && (c[e] == best);
      if (mutual) ok = ok && (c[e] == __uint_as_float(colmax_bits[j + e]));
      if (ok && j + e < first) first = j + e;
    }
  };
  if (NREG > 0) {
    float cc[NREG > 0 ? NREG : 1][VEC];
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int j = (lane + 64 * k) * VEC;
      if (j < N) {
        conf_piece(j, cc[k]);
        emit(j, cc[k]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) best = fmaxf(best, cc[k][e]);
      }
    }
    best = wave_max(best);
#pragma unroll
    for (int k = 0; k < NREG; ++k) {
      const int j = (lane + 64 * k) * VEC;
      if (j < N) pick(j, cc[k]);
    }
  } else {
    for (int j = lane * VEC; j < N; j += 64 * VEC) {
      float c[VEC];
      conf_piece(j, c);
      emit(j, c);
#pragma unroll
      for (int e = 0; e < VEC; ++e) best = fmaxf(best, c[e]);
    }
    best = wave_max(best);
    for (int j = lane * VEC; j < N; j += 64 * VEC) {
      float c[VEC];
      conf_piece(j, c);
      pick(j, c);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
  if (lane == 0) {
    sel_j[row] = (first == 0x7fffffff) ? -1 : first;
    sel_conf[row] = best;
  }
}

// ordered compaction by a single workgroup (M <= a few 10k): (i, j, conf) of rows with a match, sorted by i
__global__ void __launch_bounds__(1024) compact_kernel(const int* __restrict__ sel_j, const float* __restrict__ sel_conf, int M,
                                                        int64_t* __restrict__ out_i, int64_t* __restrict__ out_j,
                                                        float* __restrict__ out_conf, int* __restrict__ count) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int start = 0; start < M; start += 1024) {
    const int i = start + tid;
    const int has = (i < M && sel_j[i] >= 0) ? 1 : 0;
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
      const int p = base + woff + prefix;
      out_i[p] = i;
      out_j[p] = sel_j[i];
      out_conf[p] = sel_conf[i];
    }
    __syncthreads();
    if (tid == 0) s_base = base + total;
    __syncthreads();
  }
  if (tid == 0) *count = s_base;
}

// ---- training: focal loss on the dual-softmax confidence and its gradient ------------------------------------------
// reference: compute_matching_loss(conf, conf_gt, alpha=0.25, gamma=2.0), utils/metrics.py:372-380:
//     c = clamp(conf, 1e-6, 1 - 1e-6) [clamp=True; the coarse-only model passes clamp=False];  loss = mean_{gt=1} -alpha (1-c)^gamma log c  +  mean_{gt=0} -alpha c^gamma log(1-c)
// (the means run over the whole batch).  acc = {sum_pos, sum_neg, n_pos, n_neg} in double precision.
__device__ __forceinline__ float pow_gamma(float x, float gamma) { return gamma == 2.0f ? x * x : powf(x, gamma); }
__device__ __forceinline__ float pow_gamma_m1(float x, float gamma) { return gamma == 2.0f ? x : powf(x, gamma - 1.0f); }

// loss term of one entry
__device__ __forceinline__ float focal_term(float conf, bool pos, float alpha, float gamma, bool clamp) {
  const float c = clamp ? fminf(fmaxf(conf, 1e-6f), 1.0f - 1e-6f) : conf;
  return pos ? -alpha * pow_gamma(1.0f - c, gamma) * logf(c) : -alpha * pow_gamma(c, gamma) * logf(1.0f - c);
}
// t = conf * d(loss)/d(conf) of one entry (wp = 1/n_pos, wn = 1/n_neg); torch.clamp passes the gradient on [min, max]
__device__ __forceinline__ float focal_t(float conf, bool pos, float alpha, float gamma, float wp, float wn, bool clamp) {
  if (clamp && !(conf >= 1e-6f && conf <= 1.0f - 1e-6f)) return 0.f;
  const float c = conf;
  const float g = pos ? -alpha * wp * (pow_gamma(1.0f - c, gamma) / c - gamma * pow_gamma_m1(1.0f - c, gamma) * logf(c))
                      : -alpha * wn * (gamma * pow_gamma_m1(c, gamma) * logf(1.0f - c) - pow_gamma(c, gamma) / (1.0f - c));
  return conf * g;
}

// 0x80 in every byte of v that is zero (exact: no carries between bytes)
__device__ __forceinline__ unsigned zero_bytes(unsigned v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }

// positives (bytes == 1) and negatives (bytes == 0) of the ground-truth mask.  Round 6: 16 bytes per load and two popcounts per word -- the
// byte-per-thread loop of rounds 1-5 took 0.16 ms for the 23 MB of a 4800 x 4800 mask (every step of the matching term and of training).
__global__ void __launch_bounds__(256) focal_count_kernel(const uint8_t* __restrict__ gt, size_t total, double* __restrict__ acc) {
  unsigned int pos = 0, neg = 0;
  const size_t head = (16 - ((size_t)gt & 15)) & 15;  // bytes in front of the first 16-byte boundary
  const size_t h = head < total ? head : total;
  const size_t nvec = (total - h) / 16;
  const uint4* v4 = reinterpret_cast<const uint4*>(gt + h);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    const uint4 w = v4[i];
    neg += __popc(zero_bytes(w.x)) + __popc(zero_bytes(w.y)) + __popc(zero_bytes(w.z)) + __popc(zero_bytes(w.w));
    pos += __popc(zero_bytes(w.x ^ 0x01010101u)) + __popc(zero_bytes(w.y ^ 0x01010101u)) + __popc(zero_bytes(w.z ^ 0x01010101u)) +
           __popc(zero_bytes(w.w ^ 0x01010101u));
  }
  if (blockIdx.x == 0 && threadIdx.x < 32) {  // the unaligned head and the tail behind the last whole vector: at most 15 bytes each
    const size_t tail0 = h + nvec * 16;
    const size_t i = threadIdx.x < 16 ? (size_t)threadIdx.x : tail0 + (threadIdx.x - 16);
    const bool in = threadIdx.x < 16 ? i < h : i < total;
    if (in) {
      const uint8_t g = gt[i];
      pos += g == 1;
      neg += g == 0;
    }
  }
  float fp = wave_sum((float)pos), fn = wave_sum((float)neg);  // < 2^24 per wavefront: exact
  // one pair of fp64 atomics per WORKGROUP (integers: exact in any order) -- 16 k of them on two addresses, one pair per wavefront of a
  // 4096-workgroup grid, were what the kernel's 0.16 ms consisted of
  __shared__ float part[2][4];
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = fp;
    part[1][threadIdx.x >> 6] = fn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(acc + 2, (double)part[0][0] + (double)part[0][1] + (double)part[0][2] + (double)part[0][3]);
    atomicAdd(acc + 3, (double)part[1][0] + (double)part[1][1] + (double)part[1][2] + (double)part[1][3]);
  }
}

// one wavefront per row: loss sums and row sums of t
__global__ void __launch_bounds__(256) focal_rows_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ gt, int M, int N,
                                                          const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                          const float* __restrict__ cmax, const float* __restrict__ csum, float alpha,
                                                          float gamma, int clamp, double* __restrict__ acc, float* __restrict__ row_t) {
  // Round 6: a wavefront takes rows row0, row0 + 4 gridDim, ...; the workgroup's loss sums leave as ONE pair of fp64 atomics.  One pair per ROW
  // (rounds 1-5) was 9600 atomics on two addresses at 4800 rows: ~0.1 of the kernel's 0.157 ms was their serialisation, not its arithmetic.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float wp = (float)(1.0 / acc[2]), wn = (float)(1.0 / acc[3]);
  double lp_wg = 0.0, ln_wg = 0.0;
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float rm = rmax[row], rs = rsum[row];
    float lp = 0.f, ln = 0.f, ts = 0.f;
    for (int j = lane; j < N; j += 64) {
      const float conf = conf_value(sim[(size_t)row * N + j], cmax[j], csum[j], rm, rs);
      const uint8_t g = gt[(size_t)row * N + j];
      if (g == 1) lp += focal_term(conf, true, alpha, gamma, clamp);
      else if (g == 0) ln += focal_term(conf, false, alpha, gamma, clamp);
      if (g <= 1) ts += focal_t(conf, g == 1, alpha, gamma, wp, wn, clamp);
    }
    lp = wave_sum(lp);
    ln = wave_sum(ln);
    ts = wave_sum(ts);
    lp_wg += (double)lp;
    ln_wg += (double)ln;
    if (lane == 0) row_t[row] = ts;
  }
  __shared__ double part[2][4];
  if (lane == 0) {
    part[0][wave] = lp_wg;
    part[1][wave] = ln_wg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(acc + 0, (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
    atomicAdd(acc + 1, (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]));
  }
}

// column sums of t: grid (ceil(N/64), COL_CHUNKS), block 256 = 64 columns x 4 row lanes; col_t zeroed by the caller
__global__ void __launch_bounds__(256) focal_cols_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ gt, int M, int N,
                                                          const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                          const float* __restrict__ cmax, const float* __restrict__ csum, float alpha,
                                                          float gamma, int clamp, const double* __restrict__ acc, float* __restrict__ col_t) {
  __shared__ float st[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int rows_per = (M + COL_CHUNKS - 1) / COL_CHUNKS;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const float wp = (float)(1.0 / acc[2]), wn = (float)(1.0 / acc[3]);
  float ts = 0.f;
  if (col < N) {
    const float cm = cmax[col], cs = csum[col];
    for (int i = r0 + ty; i < r1; i += 4) {
      const float conf = conf_value(sim[(size_t)i * N + col], cm, cs, rmax[i], rsum[i]);
      const uint8_t g = gt[(size_t)i * N + col];
      if (g <= 1) ts += focal_t(conf, g == 1, alpha, gamma, wp, wn, clamp);
    }
  }
  st[ty][tx] = ts;
  __syncthreads();
  if (ty == 0 && col < N) atomicAdd(col_t + col, (st[0][tx] + st[1][tx]) + (st[2][tx] + st[3][tx]));
}

// Round 6: focal_rows_kernel + focal_cols_kernel as ONE pass over the similarity matrix and the mask (each of the two read both and
// evaluated conf and t for every entry: 2 x 115 MB and twice the transcendentals at 4800 x 4800).  Grid (ceil(N / 256), COL_CHUNKS): a
// workgroup takes rows [r0, r1) x 256 columns, wavefront w the rows r0 + w, r0 + w + 4, ..., a lane four consecutive columns.  Nothing is
// accumulated with atomics: the row sums of t leave as one partial per (column chunk, row), the column sums as one per (row chunk, column),
// the loss sums as one pair per workgroup; focal_merge_kernel adds them in a fixed order (row_t / col_t / loss are deterministic now).
// Needs N % 4 == 0 and a 4-byte aligned mask row; everything else takes the two-kernel form above.
__global__ void __launch_bounds__(256) focal_tile_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ gt, int M, int N,
                                                          const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                          const float* __restrict__ cmax, const float* __restrict__ csum, float alpha,
                                                          float gamma, int clamp, const double* __restrict__ acc,
                                                          float* __restrict__ row_part, float* __restrict__ col_part,
                                                          double* __restrict__ loss_part) {
  __shared__ float st[4][256];
  __shared__ double part[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const int rows_per = (M + COL_CHUNKS - 1) / COL_CHUNKS;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const float wp = (float)(1.0 / acc[2]), wn = (float)(1.0 / acc[3]);
  const bool in = col < N;
  float ct[4] = {0.f, 0.f, 0.f, 0.f};
  float lp = 0.f, ln = 0.f;
  Piece<4> cm{}, cs{};
  if (in) {
    cm = load_piece<4>(cmax + col);
    cs = load_piece<4>(csum + col);
  }
  for (int i = r0 + wave; i < r1; i += 4) {
    float ts = 0.f;
    if (in) {
      const float rm = rmax[i], rs = rsum[i];
      const Piece<4> v = load_piece<4>(sim + (size_t)i * N + col);
      const unsigned g4 = *reinterpret_cast<const unsigned*>(gt + (size_t)i * N + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned g = (g4 >> (8 * e)) & 0xffu;
        if (g <= 1) {
          const float conf = conf_value(v.v[e], cm.v[e], cs.v[e], rm, rs);
          if (g == 1) lp += focal_term(conf, true, alpha, gamma, clamp);
          else ln += focal_term(conf, false, alpha, gamma, clamp);
          const float t = focal_t(conf, g == 1, alpha, gamma, wp, wn, clamp);
          ts += t;
          ct[e] += t;
        }
      }
    }
    ts = wave_sum(ts);
    if (lane == 0) row_part[(size_t)blockIdx.x * M + i] = ts;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) st[wave][lane * 4 + e] = ct[e];
  lp = wave_sum(lp);
  ln = wave_sum(ln);
  if (lane == 0) {
    part[0][wave] = (double)lp;
    part[1][wave] = (double)ln;
  }
  __syncthreads();
  if (wave == 0 && in) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (st[0][lane * 4 + e] + st[1][lane * 4 + e]) + (st[2][lane * 4 + e] + st[3][lane * 4 + e]);
    *reinterpret_cast<f32x4*>(col_part + (size_t)blockIdx.y * N + col) = o;
  }
  if (threadIdx.x == 0) {
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    loss_part[2 * wg] = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
    loss_part[2 * wg + 1] = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
  }
}

// row_t[i] = sum over the nx column chunks, col_t[j] = sum over the COL_CHUNKS row chunks, acc[0..1] += the workgroups' loss sums
__global__ void __launch_bounds__(256) focal_merge_kernel(const float* __restrict__ row_part, const float* __restrict__ col_part,
                                                           const double* __restrict__ loss_part, int M, int N, int nx,
                                                           float* __restrict__ row_t, float* __restrict__ col_t, double* __restrict__ acc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < M) {
    float s = 0.f;
    for (int c = 0; c < nx; ++c) s += row_part[(size_t)c * M + i];
    row_t[i] = s;
  }
  if (i < N) {
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < COL_CHUNKS; ++r) s += col_part[(size_t)r * N + i];
    col_t[i] = s;
  }
  if (blockIdx.x == 0) {
    __shared__ double red[2][256];
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nx * COL_CHUNKS; k += 256) {
      a += loss_part[2 * k];
      b += loss_part[2 * k + 1];
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        red[0][threadIdx.x] += red[0][threadIdx.x + o];
        red[1][threadIdx.x] += red[1][threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {  // (stream-ordered: the pairs of a batch add up one launch after the other)
      acc[0] += red[0][0];
      acc[1] += red[1][0];
    }
  }
}

// d(loss)/d(sim) = 2 t - A col_t[n] - B row_t[m]  with A / B the column / row soft-max factors; zero at masked entries
// (masked_fill has no gradient there).  Writes ddot = gl * scale * dsim (gradient w.r.t. the un-scaled dot products)
// and adds gl * sum dsim * dot to dscale (gradient of the learned temperature).
template <int VEC>
__global__ void __launch_bounds__(256) focal_bwd_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ gt,
                                                         const uint8_t* __restrict__ im_mask, const uint8_t* __restrict__ pt_mask, int M,
                                                         int N, const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                         const float* __restrict__ cmax, const float* __restrict__ csum, float alpha,
                                                         float gamma, int clamp, float scale, const float* __restrict__ gl,
                                                         const double* __restrict__ acc, const float* __restrict__ row_t,
                                                         const float* __restrict__ col_t, float* __restrict__ ddot,
                                                         double* __restrict__ dscale) {
  // (VEC = 4, round 6: 16-byte loads and stores -- N % 4 == 0 and a 4-byte aligned mask; VEC = 1: any N)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float wp = (float)(1.0 / acc[2]), wn = (float)(1.0 / acc[3]);
  const float up = gl ? *gl : 1.0f;
  double ts_wg = 0.0;  // (one fp64 atomic per workgroup, not per row: see focal_rows_kernel)
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    const float rm = rmax[row], rs = rsum[row], rt = row_t[row];
    const bool rk = im_mask ? im_mask[row] != 0 : true;
    float tsum = 0.f;
    for (int j = lane * VEC; j < N; j += 64 * VEC) {
      const Piece<VEC> v = load_piece<VEC>(sim + (size_t)row * N + j);
      Piece<VEC> d;
      unsigned g4 = 0, k4 = 0x01010101u;
      if (VEC == 4) {
        g4 = *reinterpret_cast<const unsigned*>(gt + (size_t)row * N + j);
        if (pt_mask) k4 = *reinterpret_cast<const unsigned*>(pt_mask + j);
      } else {
        g4 = gt[(size_t)row * N + j];
        if (pt_mask) k4 = pt_mask[j];
      }
      Piece<VEC> cm{}, cs{}, ctj{};
      if (rk) {
        cm = load_piece<VEC>(cmax + j);
        cs = load_piece<VEC>(csum + j);
        ctj = load_piece<VEC>(col_t + j);
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const bool keep = rk && ((k4 >> (8 * e)) & 0xffu) != 0;
        float de = 0.f;
        if (keep) {
          const float conf = conf_value(v.v[e], cm.v[e], cs.v[e], rm, rs);
          const unsigned g = (g4 >> (8 * e)) & 0xffu;
          const float t = g <= 1 ? focal_t(conf, g == 1, alpha, gamma, wp, wn, clamp) : 0.f;
          const float A = exp_fast(v.v[e] - cm.v[e]) * cs.v[e], B = exp_fast(v.v[e] - rm) * rs;
          de = up * ((2.0f * t - A * ctj.v[e]) - B * rt);
          tsum = NM_FMA(de, v.v[e], tsum);
        }
        d.v[e] = de * scale;
      }
      if (VEC == 4) {
        f32x4 o;
        o[0] = d.v[0]; o[1] = d.v[1 % VEC]; o[2] = d.v[2 % VEC]; o[3] = d.v[3 % VEC];
        *reinterpret_cast<f32x4*>(ddot + (size_t)row * N + j) = o;
      } else {
        ddot[(size_t)row * N + j] = d.v[0];
      }
    }
    ts_wg += (double)wave_sum(tsum) / (double)scale;
  }
  __shared__ double part[4];
  if (lane == 0) part[wave] = ts_wg;
  __syncthreads();
  if (threadIdx.x == 0 && dscale) atomicAdd(dscale, (part[0] + part[1]) + (part[2] + part[3]));
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Workspace {
  float *sim, *imn, *ptn, *rmax, *rsum, *cmax, *csum, *pmax, *psum, *sel_conf;
  unsigned int* colmax;
  int* sel_j;
  void* blob;  // split / re-ordered point features for the bf16x3 similarity GEMM
  size_t bytes;
};

Workspace carve(void* base, int M, int N, int C) {
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t n) {
    void* p = base ? (char*)base + off : nullptr;
    off += align256(n);
    return p;
  };
  w.sim = (float*)take((size_t)M * N * 4);
  w.imn = (float*)take((size_t)M * C * 4);
  w.ptn = (float*)take((size_t)N * C * 4);
  w.rmax = (float*)take((size_t)M * 4);
  w.rsum = (float*)take((size_t)M * 4);
  w.cmax = (float*)take((size_t)N * 4);
  w.csum = (float*)take((size_t)N * 4);
  w.pmax = (float*)take((size_t)COL_CHUNKS * N * 4);
  w.psum = (float*)take((size_t)COL_CHUNKS * N * 4);
  w.colmax = (unsigned int*)take((size_t)N * 4);
  w.sel_j = (int*)take((size_t)M * 4);
  w.sel_conf = (float*)take((size_t)M * 4);
  w.blob = take(nm_linear_blob_bytes_bf16x3(N, C));
  w.bytes = off;
  return w;
}

}  // namespace

extern "C" size_t nm_match_workspace_bytes(int M, int N, int C) {
  if (M <= 0 || N <= 0 || C <= 0) return 0;
  return carve(nullptr, M, N, C).bytes;
}

extern "C" int nm_dual_softmax_match(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                                     const uint8_t* pt_mask, float threshold, int mutual, float* conf, float* im_norm,
                                     float* pt_norm, int64_t* out_i, int64_t* out_j, float* out_conf, int* count, void* workspace,
                                     size_t workspace_bytes, nmStream_t stream) {
  return nm_dual_softmax_match_ex(im, pt, M, N, C, scale, im_mask, pt_mask, threshold, mutual, 0, conf, im_norm, pt_norm, out_i, out_j,
                                  out_conf, count, workspace, workspace_bytes, stream);
}

extern "C" int nm_dual_softmax_match_ex(const float* im, const float* pt, int M, int N, int C, float scale, const uint8_t* im_mask,
                                        const uint8_t* pt_mask, float threshold, int mutual, int flags, float* conf, float* im_norm,
                                        float* pt_norm, int64_t* out_i, int64_t* out_j, float* out_conf, int* count, void* workspace,
                                        size_t workspace_bytes, nmStream_t stream) {
  const bool stats_only = (flags & NM_MATCH_STATS_ONLY) != 0;
  NM_CHECK_ARG(im && pt && (stats_only || (out_i && out_j && out_conf && count)) && workspace && M > 0 && N > 0 && C > 0);
  if (C != 64 && C != 128 && C != 256 && C != 512) return NM_ERR_UNSUPPORTED;
  Workspace w = carve(workspace, M, N, C);
  if (workspace_bytes < w.bytes) return NM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* imn = im_norm ? im_norm : w.imn;
  float* ptn = pt_norm ? pt_norm : w.ptn;
  auto l2 = [&](const float* x, int rows, float* y) {
    const int grid = (rows + 3) / 4;
    switch (C) {
      case 64: l2norm_kernel<1><<<grid, 256, 0, s>>>(x, rows, y); break;
      case 128: l2norm_kernel<2><<<grid, 256, 0, s>>>(x, rows, y); break;
      case 256: l2norm_kernel<4><<<grid, 256, 0, s>>>(x, rows, y); break;
      default: l2norm_kernel<8><<<grid, 256, 0, s>>>(x, rows, y); break;
    }
  };
  l2(im, M, imn);
  l2(pt, N, ptn);
  int rc;
  if ((flags & NM_MATCH_BF16X3) && N % 8 == 0) rc = nm_internal_sim_bf16x3(imn, ptn, M, N, C, scale, im_mask, pt_mask, w.sim, w.blob, s);
  else rc = nm_internal_sim(imn, ptn, M, N, C, scale, im_mask, pt_mask, w.sim, s);
  if (rc != NM_OK) return rc;
  const bool vec = (N % 4 == 0);
  const bool cached = vec && N <= 64 * 4 * ROW_NREG;
  if (cached) row_stats_kernel<4, ROW_NREG><<<(M + 3) / 4, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum);
  else if (vec) row_stats_kernel<4, 0><<<(M + 3) / 4, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum);
  else row_stats_kernel<1, 0><<<(M + 3) / 4, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum);
  const int cw = vec ? 256 : 64;
  dim3 cgrid((N + cw - 1) / cw, COL_CHUNKS);
  if (vec) col_stats_partial_kernel<4><<<cgrid, 256, 0, s>>>(w.sim, M, N, w.pmax, w.psum);
  else col_stats_partial_kernel<1><<<cgrid, 256, 0, s>>>(w.sim, M, N, w.pmax, w.psum);
  col_stats_merge_kernel<<<(N + 63) / 64, 256, 0, s>>>(w.pmax, w.psum, N, w.cmax, w.csum);
  if (stats_only) return nm_launch_status();
  if (mutual) {
    if (hipMemsetAsync(w.colmax, 0, (size_t)N * 4, s) != hipSuccess) return NM_ERR_LAUNCH;
    if (vec) col_confmax_kernel<4><<<cgrid, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum, w.cmax, w.csum, w.colmax);
    else col_confmax_kernel<1><<<cgrid, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum, w.cmax, w.csum, w.colmax);
  }
  if (cached) row_select_kernel<4, ROW_NREG><<<(M + 3) / 4, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum, w.cmax, w.csum, w.colmax, threshold,
                                                                         mutual, conf, w.sel_j, w.sel_conf);
  else if (vec) row_select_kernel<4, 0><<<(M + 3) / 4, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum, w.cmax, w.csum, w.colmax, threshold,
                                                                   mutual, conf, w.sel_j, w.sel_conf);
  else row_select_kernel<1, 0><<<(M + 3) / 4, 256, 0, s>>>(w.sim, M, N, w.rmax, w.rsum, w.cmax, w.csum, w.colmax, threshold, mutual,
                                                           conf, w.sel_j, w.sel_conf);
  compact_kernel<<<1, 1024, 0, s>>>(w.sel_j, w.sel_conf, M, out_i, out_j, out_conf, count);
  return nm_launch_status();
}

extern "C" int nm_focal_count(const uint8_t* conf_gt, size_t total, double* acc, nmStream_t stream) {
  NM_CHECK_ARG(conf_gt && acc && total > 0);
  const unsigned grid = (unsigned)((total / 16 + 256 * 4 - 1) / (256 * 4) + 1);  // ~4 vectors per thread, at most two workgroups per CU
  focal_count_kernel<<<grid < 512 ? grid : 512, 256, 0, (hipStream_t)stream>>>(conf_gt, total, acc);
  return nm_launch_status();
}

extern "C" int nm_match_focal_loss(const uint8_t* conf_gt, int M, int N, int C, float alpha, float gamma, int clamp, void* workspace,
                                   size_t workspace_bytes, double* acc, float* row_t, float* col_t, nmStream_t stream) {
  NM_CHECK_ARG(conf_gt && workspace && acc && row_t && col_t && M > 0 && N > 0 && C > 0);
  Workspace w = carve(workspace, M, N, C);
  if (workspace_bytes < w.bytes) return NM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  // one pass + merge when the partial sums fit the two column-statistics scratch areas of the workspace (free since the statistics were merged)
  const int nx = (N + 255) / 256;
  const size_t cap = (size_t)COL_CHUNKS * N * 4;
  if (N % 4 == 0 && ((size_t)conf_gt & 3) == 0 && align256((size_t)nx * M * 4) + (size_t)nx * COL_CHUNKS * 16 <= cap) {
    float* row_part = w.psum;
    double* loss_part = reinterpret_cast<double*>(reinterpret_cast<char*>(w.psum) + align256((size_t)nx * M * 4));
    focal_tile_kernel<<<dim3(nx, COL_CHUNKS), 256, 0, s>>>(w.sim, conf_gt, M, N, w.rmax, w.rsum, w.cmax, w.csum, alpha, gamma, clamp, acc, row_part,
                                                            w.pmax, loss_part);
    focal_merge_kernel<<<((M > N ? M : N) + 255) / 256, 256, 0, s>>>(row_part, w.pmax, loss_part, M, N, nx, row_t, col_t, acc);
    return nm_launch_status();
  }
  if (hipMemsetAsync(col_t, 0, (size_t)N * sizeof(float), s) != hipSuccess) return NM_ERR_LAUNCH;
  focal_rows_kernel<<<(M + 7) / 8, 256, 0, s>>>(w.sim, conf_gt, M, N, w.rmax, w.rsum, w.cmax, w.csum, alpha, gamma, clamp, acc, row_t);
  focal_cols_kernel<<<dim3((N + 63) / 64, COL_CHUNKS), 256, 0, s>>>(w.sim, conf_gt, M, N, w.rmax, w.rsum, w.cmax, w.csum, alpha, gamma,
                                                                    clamp, acc, col_t);
  return nm_launch_status();
}

extern "C" int nm_match_focal_loss_bwd(const uint8_t* conf_gt, const uint8_t* im_mask, const uint8_t* pt_mask, int M, int N, int C,
                                       float alpha, float gamma, int clamp, float scale, const float* grad_loss, void* workspace,
                                       size_t workspace_bytes, const double* acc, const float* row_t, const float* col_t, float* ddot,
                                       double* dscale, nmStream_t stream) {
  NM_CHECK_ARG(conf_gt && workspace && acc && row_t && col_t && ddot && M > 0 && N > 0 && C > 0 && scale != 0.f);
  Workspace w = carve(workspace, M, N, C);
  if (workspace_bytes < w.bytes) return NM_ERR_WORKSPACE;
  if (N % 4 == 0 && ((size_t)conf_gt & 3) == 0 && (!pt_mask || ((size_t)pt_mask & 3) == 0))
    focal_bwd_kernel<4><<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(w.sim, conf_gt, im_mask, pt_mask, M, N, w.rmax, w.rsum, w.cmax, w.csum,
                                                                      alpha, gamma, clamp, scale, grad_loss, acc, row_t, col_t, ddot, dscale);
  else
    focal_bwd_kernel<1><<<(M + 3) / 4, 256, 0, (hipStream_t)stream>>>(w.sim, conf_gt, im_mask, pt_mask, M, N, w.rmax, w.rsum, w.cmax, w.csum,
                                                                      alpha, gamma, clamp, scale, grad_loss, acc, row_t, col_t, ddot, dscale);
  return nm_launch_status();
}
