// Training-side kernels of the matcher head: backward passes of the layers that
// gemm*.hip / attention.hip / match.hip run forward.  Everything is fp32; reductions over rows (weight / bias / LayerNorm
// parameter gradients) are two-stage or atomic as noted per kernel.
//
//   nm_linear_wgrad    dW[N,K] = dy[M,N]^T . x[M,K]        (nn.Linear weight gradient; also d(pt) of the similarity GEMM)
//   nm_col_sum         db[N]   = sum_m dy[m,:]             (bias gradient)
//   nm_gelu / _bwd     exact-erf GELU and its derivative    (nn.GELU(), reference modules/attention.py:136-154)
//   nm_layernorm_bwd   dx, dgamma, dbeta                    (nn.LayerNorm, eps inside the square root)
//   nm_l2norm_bwd      d f  of  f / (|f| + 1e-6)            (coarse_matching normalisation, c2f_trainer.py:290-291)
#include "common.h"

#include <algorithm>

namespace {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------- weight gradient
// One workgroup = one 64x64 tile of dW over one slice of the M rows; its 4 wavefronts own the four 32x32 blocks.
// v_mfma_f32_32x32x2f32 computes D[i][j] += sum_{kk<2} A[i][kk] B[kk][j] with lane = (i or j) + 32 kk, i.e. both operands
// are (row m, column) elements of the row-major dy / x -- no transposes anywhere.  Rows are staged through LDS in chunks of
// 32 (coalesced 16-byte loads, register double buffering: the loads of chunk c+1 are in flight while chunk c feeds the
// matrix cores), read back one float per lane per MFMA (conflict-free: 32 consecutive floats per half wavefront).
// Two accumulators per wavefront (even / odd row pairs) keep consecutive MFMAs independent.  With 64x64 tiles every row of
// dy / x is read by N/64 resp. K/64 workgroups: at N = K = 256 the kernel is bound by that 4x re-read (an MFMA-free build
// takes 80 % of the time), larger tiles would need more row slices (partial-sum traffic) to fill the chip -- left as is.
// grid (ceil(N/64) * ceil(K/64), splits); block 256.  splits > 1: partial tiles go to `part` [splits][N][K] and
// wgrad_reduce_kernel sums them in a fixed order (deterministic; no atomics).
constexpr int WG_CHUNK = 32;
__global__ void __launch_bounds__(256) wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int N, int K,
                                                     int rows_per, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sm[2][2][WG_CHUNK * 64];  // [stage][dy | x][row][64 columns]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hi = lane >> 5;
  const int tiles_k = (K + 63) / 64;
  const int n0 = (blockIdx.x / tiles_k) * 64, k0 = (blockIdx.x % tiles_k) * 64;
  const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
  const int nb = wave >> 1, kb = wave & 1;  // this wavefront's 32x32 block of the tile
  // staging role: thread t moves the 16-byte piece (row t/16 (+16), columns 4 (t%16) ..) of both matrices
  const int srow = tid >> 4, scol = (tid & 15) * 4;
  const bool vn = n0 + scol < N, vk = k0 + scol < K;  // N, K are multiples of 4: a piece is inside or outside
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int PF = 4;  // chunks of rows in flight global -> registers (memory latency >> the 0.4 us a chunk spends in the MFMAs)
  f32x4 gd[PF][2], gx[PF][2];
  auto gload = [&](int slot, int mc) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = mc + srow + 16 * h;
      gd[slot][h] = (m < m1 && vn) ? *reinterpret_cast<const f32x4*>(dy + (size_t)m * N + n0 + scol) : zero4;
      gx[slot][h] = (m < m1 && vk) ? *reinterpret_cast<const f32x4*>(x + (size_t)m * K + k0 + scol) : zero4;
    }
  };
  auto sstore = [&](int st, int slot) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f32x4*>(&sm[st][0][(srow + 16 * h) * 64 + scol]) = gd[slot][h];
      *reinterpret_cast<f32x4*>(&sm[st][1][(srow + 16 * h) * 64 + scol]) = gx[slot][h];
    }
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < PF; ++i) gload(i, m0 + i * WG_CHUNK);  // (rows past m1 load nothing: zeros)
  sstore(0, 0);
  __syncthreads();
  for (int mc0 = m0; mc0 < m1; mc0 += PF * WG_CHUNK) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int mc = mc0 + i * WG_CHUNK, st = i & 1;  // (PF is even: the LDS stage of a chunk is its register slot parity)
      if (mc < m1) {                                   // uniform
        gload(i, mc + PF * WG_CHUNK);                  // slot i was copied to LDS one iteration ago
        const float* a = &sm[st][0][hi * 64 + nb * 32 + j];
        const float* b = &sm[st][1][hi * 64 + kb * 32 + j];
        float av[WG_CHUNK / 2], bv[WG_CHUNK / 2];
#pragma unroll
        for (int s2 = 0; s2 < WG_CHUNK / 2; ++s2) {
          av[s2] = a[(2 * s2) * 64];
          bv[s2] = b[(2 * s2) * 64];
        }
#pragma unroll
        for (int s2 = 0; s2 < WG_CHUNK / 2; s2 += 2) {
          acc0 = MFMA32(av[s2], bv[s2], acc0);
          acc1 = MFMA32(av[s2 + 1], bv[s2 + 1], acc1);
        }
        if (mc + WG_CHUNK < m1) sstore(st ^ 1, (i + 1) % PF);
        __syncthreads();
      }
    }
  }
  // register r of lane (j, hi) <-> row (r & 3) + 8 (r >> 2) + 4 hi of the block, column j
  float* o = out + (size_t)blockIdx.y * N * K;
  const int kc = k0 + 32 * kb + j;
  if (kc < K) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (n < N) o[(size_t)n * K + kc] = acc0[r] + acc1[r];
    }
  }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, int splits, size_t total, int accumulate,
                                                            float* __restrict__ dw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float s = accumulate ? dw[i] : 0.f;
  for (int p = 0; p < splits; ++p) s += part[(size_t)p * total + i];
  dw[i] = s;
}

// ---------------------------------------------------------------------------------------------------- weight gradient, split bf16
// Round 6.  dW = dy^T . x contracts over the ROWS of both operands; an MFMA operand holds, per lane, eight values of the contraction
// index for ONE output row / column.  Two freedoms make that loadable straight from global memory, with no LDS transposition:
//   * which eight rows a lane's register slots mean is free as long as both operands agree: rows mc + 8 (lane / 32) + r;
//   * which column of a 32-wide block a lane stands for is free as long as the store follows: lane i loads dy[m][n0 + 2 i .. + 1] (8 bytes) and
//     x[m][k0 + 4 i .. + 3] (16 bytes) -- 256 / 512 contiguous bytes per half wavefront and row -- and uses the two / four values for the same
//     lane position of two / four DIFFERENT blocks: block a of dy holds the columns n0 + 2 i + a, block b of x the columns k0 + 4 i + b.
// A wavefront accumulates a 64 x 128 tile of dW (2 x 4 blocks; sixteen load instructions per 16 rows for 24 MFMAs); each fp32 value is split
// on the fly (hi = bf16(v), lo = bf16(v - hi)), a product is hi.hi + hi.lo + lo.hi with fp32 accumulation -- the arithmetic of gemm_bf16.hip, i.e.
// of the dX GEMMs of the same backward pass.  Loads run two steps ahead (register ring of three, unconditional: out-of-range rows / columns are
// clamped to an address inside the matrix and zeroed by a select -- a predicated load is a branch and costs the compiler its counted waits).
// The four wavefronts of a workgroup take four quarters of the workgroup's row slice and add their tiles through LDS (128 KB), so the partial
// sums that go to memory -- `out` [slices][N][K], summed in a fixed order by wgrad_reduce4_kernel -- are a quarter of the wavefront count.
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// db (may be NULL): the column sums of dy (the bias gradient of the same layer) -- the workgroups of the first tile column add up the dy values
// they load anyway; one fp32 atomic per column and workgroup (order-dependent in the last bits, like col_sum_kernel), db zeroed by the caller.
__global__ void __launch_bounds__(256) wgrad_bf16x3_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int N, int K,
                                                           int rows_per, float* __restrict__ out, float* __restrict__ db) {
  __shared__ float red[4][64 * 128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
  const int tiles_k = (K + 127) / 128;
  const int n0 = (blockIdx.x / tiles_k) * 64, k0 = (blockIdx.x % tiles_k) * 128;
  const int rpw = ((rows_per + 3) / 4 + 15) / 16 * 16;  // rows per wavefront
  const int m0 = blockIdx.y * rows_per + wave * rpw;
  const int m1 = min(min(M, (int)blockIdx.y * rows_per + rows_per), m0 + rpw);
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
  const bool cn = n0 + 2 * i < N, ck = k0 + 4 * i < K;  // (N even, K a multiple of 4: a piece is inside or outside)
  const float* dcol = dy + (cn ? n0 + 2 * i : 0);
  const float* xcol = x + (ck ? k0 + 4 * i : 0);
  f32x2 sa[3][8];
  f32x4 sb[3][8];
  const bool want_db = db != nullptr && (blockIdx.x % tiles_k) == 0;  // workgroup-uniform
  float csum[2] = {0.f, 0.f};
  auto load = [&](int st, int mc) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int m = min(mc + 8 * h + r, M - 1);
      sa[st][r] = *reinterpret_cast<const f32x2*>(dcol + (size_t)m * N);
      sb[st][r] = *reinterpret_cast<const f32x4*>(xcol + (size_t)m * K);
    }
  };
  auto products = [&](int st, int mc) {
    wg_bf16x8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bool in = mc + 8 * h + r < m1;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float v = (in && cn) ? sa[st][r][a] : 0.f;
        csum[a] += v;
        const __bf16 hv = (__bf16)v;
        ah[a][r] = hv;
        al[a][r] = (__bf16)(v - (float)hv);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float v = (in && ck) ? sb[st][r][b] : 0.f;
        const __bf16 hv = (__bf16)v;
        bh[b][r] = hv;
        bl[b][r] = (__bf16)(v - (float)hv);
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
      }
  };
  if (m0 < m1) {  // (wavefront-uniform; a wavefront without rows contributes its zeros below)
    load(0, m0);
    load(1, m0 + 16);
    for (int mc = m0; mc < m1; mc += 48) {
#pragma unroll
      for (int st = 0; st < 3; ++st) {
        const int cur = mc + 16 * st;
        if (cur < m1) {  // uniform
          load((st + 2) % 3, cur + 32);
          products(st, cur);
        }
      }
    }
  }
  // the four wavefronts' tiles -> one: register v of block (a, b) is row 8 (v / 4) + 4 h + v % 4 of dy-block a, lane position i of x-block b
#pragma unroll
  for (int v = 0; v < 16; ++v)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) red[wave][((v * 2 + a) * 4 + b) * 64 + lane] = acc[a][b][v];
  __syncthreads();
  if (want_db) {
    // (red is read below only after this; the column sums go through the shared array's last row... a separate small array keeps it simple)
    __shared__ float cs[4][2][32];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float t = csum[a] + __shfl_xor(csum[a], 32, 64);  // the two row halves
      if (h == 0) cs[wave][a][i] = t;
    }
    __syncthreads();
    if (wave == 0 && h == 0 && cn) {
#pragma unroll
      for (int a = 0; a < 2; ++a) atomicAdd(db + n0 + 2 * i + a, (cs[0][a][i] + cs[1][a][i]) + (cs[2][a][i] + cs[3][a][i]));
    }
  }
  float* o = out + (size_t)blockIdx.y * N * K;
#pragma unroll
  for (int vv = 0; vv < 4; ++vv) {
    const int v = 4 * wave + vv;  // this wavefront sums and stores the registers 4 wave .. 4 wave + 3 of every block
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      f32x4 sum;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int e = ((v * 2 + a) * 4 + b) * 64 + lane;
        sum[b] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
      }
      const int n = n0 + 2 * (8 * (v >> 2) + 4 * h + (v & 3)) + a;  // block a holds the columns n0 + 2 (position) + a of dy
      if (n < N && ck) *reinterpret_cast<f32x4*>(o + (size_t)n * K + k0 + 4 * i) = sum;
    }
  }
}

// dw[i] (+)= sum over the slices, four independent partial sums (slices p = 0, 4, 8, ...; 1, 5, ...; ...) added in a fixed order
__global__ void __launch_bounds__(256) wgrad_reduce4_kernel(const float* __restrict__ part, int splits, size_t total4, int accumulate,
                                                             float* __restrict__ dw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int p = 0;
  for (; p + 3 < splits; p += 4) {
    s0 += p4[(size_t)p * total4 + i];
    s1 += p4[(size_t)(p + 1) * total4 + i];
    s2 += p4[(size_t)(p + 2) * total4 + i];
    s3 += p4[(size_t)(p + 3) * total4 + i];
  }
  for (; p < splits; ++p) s0 += p4[(size_t)p * total4 + i];
  f32x4 s = (s0 + s1) + (s2 + s3);
  if (accumulate) s += reinterpret_cast<const f32x4*>(dw)[i];
  reinterpret_cast<f32x4*>(dw)[i] = s;
}

// ---------------------------------------------------------------------------------------------------- column sums
// grid (ceil(N/256), chunks); block 256: thread = column, rows of the chunk in sequence; merged with float atomics
// (order-dependent in the last bits; `out` must be zeroed or hold the value to accumulate onto).
__global__ void __launch_bounds__(256) col_sum_kernel(const float* __restrict__ dy, int M, int N, int rows_per, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int m = m0;
  for (; m + 3 < m1; m += 4) {
    s0 += dy[(size_t)m * N + c];
    s1 += dy[(size_t)(m + 1) * N + c];
    s2 += dy[(size_t)(m + 2) * N + c];
    s3 += dy[(size_t)(m + 3) * N + c];
  }
  for (; m < m1; ++m) s0 += dy[(size_t)m * N + c];
  atomicAdd(out + c, (s0 + s1) + (s2 + s3));
}

// ---------------------------------------------------------------------------------------------------- GELU
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_d(float v) {
  const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
  return cdf + v * pdf;
}
__global__ void __launch_bounds__(256) gelu_kernel(const float* __restrict__ u, size_t n4, float* __restrict__ h) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(u)[i];
  reinterpret_cast<f32x4*>(h)[i] = f32x4{gelu_f(v[0]), gelu_f(v[1]), gelu_f(v[2]), gelu_f(v[3])};
}
// du = dh where the forward's ReLU output h is positive, else 0 (nn.ReLU, the FeedForwardNetwork's other activation: attention.py:136-154)
__global__ void __launch_bounds__(256) relu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dh, size_t n4, float* __restrict__ du) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(h)[i], g = reinterpret_cast<const f32x4*>(dh)[i];
  reinterpret_cast<f32x4*>(du)[i] = f32x4{v[0] > 0.f ? g[0] : 0.f, v[1] > 0.f ? g[1] : 0.f, v[2] > 0.f ? g[2] : 0.f, v[3] > 0.f ? g[3] : 0.f};
}

__global__ void __launch_bounds__(256) gelu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dh, size_t n4,
                                                        float* __restrict__ du) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(u)[i];
  const f32x4 g = reinterpret_cast<const f32x4*>(dh)[i];
  reinterpret_cast<f32x4*>(du)[i] = f32x4{g[0] * gelu_d(v[0]), g[1] * gelu_d(v[1]), g[2] * gelu_d(v[2]), g[3] * gelu_d(v[3])};
}

// ---------------------------------------------------------------------------------------------------- LayerNorm backward
// One wavefront per row at a time (dim = 64 PER), rows grid-strided so that the parameter gradients are summed in
// registers, across the workgroup's 4 wavefronts through LDS, and merged with 2 dim atomics per workgroup.
//   xh = (x - mean) rstd;  g = dy gamma;  dx = rstd (g - mean(g) - xh mean(g xh));  dgamma += dy xh;  dbeta += dy
template <int PER>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ dy, int rows, float eps, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  // Round 6: a lane holds PER CONSECUTIVE columns (one 16-byte load per operand at dim 256 instead of four 4-byte ones) and the next row's
  // loads are issued before this row's four lane reductions; two workgroups per CU (the first version ran one wavefront per SIMD through a
  // chain of load -> reduce -> reduce -> reduce -> store per row with nothing to hide its latency behind: 24 us for 7200 x 256).
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  constexpr int dim = 64 * PER;
  typedef float vec_t __attribute__((ext_vector_type(PER)));
  float gm[PER], dg[PER], db[PER];
  {
    const vec_t g = *reinterpret_cast<const vec_t*>(gamma + lane * PER);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      gm[i] = PER == 1 ? gamma[lane] : g[i];
      dg[i] = 0.f;
      db[i] = 0.f;
    }
  }
  vec_t vn = {}, dn = {};
  if (wave < rows) {
    vn = *reinterpret_cast<const vec_t*>(x + (size_t)wave * dim + lane * PER);
    dn = *reinterpret_cast<const vec_t*>(dy + (size_t)wave * dim + lane * PER);
  }
  for (int row = wave; row < rows; row += nwaves) {
    float v[PER], d[PER], s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      v[i] = vn[i];
      d[i] = dn[i];
      s += v[i];
    }
    if (row + nwaves < rows) {
      vn = *reinterpret_cast<const vec_t*>(x + (size_t)(row + nwaves) * dim + lane * PER);
      dn = *reinterpret_cast<const vec_t*>(dy + (size_t)(row + nwaves) * dim + lane * PER);
    }
    const float mean = wave_sum(s) / (float)dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      v[i] -= mean;
      q = NM_FMA(v[i], v[i], q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)dim + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      v[i] *= rstd;  // xh
      const float g = d[i] * gm[i];
      sg += g;
      sgx = NM_FMA(g, v[i], sgx);
      dg[i] = NM_FMA(d[i], v[i], dg[i]);
      db[i] += d[i];
    }
    const float mg = wave_sum(sg) / (float)dim, mgx = wave_sum(sgx) / (float)dim;
    vec_t o;
#pragma unroll
    for (int i = 0; i < PER; ++i) o[i] = rstd * ((d[i] * gm[i] - mg) - v[i] * mgx);
    *reinterpret_cast<vec_t*>(dx + (size_t)row * dim + lane * PER) = o;
  }
  if (!dgamma) return;  // frozen parameters (the iNeRF matching term): input gradient only -- no reduction, no atomics, nothing to zero
  __shared__ float red[4][2][64 * PER];  // [wavefront][dgamma | dbeta][column]
  const int w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    red[w][0][lane * PER + i] = dg[i];
    red[w][1][lane * PER + i] = db[i];
  }
  __syncthreads();
  if (w > 0) return;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;  // consecutive lanes, consecutive addresses
    atomicAdd(dgamma + c, (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]));
    atomicAdd(dbeta + c, (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]));
  }
}

// ---------------------------------------------------------------------------------------------------- l2-normalise backward
// y = f / (r + e), r = |f|:   df = dy / (r + e) - f (f . dy) / (r (r + e)^2)
template <int PER>
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const float* __restrict__ f, const float* __restrict__ dy, int rows,
                                                          float* __restrict__ df) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[PER], d[PER], q = 0.f, p = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    v[i] = f[(size_t)row * 64 * PER + lane + 64 * i];
    d[i] = dy[(size_t)row * 64 * PER + lane + 64 * i];
    q = NM_FMA(v[i], v[i], q);
    p = NM_FMA(v[i], d[i], p);
  }
  const float r = sqrtf(wave_sum(q)), den = r + 1e-6f;
  p = wave_sum(p);
  const float a = 1.0f / den, b = r > 0.f ? p / (r * den * den) : 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) df[(size_t)row * 64 * PER + lane + 64 * i] = d[i] * a - v[i] * b;
}

}  // namespace

#ifndef WG_MAX_SPLITS
#define WG_MAX_SPLITS 16
#endif
#ifndef WG_TARGET
#define WG_TARGET 512
#endif
// slices of the rows for the split-bf16 kernel: enough workgroups for the chip, at least 256 rows per workgroup (64 per wavefront), the partial
// tiles that go through memory bounded by 16 MB
static int wb_slices(int M, int N, int K) {
  const int tiles = ((N + 63) / 64) * ((K + 127) / 128);
  long long s = (384 + tiles - 1) / tiles;
  s = std::min<long long>(s, 128);
  s = std::min<long long>(s, (M + 255) / 256);
  s = std::min<long long>(s, (16ll << 20) / ((long long)N * K * 4));
  return (int)std::max<long long>(s, 1);
}

extern "C" size_t nm_linear_wgrad_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const size_t a = (size_t)WG_MAX_SPLITS * N * K * sizeof(float);  // the fp32 kernel: at most WG_MAX_SPLITS row slices
  const size_t b = (size_t)wb_slices(M, N, K) * N * K * sizeof(float);
  return a > b ? a : b;
}

extern "C" int nm_linear_wgrad(const float* dy, const float* x, int M, int N, int K, int accumulate, float* dw, void* workspace,
                               size_t workspace_bytes, nmStream_t stream) {
  NM_CHECK_ARG(dy && x && dw && M > 0 && N > 0 && K > 0);
  if (N % 4 != 0 || K % 4 != 0) return NM_ERR_UNSUPPORTED;  // 16-byte row pieces
  hipStream_t s = (hipStream_t)stream;
  const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
  // enough workgroups to fill 256 CUs twice over, slices of at least 256 rows (64 per wavefront)
  int splits = (WG_TARGET + tiles - 1) / tiles;
  splits = max(1, min(min(splits, WG_MAX_SPLITS), (M + 255) / 256));
  int rows_per = ((M + splits - 1) / splits + WG_CHUNK - 1) / WG_CHUNK * WG_CHUNK;
  splits = (M + rows_per - 1) / rows_per;
  if (splits == 1 && !accumulate) {
    wgrad_kernel<<<dim3(tiles, 1), 256, 0, s>>>(dy, x, M, N, K, rows_per, dw);
    return nm_launch_status();
  }
  const size_t total = (size_t)N * K;
  if (!workspace || workspace_bytes < (size_t)splits * total * sizeof(float)) return NM_ERR_WORKSPACE;
  float* part = (float*)workspace;
  wgrad_kernel<<<dim3(tiles, splits), 256, 0, s>>>(dy, x, M, N, K, rows_per, part);
  wgrad_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, s>>>(part, splits, total, accumulate, dw);
  return nm_launch_status();
}

extern "C" int nm_linear_wgrad_bf16x3(const float* dy, const float* x, int M, int N, int K, int accumulate, float* dw, void* workspace,
                                      size_t workspace_bytes, nmStream_t stream) {
  return nm_linear_wgrad_bias_bf16x3(dy, x, M, N, K, accumulate, dw, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int nm_linear_wgrad_bias_bf16x3(const float* dy, const float* x, int M, int N, int K, int accumulate, float* dw, float* db,
                                           void* workspace, size_t workspace_bytes, nmStream_t stream) {
  NM_CHECK_ARG(dy && x && dw && M > 0 && N > 0 && K > 0);
  if (N % 2 != 0 || K % 4 != 0 || ((size_t)dy & 7) || ((size_t)x & 15) || ((size_t)dw & 15)) return NM_ERR_UNSUPPORTED;  // 8-byte pieces of dy rows, 16-byte pieces of x / dw rows
  hipStream_t s = (hipStream_t)stream;
  const int tiles = ((N + 63) / 64) * ((K + 127) / 128);
  int splits = wb_slices(M, N, K);
  const int rows_per = ((M + splits - 1) / splits + 63) / 64 * 64;
  splits = (M + rows_per - 1) / rows_per;
  if (db && !accumulate && hipMemsetAsync(db, 0, (size_t)N * sizeof(float), s) != hipSuccess) return NM_ERR_LAUNCH;
  if (splits == 1 && !accumulate) {
    wgrad_bf16x3_kernel<<<dim3(tiles, 1), 256, 0, s>>>(dy, x, M, N, K, rows_per, dw, db);
    return nm_launch_status();
  }
  const size_t total = (size_t)N * K;
  if (!workspace || workspace_bytes < (size_t)splits * total * sizeof(float)) return NM_ERR_WORKSPACE;
  float* part = (float*)workspace;
  wgrad_bf16x3_kernel<<<dim3(tiles, splits), 256, 0, s>>>(dy, x, M, N, K, rows_per, part, db);
  wgrad_reduce4_kernel<<<(unsigned)((total / 4 + 255) / 256), 256, 0, s>>>(part, splits, total / 4, accumulate, dw);
  return nm_launch_status();
}

extern "C" int nm_col_sum(const float* dy, int M, int N, int accumulate, float* out, nmStream_t stream) {
  NM_CHECK_ARG(dy && out && M > 0 && N > 0);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s) != hipSuccess) return NM_ERR_LAUNCH;
  const int chunks = max(1, min(256, M / 64));
  const int rows_per = (M + chunks - 1) / chunks;
  col_sum_kernel<<<dim3((N + 255) / 256, (M + rows_per - 1) / rows_per), 256, 0, s>>>(dy, M, N, rows_per, out);
  return nm_launch_status();
}

extern "C" int nm_gelu(const float* u, size_t n, float* h, nmStream_t stream) {
  NM_CHECK_ARG(u && h && n > 0);
  if (n % 4 != 0) return NM_ERR_UNSUPPORTED;
  gelu_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(u, n / 4, h);
  return nm_launch_status();
}

extern "C" int nm_relu_bwd(const float* h, const float* dh, size_t n, float* du, nmStream_t stream) {
  NM_CHECK_ARG(h && dh && du && n > 0 && n % 4 == 0);
  relu_bwd_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(h, dh, n / 4, du);
  return nm_launch_status();
}

extern "C" int nm_gelu_bwd(const float* u, const float* dh, size_t n, float* du, nmStream_t stream) {
  NM_CHECK_ARG(u && dh && du && n > 0);
  if (n % 4 != 0) return NM_ERR_UNSUPPORTED;
  gelu_bwd_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(u, dh, n / 4, du);
  return nm_launch_status();
}

extern "C" int nm_layernorm_bwd(const float* x, const float* gamma, const float* dy, int rows, int dim, float eps, float* dx,
                                float* dgamma, float* dbeta, nmStream_t stream) {
  NM_CHECK_ARG(x && gamma && dy && dx && ((dgamma != nullptr) == (dbeta != nullptr)) && rows > 0);
  hipStream_t s = (hipStream_t)stream;
  const int grid = max(1, min((rows + 3) / 4, (dgamma ? 1 : 2) * nm_cu_count()));  // (with parameter gradients: 2 dim atomics per workgroup)
  switch (dim) {
    case 64: layernorm_bwd_kernel<1><<<grid, 256, 0, s>>>(x, gamma, dy, rows, eps, dx, dgamma, dbeta); break;
    case 128: layernorm_bwd_kernel<2><<<grid, 256, 0, s>>>(x, gamma, dy, rows, eps, dx, dgamma, dbeta); break;
    case 256: layernorm_bwd_kernel<4><<<grid, 256, 0, s>>>(x, gamma, dy, rows, eps, dx, dgamma, dbeta); break;
    case 512: layernorm_bwd_kernel<8><<<grid, 256, 0, s>>>(x, gamma, dy, rows, eps, dx, dgamma, dbeta); break;
    default: return NM_ERR_UNSUPPORTED;
  }
  return nm_launch_status();
}

extern "C" int nm_l2norm_bwd(const float* f, const float* dy, int rows, int dim, float* df, nmStream_t stream) {
  NM_CHECK_ARG(f && dy && df && rows > 0);
  hipStream_t s = (hipStream_t)stream;
  const int grid = (rows + 3) / 4;
  switch (dim) {
    case 64: l2norm_bwd_kernel<1><<<grid, 256, 0, s>>>(f, dy, rows, df); break;
    case 128: l2norm_bwd_kernel<2><<<grid, 256, 0, s>>>(f, dy, rows, df); break;
    case 256: l2norm_bwd_kernel<4><<<grid, 256, 0, s>>>(f, dy, rows, df); break;
    case 512: l2norm_bwd_kernel<8><<<grid, 256, 0, s>>>(f, dy, rows, df); break;
    default: return NM_ERR_UNSUPPORTED;
  }
  return nm_launch_status();
}
