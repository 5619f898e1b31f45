// iNeRF pose refinement: the differentiable fine pass of
// NeRFMatchEvaluator.inerf_refinement (nerfmatch/nerfmatch_evaluator.py:348-430) as forward / backward kernel pairs.
//
// The gradient of the photometric loss reaches the pose only through the ray origins o and view directions v of the
// FINE pass (the samplers and the Gaussians' variances see detached rays, the coarse network runs under no_grad):
//   x_pts = IPE(o + t_mean * v, var)          -> nm_inerf_encode      / nm_inerf_encode_bwd
//   x_dir = PE(v), appearance row                (same kernels)
//   8x256 MLP + heads                          -> nm_linear / nm_linear_bf16x3, forward and (transposed) backward
//   white-background compositing, delta * |d|  -> nm_inerf_composite  / nm_inerf_composite_bwd
// Samples s >= S_act are not evaluated: with the reference's randomized resampler the intervals s > S/2 have zero
// width, i.e. weight 0 and -- because d(alpha)/d(sigma) = delta exp(-sigma delta) = 0 and delta = dz |d| with dz = 0 --
// gradient exactly 0 (cf. NM_NERF_ZERO_TAIL).  S_act = S evaluates everything.
#include "common.h"

namespace {

constexpr int XI = 96;   // IPE columns (90 used)
constexpr int XD = 48;   // view-direction PE (27) | appearance row (16) | padding
constexpr float HALF_PI_F = 1.57079637050628662109375f;

struct Gauss {
  float t_mean, var[3];
};

// conical frustum -> Gaussian of interval [t0, t1] (render_utils.py:365-374, :326-339); d = rays[3:6]
__device__ __forceinline__ Gauss frustum(float t0, float t1, const float* d, float radius) {
  const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
  const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
  const float denom = fmaxf(1.1920928955078125e-07f, 3.0f * mu2 + hw2);
  Gauss g;
  g.t_mean = mu + (2.0f * mu * hw2) / denom;
  const float t_var = hw2 / 3.0f - (float)(4.0 / 15.0) * ((hw4 * (12.0f * mu2 - hw2)) / (denom * denom));
  const float r_var = (radius * radius) * ((mu2 / 4.0f + (float)(5.0 / 12.0) * hw2) - (float)(4.0 / 15.0) * hw4 / denom);
  const float dsq[3] = {d[0] * d[0], d[1] * d[1], d[2] * d[2]};
  const float dmag = fmaxf(1e-10f, (dsq[0] + dsq[1]) + dsq[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) g.var[a] = t_var * dsq[a] + r_var * (1.0f - dsq[a] / dmag);
  return g;
}

// 16 threads per sample (round 4; was one thread per output element, each recomputing the sample's Gaussian): thread i < 15 writes the
// six IPE columns of frequency 2^i (3 axes x {sin, sin(. + pi/2)}), thread 15 the view-direction row xd and the padding of xi
__global__ void inerf_encode_kernel(const float* __restrict__ rays, const float* __restrict__ z, int R, int S, int Sa,
                                    const float* __restrict__ app_row, float* __restrict__ xi, float* __restrict__ xd) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = idx >> 4;
  const int i = (int)(idx & 15);
  if (n >= (size_t)R * Sa) return;
  const int r = (int)(n / Sa), s = (int)(n % Sa);
  const float* rp = rays + (size_t)r * 12;
  if (i < 15) {
    const Gauss g = frustum(z[(size_t)r * (S + 1) + s], z[(size_t)r * (S + 1) + s + 1], rp + 3, rp[11]);
    const float sc = (float)(1 << i);
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const float mean = rp[ax] + g.t_mean * rp[8 + ax];
      const float xe = mean * sc;
      const float damp = expf(-0.5f * (g.var[ax] * (sc * sc)));
      xi[n * XI + i * 3 + ax] = damp * nm_sinf(xe);
      xi[n * XI + 45 + i * 3 + ax] = damp * nm_sinf(xe + HALF_PI_F);
    }
  } else {
#pragma unroll
    for (int f = 90; f < XI; ++f) xi[n * XI + f] = 0.f;
    for (int c = 0; c < XD; ++c) {
      float v = 0.f;
      if (c < 24) {
        const int k = (c % 12) / 3, ax = c % 3;
        const float xe = rp[8 + ax] * (float)(1 << k);
        v = nm_sinf(c < 12 ? xe : xe + HALF_PI_F);
      } else if (c < 27) {
        v = rp[8 + (c - 24)];
      } else if (c < 43) {
        v = app_row ? app_row[c - 27] : 0.f;
      }
      xd[n * XD + c] = v;
    }
  }
}

// g_o[r], g_v[r] = d loss / d origin, d loss / d view direction of ray r.  One workgroup (256 threads) per ray.
// Round 6: the first version ran one thread per SAMPLE -- every load instruction touched 64 rows of 384 bytes, and the 33 sums of a ray went
// through LDS atomics on one address each (128-way serialised): 380 us per step for 540 MB, 9 % of an iNeRF step.  Now a thread takes
// (sample, frequency, axis) ITEMS in the order the gradient rows lie in memory (45 consecutive floats per half row), the view-direction rows
// are read as whole 16-byte pieces, and the sums are lane reductions + one LDS round.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
constexpr int ENC_MAX_S = 1024;  // samples per ray this kernel holds Gaussians for (nm_inerf_encode_bwd* refuse more)
template <bool TWO>
__global__ void __launch_bounds__(256) inerf_encode_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ z, int R, int S,
                                                                int Sa, const float* __restrict__ gxi, const float* __restrict__ gxi2,
                                                                const float* __restrict__ gxd, float* __restrict__ g_o, float* __restrict__ g_v) {
  __shared__ float gauss[ENC_MAX_S][4];  // t_mean, var[3] per sample
  __shared__ float red[21][XD + 1];      // view-direction column sums per row group
  __shared__ float part[4][6];
  __shared__ float acc[6 + 27];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* rp = rays + (size_t)r * 12;
  for (int sidx = tid; sidx < Sa; sidx += 256) {
    const Gauss g = frustum(z[(size_t)r * (S + 1) + sidx], z[(size_t)r * (S + 1) + sidx + 1], rp + 3, rp[11]);
    gauss[sidx][0] = g.t_mean;
    gauss[sidx][1] = g.var[0];
    gauss[sidx][2] = g.var[1];
    gauss[sidx][3] = g.var[2];
  }
  __syncthreads();
  const float o3[3] = {rp[0], rp[1], rp[2]}, v3[3] = {rp[8], rp[9], rp[10]};
  const float* gi = gxi + (size_t)r * Sa * XI;
  const float* gi2 = gxi2 + (size_t)r * Sa * XI;  // TWO: second contribution to d loss / d xi (the skip connection's, nm_nerf_points_bwd_bf16x3)
  float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};  // sum of d loss / d mean, and of the same times t_mean
  for (int p = tid; p < Sa * 45; p += 256) {
    const int sidx = p / 45, c = p - sidx * 45;
    const int i = c / 3, ax = c - i * 3;
    const size_t e = (size_t)sidx * XI + c;
    float ga = gi[e], gb = gi[e + 45];
    if constexpr (TWO) { ga += gi2[e]; gb += gi2[e + 45]; }
    const float t_mean = gauss[sidx][0], var = gauss[sidx][1 + ax];
    const float oa = ax == 0 ? o3[0] : (ax == 1 ? o3[1] : o3[2]), va = ax == 0 ? v3[0] : (ax == 1 ? v3[1] : v3[2]);
    const float sc = (float)(1 << i);
    const float mean = oa + t_mean * va;
    const float xe = mean * sc;
    const float damp = expf(-0.5f * (var * (sc * sc)));
    // d/dx [damp sin(x)] = damp cos(x);  d/dx [damp sin(fl(x + pi/2))] = damp cos(fl(x + pi/2)) = -damp sin(x) up to the rounding of the
    // argument (<= 1 ulp of x in the phase: 1e-7 relative on a GRADIENT) -- one range reduction serves both
    float sn, cs;
    nm_sincosf(xe, sn, cs);
    const float gm = (ga * (damp * cs) + gb * (-(damp * sn))) * sc;
    const float gt = gm * t_mean;
    a[0] += ax == 0 ? gm : 0.f; a[1] += ax == 1 ? gm : 0.f; a[2] += ax == 2 ? gm : 0.f;
    b[0] += ax == 0 ? gt : 0.f; b[1] += ax == 1 ? gt : 0.f; b[2] += ax == 2 ? gt : 0.f;
  }
  // view-direction rows: 12 pieces of 16 bytes per row, thread = (piece, row group)
  {
    const int q = tid % 12, grp = tid / 12;  // 21 row groups (252 threads)
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
    if (grp < 21)
      for (int sidx = grp; sidx < Sa; sidx += 21) s4 += *reinterpret_cast<const f32x4*>(gxd + ((size_t)r * Sa + sidx) * XD + 4 * q);
    if (grp < 21) {
      red[grp][4 * q] = s4[0]; red[grp][4 * q + 1] = s4[1]; red[grp][4 * q + 2] = s4[2]; red[grp][4 * q + 3] = s4[3];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    a[k] = wave_sum(a[k]);
    b[k] = wave_sum(b[k]);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      part[wave][k] = a[k];
      part[wave][3 + k] = b[k];
    }
  }
  __syncthreads();
  if (tid < 6) acc[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
  if (tid >= 64 && tid < 64 + 27) {
    const int c = tid - 64;
    float sum = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 21; ++g2) sum += red[g2][c];
    acc[6 + c] = sum;
  }
  __syncthreads();
  if (tid < 3) {
    const int ax = tid;
    const float v = rp[8 + ax];
    float gv = acc[3 + ax] + acc[6 + 24 + ax];
    for (int k = 0; k < 4; ++k) {
      const float sc = (float)(1 << k);
      gv += (acc[6 + k * 3 + ax] * nm_cosf(v * sc) + acc[6 + 12 + k * 3 + ax] * nm_cosf(v * sc + HALF_PI_F)) * sc;
    }
    g_o[(size_t)r * 3 + ax] = acc[ax];
    g_v[(size_t)r * 3 + ax] = gv;
  }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// one thread per ray: rgb_map = sum_s w_s c_s + (1 - sum_s w_s)   (white background, render_utils.py:224-225)
__global__ void inerf_composite_kernel(const float* __restrict__ logit, const float* __restrict__ sig, int ld, const float* __restrict__ z,
                                       const float* __restrict__ rays, int R, int S, int Sa, float* __restrict__ rgb_map,
                                       float* __restrict__ w_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* rp = rays + (size_t)r * 12;
  const float dn = sqrtf((rp[3] * rp[3] + rp[4] * rp[4]) + rp[5] * rp[5]);
  float T = 1.f, acc = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int s = 0; s < Sa; ++s) {
    const size_t n = (size_t)r * Sa + s;
    const float sg = fmaxf(sig[n * ld], 0.f);
    const float delta = (z[(size_t)r * (S + 1) + s + 1] - z[(size_t)r * (S + 1) + s]) * dn;
    const float alpha = 1.0f - expf(-sg * delta);
    const float w = alpha * T;
    if (w_out) w_out[n] = w;
    c0 += w * sigmoidf(logit[n * ld]); c1 += w * sigmoidf(logit[n * ld + 1]); c2 += w * sigmoidf(logit[n * ld + 2]);
    acc += w;
    T *= (1.0f - alpha) + 1e-10f;
  }
  rgb_map[(size_t)r * 3] = c0 + (1.0f - acc);
  rgb_map[(size_t)r * 3 + 1] = c1 + (1.0f - acc);
  rgb_map[(size_t)r * 3 + 2] = c2 + (1.0f - acc);
}

// backward of the above for upstream gradients G = d loss / d rgb_map (R,3) and, optionally, g_w = d loss / d weights
// (R,Sa) (the weights also feed the matching term of the refinement):
//   g_logit (n, ld) columns 0..2, g_sig (n, ld) column 0 (other columns zero), g_d (R,3) through delta = dz |d|
__global__ void inerf_composite_bwd_kernel(const float* __restrict__ logit, const float* __restrict__ sig, int ld, const float* __restrict__ z,
                                           const float* __restrict__ rays, const float* __restrict__ G, const float* __restrict__ g_w, int R,
                                           int S, int Sa, float* __restrict__ g_logit, float* __restrict__ g_sig, float* __restrict__ g_d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* rp = rays + (size_t)r * 12;
  const float dn = sqrtf((rp[3] * rp[3] + rp[4] * rp[4]) + rp[5] * rp[5]);
  const float G0 = G[(size_t)r * 3], G1 = G[(size_t)r * 3 + 1], G2 = G[(size_t)r * 3 + 2];
  // forward sweep: park T_s in g_sig column 1 (scratch, cleared below)
  float T = 1.f;
  for (int s = 0; s < Sa; ++s) {
    const size_t n = (size_t)r * Sa + s;
    const float sg = fmaxf(sig[n * ld], 0.f);
    const float delta = (z[(size_t)r * (S + 1) + s + 1] - z[(size_t)r * (S + 1) + s]) * dn;
    g_sig[n * ld + 1] = T;
    T *= (1.0f - (1.0f - expf(-sg * delta))) + 1e-10f;
  }
  // backward sweep: rgb_map = 1 + sum_s w_s (c_s - 1), w_s = alpha_s T_s, T_s = prod_{j<s} u_j, u = 1 - alpha + 1e-10
  float B = 0.f, gnorm = 0.f;  // B = sum_{s > j} q_s w_s
  for (int s = Sa - 1; s >= 0; --s) {
    const size_t n = (size_t)r * Sa + s;
    const float raw = sig[n * ld];
    const float sg = fmaxf(raw, 0.f);
    const float dz = z[(size_t)r * (S + 1) + s + 1] - z[(size_t)r * (S + 1) + s];
    const float delta = dz * dn;
    const float ex = expf(-sg * delta);
    const float alpha = 1.0f - ex;
    const float u = (1.0f - alpha) + 1e-10f;
    const float Ts = g_sig[n * ld + 1];
    const float w = alpha * Ts;
    const float c0 = sigmoidf(logit[n * ld]), c1 = sigmoidf(logit[n * ld + 1]), c2 = sigmoidf(logit[n * ld + 2]);
    float q = (G0 * (c0 - 1.0f) + G1 * (c1 - 1.0f)) + G2 * (c2 - 1.0f);  // d loss / d w_s
    if (g_w) q += g_w[n];
    g_logit[n * ld] = G0 * w * (c0 * (1.0f - c0));
    g_logit[n * ld + 1] = G1 * w * (c1 * (1.0f - c1));
    g_logit[n * ld + 2] = G2 * w * (c2 * (1.0f - c2));
    for (int c = 3; c < ld; ++c) g_logit[n * ld + c] = 0.f;
    const float g_alpha = q * Ts - B / u;
    B += q * w;
    g_sig[n * ld] = raw > 0.f ? g_alpha * (delta * ex) : 0.f;
    for (int c = 1; c < ld; ++c) g_sig[n * ld + c] = 0.f;
    gnorm += g_alpha * (sg * ex) * dz;
  }
  const float inv = dn > 0.f ? 1.0f / dn : 0.f;
  g_d[(size_t)r * 3] = gnorm * rp[3] * inv;
  g_d[(size_t)r * 3 + 1] = gnorm * rp[4] * inv;
  g_d[(size_t)r * 3 + 2] = gnorm * rp[5] * inv;
}

// Round 6: the same two passes on the FUSED fine field's own output rows (out4 [n, 4] = rgb logits | raw sigma, what nm_nerf_points_fwd*
// writes), one WAVEFRONT per ray instead of one thread: a lane takes the samples lane, lane + 64, ..., the transmittance is a prefix product
// over lanes (six shuffle steps per 64 samples, carry between chunks), the backward's B_s = sum_{j > s} q_j w_j a suffix sum the same way; one
// 16-byte load and one 16-byte store per sample.  The one-thread-per-ray kernels above walk 128 dependent iterations on 75 wavefronts of a
// 1024-SIMD chip (58 + 66 us per step at 4800 rays) and need sigma copied out of / back into column 3 around them (two strided torch copies).
// The products / sums associate differently from the sequential loops (a scan tree): results agree to rounding.
constexpr int CMP_MAX_CHUNKS = 16;  // 64 samples each
__device__ __forceinline__ float shfl_up_f(float v, int d) { return __shfl_up(v, d, 64); }
__device__ __forceinline__ float shfl_down_f(float v, int d) { return __shfl_down(v, d, 64); }

struct CmpSample {
  float raw, sg, dz, delta, ex, alpha, u;
  f32x4 o;
};
__device__ __forceinline__ CmpSample cmp_sample(const float* __restrict__ out4, const float* __restrict__ zr, size_t n, int sidx, bool valid, float dn) {
  CmpSample c;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  c.o = valid ? *reinterpret_cast<const f32x4*>(out4 + n * 4) : zero4;
  c.raw = c.o[3];
  c.sg = fmaxf(c.raw, 0.f);
  c.dz = valid ? zr[sidx + 1] - zr[sidx] : 0.f;
  c.delta = c.dz * dn;
  c.ex = expf(-c.sg * c.delta);
  c.alpha = valid ? 1.0f - c.ex : 0.f;
  c.u = (1.0f - c.alpha) + 1e-10f;
  return c;
}
// inclusive prefix product over the lanes; returns it, *excl = the exclusive one (1 in lane 0)
__device__ __forceinline__ float prefix_prod(float u, int lane, float* excl) {
  float p = u;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float t = shfl_up_f(p, d);
    if (lane >= d) p *= t;
  }
  const float e = shfl_up_f(p, 1);
  *excl = lane == 0 ? 1.0f : e;
  return p;
}

__global__ void __launch_bounds__(256) inerf_composite4_kernel(const float* __restrict__ out4, const float* __restrict__ z,
                                                                const float* __restrict__ rays, int R, int S, int Sa, float* __restrict__ rgb_map,
                                                                float* __restrict__ w_out) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* rp = rays + (size_t)r * 12;
  const float* zr = z + (size_t)r * (S + 1);
  const float dn = sqrtf((rp[3] * rp[3] + rp[4] * rp[4]) + rp[5] * rp[5]);
  float T0 = 1.f, acc = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
  for (int base = 0; base < Sa; base += 64) {
    const int sidx = base + lane;
    const bool valid = sidx < Sa;
    const size_t n = (size_t)r * Sa + sidx;
    const CmpSample c = cmp_sample(out4, zr, n, sidx, valid, dn);
    float excl;
    const float p = prefix_prod(c.u, lane, &excl);
    const float w = c.alpha * (T0 * excl);
    if (w_out && valid) w_out[n] = w;
    c0 += w * sigmoidf(c.o[0]); c1 += w * sigmoidf(c.o[1]); c2 += w * sigmoidf(c.o[2]);
    acc += w;
    T0 *= __shfl(p, 63, 64);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_xor(acc, o, 64); c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); c2 += __shfl_xor(c2, o, 64);
  }
  if (lane == 0) {
    rgb_map[(size_t)r * 3] = c0 + (1.0f - acc);
    rgb_map[(size_t)r * 3 + 1] = c1 + (1.0f - acc);
    rgb_map[(size_t)r * 3 + 2] = c2 + (1.0f - acc);
  }
}

__global__ void __launch_bounds__(256) inerf_composite4_bwd_kernel(const float* __restrict__ out4, const float* __restrict__ z,
                                                                    const float* __restrict__ rays, const float* __restrict__ G,
                                                                    const float* __restrict__ g_w, int R, int S, int Sa, float* __restrict__ g4,
                                                                    float* __restrict__ g_d) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* rp = rays + (size_t)r * 12;
  const float* zr = z + (size_t)r * (S + 1);
  const float dn = sqrtf((rp[3] * rp[3] + rp[4] * rp[4]) + rp[5] * rp[5]);
  const float G0 = G[(size_t)r * 3], G1 = G[(size_t)r * 3 + 1], G2 = G[(size_t)r * 3 + 2];
  // forward sweep: the transmittance at the start of every chunk
  float carry[CMP_MAX_CHUNKS];
  const int nchunk = (Sa + 63) / 64;
  {
    float T0 = 1.f;
#pragma unroll
    for (int k = 0; k < CMP_MAX_CHUNKS; ++k) {
      carry[k] = T0;
      if (k < nchunk) {
        const int sidx = k * 64 + lane;
        const CmpSample c = cmp_sample(out4, zr, (size_t)r * Sa + sidx, sidx, sidx < Sa, dn);
        float excl;
        T0 *= __shfl(prefix_prod(c.u, lane, &excl), 63, 64);
      }
    }
  }
  // backward sweep: rgb_map = 1 + sum_s w_s (c_s - 1), w_s = alpha_s T_s, T_s = prod_{j<s} u_j, u = 1 - alpha + 1e-10
  float Bc = 0.f, gnorm = 0.f;  // Bc = sum of q w over the chunks behind this one
#pragma unroll
  for (int k = CMP_MAX_CHUNKS - 1; k >= 0; --k) {
    if (k >= nchunk) continue;
    const int sidx = k * 64 + lane;
    const bool valid = sidx < Sa;
    const size_t n = (size_t)r * Sa + sidx;
    const CmpSample c = cmp_sample(out4, zr, n, sidx, valid, dn);
    float excl;
    prefix_prod(c.u, lane, &excl);
    const float Ts = carry[k] * excl;
    const float w = c.alpha * Ts;
    const float s0 = sigmoidf(c.o[0]), s1 = sigmoidf(c.o[1]), s2 = sigmoidf(c.o[2]);
    float q = (G0 * (s0 - 1.0f) + G1 * (s1 - 1.0f)) + G2 * (s2 - 1.0f);  // d loss / d w_s
    if (g_w && valid) q += g_w[n];
    const float qw = valid ? q * w : 0.f;
    float sfx = qw;  // inclusive suffix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const float t = shfl_down_f(sfx, d);
      if (lane + d < 64) sfx += t;
    }
    const float nxt = shfl_down_f(sfx, 1);
    const float B = Bc + (lane == 63 ? 0.f : nxt);
    const float g_alpha = q * Ts - B / c.u;
    if (valid) {
      f32x4 g;
      g[0] = G0 * w * (s0 * (1.0f - s0));
      g[1] = G1 * w * (s1 * (1.0f - s1));
      g[2] = G2 * w * (s2 * (1.0f - s2));
      g[3] = c.raw > 0.f ? g_alpha * (c.delta * c.ex) : 0.f;
      *reinterpret_cast<f32x4*>(g4 + n * 4) = g;
      gnorm += g_alpha * (c.sg * c.ex) * c.dz;
    }
    Bc += __shfl(sfx, 0, 64);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gnorm += __shfl_xor(gnorm, o, 64);
  if (lane == 0) {
    const float inv = dn > 0.f ? 1.0f / dn : 0.f;
    g_d[(size_t)r * 3] = gnorm * rp[3] * inv;
    g_d[(size_t)r * 3 + 1] = gnorm * rp[4] * inv;
    g_d[(size_t)r * 3 + 2] = gnorm * rp[5] * inv;
  }
}

// Matching term of the refinement (nerfmatch_evaluator.py:420-428): per ray, pt_feat = sum_s w_s feats_s and
// pts = sum_s w_s mean_s, the Gaussian means o + t_mean d being those of the sampler, i.e. constants.
// One workgroup per ray, thread = feature channel (C <= blockDim); threads 0..2 also do the three coordinates.
__global__ void __launch_bounds__(256) inerf_ray_sums_kernel(const float* __restrict__ w, const float* __restrict__ feats, int C,
                                                              const float* __restrict__ rays, const float* __restrict__ z, int S, int Sa,
                                                              float* __restrict__ pt_feat, float* __restrict__ pts) {
  const int r = blockIdx.x, c = threadIdx.x;
  const float* rp = rays + (size_t)r * 12;
  const float* wr = w + (size_t)r * Sa;
  if (c < C) {
    float acc = 0.f;
    for (int s = 0; s < Sa; ++s) acc += wr[s] * feats[((size_t)r * Sa + s) * C + c];
    pt_feat[(size_t)r * C + c] = acc;
  }
  if (c < 3) {
    float acc = 0.f;
    for (int s = 0; s < Sa; ++s) {
      const Gauss g = frustum(z[(size_t)r * (S + 1) + s], z[(size_t)r * (S + 1) + s + 1], rp + 3, rp[11]);
      acc += wr[s] * (rp[3 + c] * g.t_mean + rp[c]);
    }
    pts[(size_t)r * 3 + c] = acc;
  }
}

// backward: g_feats[n][c] = w_n g_ptfeat[r][c];  g_w[n] = <g_ptfeat[r], feats[n]> + <g_pts[r], mean_n>.
// One wavefront per sample row (4 rows per workgroup), lanes stride the channels.
__global__ void __launch_bounds__(256) inerf_ray_sums_bwd_kernel(const float* __restrict__ w, const float* __restrict__ feats, int C,
                                                                  const float* __restrict__ rays, const float* __restrict__ z,
                                                                  const float* __restrict__ g_ptfeat, const float* __restrict__ g_pts, int R,
                                                                  int S, int Sa, float* __restrict__ g_feats, float* __restrict__ g_w) {
  const size_t n = (size_t)blockIdx.x * 4 + threadIdx.x / 64;
  const int lane = threadIdx.x % 64;
  if (n >= (size_t)R * Sa) return;
  const int r = (int)(n / Sa), s = (int)(n % Sa);
  const float wn = w[n];
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float g = g_ptfeat[(size_t)r * C + c];
    dot += g * feats[n * C + c];
    if (g_feats) g_feats[n * C + c] = wn * g;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (lane == 0) {
    const float* rp = rays + (size_t)r * 12;
    const Gauss g = frustum(z[(size_t)r * (S + 1) + s], z[(size_t)r * (S + 1) + s + 1], rp + 3, rp[11]);
#pragma unroll
    for (int a = 0; a < 3; ++a) dot += g_pts[(size_t)r * 3 + a] * (rp[3 + a] * g.t_mean + rp[a]);
    g_w[n] = dot;
  }
}

}  // namespace

extern "C" int nm_inerf_encode(const float* rays, const float* z, int R, int S, int S_act, const float* app_row, float* xi, float* xd,
                               nmStream_t stream) {
  NM_CHECK_ARG(rays && z && xi && xd && R > 0 && S > 0 && S_act > 0 && S_act <= S);
  const size_t total = (size_t)R * S_act * 16;
  inerf_encode_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(rays, z, R, S, S_act, app_row, xi, xd);
  return nm_launch_status();
}

extern "C" int nm_inerf_encode_bwd(const float* rays, const float* z, int R, int S, int S_act, const float* g_xi, const float* g_xd,
                                   float* g_o, float* g_v, nmStream_t stream) {
  NM_CHECK_ARG(rays && z && g_xi && g_xd && g_o && g_v && R > 0 && S > 0 && S_act > 0 && S_act <= S);
  if (S_act > ENC_MAX_S) return NM_ERR_UNSUPPORTED;
  inerf_encode_bwd_kernel<false><<<R, 256, 0, (hipStream_t)stream>>>(rays, z, R, S, S_act, g_xi, g_xi, g_xd, g_o, g_v);
  return nm_launch_status();
}

extern "C" int nm_inerf_encode_bwd2(const float* rays, const float* z, int R, int S, int S_act, const float* g_xi_a, const float* g_xi_b,
                                    const float* g_xd, float* g_o, float* g_v, nmStream_t stream) {
  NM_CHECK_ARG(rays && z && g_xi_a && g_xi_b && g_xd && g_o && g_v && R > 0 && S > 0 && S_act > 0 && S_act <= S);
  if (S_act > ENC_MAX_S) return NM_ERR_UNSUPPORTED;
  inerf_encode_bwd_kernel<true><<<R, 256, 0, (hipStream_t)stream>>>(rays, z, R, S, S_act, g_xi_a, g_xi_b, g_xd, g_o, g_v);
  return nm_launch_status();
}

// d loss / d pose (normalised-scene c2w, row-major 4x4; the homogeneous row gets zeros) from the per-ray gradients of the ray
// bundle: origin o = pose[:3, 3] for every ray; view direction = normalise(pose[:3, :3] . Kinv . (x, y, 1)) on the sub-sampled pixel
// grid (the reference's gen_rays, nerfmatch_evaluator.py:268-286: rays[:, 3:6] and rays[:, 8:11] are the same tensor, so g_view takes
// both gradients).  One workgroup: 12 sums over R rays -- replaces a dozen tiny autograd launches at the tail of every step.
struct PoseGradArgs {
  float kinv[9], rot[9];
  int H, W, ds;
};
__global__ void __launch_bounds__(256) inerf_pose_grad_kernel(PoseGradArgs a, const float* __restrict__ g_o, const float* __restrict__ g_v,
                                                              const float* __restrict__ g_d, int R, float* __restrict__ g_pose) {
  __shared__ float red[12][256];
  const int tid = threadIdx.x;
  const int nx = (a.W - a.ds / 2 + a.ds - 1) / a.ds;
  float acc[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = tid; r < R; r += 256) {
    const float x = (float)(a.ds / 2 + (r % nx) * a.ds), y = (float)(a.ds / 2 + (r / nx) * a.ds);
    float dc[3], rd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) dc[i] = (a.kinv[3 * i] * x + a.kinv[3 * i + 1] * y) + a.kinv[3 * i + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i) rd[i] = (a.rot[3 * i] * dc[0] + a.rot[3 * i + 1] * dc[1]) + a.rot[3 * i + 2] * dc[2];
    const float nrm = sqrtf((rd[0] * rd[0] + rd[1] * rd[1]) + rd[2] * rd[2]);
    float v[3], gv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i] = rd[i] / nrm;
      gv[i] = g_v[(size_t)r * 3 + i] + (g_d ? g_d[(size_t)r * 3 + i] : 0.f);
    }
    const float dot = (v[0] * gv[0] + v[1] * gv[1]) + v[2] * gv[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float gr = (gv[i] - v[i] * dot) / nrm;  // d loss / d raydir_i
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[3 * i + j] += gr * dc[j];
      acc[9 + i] += g_o[(size_t)r * 3 + i];
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) red[k][tid] = acc[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) {
#pragma unroll
      for (int k = 0; k < 12; ++k) red[k][tid] += red[k][tid + off];
    }
    __syncthreads();
  }
  if (tid < 16) {
    const int i = tid >> 2, j = tid & 3;
    g_pose[tid] = i == 3 ? 0.f : (j < 3 ? red[3 * i + j][0] : red[9 + i][0]);
  }
}

extern "C" int nm_inerf_pose_grad(const float* Kinv_host, const float* pose_host, int H, int W, int ds, const float* g_o, const float* g_v,
                                  const float* g_d, int R, float* g_pose, nmStream_t stream) {
  NM_CHECK_ARG(Kinv_host && pose_host && g_o && g_v && g_pose && H > 0 && W > 0 && ds > 0 && R > 0);
  PoseGradArgs a;
  for (int i = 0; i < 9; ++i) a.kinv[i] = Kinv_host[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) a.rot[3 * i + j] = pose_host[4 * i + j];
  a.H = H; a.W = W; a.ds = ds;
  inerf_pose_grad_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a, g_o, g_v, g_d, R, g_pose);
  return nm_launch_status();
}

extern "C" int nm_inerf_composite_ex(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays, int R, int S,
                                     int S_act, float* rgb_map, float* weights, nmStream_t stream) {
  NM_CHECK_ARG(logit_rgb && sigma_raw && z && rays && rgb_map && R > 0 && S > 0 && S_act > 0 && S_act <= S && ld >= 3);
  inerf_composite_kernel<<<(R + 63) / 64, 64, 0, (hipStream_t)stream>>>(logit_rgb, sigma_raw, ld, z, rays, R, S, S_act, rgb_map, weights);
  return nm_launch_status();
}

extern "C" int nm_inerf_composite(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays, int R, int S,
                                  int S_act, float* rgb_map, nmStream_t stream) {
  return nm_inerf_composite_ex(logit_rgb, sigma_raw, ld, z, rays, R, S, S_act, rgb_map, nullptr, stream);
}

extern "C" int nm_inerf_composite4(const float* out4, const float* z, const float* rays, int R, int S, int S_act, float* rgb_map, float* weights,
                                   nmStream_t stream) {
  NM_CHECK_ARG(out4 && z && rays && rgb_map && R > 0 && S > 0 && S_act > 0 && S_act <= S);
  inerf_composite4_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(out4, z, rays, R, S, S_act, rgb_map, weights);
  return nm_launch_status();
}

extern "C" int nm_inerf_composite4_bwd(const float* out4, const float* z, const float* rays, const float* g_rgb_map, const float* g_weights, int R,
                                       int S, int S_act, float* g_out4, float* g_d, nmStream_t stream) {
  NM_CHECK_ARG(out4 && z && rays && g_rgb_map && g_out4 && g_d && R > 0 && S > 0 && S_act > 0 && S_act <= S);
  if (S_act > 64 * CMP_MAX_CHUNKS) return NM_ERR_UNSUPPORTED;
  inerf_composite4_bwd_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(out4, z, rays, g_rgb_map, g_weights, R, S, S_act, g_out4, g_d);
  return nm_launch_status();
}

extern "C" int nm_inerf_composite_bwd_ex(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays,
                                         const float* g_rgb_map, const float* g_weights, int R, int S, int S_act, float* g_logit,
                                         float* g_sigma, float* g_d, nmStream_t stream) {
  NM_CHECK_ARG(logit_rgb && sigma_raw && z && rays && g_rgb_map && g_logit && g_sigma && g_d && R > 0 && S > 0 && S_act > 0 && S_act <= S &&
               ld >= 3);
  inerf_composite_bwd_kernel<<<(R + 63) / 64, 64, 0, (hipStream_t)stream>>>(logit_rgb, sigma_raw, ld, z, rays, g_rgb_map, g_weights, R, S,
                                                                               S_act, g_logit, g_sigma, g_d);
  return nm_launch_status();
}

extern "C" int nm_inerf_composite_bwd(const float* logit_rgb, const float* sigma_raw, int ld, const float* z, const float* rays,
                                      const float* g_rgb_map, int R, int S, int S_act, float* g_logit, float* g_sigma, float* g_d,
                                      nmStream_t stream) {
  return nm_inerf_composite_bwd_ex(logit_rgb, sigma_raw, ld, z, rays, g_rgb_map, nullptr, R, S, S_act, g_logit, g_sigma, g_d, stream);
}

extern "C" int nm_inerf_ray_sums(const float* weights, const float* feats, int C, const float* rays, const float* z, int R, int S, int S_act,
                                 float* pt_feat, float* pts, nmStream_t stream) {
  NM_CHECK_ARG(weights && feats && rays && z && pt_feat && pts && R > 0 && S > 0 && S_act > 0 && S_act <= S && C >= 3 && C <= 256);
  inerf_ray_sums_kernel<<<R, 256, 0, (hipStream_t)stream>>>(weights, feats, C, rays, z, S, S_act, pt_feat, pts);
  return nm_launch_status();
}

extern "C" int nm_inerf_ray_sums_bwd(const float* weights, const float* feats, int C, const float* rays, const float* z, const float* g_pt_feat,
                                     const float* g_pts, int R, int S, int S_act, float* g_feats, float* g_weights, nmStream_t stream) {
  NM_CHECK_ARG(weights && feats && rays && z && g_pt_feat && g_pts && g_weights && R > 0 && S > 0 && S_act > 0 && S_act <= S && C > 0);
  const size_t rows = (size_t)R * S_act;
  inerf_ray_sums_bwd_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(weights, feats, C, rays, z, g_pt_feat, g_pts, R, S,
                                                                                       S_act, g_feats, g_weights);
  return nm_launch_status();
}
