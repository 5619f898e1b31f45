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

// one workgroup (128 threads) per ray, thread = sample; g_o[r], g_v[r] = d loss / d origin, d loss / d view direction
template <bool TWO>
__global__ void __launch_bounds__(128) inerf_encode_bwd_kernel(const float* __restrict__ rays, const float* __restrict__ z, int R, int S,
                                                                int Sa, const float* __restrict__ gxi, const float* __restrict__ gxi2,
                                                                const float* __restrict__ gxd, float* __restrict__ g_o, float* __restrict__ g_v) {
  __shared__ float acc[6 + 27];
  const int r = blockIdx.x, tid = threadIdx.x;
  if (tid < 33) acc[tid] = 0.f;
  __syncthreads();
  const float* rp = rays + (size_t)r * 12;
  for (int s = tid; s < Sa; s += (int)blockDim.x) {
    const size_t n = (size_t)r * Sa + s;
    const Gauss g = frustum(z[(size_t)r * (S + 1) + s], z[(size_t)r * (S + 1) + s + 1], rp + 3, rp[11]);
    float gm[3] = {0.f, 0.f, 0.f};  // d loss / d mean
    const float* gi = gxi + n * XI;
    const float* gi2 = gxi2 + n * XI;  // TWO: second contribution to d loss / d xi (the skip connection's, nm_nerf_points_bwd_bf16x3)
#pragma unroll 1
    for (int i = 0; i < 15; ++i) {
      const float sc = (float)(1 << i);
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        const float mean = rp[ax] + g.t_mean * rp[8 + ax];
        const float xe = mean * sc;
        const float damp = expf(-0.5f * (g.var[ax] * (sc * sc)));
        // d/dx [damp sin(x)] = damp cos(x);  d/dx [damp sin(fl(x + pi/2))] = damp cos(fl(x + pi/2)) = -damp sin(x) up to the rounding of the
        // argument (<= 1 ulp of x in the phase: 1e-7 relative on a GRADIENT) -- one range reduction serves both
        float sn, cs;
        nm_sincosf(xe, sn, cs);
        const float d0 = damp * cs, d1 = -(damp * sn);
        float ga = gi[i * 3 + ax], gb = gi[45 + i * 3 + ax];
        if constexpr (TWO) { ga += gi2[i * 3 + ax]; gb += gi2[45 + i * 3 + ax]; }
        gm[ax] += (ga * d0 + gb * d1) * sc;
      }
    }
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      atomicAdd(&acc[ax], gm[ax]);
      atomicAdd(&acc[3 + ax], gm[ax] * g.t_mean);
    }
    const float* gd = gxd + n * XD;
    for (int c = 0; c < 27; ++c) atomicAdd(&acc[6 + c], gd[c]);
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
  inerf_encode_bwd_kernel<false><<<R, S_act <= 64 ? 64 : 128, 0, (hipStream_t)stream>>>(rays, z, R, S, S_act, g_xi, g_xi, g_xd, g_o, g_v);
  return nm_launch_status();
}

extern "C" int nm_inerf_encode_bwd2(const float* rays, const float* z, int R, int S, int S_act, const float* g_xi_a, const float* g_xi_b,
                                    const float* g_xd, float* g_o, float* g_v, nmStream_t stream) {
  NM_CHECK_ARG(rays && z && g_xi_a && g_xi_b && g_xd && g_o && g_v && R > 0 && S > 0 && S_act > 0 && S_act <= S);
  inerf_encode_bwd_kernel<true><<<R, S_act <= 64 ? 64 : 128, 0, (hipStream_t)stream>>>(rays, z, R, S, S_act, g_xi_a, g_xi_b, g_xd, g_o, g_v);
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
