// Ray generation, stratified sampling and hierarchical re-sampling.
// All three are tiny HBM/latency-bound kernels: one thread per ray (or per fence post), coalesced
// row-major writes.  They exist so that a localisation step never leaves the device; the reference
// builds the full-resolution ray grid on the CPU and copies 1/64 of it (render_utils.py:56-78).
#include "common.h"
#include <cstdlib>

namespace {

constexpr int RAYGEN_MAXQ = 32;  // poses per launch (kernel-argument space: 32 x 48 bytes)
struct RayGenArgs {
  float kinv[9];
  int H, W, ds, nx, ny;
  float near_plane;
  float poses[RAYGEN_MAXQ][12];  // rows 0..2 of the normalised camera-to-world matrices
};
struct RayGenPose {  // kinv + the pose of the query this workgroup row (blockIdx.y) belongs to
  float kinv[9];
  float c2w[12];
};

__device__ __forceinline__ void view_dir(const RayGenPose& a, int px, int py, float (&v)[3]) {
  const float x = (float)px, y = (float)py;
  float cam[3], wd[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) cam[j] = NM_FMA(1.0f, a.kinv[3 * j + 2], NM_FMA(y, a.kinv[3 * j + 1], x * a.kinv[3 * j]));
#pragma unroll
  for (int j = 0; j < 3; ++j)
    wd[j] = NM_FMA(cam[2], a.c2w[4 * j + 2], NM_FMA(cam[1], a.c2w[4 * j + 1], cam[0] * a.c2w[4 * j]));
  const float n = sqrtf(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
#pragma unroll
  for (int j = 0; j < 3; ++j) v[j] = wd[j] / n;
}

// grid: one thread per sub-sampled ray.  Besides its own pixel every thread evaluates the direction of the
// pixel one image row below (cone radius = distance between unit directions of row neighbours) and every
// thread of the FULL image contributes to the far-plane validity flag: the reference checks the discriminant
// for all H*W pixels, so the sub-sampled threads stride over the ds x ds block they represent.
__global__ void raygen_kernel(RayGenArgs a, float* __restrict__ rays, int* __restrict__ fallback) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;  // blockIdx.y = query of the batch
  if (idx >= a.nx * a.ny) return;
  RayGenPose pz;
#pragma unroll
  for (int i = 0; i < 9; ++i) pz.kinv[i] = a.kinv[i];
#pragma unroll
  for (int i = 0; i < 12; ++i) pz.c2w[i] = a.poses[q][i];
  rays += (size_t)q * a.nx * a.ny * 12;
  fallback += q;
  const int ix = idx % a.nx, iy = idx / a.nx;
  const int px = a.ds / 2 + ix * a.ds, py = a.ds / 2 + iy * a.ds;
  const float o[3] = {pz.c2w[3], pz.c2w[7], pz.c2w[11]};
  const float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];

  // validity of every full-resolution pixel of this thread's block (pixels [ix*ds, ix*ds+ds) x [iy*ds, ...),
  // edge blocks extend to the image border)
  const int x0 = ix * a.ds, y0 = iy * a.ds;
  const int x1 = (ix == a.nx - 1) ? a.W : x0 + a.ds, y1 = (iy == a.ny - 1) ? a.H : y0 + a.ds;
  bool bad = false;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      float v[3];
      view_dir(pz, xx, yy, v);
      const float od = o[0] * v[0] + o[1] * v[1] + o[2] * v[2];
      const float dd = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      const float disc = od * od + (1.0f - oo) * dd;
      bad |= !(disc >= 0.0f);
    }
  if (bad) atomicOr(fallback, 1);

  float v[3], vn[3];
  view_dir(pz, px, py, v);
  // neighbour along image rows (axis 0); the last row re-uses the difference of rows H-2 / H-1
  if (py + 1 < a.H)
    view_dir(pz, px, py + 1, vn);
  else
    view_dir(pz, px, py - 1, vn);
  const float e0 = v[0] - vn[0], e1 = v[1] - vn[1], e2 = v[2] - vn[2];
  const float step = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float radius = step * 2.0f / 3.4641016151377544f;
  const float od = o[0] * v[0] + o[1] * v[1] + o[2] * v[2];
  const float dd = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const float disc = od * od + (1.0f - oo) * dd;
  const float far_plane = (sqrtf(disc) - od) / dd;
  float* r = rays + (size_t)idx * 12;
  r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
  r[3] = v[0]; r[4] = v[1]; r[5] = v[2];
  r[6] = a.near_plane; r[7] = far_plane;
  r[8] = v[0]; r[9] = v[1]; r[10] = v[2];
  r[11] = radius;
}

// The same rays and the same flag with one thread per FULL-RESOLUTION pixel (round 5).  raygen_kernel's threads walk the ds x ds pixels their ray
// stands for -- 64 directions with a square root and three divisions each, on 4800 threads: 21 us of pure latency for one query.  Here
// every pixel's discriminant has its own thread (one atomic per wavefront that sees a negative one) and the threads that sit on a ray's
// centre pixel write the ray: the expressions are raygen_kernel's (bit-identical rays), the pixel set is the whole image either way.
__global__ void __launch_bounds__(256) raygen_pixels_kernel(RayGenArgs a, float* __restrict__ rays, int* __restrict__ fallback) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  const bool inside = idx < a.H * a.W;
  RayGenPose pz;
#pragma unroll
  for (int i = 0; i < 9; ++i) pz.kinv[i] = a.kinv[i];
#pragma unroll
  for (int i = 0; i < 12; ++i) pz.c2w[i] = a.poses[q][i];
  const int xx = inside ? idx % a.W : 0, yy = inside ? idx / a.W : 0;
  const float o[3] = {pz.c2w[3], pz.c2w[7], pz.c2w[11]};
  const float oo = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
  float v[3];
  view_dir(pz, xx, yy, v);
  const float od = o[0] * v[0] + o[1] * v[1] + o[2] * v[2];
  const float dd = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const float disc = od * od + (1.0f - oo) * dd;
  const bool bad = inside && !(disc >= 0.0f);
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(fallback + q, 1);
  const int h = a.ds / 2;
  if (!inside || xx % a.ds != h || yy % a.ds != h) return;
  const int ix = xx / a.ds, iy = yy / a.ds;
  if (ix >= a.nx || iy >= a.ny) return;
  float vn[3];
  // neighbour along image rows (axis 0); the last row re-uses the difference of rows H-2 / H-1
  if (yy + 1 < a.H)
    view_dir(pz, xx, yy + 1, vn);
  else
    view_dir(pz, xx, yy - 1, vn);
  const float e0 = v[0] - vn[0], e1 = v[1] - vn[1], e2 = v[2] - vn[2];
  const float step = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
  const float radius = step * 2.0f / 3.4641016151377544f;
  const float far_plane = (sqrtf(disc) - od) / dd;
  float* r = rays + ((size_t)q * a.nx * a.ny + (size_t)iy * a.nx + ix) * 12;
  r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
  r[3] = v[0]; r[4] = v[1]; r[5] = v[2];
  r[6] = a.near_plane; r[7] = far_plane;
  r[8] = v[0]; r[9] = v[1]; r[10] = v[2];
  r[11] = radius;
}

__global__ void far_fallback_kernel(float* __restrict__ rays, const int* __restrict__ fallback, int R) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
  if (idx < R && fallback[q]) rays[((size_t)q * R + idx) * 12 + 7] = 1.0f;
}

// t = near*(1-u) + far*u with u = linspace(0,1,S+1); stratified jitter between interval mid points.
__global__ void sample_coarse_kernel(const float* __restrict__ rays, const float* __restrict__ t_rand, int R, int S,
                                     float* __restrict__ t_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = S + 1;
  if (idx >= R * n) return;
  const int ray = idx / n, k = idx % n;
  const float near_plane = rays[(size_t)ray * 12 + 6], far_plane = rays[(size_t)ray * 12 + 7];
  // torch.linspace(0,1,n): step = 1/(n-1); second half is computed from the end (end - step*(n-1-i))
  const float stepu = 1.0f / (float)S;
  auto lin = [&](int i) -> float { return (i < n / 2) ? stepu * (float)i : 1.0f - stepu * (float)(n - 1 - i); };
  auto tv = [&](int i) -> float {
    const float u = lin(i);
    return near_plane * (1.0f - u) + far_plane * u;
  };
  const float tk = tv(k);
  const float lo = (k == 0) ? tk : 0.5f * (tk + tv(k - 1));
  const float hi = (k == n - 1) ? tk : 0.5f * (tv(k + 1) + tk);
  t_out[idx] = lo + (hi - lo) * t_rand[idx];
}

// One workgroup per ray.  Thread j < S+1 produces fence post j.
//   blur:  w'_i = 0.5*(max(w_{i-1},w_i) + max(w_i,w_{i+1})) + padding (edge-replicated)
//   pdf/cdf: sequential fp64 accumulation rounded to fp32 per element (what torch.cumsum does on the CPU)
//   u_j = min(j/n + j/n + jitter_j, 1-eps)   [randomized]   or linspace(0, 1-eps, n)[j]
//   interval: literal max / min selects over all cdf entries (no monotonicity assumption)
template <int MAXN>
__global__ void resample_kernel(const float* __restrict__ t_in, const float* __restrict__ weights,
                                const float* __restrict__ jitter, float jscale, int S, float padding, int randomized,
                                float* __restrict__ t_out, int* __restrict__ tail_flag) {
  __shared__ float s_w[MAXN], s_cdf[MAXN + 1], s_bins[MAXN + 1];
  __shared__ float s_sum, s_addw;
  const int ray = blockIdx.x, j = threadIdx.x, n = S + 1;
  const float* w = weights + (size_t)ray * S;
  if (j < S) {
    const float wm = w[j > 0 ? j - 1 : 0], wc = w[j], wp = w[j < S - 1 ? j + 1 : S - 1];
    s_w[j] = 0.5f * (fmaxf(wm, wc) + fmaxf(wc, wp)) + padding;
  }
  if (j < n) s_bins[j] = t_in[(size_t)ray * n + j];
  __syncthreads();
  // torch.sum over the last dim of a contiguous fp32 row is a vectorised cascade (near-pairwise) in ATen; its order cannot be
  // reproduced portably, so the sum is taken in fp64 and rounded ONCE: the correctly rounded value, <= 1 ulp from ATen's and the
  // closest one can get to the exact fence posts (round 4: against the fp64 evaluation the sequential fp32 sum of rounds 1-3
  // was 2x farther than the reference's own fp32 run, tests/test_resample_truth_gpu.py).
  // S <= 64 (round 5): one wavefront reduction instead of 64 dependent additions of thread 0 -- 4 k cycles on the critical path of every
  // ray.  The addends are fp32 values in [padding, ~1]: every partial sum is exact in fp64, so the order cannot change the result.
  double acc64 = 0.0;
  if (S <= 64 && j < 64) {
    acc64 = j < S ? (double)s_w[j] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc64 += __shfl_xor(acc64, o, 64);
  }
  if (j == 0) {
    if (S > 64)
      for (int i = 0; i < S; ++i) acc64 += (double)s_w[i];
    const float acc = (float)acc64;
    const float pad = fmaxf(0.f, 1e-5f - acc);
    s_sum = acc + pad;
    const float addw = pad / (float)S;
    s_addw = addw;
    s_cdf[0] = 0.f;
    s_cdf[S] = 1.0f;
    if (S > 64) {
      double c = 0.0;
      for (int i = 0; i < S - 1; ++i) {
        const float pdf = (s_w[i] + addw) / s_sum;
        c += (double)pdf;
        s_cdf[i + 1] = fminf(1.0f, (float)c);
      }
    }
  }
  __syncthreads();
  if (S <= 64) {
    // S <= 64: the cumulative sum as ONE wavefront scan in fp64 instead of 63 dependent additions of thread 0.  The pdf
    // entries are fp32 values >= padding / sum, so every partial sum is exact in fp64 and the order of the additions
    // cannot change the result (the sequential accumulation of torch.cumsum included).
    if (j < 64) {
      double c = (j < S - 1) ? (double)((s_w[j] + s_addw) / s_sum) : 0.0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(c, o, 64);
        if (j >= o) c += up;
      }
      if (j < S - 1) s_cdf[j + 1] = fminf(1.0f, (float)c);
    }
    __syncthreads();
  }
  if (j >= n) return;
  const float one_m_eps = 1.0f - 1.1920928955078125e-07f;
  float u;
  if (randomized) {
    const float base = (float)j * (float)(1.0 / (double)n);
    u = fminf((base + base) + jitter[(size_t)ray * n + j] * jscale, one_m_eps);  // (jscale 1: x * 1 is x)
    // NM_NERF_ZERO_TAIL premise: the fence posts j > S/2 all sit at u = 1 - eps (true for every jitter >= 0) and therefore
    // coincide.  A caller-supplied jitter that breaks it raises the flag; nm_nerf_fwd_bf16x3_ex then evaluates every sample.
    if (tail_flag && j >= S / 2 + 1 && u != one_m_eps) atomicOr(tail_flag, 1);
  } else {
    // torch.linspace(0, 1-eps, n)
    const float st = one_m_eps / (float)(n - 1);
    u = (j < n / 2) ? st * (float)j : one_m_eps - st * (float)(n - 1 - j);
  }
  float x0 = s_bins[0], x1 = s_bins[n - 1], y0 = s_cdf[0], y1 = s_cdf[n - 1];
  float bx0 = s_bins[0], by0 = s_cdf[0], bx1 = s_bins[n - 1], by1 = s_cdf[n - 1];
  for (int i = 0; i < n; ++i) {
    const float c = s_cdf[i], b = s_bins[i];
    const bool m = u >= c;
    bx0 = fmaxf(bx0, m ? b : x0);
    by0 = fmaxf(by0, m ? c : y0);
    bx1 = fminf(bx1, m ? x1 : b);
    by1 = fminf(by1, m ? y1 : c);
  }
  float fr = (u - by0) / (by1 - by0);
  if (fr != fr) fr = 0.f;  // nan_to_num(nan=0); +-inf are clipped below
  fr = fminf(fmaxf(fr, 0.f), 1.f);
  t_out[(size_t)ray * n + j] = bx0 + fr * (bx1 - bx0);
}

// The same arithmetic for rows of at most 64 intervals with SEVERAL rays per workgroup (round 5).  resample_kernel gives a ray S + 1 = 65 threads,
// i.e. two wavefronts of which the second works for ONE fence post -- and the 65-step interval search costs a wavefront the same whether one
// lane or 64 take part.  Here wavefront v < RPW owns ray v's fence posts 0 .. 63 and one more wavefront owns post 64 of all RPW rays:
// (RPW + 1) / RPW search loops per ray instead of two.  Every value is computed by the expressions of resample_kernel (bit-identical).
template <int RPW>
__global__ void __launch_bounds__((RPW + 1) * 64) resample_pack_kernel(const float* __restrict__ t_in, const float* __restrict__ weights,
                                                                       const float* __restrict__ jitter, float jscale, int R, int S, float padding,
                                                                       int randomized, float* __restrict__ t_out, int* __restrict__ tail_flag) {
  __shared__ float s_w[RPW][64], s_cdf[RPW][66], s_bins[RPW][66];
  __shared__ float s_sum[RPW], s_addw[RPW];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, n = S + 1;
  const bool strag = wv == RPW;
  const int rl = strag ? lane : wv;  // the ray (within the workgroup) and the fence post this thread produces
  const int j = strag ? 64 : lane;
  const int ray = blockIdx.x * RPW + rl;
  const bool ray_ok = rl < RPW && ray < R;
  const bool owner = !strag && ray_ok;  // the wavefront that prepares the ray's tables
  if (owner) {
    const float* w = weights + (size_t)ray * S;
    if (lane < S) {
      const float wm = w[lane > 0 ? lane - 1 : 0], wc = w[lane], wp = w[lane < S - 1 ? lane + 1 : S - 1];
      s_w[wv][lane] = 0.5f * (fmaxf(wm, wc) + fmaxf(wc, wp)) + padding;
    }
    if (lane < n) s_bins[wv][lane] = t_in[(size_t)ray * n + lane];
    if (lane == 0 && n == 65) s_bins[wv][64] = t_in[(size_t)ray * n + 64];
  }
  __syncthreads();
  if (owner) {
    double acc64 = lane < S ? (double)s_w[wv][lane] : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc64 += __shfl_xor(acc64, o, 64);
    if (lane == 0) {
      const float acc = (float)acc64;
      const float pad = fmaxf(0.f, 1e-5f - acc);
      s_sum[wv] = acc + pad;
      s_addw[wv] = pad / (float)S;
      s_cdf[wv][0] = 0.f;
      s_cdf[wv][S] = 1.0f;
    }
  }
  __syncthreads();
  if (owner) {
    double c = (lane < S - 1) ? (double)((s_w[wv][lane] + s_addw[wv]) / s_sum[wv]) : 0.0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(c, o, 64);
      if (lane >= o) c += up;
    }
    if (lane < S - 1) s_cdf[wv][lane + 1] = fminf(1.0f, (float)c);
  }
  __syncthreads();
  if (!ray_ok || j >= n) return;
  const float* bins = s_bins[rl];
  const float* cdf = s_cdf[rl];
  const float one_m_eps = 1.0f - 1.1920928955078125e-07f;
  float u;
  if (randomized) {
    const float base = (float)j * (float)(1.0 / (double)n);
    u = fminf((base + base) + jitter[(size_t)ray * n + j] * jscale, one_m_eps);  // (jscale 1: x * 1 is x)
    if (tail_flag && j >= S / 2 + 1 && u != one_m_eps) atomicOr(tail_flag, 1);
  } else {
    const float st = one_m_eps / (float)(n - 1);
    u = (j < n / 2) ? st * (float)j : one_m_eps - st * (float)(n - 1 - j);
  }
  float x0 = bins[0], x1 = bins[n - 1], y0 = cdf[0], y1 = cdf[n - 1];
  float bx0 = bins[0], by0 = cdf[0], bx1 = bins[n - 1], by1 = cdf[n - 1];
  for (int i = 0; i < n; ++i) {
    const float c = cdf[i], b = bins[i];
    const bool m = u >= c;
    bx0 = fmaxf(bx0, m ? b : x0);
    by0 = fmaxf(by0, m ? c : y0);
    bx1 = fminf(bx1, m ? x1 : b);
    by1 = fminf(by1, m ? y1 : c);
  }
  float fr = (u - by0) / (by1 - by0);
  if (fr != fr) fr = 0.f;
  fr = fminf(fmaxf(fr, 0.f), 1.f);
  t_out[(size_t)ray * n + j] = bx0 + fr * (bx1 - bx0);
}

__global__ void unnormalize_kernel(const float* __restrict__ pts, int n, float m00, float m01, float m02, float m03,
                                   float m10, float m11, float m12, float m13, float m20, float m21, float m22,
                                   float m23, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  out[3 * i + 0] = NM_FMA(1.0f, m03, NM_FMA(z, m02, NM_FMA(y, m01, x * m00)));
  out[3 * i + 1] = NM_FMA(1.0f, m13, NM_FMA(z, m12, NM_FMA(y, m11, x * m10)));
  out[3 * i + 2] = NM_FMA(1.0f, m23, NM_FMA(z, m22, NM_FMA(y, m21, x * m20)));
}

}  // namespace

extern "C" int nm_raygen_count(int H, int W, int ds) {
  if (H <= 0 || W <= 0 || ds <= 0) return 0;
  const int ny = (H - ds / 2 + ds - 1) / ds, nx = (W - ds / 2 + ds - 1) / ds;
  return nx * ny;
}

extern "C" int nm_raygen_batch(const float* Kinv_host, const float* c2w_host, int Q, int H, int W, int ds, float near_plane, float* rays,
                               int* fallback, nmStream_t stream) {
  NM_CHECK_ARG(Kinv_host && c2w_host && rays && fallback && Q > 0 && H > 1 && W > 0 && ds > 0);
  RayGenArgs a;
  for (int i = 0; i < 9; ++i) a.kinv[i] = Kinv_host[i];
  a.H = H; a.W = W; a.ds = ds;
  a.ny = (H - ds / 2 + ds - 1) / ds;
  a.nx = (W - ds / 2 + ds - 1) / ds;
  a.near_plane = near_plane;
  const int R = a.nx * a.ny;
  if (R <= 0) return NM_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(fallback, 0, (size_t)Q * sizeof(int), s) != hipSuccess) return NM_ERR_LAUNCH;
  for (int q0 = 0; q0 < Q; q0 += RAYGEN_MAXQ) {
    const int nq = Q - q0 < RAYGEN_MAXQ ? Q - q0 : RAYGEN_MAXQ;
    for (int q = 0; q < nq; ++q)
      for (int i = 0; i < 12; ++i) a.poses[q][i] = c2w_host[(size_t)(q0 + q) * 16 + i];
    float* r = rays + (size_t)q0 * R * 12;
    const char* px_env = getenv("NM_RAYGEN_PIXELS");  // "0": the one-thread-per-ray kernel (tests compare the two bit for bit)
    if (px_env && px_env[0] == '0') raygen_kernel<<<dim3((R + 63) / 64, nq), 64, 0, s>>>(a, r, fallback + q0);
    else raygen_pixels_kernel<<<dim3((H * W + 255) / 256, nq), 256, 0, s>>>(a, r, fallback + q0);
    far_fallback_kernel<<<dim3((R + 255) / 256, nq), 256, 0, s>>>(r, fallback + q0, R);
  }
  return nm_launch_status();
}

extern "C" int nm_raygen(const float* Kinv_host, const float* c2w_host, int H, int W, int ds, float near_plane,
                         float* rays, int* fallback, nmStream_t stream) {
  return nm_raygen_batch(Kinv_host, c2w_host, 1, H, W, ds, near_plane, rays, fallback, stream);
}

extern "C" int nm_sample_coarse(const float* rays, const float* t_rand, int R, int S, float* t_out, nmStream_t stream) {
  NM_CHECK_ARG(rays && t_rand && t_out && R > 0 && S > 0);
  const int total = R * (S + 1);
  sample_coarse_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(rays, t_rand, R, S, t_out);
  return nm_launch_status();
}

extern "C" int nm_resample(const float* t_in, const float* weights, const float* jitter, int R, int S, float padding,
                           int randomized, float* t_out, nmStream_t stream) {
  return nm_resample_ex(t_in, weights, jitter, R, S, padding, randomized, t_out, nullptr, stream);
}

extern "C" int nm_resample_ex(const float* t_in, const float* weights, const float* jitter, int R, int S, float padding,
                              int randomized, float* t_out, int* zero_tail_violation, nmStream_t stream) {
  return nm_resample_scaled(t_in, weights, jitter, 1.0f, R, S, padding, randomized, t_out, zero_tail_violation, stream);
}

extern "C" int nm_resample_scaled(const float* t_in, const float* weights, const float* jitter, float jitter_scale, int R, int S, float padding,
                                  int randomized, float* t_out, int* zero_tail_violation, nmStream_t stream) {
  NM_CHECK_ARG(t_in && weights && t_out && R > 0 && S > 1 && (jitter || !randomized));
  const float jscale = jitter_scale;
  int* const tf = randomized ? zero_tail_violation : nullptr;
  hipStream_t s0 = (hipStream_t)stream;
  // deterministic fence posts (linspace) never have the zero-width tail: flag = 1; otherwise the kernel raises it on violation
  if (zero_tail_violation && hipMemsetAsync(zero_tail_violation, randomized ? 0 : 1, sizeof(int), s0) != hipSuccess) return NM_ERR_LAUNCH;
  if (S + 1 > 1024) return NM_ERR_UNSUPPORTED;
  const int threads = ((S + 1 + 63) / 64) * 64;
  hipStream_t s = (hipStream_t)stream;
  // (the LDS arrays are sized by the template argument: small rows leave room for more workgroups per CU)
  constexpr int RPW = 7;  // rays per workgroup of the packed form (rows of at most 64 intervals)
  const char* pack_env = getenv("NM_RESAMPLE_PACK");  // "0": one workgroup per ray for every row length (tests compare the two forms bit for bit)
  if (S <= 64 && !(pack_env && pack_env[0] == '0')) resample_pack_kernel<RPW><<<(R + RPW - 1) / RPW, (RPW + 1) * 64, 0, s>>>(t_in, weights, jitter, jscale, R, S, padding, randomized, t_out, tf);
  else if (S + 1 <= 128) resample_kernel<128><<<R, threads, 0, s>>>(t_in, weights, jitter, jscale, S, padding, randomized, t_out, tf);
  else if (S + 1 <= 256) resample_kernel<256><<<R, threads, 0, s>>>(t_in, weights, jitter, jscale, S, padding, randomized, t_out, tf);
  else resample_kernel<1024><<<R, threads, 0, s>>>(t_in, weights, jitter, jscale, S, padding, randomized, t_out, tf);
  return nm_launch_status();
}

extern "C" int nm_unnormalize_points(const float* pts, const float* m, int n, float* out, nmStream_t stream) {
  NM_CHECK_ARG(pts && m && out && n > 0);
  unnormalize_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(pts, n, m[0], m[1], m[2], m[3], m[4], m[5], m[6],
                                                                       m[7], m[8], m[9], m[10], m[11], out);
  return nm_launch_status();
}
