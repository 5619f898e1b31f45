// Fused per-ray NeRF evaluation for gfx950:
//   conical frustum -> Gaussian -> integrated positional encoding -> 8x256 ReLU MLP (+ density / feature /
//   view / rgb heads) -> alpha compositing along the ray -> weighted feature / point / colour sums.
//
// Design (one workgroup = 4 wavefronts = 128 samples, one wavefront per SIMD, whole 512-register file):
//   * Every wavefront owns 32 samples.  A layer is  H_out^T[256 x 32] = W[256 x K] . H_in^T[K x 32]  computed with
//     v_mfma_f32_32x32x2_f32 (exact fp32, the 157 TFLOP/s matrix path of MI355X): the weight matrix is the A
//     operand (row = output neuron), the activations are the B operand (column = sample).
//   * The MFMA result layout (lane = sample + 32*half, register r <-> neuron (r&3) + 8*(r>>2) + 4*half of a
//     32-neuron block) is exactly a valid B-operand layout for the next layer when the two K-indices of a
//     k-step are taken to be (neuron n, neuron n+4): activations NEVER leave the register file between layers
//     - no LDS round trip, no shuffles.  The host packs the weights in the matching order (nm_nerf_pack),
//     so a wavefront streams its A operands with fully coalesced 16-byte loads (L2 resident: 2.4 MB/model).
//   * Biases, the density/rgb head vectors and per-sample scalars live in LDS; the tapped 256-d activations of
//     the 128 samples are parked in LDS (padded rows, conflict-free 16-byte stores) until the compositing
//     weights are known, then reduced per ray by 256 threads (thread = feature channel).
//   * The transmittance scan along the ray is a wavefront shuffle scan (segments of min(S,64) lanes), carried
//     across wavefronts / 128-sample chunks through LDS for S >= 128.
// HBM traffic per sample is ~100 B in / ~10 B out against 1.2 MFLOP: the kernel is bound by the fp32 MFMA rate.
#include "common.h"

namespace {

constexpr int TILE = 128;          // samples per workgroup pass
constexpr int XK = 45;             // k-steps of the 90-d IPE input (pair = sin / cos(=sin(.+pi/2)) of one (scale, axis))
constexpr int HK = 128;            // k-steps of a 256-d hidden input
constexpr int VK = 150;            // views layer: 128 (feature) + 14 (27-d direction PE, padded) + 8 (16-d appearance)
constexpr int STASH_LD = 260;      // padded row (floats) of the tapped-feature stash: 1040 B -> conflict-free b128 stores

// ---- blob layout (floats) -------------------------------------------------------------------------------------
constexpr int OFF_BIAS = 0;        // [9][256]: pts layers 0..7, feature_linear
constexpr int OFF_BVIEWS = 2304;   // [128]
constexpr int OFF_WALPHA = 2432;   // [256]
constexpr int OFF_WRGB = 2688;     // [3][128]
constexpr int OFF_MISC = 3072;     // alpha bias, rgb bias x3
constexpr int SMALL = 3088;        // floats copied to LDS
constexpr int OFF_WX0 = SMALL;                 // layer 0:   [45][2][64][4]
constexpr int OFF_WX5 = OFF_WX0 + XK * 512;    // layer 5 (skip part)
constexpr int OFF_WH = OFF_WX5 + XK * 512;     // hidden parts of layers 1..7 and feature_linear: 8 x [128][2][64][4]
constexpr int OFF_WV = OFF_WH + 8 * HK * 512;  // views: [150][1][64][4]
constexpr int BLOB_FLOATS = OFF_WV + VK * 256;

struct NerfArgs {
  const float* blob;
  const float* rays;
  const float* t;
  const float* app_row;
  float* weights;
  float* feat;
  float* pts;
  float* rgb;
  float* depth;
  float* acc;
  float* raw;
  float* sfeat;
  int R, S, tap, white_bg, flags;
  float var_scale;
  int* run_if;  // nm_nerf_fwd_guarded: device int[16] -- the launch does nothing unless bit 0 of [0] is set (NULL: always run)
};

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// tuning knobs (A/B-tested on MI355X)
#ifndef NM_GROUP
#define NM_GROUP 4   // k-steps per weight prefetch group (2 buffers of NM_GROUP * NOBG float4 each)
#endif
#ifndef NM_XPIPE
#define NM_XPIPE 0   // 1: produce the B operands of group g+1 while group g's MFMAs issue
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using wsrc_t = __amdgpu_buffer_rsrc_t;

// 16-byte weight fetch: wave-uniform descriptor + scalar byte offset + constant per-lane offset (lane*16).
// Buffer addressing keeps the whole weight stream free of per-load 64-bit address registers.
__device__ __forceinline__ f32x4 wload(wsrc_t rs, int lane_off, int soff) {
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, soff, 0);
  return __builtin_bit_cast(f32x4, r);
}

// acc[4*NOBG] (32-neuron output blocks) += W-part . x over NKS k-steps.  Weight stream: [ks][obg][lane] float4
// (4 output blocks per float4) starting `base` bytes into the blob.  Loads are grouped G k-steps at a time and
// double buffered so that a group's loads are in flight while the previous group's MFMAs issue.  `xs(ks)` yields
// the B operand of k-step ks (a register of the resident activation array, or a value generated on the fly).
template <int NKS, int NOBG, typename XS>
__device__ __forceinline__ void gemm_part(f32x16 (&acc)[4 * NOBG], wsrc_t rs, int lane_off, int base, XS xs) {
  constexpr int G = NM_GROUP;
  constexpr int NG = (NKS + G - 1) / G;
  f32x4 bufA[G * NOBG], bufB[G * NOBG];
  auto load = [&](f32x4(&buf)[G * NOBG], int g) {
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (g * G + j < NKS) {
#pragma unroll
        for (int o = 0; o < NOBG; ++o) buf[j * NOBG + o] = wload(rs, lane_off, base + ((g * G + j) * NOBG + o) * 1024);
      }
  };
  float xq[2][G];
  auto fetch_x = [&](float(&dst)[G], int g) {
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (g * G + j < NKS) dst[j] = xs(g * G + j);
  };
  auto compute = [&](const f32x4(&buf)[G * NOBG], const float(&xv4)[G], int g) {
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (g * G + j < NKS) {
        const float xv = NM_XPIPE ? xv4[j] : xs(g * G + j);
#pragma unroll
        for (int o = 0; o < NOBG; ++o) {
          const f32x4 w = buf[j * NOBG + o];
          acc[4 * o + 0] = MFMA32(w[0], xv, acc[4 * o + 0]);
          acc[4 * o + 1] = MFMA32(w[1], xv, acc[4 * o + 1]);
          acc[4 * o + 2] = MFMA32(w[2], xv, acc[4 * o + 2]);
          acc[4 * o + 3] = MFMA32(w[3], xv, acc[4 * o + 3]);
        }
      }
  };
  load(bufA, 0);
  if (NM_XPIPE) fetch_x(xq[0], 0);
#pragma unroll
  for (int g = 0; g < NG; g += 2) {
    if (g + 1 < NG) load(bufB, g + 1);
    if (NM_XPIPE && g + 1 < NG) fetch_x(xq[1], g + 1);
    compute(bufA, xq[0], g);
    __builtin_amdgcn_sched_barrier(0);
    if (g + 2 < NG) load(bufA, g + 2);
    if (NM_XPIPE && g + 2 < NG) fetch_x(xq[0], g + 2);
    if (g + 1 < NG) compute(bufB, xq[1], g + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Opaque copy of a lane-varying int: values derived from the copy cannot be hoisted above this point.  Used so that
// cheap epilogue-only quantities (LDS addresses, view-direction encodings) are recomputed where they are needed
// instead of being kept live (= spilled to scratch) across the ~9,500 MFMAs of the MLP.
__device__ __forceinline__ int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// neuron index inside a 32-block held by (register r, half hi)
__host__ __device__ __forceinline__ constexpr int nrow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// One 128-sample tile (`bid` = tile index; `sm` = the workgroup's single LDS object)
__device__ __forceinline__ void nerf_fwd_tile(const NerfArgs& a, const int bid, float* const sm) {
  float* const sm_small = sm;
  float* const sm_stash = sm + SMALL;
  float* const sm_sigma = sm_stash + TILE * STASH_LD;  // [128]
  float* const sm_rgb = sm_sigma + TILE;               // [3][128]
  float* const sm_t0 = sm_rgb + 3 * TILE;              // [128]
  float* const sm_t1 = sm_t0 + TILE;                   // [128]
  float* const sm_mean = sm_t1 + TILE;                 // [3][128]
  float* const sm_dn = sm_mean + 3 * TILE;             // [128] |d| of the sample's ray
  float* const sm_w = sm_dn + TILE;                    // [128]
  float* const sm_misc = sm_w + TILE;                  // [32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = lane & 31, hi = lane >> 5;
  const int S = a.S, R = a.R;
  const int SP = S < TILE ? S : TILE;     // samples of one ray inside a 128-sample pass
  const int nr = TILE / SP;               // rays per workgroup
  const int nchunks = (S + TILE - 1) / TILE;
  const bool need_rgb = !(a.flags & NM_NERF_SKIP_RGB);
  const bool feat_max = (a.flags & NM_NERF_FEAT_MAX) != 0;
  const bool need_stash = (a.feat != nullptr) || (a.sfeat != nullptr);
  const int tap = (a.tap < 0 || a.tap > 7) ? 7 : a.tap;

  for (int i = tid; i < SMALL / 4; i += 256) reinterpret_cast<f32x4*>(sm_small)[i] = reinterpret_cast<const f32x4*>(a.blob)[i];
  __syncthreads();  // layer 0 reads biases other wavefronts copied (the first barrier of the layer loop comes later)

  // ---- the sample this lane feeds into the MLP --------------------------------------------------------------
  const int js = wave * 32 + s;            // sample slot inside the tile
  const int rl = js / SP;                  // ray slot inside the tile
  const int ray = bid * nr + rl;
  const int rc = ray < R ? ray : R - 1;    // clamp: out-of-range slots recompute the last ray, writes are masked
  const float* rp = a.rays + (size_t)rc * 12;
  const float o0 = rp[0], o1 = rp[1], o2 = rp[2], d0 = rp[3], d1 = rp[4], d2 = rp[5];
  const float radius = rp[11];
  const float dsq0 = d0 * d0, dsq1 = d1 * d1, dsq2 = d2 * d2;
  const float dmag = fmaxf(1e-10f, (dsq0 + dsq1) + dsq2);
  const float dnorm = sqrtf((dsq0 + dsq1) + dsq2);
  const float nul0 = 1.0f - dsq0 / dmag, nul1 = 1.0f - dsq1 / dmag, nul2 = 1.0f - dsq2 / dmag;

  // ---- per-thread state of the reduction phase (valid for thread roles described below) -----------------------
  float red_acc = 0.f;      // threads < 8*nr: running sum of quantity q for ray slot r
  float feat_acc[4] = {0.f, 0.f, 0.f, 0.f};
  float carryT = 1.f;       // transmittance at the start of the current 128-sample chunk (S > 128 only)
  float best_w = -1.f;      // FEAT_MAX: best weight so far (threads < 8*nr with q == 0; feature threads mirror via LDS)

  const wsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.blob), 0, BLOB_FLOATS * 4, 0x00020000);
  const int lane_off = lane * 16;

  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int sidx = chunk * TILE + (js % SP);
    const float t0 = a.t[(size_t)rc * (S + 1) + sidx];
    const float t1 = a.t[(size_t)rc * (S + 1) + sidx + 1];

    // conical frustum -> Gaussian (stable form), lifted to 3-D with a diagonal covariance
    const float mu = (t0 + t1) / 2.0f, hw = (t1 - t0) / 2.0f;
    const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
    const float denom = fmaxf(1.1920928955078125e-07f, 3.0f * mu2 + hw2);
    const float t_mean = mu + (2.0f * mu * hw2) / denom;
    const float t_var = hw2 / 3.0f - (float)(4.0 / 15.0) * ((hw4 * (12.0f * mu2 - hw2)) / (denom * denom));
    const float r_var = (radius * radius) * ((mu2 / 4.0f + (float)(5.0 / 12.0) * hw2) - (float)(4.0 / 15.0) * hw4 / denom);
    float mean[3] = {d0 * t_mean + o0, d1 * t_mean + o1, d2 * t_mean + o2};
    float var[3] = {t_var * dsq0 + r_var * nul0, t_var * dsq1 + r_var * nul1, t_var * dsq2 + r_var * nul2};
    if (a.var_scale > 0.f) {
      var[0] *= a.var_scale; var[1] *= a.var_scale; var[2] *= a.var_scale;
    }

    // integrated positional encoding, generated per k-step (scale i, axis ax) right where the MFMA consumes it:
    // lanes 0-31 hold the sin entry, lanes 32-63 the sin(. + pi/2) entry of the same (i, ax) - the two K indices
    // of one MFMA step.  It is evaluated twice (layers 0 and 5) instead of being kept in 45 registers.
    auto ipe_at = [&](int ks) -> float {
      const int i = ks / 3, ax = ks % 3;
      const float sc = (float)(1 << i);
      const float xe = mean[ax] * sc;
      const float arg = hi ? (xe + 1.57079637050628662109375f) : xe;
      const float ye = var[ax] * (sc * sc);
      return expf(-0.5f * ye) * nm_sinf(arg);
    };

    // ---- the 8 pts layers + feature_linear, activations register resident ------------------------------------
    float x[HK];
    float sig_part = 0.f;
#pragma unroll 1
    for (int l = 0; l < 9; ++l) {
      if (l == 8 && !need_rgb) break;
      f32x16 acc[8];
      const float* bl = sm_small + OFF_BIAS + l * 256 + 4 * hi;
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bl + ob * 32 + 8 * q);
          acc[ob][4 * q + 0] = b[0]; acc[ob][4 * q + 1] = b[1]; acc[ob][4 * q + 2] = b[2]; acc[ob][4 * q + 3] = b[3];
        }
      if (l == 0 || l == 5) gemm_part<XK, 2>(acc, wrs, lane_off, (l == 0 ? OFF_WX0 : OFF_WX5) * 4, ipe_at);
      if (l != 0) gemm_part<HK, 2>(acc, wrs, lane_off, (OFF_WH + (l - 1) * HK * 512) * 4, [&](int ks) -> float { return x[ks]; });
      const float floor_v = (l < 8) ? 0.f : -__builtin_inff();  // feature_linear has no activation
#pragma unroll
      for (int ob = 0; ob < 8; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[ob * 16 + r] = fmaxf(acc[ob][r], floor_v);

      if (l == tap && need_stash) {
        float* row = sm_stash + js * STASH_LD + 4 * hi;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v4 = {x[ob * 16 + 4 * q], x[ob * 16 + 4 * q + 1], x[ob * 16 + 4 * q + 2], x[ob * 16 + 4 * q + 3]};
            *reinterpret_cast<f32x4*>(row + ob * 32 + 8 * q) = v4;
          }
      }
      if (l == 7) {
        const float* wa = sm_small + OFF_WALPHA + 4 * hi;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 w4v = *reinterpret_cast<const f32x4*>(wa + ob * 32 + 8 * q);
            p0 = NM_FMA(x[ob * 16 + 4 * q + 0], w4v[0], p0);
            p1 = NM_FMA(x[ob * 16 + 4 * q + 1], w4v[1], p1);
            p2 = NM_FMA(x[ob * 16 + 4 * q + 2], w4v[2], p2);
            p3 = NM_FMA(x[ob * 16 + 4 * q + 3], w4v[3], p3);
          }
        sig_part = (p0 + p1) + (p2 + p3);
      }
    }
    const float sigma_raw = (sig_part + nm_shfl_xor32(sig_part)) + sm_small[OFF_MISC];

    // ---- views layer + rgb head -------------------------------------------------------------------------------
    float c_r = 0.f, c_g = 0.f, c_b = 0.f;
    if (need_rgb) {
      float vx[VK - HK];  // 12 direction-PE pairs, raw direction (x,y) (z,0), 8 appearance pairs
      const float* rp2 = a.rays + (size_t)launder(rc) * 12;
      const float v0 = rp2[8], v1 = rp2[9], v2 = rp2[10];
      const int hi = launder(lane) >> 5;
      const float vd[3] = {v0, v1, v2};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const float xe = vd[ax] * (float)(1 << i);
          vx[i * 3 + ax] = nm_sinf(hi ? (xe + 1.57079637050628662109375f) : xe);
        }
      vx[12] = hi ? v1 : v0;
      vx[13] = hi ? 0.f : v2;
#pragma unroll
      for (int j = 0; j < 8; ++j) vx[14 + j] = a.app_row ? a.app_row[2 * j + hi] : 0.f;
      f32x16 av[4];
      const float* bv = sm_small + OFF_BVIEWS + 4 * hi;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(bv + ob * 32 + 8 * q);
          av[ob][4 * q + 0] = b[0]; av[ob][4 * q + 1] = b[1]; av[ob][4 * q + 2] = b[2]; av[ob][4 * q + 3] = b[3];
        }
      gemm_part<HK, 1>(av, wrs, lane_off, OFF_WV * 4, [&](int ks) -> float { return x[ks]; });
      gemm_part<VK - HK, 1>(av, wrs, lane_off, (OFF_WV + HK * 256) * 4, [&](int ks) -> float { return vx[ks]; });
      const float* wr = sm_small + OFF_WRGB + 4 * hi;
      float pr = 0.f, pg = 0.f, pb = 0.f;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 wr4 = *reinterpret_cast<const f32x4*>(wr + ob * 32 + 8 * q);
          const f32x4 wg4 = *reinterpret_cast<const f32x4*>(wr + 128 + ob * 32 + 8 * q);
          const f32x4 wb4 = *reinterpret_cast<const f32x4*>(wr + 256 + ob * 32 + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float hv = fmaxf(av[ob][4 * q + e], 0.f);
            pr = NM_FMA(hv, wr4[e], pr);
            pg = NM_FMA(hv, wg4[e], pg);
            pb = NM_FMA(hv, wb4[e], pb);
          }
        }
      pr = (pr + nm_shfl_xor32(pr)) + sm_small[OFF_MISC + 1];
      pg = (pg + nm_shfl_xor32(pg)) + sm_small[OFF_MISC + 2];
      pb = (pb + nm_shfl_xor32(pb)) + sm_small[OFF_MISC + 3];
      c_r = 1.0f / (1.0f + expf(-pr));
      c_g = 1.0f / (1.0f + expf(-pg));
      c_b = 1.0f / (1.0f + expf(-pb));
    }
    const int jsw = launder(js);
    if ((launder(lane) >> 5) == 0) {
      const int js = jsw;
      sm_sigma[js] = sigma_raw;
      sm_rgb[js] = c_r; sm_rgb[TILE + js] = c_g; sm_rgb[2 * TILE + js] = c_b;
      sm_t0[js] = t0; sm_t1[js] = t1;
      sm_mean[js] = mean[0]; sm_mean[TILE + js] = mean[1]; sm_mean[2 * TILE + js] = mean[2];
      sm_dn[js] = dnorm;
    }
    __syncthreads();

    // ---- alpha compositing: thread j < 128 owns sample slot j; wavefront shuffle scan of (1 - alpha + 1e-10) ----
    float alpha = 0.f, incl = 1.f;
    const int tid = launder(threadIdx.x), lane = tid & 63, wave = tid >> 6;
    if (tid < TILE) {
      const float sg = fmaxf(sm_sigma[tid], 0.f);
      const float delta = (sm_t1[tid] - sm_t0[tid]) * sm_dn[tid];
      alpha = 1.0f - expf(-sg * delta);
      incl = (1.0f - alpha) + 1e-10f;
      const int seg = SP < 64 ? SP : 64;
#pragma unroll
      for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const float up = __shfl_up(incl, dlt, 64);
        if (dlt < seg && (lane & (seg - 1)) >= dlt) incl *= up;
      }
      if (lane == 63) sm_misc[wave] = incl;  // product over this wavefront's last segment (whole wave when SP >= 64)
    }
    __syncthreads();
    float wgt = 0.f;
    if (tid < TILE) {
      const int seg = SP < 64 ? SP : 64;
      float excl = __shfl_up(incl, 1, 64);
      if ((lane & (seg - 1)) == 0) excl = 1.f;
      if (SP == TILE && wave == 1) excl *= sm_misc[0];
      excl *= carryT;
      wgt = alpha * excl;
      sm_w[tid] = wgt;
      const int r2 = tid / SP, ray2 = bid * nr + r2;
      if (ray2 < R) {
        const int s2 = chunk * TILE + tid % SP;
        a.weights[(size_t)ray2 * S + s2] = wgt;
        if (a.raw) {
          f32x4 rv = {sm_rgb[tid], sm_rgb[TILE + tid], sm_rgb[2 * TILE + tid], sm_sigma[tid]};
          *reinterpret_cast<f32x4*>(a.raw + ((size_t)ray2 * S + s2) * 4) = rv;
        }
      }
    }
    if (nchunks > 1) carryT = carryT * (sm_misc[0] * sm_misc[1]);  // S > 128: one ray per workgroup
    __syncthreads();

    // ---- per-ray sums: thread (r, q) for q in {acc, r, g, b, depth, x, y, z} ----------------------------------
    if (tid < 8 * nr) {
      const int q = tid & 7, r2 = tid >> 3;
      const float* wv = sm_w + r2 * SP;
      if (!feat_max || q < 5) {
        // q selects one LDS array (or constants); 8 independent loads in flight per step (SP is a multiple of 32)
        const float* va = q == 0 ? nullptr : q <= 3 ? sm_rgb + (q - 1) * TILE + r2 * SP : q == 4 ? sm_t0 + r2 * SP : sm_mean + (q - 5) * TILE + r2 * SP;
        const float* vb = q == 4 ? sm_t1 + r2 * SP : nullptr;
        float sum = 0.f;
        for (int k0 = 0; k0 < SP; k0 += 8) {
          float wk[8], xk[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            wk[e] = wv[k0 + e];
            xk[e] = va ? va[k0 + e] : 1.0f;
            if (vb) xk[e] = 0.5f * (xk[e] + vb[k0 + e]);
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) sum += wk[e] * xk[e];
        }
        red_acc += sum;
      }
      if (feat_max) {
        // first maximum of the weights (torch.max semantics); strict > across chunks keeps the earliest
        float bw = wv[0];
        int bi = 0;
        for (int k = 1; k < SP; ++k)
          if (wv[k] > bw) { bw = wv[k]; bi = k; }
        const bool better = bw > best_w;
        if (better) best_w = bw;
        if (q == 0) sm_misc[8 + r2] = better ? __int_as_float(r2 * SP + bi) : __int_as_float(-1);
        if (q >= 5 && better) red_acc = sm_mean[(q - 5) * TILE + r2 * SP + bi];
      }
    }
    if (feat_max) __syncthreads();
    if (a.feat) {
#pragma unroll
      for (int r2 = 0; r2 < 4; ++r2)
        if (r2 < nr) {
          if (feat_max) {
            const int bi = __float_as_int(sm_misc[8 + r2]);
            if (bi >= 0) feat_acc[r2] = sm_stash[bi * STASH_LD + tid];
          } else {
            float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f;  // 4 independent chains keep 8 LDS loads in flight
            const float* st = sm_stash + (r2 * SP) * STASH_LD + tid;
            const float* ww = sm_w + r2 * SP;
            for (int k = 0; k < SP; k += 4) {
              f0 = NM_FMA(ww[k], st[k * STASH_LD], f0);
              f1 = NM_FMA(ww[k + 1], st[(k + 1) * STASH_LD], f1);
              f2 = NM_FMA(ww[k + 2], st[(k + 2) * STASH_LD], f2);
              f3 = NM_FMA(ww[k + 3], st[(k + 3) * STASH_LD], f3);
            }
            feat_acc[r2] += (f0 + f1) + (f2 + f3);
          }
        }
    }
    if (a.sfeat) {
      for (int k = 0; k < TILE; ++k) {
        const int r2 = k / SP, ray2 = bid * nr + r2;
        if (ray2 < R) a.sfeat[((size_t)ray2 * S + chunk * TILE + k % SP) * 256 + tid] = sm_stash[k * STASH_LD + tid];
      }
    }
    __syncthreads();  // stash / scratch are rewritten by the next chunk
  }

  // ---- final per-ray writes ---------------------------------------------------------------------------------------
  if (tid < 8 * nr) {
    const int q = tid & 7, r2 = tid >> 3, ray2 = bid * nr + r2;
    const float accv = __shfl(red_acc, lane & ~7, 64);
    if (ray2 < R) {
      if (q == 0) { if (a.acc) a.acc[ray2] = red_acc; }
      else if (q <= 3) { if (a.rgb && need_rgb) a.rgb[(size_t)ray2 * 3 + (q - 1)] = a.white_bg ? red_acc + (1.0f - accv) : red_acc; }
      else if (q == 4) { if (a.depth) a.depth[ray2] = red_acc; }
      else { if (a.pts) a.pts[(size_t)ray2 * 3 + (q - 5)] = red_acc; }
    }
  }
  if (a.feat) {
#pragma unroll
    for (int r2 = 0; r2 < 4; ++r2) {
      const int ray2 = bid * nr + r2;
      if (r2 < nr && ray2 < R) a.feat[(size_t)ray2 * 256 + tid] = feat_acc[r2];
    }
  }
}


__global__ void __launch_bounds__(256, 1) nerf_fwd_kernel(NerfArgs a) {
  // single LDS object (cdna guide: a second __shared__ object can de-pipeline the loads)
  __shared__ __attribute__((aligned(16))) float sm[SMALL + TILE * STASH_LD + TILE * 12 + 32];
  nerf_fwd_tile(a, blockIdx.x, sm);
}

// Guarded form (nm_nerf_fwd_guarded: the fall-back behind the fp16x3 kernel's saturation flag).  The decision is taken on the
// device -- and it must be CHEAP when the flag is down, which it almost always is: this kernel's 133 KB of LDS allow one workgroup
// per CU, so a grid of one workgroup per tile (38,400 for 16 queries) takes ~0.6 ms just to start and exit.  Hence a persistent
// grid of at most one workgroup per CU that walks the tiles: 256 workgroups read the flag and leave.
__global__ void __launch_bounds__(256, 1) nerf_fwd_guarded_kernel(NerfArgs a, int ntiles) {
  __shared__ __attribute__((aligned(16))) float sm[SMALL + TILE * STASH_LD + TILE * 12 + 32];
  if (!(*a.run_if & 1)) return;
  for (int bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
    nerf_fwd_tile(a, bid, sm);
    __syncthreads();  // the next tile re-uses the LDS
  }
  // The flag is CONSUMED: the last workgroup to finish (every workgroup read the flag before it got here) clears bit 0 and counts the
  // event in run_if[11], so that one saturating launch does not send every later launch on this status block through the fp32 kernel.
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(a.run_if + 12, 1) == (int)gridDim.x - 1) {
      a.run_if[12] = 0;
      atomicAdd(a.run_if + 11, 1);
      atomicAnd(a.run_if, ~1);
    }
  }
}

}  // namespace

extern "C" size_t nm_nerf_blob_floats(void) { return (size_t)BLOB_FLOATS; }

// Host-side packing of one MLP into MFMA A-operand order.  For a k-step `ks`, output-block group `obg` and lane
// (row = lane & 31, half = lane >> 5) the float4 holds W[32*(4*obg + c) + row][in(ks, half)], c = 0..3.
extern "C" int nm_nerf_pack(const nmNerfWeights* w, float* blob) {
  if (!w || !blob) return NM_ERR_ARG;
  for (int i = 0; i < 8; ++i)
    if (!w->pts_w[i] || !w->pts_b[i]) return NM_ERR_ARG;
  if (!w->alpha_w || !w->alpha_b || !w->feat_w || !w->feat_b || !w->views_w || !w->views_b || !w->rgb_w || !w->rgb_b)
    return NM_ERR_ARG;
  if (w->app_dim != 0 && w->app_dim != 16) return NM_ERR_UNSUPPORTED;
  for (size_t i = 0; i < (size_t)BLOB_FLOATS; ++i) blob[i] = 0.f;
  for (int l = 0; l < 8; ++l)
    for (int n = 0; n < 256; ++n) blob[OFF_BIAS + l * 256 + n] = w->pts_b[l][n];
  for (int n = 0; n < 256; ++n) blob[OFF_BIAS + 8 * 256 + n] = w->feat_b[n];
  for (int n = 0; n < 128; ++n) blob[OFF_BVIEWS + n] = w->views_b[n];
  for (int n = 0; n < 256; ++n) blob[OFF_WALPHA + n] = w->alpha_w[n];
  for (int n = 0; n < 384; ++n) blob[OFF_WRGB + n] = w->rgb_w[n];
  blob[OFF_MISC] = w->alpha_b[0];
  for (int c = 0; c < 3; ++c) blob[OFF_MISC + 1 + c] = w->rgb_b[c];

  auto hid_in = [](int ks, int half) { return 32 * (ks / 16) + nrow(ks % 16, half); };
  // IPE parts (layers 0 and 5): input index = (scale*3 + axis) + 45*half
  for (int part = 0; part < 2; ++part) {
    const float* W = part == 0 ? w->pts_w[0] : w->pts_w[5];
    const int ld = part == 0 ? 90 : 346;
    float* dst = blob + (part == 0 ? OFF_WX0 : OFF_WX5);
    for (int ks = 0; ks < XK; ++ks)
      for (int obg = 0; obg < 2; ++obg)
        for (int lane = 0; lane < 64; ++lane)
          for (int c = 0; c < 4; ++c)
            dst[((ks * 2 + obg) * 64 + lane) * 4 + c] = W[(size_t)(32 * (4 * obg + c) + (lane & 31)) * ld + ks + 45 * (lane >> 5)];
  }
  // hidden parts: layers 1..7 (layer 5: columns 90..345) and feature_linear
  for (int l = 1; l <= 8; ++l) {
    const float* W = l < 8 ? w->pts_w[l] : w->feat_w;
    const int ld = l == 5 ? 346 : 256, col0 = l == 5 ? 90 : 0;
    float* dst = blob + OFF_WH + (size_t)(l - 1) * HK * 512;
    for (int ks = 0; ks < HK; ++ks)
      for (int obg = 0; obg < 2; ++obg)
        for (int lane = 0; lane < 64; ++lane)
          for (int c = 0; c < 4; ++c)
            dst[((ks * 2 + obg) * 64 + lane) * 4 + c] =
                W[(size_t)(32 * (4 * obg + c) + (lane & 31)) * ld + col0 + hid_in(ks, lane >> 5)];
  }
  // views: [feature 256 | dir PE 27 | app 16]
  {
    const int ld = 283 + w->app_dim;
    float* dst = blob + OFF_WV;
    for (int ks = 0; ks < VK; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const int half = lane >> 5;
        int in;
        if (ks < HK) in = hid_in(ks, half);
        else if (ks < HK + 12) in = 256 + (ks - HK) + 12 * half;          // sin block | sin(.+pi/2) block
        else if (ks == HK + 12) in = 256 + 24 + half;                      // raw x, y
        else if (ks == HK + 13) in = half ? -1 : 256 + 26;                 // raw z, pad
        else in = w->app_dim ? 283 + 2 * (ks - HK - 14) + half : -1;       // appearance embedding
        for (int c = 0; c < 4; ++c)
          dst[(ks * 64 + lane) * 4 + c] = in < 0 ? 0.f : w->views_w[(size_t)(32 * c + (lane & 31)) * ld + in];
      }
  }
  return NM_OK;
}

static int nerf_fwd_launch(const float* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                           int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                           float* rgb, float* depth, float* acc, float* raw, float* sample_feat, int* run_if, nmStream_t stream) {
  NM_CHECK_ARG(blob && rays && t && weights && R > 0 && S > 0);
  if (!(S == 32 || S == 64 || (S % 128) == 0)) return NM_ERR_UNSUPPORTED;
  if (tap_layer > 7) return NM_ERR_ARG;
  NerfArgs a;
  a.blob = blob; a.rays = rays; a.t = t; a.app_row = app_row;
  a.weights = weights; a.feat = feat; a.pts = pts; a.rgb = rgb; a.depth = depth; a.acc = acc; a.raw = raw; a.sfeat = sample_feat;
  a.R = R; a.S = S; a.tap = tap_layer; a.white_bg = white_bg; a.flags = flags; a.var_scale = var_scale;
  a.run_if = run_if;
  const int SP = S < TILE ? S : TILE, nr = TILE / SP;
  const int ntiles = (R + nr - 1) / nr;
  if (run_if) {
    const int ncu = nm_stream_cus(stream);
    nerf_fwd_guarded_kernel<<<ntiles < ncu ? ntiles : ncu, 256, 0, (hipStream_t)stream>>>(a, ntiles);
  } else {
    nerf_fwd_kernel<<<ntiles, 256, 0, (hipStream_t)stream>>>(a);
  }
  return nm_launch_status();
}

extern "C" int nm_nerf_fwd(const float* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                           int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                           float* rgb, float* depth, float* acc, float* raw, float* sample_feat, nmStream_t stream) {
  return nerf_fwd_launch(blob, rays, t, app_row, R, S, tap_layer, white_bg, var_scale, flags, weights, feat, pts, rgb, depth, acc, raw,
                         sample_feat, nullptr, stream);
}

extern "C" int nm_nerf_fwd_guarded(const float* blob, const float* rays, const float* t, const float* app_row, int R, int S,
                                   int tap_layer, int white_bg, float var_scale, int flags, float* weights, float* feat, float* pts,
                                   float* rgb, float* depth, float* acc, float* raw, float* sample_feat, int* run_if,
                                   nmStream_t stream) {
  if (!run_if) return NM_ERR_ARG;
  return nerf_fwd_launch(blob, rays, t, app_row, R, S, tap_layer, white_bg, var_scale, flags & ~NM_NERF_ZERO_TAIL, weights, feat, pts, rgb,
                         depth, acc, raw, sample_feat, run_if, stream);
}
