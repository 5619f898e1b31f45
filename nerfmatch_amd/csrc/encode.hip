// Stand-alone encoders: the reference's encoding MODULES as callable kernels (round 5, SURVEY 8a row N0 / 8b: the evaluator reaches
// into `renderer.xyz_encoder(mean, var)` / `renderer.dirs_encoder(viewdirs)` directly, nerfmatch_evaluator.py:385-393).
//
//   nm_mip_encode     PositionalEncodingMIP.forward(x, y)   nerfmatch/nerf/embedding.py:66-84
//                     x_enc = [x * 2^s]_(s, axis) ++ the same + fl32(pi/2);  with y (IPE):  x_ret = exp(-y_enc / 2) sin(x_enc),
//                     y_ret = max(0, (1 - exp(-2 y_enc) cos(2 x_enc)) / 2 - x_ret^2),  y_enc = [y * 4^s] twice;  without y (PE):
//                     out = [sin(x_enc) | x]
//   nm_fourier_embed  FourierEmbedding.forward(x)           embedding.py:35-46:  [x | sin(2^0 x) | cos(2^0 x) | sin(2^1 x) | ...]
//
// `arith` selects the transcendental arithmetic of nm_mip_encode's x_ret: 0 = expf + the fp64-reduced sine (what nerf_fwd.hip, the
// pointwise iNeRF kernels and nm_inerf_encode evaluate), 1 = exp2-based exponential + the fp32 Cody-Waite sine `sin32` -- the very
// device functions the split render kernels (nerf_fwd_bf16.hip) inline, so that their encoding can be pinned against the reference's
// values directly and not only through the MLP's outputs (tests/test_encoders_gpu.py).
// The inside of the fused kernels never calls these entry points: there the encodings are produced in registers / LDS as MFMA operands.
#include "common.h"
#include "nerf_bf16_common.h"

namespace {

constexpr float HALF_PI_F32 = 1.57079637050628662109375f;  // fl32(0.5 * pi): the reference adds a python float to an fp32 tensor

// one thread per (row, s, axis): both halves of the encoding (phase 0 and fl32(pi/2))
template <int ARITH>
__global__ void __launch_bounds__(256) mip_encode_kernel(const float* __restrict__ x, const float* __restrict__ y, size_t n, int D, int min_deg,
                                                         int F, float* __restrict__ x_ret, float* __restrict__ y_ret) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int FD = F * D;
  if (idx >= n * (size_t)FD) return;
  const size_t row = idx / FD;
  const int c = (int)(idx % FD), s = c / D, ax = c % D;
  const float sc = (float)(1u << (min_deg + s));  // (scales are int64 powers of two in the reference: the products are exact)
  const float xv = x[row * D + ax];
  const float xe0 = xv * sc, xe1 = xe0 + HALF_PI_F32;
  if (!y) {  // PE: [sin(x_enc) (2 F D) | x (D)]
    float* o = x_ret + row * (size_t)(2 * FD + D);
    o[c] = nm_sinf(xe0);
    o[FD + c] = nm_sinf(xe1);
    if (s == 0) o[2 * FD + ax] = xv;
    return;
  }
  const float ye = y[row * D + ax] * (sc * sc);
  float damp;
  if (ARITH == 1) damp = __builtin_amdgcn_exp2f((-0.5f * ye) * 1.44269504088896340736f);
  else damp = expf(-0.5f * ye);
  const float s0 = ARITH == 1 ? nmbf::sin32(xe0) : nm_sinf(xe0);
  const float s1 = ARITH == 1 ? nmbf::sin32(xe1) : nm_sinf(xe1);
  const float r0 = damp * s0, r1 = damp * s1;
  float* o = x_ret + row * (size_t)(2 * FD);
  o[c] = r0;
  o[FD + c] = r1;
  if (y_ret) {
    const float e2 = expf(-2.0f * ye);
    float* v = y_ret + row * (size_t)(2 * FD);
    v[c] = fmaxf(0.f, 0.5f * (1.0f - e2 * nm_cosf(2.0f * xe0)) - r0 * r0);
    v[FD + c] = fmaxf(0.f, 0.5f * (1.0f - e2 * nm_cosf(2.0f * xe1)) - r1 * r1);
  }
}

// one thread per (row, f, axis): sin and cos of 2^f x
__global__ void __launch_bounds__(256) fourier_embed_kernel(const float* __restrict__ x, size_t n, int D, int F, float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int FD = F * D;
  if (idx >= n * (size_t)FD) return;
  const size_t row = idx / FD;
  const int c = (int)(idx % FD), f = c / D, ax = c % D;
  const float xv = x[row * D + ax];
  const float arg = ((float)(1u << f) * xv) * 1.0f;  // freq * x * scale, scale = 1
  float sn, cs;
  nm_sincosf(arg, sn, cs);
  float* o = out + row * (size_t)(D + 2 * FD);
  if (f == 0) o[ax] = xv;
  o[D + 2 * f * D + ax] = sn;
  o[D + (2 * f + 1) * D + ax] = cs;
}

}  // namespace

extern "C" int nm_mip_encode(const float* x, const float* y, size_t n, int D, int min_deg, int num_freqs, int arith, float* x_ret, float* y_ret,
                             nmStream_t stream) {
  NM_CHECK_ARG(x && x_ret && D > 0 && num_freqs > 0 && min_deg >= 0 && min_deg + num_freqs <= 31 && (arith == 0 || arith == 1));
  if (!y && y_ret) return NM_ERR_ARG;
  if (n == 0) return NM_OK;
  const size_t total = n * (size_t)(num_freqs * D);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (arith == 1) mip_encode_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n, D, min_deg, num_freqs, x_ret, y_ret);
  else mip_encode_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(x, y, n, D, min_deg, num_freqs, x_ret, y_ret);
  return nm_launch_status();
}

extern "C" int nm_fourier_embed(const float* x, size_t n, int D, int num_freqs, float* out, nmStream_t stream) {
  NM_CHECK_ARG(x && out && D > 0 && num_freqs > 0 && num_freqs <= 31);
  if (n == 0) return NM_OK;
  const size_t total = n * (size_t)(num_freqs * D);
  fourier_embed_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, n, D, num_freqs, out);
  return nm_launch_status();
}
