// ABI version / error strings of the C boundary (include/nerfmatch_amd.h).
#include "common.h"

extern "C" int nm_abi_version(void) { return 1; }

extern "C" const char* nm_error_string(int code) {
  switch (code) {
    case NM_OK: return "ok";
    case NM_ERR_ARG: return "invalid argument (null pointer or non-positive size)";
    case NM_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case NM_ERR_LAUNCH: return "HIP launch error";
    case NM_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

// Measurement aid for bench.py (not on any product path): a bare stream of v_mfma_f32_32x32x16_f16, 24 per round on 8 accumulators,
// one wavefront per SIMD on every CU -- the matrix rate this chip SUSTAINS at its power limit (the 2.5 PFLOP/s of the data sheet
// assume 2.4 GHz; a dense MFMA stream clocks lower, profiles/r3_ubench_fillers.log).  `rounds` rounds per wavefront; `sink`
// receives one float per thread (grid * 256).  FLOP of a launch = grid * 4 * rounds * 24 * 32768.
namespace {
typedef _Float16 probe_f16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256, 1) mfma_probe_kernel(float* sink, int rounds) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  probe_f16x8 a, x;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (float)((lane + i) & 31)); x[i] = (_Float16)(0.002f * (float)((lane * 7 + i) & 31)); }
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int m = 0; m < 24; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, x, acc[m & 7], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[b][i];
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

extern "C" int nm_probe_mfma_f16(float* sink, int workgroups, int rounds, nmStream_t stream) {
  NM_CHECK_ARG(sink && workgroups > 0 && rounds > 0);
  mfma_probe_kernel<<<workgroups, 256, 0, (hipStream_t)stream>>>(sink, rounds);
  return nm_launch_status();
}
