// Shared device helpers for the gfx950 kernels of nerfmatch_amd.  Wavefront = 64 lanes throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nerfmatch_amd.h"

#define NM_CHECK_ARG(cond) \
  do {                     \
    if (!(cond)) return NM_ERR_ARG; \
  } while (0)

static inline int nm_launch_status() { return hipGetLastError() == hipSuccess ? NM_OK : NM_ERR_LAUNCH; }
// compute units of the current device (persistent kernels launch one workgroup per CU); queried once per process --
// hipDeviceGetAttribute costs tens of microseconds per call, more than a small kernel
static inline int nm_cu_count() {
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cached = n;
  return n;
}

// compute units a persistent kernel launched on `stream` may use: the partition's size for a stream made by nm_stream_create_cu_mask
// (capi.hip), the device's CU count for every other stream
extern "C" int nm_stream_cus(nmStream_t stream);

// Counted waits on the in-order VMEM counter ("at most n of my loads / LDS-DMA pieces may still be in flight") are what the
// software pipelines of these kernels rest on, and n is derived by hand from the number and ORDER of the VMEM instructions the
// compiler emits.  A compiler that splits, merges or moves one of them would turn a counted wait into a silent stale-LDS read
// (ADVICE r3).  Guard: every counted wait goes through this macro, and build.py also produces lib/libnerfmatch_amd_safewait.so
// with -DNM_SAFE_WAIT, where they all become vmcnt(0) -- slower, independent of the count; tests/test_safe_wait_gpu.py requires
// the two libraries to agree BIT FOR BIT on every kernel family that uses counted waits.
#ifdef NM_SAFE_WAIT
#define NM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define NM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// The build uses -ffp-contract=off: every fused multiply-add below is written explicitly so that the
// plain mul/add sequences of the reference's elementwise torch ops are reproduced op for op.
#define NM_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// sin / cos of an fp32 argument of any magnitude the path produces (|x| up to ~1e7): range reduction and
// polynomial in fp64 (musl __sindf/__cosdf coefficients), result rounded once to fp32 (< 1 ulp).
// Used for the integrated positional encoding (2^14 * x), the view-direction PE and the 3-D Fourier embedding.
__device__ __forceinline__ float nm_sincos_sel(float x, int quadrant_shift) {
  const double xd = (double)x;
  const double n = __builtin_rint(xd * 0.63661977236758134308);  // x * 2/pi
  double r = __builtin_fma(-n, 1.57079632673412561417e+00, xd);  // pi/2 split in two (Cody-Waite): hi
  r = __builtin_fma(-n, 6.07710050650619224932e-11, r);          // lo
  const int q = (int)n + quadrant_shift;
  const double z = r * r;
  // sin(r), |r| <= pi/4
  const double S1 = -0.166666666416265235595, S2 = 0.0083333293858894631756, S3 = -0.000198393348360966317347,
               S4 = 0.0000027183114939898219064;
  const double w = z * z;
  const double sr = (r + (z * r) * (S1 + z * S2)) + (z * r) * w * (S3 + z * S4);
  // cos(r)
  const double C0 = -0.499999997251031003120, C1 = 0.0416666233237390631894, C2 = -0.00138867637746099294692,
               C3 = 0.0000243904487962774090654;
  const double cr = ((1.0 + z * C0) + w * C1) + (w * z) * (C2 + z * C3);
  const double res = (q & 1) ? cr : sr;
  return (float)((q & 2) ? -res : res);
}
__device__ __forceinline__ float nm_sinf(float x) { return nm_sincos_sel(x, 0); }
// both at once from ONE range reduction (backward kernels: d sin = cos, d cos = -sin)
__device__ __forceinline__ void nm_sincosf(float x, float& sn, float& cs) {
  const double xd = (double)x;
  const double n = __builtin_rint(xd * 0.63661977236758134308);
  double r = __builtin_fma(-n, 1.57079632673412561417e+00, xd);
  r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
  const int q = (int)n;
  const double z = r * r;
  const double S1 = -0.166666666416265235595, S2 = 0.0083333293858894631756, S3 = -0.000198393348360966317347,
               S4 = 0.0000027183114939898219064;
  const double w = z * z;
  const double sr = (r + (z * r) * (S1 + z * S2)) + (z * r) * w * (S3 + z * S4);
  const double C0 = -0.499999997251031003120, C1 = 0.0416666233237390631894, C2 = -0.00138867637746099294692,
               C3 = 0.0000243904487962774090654;
  const double cr = ((1.0 + z * C0) + w * C1) + (w * z) * (C2 + z * C3);
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;  // quadrant 1, 3: sin <-> cos
  sn = (float)((q & 2) ? -s0 : s0);
  cs = (float)(((q + 1) & 2) ? -c0 : c0);
}
__device__ __forceinline__ float nm_cosf(float x) { return nm_sincos_sel(x, 1); }

__device__ __forceinline__ float nm_shfl_xor32(float v) { return __shfl_xor(v, 32, 64); }

// Exchange between the two 32-lane halves of a wavefront without the LDS round trip of ds_bpermute (gfx950:
// v_permlane32_swap_b32 swaps the upper half of its first operand with the lower half of its second).  After the call
// lo = the value of lane (l & 31), hi = the value of lane (l & 31) + 32, in BOTH halves: op(lo, hi) is the xor-32 reduction.
__device__ __forceinline__ void nm_swap32(float x, float& lo, float& hi) {
  lo = x;
  hi = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
}
// v_max3_f32 / v_max_f32 without the canonicalising max(x, x) the compiler puts in front of fmaxf on MFMA results
__device__ __forceinline__ float nm_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float nm_max16(const f32x16& v) {
  float a = nm_max3(v[0], v[1], v[2]), b = nm_max3(v[7], v[8], v[9]);
  a = nm_max3(a, v[3], v[4]);
  b = nm_max3(b, v[10], v[11]);
  a = nm_max3(a, v[5], v[6]);
  b = nm_max3(b, v[12], v[13]);
  a = nm_max3(a, v[14], v[15]);
  return nm_max3(a, b, b);
}

// Sum over the 32 lanes of each wavefront half with DPP (VALU only, no LDS round trips):
// xor-1 and xor-2 inside quads, row_half_mirror (8), row_mirror (16), then row_bcast15 adds row 0's total into row 1
// (and row 2's into row 3).  The full 32-lane sum is valid in lanes 16..31 (half 0) and 48..63 (half 1).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float nm_dpp(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float nm_half_sum_dpp(float x) {
  x += nm_dpp<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
  x += nm_dpp<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
  x += nm_dpp<0x141, 0xF>(x);  // row_half_mirror
  x += nm_dpp<0x140, 0xF>(x);  // row_mirror
  x += nm_dpp<0x142, 0xA>(x);  // row_bcast15 into rows 1 and 3
  return x;
}

// The same reduction for 8 independent values with the adds fused into the DPP instructions (5 instead of 10
// instructions per value).  The 8 chains are interleaved, so dependent DPP reads are 7 instructions apart and only
// the entry needs the 2 wait states between a VALU write and a DPP read of the same register.
__device__ __forceinline__ void nm_half_sum_dpp8(float (&v)[8]) {
#define NM_DPP_STEP(ctrl)                                   \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n"                     \
  "v_add_f32_dpp %1, %1, %1 " ctrl "\n"                     \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n"                     \
  "v_add_f32_dpp %3, %3, %3 " ctrl "\n"                     \
  "v_add_f32_dpp %4, %4, %4 " ctrl "\n"                     \
  "v_add_f32_dpp %5, %5, %5 " ctrl "\n"                     \
  "v_add_f32_dpp %6, %6, %6 " ctrl "\n"                     \
  "v_add_f32_dpp %7, %7, %7 " ctrl "\n"
  asm("s_nop 1\n"
      NM_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
      NM_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
      NM_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
      NM_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
      NM_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#undef NM_DPP_STEP
}
