// Shared between the two generations of the split-bf16 fused NeRF kernel (nerf_fwd_bf16.hip: one wavefront per SIMD;
// nerf_fwd_bf16_2w.hip: two wavefronts per SIMD): blob / LDS constants, the kernel argument block, small device helpers.
#pragma once
#include "common.h"
#include <string.h>

namespace nmbf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TILE = 128;
constexpr int SLOT_BYTES = 16384;
constexpr int SLOT_FLOATS = SLOT_BYTES / 4;
constexpr int NRING = 4;
constexpr int WS_WORKGROUPS = 512;  // upper bound of the persistent grid (one workgroup per CU); sizes the workspace
constexpr int XS = 6;    // K-steps of the 90(->96)-d IPE input
constexpr int HS = 16;   // K-steps of a 256-d hidden input
constexpr int VS = 3;    // K-steps of the 43(->48)-d [direction PE | appearance] input of the views layer
constexpr int NSLOT_NORGB = XS + 4 * HS + (XS + HS) + 2 * HS;       // layers 0..7          = 124
constexpr int NSLOT_FULL = NSLOT_NORGB + (HS + VS);                  // + views (feature_linear folded in) = 143

// small-parameter block (fp32), same layout as nerf_fwd.hip
constexpr int OFF_BIAS = 0, OFF_BVIEWS = 2304, OFF_WALPHA = 2432, OFF_WRGB = 2688, OFF_MISC = 3072;
// Round 4, fp16x3 only (the other modes carry ones): power-of-two operand scaling chosen at pack time (nm_nerf_pack_fp16x3_scaled).
//   OFF_SCALE   [16]  s_l, l = 0..8: the finished layer l is re-packed as fma(acc, s_l, bias'_l) -- s_l = 2^(c_{l+1} - A_l) takes the
//                     accumulator from its scale A_l (weight scale + input scale) to the input scale c_{l+1} of the next layer,
//                     bias'_l = bias_l * 2^c_{l+1} (stored in OFF_BIAS); exact: a power of two commutes with every rounding
//   OFF_DESCALE [8]   2^-c_{l+1}: back to true units, applied once per ray to the weight that multiplies the tapped activations
//   OFF_INSCALE [4]   2^c of the kernel-made inputs: IPE, direction PE, appearance row, (pad)
constexpr int OFF_SCALE = 3088, OFF_DESCALE = 3104, OFF_INSCALE = 3112, SMALL = 3120;
constexpr int NRANGE = 10;  // range telemetry slots: re-packed output of layers 0..8, views-layer extra inputs
constexpr int SMALL_PAD = 4096;  // floats reserved in the blob / LDS (16 KiB)
constexpr int NSLOT_PAD = 4;  // zero slots behind the last one: the two-wavefront kernel's weight stream runs that far past the end
constexpr size_t BLOB_BYTES = (size_t)SMALL_PAD * 4 + (size_t)(NSLOT_FULL + NSLOT_PAD) * SLOT_BYTES;

// LDS map (floats)
constexpr int LDS_SMALL = 0;
constexpr int LDS_RING = SMALL_PAD;
constexpr int LDS_IPE = LDS_RING + NRING * SLOT_FLOATS;        // [4 waves][XS][2][64][4 floats]
constexpr int LDS_SCR = LDS_IPE + 4 * XS * 2 * 64 * 4;          // per-sample scratch, see below
constexpr int LDS_FEAT = LDS_SCR + TILE * 12 + 32;              // [4 waves][256] partial feature sums
constexpr int LDS_EX = LDS_FEAT + 4 * 256;                     // [4 ray slots][48] views-layer extra inputs
constexpr int LDS_LEFT = LDS_EX + 4 * 48;                     // leftover list: [128] ray index, [128] transmittance
constexpr int LDS_RNG = LDS_LEFT + 2 * TILE;                   // [NRANGE][256] per-thread running maxima of the re-packed |values| (bits)
constexpr int LDS_TOTAL = LDS_RNG + NRANGE * 256;

struct NerfArgs {
  const char* blob;
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
  float* ws;  // [gridDim.x][4 wavefronts][32][64 lanes][4]: tapped activations of the tile in flight (fp32)
  int R, S, tap, white_bg, flags, ntiles;
  int Sa;    // samples per ray evaluated by the regular tiles (= S, or S/2 with NM_NERF_ZERO_TAIL)
  int left;  // 1: sample Sa of every ray is evaluated by "leftover" passes, samples > Sa have zero width (weight 0)
  int ntiles_full;       // tile count of the full evaluation (Sa = S)
  const int* tail_viol;  // device flag raised by nm_resample_ex when the zero-width premise does NOT hold: evaluate everything
  float var_scale;
  int* status;  // fp16x3: device int32[16] or NULL -- [0] |= 1 when an operand reached the fp16 limit, [1 + k] = max bits of range slot k
};

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// -DNM_TRACE: profiling build only -- the `raw` output becomes a [grid][32] table of s_memtime stamps of wavefront 0
#ifndef NM_TRACE
#define NM_TRACE 0
#endif

__host__ __device__ __forceinline__ constexpr int nrow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// the same for a wavefront-uniform value that must stay in an SGPR
__device__ __forceinline__ int launder_s(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

#if NM_TRACE
#define TRACE(i)                                                                                               \
  do {                                                                                                         \
    if (a.raw && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(a.raw)[bid * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define TRACE(i) do { } while (0)
#endif

// x = hi + lo with hi, lo bf16 (round to nearest even): 16 bits of mantissa survive.
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

// Element of a finished accumulator: one 32-bit cross-class copy (v_accvgpr_read_b32) at the point of use.  Without the
// opaque asm the compiler copies whole 16-register tuples around (and through scratch when it runs out of registers).
__device__ __forceinline__ float acc_read(float av) {
  asm volatile("" : "+v"(av));
  return av;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32 (round to nearest even)
  return __builtin_bit_cast(unsigned, bf16x2{(__bf16)a, (__bf16)b});
}
// Opaque identity: keeps a value (and the instructions that made it) in the basic block and at the position it was
// written -- without it LLVM sinks the re-packing arithmetic out of the MFMA stream into the block of its first use.
template <class T>
__device__ __forceinline__ void pin(T& v) {
  asm volatile("" : "+v"(v));
}
// fp32 sine for the positional encoding: q = rint(x / pi), 4-term Cody-Waite reduction (q * 3.140625 is exact up to
// q = 2^16), odd polynomial of degree 9 (SLEEF's sinf coefficients).  |error| <= 1e-7 for |x| < 6.5e4 (checked against
// fp64 on 8e4 random arguments).
__device__ __forceinline__ float sin32(float x) {
  const float q = __builtin_rintf(x * 0.318309886183790671537767526745028724f);
  float d = __builtin_fmaf(q, -3.140625f, x);
  d = __builtin_fmaf(q, -0.0009670257568359375f, d);
  d = __builtin_fmaf(q, -6.2771141529083251953e-07f, d);
  d = __builtin_fmaf(q, -1.2154201256553420762e-10f, d);
  const float s = d * d;
  d = ((int)q & 1) ? -d : d;
  float u = 2.6083159809786593541503e-06f;
  u = __builtin_fmaf(u, s, -0.0001981069071916863322258f);
  u = __builtin_fmaf(u, s, 0.00833307858556509017944336f);
  u = __builtin_fmaf(u, s, -0.166666597127914428710938f);
  return __builtin_fmaf(s, u * d, d);
}


}  // namespace nmbf
