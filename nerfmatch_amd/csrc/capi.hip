// ABI version / error strings of the C boundary (include/nerfmatch_amd.h).
#include "common.h"

#include <atomic>
#include <mutex>

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

// ---- compute-unit partitions (round 6) ---------------------------------------------------------------------------------------------
// A stream made by nm_stream_create_cu_mask runs its kernels on a subset of the device's compute units (hipExtStreamCreateWithCUMask);
// the persistent kernels (one workgroup per CU, static tile stride) size their grids by nm_stream_cus(stream) -- a grid of 256 workgroups
// on a 176-CU partition would run its last 80 workgroups as a second round.  The registry is the library's only process-wide state: up to
// NM_MAX_PART_STREAMS records (stream handle, CU count), written under a mutex, read lock-free by the launch paths.
namespace {
constexpr int NM_MAX_PART_STREAMS = 16;
struct PartRec { std::atomic<void*> s; std::atomic<int> cus; };
PartRec g_parts[NM_MAX_PART_STREAMS];
std::mutex g_parts_mu;
}  // namespace

extern "C" int nm_stream_cus(nmStream_t stream) {
  if (stream)
    for (int i = 0; i < NM_MAX_PART_STREAMS; ++i)
      if (g_parts[i].s.load(std::memory_order_acquire) == stream) return g_parts[i].cus.load(std::memory_order_relaxed);
  return nm_cu_count();
}

extern "C" int nm_stream_create_cu_mask(const uint32_t* mask_host, int n_words, nmStream_t* stream) {
  NM_CHECK_ARG(stream && mask_host && n_words > 0 && n_words <= 32);
  const int ncu = nm_cu_count();
  // Bit i of the mask selects compute unit i / 8 of XCD i % 8 on this chip, and within an XCD consecutive units alternate over its four
  // shader engines (scripts/ubench/cumask_probe.hip, profiles/r6_cumask_probe.log): a contiguous range of bits is spread evenly over the
  // eight XCDs.  The dispatcher deals the workgroups of a grid round-robin over XCDs and shader engines WITHOUT looking at the mask, so a
  // persistent one-workgroup-per-CU grid needs the same number of units in every (XCD, engine) -- a multiple of 32 bits in a contiguous
  // range -- or its surplus workgroups run as a second round (176 units: 2x the time of 160, profiles/r6_ab_render_stream.log).
  int cus = 0;
  for (int w = 0; w < n_words; ++w) {
    uint32_t m = mask_host[w];
    if (32 * w + 32 > ncu) m &= (32 * w >= ncu) ? 0u : ((1u << (ncu - 32 * w)) - 1u);
    cus += __builtin_popcount(m);
  }
  if (cus <= 0) return NM_ERR_ARG;
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_host) != hipSuccess) { (void)hipGetLastError(); return NM_ERR_LAUNCH; }
  std::lock_guard<std::mutex> lk(g_parts_mu);
  for (int i = 0; i < NM_MAX_PART_STREAMS; ++i)
    if (g_parts[i].s.load(std::memory_order_relaxed) == nullptr) {
      g_parts[i].cus.store(cus, std::memory_order_relaxed);
      g_parts[i].s.store((void*)s, std::memory_order_release);
      *stream = (nmStream_t)s;
      return NM_OK;
    }
  (void)hipStreamDestroy(s);
  return NM_ERR_WORKSPACE;  // registry full
}

extern "C" int nm_stream_destroy(nmStream_t stream) {
  NM_CHECK_ARG(stream);
  {
    std::lock_guard<std::mutex> lk(g_parts_mu);
    bool found = false;
    for (int i = 0; i < NM_MAX_PART_STREAMS; ++i)
      if (g_parts[i].s.load(std::memory_order_relaxed) == stream) { g_parts[i].s.store(nullptr, std::memory_order_release); found = true; }
    if (!found) return NM_ERR_ARG;  // not one of ours: the caller destroys its own streams
  }
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? NM_OK : NM_ERR_LAUNCH;
}

// ---- parameter fingerprints (round 6) -----------------------------------------------------------------------------
// The python side caches copies DERIVED from parameters (packed / split / transposed weight blobs, the temperature's host value) under
// (data_ptr, _version) keys.  A write through `.data` -- p.data.clamp_(), the idiom for clamping a learned temperature -- changes the
// values and neither key: the caches would serve old weights, silently.  One launch per forward pass sums every parameter's words with
// position-dependent odd multipliers (64-bit wrap-around arithmetic: exact in any order) and compares with the sums taken when the caches
// were filled; a mismatch raises a device flag that travels to the host with the match counts the pass reads back anyway -- the pass is
// then repeated on fresh copies.  No synchronisation of its own.
namespace {
constexpr int FP_CHUNK = 4096;  // words per workgroup (ops.ParamGuard.CHUNK): four 16-byte loads per thread, all in flight at once
__global__ void __launch_bounds__(256) fingerprint_kernel(const void* const* __restrict__ ptrs, const long long* __restrict__ words,
                                                          const int* __restrict__ blk_tensor, const long long* __restrict__ blk_off, int n_tensors,
                                                          unsigned long long* __restrict__ cur, unsigned long long* __restrict__ ref,
                                                          int* __restrict__ ctrl, int baseline) {
  const int t = blk_tensor[blockIdx.x];
  const long long off = blk_off[blockIdx.x], nw = words[t];
  const unsigned* p = reinterpret_cast<const unsigned*>(ptrs[t]);
  const long long end = (off + FP_CHUNK < nw) ? off + FP_CHUNK : nw;
  unsigned long long s = 0;
  if ((reinterpret_cast<uintptr_t>(p + off) & 15) == 0) {
    // (a first version read one word per thread and iteration, 64 dependent-latency rounds per workgroup: 33 us for the matcher's 13 MB)
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = off + 4 * (k * 256 + (int)threadIdx.x);
      v[k] = (i + 3 < end) ? *reinterpret_cast<const uint4*>(p + i) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long i = off + 4 * (k * 256 + (int)threadIdx.x);
      const unsigned long long m = (unsigned long long)(2 * i + 1);
      if (i + 3 < end) s += v[k].x * m + v[k].y * (m + 2) + v[k].z * (m + 4) + v[k].w * (m + 6);
      else for (long long j = i; j < end; ++j) s += (unsigned long long)p[j] * (unsigned long long)(2 * j + 1);
    }
  } else {
    for (long long i = off + threadIdx.x; i < end; i += 256) s += (unsigned long long)p[i] * (unsigned long long)(2 * i + 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ unsigned long long part[4];
  __shared__ int last;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(cur + t, part[0] + part[1] + part[2] + part[3]);
    __threadfence();
    last = atomicAdd(ctrl, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  int diff = 0;
  for (int i = threadIdx.x; i < n_tensors; i += 256) {
    const unsigned long long c = __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (baseline) ref[i] = c;
    else diff |= (c != ref[i]);
    cur[i] = 0;  // ready for the next launch
  }
  diff = __syncthreads_or(diff);
  if (threadIdx.x == 0) {
    if (baseline) ctrl[1] = 0;
    else if (diff) ctrl[1] = 1;  // sticky until the next baseline
    ctrl[0] = 0;
  }
}
}  // namespace

extern "C" int nm_params_fingerprint(const void* const* ptrs_dev, const long long* words_dev, const int* blk_tensor_dev, const long long* blk_off_dev,
                                     int n_tensors, int n_blocks, unsigned long long* cur_dev, unsigned long long* ref_dev, int* ctrl_dev, int baseline,
                                     nmStream_t stream) {
  NM_CHECK_ARG(ptrs_dev && words_dev && blk_tensor_dev && blk_off_dev && cur_dev && ref_dev && ctrl_dev && n_tensors > 0 && n_blocks > 0);
  fingerprint_kernel<<<n_blocks, 256, 0, (hipStream_t)stream>>>(ptrs_dev, words_dev, blk_tensor_dev, blk_off_dev, n_tensors, cur_dev, ref_dev, ctrl_dev,
                                                                baseline);
  return nm_launch_status();
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
