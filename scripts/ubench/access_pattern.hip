// Microbenchmark: global memory throughput of the GEMM kernels' access patterns on an M x 256 fp32 matrix.
//   MODE 0  load, lane = row: lane (r, half) reads 32 B of row r per K-step (gemm_bf16x3_kernel's x loads)
//   MODE 1  load, coalesced: 8 lanes x 16 B = one 128-B line per row, 8 rows per instruction
//   MODE 2  store, lane = row: lane (r, half) writes 16 B pieces of row r (the epilogue's pattern)
//   MODE 3  store, coalesced: 32 lanes x 16 B = 512 contiguous bytes of a row, 2 rows per instruction
//   MODE 4  store of one 128-column chunk of an M x 768 matrix per wavefront (lane = row), chunk = blockIdx % 6: the y tiles of
//           the N = 768 GEMM (512-byte row segments at a 3 KiB stride)
//   MODE 5  MODE 0 + MODE 2 together (read x, write y: the N = 256 GEMM's traffic without the arithmetic)
// Build: hipcc --offload-arch=gfx950 -O3 access_pattern.hip -o access_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ x, float* __restrict__ y, int M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hi = lane >> 5;
  const int tile = blockIdx.x * 4 + wave;  // 32 rows x 256 columns per wavefront
  if (MODE != 4 && tile * 32 >= M) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    const float* p = x + (size_t)(tile * 32 + r) * 256 + 8 * hi;
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      acc += *reinterpret_cast<const f32x4*>(p + 16 * ks);
      acc += *reinterpret_cast<const f32x4*>(p + 16 * ks + 4);
    }
  } else if (MODE == 1) {
    const float* p = x + (size_t)(tile * 32 + (lane >> 3)) * 256 + 4 * (lane & 7);
#pragma unroll 4
    for (int kp = 0; kp < 8; ++kp)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += *reinterpret_cast<const f32x4*>(p + (size_t)(8 * i) * 256 + 32 * kp);
  } else if (MODE == 4) {
    // blockIdx -> (row group of 4 tiles, chunk): consecutive blocks = the 6 chunks of the same rows
    const int chunk = blockIdx.x % 6, rt = (blockIdx.x / 6) * 4 + wave;
    if (rt * 32 >= M) return;
    float* p = y + (size_t)(rt * 32 + r) * 768 + chunk * 128 + 4 * hi;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
#pragma unroll
    for (int c = 0; c < 16; ++c) *reinterpret_cast<f32x4*>(p + 8 * c) = v;
  } else if (MODE == 2) {
    float* p = y + (size_t)(tile * 32 + r) * 256 + 4 * hi;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
#pragma unroll
    for (int c = 0; c < 32; ++c) *reinterpret_cast<f32x4*>(p + 8 * c) = v;
  } else {
    float* p = y + (size_t)(tile * 32 + hi) * 256 + 4 * r;
    const f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      *reinterpret_cast<f32x4*>(p + (size_t)(2 * i) * 256) = v;
      *reinterpret_cast<f32x4*>(p + (size_t)(2 * i) * 256 + 128) = v;
    }
  }
  if (MODE == 5) {
    const float* p = x + (size_t)(tile * 32 + r) * 256 + 8 * hi;
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
      acc += *reinterpret_cast<const f32x4*>(p + 16 * ks);
      acc += *reinterpret_cast<const f32x4*>(p + 16 * ks + 4);
    }
    float* py = y + (size_t)(tile * 32 + r) * 256 + 4 * hi;
#pragma unroll
    for (int c = 0; c < 32; ++c) *reinterpret_cast<f32x4*>(py + 8 * c) = acc + (float)c;
    return;
  }
  if (MODE == 6 || MODE == 7) {  // 6: coalesced read + coalesced write; 7: lane=row read + coalesced write
    f32x4 v[16];
    if (MODE == 6) {
      const float* p = x + (size_t)(tile * 32 + (lane >> 3)) * 256 + 4 * (lane & 7);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(8 * (i & 3)) * 256 + 32 * (i >> 2));
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] += *reinterpret_cast<const f32x4*>(p + (size_t)(8 * (i & 3)) * 256 + 32 * (i >> 2) + 128);
    } else {
      const float* p = x + (size_t)(tile * 32 + r) * 256 + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) v[ks] = *reinterpret_cast<const f32x4*>(p + 16 * ks) + *reinterpret_cast<const f32x4*>(p + 16 * ks + 4);
    }
    float* py = y + (size_t)(tile * 32 + hi) * 256 + 4 * r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      *reinterpret_cast<f32x4*>(py + (size_t)(2 * i) * 256) = v[i];
      *reinterpret_cast<f32x4*>(py + (size_t)(2 * i) * 256 + 128) = v[i] + 1.f;
    }
    return;
  }
  if (MODE < 2 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) y[0] = acc[0];
}

template <int MODE>
void run(const float* x, float* y, int M, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = ((M / 32 + 3) / 4) * (MODE == 4 ? 6 : 1);
  for (int i = 0; i < 3; ++i) k<MODE><<<grid, 256>>>(x, y, M);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) k<MODE><<<grid, 256>>>(x, y, M);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)M * 1024 * (MODE == 4 ? 3 : MODE >= 5 ? 2 : 1);
  printf("%-36s %8.1f us  %6.2f TB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
}

int main() {
  const int M = 76800;
  float *x, *y;
  hipMalloc(&x, (size_t)M * 1024);
  hipMalloc(&y, (size_t)M * 3072);
  hipMemset(x, 0, (size_t)M * 1024);
  run<0>(x, y, M, "load  lane=row (32 B/lane)");
  run<1>(x, y, M, "load  coalesced");
  run<2>(x, y, M, "store lane=row (16 B/lane)");
  run<3>(x, y, M, "store coalesced");
  run<4>(x, y, M, "store 128-col chunks of 768 (lane=row)");
  run<5>(x, y, M, "read x + write y (lane=row)");
  run<6>(x, y, M, "read x + write y (coalesced)");
  run<7>(x, y, M, "read lane=row + write coalesced");
  return 0;
}
