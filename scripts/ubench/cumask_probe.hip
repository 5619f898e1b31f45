// CU-masked streams on gfx950 (round 6): does hipExtStreamCreateWithCUMask work for an ordinary user on this pool, which (XCC, SE, CU) does
// bit i of the mask select, what does hipExtStreamGetCUMask cost per call, and do a long kernel on a masked stream and short kernels on an
// unmasked one really run side by side?
//   hipcc -O2 --offload-arch=gfx950 scripts/ubench/cumask_probe.hip -o scripts/ubench/cumask_probe && scripts/ubench/cumask_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);         \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void where_kernel(uint32_t* out, int spin) {
  uint32_t xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // keep the workgroup resident for a while so that the grid has to spread over every CU the queue may use
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc & 0xf;
    out[2 * blockIdx.x + 1] = hwid;
  }
}

__global__ void spin_kernel(long long ticks, long long* stamp) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) { stamp[0] = t0; stamp[1] = wall_clock64(); }
}

__global__ void short_kernel(long long* stamp, int i) {
  if (threadIdx.x == 0 && blockIdx.x == 0) stamp[i] = wall_clock64();
}

static int survey(const std::vector<uint32_t>& mask, const char* what, uint32_t* d_out, std::vector<uint32_t>& h) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask -> %s\n", what, hipGetErrorString(e)); return 1; }
  const int G = 4096;
  where_kernel<<<G, 64, 0, s>>>(d_out, 20000);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h.data(), d_out, G * 8, hipMemcpyDeviceToHost));
  std::map<int, std::set<uint32_t>> per;
  for (int b = 0; b < G; ++b) per[h[2 * b]].insert((h[2 * b + 1] >> 8) & 0xff);  // cu_id[11:8] sh_id[12] se_id[15:13]
  int tot = 0;
  printf("%-34s", what);
  for (auto& kv : per) { printf(" xcc%d:%2zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
  printf("  total %d CUs\n", tot);
  uint32_t back[8] = {0};
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 1000; ++i) hipExtStreamGetCUMask(s, 8, back);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1000;
  int pc = 0;
  for (int i = 0; i < 8; ++i) pc += __builtin_popcount(back[i]);
  printf("    hipExtStreamGetCUMask: %d bits set, %.2f us per call\n", pc, us);
  CK(hipStreamDestroy(s));
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
  uint32_t* d_out;
  CK(hipMalloc(&d_out, 4096 * 8));
  std::vector<uint32_t> h(4096 * 2);
  auto low = [](int n) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < n; ++i) m[i / 32] |= 1u << (i % 32); return m; };
  survey(low(256), "all 256 bits", d_out, h);
  survey(low(176), "bits 0..175", d_out, h);
  survey(low(128), "bits 0..127", d_out, h);
  survey(low(64), "bits 0..63", d_out, h);
  survey(low(8), "bits 0..7", d_out, h);
  { std::vector<uint32_t> m(8, 0); for (int i = 176; i < 256; ++i) m[i / 32] |= 1u << (i % 32); survey(m, "bits 176..255", d_out, h); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 256; i += 2) m[i / 32] |= 1u << (i % 32); survey(m, "even bits", d_out, h); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xffffffffu; survey(m, "word 0 only", d_out, h); }
  {
    hipStream_t s0;
    CK(hipStreamCreate(&s0));
    uint32_t back[8] = {0};
    hipError_t e = hipExtStreamGetCUMask(s0, 8, back);
    int pc = 0;
    for (int i = 0; i < 8; ++i) pc += __builtin_popcount(back[i]);
    printf("plain stream: hipExtStreamGetCUMask -> %s, %d bits set\n", hipGetErrorString(e), pc);
    e = hipExtStreamGetCUMask(nullptr, 8, back);
    pc = 0;
    for (int i = 0; i < 8; ++i) pc += __builtin_popcount(back[i]);
    printf("null stream: hipExtStreamGetCUMask -> %s, %d bits set\n", hipGetErrorString(e), pc);
  }
  // side by side: a 2 ms spin on a stream masked to 176 CUs (176 workgroups of 256 threads), twenty dependent short kernels on a plain stream
  {
    hipStream_t a, b;
    auto m = low(176);
    CK(hipExtStreamCreateWithCUMask(&a, 8, m.data()));
    CK(hipStreamCreate(&b));
    long long* st;
    CK(hipMalloc(&st, 64 * 8));
    CK(hipMemset(st, 0, 64 * 8));
    for (int rep = 0; rep < 2; ++rep) {
      spin_kernel<<<176, 256, 0, a>>>(200000, st);  // 100 MHz wall clock: 2 ms
      for (int i = 0; i < 20; ++i) short_kernel<<<256, 256, 0, b>>>(st, 2 + i);
      CK(hipDeviceSynchronize());
    }
    long long hs[64];
    CK(hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost));
    printf("masked spin: start 0, end %.1f us; short kernels on the plain stream at", (hs[1] - hs[0]) / 100.0);
    for (int i = 0; i < 20; i += 4) printf(" %.1f", (hs[2 + i] - hs[0]) / 100.0);
    printf(" us (inside the spin = side by side)\n");
    // and the other way round: the spin needs every CU (256 workgroups x 1024 threads would do; here 256 x 256 with 64 KB LDS each is not needed -- a full-mask persistent grid)
    CK(hipMemset(st, 0, 64 * 8));
    hipStream_t c;
    CK(hipStreamCreate(&c));
    spin_kernel<<<256, 1024, 0, c>>>(200000, st);
    for (int i = 0; i < 20; ++i) short_kernel<<<256, 256, 0, b>>>(st, 2 + i);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hs, st, sizeof(hs), hipMemcpyDeviceToHost));
    printf("unmasked spin (256 x 1024 threads): end %.1f us; short kernels at", (hs[1] - hs[0]) / 100.0);
    for (int i = 0; i < 20; i += 4) printf(" %.1f", (hs[2 + i] - hs[0]) / 100.0);
    printf(" us\n");
  }
  return 0;
}
