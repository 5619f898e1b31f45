// micro test of gate_byte (nerf_fwd_bf16.hip): 4 packed words -> 8 gate bits
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned gate_byte(const u32x4& h) {
  // v_pk_min_u16 against (1, 1): 0 / 1 per half.  (Inline asm on scalars: the vector-typed __builtin_elementwise_min on a bit-cast
  // element of the ext-vector came out reading word 0 four times -- scripts/ubench/gate_byte.hip.)
  const unsigned w0 = h[0], w1 = h[1], w2 = h[2], w3 = h[3];
  unsigned m0, m1, m2, m3;
  const unsigned one = 0x00010001u;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m0) : "v"(w0), "v"(one));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m1) : "v"(w1), "v"(one));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m2) : "v"(w2), "v"(one));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m3) : "v"(w3), "v"(one));
  const unsigned t = m0 | (m1 << 1) | (m2 << 2) | (m3 << 3);
  return (t & 0xfu) | ((t >> 12) & 0xf0u);
}
__global__ void k(const u32x4* in, unsigned* out) { out[threadIdx.x] = gate_byte(in[threadIdx.x]); }
int main() {
  u32x4 h[4] = {{0x3f800000u, 0x00003f80u, 0x3f803f80u, 0u}, {0x00003f80u, 0x00003f80u, 0x00003f80u, 0x00003f80u}, {0u, 0u, 0u, 0x3f800000u}, {0x3f803f80u, 0u, 0u, 0x3f803f80u}};
  u32x4* d; unsigned* o; unsigned r[4];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 4>>>(d, o); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  // expected: word k low half -> bit k, high half -> bit 4 + k
  const unsigned exp[4] = {0x10 | 0x02 | 0x04 | 0x40, 0x0f, 0x80, 0x11 | 0x88};
  for (int i = 0; i < 4; ++i) printf("case %d: got 0x%02x expected 0x%02x\n", i, r[i], exp[i]);
  return 0;
}
