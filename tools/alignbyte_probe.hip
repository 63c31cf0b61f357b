// Does v_alignbyte_b32 use only bits [1:0] of its shift operand on gfx950?  (If so, a reference's LDS byte address is its own shift.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  const unsigned s = threadIdx.x;  // 0..63
  out[s] = __builtin_amdgcn_alignbyte(0x88776655u, 0x44332211u, s);
}
int main() {
  unsigned *d, h[64];
  hipMalloc(&d, 256);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  int same = 1;
  for (int i = 0; i < 64; ++i) { if (h[i] != h[i & 3]) same = 0; }
  for (int i = 0; i < 12; ++i) printf("shift %d -> %08x\n", i, h[i]);
  printf("alignbyte_uses_low_2_bits_only=%d\n", same);
  return 0;
}
