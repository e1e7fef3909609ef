// Probe ds_read_b64_tr_b16 semantics on gfx950: LDS holds s[i] = i (u16); lane l reads with byte
// address given by pattern P; prints the 4 u16 each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int pattern) {
  __shared__ unsigned short s[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) s[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (pattern == 0) addr = l * 8;                       // consecutive 8-byte chunks
  else if (pattern == 1) addr = (l & 15) * 256 + (l >> 4) * 8;   // 16 rows of 128 elements, lane group picks column chunk
  else addr = (l & 15) * 32 + (l >> 4) * 8;             // 16 rows of 16 elements (32 B rows)
  addr += (unsigned)(size_t)s;   // LDS base offset (generic->lds low bits)
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[l * 4 + 0] = v & 0xFFFF; out[l * 4 + 1] = (v >> 16) & 0xFFFF;
  out[l * 4 + 2] = (v >> 32) & 0xFFFF; out[l * 4 + 3] = (v >> 48) & 0xFFFF;
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int p = 0; p < 3; ++p) {
    probe<<<1, 64>>>(d, p); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], (l % 2) ? "\n" : "   |");
  }
  return 0;
}
