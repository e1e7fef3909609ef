// LDS-DMA (global_load_lds_dwordx4) fill rate from an L2-resident region, alone and beside v_mfma_f32_32x32x16_bf16:
// what bounds a conv kernel that stages weights per tap.  Each wave issues PIECES 1-KiB pieces per round into its own LDS
// slots from a `region`-byte window (all workgroups read the same window: weights), waits, repeats.
//   hipcc --offload-arch=gfx950 -O3 tools/dma_rate.hip -o /tmp/dma_rate && /tmp/dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(const char* src, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

template <int PIECES, int MFMAS>
__global__ __launch_bounds__(512) void k_dma(const char* __restrict__ src, size_t region, int rounds, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(0.125f * (threadIdx.x & 7)); bb[e] = (__bf16)(1.0f / (1 + (threadIdx.x & 3))); }
  size_t off = ((size_t)blockIdx.x * 7919 * 1024 + (size_t)wave * PIECES * 1024) % region;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      dma16(src + off + lane * 16, smem + (size_t)(wave * PIECES + p) * 1024);
      off += 1024;
      if (off >= region) off -= region;
    }
    off += (size_t)(nw - 1) * PIECES * 1024;
    off %= region;
#pragma unroll
    for (int m = 0; m < MFMAS; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[m & 3], 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) s += acc[t][e];
  s += (float)smem[lane * 4];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int PIECES, int MFMAS>
void run(const char* src, size_t region, int threads, int wg_per_cu, float* out) {
  const int cus = 256, grid = cus * wg_per_cu, rounds = 4000;
  const size_t lds = (size_t)(threads / 64) * PIECES * 1024;
  (void)hipFuncSetAttribute((const void*)k_dma<PIECES, MFMAS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k_dma<PIECES, MFMAS><<<grid, threads, lds>>>(src, region, 100, out);
  (void)hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_dma<PIECES, MFMAS><<<grid, threads, lds>>>(src, region, rounds, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double bytes = (double)grid * (threads / 64) * PIECES * 1024.0 * rounds;
  const double fl = (double)grid * (threads / 64) * MFMAS * (double)rounds * 2.0 * 32 * 32 * 16;
  printf("region %7zu KB  %3d thr x %d WG/CU  %2d pieces + %2d MFMA per wave-round: %7.3f ms  DMA %6.2f TB/s  MFMA %7.1f TFLOP/s\n",
         region >> 10, threads, wg_per_cu, PIECES, MFMAS, best, bytes / best / 1e9,
         fl / best / 1e9);
}

int main() {
  char* src;
  const size_t cap = 64u << 20;
  (void)hipMalloc(&src, cap);
  (void)hipMemset(src, 1, cap);
  float* out;
  (void)hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
  for (size_t region : {(size_t)288 << 10, (size_t)3 << 20, (size_t)48 << 20}) {
    run<4, 0>(src, region, 256, 2, out);
    run<8, 0>(src, region, 256, 2, out);
    run<4, 0>(src, region, 512, 1, out);
    run<8, 0>(src, region, 512, 1, out);
    run<2, 16>(src, region, 512, 1, out);
    run<4, 16>(src, region, 512, 1, out);
    run<4, 16>(src, region, 256, 2, out);
    run<2, 8>(src, region, 256, 2, out);
    run<4, 8>(src, region, 256, 2, out);
    run<6, 16>(src, region, 256, 2, out);
  }
  run<0, 16>(src, 288 << 10, 512, 1, out);
  run<0, 16>(src, 288 << 10, 256, 2, out);
  return 0;
}
