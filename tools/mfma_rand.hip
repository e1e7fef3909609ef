// v_mfma_f32_32x32x16_bf16 issue rate with RANDOM operands rotating through 8 fragment pairs (data-dependent power: the
// chip clocks to its power budget) vs constant operands, 2 workgroups x 4 waves or 1 x 8 per CU, plus the shader clock
// measured with s_memtime against the wall clock.   hipcc --offload-arch=gfx950 -O3 tools/mfma_rand.hip -o /tmp/mfma_rand
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(512) void k(const s8* __restrict__ ops, float* out, long long* cyc, int iters, int random) {
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  s8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = ops[(random ? (i * 2) * 64 + (threadIdx.x & 63) : 0)];
    b[i] = ops[(random ? (i * 2 + 1) * 64 + (threadIdx.x & 63) : 1)];
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int t = 0; t < NACC; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[(i + t) & 7]), acc[t], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int e = 0; e < 16; ++e) s += acc[t][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int n = 16 * 64;
  s8* h = (s8*)malloc(n * sizeof(s8));
  srand(1);
  for (int i = 0; i < n; ++i)
    for (int e = 0; e < 8; ++e) {
      float v = (float)rand() / RAND_MAX * 2.f - 1.f;
      unsigned u;
      memcpy(&u, &v, 4);
      h[i][e] = (short)(u >> 16);
    }
  h[0] = h[2]; h[1] = h[3];
  s8* d; float* out; long long* cyc;
  (void)hipMalloc(&d, n * sizeof(s8));
  (void)hipMemcpy(d, h, n * sizeof(s8), hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 512 * 512 * sizeof(float));
  (void)hipMalloc(&cyc, 512 * sizeof(long long));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int cfg = 0; cfg < 2; ++cfg)
    for (int random = 0; random < 2; ++random) {
      const int threads = cfg ? 512 : 256, grid = cfg ? 256 : 512, iters = 4000;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k<4><<<grid, threads>>>(d, out, cyc, iters, random);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        long long c0;
        (void)hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
        const double fl = (double)grid * (threads / 64) * iters * 8 * 4 * 2.0 * 32 * 32 * 16;
        if (rep == 2)
          printf("%s operands, %d thr x %d WG/CU: %.3f ms  %.0f TFLOP/s  block 0: %lld cycles -> shader clock %.2f GHz, %.1f cycles per MFMA per SIMD\n",
                 random ? "random  " : "constant", threads, cfg ? 1 : 2, ms, fl / ms / 1e9, c0, c0 / (ms * 1e6),
                 (double)c0 / (iters * 8 * 4 * 2.0));
      }
    }
  return 0;
}
