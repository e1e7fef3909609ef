// Sustained MFMA issue rate with nothing else going on: every wave of a full grid (2 workgroups of 4 waves per CU) runs
// independent MFMAs on eight accumulators, no memory traffic.  Prints TFLOP/s for the fp32 (16x16x4) and bf16 (16x16x32)
// instructions -- the ceiling the convolution kernels can be compared with on THIS machine (clocks under matrix load are
// not the nominal 2.4 GHz).   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool BF>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters) {
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f / (float)(1 + (threadIdx.x & 3));
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)a; bb[e] = (__bf16)b; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if constexpr (BF) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_mfma32(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  bf16x8 ab, bb;
  for (int e = 0; e < 8; ++e) { ab[e] = (__bf16)(0.125f * (threadIdx.x & 7)); bb[e] = (__bf16)(1.0f / (1 + (threadIdx.x & 3))); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) s += acc[t][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma32_f32(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const float a = (float)(threadIdx.x & 7) * 0.125f, b = 1.0f / (float)(1 + (threadIdx.x & 3));
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) s += acc[t][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
void run32_f32(int wgs_per_cu) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, grid = cus * wgs_per_cu, iters = 20000;
  float* out;
  (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k_mfma32_f32<<<grid, 256>>>(out, 1000);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_mfma32_f32<<<grid, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
    printf("fp32 v_mfma_f32_32x32x2_f32   %d CUs x %d workgroups: %.3f ms  %.1f TFLOP/s  (%.1f %% of the nominal 157)\n", cus, wgs_per_cu, ms,
           fl / ms / 1e9, 100.0 * fl / ms / 1e9 / 157.3);
  }
  (void)hipFree(out);
}
void run32(int wgs_per_cu) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, grid = cus * wgs_per_cu, iters = 40000;
  float* out;
  (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k_mfma32<<<grid, 256>>>(out, 1000);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    k_mfma32<<<grid, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
    printf("bf16 v_mfma_f32_32x32x16_bf16  %d CUs x %d workgroups: %.3f ms  %.1f TFLOP/s  (%.1f %% of the nominal 2517)\n", cus, wgs_per_cu, ms,
           fl / ms / 1e9, 100.0 * fl / ms / 1e9 / 2516.6);
  }
  (void)hipFree(out);
}

template <bool BF>
void run(const char* name, double flop_per_instr, double nominal_tf, int wgs_per_cu, int cus_used = 0) {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = cus_used ? cus_used : p.multiProcessorCount, grid = cus * wgs_per_cu, iters = BF ? 40000 : 20000;
  nominal_tf = nominal_tf * cus / p.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k_mfma<BF><<<grid, 256>>>(out, 1000);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k_mfma<BF><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 8 * flop_per_instr;
    printf("%s  %d CUs x %d workgroups: %.3f ms  %.1f TFLOP/s  (%.1f %% of the nominal %.0f; implied clock %.2f GHz)\n", name, cus,
           wgs_per_cu, ms, fl / ms / 1e9, 100.0 * fl / ms / 1e9 / nominal_tf, nominal_tf,
           fl / ms / 1e9 / nominal_tf * 2.4);
  }
  hipFree(out);
}

int main() {
  run<false>("fp32 v_mfma_f32_16x16x4_f32 ", 2.0 * 16 * 16 * 4, 157.3, 2);
  run<false>("fp32 v_mfma_f32_16x16x4_f32 ", 2.0 * 16 * 16 * 4, 157.3, 1);
  run<true>("bf16 v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 2516.6, 2);
  run<true>("bf16 v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 2516.6, 4);
  run<true>("bf16 v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 2516.6, 8);
  run<false>("fp32 v_mfma_f32_16x16x4_f32 ", 2.0 * 16 * 16 * 4, 157.3, 4);
  run32_f32(1);
  run32_f32(2);
  run32_f32(4);
  run32(2);
  run32(4);
  // a few workgroups only (the chip draws little power): is the rate per CU the same?  (one workgroup per XCD-round-robin slot)
  run<true>("bf16, 8 workgroups           ", 2.0 * 16 * 16 * 32, 2516.6, 1, 8);
  run<false>("fp32, 8 workgroups           ", 2.0 * 16 * 16 * 4, 157.3, 1, 8);
  return 0;
}
