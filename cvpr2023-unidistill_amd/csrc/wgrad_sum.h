// Fixed-order reduction of the weight-gradient kernels' per-slice partial sums (conv2d.hip, conv2d_f32_wgrad.hip):
// deterministic split-K without atomics.
#ifndef UD_WGRAD_SUM_H_
#define UD_WGRAD_SUM_H_
#include "ud_common.h"

namespace {

// out[i] = sum over slices of partial[s][i], i in float4 units: a workgroup owns 64 float4 outputs, its four
// waves each add a quarter of the slices (ascending), the four sub-sums are combined in wave order.
__global__ __launch_bounds__(256) void k_wgrad_sum(const float* __restrict__ partial, int slices, size_t n,
                                                   float* __restrict__ out) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t i4 = (size_t)blockIdx.x * 64 + lane, n4 = n / 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i4 < n4) {
    const int per = (slices + 3) / 4, s0 = wave * per, s1 = min(slices, s0 + per);
    auto ld = [&](int s) { return *reinterpret_cast<const float4*>(partial + (size_t)s * n + 4 * i4); };
    int s = s0;
    for (; s + 3 < s1; s += 4) {          // four loads in flight per trip; the additions keep the ascending order
      const float4 v0 = ld(s), v1 = ld(s + 1), v2 = ld(s + 2), v3 = ld(s + 3);
      a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
      a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
      a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
      a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
    }
    for (; s < s1; ++s) {
      const float4 v = ld(s);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  part[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && i4 < n4) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 v = part[w][lane];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(out + 4 * i4) = a;
  }
}

// n % 4 == 0
inline int ud_wgrad_sum(const float* partial, int slices, size_t n, float* out, hipStream_t stream) {
  k_wgrad_sum<<<ud_div_up((long long)(n / 4), 64), 256, 0, stream>>>(partial, slices, n, out);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

}  // namespace
#endif  // UD_WGRAD_SUM_H_
