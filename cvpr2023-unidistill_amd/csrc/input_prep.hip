// LiDAR input side: rigid / affine transforms of point clouds on the device.
//
// Reference (numpy, in the data loader): CollectLidarSweeps.forward
// (unidistill/data/multisensorfusion/transforms3d.py:379-414) moves every sweep into the key frame --
// xyz <- (M @ [x y z 1]^T)[:3] with M = inv(lidar_to_ego) @ inv(ego_to_global) @ sweep_pose @ lidar_to_ego
// in float64, stored back into the float32 cloud -- and writes the time lag into the 5th column;
// BevAffineTransformation.forward (:417-443) applies functional.bev_transform's matrix the same way.
// One launch handles a whole batch of segments (key frame + sweeps of every sample), each with its own
// matrix: out row r of segment s = transform(in row r).  float64 arithmetic in the reference's order
// (((m0 x + m1 y) + m2 z) + m3), no contraction, so the float32 results match numpy's bit for bit.
#include "ud_common.h"
#include "ud_prof.h"

namespace {

__global__ __launch_bounds__(256) void k_points_transform(const float* __restrict__ in, float* __restrict__ out,
                                                          const int64_t* __restrict__ seg,
                                                          const double* __restrict__ mats,
                                                          const float* __restrict__ last, int D) {
  const int s = blockIdx.y;
  const int64_t begin = seg[s], n = seg[s + 1] - begin;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* m = mats + (size_t)s * 16;
  const float* src = in + (size_t)(begin + i) * D;
  float* dst = out + (size_t)(begin + i) * D;
  const double x = src[0], y = src[1], z = src[2];
  float r[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) r[k] = (float)(((m[4 * k] * x + m[4 * k + 1] * y) + m[4 * k + 2] * z) + m[4 * k + 3]);
  for (int c = 3; c < D; ++c) dst[c] = src[c];          // before xyz: in-place calls read src first
  dst[0] = r[0];
  dst[1] = r[1];
  dst[2] = r[2];
  if (last) {
    const float v = last[s];
    if (v == v) dst[D - 1] = v;                         // NaN = keep the column
  }
}

}  // namespace

extern "C" int ud_points_transform(const float* in, float* out, const int64_t* seg, const double* mats,
                                   const float* last, int S, int D, int64_t max_rows, ud_stream_t stream_) {
  if (S == 0 || max_rows == 0) return UD_OK;
  if (!in || !out || !seg || !mats || S < 0 || S > 65535 || D < 3 || max_rows < 0) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("input.k_points_transform", stream);
  k_points_transform<<<dim3((unsigned)ud_div_up((long long)max_rows, 256), S), 256, 0, stream>>>(in, out, seg, mats,
                                                                                                  last, D);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
