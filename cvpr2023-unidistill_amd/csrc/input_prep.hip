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

// ---- camera side ----------------------------------------------------------------------------------------
// ImageNormalize.forward (transforms3d.py:350-368) -> mmcv.imnormalize(np.array(img), mean, std, to_rgb) (mmcv is a
// third-party dependency, absent from the tree: published algorithm = float32 image, optional channel reversal
// (cvtColor BGR2RGB), img - float32(mean), then * float32(1 / float64(std)), in that order), followed by the
// dataset's HWC -> CHW permute + stack (nuscenes_multimodal.py:262-293).  One launch for all images of a batch;
// the loader ships uint8 pixels (a quarter of the bytes of the reference's float32 tensors).
// out_cl == 0: out f32 [NI][3][H][W] (the reference's layout); out_cl == 1: [NI][H][W][3] memory (channels-last view).
__global__ __launch_bounds__(256) void k_image_normalize(const unsigned char* __restrict__ img,
                                                         float* __restrict__ out, float m0, float m1, float m2,
                                                         float s0, float s1, float s2, int to_rgb, long long npix_img,
                                                         long long total, int out_cl) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;      // one pixel of one image
  if (t >= total) return;
  const unsigned char* p = img + t * 3;
  float v0 = (float)p[0], v1 = (float)p[1], v2 = (float)p[2];
  if (to_rgb) {
    const float tmp = v0;
    v0 = v2;
    v2 = tmp;
  }
  v0 = __fmul_rn(__fsub_rn(v0, m0), s0);
  v1 = __fmul_rn(__fsub_rn(v1, m1), s1);
  v2 = __fmul_rn(__fsub_rn(v2, m2), s2);
  if (out_cl) {
    out[t * 3 + 0] = v0;
    out[t * 3 + 1] = v1;
    out[t * 3 + 2] = v2;
  } else {
    const long long im = t / npix_img, q = t - im * npix_img;
    float* o = out + im * 3 * npix_img + q;
    o[0] = v0;
    o[npix_img] = v1;
    o[2 * npix_img] = v2;
  }
}

// collate_fn's fill_batch_tensor for ragged samples (nuscenes_multimodal.py:441-463): out[b, :len_b] = sample b,
// zero rows up to the longest sample.  Up to kCollateMax samples per launch, their device pointers by value.
constexpr int kCollateMax = 32;
struct CollateArgs {
  const float* src[kCollateMax];
  long long rows[kCollateMax];
};
__global__ __launch_bounds__(256) void k_collate_pad(CollateArgs a, float* __restrict__ out, long long L, int W) {
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // element of sample b's padded [L, W] slab
  if (i >= L * W) return;
  const long long r = i / W;
  out[(size_t)b * L * W + i] = (r < a.rows[b]) ? a.src[b][i] : 0.0f;
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

extern "C" int ud_image_normalize(const unsigned char* img, float* out, const float* mean, const float* std,
                                  int to_rgb, int NI, int H, int W, int out_channels_last, ud_stream_t stream_) {
  if (NI == 0) return UD_OK;
  if (!img || !out || !mean || !std || NI < 0 || H <= 0 || W <= 0) return UD_ERR_INVALID_ARG;
  for (int c = 0; c < 3; ++c)
    if (!(std[c] != 0.0f)) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const long long npix = (long long)H * W, total = npix * NI;
  // mmcv: mean -> float64, stdinv = 1 / float64(std); OpenCV applies both to the float32 image in float32
  const float s0 = (float)(1.0 / (double)std[0]), s1 = (float)(1.0 / (double)std[1]), s2 = (float)(1.0 / (double)std[2]);
  UdProfScope prof("input.k_image_normalize", stream);
  k_image_normalize<<<ud_div_up(total, 256), 256, 0, stream>>>(img, out, mean[0], mean[1], mean[2], s0, s1, s2,
                                                               to_rgb ? 1 : 0, npix, total, out_channels_last ? 1 : 0);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_collate_pad(const float* const* samples, const int64_t* rows, int B, int64_t L, int W, float* out,
                              ud_stream_t stream_) {
  if (B == 0 || L == 0) return UD_OK;
  if (!samples || !rows || !out || B < 0 || L < 0 || W <= 0) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  for (int b0 = 0; b0 < B; b0 += kCollateMax) {
    CollateArgs a;
    const int nb = B - b0 < kCollateMax ? B - b0 : kCollateMax;
    for (int i = 0; i < kCollateMax; ++i) {
      a.src[i] = i < nb ? samples[b0 + i] : nullptr;
      a.rows[i] = i < nb ? rows[b0 + i] : 0;
      if (i < nb && (a.rows[i] < 0 || a.rows[i] > L || (a.rows[i] > 0 && !a.src[i]))) return UD_ERR_INVALID_ARG;
    }
    k_collate_pad<<<dim3((unsigned)ud_div_up((long long)L * W, 256), nb), 256, 0, stream>>>(
        a, out + (size_t)b0 * L * W, L, W);
    UD_LAUNCH_CHECK();
  }
  return UD_OK;
}
