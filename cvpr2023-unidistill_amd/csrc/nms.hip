// Rotated-BEV IoU and greedy NMS for the proposal layer (eval / predict_boxes_when_training path).
//
// Replaces `iou3d_nms_cuda.nms_gpu` (reference unidistill/layers/head/det3d/generate_proposals/
// centerpoint_gen_proposals.py:85-105; the extension binary is not in the reference tree -- it is the
// OpenPCDet iou3d_nms op, whose published behaviour is restated here): boxes are
// (x, y, z, dx, dy, dz, heading), already sorted by descending score; box i suppresses every later box
// j with BEV IoU(i, j) > thresh unless i itself was suppressed.  BEV footprint = dx x dy rectangle
// rotated by heading about (x, y).  Intersection = Sutherland-Hodgman clipping of rectangle A by the
// four half-planes of rectangle B, shoelace area; IoU = inter / max(area_a + area_b - inter, 1e-8).
//
//   k_nms_mask   : one thread per (box i, block of 64 later boxes) -> 64 suppression bits
//   k_nms_reduce : ONE wave walks the boxes in score order (the inherently sequential part, kept on
//                  the device so the proposal layer never synchronises with the host): lane w owns
//                  word w of the running "removed" bitmap; keep list and count are written out.
#include "ud_common.h"
#include "ud_prof.h"
#include "bev_iou.h"

namespace {

using ud_iou::iou_bev;

__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ boxes, int N, float thresh,
                                                  unsigned long long* __restrict__ mask, int words) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)N * words) return;
  const int i = (int)(t / words), wj = (int)(t - (long long)i * words);
  unsigned long long bits = 0ull;
  const int j0 = wj * 64;
  if (j0 + 63 > i) {                              // only later boxes can be suppressed by i
    float bi[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) bi[k] = boxes[(size_t)i * 7 + k];
    for (int jj = 0; jj < 64; ++jj) {
      const int j = j0 + jj;
      if (j <= i || j >= N) continue;
      float bj[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) bj[k] = boxes[(size_t)j * 7 + k];
      if (iou_bev(bi, bj) > thresh) bits |= 1ull << jj;
    }
  }
  mask[(size_t)i * words + wj] = bits;
}

__global__ __launch_bounds__(64) void k_nms_reduce(const unsigned long long* __restrict__ mask, int N,
                                                   int words, long long* __restrict__ keep,
                                                   int* __restrict__ num_keep) {
  const int lane = threadIdx.x;
  // lane w owns removed-bitmap words w, w + 64, ... (N <= 64 * 64 * kWordsPerLane)
  constexpr int kWordsPerLane = 4;
  unsigned long long removed[kWordsPerLane] = {0ull, 0ull, 0ull, 0ull};
  int count = 0;
  for (int i = 0; i < N; ++i) {
    const int w = i >> 6, owner = w & 63, slot = w >> 6;
    unsigned long long word = 0ull;
#pragma unroll
    for (int s = 0; s < kWordsPerLane; ++s)
      if (s == slot) word = removed[s];
    const unsigned lo = __shfl((unsigned)(word & 0xFFFFFFFFull), owner);
    const unsigned hi = __shfl((unsigned)(word >> 32), owner);
    const unsigned long long cur = ((unsigned long long)hi << 32) | lo;
    if (!((cur >> (i & 63)) & 1ull)) {            // wave-uniform
      if (lane == 0) keep[count] = i;
      ++count;
#pragma unroll
      for (int s = 0; s < kWordsPerLane; ++s) {
        const int wj = lane + 64 * s;
        if (wj < words) removed[s] |= mask[(size_t)i * words + wj];
      }
    }
  }
  if (lane == 0) *num_keep = count;
  for (int k = count + lane; k < N; k += 64) keep[k] = -1;
}

__global__ __launch_bounds__(256) void k_iou_matrix(const float* __restrict__ a, int Na,
                                                    const float* __restrict__ b, int Nb,
                                                    float* __restrict__ iou) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)Na * Nb) return;
  const int i = (int)(t / Nb), j = (int)(t - (long long)i * Nb);
  float ba[7], bb[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { ba[k] = a[(size_t)i * 7 + k]; bb[k] = b[(size_t)j * 7 + k]; }
  iou[t] = iou_bev(ba, bb);
}

constexpr int kMaxBoxes = 64 * 64 * 4;   // 16384

}  // namespace

extern "C" {

size_t ud_nms_bev_workspace_bytes(int N) {
  if (N <= 0) return 0;
  const int words = (N + 63) / 64;
  return ud_align_up((size_t)N * words * sizeof(unsigned long long));
}

int ud_nms_rotated_bev(const float* boxes, int N, float thresh, long long* keep, int* num_keep,
                       void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (N < 0 || !num_keep) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (N == 0) {
    UD_HIP_TRY(hipMemsetAsync(num_keep, 0, sizeof(int), stream));
    return UD_OK;
  }
  if (!boxes || !keep) return UD_ERR_INVALID_ARG;
  if (N > kMaxBoxes) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_nms_bev_workspace_bytes(N)) return UD_ERR_WORKSPACE;
  const int words = (N + 63) / 64;
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(workspace);
  UdProfScope prof("nms.rotated_bev", stream);
  k_nms_mask<<<ud_div_up((long long)N * words, 256), 256, 0, stream>>>(boxes, N, thresh, mask, words);
  UD_LAUNCH_CHECK();
  k_nms_reduce<<<1, 64, 0, stream>>>(mask, N, words, keep, num_keep);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

int ud_boxes_iou_bev(const float* a, int Na, const float* b, int Nb, float* iou, ud_stream_t stream_) {
  if (Na < 0 || Nb < 0) return UD_ERR_INVALID_ARG;
  if (Na == 0 || Nb == 0) return UD_OK;
  if (!a || !b || !iou) return UD_ERR_INVALID_ARG;
  k_iou_matrix<<<ud_div_up((long long)Na * Nb, 256), 256, 0, (hipStream_t)stream_>>>(a, Na, b, Nb, iou);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

}  // extern "C"
