// Proposal layer of the CenterPoint heads on the device: heat-map top-K, box decode, IoU-aware score,
// range / score filter, rotated NMS and the padded roi tensors -- all tasks and all samples of a batch in
// THREE launches, no host round trip.
//
// Reference: IouAwareGenProposals.proposal_layer (unidistill/layers/head/det3d/generate_proposals/
// iou_aware_gen_proposals.py:43-139), CenterPointGenProposals._topk / _nms_gpu_3d /
// generate_predicted_boxes (centerpoint_gen_proposals.py:66-105, 232-340).  The reference runs, per task,
// two torch.topk, seven gathers, a boolean-mask selection per sample (a host sync each), a sort and the
// `iou3d_nms_cuda.nms_gpu` call with a CPU keep tensor, then concatenates per sample in python.
//
//   k_prop_decode : one 1024-thread workgroup per (sample, task).
//       top-K of the nc*H*W sigmoid scores == the reference's per-class top-K followed by the top-K of
//       their union.  Keys are 64 bit, (score bits << 32) | ~flat index, hence unique: an LDS radix select
//       (8-bit digits from the top, stops as soon as the K-th key is isolated) finds the K-th key, the keys
//       above it are compacted and bitonic-sorted in LDS: descending score, ascending (class, pixel) on equal
//       scores.  Each thread then decodes its ranks in registers (same fp32 operations in the same order as
//       the reference's tensor code), applies the centre-range and score masks, and the survivors are
//       sorted by the NMS score (descending, rank order on ties) into box rows [x y z dx dy dz rot (vx vy)
//       score label] + a device count.
//   k_prop_mask   : 64 suppression bits per (box i, word of later boxes) of every (sample, task), bounded by
//       the device counts (csrc/bev_iou.h: the IoU of nms.hip).
//   k_prop_select : one workgroup per sample, wave t walks task t's boxes in NMS-score order (the inherently
//       sequential part) up to post_max keepers; the tasks' keepers are then packed back to back into
//       rois / roi_scores / roi_labels (zero padded, labels = 1 + class offset of the task + class), and the
//       per-sample total is written: the ONE number the host reads to slice pred_dicts.
#include "ud_common.h"
#include "ud_prof.h"
#include "bev_iou.h"

namespace {

constexpr int kMaxTasks = 8;
constexpr int kMaxK = 2048;          // top-K candidates per (sample, task): LDS sort width
constexpr int kMaxPost = 512;        // kept boxes per task
constexpr int kRow = 11;             // floats per sorted candidate row: 9 box values, score, label
constexpr int kThreads = 1024;

struct Desc {                        // one head tensor [B, c, H*W] through its strides (elements)
  const float* p;
  long long sb, sc, sp;
};
struct Task {
  Desc d[7];                         // hm, reg, height, dim, rot, vel, iou
  int nc, cls_off;
  float alpha;
};
struct Params {
  Task t[kMaxTasks];
  int B, T, H, W, K, nbox, no_log, iou_aware, post_max;
  float osf, vs0, vs1, pc0, pc1, score_thr, nms_thr;
  float lo[3], hi[3];
};

__device__ __forceinline__ float ld(const Desc& d, int b, int c, int pix) {
  return d.p[(long long)b * d.sb + (long long)c * d.sc + (long long)pix * d.sp];
}

__device__ __forceinline__ unsigned long long score_key(const Task& tk, int b, int HW, int i) {
  const int c = i / HW, pix = i - c * HW;
  const float s = 1.0f / (1.0f + expf(-ld(tk.d[0], b, c, pix)));       // torch.sigmoid
  return ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(~(unsigned)i);
}

// descending bitonic sort of n (power of two) 64-bit keys in LDS by the whole workgroup
__device__ __forceinline__ void bitonic_desc(unsigned long long* keys, int n) {
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += kThreads) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = keys[i], b = keys[p];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[p] = a; }
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(kThreads) void k_prop_decode(Params P, float* __restrict__ rows,
                                                           int* __restrict__ counts) {
  __shared__ unsigned long long keys[kMaxK];
  __shared__ int pos_of_rank[kMaxK];
  __shared__ int hist[256];
  __shared__ int s_cnt, s_digit, s_k, s_valid;
  const int bt = blockIdx.x, b = bt / P.T, t = bt - b * P.T, tid = threadIdx.x;
  const Task& tk = P.t[t];
  const int HW = P.H * P.W, n = tk.nc * HW;
  const int K = min(P.K, n);
  int n2 = 1;
  while (n2 < K) n2 <<= 1;

  // ---- radix select of the K-th largest key ----------------------------------------------------
  unsigned long long prefix = 0ull;        // digits fixed so far (high bits)
  int shift = 64, krem = K;                // K-th largest among the keys that match `prefix`
  bool isolated = false;
  while (shift > 0 && !isolated) {
    shift -= 8;
    for (int i = tid; i < 256; i += kThreads) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += kThreads) {
      const unsigned long long key = score_key(tk, b, HW, i);
      if (shift == 56 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
        atomicAdd(&hist[(int)((key >> shift) & 255ull)], 1);
    }
    __syncthreads();
    if (tid < 64) {                        // one wave: 4 bins per lane, suffix sums from the top digit
      const int base = 252 - 4 * tid;      // lane 0 owns the top bins 252..255
      const int h3 = hist[base + 3], h2 = hist[base + 2], h1 = hist[base + 1], h0 = hist[base];
      const int mine = h3 + h2 + h1 + h0;
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (tid >= o) incl += v;
      }
      const int above = incl - mine;       // keys with a larger digit than this lane's bins
      if (above < krem && krem <= incl) {  // exactly one lane
        int acc = above, dsel = base + 3, cnt = h3;
        if (acc + h3 >= krem) { dsel = base + 3; cnt = h3; }
        else if ((acc += h3) + h2 >= krem) { dsel = base + 2; cnt = h2; }
        else if ((acc += h2) + h1 >= krem) { dsel = base + 1; cnt = h1; }
        else { acc += h1; dsel = base; cnt = h0; }
        s_digit = dsel;
        s_k = krem - acc;                  // rank inside the selected bin
        s_cnt = cnt;
      }
    }
    __syncthreads();
    prefix |= (unsigned long long)s_digit << shift;
    krem = s_k;
    isolated = (s_cnt == 1);
    __syncthreads();
  }
  // every key whose top bits are >= prefix's is among the K largest (exactly K of them)
  if (tid == 0) { s_cnt = 0; s_valid = 0; }
  for (int i = tid; i < n2; i += kThreads) keys[i] = 0ull;
  __syncthreads();
  for (int i = tid; i < n; i += kThreads) {
    const unsigned long long key = score_key(tk, b, HW, i);
    if ((key >> shift) >= (prefix >> shift)) {
      const int slot = atomicAdd(&s_cnt, 1);
      if (slot < n2) keys[slot] = key;
    }
  }
  __syncthreads();
  bitonic_desc(keys, n2);

  // ---- decode my ranks in registers -------------------------------------------------------------
  constexpr int kPer = kMaxK / kThreads;
  float row[kPer][kRow];
  unsigned long long key2[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int r = tid + q * kThreads;
    key2[q] = 0ull;
    if (r < K) {
      const unsigned long long key = keys[r];
      const float score = __uint_as_float((unsigned)(key >> 32));
      const int i = (int)(~(unsigned)key);
      const int c = i / HW, pix = i - c * HW;
      const float ys = (float)(pix / P.W), xs = (float)(pix % P.W);
      float x = xs + ld(tk.d[1], b, 0, pix);
      float y = ys + ld(tk.d[1], b, 1, pix);
      const float rot = atan2f(ld(tk.d[4], b, 0, pix), ld(tk.d[4], b, 1, pix));
      const float z = ld(tk.d[2], b, 0, pix);
      x = x * P.osf * P.vs0 + P.pc0;
      y = y * P.osf * P.vs1 + P.pc1;
      row[q][0] = x; row[q][1] = y; row[q][2] = z;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float dv = ld(tk.d[3], b, k, pix);
        if (!P.no_log) dv = fminf(fmaxf(expf(dv), 0.001f), 30.f);
        row[q][3 + k] = dv;
      }
      row[q][6] = rot;
      row[q][7] = row[q][8] = 0.f;
      if (P.nbox == 9) { row[q][7] = ld(tk.d[5], b, 0, pix); row[q][8] = ld(tk.d[5], b, 1, pix); }
      row[q][9] = score;
      row[q][10] = (float)c;
      float nms = score;
      if (P.iou_aware) {
        const float iou = fminf(fmaxf(ld(tk.d[6], b, 0, pix) / 2.f + 0.5f, 0.f), 1.f);
        nms = powf(score, 1.f - tk.alpha) * powf(iou, tk.alpha);
      }
      const bool ok = x >= P.lo[0] && y >= P.lo[1] && z >= P.lo[2] && x <= P.hi[0] && y <= P.hi[1] &&
                      z <= P.hi[2] && score > P.score_thr;
      if (ok) {
        // +1 keeps a survivor's key above the 0 of a masked rank even at nms == 0
        key2[q] = (((unsigned long long)__float_as_uint(fmaxf(nms, 0.f)) + 1ull) << 32) | (unsigned)(~(unsigned)r);
        atomicAdd(&s_valid, 1);
      }
    }
  }
  __syncthreads();                      // all reads of keys[] done
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int r = tid + q * kThreads;
    if (r < n2) keys[r] = key2[q];
  }
  __syncthreads();
  bitonic_desc(keys, n2);
  for (int j = tid; j < n2; j += kThreads) {
    const unsigned long long k2 = keys[j];
    if (k2 != 0ull) pos_of_rank[(int)(~(unsigned)k2)] = j;
  }
  __syncthreads();
  float* out = rows + (size_t)bt * P.K * kRow;
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int r = tid + q * kThreads;
    if (r < K && key2[q] != 0ull) {
      float* o = out + (size_t)pos_of_rank[r] * kRow;
#pragma unroll
      for (int k = 0; k < kRow; ++k) o[k] = row[q][k];
    }
  }
  if (tid == 0) counts[bt] = s_valid;
}

__global__ __launch_bounds__(256) void k_prop_mask(const float* __restrict__ rows, const int* __restrict__ counts,
                                                   int K, int words, float thresh,
                                                   unsigned long long* __restrict__ mask) {
  const int bt = blockIdx.y;
  const int N = counts[bt];
  const long long tt = (long long)blockIdx.x * 256 + threadIdx.x;
  if (tt >= (long long)N * words) return;
  const int i = (int)(tt / words), wj = (int)(tt - (long long)i * words);
  const float* boxes = rows + (size_t)bt * K * kRow;
  unsigned long long bits = 0ull;
  const int j0 = wj * 64;
  if (j0 + 63 > i && j0 < N) {
    float bi[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) bi[k] = boxes[(size_t)i * kRow + k];
    for (int jj = 0; jj < 64; ++jj) {
      const int j = j0 + jj;
      if (j <= i || j >= N) continue;
      float bj[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) bj[k] = boxes[(size_t)j * kRow + k];
      if (ud_iou::iou_bev(bi, bj) > thresh) bits |= 1ull << jj;
    }
  }
  mask[((size_t)bt * K + i) * words + wj] = bits;
}

__global__ __launch_bounds__(64 * kMaxTasks) void k_prop_select(
    Params P, const float* __restrict__ rows, const int* __restrict__ counts,
    const unsigned long long* __restrict__ mask, int words, float* __restrict__ rois,
    float* __restrict__ roi_scores, long long* __restrict__ roi_labels, int* __restrict__ totals) {
  __shared__ short keep[kMaxTasks][kMaxPost];
  __shared__ int kept[kMaxTasks];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave < P.T) {
    const int bt = b * P.T + wave, N = counts[bt];
    const unsigned long long* m = mask + (size_t)bt * P.K * words;
    unsigned long long removed = 0ull;             // lane w owns word w (words <= 32)
    int count = 0;
    for (int i = 0; i < N && count < P.post_max; ++i) {
      const int w = i >> 6;
      const unsigned lo = __shfl((unsigned)(removed & 0xFFFFFFFFull), w);
      const unsigned hi = __shfl((unsigned)(removed >> 32), w);
      const unsigned long long cur = ((unsigned long long)hi << 32) | lo;
      if (!((cur >> (i & 63)) & 1ull)) {           // wave-uniform
        if (lane == 0) keep[wave][count] = (short)i;
        ++count;
        if (lane < words) removed |= m[(size_t)i * words + lane];
      }
    }
    if (lane == 0) kept[wave] = count;
  }
  __syncthreads();
  const int num_rois = P.post_max * P.T;
  int total = 0;
  for (int t = 0; t < P.T; ++t) total += kept[t];
  for (int r = tid; r < num_rois; r += blockDim.x) {
    float* o = rois + ((size_t)b * num_rois + r) * P.nbox;
    float sc = 0.f;
    long long lb = 0;
    if (r < total) {
      int t = 0, off = 0;
      while (r >= off + kept[t]) { off += kept[t]; ++t; }
      const float* src = rows + ((size_t)(b * P.T + t) * P.K + keep[t][r - off]) * kRow;
      for (int k = 0; k < P.nbox; ++k) o[k] = src[k];
      sc = src[9];
      lb = (long long)src[10] + 1 + P.t[t].cls_off;
    } else {
      for (int k = 0; k < P.nbox; ++k) o[k] = 0.f;
    }
    roi_scores[(size_t)b * num_rois + r] = sc;
    roi_labels[(size_t)b * num_rois + r] = lb;
  }
  if (tid == 0) totals[b] = total;
}

}  // namespace

extern "C" {

size_t ud_proposal_workspace_bytes(int B, int T, int K) {
  if (B <= 0 || T <= 0 || T > kMaxTasks || K <= 0 || K > kMaxK) return 0;
  const int words = (K + 63) / 64;
  return ud_align_up((size_t)B * T * K * kRow * sizeof(float)) + ud_align_up((size_t)B * T * sizeof(int)) +
         ud_align_up((size_t)B * T * K * words * sizeof(unsigned long long));
}

int ud_proposal_layer(const float* const* heads, const long long* strides, const int* num_classes,
                      const int* class_offsets, const float* iou_alpha, int B, int T, int H, int W, int K,
                      int post_max, int box_dim, int no_log, float out_size_factor, float voxel_x,
                      float voxel_y, float pc_x, float pc_y, const float* center_range, float score_threshold,
                      float nms_threshold, float* rois, float* roi_scores, long long* roi_labels,
                      int* num_boxes, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (B <= 0 || T <= 0 || T > kMaxTasks || H <= 0 || W <= 0 || K <= 0 || post_max <= 0) return UD_ERR_INVALID_ARG;
  if (!heads || !strides || !num_classes || !class_offsets || !center_range || !rois || !roi_scores ||
      !roi_labels || !num_boxes)
    return UD_ERR_INVALID_ARG;
  if (box_dim != 7 && box_dim != 9) return UD_ERR_INVALID_ARG;
  if (K > kMaxK || post_max > kMaxPost) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_proposal_workspace_bytes(B, T, K)) return UD_ERR_WORKSPACE;
  Params P;
  for (int t = 0; t < T; ++t) {
    for (int j = 0; j < 7; ++j) {
      Desc& d = P.t[t].d[j];
      d.p = heads[t * 7 + j];
      d.sb = strides[(t * 7 + j) * 3];
      d.sc = strides[(t * 7 + j) * 3 + 1];
      d.sp = strides[(t * 7 + j) * 3 + 2];
      const bool optional = (j == 5 && box_dim == 7) || (j == 6 && !iou_alpha);
      if (!d.p && !optional) return UD_ERR_INVALID_ARG;
    }
    P.t[t].nc = num_classes[t];
    P.t[t].cls_off = class_offsets[t];
    P.t[t].alpha = iou_alpha ? iou_alpha[t] : 0.f;
    if (P.t[t].nc <= 0 || (long long)P.t[t].nc * H * W > 0x7FFFFFFFll) return UD_ERR_INVALID_ARG;
  }
  P.B = B; P.T = T; P.H = H; P.W = W; P.K = K; P.nbox = box_dim; P.no_log = no_log;
  P.iou_aware = iou_alpha != nullptr; P.post_max = post_max;
  P.osf = out_size_factor; P.vs0 = voxel_x; P.vs1 = voxel_y; P.pc0 = pc_x; P.pc1 = pc_y;
  P.score_thr = score_threshold; P.nms_thr = nms_threshold;
  for (int k = 0; k < 3; ++k) { P.lo[k] = center_range[k]; P.hi[k] = center_range[3 + k]; }
  hipStream_t stream = (hipStream_t)stream_;
  UdArena ar(workspace, workspace_bytes);
  const int words = (K + 63) / 64;
  float* rows = ar.take<float>((size_t)B * T * K * kRow);
  int* counts = ar.take<int>((size_t)B * T);
  unsigned long long* mask = ar.take<unsigned long long>((size_t)B * T * K * words);
  if (!ar.ok()) return UD_ERR_WORKSPACE;
  UdProfScope prof("proposals.layer", stream);
  k_prop_decode<<<B * T, kThreads, 0, stream>>>(P, rows, counts);
  UD_LAUNCH_CHECK();
  k_prop_mask<<<dim3(ud_div_up((long long)K * words, 256), B * T), 256, 0, stream>>>(rows, counts, K, words,
                                                                                    nms_threshold, mask);
  UD_LAUNCH_CHECK();
  k_prop_select<<<B, 64 * kMaxTasks, 0, stream>>>(P, rows, counts, mask, words, rois, roi_scores, roi_labels,
                                                   num_boxes);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

}  // extern "C"
