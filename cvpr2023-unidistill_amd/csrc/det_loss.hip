// Detection loss of the CenterPoint heads as four kernels instead of ~330 tiny tensor ops.
//
// Reference: CenterHeadIouAware.get_loss (unidistill/layers/head/det3d/center_head_iou_aware.py:55-298)
// with FocalLoss / CenterNetRegLoss (unidistill/layers/losses/det3d.py:287-421) and the clamped sigmoid
// (center_head.py:153-155).  All tasks are evaluated together; head tensors are addressed through a
// pointer table (any [B, c, H, W] tensors with contiguous H*W planes, e.g. views of the packed head
// output), targets are the task-stacked tensors of the assigner.
//
//   focal : prob = clamp(sigmoid(x), 1e-4, 1 - 1e-4)  (also an output: the response distillation reads it)
//           pos[t] = sum log(p) (1-p)^g a [gt == 1],   neg[t] = sum log(1 - p + 1e-4) p^g (1-a) [gt == 0]
//   reg   : at every assigned slot (t, b, k) the 11 head values at pixel ind are gathered once;
//           box[t][j] = sum |g_j m_j - tgt_j m_j| / (num_obj[t] + 1e-4);  axis-aligned 3-D IoU loss and the
//           IoU-aware L1 against the nearest-BEV IoU of the (detached) decoded boxes.  The local
//           derivatives are produced in the same pass, so the backward only scales and scatters them.
// Reductions are two-stage and ordered (deterministic); nothing synchronises with the host.
#include "ud_common.h"
#include "ud_prof.h"

namespace {

constexpr int kMaxTasks = 8;
constexpr int kNG = 11;            // gathered values per slot: reg2 height1 dim3 rot2 vel2 iou1
constexpr int kNB = 10;            // box code size (nuScenes)
constexpr int kLoc = 17;           // saved local derivatives per slot: box 10, iou-loss 6, aware 1
constexpr int kFocalBlocks = 64;
constexpr int kRegChunks = 8;

struct FocalArgs {
  const float* hm[kMaxTasks];      // logits of task t: [B, ncls[t], HW]
  long long bstride[kMaxTasks];    // elements between consecutive batch entries
  int ncls[kMaxTasks];
  int T, B, ncm, HW;
  float alpha, gamma;
};

__device__ __forceinline__ float clamped_sigmoid(float x) {
  const float s = 1.f / (1.f + expf(-x));
  return fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
}

// grid (kFocalBlocks, T); partial[t][block][2]
__global__ __launch_bounds__(256) void k_focal_fwd(FocalArgs a, const float* __restrict__ gt,
                                                   float* __restrict__ prob, float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int t = blockIdx.y;
  const long long n = (long long)a.B * a.ncm * a.HW;
  float pos = 0.f, neg = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int p = (int)(i % a.HW);
    const int c = (int)((i / a.HW) % a.ncm);
    const int b = (int)(i / ((long long)a.HW * a.ncm));
    const size_t o = (size_t)t * n + i;
    float pr = 0.f;
    if (c < a.ncls[t]) {
      pr = clamped_sigmoid(a.hm[t][(size_t)b * a.bstride[t] + (size_t)c * a.HW + p]);
      const float g = gt[o];
      if (g == 1.f) pos += logf(pr) * powf(1.f - pr, a.gamma) * a.alpha;
      else if (g == 0.f) neg += logf(1.f - pr + 1e-4f) * powf(pr, a.gamma) * (1.f - a.alpha);
    }
    prob[o] = pr;
  }
  pos = ud_wave_sum(pos);
  neg = ud_wave_sum(neg);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = pos; red[1][wave] = neg; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const float* r = red[threadIdx.x];
    partial[((size_t)t * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
  }
}

// out[t][w] = sum over blocks (ordered)
__global__ void k_sum_partials(const float* __restrict__ partial, int nblk, int width, int total,
                               float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // i = t * width + w
  if (i >= total) return;
  const int t = i / width, w = i - t * width;
  float acc = 0.f;
  for (int b = 0; b < nblk; ++b) acc += partial[((size_t)t * nblk + b) * width + w];
  out[i] = acc;
}

// dlogit[t][b][c][p] = (g_prob + g_pos[t] dpos/dp + g_neg[t] dneg/dp) * dp/dx
__global__ __launch_bounds__(256) void k_focal_bwd(FocalArgs a, const float* __restrict__ gt,
                                                   const float* __restrict__ prob,
                                                   const float* __restrict__ g_prob,
                                                   const float* __restrict__ g_pos,
                                                   const float* __restrict__ g_neg,
                                                   float* __restrict__ dlogit) {
  const long long n = (long long)a.B * a.ncm * a.HW;
  const long long total = n * a.T;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
    const int t = (int)(o / n);
    const int c = (int)((o / a.HW) % a.ncm);
    float d = 0.f;
    if (c < a.ncls[t]) {
      const float p = prob[o], g = gt[o];
      float dp = g_prob ? g_prob[o] : 0.f;
      if (g == 1.f) {
        const float q = 1.f - p;
        dp += g_pos[t] * a.alpha * (powf(q, a.gamma) / p - a.gamma * powf(q, a.gamma - 1.f) * logf(p));
      } else if (g == 0.f) {
        const float u = 1.f - p + 1e-4f;
        dp += g_neg[t] * (1.f - a.alpha) * (a.gamma * powf(p, a.gamma - 1.f) * logf(u) - powf(p, a.gamma) / u);
      }
      // clamp passes the gradient only strictly inside (the clamped value equals a bound otherwise)
      if (p > 1e-4f && p < 1.f - 1e-4f) d = dp * p * (1.f - p);
    }
    dlogit[o] = d;
  }
}

// ---- gathered regression / IoU terms --------------------------------------------------------------
struct RegArgs {
  const float* head[kMaxTasks][kNG];   // pointer to the channel plane of gathered value j of task t
  long long bstride[kMaxTasks][kNG];   // elements between consecutive batch entries of that tensor
  int T, B, K, HW, nb;                 // nb = box code size (10 or 8)
  float sx, sy;                        // stride * voxel size (x, y)
};

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }

__device__ __forceinline__ void decode_whl(float e, float* w, float* dw) {
  const float ec = fminf(e, 80.f);
  const float ex = expf(ec);
  *w = fminf(fmaxf(ex, 0.001f), 30.f);
  *dw = (e <= 80.f && ex >= 0.001f && ex <= 30.f) ? ex : 0.f;
}

// overlap = clamp(min(pc+pe/2, tc+te/2) - max(pc-pe/2, tc-te/2), min=1e-3) and d/dpc, d/dpe
__device__ __forceinline__ float overlap(float pc, float pe, float tc, float te, float* dpc, float* dpe) {
  const float A = pc + 0.5f * pe, B = tc + 0.5f * te, C = pc - 0.5f * pe, D = tc - 0.5f * te;
  const float raw = fminf(A, B) - fmaxf(C, D);
  const float on = raw >= 1e-3f ? 1.f : 0.f;
  const float dhi_c = A <= B ? 1.f : 0.f, dlo_c = C >= D ? 1.f : 0.f;
  *dpc = on * (dhi_c - dlo_c);
  *dpe = on * (0.5f * dhi_c + 0.5f * dlo_c);
  return fmaxf(raw, 1e-3f);
}

__device__ __forceinline__ void aligned_bev(float x, float y, float w, float l, float rot, float* x0,
                                            float* y0, float* x1, float* y1) {
  const float pi = 3.14159265358979323846f;
  const float r = fabsf(rot - floorf(rot / pi + 0.5f) * pi);
  const float dx = r < pi / 4 ? w : l, dy = r < pi / 4 ? l : w;
  *x0 = x - dx / 2; *y0 = y - dy / 2; *x1 = x + dx / 2; *y1 = y + dy / 2;
}

// grid (kRegChunks, T); partial[t][chunk][nb + 2]; loc[t][b][k][kLoc]
__global__ __launch_bounds__(256) void k_reg_fwd(RegArgs a, const long long* __restrict__ ind,
                                                 const unsigned char* __restrict__ mask,
                                                 const float* __restrict__ tgt, int tgt_dim,
                                                 const float* __restrict__ num_obj,
                                                 float* __restrict__ partial, float* __restrict__ loc) {
  __shared__ float red[4][kNB + 2];
  const int t = blockIdx.y;
  const int slots = a.B * a.K;
  const float den_box = num_obj[t] + 1e-4f, den_iou = fmaxf(num_obj[t], 1.f);
  float acc[kNB + 2];
#pragma unroll
  for (int j = 0; j < kNB + 2; ++j) acc[j] = 0.f;
  for (int s = blockIdx.x * 256 + threadIdx.x; s < slots; s += gridDim.x * 256) {
    const size_t so = (size_t)t * slots + s;
    float* L = loc + so * kLoc;
    const bool m = mask[so] != 0;
    if (!m) {
#pragma unroll
      for (int j = 0; j < kLoc; ++j) L[j] = 0.f;
      continue;
    }
    const int b = s / a.K;
    const long long pix = ind[so];
    float g[kNG], tg[kNB];
#pragma unroll
    for (int j = 0; j < kNG; ++j)
      g[j] = (j < a.nb || j == kNG - 1) ? a.head[t][j][(size_t)b * a.bstride[t][j] + pix] : 0.f;
#pragma unroll
    for (int j = 0; j < kNB; ++j) tg[j] = j < a.nb ? tgt[so * tgt_dim + j] : 0.f;
    // masked L1 per code dimension
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
      const float mj = (j < a.nb && !isnan(tg[j])) ? 1.f : 0.f;
      const float d = g[j] * mj - tg[j] * mj;      // a NaN target propagates exactly like the reference's target * mask
      acc[j] += fabsf(d);
      L[j] = sgn(d) * mj / den_box;
    }
    // decoded boxes
    float pw[3], dpw[3], tw[3], dtw;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      decode_whl(g[3 + q], &pw[q], &dpw[q]);
      decode_whl(tg[3 + q], &tw[q], &dtw);
    }
    const float px = g[0] * a.sx, py = g[1] * a.sy, pz = g[2];
    const float tx = tg[0] * a.sx, ty = tg[1] * a.sy, tz = tg[2];
    float dxc, dxe, dyc, dye, dzc, dze;
    const float ox = overlap(px, pw[0], tx, tw[0], &dxc, &dxe);
    const float oy = overlap(py, pw[2], ty, tw[2], &dyc, &dye);
    const float oz = overlap(pz, pw[1], tz, tw[1], &dzc, &dze);
    const float inter = ox * oy * oz;
    const float vraw = pw[0] * pw[2] * pw[1];
    const float vp = fmaxf(vraw, 1e-3f), von = vraw >= 1e-3f ? 1.f : 0.f;
    const float vt = fmaxf(tw[0] * tw[2] * tw[1], 1e-3f);
    const float U = vp + vt - inter;
    const float iou = inter / U;
    acc[kNB] += 1.f - fminf(fmaxf(iou, 0.f), 1.f);
    const float dL = (iou >= 0.f && iou <= 1.f) ? -1.f / den_iou : 0.f;          // d loss / d iou
    // d iou / d(inter), d iou / d(vp)
    const float di_dinter = (U + inter) / (U * U), di_dvp = -inter / (U * U);
    const float dinter_x = oy * oz, dinter_y = ox * oz, dinter_z = ox * oy;
    // centre terms (e0, e1, e2) and size terms (e3 -> w0 (x), e4 -> w1 (z), e5 -> w2 (y))
    L[kNB + 0] = dL * di_dinter * dinter_x * dxc * a.sx;
    L[kNB + 1] = dL * di_dinter * dinter_y * dyc * a.sy;
    L[kNB + 2] = dL * di_dinter * dinter_z * dzc;
    L[kNB + 3] = dL * (di_dinter * dinter_x * dxe + di_dvp * von * pw[2] * pw[1]) * dpw[0];
    L[kNB + 4] = dL * (di_dinter * dinter_z * dze + di_dvp * von * pw[0] * pw[2]) * dpw[1];
    L[kNB + 5] = dL * (di_dinter * dinter_y * dye + di_dvp * von * pw[0] * pw[1]) * dpw[2];
    // IoU-aware target: nearest-BEV IoU of the decoded target and (detached) prediction
    float a0x, a0y, a1x, a1y, b0x, b0y, b1x, b1y;
    aligned_bev(tx, ty, tw[0], tw[1], atan2f(tg[6], tg[7]), &a0x, &a0y, &a1x, &a1y);
    aligned_bev(px, py, pw[0], pw[1], atan2f(g[6], g[7]), &b0x, &b0y, &b1x, &b1y);
    const float iw = fmaxf(fminf(a1x, b1x) - fmaxf(a0x, b0x), 0.f);
    const float ih = fmaxf(fminf(a1y, b1y) - fmaxf(a0y, b0y), 0.f);
    const float ib = iw * ih;
    const float area = (a1x - a0x) * (a1y - a0y) + (b1x - b0x) * (b1y - b0y) - ib;
    const float tar = 2.f * (ib / fmaxf(area, 1e-6f) - 0.5f);
    const float ma = isnan(tar) ? 0.f : 1.f;
    const float da = g[kNG - 1] * ma - tar * ma;
    acc[kNB + 1] += fabsf(da);
    L[kNB + 6] = sgn(da) * ma / den_box;
  }
#pragma unroll
  for (int j = 0; j < kNB + 2; ++j) acc[j] = ud_wave_sum(acc[j]);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int j = 0; j < kNB + 2; ++j) red[wave][j] = acc[j];
  __syncthreads();
  if (threadIdx.x < kNB + 2) {
    const int j = threadIdx.x;
    float v = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    v /= (j == kNB) ? den_iou : den_box;
    partial[((size_t)t * gridDim.x + blockIdx.x) * (kNB + 2) + j] = v;
  }
}

// dhead[t][b][j][pix] = g_box[t][j] * L_box[j] + g_iou[t] * L_iou[j] (+ g_aw[t] * L_aw for the iou head)
__global__ __launch_bounds__(256) void k_reg_bwd(RegArgs a, const long long* __restrict__ ind,
                                                 const unsigned char* __restrict__ mask,
                                                 const float* __restrict__ loc,
                                                 const float* __restrict__ g_box,
                                                 const float* __restrict__ g_iou,
                                                 const float* __restrict__ g_aw, float* __restrict__ dhead) {
  const int slots = a.B * a.K;
  const long long total = (long long)a.T * slots;
  for (long long so = (long long)blockIdx.x * 256 + threadIdx.x; so < total; so += (long long)gridDim.x * 256) {
    if (!mask[so]) continue;
    const int t = (int)(so / slots), s = (int)(so - (long long)t * slots), b = s / a.K;
    const float* L = loc + so * kLoc;
    const long long pix = ind[so];
    float* dst = dhead + (((size_t)t * a.B + b) * kNG) * a.HW + pix;
#pragma unroll
    for (int j = 0; j < kNG; ++j) {
      float v = 0.f;
      if (j < a.nb) v += g_box[t * kNB + j] * L[j];
      if (j < 6) v += g_iou[t] * L[kNB + j];
      if (j == kNG - 1) v = g_aw[t] * L[kNB + 6];
      if (j < a.nb || j == kNG - 1) dst[(size_t)j * a.HW] = v;
    }
  }
}

}  // namespace

extern "C" {

size_t ud_det_loss_workspace_bytes(int T) {
  if (T <= 0 || T > kMaxTasks) return 0;
  return ud_align_up((size_t)T * kFocalBlocks * 2 * sizeof(float)) +
         ud_align_up((size_t)T * kRegChunks * (kNB + 2) * sizeof(float));
}

// hm[t]: device pointer to task t's logits [B, ncls[t], HW] with hm_bstride[t] elements between batch
// entries.  gt, prob: [T, B, ncm, HW] fp32.  pos_neg: [T, 2] = (pos[t], neg[t]).
int ud_det_focal_fwd(const float* const* hm, const long long* hm_bstride, const int* ncls, int T, int B,
                     int ncm, int HW, const float* gt, float alpha, float gamma, float* prob,
                     float* pos_neg, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!hm || !hm_bstride || !ncls || !gt || !prob || !pos_neg || T <= 0 || T > kMaxTasks || B <= 0 ||
      ncm <= 0 || HW <= 0)
    return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_det_loss_workspace_bytes(T)) return UD_ERR_WORKSPACE;
  FocalArgs a;
  for (int t = 0; t < T; ++t) {
    if (!hm[t] || ncls[t] <= 0 || ncls[t] > ncm) return UD_ERR_INVALID_ARG;
    a.hm[t] = hm[t]; a.bstride[t] = hm_bstride[t]; a.ncls[t] = ncls[t];
  }
  a.T = T; a.B = B; a.ncm = ncm; a.HW = HW; a.alpha = alpha; a.gamma = gamma;
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = reinterpret_cast<float*>(workspace);
  UdProfScope prof("det_loss.focal_fwd", stream);
  k_focal_fwd<<<dim3(kFocalBlocks, T), 256, 0, stream>>>(a, gt, prob, partial);
  UD_LAUNCH_CHECK();
  k_sum_partials<<<1, 64, 0, stream>>>(partial, kFocalBlocks, 2, T * 2, pos_neg);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// dlogit: [T, B, ncm, HW]; g_prob may be NULL; g_pos / g_neg: [T] device.
int ud_det_focal_bwd(const int* ncls, int T, int B, int ncm, int HW, const float* gt, const float* prob,
                     const float* g_prob, const float* g_pos, const float* g_neg, float alpha,
                     float gamma, float* dlogit, ud_stream_t stream_) {
  if (!ncls || !gt || !prob || !g_pos || !g_neg || !dlogit || T <= 0 || T > kMaxTasks || B <= 0 ||
      ncm <= 0 || HW <= 0)
    return UD_ERR_INVALID_ARG;
  FocalArgs a;
  for (int t = 0; t < T; ++t) { a.hm[t] = nullptr; a.bstride[t] = 0; a.ncls[t] = ncls[t]; }
  a.T = T; a.B = B; a.ncm = ncm; a.HW = HW; a.alpha = alpha; a.gamma = gamma;
  const long long total = (long long)T * B * ncm * HW;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  k_focal_bwd<<<(int)blocks, 256, 0, (hipStream_t)stream_>>>(a, gt, prob, g_prob, g_pos, g_neg, dlogit);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// head[t*11 + j]: pointer to the channel plane (batch 0) of gathered value j of task t, j in the order
// reg.x reg.y height dim0 dim1 dim2 rot.sin rot.cos vel.x vel.y iou; head_bstride likewise.
// ind i64 / mask u8 / tgt f32 [T, B, K(, tgt_dim)]; num_obj f32[T] (device).
// losses: [T, nb + 2] = box[0..nb), iou_loss, iou_aware (already normalised); loc: [T, B, K, 17].
int ud_det_reg_fwd(const float* const* head, const long long* head_bstride, int T, int B, int K, int HW,
                   int nb, const long long* ind, const unsigned char* mask, const float* tgt, int tgt_dim,
                   const float* num_obj, float sx, float sy, float* losses, float* loc, void* workspace,
                   size_t workspace_bytes, ud_stream_t stream_) {
  if (!head || !head_bstride || !ind || !mask || !tgt || !num_obj || !losses || !loc || T <= 0 ||
      T > kMaxTasks || B <= 0 || K <= 0 || HW <= 0 || (nb != 10 && nb != 8) || tgt_dim < nb)
    return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_det_loss_workspace_bytes(T)) return UD_ERR_WORKSPACE;
  RegArgs a;
  for (int t = 0; t < T; ++t)
    for (int j = 0; j < kNG; ++j) {
      a.head[t][j] = head[t * kNG + j];
      a.bstride[t][j] = head_bstride[t * kNG + j];
      if ((j < nb || j == kNG - 1) && !a.head[t][j]) return UD_ERR_INVALID_ARG;
    }
  a.T = T; a.B = B; a.K = K; a.HW = HW; a.nb = nb; a.sx = sx; a.sy = sy;
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) +
                                            ud_align_up((size_t)T * kFocalBlocks * 2 * sizeof(float)));
  UdProfScope prof("det_loss.reg_fwd", stream);
  k_reg_fwd<<<dim3(kRegChunks, T), 256, 0, stream>>>(a, ind, mask, tgt, tgt_dim, num_obj, partial, loc);
  UD_LAUNCH_CHECK();
  // partial rows are kNB + 2 wide; the caller's losses rows are nb + 2 wide
  k_sum_partials<<<1, 128, 0, stream>>>(partial, kRegChunks, kNB + 2, T * (kNB + 2), losses);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// dhead: [T, B, 11, HW] fp32, zero-initialised by the caller; g_box [T, 10], g_iou [T], g_aw [T] device.
int ud_det_reg_bwd(int T, int B, int K, int HW, int nb, const long long* ind, const unsigned char* mask,
                   const float* loc, const float* g_box, const float* g_iou, const float* g_aw,
                   float* dhead, ud_stream_t stream_) {
  if (!ind || !mask || !loc || !g_box || !g_iou || !g_aw || !dhead || T <= 0 || T > kMaxTasks || B <= 0 ||
      K <= 0 || HW <= 0 || (nb != 10 && nb != 8))
    return UD_ERR_INVALID_ARG;
  RegArgs a;
  a.T = T; a.B = B; a.K = K; a.HW = HW; a.nb = nb; a.sx = a.sy = 0.f;
  const long long total = (long long)T * B * K;
  k_reg_bwd<<<ud_div_up(total, 256), 256, 0, (hipStream_t)stream_>>>(a, ind, mask, loc, g_box, g_iou, g_aw, dhead);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

}  // extern "C"
