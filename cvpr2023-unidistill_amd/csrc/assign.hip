// FCOS-style target assignment of the CenterPoint heads as one kernel.
//
// Reference: FCOSAssigner.assign_targets (unidistill/layers/head/det3d/target_assigner/
// fcos_assigner.py:73-285): per task, the `topk` anchors nearest to every ground-truth centre are
// positives; each positive anchor is assigned its nearest box (task order, first on ties); positives are
// compacted in ascending anchor order into K slots with the box encoding relative to the anchor, and
// a one-hot class heat map is written.  The tensor-op formulation (layers/center_head.py) issues ~180
// launches per step; here one workgroup owns one (task, sample) pair:
//   boxes -> LDS, task members ranked by (class offset, index); per member one wave picks the k
//   smallest of the 81 window candidates (the k <= 9 nearest anchors of a point on the regular anchor
//   grid lie within +-4 cells of the clamped nearest cell); positives are bits of an LDS bitmap whose
//   prefix popcount gives the ascending slot; nearest-box search, encoding and heat map follow.
// Integer outputs are bit-identical to the tensor-op path; float encodings use the same operations.
#include "ud_common.h"
#include "ud_prof.h"

namespace {

constexpr int kMaxBoxes = 512;
constexpr int kMaxBoxDim = 12;      // columns of a ground-truth row incl. the class column
constexpr int kMaxClasses = 64;

struct AssignArgs {
  int T, B, M, cols;                // cols = box columns incl. class
  int w, h, K, topk, ncm, enc_dim;
  float osf, pc0, pc1, vs0, vs1;
  signed char task_of[kMaxClasses]; // by 1-based class id (0 = none)
  signed char off_of[kMaxClasses];
};

__device__ __forceinline__ float limit_period_2pi(float v) {
  const float period = 6.283185307179586f;
  return v - floorf(v / period + 0.5f) * period;
}

__global__ __launch_bounds__(256) void k_assign_targets(AssignArgs a, const float* __restrict__ gt,
                                                        float* __restrict__ hm, long long* __restrict__ ind,
                                                        unsigned char* __restrict__ mask,
                                                        long long* __restrict__ cat, float* __restrict__ enc) {
  __shared__ float box[kMaxBoxes * kMaxBoxDim];
  __shared__ short order[kMaxBoxes];          // member indices in task order
  __shared__ short coff[kMaxBoxes];           // class offset of box m in this task, -1 = not a member
  __shared__ float mcx[kMaxBoxes], mcy[kMaxBoxes];   // member centres (task order), anchor-grid units * osf
  __shared__ unsigned long long bits[(180 * 180 + 63) / 64 + 1];
  __shared__ int prefix[(180 * 180 + 63) / 64 + 1];
  __shared__ int s_last, s_nm;
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = a.w * a.h, words = (A + 63) / 64;
  const float* g = gt + (size_t)b * a.M * a.cols;
  if (tid == 0) { s_last = 0; s_nm = 0; }
  for (int i = tid; i < a.M * a.cols; i += 256) box[i] = g[i];
  for (int i = tid; i < words; i += 256) bits[i] = 0ull;
  __syncthreads();
  // rows up to the last one whose box columns do not sum to zero are valid (row 0 always)
  for (int m = tid; m < a.M; m += 256) {
    float s = 0.f;
    for (int c = 0; c < a.cols - 1; ++c) s += box[m * a.cols + c];
    if (s != 0.f) atomicMax(&s_last, m);
  }
  __syncthreads();
  const int last = s_last;
  for (int m = tid; m < a.M; m += 256) {
    long long cls = (long long)box[m * a.cols + a.cols - 1];
    cls = cls < 0 ? 0 : (cls > kMaxClasses - 1 ? kMaxClasses - 1 : cls);
    const bool member = m <= last && a.task_of[cls] == t;
    coff[m] = member ? a.off_of[cls] : -1;
  }
  __syncthreads();
  // task order = ascending (class offset, original index)
  for (int m = tid; m < a.M; m += 256) {
    if (coff[m] < 0) continue;
    int r = 0;
    for (int q = 0; q < a.M; ++q)
      if (coff[q] >= 0 && (coff[q] < coff[m] || (coff[q] == coff[m] && q < m))) ++r;
    order[r] = (short)m;
    mcx[r] = (box[m * a.cols + 0] - a.pc0) / a.vs0;
    mcy[r] = (box[m * a.cols + 1] - a.pc1) / a.vs1;
    atomicAdd(&s_nm, 1);
  }
  __syncthreads();
  const int nm = s_nm;
  // the topk nearest anchors of every member: one wave per member, 81 window candidates
  for (int r = wave; r < nm; r += 4) {
    const float px = mcx[r], py = mcy[r];
    int cxi = (int)rintf(px / a.osf), cyi = (int)rintf(py / a.osf);
    cxi = cxi < 0 ? 0 : (cxi > a.w - 1 ? a.w - 1 : cxi);
    cyi = cyi < 0 ? 0 : (cyi > a.h - 1 ? a.h - 1 : cyi);
    unsigned long long key[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = lane + 64 * s;
      key[s] = ~0ull;
      if (c < 81) {
        const int ix = cxi + (c % 9) - 4, iy = cyi + (c / 9) - 4;
        if (ix >= 0 && ix < a.w && iy >= 0 && iy < a.h) {
          const float dx = (float)ix * a.osf - px, dy = (float)iy * a.osf - py;
          const float d = dx * dx + dy * dy;
          key[s] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(iy * a.w + ix);
        }
      }
    }
    for (int it = 0; it < a.topk; ++it) {
      unsigned long long best = key[0] < key[1] ? key[0] : key[1];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(best & 0xFFFFFFFFull), o);
        const unsigned hi = __shfl_xor((unsigned)(best >> 32), o);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        best = other < best ? other : best;
      }
      if (best == ~0ull) break;
      if (key[0] == best) key[0] = ~0ull;
      if (key[1] == best) key[1] = ~0ull;
      if (lane == 0) {
        const unsigned aidx = (unsigned)(best & 0xFFFFFFFFull);
        atomicOr(&bits[aidx >> 6], 1ull << (aidx & 63));
      }
    }
  }
  __syncthreads();
  // exclusive prefix popcount over the bitmap words (one wave)
  if (wave == 0) {
    int carry = 0;
    for (int base = 0; base < words; base += 64) {
      const int i = base + lane;
      const int c = i < words ? __popcll(bits[i]) : 0;
      int incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if (i < words) prefix[i] = carry + incl - c;
      carry += __shfl(incl, 63);
    }
  }
  // clear this (task, sample)'s outputs
  const size_t tb = (size_t)t * a.B + b;
  for (int i = tid; i < a.ncm * A; i += 256) hm[tb * a.ncm * A + i] = 0.f;
  for (int k = tid; k < a.K; k += 256) {
    ind[tb * a.K + k] = 0;
    mask[tb * a.K + k] = 0;
    cat[tb * a.K + k] = 0;
  }
  for (int i = tid; i < a.K * a.enc_dim; i += 256) enc[tb * a.K * a.enc_dim + i] = 0.f;
  __syncthreads();
  // positives: nearest member (task order, first on ties), heat map, slot outputs
  for (int wi = tid; wi < words; wi += 256) {
    unsigned long long wbits = bits[wi];
    int rank = prefix[wi];
    while (wbits) {
      const int bit = __ffsll((long long)wbits) - 1;
      wbits &= wbits - 1;
      const int aidx = wi * 64 + bit;
      const float ax = (float)(aidx % a.w) * a.osf, ay = (float)(aidx / a.w) * a.osf;
      float best = 3.4e38f;
      int gr = 0;
      for (int r = 0; r < nm; ++r) {
        const float dx = ax - mcx[r], dy = ay - mcy[r];
        const float d = dx * dx + dy * dy;
        if (d < best) { best = d; gr = r; }
      }
      const int m = order[gr];
      const int cls_off = coff[m] < 0 ? 0 : coff[m];
      hm[(tb * a.ncm + cls_off) * A + aidx] = 1.f;
      if (rank < a.K) {
        const size_t so = tb * a.K + rank;
        ind[so] = aidx;
        mask[so] = 1;
        cat[so] = coff[m];
        const float* bx = box + m * a.cols;
        float* e = enc + so * a.enc_dim;
        const float cx = (bx[0] - a.pc0) / a.vs0, cy = (bx[1] - a.pc1) / a.vs1;
        const float dxv = bx[3] / a.vs0, dyv = bx[4] / a.vs1;
        const float yaw = limit_period_2pi(bx[6]);
        e[0] = (cx - ax) / a.osf;
        e[1] = (cy - ay) / a.osf;
        e[2] = bx[2];
        e[3] = logf(dxv * a.vs0);
        e[4] = logf(dyv * a.vs1);
        e[5] = logf(bx[5]);
        e[6] = sinf(yaw);
        e[7] = cosf(yaw);
        for (int c = 7; c < a.cols - 1 && 8 + (c - 7) < a.enc_dim; ++c) e[8 + (c - 7)] = bx[c];
      }
      ++rank;
    }
  }
}

}  // namespace

// gt f32[B, M, cols] (last column = 1-based class id); task_of / off_of i8[n_classes] by class id.
// hm f32[T,B,ncm,h*w], ind i64[T,B,K], mask u8[T,B,K], cat i64[T,B,K], enc f32[T,B,K,enc_dim].
extern "C" int ud_assign_targets(const float* gt, int B, int M, int cols, const signed char* task_of,
                                 const signed char* off_of, int n_classes, int T, int ncm, int w, int h,
                                 int K, int topk, int enc_dim, float osf, float pc0, float pc1, float vs0,
                                 float vs1, float* hm, long long* ind, unsigned char* mask, long long* cat,
                                 float* enc, ud_stream_t stream_) {
  if (!gt || !task_of || !off_of || !hm || !ind || !mask || !cat || !enc || B <= 0 || M <= 0 || T <= 0 ||
      ncm <= 0 || w <= 0 || h <= 0 || K <= 0 || topk <= 0 || n_classes <= 0)
    return UD_ERR_INVALID_ARG;
  if (M > kMaxBoxes || cols > kMaxBoxDim || cols < 8 || n_classes > kMaxClasses || topk > 9 || w < 9 || h < 9 ||
      w * h > 180 * 180 || enc_dim < 8)
    return UD_ERR_UNSUPPORTED;
  AssignArgs a;
  a.T = T; a.B = B; a.M = M; a.cols = cols; a.w = w; a.h = h; a.K = K; a.topk = topk; a.ncm = ncm;
  a.enc_dim = enc_dim; a.osf = osf; a.pc0 = pc0; a.pc1 = pc1; a.vs0 = vs0; a.vs1 = vs1;
  for (int i = 0; i < kMaxClasses; ++i) {
    a.task_of[i] = i < n_classes ? task_of[i] : -1;
    a.off_of[i] = i < n_classes ? off_of[i] : 0;
  }
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("assign.k_assign_targets", stream);
  k_assign_targets<<<dim3(T, B), 256, 0, stream>>>(a, gt, hm, ind, mask, cat, enc);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
