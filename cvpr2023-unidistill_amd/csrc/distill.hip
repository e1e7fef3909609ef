// Fused distillation-loss kernels for MI355X / gfx950.
//
// Reference (module-level functions of the four *_distill_* experiment files, e.g.
// unidistill/exps/multisensor_fusion/nuscenes/BEVFusion/
//   BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar.py):
//   :73-97   center_to_corner_box2d        :100-178 calculate_box_mask_gaussian (host numpy!)
//   :196-245 FeatureDistillLoss            :248-323 BEVDistillLoss
//   :326-385 ResponseDistillLoss           :449-455, :466-483 valid-box scan + BEV-pixel corners
//
// The reference builds the 9 key points with ~10 small torch ops, runs two full grid_sample
// launches per loss, concatenates 24 head tensors, and computes the gaussian mask with numpy loops
// on the host (device->host sync + host->device copy per step).  Here each loss is one forward and
// one backward kernel reading the BEV maps in place; everything stays on the device.
#include "ud_common.h"
#include "ud_prof.h"

namespace {

// element (b, c, y, x) of a BEV map lives at b*sb + c*sc + y*sy + x*sx
struct MapView {
  const float* p;
  long long sb, sc, sy, sx;
};
struct MapViewW {
  float* p;
  long long sb, sc, sy, sx;
};

struct Bilin {
  int x0, y0;
  float w[4];   // nw, ne, sw, se
  bool in[4];
};

// grid_sample(align_corners=False, bilinear, zeros) coordinates for key point (px0, px1) given in
// BEV pixels: the reference normalises column 0 by w and column 1 by h, then SWAPS the columns
// (distill_lidar.py:224-226), so the width-direction sample coordinate comes from px1.
__device__ __forceinline__ Bilin make_bilin(float px0, float px1, int H, int W) {
  const float a0 = (px0 - W / 2.0f) / (W / 2.0f);
  const float a1 = (px1 - H / 2.0f) / (H / 2.0f);
  const float gx = a1, gy = a0;
  const float ix = ((gx + 1.0f) * W - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * H - 1.0f) / 2.0f;
  const float fx = floorf(ix), fy = floorf(iy);
  Bilin b;
  b.x0 = (int)fx;
  b.y0 = (int)fy;
  const float x1 = fx + 1.0f, y1 = fy + 1.0f;
  b.w[0] = (x1 - ix) * (y1 - iy);
  b.w[1] = (ix - fx) * (y1 - iy);
  b.w[2] = (x1 - ix) * (iy - fy);
  b.w[3] = (ix - fx) * (iy - fy);
  const bool xin0 = b.x0 >= 0 && b.x0 < W, xin1 = b.x0 + 1 >= 0 && b.x0 + 1 < W;
  const bool yin0 = b.y0 >= 0 && b.y0 < H, yin1 = b.y0 + 1 >= 0 && b.y0 + 1 < H;
  b.in[0] = xin0 && yin0;
  b.in[1] = xin1 && yin0;
  b.in[2] = xin0 && yin1;
  b.in[3] = xin1 && yin1;
  return b;
}

__device__ __forceinline__ float sample(const MapView& m, int b, int c, const Bilin& q) {
  const float* base = m.p + b * m.sb + c * m.sc;
  float v = 0.f;
  if (q.in[0]) v += base[q.y0 * m.sy + q.x0 * m.sx] * q.w[0];
  if (q.in[1]) v += base[q.y0 * m.sy + (q.x0 + 1) * m.sx] * q.w[1];
  if (q.in[2]) v += base[(q.y0 + 1) * m.sy + q.x0 * m.sx] * q.w[2];
  if (q.in[3]) v += base[(q.y0 + 1) * m.sy + (q.x0 + 1) * m.sx] * q.w[3];
  return v;
}

// 9 key points of a box from its 4 BEV corners: corners, centre, 4 edge mid-points (:200-223)
__device__ __forceinline__ void key_point(const float* __restrict__ c8, int kp, float& x, float& y) {
  if (kp < 4) {
    x = c8[kp * 2];
    y = c8[kp * 2 + 1];
  } else if (kp == 4) {
    x = (((c8[0] + c8[2]) + c8[4]) + c8[6]) / 4.0f;
    y = (((c8[1] + c8[3]) + c8[5]) + c8[7]) / 4.0f;
  } else {
    const int pairs[4][2] = {{0, 1}, {1, 2}, {2, 3}, {0, 3}};
    const int i = pairs[kp - 5][0], j = pairs[kp - 5][1];
    x = (c8[i * 2] + c8[j * 2]) / 2.0f;
    y = (c8[i * 2 + 1] + c8[j * 2 + 1]) / 2.0f;
  }
}

// ---- box corners in BEV pixels + validity ------------------------------------------------------
__global__ void k_box_corners(const float* __restrict__ gt, int B, int M, int S, double pc0,
                              double pc1, float px0, float px1, float* __restrict__ corners,
                              unsigned char* __restrict__ valid) {
  const int b = blockIdx.x;
  // last row (searching from the end, never below row 0) whose entries do not sum to zero
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    int cnt = M - 1;
    while (cnt > 0) {
      float s = 0.f;
      for (int k = 0; k < S; ++k) s += gt[((size_t)b * M + cnt) * S + k];
      if (s != 0.f) break;
      --cnt;
    }
    s_last = cnt;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    const float* g = gt + ((size_t)b * M + m) * S;
    valid[b * M + m] = (m <= s_last) ? 1 : 0;
    const float sn = sinf(g[6]), cs = cosf(g[6]);
    const double nx[4] = {-0.5, -0.5, 0.5, 0.5}, ny[4] = {-0.5, 0.5, 0.5, -0.5};
    for (int k = 0; k < 4; ++k) {
      const double lx = (double)g[3] * nx[k], ly = (double)g[4] * ny[k];
      const double wx = lx * (double)cs - ly * (double)sn + (double)g[0];
      const double wy = lx * (double)sn + ly * (double)cs + (double)g[1];
      const float fx = (float)wx, fy = (float)wy;  // stored into a float32 tensor (:476)
      corners[((size_t)(b * M + m) * 4 + k) * 2 + 0] = (fx - (float)pc0) / px0;
      corners[((size_t)(b * M + m) * 4 + k) * 2 + 1] = (fy - (float)pc1) / px1;
    }
  }
}

// ---- feature distillation ----------------------------------------------------------------------
// one 256-thread workgroup per box: box_loss[b,m] = valid * mean_kp mean_c |s - t| at the 9 key points
template <bool BWD>
__global__ __launch_bounds__(256) void k_feat(MapView s, MapView t, const float* __restrict__ corners,
                                              const unsigned char* __restrict__ valid, int M, int C,
                                              int H, int W, float* __restrict__ box_loss,
                                              float* __restrict__ dkp, const float* __restrict__ gscale) {
  __shared__ float s_kp[9];
  const int bm = blockIdx.x, b = bm / M;
  const int lane = ud_lane(), wave = threadIdx.x >> 6;
  if (!valid[bm]) {
    if (!BWD && threadIdx.x == 0) box_loss[bm] = 0.f;
    return;
  }
  const float* c8 = corners + (size_t)bm * 8;
  const float g = BWD ? (*gscale) / (float)(C * 9) : 0.f;
  for (int kp = wave; kp < 9; kp += 4) {
    float x, y;
    key_point(c8, kp, x, y);
    const Bilin q = make_bilin(x, y, H, W);
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float d = sample(s, b, c, q) - sample(t, b, c, q);
      if (BWD) {
        const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        dkp[((size_t)bm * 9 + kp) * C + c] = sg * g;       // gradient w.r.t. the sampled value; k_box_scatter spreads it
      } else {
        acc += fabsf(d);
      }
    }
    if (!BWD) {
      acc = ud_wave_sum(acc);
      if (lane == 0) s_kp[kp] = acc / (float)C;
    }
  }
  if (!BWD) {
    __syncthreads();
    if (threadIdx.x == 0) {
      float tot = 0.f;
      for (int k = 0; k < 9; ++k) tot += s_kp[k];
      box_loss[bm] = tot / 9.0f;
    }
  }
}

// ---- relation (BEV) distillation -----------------------------------------------------------------
// one workgroup per box: rows f_k = sampled [9, C]; normalise f/(|f| + 1e-4); 9x9 Gram; L1 between
// the student's and the teacher's Gram, mean over the 81 entries.
template <bool BWD>
__global__ __launch_bounds__(256) void k_rel(MapView s, MapView t, const float* __restrict__ corners,
                                             const unsigned char* __restrict__ valid, int M, int C,
                                             int H, int W, float* __restrict__ box_loss,
                                             float* __restrict__ dkp, const float* __restrict__ gscale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Fs = reinterpret_cast<float*>(smem);  // [9][C] student samples (raw, then normalised)
  float* Ft = Fs + 9 * C;                       // [9][C] teacher
  float* Gd = Ft + 9 * C;                       // [81]   Gs - Gt (fwd) / dL/dGs (bwd)
  float* nrm = Gd + 81;                         // [18]   |f| student 0..8, teacher 9..17
  float* red = nrm + 18;                        // [4]
  const int bm = blockIdx.x, b = bm / M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!valid[bm]) {
    if (!BWD && tid == 0) box_loss[bm] = 0.f;
    return;
  }
  const float* c8 = corners + (size_t)bm * 8;
  for (int kp = wave; kp < 9; kp += 4) {
    float x, y;
    key_point(c8, kp, x, y);
    const Bilin q = make_bilin(x, y, H, W);
    float ns = 0.f, nt = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float a = sample(s, b, c, q), d = sample(t, b, c, q);
      Fs[kp * C + c] = a;
      Ft[kp * C + c] = d;
      ns += a * a;
      nt += d * d;
    }
    ns = ud_wave_sum(ns);
    nt = ud_wave_sum(nt);
    if (lane == 0) {
      nrm[kp] = sqrtf(ns);
      nrm[9 + kp] = sqrtf(nt);
    }
  }
  __syncthreads();
  // Gram entries: pair p = i*9 + j; normalised rows are formed on the fly
  for (int p = wave; p < 81; p += 4) {
    const int i = p / 9, j = p - i * 9;
    const float si = nrm[i] + 1e-4f, sj = nrm[j] + 1e-4f;
    const float ti = nrm[9 + i] + 1e-4f, tj = nrm[9 + j] + 1e-4f;
    float ds = 0.f, dt = 0.f;
    for (int c = lane; c < C; c += 64) {
      ds += (Fs[i * C + c] / si) * (Fs[j * C + c] / sj);
      dt += (Ft[i * C + c] / ti) * (Ft[j * C + c] / tj);
    }
    ds = ud_wave_sum(ds);
    dt = ud_wave_sum(dt);
    if (lane == 0) Gd[p] = ds - dt;
  }
  __syncthreads();
  if (!BWD) {
    if (tid == 0) {
      float tot = 0.f;
      for (int p = 0; p < 81; ++p) tot += fabsf(Gd[p]);
      box_loss[bm] = tot / 81.0f;
    }
    return;
  }
  // ---- backward w.r.t. the student map
  const float g = (*gscale) / 81.0f;
  if (tid < 81) {
    const float d = Gd[tid];
    Gd[tid] = ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) * g;
  }
  __syncthreads();
  for (int kp = wave; kp < 9; kp += 4) {
    const float n = nrm[kp], den = n + 1e-4f;
    // dFhat[kp][c] = sum_j (dG[kp][j] + dG[j][kp]) * Fhat[j][c];  dot = f . dFhat
    float dot = 0.f;
    for (int c = lane; c < C; c += 64) {
      float dh = 0.f;
      for (int j = 0; j < 9; ++j)
        dh += (Gd[kp * 9 + j] + Gd[j * 9 + kp]) * (Fs[j * C + c] / (nrm[j] + 1e-4f));
      dot += Fs[kp * C + c] * dh;
    }
    dot = ud_wave_sum(dot);
    const float k2 = (n > 0.f) ? dot / (n * den * den) : 0.f;
    for (int c = lane; c < C; c += 64) {
      float dh = 0.f;
      for (int j = 0; j < 9; ++j)
        dh += (Gd[kp * 9 + j] + Gd[j * 9 + kp]) * (Fs[j * C + c] / (nrm[j] + 1e-4f));
      dkp[((size_t)bm * 9 + kp) * C + c] = dh / den - Fs[kp * C + c] * k2;
    }
  }
  (void)red;
}

// ---- deterministic scatter of the key-point gradients -----------------------------------------------------------------
// d loss / d map[b, c, pixel] = sum over (box m, key point kp, bilinear corner i) of dkp[b, m, kp, c] * w_i.  Footprints
// overlap (a small box puts all nine key points into one or two BEV cells; neighbouring boxes share pixels), so adding
// the terms with float atomics -- the first version -- made the sum depend on the arrival order whenever three or more
// terms met (a run-to-run difference of one ulp in every gradient upstream, seen as a flaky bit-reproducibility test on
// cold GPUs).  Here every (m, kp, i) term of a sample gets the key (pixel, term index); the keys of a sample are sorted,
// runs of equal pixels are found, and a lane (= channel) adds the terms of a pixel in term order and stores the pixel
// once -- no atomics, fixed order.  Two sorters, chosen by the number of terms T = 36 * M (M = padded box count):
//   * T <= 16384 (M <= 455): bitonic sort in LDS (dynamic, 10 bytes per term), repeated by the (64 channels, 1/8 of the
//     runs) workgroups of a sample, which is cheaper than a second launch;
//   * larger: rank sort through the workspace (every key is unique, so its rank = the number of smaller keys; O(T^2)
//     compares, no size limit), then one pass over the sorted keys.
constexpr int kMinTerms = 2048;                // 8 sorted entries per thread at least
constexpr int kMaxLdsTerms = 16384;            // 160 KB of LDS would hold 16384 * 10 bytes
constexpr int kRunSplit = 8;                   // workgroups sharing the pixels of one (sample, 64 channels)

__device__ __forceinline__ bool term_pixel(const float* __restrict__ corners, const unsigned char* __restrict__ valid,
                                           int b, int M, int t, int H, int W, int& pix, float& wv) {
  const int m = t / 36, kp = (t - m * 36) >> 2, i = t & 3;
  if (!valid[b * M + m]) return false;
  float x, y;
  key_point(corners + (size_t)(b * M + m) * 8, kp, x, y);
  const Bilin q = make_bilin(x, y, H, W);
  if (!q.in[i]) return false;
  pix = (q.y0 + (i >> 1)) * W + q.x0 + (i & 1);
  wv = q.w[i];
  return true;
}

__global__ __launch_bounds__(256) void k_box_scatter(const float* __restrict__ dkp, const float* __restrict__ corners,
                                                     const unsigned char* __restrict__ valid, int M, int C, int H,
                                                     int W, MapViewW gs, int T, int tbits, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem_sc[];
  unsigned* keys = reinterpret_cast<unsigned*>(smem_sc);                    // [T]
  float* wts = reinterpret_cast<float*>(keys + T);                          // [T]
  unsigned short* starts = reinterpret_cast<unsigned short*>(wts + T);      // [T + 1] first sorted index of every run
  __shared__ int s_wave[4], s_runs;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nterm = M * 36;
  const unsigned tmask = (1u << tbits) - 1u;
  for (int t = tid; t < T; t += 256) {
    unsigned key = 0xFFFFFFFFu;
    float wv = 0.f;
    int pix;
    if (t < nterm && term_pixel(corners, valid, b, M, t, H, W, pix, wv)) key = ((unsigned)pix << tbits) | (unsigned)t;
    keys[t] = key;
    wts[t] = (key != 0xFFFFFFFFu) ? wv : 0.f;
  }
  __syncthreads();
  for (int k = 2; k <= T; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < T; t += 256) {
        const int p = t ^ j;
        if (p > t) {
          const unsigned a = keys[t], c2 = keys[p];
          const bool up = (t & k) == 0;
          if ((a > c2) == up) { keys[t] = c2; keys[p] = a; }
        }
      }
      __syncthreads();
    }
  // run starts: thread t owns sorted entries [E t, E t + E); block-wide exclusive scan of the per-thread start counts
  const int E = T >> 8;                                // 8 .. 64
  int cnt = 0;
  unsigned long long flags = 0ull;
  for (int e = 0; e < E; ++e) {
    const int i = tid * E + e;
    const unsigned key = keys[i];
    const bool st = key != 0xFFFFFFFFu && (i == 0 || (keys[i - 1] >> tbits) != (key >> tbits));
    flags |= (unsigned long long)st << e;
    cnt += st;
  }
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  int off = inc - cnt;
  for (int w2 = 0; w2 < wave; ++w2) off += s_wave[w2];
  for (int e = 0; e < E; ++e)
    if ((flags >> e) & 1ull) starts[off++] = (unsigned short)(tid * E + e);
  if (tid == 255) s_runs = off;                   // total number of runs
  // number of valid (sorted-to-the-front) entries = end of the last run
  __syncthreads();
  const int nruns = s_runs;
  if (tid == 0) {
    int lo = 0, hi = T;                             // binary search for the first invalid key
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (keys[mid] == 0xFFFFFFFFu) hi = mid; else lo = mid + 1;
    }
    starts[nruns] = (unsigned short)lo;
  }
  __syncthreads();
  const int c = blockIdx.y * 64 + lane;
  if (c >= C) return;
  float* __restrict__ base = gs.p + b * gs.sb + c * gs.sc;
  const float* __restrict__ dk = dkp + (size_t)b * M * 9 * C + c;
  for (int r = blockIdx.z + kRunSplit * wave; r < nruns; r += kRunSplit * 4) {
    const int i0 = starts[r], i1 = starts[r + 1];
    const int pix = (int)(keys[i0] >> tbits);
    float acc = 0.f;
    for (int i = i0; i < i1; ++i) {
      const int t = (int)(keys[i] & tmask);
      acc += dk[(size_t)((t / 36) * 9 + ((t % 36) >> 2)) * C] * wts[t];
    }
    float* dst = base + (pix / W) * gs.sy + (pix % W) * gs.sx;
    *dst = accumulate ? *dst + acc : acc;       // accumulate: the map already holds the other consumers' gradient
  }
}

// -- the rank-sort path (T > kMaxLdsTerms).  Keys are 64-bit (pixel << 32 | term), invalid terms sort to the end.
__global__ __launch_bounds__(256) void k_terms_make(const float* __restrict__ corners,
                                                    const unsigned char* __restrict__ valid, int M, int H, int W,
                                                    int T, unsigned long long* __restrict__ keys,
                                                    float* __restrict__ wts) {
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  int pix = 0;
  float wv = 0.f;
  const bool ok = term_pixel(corners, valid, b, M, t, H, W, pix, wv);
  keys[(size_t)b * T + t] = ((unsigned long long)(ok ? (unsigned)pix : 0xFFFFFFFFu) << 32) | (unsigned)t;
  wts[(size_t)b * T + t] = ok ? wv : 0.f;
}

__global__ __launch_bounds__(256) void k_terms_rank(const unsigned long long* __restrict__ keys, int T,
                                                    unsigned long long* __restrict__ sorted) {
  __shared__ unsigned long long tile[256];
  const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long* kb = keys + (size_t)b * T;
  const unsigned long long mine = (t < T) ? kb[t] : ~0ull;
  int rank = 0;
  for (int j0 = 0; j0 < T; j0 += 256) {
    __syncthreads();
    tile[threadIdx.x] = (j0 + threadIdx.x < T) ? kb[j0 + threadIdx.x] : ~0ull;
    __syncthreads();
#pragma unroll 8
    for (int j = 0; j < 256; ++j) rank += tile[j] < mine;
  }
  if (t < T) sorted[(size_t)b * T + rank] = mine;
}

// one wave per 64 sorted entries; a lane is a channel.  The wave visits the run starts among its entries in order and
// follows each run to its end (possibly into the next wave's entries).
__global__ __launch_bounds__(256) void k_box_scatter_sorted(const float* __restrict__ dkp,
                                                            const unsigned long long* __restrict__ sorted,
                                                            const float* __restrict__ wts, int M, int C, int W,
                                                            int T, MapViewW gs, int accumulate) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long* sk = sorted + (size_t)b * T;
  const float* wt = wts + (size_t)b * T;
  const int i = (blockIdx.z * 4 + wave) * 64 + lane;
  bool st = false;
  if (i < T) {
    const unsigned pix = (unsigned)(sk[i] >> 32);
    st = pix != 0xFFFFFFFFu && (i == 0 || (unsigned)(sk[i - 1] >> 32) != pix);
  }
  unsigned long long todo = __ballot(st);
  const int c = blockIdx.y * 64 + lane;
  if (c >= C) return;
  float* __restrict__ base = gs.p + b * gs.sb + c * gs.sc;
  const float* __restrict__ dk = dkp + (size_t)b * M * 9 * C + c;
  const int i_base = (blockIdx.z * 4 + wave) * 64;
  while (todo) {
    const int first = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    int j = i_base + first;
    const unsigned pix = (unsigned)(sk[j] >> 32);
    float acc = 0.f;
    for (; j < T && (unsigned)(sk[j] >> 32) == pix; ++j) {
      const int t = (int)(unsigned)sk[j];
      acc += dk[(size_t)((t / 36) * 9 + ((t % 36) >> 2)) * C] * wt[t];
    }
    float* dst = base + ((int)pix / W) * gs.sy + ((int)pix % W) * gs.sx;
    *dst = accumulate ? *dst + acc : acc;
  }
}

// ---- gaussian box mask ---------------------------------------------------------------------------
struct BoxG {
  int cx, cy, r;
};

__device__ double gaussian_radius_d(double height, double width, double min_overlap) {
  const double a1 = 1.0, b1 = height + width;
  const double c1 = width * height * (1.0 - min_overlap) / (1.0 + min_overlap);
  const double r1 = (b1 + sqrt(b1 * b1 - 4.0 * a1 * c1)) / 2.0;
  const double a2 = 4.0, b2 = 2.0 * (height + width);
  const double c2 = (1.0 - min_overlap) * width * height;
  const double r2 = (b2 + sqrt(b2 * b2 - 4.0 * a2 * c2)) / 2.0;
  const double a3 = 4.0 * min_overlap, b3 = -2.0 * min_overlap * (height + width);
  const double c3 = (min_overlap - 1.0) * width * height;
  const double r3 = (b3 + sqrt(b3 * b3 - 4.0 * a3 * c3)) / 2.0;
  return fmin(r1, fmin(r2, r3));
}

__global__ void k_mask_boxes(const float* __restrict__ gt, int B, int M, int S, double pc0,
                             double pc1, double px0, double px1, BoxG* __restrict__ boxes,
                             int* __restrict__ nbox) {
  const int b = blockIdx.x;
  __shared__ int s_n;
  if (threadIdx.x == 0) {  // boxes are drawn until the first all-zero row (:112-113)
    int n = 0;
    for (; n < M; ++n) {
      float s = 0.f;
      for (int k = 0; k < S; ++k) s += gt[((size_t)b * M + n) * S + k];
      if (s == 0.f) break;
    }
    s_n = n;
    nbox[b] = n;
  }
  __syncthreads();
  for (int m = threadIdx.x; m < s_n; m += blockDim.x) {
    const float* g = gt + ((size_t)b * M + m) * S;
    const double w = (double)g[3] / px0, h = (double)g[4] / px1;
    double r = gaussian_radius_d(w, h, 0.7);
    BoxG o;
    o.r = (r > 0.0) ? (int)r : 0;  // max(0, int(radius)); NaN -> 0
    o.cx = (int)(((double)g[0] - pc0) / px0);
    o.cy = (int)(((double)g[1] - pc1) / px1);
    boxes[b * M + m] = o;
  }
}

__global__ __launch_bounds__(256) void k_mask_pixels(const BoxG* __restrict__ boxes,
                                                     const int* __restrict__ nbox, int M, int H,
                                                     int W, float* __restrict__ mask) {
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int py = pix / W, px = pix - py * W;
  float v = 0.f;
  const int n = nbox[b];
  for (int m = 0; m < n; ++m) {
    const BoxG o = boxes[b * M + m];
    const int dx = px - o.cx, dy = py - o.cy;
    if (dx < -o.r || dx > o.r || dy < -o.r || dy > o.r) continue;
    // draw_umich_gaussian slices are empty when the centre lies more than r outside the low edge
    if (o.cx + o.r + 1 <= 0 || o.cy + o.r + 1 <= 0) continue;
    const double sigma = (2.0 * o.r + 1.0) / 6.0;
    const float gval = (float)exp(-(double)(dx * dx + dy * dy) / (2.0 * sigma * sigma));
    v = fmaxf(v, gval);
  }
  mask[(size_t)b * H * W + pix] = v;
}

// ---- response distillation -------------------------------------------------------------------------
constexpr int kMaxHm = 8;
constexpr int kMaxReg = 48;
struct RespStride {
  int sb, sc, sp;       // element strides of batch, channel and (linearised) pixel: NCHW = (ch HW, HW, 1), a channels-last map = (HW ch, 1, ch)
};
struct RespArgs {
  const float* s_hm[kMaxHm];
  const float* t_hm[kMaxHm];
  float* g_hm[kMaxHm];
  int hm_ch[kMaxHm];
  RespStride s_hm_st[kMaxHm], t_hm_st[kMaxHm], g_hm_st[kMaxHm];
  int n_hm;
  const float* s_reg[kMaxReg];
  const float* t_reg[kMaxReg];
  float* g_reg[kMaxReg];
  int reg_ch[kMaxReg];
  RespStride s_reg_st[kMaxReg], t_reg_st[kMaxReg], g_reg_st[kMaxReg];
  int n_reg;
  int reg_total;
  float lo, hi;  // teacher sigmoid clamp
};

__device__ __forceinline__ float teacher_prob(float logit, float lo, float hi) {
  const float y = 1.0f / (1.0f + expf(-(logit / 2.0f)));
  return fminf(fmaxf(y, lo), hi);
}

// one thread per BEV pixel; every tensor [B, ch, H, W] comes with its (batch, channel, pixel) strides -- dense NCHW, channels-last
// maps and channel slices of a packed channels-last head output are all read in place
template <bool BWD>
__global__ __launch_bounds__(256) void k_resp(RespArgs a, const float* __restrict__ mask, int HW,
                                              float* __restrict__ partial,
                                              const float* __restrict__ gscale_cls,
                                              const float* __restrict__ gscale_reg) {
  __shared__ float s_c[4], s_r[4];
  const int b = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool act = pix < HW;
  float dc = 0.f, dr = 0.f;
  if (act) {
    const float mk = mask[(size_t)b * HW + pix];
    // class response: max over all heat-map channels of all tasks
    float smax = -INFINITY, tmax = -INFINITY;
    int arg_t = 0, arg_c = 0;
    for (int i = 0; i < a.n_hm; ++i)
      for (int c = 0; c < a.hm_ch[i]; ++c) {
        const float sv = a.s_hm[i][(size_t)b * a.s_hm_st[i].sb + (size_t)c * a.s_hm_st[i].sc + (size_t)pix * a.s_hm_st[i].sp];
        if (sv > smax) {
          smax = sv;
          arg_t = i;
          arg_c = c;
        }
        tmax = fmaxf(tmax, teacher_prob(a.t_hm[i][(size_t)b * a.t_hm_st[i].sb + (size_t)c * a.t_hm_st[i].sc +
                                                  (size_t)pix * a.t_hm_st[i].sp], a.lo, a.hi));
      }
    const float d = smax - tmax;
    if (!BWD) {
      dc = fabsf(d) * mk;
    } else {
      const float gc = (*gscale_cls) * mk * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
      for (int i = 0; i < a.n_hm; ++i)
        for (int c = 0; c < a.hm_ch[i]; ++c)
          a.g_hm[i][(size_t)b * a.g_hm_st[i].sb + (size_t)c * a.g_hm_st[i].sc + (size_t)pix * a.g_hm_st[i].sp] =
              (i == arg_t && c == arg_c) ? gc : 0.f;
    }
    // box-regression response: mean over all regression channels of |s - t|
    float acc = 0.f;
    const float gr = BWD ? (*gscale_reg) * mk / (float)a.reg_total : 0.f;
    for (int i = 0; i < a.n_reg; ++i)
      for (int c = 0; c < a.reg_ch[i]; ++c) {
        const float e = a.s_reg[i][(size_t)b * a.s_reg_st[i].sb + (size_t)c * a.s_reg_st[i].sc + (size_t)pix * a.s_reg_st[i].sp] -
                        a.t_reg[i][(size_t)b * a.t_reg_st[i].sb + (size_t)c * a.t_reg_st[i].sc + (size_t)pix * a.t_reg_st[i].sp];
        if (!BWD)
          acc += fabsf(e);
        else
          a.g_reg[i][(size_t)b * a.g_reg_st[i].sb + (size_t)c * a.g_reg_st[i].sc + (size_t)pix * a.g_reg_st[i].sp] =
              gr * ((e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f));
      }
    dr = (acc / (float)a.reg_total) * mk;
  }
  if (BWD) return;
  dc = ud_wave_sum(dc);
  dr = ud_wave_sum(dr);
  const int lane = ud_lane(), wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_c[wave] = dc;
    s_r[wave] = dr;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[blk * 2 + 0] = ((s_c[0] + s_c[1]) + s_c[2]) + s_c[3];
    partial[blk * 2 + 1] = ((s_r[0] + s_r[1]) + s_r[2]) + s_r[3];
  }
}

}  // namespace

// gt f32[B,M,S] (x,y,z,dx,dy,dz,yaw,...) -> corners_px f32[B,M,4,2] in BEV pixels
// ((corner - pc_min) / pixel_size) and valid u8[B,M] = rows up to the last non-zero row.
extern "C" int ud_distill_box_corners(const float* gt, int B, int M, int S, double pc_min_x,
                                      double pc_min_y, double pixel_x, double pixel_y,
                                      float* corners_px, unsigned char* valid,
                                      ud_stream_t stream_) {
  if (!gt || !corners_px || !valid || B <= 0 || M <= 0 || S < 7) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  k_box_corners<<<B, 128, 0, stream>>>(gt, B, M, S, pc_min_x, pc_min_y, (float)pixel_x,
                                       (float)pixel_y, corners_px, valid);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

static MapView mv(const float* p, const int64_t* st) { return MapView{p, st[0], st[1], st[2], st[3]}; }

// Feature (kind 0) / relation (kind 1) distillation, forward: box_loss f32[B,M] (0 for invalid
// boxes).  s/t are BEV maps [B,C,H,W] addressed through element strides (sb,sc,sy,sx) each.
extern "C" int ud_distill_box_fwd(int kind, const float* s, const int64_t* s_strides,
                                  const float* t, const int64_t* t_strides,
                                  const float* corners_px, const unsigned char* valid, int B, int M,
                                  int C, int H, int W, float* box_loss, ud_stream_t stream_) {
  if (!s || !t || !s_strides || !t_strides || !corners_px || !valid || !box_loss)
    return UD_ERR_INVALID_ARG;
  if (B <= 0 || M <= 0 || C <= 0 || H <= 0 || W <= 0 || (kind != 0 && kind != 1))
    return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (kind == 0) {
    UdProfScope prof("distill.k_feat", stream);
    k_feat<false><<<B * M, 256, 0, stream>>>(mv(s, s_strides), mv(t, t_strides), corners_px, valid,
                                             M, C, H, W, box_loss, nullptr, nullptr);
  } else {
    const size_t lds = (size_t)(18 * C + 81 + 18 + 4) * sizeof(float);
    if (lds > 160 * 1024) return UD_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_rel<false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    UdProfScope prof("distill.k_rel", stream);
    k_rel<false><<<B * M, 256, lds, stream>>>(mv(s, s_strides), mv(t, t_strides), corners_px, valid,
                                              M, C, H, W, box_loss, nullptr, nullptr);
  }
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// Backward w.r.t. the student map: gs (pre-zeroed by the caller, strides gs_strides) =
// d(sum_boxes box_loss * (*gscale))/ds.  gscale is a DEVICE scalar (upstream grad / normaliser),
// so no host sync is needed between the losses and their backward.  Two launches: the per-(box, key point, channel)
// gradients into `workspace` (ud_distill_box_bwd_workspace_bytes), then their deterministic scatter (k_box_scatter).
extern "C" size_t ud_distill_box_bwd_workspace_bytes(int B, int M, int C) {
  if (B <= 0 || M <= 0 || C <= 0) return 0;
  // key-point gradients + (rank-sort path only, but sized always: 20 bytes per term) keys, sorted keys, weights
  return ud_align_up((size_t)B * M * 9 * C * sizeof(float)) +
         2 * ud_align_up((size_t)B * M * 36 * sizeof(unsigned long long)) + ud_align_up((size_t)B * M * 36 * sizeof(float));
}

namespace {
int box_bwd_impl(int kind, const float* s, const int64_t* s_strides, const float* t, const int64_t* t_strides,
                 const float* corners_px, const unsigned char* valid, int B, int M, int C, int H, int W,
                 const float* gscale, float* gs, const int64_t* gs_strides, void* workspace, size_t workspace_bytes,
                 ud_stream_t stream_, int accumulate) {
  if (!s || !t || !s_strides || !t_strides || !corners_px || !valid || !gscale || !gs || !gs_strides)
    return UD_ERR_INVALID_ARG;
  if (B <= 0 || M <= 0 || C <= 0 || H <= 0 || W <= 0 || (kind != 0 && kind != 1))
    return UD_ERR_INVALID_ARG;
  if ((long long)H * W >= (1ll << 31) || (long long)M * 36 >= (1ll << 31)) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_distill_box_bwd_workspace_bytes(B, M, C)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* dkp = reinterpret_cast<float*>(workspace);
  MapViewW g{gs, gs_strides[0], gs_strides[1], gs_strides[2], gs_strides[3]};
  if (kind == 0) {
    k_feat<true><<<B * M, 256, 0, stream>>>(mv(s, s_strides), mv(t, t_strides), corners_px, valid,
                                            M, C, H, W, nullptr, dkp, gscale);
  } else {
    const size_t lds = (size_t)(18 * C + 81 + 18 + 4) * sizeof(float);
    if (lds > 160 * 1024) return UD_ERR_UNSUPPORTED;
    if (lds > 64 * 1024)
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_rel<true>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_rel<true><<<B * M, 256, lds, stream>>>(mv(s, s_strides), mv(t, t_strides), corners_px, valid,
                                             M, C, H, W, nullptr, dkp, gscale);
  }
  UD_LAUNCH_CHECK();
  const int nterm = M * 36;
  int T = kMinTerms, tbits = 11;
  while (T < nterm) { T <<= 1; ++tbits; }
  if (T <= kMaxLdsTerms && (long long)H * W < (1ll << (32 - tbits))) {
    const size_t lds = (size_t)T * 10 + 16;
    if (lds > 64 * 1024)
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_box_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_box_scatter<<<dim3(B, ud_div_up(C, 64), kRunSplit), 256, lds, stream>>>(dkp, corners_px, valid, M, C, H, W, g,
                                                                                 T, tbits, accumulate);
    UD_LAUNCH_CHECK();
    return UD_OK;
  }
  // rank-sort path
  const size_t need = ud_align_up((size_t)B * M * 9 * C * sizeof(float)) +
                      2 * ud_align_up((size_t)B * nterm * sizeof(unsigned long long)) +
                      ud_align_up((size_t)B * nterm * sizeof(float));
  if (workspace_bytes < need) return UD_ERR_WORKSPACE;
  UdArena a(workspace, workspace_bytes);
  a.take<float>((size_t)B * M * 9 * C);
  unsigned long long* keys = a.take<unsigned long long>((size_t)B * nterm);
  unsigned long long* sorted = a.take<unsigned long long>((size_t)B * nterm);
  float* wts = a.take<float>((size_t)B * nterm);
  const dim3 gt_(ud_div_up(nterm, 256), B);
  k_terms_make<<<gt_, 256, 0, stream>>>(corners_px, valid, M, H, W, nterm, keys, wts);
  UD_LAUNCH_CHECK();
  k_terms_rank<<<gt_, 256, 0, stream>>>(keys, nterm, sorted);
  UD_LAUNCH_CHECK();
  k_box_scatter_sorted<<<dim3(B, ud_div_up(C, 64), ud_div_up(nterm, 256)), 256, 0, stream>>>(dkp, sorted, wts, M, C, W,
                                                                                            nterm, g, accumulate);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
}  // namespace

extern "C" int ud_distill_box_bwd(int kind, const float* s, const int64_t* s_strides, const float* t, const int64_t* t_strides,
                                  const float* corners_px, const unsigned char* valid, int B, int M, int C, int H, int W,
                                  const float* gscale, float* gs, const int64_t* gs_strides, void* workspace,
                                  size_t workspace_bytes, ud_stream_t stream) {
  return box_bwd_impl(kind, s, s_strides, t, t_strides, corners_px, valid, B, M, C, H, W, gscale, gs, gs_strides, workspace,
                      workspace_bytes, stream, 0);
}

// The same gradient ADDED to a map that already holds the feature's gradient from its other consumer (the BEV trunk / the head):
// the touched pixels are read, added to and written back once each (same order: deterministic); nothing else is touched, so
// neither a zeroed map nor autograd's three-pass add of two dense maps is needed (4 x 512 x 180 x 180: 265 MB each).
extern "C" int ud_distill_box_bwd_acc(int kind, const float* s, const int64_t* s_strides, const float* t,
                                      const int64_t* t_strides, const float* corners_px, const unsigned char* valid, int B, int M,
                                      int C, int H, int W, const float* gscale, float* gs, const int64_t* gs_strides,
                                      void* workspace, size_t workspace_bytes, ud_stream_t stream) {
  return box_bwd_impl(kind, s, s_strides, t, t_strides, corners_px, valid, B, M, C, H, W, gscale, gs, gs_strides, workspace,
                      workspace_bytes, stream, 1);
}

// calculate_box_mask_gaussian on the device: gt f32[B,M,S] -> mask f32[B,H,W].
// workspace: B*M*12 + B*4 bytes.
extern "C" size_t ud_distill_mask_workspace_bytes(int B, int M) {
  if (B <= 0 || M <= 0) return 0;
  return ud_align_up((size_t)B * M * sizeof(BoxG)) + ud_align_up((size_t)B * sizeof(int));
}

extern "C" int ud_distill_gaussian_mask(const float* gt, int B, int M, int S, double pc_min_x,
                                        double pc_min_y, double pixel_x, double pixel_y, int H,
                                        int W, float* mask, void* workspace, size_t workspace_bytes,
                                        ud_stream_t stream_) {
  if (!gt || !mask || B <= 0 || M <= 0 || S < 5 || H <= 0 || W <= 0) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_distill_mask_workspace_bytes(B, M)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  UdArena a(workspace, workspace_bytes);
  BoxG* boxes = a.take<BoxG>((size_t)B * M);
  int* nbox = a.take<int>(B);
  k_mask_boxes<<<B, 128, 0, stream>>>(gt, B, M, S, pc_min_x, pc_min_y, pixel_x, pixel_y, boxes, nbox);
  UD_LAUNCH_CHECK();
  dim3 grid(ud_div_up((long long)H * W, 256), B);
  k_mask_pixels<<<grid, 256, 0, stream>>>(boxes, nbox, M, H, W, mask);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// strides: per tensor (sb, sc, sp) triples, or nullptr = dense NCHW
static bool resp_stride(RespStride& d, const int64_t* st, int i, int ch, int HW) {
  if (!st) {
    d = RespStride{ch * HW, HW, 1};
    return true;
  }
  const int64_t lim = 1ll << 31;
  if (st[3 * i] < 0 || st[3 * i] >= lim || st[3 * i + 1] < 0 || st[3 * i + 1] >= lim || st[3 * i + 2] < 0 || st[3 * i + 2] >= lim)
    return false;
  d = RespStride{(int)st[3 * i], (int)st[3 * i + 1], (int)st[3 * i + 2]};
  return true;
}

static int fill_resp(RespArgs& a, const float* const* s_hm, const float* const* t_hm,
                     float* const* g_hm, const int* hm_ch, int n_hm, const float* const* s_reg,
                     const float* const* t_reg, float* const* g_reg, const int* reg_ch, int n_reg,
                     float lo, float hi, int HW, const int64_t* s_hm_st = nullptr, const int64_t* t_hm_st = nullptr,
                     const int64_t* g_hm_st = nullptr, const int64_t* s_reg_st = nullptr, const int64_t* t_reg_st = nullptr,
                     const int64_t* g_reg_st = nullptr) {
  if (n_hm <= 0 || n_hm > kMaxHm || n_reg <= 0 || n_reg > kMaxReg) return UD_ERR_UNSUPPORTED;
  a.n_hm = n_hm;
  a.n_reg = n_reg;
  a.reg_total = 0;
  a.lo = lo;
  a.hi = hi;
  for (int i = 0; i < n_hm; ++i) {
    if (!s_hm[i] || !t_hm[i] || hm_ch[i] <= 0) return UD_ERR_INVALID_ARG;
    a.s_hm[i] = s_hm[i];
    a.t_hm[i] = t_hm[i];
    a.g_hm[i] = g_hm ? g_hm[i] : nullptr;
    a.hm_ch[i] = hm_ch[i];
    if (!resp_stride(a.s_hm_st[i], s_hm_st, i, hm_ch[i], HW) || !resp_stride(a.t_hm_st[i], t_hm_st, i, hm_ch[i], HW) ||
        !resp_stride(a.g_hm_st[i], g_hm_st, i, hm_ch[i], HW))
      return UD_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < n_reg; ++i) {
    if (!s_reg[i] || !t_reg[i] || reg_ch[i] <= 0) return UD_ERR_INVALID_ARG;
    a.s_reg[i] = s_reg[i];
    a.t_reg[i] = t_reg[i];
    a.g_reg[i] = g_reg ? g_reg[i] : nullptr;
    a.reg_ch[i] = reg_ch[i];
    a.reg_total += reg_ch[i];
    if (!resp_stride(a.s_reg_st[i], s_reg_st, i, reg_ch[i], HW) || !resp_stride(a.t_reg_st[i], t_reg_st, i, reg_ch[i], HW) ||
        !resp_stride(a.g_reg_st[i], g_reg_st, i, reg_ch[i], HW))
      return UD_ERR_UNSUPPORTED;
  }
  return UD_OK;
}

// ResponseDistillLoss forward.  Host arrays of device pointers: n_hm heat-map tensors
// [B,hm_ch[i],H,W] (student: post-sigmoid probabilities; teacher: logits -> clamp(sigmoid(x/2)))
// and n_reg regression tensors [B,reg_ch[i],H,W] (all dense NCHW).  partial f32[nblk,2] receives
// per-workgroup sums of (|cls diff|*mask, mean|reg diff|*mask); nblk = B*ceil(H*W/256).
extern "C" int ud_distill_resp_fwd(const float* const* s_hm, const float* const* t_hm,
                                   const int* hm_ch, int n_hm, const float* const* s_reg,
                                   const float* const* t_reg, const int* reg_ch, int n_reg,
                                   const float* mask, int B, int H, int W, float clamp_lo,
                                   float clamp_hi, float* partial, ud_stream_t stream_) {
  if (!s_hm || !t_hm || !hm_ch || !s_reg || !t_reg || !reg_ch || !mask || !partial)
    return UD_ERR_INVALID_ARG;
  RespArgs a;
  int rc = fill_resp(a, s_hm, t_hm, nullptr, hm_ch, n_hm, s_reg, t_reg, nullptr, reg_ch, n_reg,
                     clamp_lo, clamp_hi, H * W);
  if (rc != UD_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(ud_div_up((long long)H * W, 256), B);
  UdProfScope prof("distill.k_resp", stream);
  k_resp<false><<<grid, 256, 0, stream>>>(a, mask, H * W, partial, nullptr, nullptr);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// Backward: writes EVERY element of the student-side grads g_hm[i] / g_reg[i] (same shapes as the
// student tensors).  gscale_cls / gscale_reg are device scalars (upstream grad / normaliser).
extern "C" int ud_distill_resp_bwd(const float* const* s_hm, const float* const* t_hm,
                                   float* const* g_hm, const int* hm_ch, int n_hm,
                                   const float* const* s_reg, const float* const* t_reg,
                                   float* const* g_reg, const int* reg_ch, int n_reg,
                                   const float* mask, int B, int H, int W, float clamp_lo,
                                   float clamp_hi, const float* gscale_cls,
                                   const float* gscale_reg, ud_stream_t stream_) {
  if (!s_hm || !t_hm || !g_hm || !hm_ch || !s_reg || !t_reg || !g_reg || !reg_ch || !mask ||
      !gscale_cls || !gscale_reg)
    return UD_ERR_INVALID_ARG;
  RespArgs a;
  int rc = fill_resp(a, s_hm, t_hm, g_hm, hm_ch, n_hm, s_reg, t_reg, g_reg, reg_ch, n_reg, clamp_lo,
                     clamp_hi, H * W);
  if (rc != UD_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(ud_div_up((long long)H * W, 256), B);
  k_resp<true><<<grid, 256, 0, stream>>>(a, mask, H * W, nullptr, gscale_cls, gscale_reg);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// The same two passes on tensors read IN PLACE: every tensor [B, ch, H, W] comes with an (sb, sc, sp) element-stride triple (batch,
// channel, linearised pixel; the map must be pixel-linear: stride(H) == W * stride(W)) -- channels-last maps and channel slices of the
// packed head output need no dense NCHW copies (84 small copies per distillation step).  *_st: int64[n][3], host arrays.
extern "C" int ud_distill_resp_fwd_strided(const float* const* s_hm, const int64_t* s_hm_st, const float* const* t_hm,
                                           const int64_t* t_hm_st, const int* hm_ch, int n_hm, const float* const* s_reg,
                                           const int64_t* s_reg_st, const float* const* t_reg, const int64_t* t_reg_st,
                                           const int* reg_ch, int n_reg, const float* mask, int B, int H, int W,
                                           float clamp_lo, float clamp_hi, float* partial, ud_stream_t stream_) {
  if (!s_hm || !t_hm || !hm_ch || !s_reg || !t_reg || !reg_ch || !mask || !partial || !s_hm_st || !t_hm_st || !s_reg_st || !t_reg_st)
    return UD_ERR_INVALID_ARG;
  RespArgs a;
  int rc = fill_resp(a, s_hm, t_hm, nullptr, hm_ch, n_hm, s_reg, t_reg, nullptr, reg_ch, n_reg, clamp_lo, clamp_hi, H * W,
                     s_hm_st, t_hm_st, nullptr, s_reg_st, t_reg_st, nullptr);
  if (rc != UD_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(ud_div_up((long long)H * W, 256), B);
  UdProfScope prof("distill.k_resp", stream);
  k_resp<false><<<grid, 256, 0, stream>>>(a, mask, H * W, partial, nullptr, nullptr);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_distill_resp_bwd_strided(const float* const* s_hm, const int64_t* s_hm_st, const float* const* t_hm,
                                           const int64_t* t_hm_st, float* const* g_hm, const int64_t* g_hm_st, const int* hm_ch,
                                           int n_hm, const float* const* s_reg, const int64_t* s_reg_st,
                                           const float* const* t_reg, const int64_t* t_reg_st, float* const* g_reg,
                                           const int64_t* g_reg_st, const int* reg_ch, int n_reg, const float* mask, int B, int H,
                                           int W, float clamp_lo, float clamp_hi, const float* gscale_cls,
                                           const float* gscale_reg, ud_stream_t stream_) {
  if (!s_hm || !t_hm || !g_hm || !hm_ch || !s_reg || !t_reg || !g_reg || !reg_ch || !mask || !gscale_cls || !gscale_reg ||
      !s_hm_st || !t_hm_st || !g_hm_st || !s_reg_st || !t_reg_st || !g_reg_st)
    return UD_ERR_INVALID_ARG;
  RespArgs a;
  int rc = fill_resp(a, s_hm, t_hm, g_hm, hm_ch, n_hm, s_reg, t_reg, g_reg, reg_ch, n_reg, clamp_lo, clamp_hi, H * W, s_hm_st,
                     t_hm_st, g_hm_st, s_reg_st, t_reg_st, g_reg_st);
  if (rc != UD_OK) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  dim3 grid(ud_div_up((long long)H * W, 256), B);
  k_resp<true><<<grid, 256, 0, stream>>>(a, mask, H * W, nullptr, gscale_cls, gscale_reg);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
