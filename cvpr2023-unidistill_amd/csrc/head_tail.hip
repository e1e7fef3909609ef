// Detection-head tail: BatchNorm -> ReLU -> per-head 3x3 conv (64 -> k<=3) for all G packed heads.
//
// Reference: SepHead (unidistill/layers/head/det3d/center_head.py:311-362) builds, per task and per
// head, Conv3x3(64->64) + BN + ReLU + Conv3x3(64->k).  After packing the G = tasks x heads first
// convs into one conv (layers/center_head.py: PackedSepHeads) the hidden tensor y is
// [B, H, W, G*64] bf16 -- 697 MB at B=4 on the 180x180 nuScenes BEV grid.  The library path used to
// run BN, ReLU and a block-diagonal dense conv over it (3 r/w passes + 42x redundant MFMA work:
// 2.1 ms fwd + 2.9 ms bwd for the conv alone).  Here the tail is HBM-bound by construction:
//
//   fwd   : one read of y; BN scale/shift + ReLU are applied while the tile is staged in LDS;
//           the 64->k conv is 18 v_mfma_f32_16x16x32_bf16 per 16 pixels (N padded 3 -> 16).
//   stats : per-channel batch mean / variance of y (training-mode BN), pivoted sums, two-stage
//           ordered reduction (deterministic).
//   bwd   : (1) dW2 from the same staged ReLU(BN(y)) tile; (2) da = conv^T(dz) recomputed on the
//           fly (K = 27 (j,tap) pairs -> one MFMA per 16 channels x 16 pixels), masked by ReLU and
//           reduced to dgamma/dbeta; (3) the same recomputation fused with the BN backward formula
//           writes dy.  The 697 MB gradient of the hidden tensor is written once and never re-read.
//
// Layouts: y, dy  [B][H][W][G*64] bf16 (channels-last);  z, dz  [B][G*kmax][H][W] fp32 (planar);
//          w2, dw2 [G][kmax][9][64] fp32 (tap = ky*3+kx);  per-channel vectors fp32 [G*64].
#include "ud_common.h"
#include "ud_prof.h"

namespace {

constexpr int kHC = 64;              // hidden channels per head (head_conv)
constexpr int kT = 16;               // spatial tile edge (pixels)
constexpr int kHalo = kT + 2;
constexpr int kQ = kHalo * kHalo;    // 324 staged pixels
constexpr int kPix = kHC + 8;        // bf16 elements per staged pixel: 144 B rows keep b128 reads spread
constexpr int kMaxOut = 3;           // outputs per head handled by the MFMA packing (27 <= 32)
constexpr int kA2 = 40;              // bf16 elements per pixel of the (j,tap) image (32 + pad)
constexpr int kWgradSlices = 12;   // 42 heads x 12 = 504 persistent blocks <= 2 per CU
constexpr int kBwdSlices = 96;
constexpr int kStatSlices = 1024;   // upper bound; the launch picks ~2048 / G pixel slices

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TailGeom {
  int B, H, W, G, kmax, C, tiles_x, tiles_y;
  __host__ __device__ int tiles() const { return tiles_x * tiles_y; }
};

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ unsigned pack2(float a, float b) { return ud_pack_bf16x2(a, b); }
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  union { unsigned u[4]; bf16x8 h; } r;
  r.u[0] = pack2(v[0], v[1]); r.u[1] = pack2(v[2], v[3]);
  r.u[2] = pack2(v[4], v[5]); r.u[3] = pack2(v[6], v[7]);
  return r.h;
}

// Stage ReLU(y*scale+shift) of one 18x18 halo tile of head g into LDS as bf16: [q][kPix] rows
// (TRANSPOSED = false, A operand of the forward conv) or [c][kRS] columns (TRANSPOSED = true, the
// pixel-contiguous operand of the weight-gradient GEMM).  All global loads are issued before the
// first use so a block pays one HBM latency per tile, not eleven.  Pixels outside the image are the
// conv's zero padding (applied AFTER BN+ReLU, as in the reference).
constexpr int kStageIters = (kQ * 8 + 255) / 256;   // 11
constexpr int kRS = 360;                            // bf16 elements per row of the transposed images

struct TileLoads {
  uint4 raw[kStageIters];
  unsigned inside;
};

__device__ __forceinline__ void issue_tile_loads(const unsigned short* __restrict__ y,
                                                 const TailGeom& gm, int b, int g, int ty0, int tx0,
                                                 TileLoads& ld) {
  const int tid = threadIdx.x, chunk = tid & 7;
  ld.inside = 0u;
#pragma unroll
  for (int it = 0; it < kStageIters; ++it) {
    const int q = (tid + 256 * it) >> 3;
    const int qy = q / kHalo, qx = q - qy * kHalo;
    const int gy = ty0 + qy - 1, gx = tx0 + qx - 1;
    ld.raw[it] = make_uint4(0u, 0u, 0u, 0u);
    if (q < kQ && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W) {
      ld.inside |= 1u << it;
      ld.raw[it] = *reinterpret_cast<const uint4*>(
          y + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.C + g * kHC + chunk * 8);
    }
  }
}

template <bool TRANSPOSED>
__device__ __forceinline__ void commit_tile(const TileLoads& ld, const float* s, const float* t,
                                            unsigned short* img) {
  const int tid = threadIdx.x, chunk = tid & 7;
#pragma unroll
  for (int it = 0; it < kStageIters; ++it) {
    const int q = (tid + 256 * it) >> 3;
    if (q >= kQ) break;
    const unsigned w[4] = {ld.raw[it].x, ld.raw[it].y, ld.raw[it].z, ld.raw[it].w};
    unsigned r[4] = {0u, 0u, 0u, 0u};
    if ((ld.inside >> it) & 1u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = fmaxf(fmaf(bf_lo(w[i]), s[2 * i], t[2 * i]), 0.f);
        const float hi = fmaxf(fmaf(bf_hi(w[i]), s[2 * i + 1], t[2 * i + 1]), 0.f);
        r[i] = pack2(lo, hi);
      }
    }
    if (TRANSPOSED) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        img[(chunk * 8 + 2 * i) * kRS + q] = (unsigned short)(r[i] & 0xFFFFu);
        img[(chunk * 8 + 2 * i + 1) * kRS + q] = (unsigned short)(r[i] >> 16);
      }
    } else {
      *reinterpret_cast<uint4*>(img + q * kPix + chunk * 8) = make_uint4(r[0], r[1], r[2], r[3]);
    }
  }
}

__device__ __forceinline__ void load_chunk_constants(const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int g,
                                                     float* s, float* t) {
  const int chunk = threadIdx.x & 7;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s[e] = scale[g * kHC + chunk * 8 + e];
    t[e] = shift[g * kHC + chunk * 8 + e];
  }
}

template <bool TRANSPOSED>
__device__ __forceinline__ void stage_activation_tile(const unsigned short* __restrict__ y,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift,
                                                      const TailGeom& gm, int b, int g, int ty0,
                                                      int tx0, unsigned short* img) {
  TileLoads ld;
  issue_tile_loads(y, gm, b, g, ty0, tx0, ld);
  float s[8], t[8];
  load_chunk_constants(scale, shift, g, s, t);
  commit_tile<TRANSPOSED>(ld, s, t, img);
}

// ---- forward -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tail_fwd(const unsigned short* __restrict__ y,
                                                  const float* __restrict__ scale,
                                                  const float* __restrict__ shift,
                                                  const float* __restrict__ w2,
                                                  const float* __restrict__ b2,
                                                  float* __restrict__ z, TailGeom gm) {
  __shared__ __attribute__((aligned(16))) unsigned short img[kQ * kPix];
  __shared__ __attribute__((aligned(16))) unsigned short wl[kMaxOut * 9 * kHC];
  const int tile = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int ty0 = (tile / gm.tiles_x) * kT, tx0 = (tile % gm.tiles_x) * kT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, kg = lane >> 4;
  stage_activation_tile<false>(y, scale, shift, gm, b, g, ty0, tx0, img);
  // weights of the (<= 3) real outputs as bf16 in LDS; MFMA columns n >= kmax multiply zeros
  for (int idx = threadIdx.x; idx < gm.kmax * 9 * kHC / 2; idx += 256) {
    const float2 v = *reinterpret_cast<const float2*>(w2 + (size_t)g * gm.kmax * 9 * kHC + 2 * idx);
    reinterpret_cast<unsigned*>(wl)[idx] = pack2(v.x, v.y);
  }
  __syncthreads();
  const float bias = (n < gm.kmax) ? b2[g * gm.kmax + n] : 0.f;
  const int Cz = gm.G * gm.kmax;
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int ly = wave * 4 + rr;
    if (ty0 + ly >= gm.H) break;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int q = (ly + tap / 3) * kHalo + (n + tap % 3);   // (ly + 1 + dy, n + 1 + dx)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(img + q * kPix + 32 * h + 8 * kg);
        bf16x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < gm.kmax) w = *reinterpret_cast<const bf16x8*>(wl + (n * 9 + tap) * kHC + 32 * h + 8 * kg);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, w, acc, 0, 0, 0);
      }
    }
    if (n < gm.kmax) {
      const int gy = ty0 + ly, gx = tx0 + 4 * kg;
      float* dst = z + ((size_t)(b * Cz + g * gm.kmax + n) * gm.H + gy) * gm.W + gx;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (gx + r < gm.W) dst[r] = acc[r] + bias;
    }
  }
}

// ---- batch statistics ----------------------------------------------------------------------------
// partial[slice][C][2] = sum(x - pivot), sum((x - pivot)^2) with pivot = y[pixel 0][c].
__global__ __launch_bounds__(256) void k_stats_partial(const unsigned short* __restrict__ y,
                                                       long long P, int C, float* __restrict__ partial) {
  __shared__ float red[32][kHC + 1][2];
  const int g = blockIdx.y, tid = threadIdx.x, chunk = tid & 7, pl = tid >> 3;
  float piv[8], s1[8], s2[8];
  {
    const uint4 raw = *reinterpret_cast<const uint4*>(y + g * kHC + chunk * 8);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { piv[2 * i] = bf_lo(w[i]); piv[2 * i + 1] = bf_hi(w[i]); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  for (long long p = (long long)blockIdx.x * 32 + pl; p < P; p += (long long)gridDim.x * 32) {
    const uint4 raw = *reinterpret_cast<const uint4*>(y + (size_t)p * C + g * kHC + chunk * 8);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = bf_lo(w[i]) - piv[2 * i], c = bf_hi(w[i]) - piv[2 * i + 1];
      s1[2 * i] += a; s2[2 * i] += a * a;
      s1[2 * i + 1] += c; s2[2 * i + 1] += c * c;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[pl][chunk * 8 + e][0] = s1[e]; red[pl][chunk * 8 + e][1] = s2[e]; }
  __syncthreads();
  if (tid < 2 * kHC) {
    const int c = tid >> 1, w = tid & 1;
    float a = 0.f;
    for (int i = 0; i < 32; ++i) a += red[i][c][w];
    partial[((size_t)blockIdx.x * C + g * kHC + c) * 2 + w] = a;
  }
}

__global__ void k_stats_final(const unsigned short* __restrict__ y, const float* __restrict__ partial,
                              int slices, long long P, int C, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, float* __restrict__ mean,
                              float* __restrict__ var, float* __restrict__ invstd,
                              float* __restrict__ scale, float* __restrict__ shift,
                              float* __restrict__ running_mean, float* __restrict__ running_var,
                              float momentum, long long* __restrict__ batches_tracked) {
  // one wave per channel: lanes stride over the slices, then a fixed-order butterfly (deterministic)
  const int c = blockIdx.x, lane = threadIdx.x;
  if (batches_tracked && c == 0 && lane == 0) *batches_tracked += 1;   // nn.BatchNorm's step counter
  double a = 0.0, q = 0.0;
  for (int s = lane; s < slices; s += 64) {
    a += partial[((size_t)s * C + c) * 2];
    q += partial[((size_t)s * C + c) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    q += __shfl_xor(q, o);
  }
  if (lane != 0) return;
  const double piv = __uint_as_float((unsigned)y[c] << 16);
  const double m = a / (double)P;
  double v = q / (double)P - m * m;
  if (v < 0.0) v = 0.0;
  const float mu = (float)(piv + m), is = (float)(1.0 / sqrt(v + (double)eps));
  mean[c] = mu;
  var[c] = (float)v;
  invstd[c] = is;
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - mu * sc;
  if (running_mean) {   // nn.BatchNorm2d bookkeeping: unbiased variance goes into the buffer
    const double unbiased = v * ((double)P / (double)(P > 1 ? P - 1 : 1));
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// ---- weight gradient of the per-head conv --------------------------------------------------------
// dW2[g][j][tap][c] = sum_p dz[j][p] * a[p + tap][c] as one GEMM per tile with the staged pixels q as
// the reduction index:  D[c][(j,tap)] += sum_q aT[c][q] * dT[(j,tap)][q],  dT[(j,tap)][q] = dz[j][q - tap]
// (zero unless q - tap is a pixel of this tile).  Both operands are pixel-contiguous bf16 images in
// LDS; wave w owns channels 16w..16w+15 and keeps its 16 x 32 accumulators across all tiles of the
// persistent block (slice, g).
constexpr int kKSteps = (kQ + 31) / 32;             // 11 (q padded to 352 <= kRS)

__global__ __launch_bounds__(256) void k_tail_wgrad(const unsigned short* __restrict__ y,
                                                    const float* __restrict__ scale,
                                                    const float* __restrict__ shift,
                                                    const float* __restrict__ dz,
                                                    float* __restrict__ partial, TailGeom gm) {
  __shared__ __attribute__((aligned(16))) unsigned short aT[kHC * kRS];
  __shared__ __attribute__((aligned(16))) unsigned short dT[32 * kRS];
  __shared__ float dzs[kMaxOut][kT * kT];
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  const int Cz = gm.G * gm.kmax;
  for (int idx = tid; idx < kHC * (kRS - kQ); idx += 256)          // K padding columns stay zero
    aT[(idx / (kRS - kQ)) * kRS + kQ + idx % (kRS - kQ)] = 0;
  for (int idx = tid; idx < 32 * kRS; idx += 256) dT[idx] = 0;      // padding columns and rows 27..31
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const int total = gm.B * gm.tiles();
  float cs[8], ct[8];
  load_chunk_constants(scale, shift, g, cs, ct);
  TileLoads ld;
  float dzr[kMaxOut];
  // loads of tile `it` (y pieces + this thread's dz pixel) -> registers
  auto issue = [&](int it) {
    const int b = it / gm.tiles(), tile = it - b * gm.tiles();
    const int ty0 = (tile / gm.tiles_x) * kT, tx0 = (tile % gm.tiles_x) * kT;
    issue_tile_loads(y, gm, b, g, ty0, tx0, ld);
    const int gy = ty0 + (tid >> 4), gx = tx0 + (tid & 15);
    const bool in = gy < gm.H && gx < gm.W;
#pragma unroll
    for (int j = 0; j < kMaxOut; ++j)
      dzr[j] = (in && j < gm.kmax)
                   ? dz[((size_t)(b * Cz + g * gm.kmax + j) * gm.H + gy) * gm.W + gx] : 0.f;
  };
  if ((int)blockIdx.x < total) issue(blockIdx.x);
  for (int it = blockIdx.x; it < total; it += gridDim.x) {
    __syncthreads();                                   // previous tile fully consumed
#pragma unroll
    for (int j = 0; j < kMaxOut; ++j) dzs[j][tid] = dzr[j];
    commit_tile<true>(ld, cs, ct, aT);
    if (it + (int)gridDim.x < total) issue(it + gridDim.x);   // next tile's HBM latency hides below
    __syncthreads();
    for (int idx = tid; idx < 27 * kQ; idx += 256) {
      const int k = idx / kQ, q = idx - k * kQ;
      const int j = k / 9, tap = k - j * 9;
      const int qy = q / kHalo, qx = q - qy * kHalo;
      const int py = qy - tap / 3, px = qx - tap % 3;  // tile pixel whose (tap)-neighbour is q
      float v = 0.f;
      if (py >= 0 && py < kT && px >= 0 && px < kT) v = dzs[j][py * kT + px];
      dT[k * kRS + q] = (unsigned short)(pack2(v, 0.f) & 0xFFFFu);
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(aT + (16 * wave + n) * kRS + 32 * ks + 8 * kg);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const bf16x8 bb = *reinterpret_cast<const bf16x8*>(dT + (16 * nt + n) * kRS + 32 * ks + 8 * kg);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[nt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int k = 16 * nt + n, j = k / 9, tap = k - j * 9;
    if (k < 27 && j < gm.kmax) {
      float* dst = partial + ((((size_t)blockIdx.x * gm.G + g) * gm.kmax + j) * 9 + tap) * kHC +
                   16 * wave + 4 * kg;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r] = acc[nt][r];
    }
  }
}

__global__ void k_sum_slices(const float* __restrict__ partial, int slices, size_t n,
                             float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = 0; s < slices; ++s) a += partial[(size_t)s * n + i];
  out[i] = a;
}

// ---- input gradient: da = conv^T(dz), ReLU mask, BatchNorm backward --------------------------------
// MFMA roles: rows = 16 hidden channels (weights, registers), cols = 16 pixels, K = 32 >= 27 (j,tap).
// Row m of MFMA tile t is channel 32*(t>>1) + 8*(m>>2) + 4*(t&1) + (m&3), so that lane group gq
// ends up with 8 consecutive channels per tile pair = one 16-byte piece of y / dy.
struct BwdConst { const float *scale, *shift, *mean, *k0, *k2; };

template <bool WRITE_DY>
__global__ __launch_bounds__(256) void k_tail_bwd(const unsigned short* __restrict__ y,
                                                  const float* __restrict__ dz,
                                                  const float* __restrict__ w2, BwdConst cst,
                                                  unsigned short* __restrict__ dy,
                                                  float* __restrict__ partial, TailGeom gm) {
  __shared__ float dzh[kMaxOut][kQ];
  __shared__ __attribute__((aligned(16))) unsigned short a2[kT * kT * kA2];
  __shared__ float red[4][4][16][2];
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, gq = lane >> 4;
  const int Cz = gm.G * gm.kmax;
  bf16x8 wfr[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ch = 32 * (t >> 1) + 8 * (n >> 2) + 4 * (t & 1) + (n & 3);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * gq + e, j = k / 9, tap = k - j * 9;
      v[e] = (k < 27 && j < gm.kmax) ? w2[((size_t)(g * gm.kmax + j) * 9 + tap) * kHC + ch] : 0.f;
    }
    wfr[t] = pack8(v);
  }
  // per-channel constants of this head, staged once: [scale, shift, mean|k0, k2][64]
  __shared__ __attribute__((aligned(16))) float cst_s[4][kHC];
  if (tid < kHC) {
    const int c = g * kHC + tid;
    cst_s[0][tid] = cst.scale[c];
    cst_s[1][tid] = cst.shift[c];
    cst_s[2][tid] = WRITE_DY ? cst.k0[c] : cst.mean[c];
    cst_s[3][tid] = WRITE_DY ? cst.k2[c] : 0.f;
  }
  // WRITE_DY keeps both channel halves of a pixel together (full 128-byte lines are written at once),
  // so its constants live in registers; the reduction variant walks one half at a time from LDS.
  // per-lane channel constants: channels 32*u + 8*gq + e  (index u*8+e)
  float rcs[16], rct[16], rc3[16], rc4[16];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = g * kHC + 32 * u + 8 * gq + e;
      rcs[u * 8 + e] = cst.scale[c];
      rct[u * 8 + e] = cst.shift[c];
      rc3[u * 8 + e] = WRITE_DY ? cst.k0[c] : cst.mean[c];
      rc4[u * 8 + e] = WRITE_DY ? cst.k2[c] : 0.f;
    }
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;

  const int total = gm.B * gm.tiles();
  for (int it = blockIdx.x; it < total; it += gridDim.x) {
    const int b = it / gm.tiles(), tile = it - b * gm.tiles();
    const int ty0 = (tile / gm.tiles_x) * kT, tx0 = (tile % gm.tiles_x) * kT;
    __syncthreads();
    for (int idx = tid; idx < kMaxOut * kQ; idx += 256) {
      const int j = idx / kQ, q = idx - j * kQ;
      const int qy = q / kHalo, qx = q - qy * kHalo;
      const int gy = ty0 + qy - 1, gx = tx0 + qx - 1;
      float v = 0.f;
      if (j < gm.kmax && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W)
        v = dz[((size_t)(b * Cz + g * gm.kmax + j) * gm.H + gy) * gm.W + gx];
      dzh[j][q] = v;
    }
    __syncthreads();
    {  // (j,tap) image: a2[p][j*9+tap] = dz[j][p - tap]
      const int ly = tid >> 4, lx = tid & 15;
      float v[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const int j = k / 9, tap = k - j * 9;
        v[k] = (k < 27) ? dzh[j][(ly + 2 - tap / 3) * kHalo + (lx + 2 - tap % 3)] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<bf16x8*>(a2 + tid * kA2 + 8 * c) = pack8(v + 8 * c);
    }
    __syncthreads();
    if constexpr (WRITE_DY) {
#pragma unroll 1
    for (int rr = 0; rr < 4; ++rr) {
      const int ly = wave * 4 + rr, gy = ty0 + ly, gx = tx0 + n;
      if (gy >= gm.H) break;
      const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(a2 + (ly * kT + n) * kA2 + 8 * gq);
      f32x4 d[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[t], bfrag, zero, 0, 0, 0);
      }
      if (gx < gm.W) {
        const size_t pix = ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.C + g * kHC;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint4 raw = *reinterpret_cast<const uint4*>(y + pix + 32 * u + 8 * gq);
          const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
          float out[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float yv = (e & 1) ? bf_hi(w[e >> 1]) : bf_lo(w[e >> 1]);
            const float da = d[2 * u + (e >> 2)][e & 3];
            const int i = u * 8 + e;
            const float dr = (fmaf(yv, rcs[i], rct[i]) > 0.f) ? da : 0.f;
            if (WRITE_DY) {
              out[e] = fmaf(rcs[i], dr, fmaf(rc4[i], yv, rc3[i]));
            } else {
              s1[i] += dr;
              s2[i] += dr * (yv - rc3[i]);
            }
          }
          if (WRITE_DY) {
            const uint4 o = make_uint4(pack2(out[0], out[1]), pack2(out[2], out[3]),
                                       pack2(out[4], out[5]), pack2(out[6], out[7]));
            *reinterpret_cast<uint4*>(dy + pix + 32 * u + 8 * gq) = o;
          }
        }
      }
    }
    } else {
#pragma unroll
    for (int u = 0; u < 2; ++u) {                      // channels 32u + 8gq .. +7 of this lane
      uint4 raw[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {                 // the wave's four rows in flight together
        const int gy = ty0 + wave * 4 + rr, gx = tx0 + n;
        raw[rr] = make_uint4(0u, 0u, 0u, 0u);
        if (gy < gm.H && gx < gm.W)
          raw[rr] = *reinterpret_cast<const uint4*>(
              y + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.C + g * kHC + 32 * u + 8 * gq);
      }
      float cs[8], ct[8], c3[8], c4[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c0 = 32 * u + 8 * gq + 4 * h;
        const float4 a0 = *reinterpret_cast<const float4*>(&cst_s[0][c0]);
        const float4 a1 = *reinterpret_cast<const float4*>(&cst_s[1][c0]);
        const float4 a2v = *reinterpret_cast<const float4*>(&cst_s[2][c0]);
        const float4 a3 = *reinterpret_cast<const float4*>(&cst_s[3][c0]);
        cs[4 * h] = a0.x; cs[4 * h + 1] = a0.y; cs[4 * h + 2] = a0.z; cs[4 * h + 3] = a0.w;
        ct[4 * h] = a1.x; ct[4 * h + 1] = a1.y; ct[4 * h + 2] = a1.z; ct[4 * h + 3] = a1.w;
        c3[4 * h] = a2v.x; c3[4 * h + 1] = a2v.y; c3[4 * h + 2] = a2v.z; c3[4 * h + 3] = a2v.w;
        c4[4 * h] = a3.x; c4[4 * h + 1] = a3.y; c4[4 * h + 2] = a3.z; c4[4 * h + 3] = a3.w;
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int ly = wave * 4 + rr, gy = ty0 + ly, gx = tx0 + n;
        if (gy >= gm.H) break;
        const bf16x8 bfrag = *reinterpret_cast<const bf16x8*>(a2 + (ly * kT + n) * kA2 + 8 * gq);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 d[2];
        d[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[2 * u], bfrag, zero, 0, 0, 0);
        d[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[2 * u + 1], bfrag, zero, 0, 0, 0);
        if (gx < gm.W) {
          const unsigned w[4] = {raw[rr].x, raw[rr].y, raw[rr].z, raw[rr].w};
          float out[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float yv = (e & 1) ? bf_hi(w[e >> 1]) : bf_lo(w[e >> 1]);
            const float da = d[e >> 2][e & 3];
            const float dr = (fmaf(yv, cs[e], ct[e]) > 0.f) ? da : 0.f;
            if (WRITE_DY) {
              out[e] = fmaf(cs[e], dr, fmaf(c4[e], yv, c3[e]));
            } else {
              s1[u * 8 + e] += dr;
              s2[u * 8 + e] += dr * (yv - c3[e]);
            }
          }
          if (WRITE_DY) {
            const uint4 o = make_uint4(pack2(out[0], out[1]), pack2(out[2], out[3]),
                                       pack2(out[4], out[5]), pack2(out[6], out[7]));
            *reinterpret_cast<uint4*>(dy + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.C + g * kHC +
                                      32 * u + 8 * gq) = o;
          }
        }
      }
    }
    }
  }
  if (!WRITE_DY) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s1[i] += __shfl_xor(s1[i], o);
        s2[i] += __shfl_xor(s2[i], o);
      }
      if (n == 0) { red[wave][gq][i][0] = s1[i]; red[wave][gq][i][1] = s2[i]; }
    }
    __syncthreads();
    if (tid < 2 * kHC) {
      const int c = tid >> 1, w = tid & 1;            // channel c = 32u + 8gq + e
      const int u = c >> 5, q4 = (c >> 3) & 3, e = c & 7;
      const float v = (red[0][q4][u * 8 + e][w] + red[1][q4][u * 8 + e][w]) +
                      (red[2][q4][u * 8 + e][w] + red[3][q4][u * 8 + e][w]);
      partial[((size_t)blockIdx.x * gm.C + g * kHC + c) * 2 + w] = v;
    }
  }
}

// dbeta, dgamma and the per-channel constants of dy = scale*dr + k2*y + k0.
__global__ void k_bn_bwd_final(const float* __restrict__ partial, int slices, int C, long long P,
                               const float* __restrict__ scale, const float* __restrict__ mean,
                               const float* __restrict__ invstd, float* __restrict__ dgamma,
                               float* __restrict__ dbeta, float* __restrict__ k0,
                               float* __restrict__ k2) {
  const int c = blockIdx.x, lane = threadIdx.x;      // one wave per channel, fixed-order reduction
  float a = 0.f, q = 0.f;
  for (int s = lane; s < slices; s += 64) {
    a += partial[((size_t)s * C + c) * 2];
    q += partial[((size_t)s * C + c) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    q += __shfl_xor(q, o);
  }
  if (lane != 0) return;
  const float is = invstd[c], dg = q * is;
  dbeta[c] = a;
  dgamma[c] = dg;
  const float inv_p = 1.0f / (float)P;
  const float kk2 = -scale[c] * (dg * inv_p) * is;
  k2[c] = kk2;
  k0[c] = -scale[c] * (a * inv_p) - kk2 * mean[c];
}

bool geom_ok(int B, int H, int W, int G, int kmax) {
  return B > 0 && H > 0 && W > 0 && G > 0 && kmax > 0 && kmax <= kMaxOut &&
         (long long)B * H * W * G * kHC < (1LL << 40);
}

TailGeom make_geom(int B, int H, int W, int G, int kmax) {
  TailGeom gm;
  gm.B = B; gm.H = H; gm.W = W; gm.G = G; gm.kmax = kmax; gm.C = G * kHC;
  gm.tiles_x = ud_div_up(W, kT);
  gm.tiles_y = ud_div_up(H, kT);
  return gm;
}

struct TailWs {
  float *stat_partial, *wgrad_partial, *bwd_partial, *k0, *k2;
};

size_t carve(UdArena& ar, int G, TailWs* w) {
  const int C = G * kHC, kmax = kMaxOut;
  w->stat_partial = ar.take<float>((size_t)kStatSlices * C * 2);
  w->wgrad_partial = ar.take<float>((size_t)kWgradSlices * G * kmax * 9 * kHC);
  w->bwd_partial = ar.take<float>((size_t)kBwdSlices * C * 2);
  w->k0 = ar.take<float>(C);
  w->k2 = ar.take<float>(C);
  return ar.used;
}

}  // namespace

extern "C" {

size_t ud_head_tail_workspace_bytes(int G) {
  if (G <= 0) return 0;
  UdArena ar(nullptr, 0);
  TailWs w;
  return carve(ar, G, &w);
}

int ud_head_tail_stats(const void* y, int B, int H, int W, int G, const float* gamma,
                       const float* beta, float eps, float* mean, float* var, float* invstd,
                       float* scale, float* shift, float* running_mean, float* running_var,
                       float momentum, long long* batches_tracked, void* workspace,
                       size_t workspace_bytes, ud_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!y || !gamma || !beta || !mean || !var || !invstd || !scale || !shift || !geom_ok(B, H, W, G, 1) ||
      ((running_mean == nullptr) != (running_var == nullptr)))
    return UD_ERR_INVALID_ARG;
  UdArena ar(workspace, workspace_bytes);
  TailWs w;
  carve(ar, G, &w);
  if (!ar.ok()) return UD_ERR_WORKSPACE;
  UdProfScope prof("head_tail.stats", stream);
  const long long P = (long long)B * H * W;
  const int C = G * kHC;
  long long slices = 2048 / G;
  if (slices > (P + 31) / 32) slices = (P + 31) / 32;
  if (slices < 1) slices = 1;
  if (slices > kStatSlices) slices = kStatSlices;
  k_stats_partial<<<dim3((int)slices, G), 256, 0, stream>>>((const unsigned short*)y, P, C, w.stat_partial);
  UD_LAUNCH_CHECK();
  k_stats_final<<<C, 64, 0, stream>>>((const unsigned short*)y, w.stat_partial,
                                                        (int)slices, P, C, gamma, beta, eps, mean, var,
                                                        invstd, scale, shift, running_mean,
                                                        running_var, momentum, batches_tracked);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

int ud_head_tail_fwd(const void* y, const float* scale, const float* shift, const float* w2,
                     const float* b2, float* z, int B, int H, int W, int G, int kmax,
                     ud_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!y || !scale || !shift || !w2 || !b2 || !z) return UD_ERR_INVALID_ARG;
  if (!geom_ok(B, H, W, G, kmax)) return (kmax > kMaxOut) ? UD_ERR_UNSUPPORTED : UD_ERR_INVALID_ARG;
  const TailGeom gm = make_geom(B, H, W, G, kmax);
  UdProfScope prof("head_tail.k_fwd", stream);
  k_tail_fwd<<<dim3(gm.tiles(), G, B), 256, 0, stream>>>((const unsigned short*)y, scale, shift, w2, b2, z, gm);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

int ud_head_tail_bwd(const void* y, const float* dz, const float* w2, const float* scale,
                     const float* shift, const float* mean, const float* invstd, void* dy,
                     float* dw2, float* dgamma, float* dbeta, int B, int H, int W, int G, int kmax,
                     void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!y || !dz || !w2 || !scale || !shift || !mean || !invstd || !dy || !dw2 || !dgamma || !dbeta)
    return UD_ERR_INVALID_ARG;
  if (!geom_ok(B, H, W, G, kmax)) return (kmax > kMaxOut) ? UD_ERR_UNSUPPORTED : UD_ERR_INVALID_ARG;
  UdArena ar(workspace, workspace_bytes);
  TailWs w;
  carve(ar, G, &w);
  if (!ar.ok()) return UD_ERR_WORKSPACE;
  const TailGeom gm = make_geom(B, H, W, G, kmax);
  const long long P = (long long)B * H * W;
  const int C = gm.C;
  {
    UdProfScope prof("head_tail.k_wgrad", stream);
    k_tail_wgrad<<<dim3(kWgradSlices, G), 256, 0, stream>>>((const unsigned short*)y, scale, shift, dz,
                                                            w.wgrad_partial, gm);
    UD_LAUNCH_CHECK();
    const size_t nW = (size_t)G * kmax * 9 * kHC;
    k_sum_slices<<<ud_div_up((long long)nW, 256), 256, 0, stream>>>(w.wgrad_partial, kWgradSlices, nW, dw2);
    UD_LAUNCH_CHECK();
  }
  BwdConst cst{scale, shift, mean, w.k0, w.k2};
  {
    UdProfScope prof("head_tail.k_bn_sums", stream);
    k_tail_bwd<false><<<dim3(kBwdSlices, G), 256, 0, stream>>>((const unsigned short*)y, dz, w2, cst,
                                                               nullptr, w.bwd_partial, gm);
    UD_LAUNCH_CHECK();
    k_bn_bwd_final<<<C, 64, 0, stream>>>(w.bwd_partial, kBwdSlices, C, P, scale, mean,
                                                           invstd, dgamma, dbeta, w.k0, w.k2);
    UD_LAUNCH_CHECK();
  }
  {
    UdProfScope prof("head_tail.k_dy", stream);
    k_tail_bwd<true><<<dim3(128, G), 256, 0, stream>>>((const unsigned short*)y, dz, w2, cst,
                                                                  (unsigned short*)dy, nullptr, gm);
    UD_LAUNCH_CHECK();
  }
  return UD_OK;
}

}  // extern "C"
