// The 42 second convolutions of the CenterPoint head (SepHead: conv3x3 64->64, BN, ReLU, conv3x3 64->k with
// k <= 3; reference layers/head/det3d/center_head.py:311-362) in FP32 -- the reference's arithmetic.
//
// All SepHeads share their input, so the hidden tensor is a[B,H,W,G*64] (G = 42 stacks, 1.39 GB in fp32 at
// B = 4) and the second layer is a GROUPED 3x3 convolution with 64 inputs and k <= 4 outputs per group: 19
// GFLOP that the libraries run either as a block-diagonal dense conv (42x the FLOPs: 6.2 ms forward, 19 ms
// forward+backward) or as a 42-group conv (slower still).  It is HBM-bound by construction, so these are
// streaming VALU kernels (exact fp32 FMAs), one workgroup per (16-pixel-wide strip of up to 48 rows, group):
//   k_gtail_fwd    no LDS: lane = (pixel column of a 4-wide strip, channel quad), 3 x 3 register window walking down a strip
//   k_gtail_dgrad  da[q, c] = sum_{tap,k} dz[q - tap, k] * w[k, tap, c]: same lane map, dz window in registers, 16-byte stores
//   k_gtail_wgrad  dW[k, tap, c] = sum_q a[q, c] * dz[q - tap, k]: the forward's lane map and strip walk, 9 * KM 4-channel accumulators,
//                  a workgroup walks a slice of the strips; slices are summed in a fixed order (deterministic)
#include "ud_common.h"
#include "ud_prof.h"

namespace {

constexpr int kHC = 64;                       // channels per group
constexpr int kTW = 16, kTH = 8, kHW = kTW + 2;   // strip width, rows per unrolled block, halo width
constexpr int kKMax = 4;                      // outputs per group
constexpr int kStripMax = 48;                 // rows a workgroup of the strip-walking kernels covers

struct GTail {
  int B, H, W, G, KM, tiles_x, tiles_y;
};

// Sum over each ROW of 16 lanes with DPP adds (VALU rate; a shuffle butterfly goes through the LDS pipe): row_shr
// 1/2/4/8 with zero fill; lane 15 of every row holds its row's total.
__device__ __forceinline__ float dpp_row_sum_to_lane15(float v) {
#define UD_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true))
  UD_DPP_ADD(0x111);   // row_shr:1
  UD_DPP_ADD(0x112);   // row_shr:2
  UD_DPP_ADD(0x114);   // row_shr:4
  UD_DPP_ADD(0x118);   // row_shr:8
#undef UD_DPP_ADD
  return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  return fmaf(a.w, b.w, acc);
}

// z[pix, g*KM + k] = bias + sum_{tap, c} a[pix + tap - 1, g*64 + c] * w[g][k][tap][c]
// No LDS.  lane = (pixel column of a 4-wide strip, channel quad): a wave walks its strip down the 8 rows of the tile
// with a 3 x 3 window of 16-byte pieces in registers (3 new loads per pixel, the window's other 6 pieces are reused),
// the lane's 9 * KM weight quads stay in registers, and each partial sum is reduced over the 16 lanes of its DPP row.
// ~19 instructions per output pixel and wave.  (Earlier formulations, all 1.7-2.1 ms at B = 4: halo in LDS with a
// thread per pixel; lane = channel with nine row reads per pixel and shuffle reductions; lane = channel with a register
// halo and 64-lane DPP reductions -- instruction-bound: 140 instructions per pixel.)
__device__ __attribute__((aligned(16))) unsigned int g_zero_f4[4];
template <int KM, bool BN>
__global__ __launch_bounds__(256) void k_gtail_fwd(const float* __restrict__ a, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ z, GTail t,
                                                   int strip_rows, int strips, const float* __restrict__ bn_scale,
                                                   const float* __restrict__ bn_shift) {
  // groups vary fastest over the grid: the workgroups running together consume whole pixel rows
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // A workgroup walks a 16-pixel-wide strip of `strip_rows` rows (a multiple of kTH): the window keeps sliding across the
  // 8-row blocks, so the lane's 9 * KM weight quads (27 KB per wave, more than a block's input) and the two halo rows are
  // fetched once per strip instead of once per block.  With one 8-row block per workgroup (the first version) a workgroup
  // lived ~4 us, about half of it the latency of its weight and first-row loads, with only two workgroups per CU
  // (252 VGPRs) to hide it: 1.0 ms for the 1.39 GB tensor; fetching rows further ahead changed nothing.
  int bidx = blockIdx.y;
  const int tx = bidx % t.tiles_x;
  bidx /= t.tiles_x;
  const int strip = bidx % strips, b = bidx / strips;
  const int tx0 = tx * kTW, ty0 = strip * strip_rows;
  const int y_end = min(t.H, ty0 + strip_rows);
  const int Ct = t.G * kHC, Zt = t.G * KM;
  const int ps = lane >> 4, cq = lane & 15;
  float4 wr[KM][9];
#pragma unroll
  for (int k = 0; k < KM; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      wr[k][tap] = *reinterpret_cast<const float4*>(w + ((size_t)(g * KM + k) * 9 + tap) * kHC + 4 * cq);
  const int gx = tx0 + 4 * wave + ps;
  if (tx0 + 4 * wave >= t.W) return;
  const float* ab = a + (size_t)b * t.H * t.W * Ct + g * kHC + 4 * cq;
  // out-of-image pieces read a zero page: `ok ? *p : zero4` on float4 structs is compiled as a load through a select of
  // two addresses (one of them a scratch copy of zero4), i.e. flat loads
  const float* zero = reinterpret_cast<const float*>(g_zero_f4);
  // The three pieces (gx-1, gx, gx+1) of the next input row: pointers stepped by one image row per call.  (Rebuilding
  // ((yy * W + xx) * Ct) per piece cost ten quarter-rate 32 / 64-bit multiplies per row -- as many issue cycles as the
  // row's 71 FMA instructions.)
  const float* pp[3];
  bool xok[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int xx = gx - 1 + d;
    xok[d] = xx >= 0 && xx < t.W;
    pp[d] = ab + ((long long)(ty0 - 1) * t.W + (xok[d] ? xx : 0)) * Ct;
  }
  const size_t rstride = (size_t)t.W * Ct;
  int yy_next = ty0 - 1;
  // bn_scale != nullptr: the input is the first convolution's raw output; BatchNorm (folded scale / shift) + ReLU are applied to
  // every piece as it is loaded (padding stays zero) -- the hidden tensor is neither written normalised nor read back
  float4 bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (BN) {
    bsc = *reinterpret_cast<const float4*>(bn_scale + g * kHC + 4 * cq);
    bsh = *reinterpret_cast<const float4*>(bn_shift + g * kHC + 4 * cq);
  }
  auto load_next = [&](float4* r) {
    const bool yok = yy_next >= 0 && yy_next < t.H;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      r[d] = *reinterpret_cast<const float4*>((yok && xok[d]) ? pp[d] : zero);
      pp[d] += rstride;
    }
    ++yy_next;
  };
  // BN: applied when a row ENTERS the 3-row window (it was fetched PD steps earlier and stays in flight until then), once per
  // row; `yy` = the row's image row
  auto bn_row = [&](float4* r, int yy) {
    const bool yok = yy >= 0 && yy < t.H;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool ok = yok && xok[d];
      r[d].x = ok ? fmaxf(fmaf(r[d].x, bsc.x, bsh.x), 0.f) : 0.f;
      r[d].y = ok ? fmaxf(fmaf(r[d].y, bsc.y, bsh.y), 0.f) : 0.f;
      r[d].z = ok ? fmaxf(fmaf(r[d].z, bsc.z, bsh.z), 0.f) : 0.f;
      r[d].w = ok ? fmaxf(fmaf(r[d].w, bsc.w, bsh.w), 0.f) : 0.f;
    }
  };
  constexpr int PD = KM >= 4 ? 2 : 3;              // rows fetched ahead of their first use
  float4 rows[kTH + 2 + PD][3];                    // rows[r] = input row (block's first row) - 1 + r
#pragma unroll
  for (int r = 0; r < 2 + PD; ++r) load_next(rows[r]);
  if constexpr (BN) {
    bn_row(rows[0], ty0 - 1);
    bn_row(rows[1], ty0);
  }
  float bv[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) bv[k] = bias ? bias[g * KM + k] : 0.f;
  float* zo = z + ((size_t)(b * t.H + ty0) * t.W + gx) * Zt + g * KM;      // this lane's output pixel, stepped per row
  const size_t zstride = (size_t)t.W * Zt;
  for (int y0 = ty0; y0 < y_end; y0 += kTH) {
#pragma unroll
    for (int y = 0; y < kTH; ++y) {
      const int gy = y0 + y;
      load_next(rows[y + 2 + PD]);
      if constexpr (BN) bn_row(rows[y + 2], gy + 1);
      float acc[KM];
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        float s = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) s = dot4(rows[y + tap / 3][tap % 3], wr[k][tap], s);
        acc[k] = dpp_row_sum_to_lane15(s);
      }
      if (cq == 15 && gy < y_end && gx < t.W) {
#pragma unroll
        for (int k = 0; k < KM; ++k) zo[k] = acc[k] + bv[k];
      }
      zo += zstride;
    }
#pragma unroll
    for (int r = 0; r < 2 + PD; ++r)
#pragma unroll
      for (int d = 0; d < 3; ++d) rows[r][d] = rows[r + kTH][d];
  }
}

// da[q, g*64 + c] = sum_{tap, k} dz[q - (tap - 1), g*KM + k] * w[g][k][tap][c]
// MODE 0 stores da.  MODE 1 / 2 are the BatchNorm backward of the fused (relu(bn(y)) -> tail) block computed ON the data
// gradient instead of after it: da is recomputed in both passes and never stored (1.39 GB at B = 4, written once and read
// twice by the separate kernels).  MODE 1: partial[slice][c] = (sum dr, sum dr * (y - mean)) with dr = da where
// y * scale + shift > 0; MODE 2: dy = scale * dr + k2 * y + k0 (k0 / k2 from k_gtail_bn_final).  MODE 3 = MODE 2 + the per-channel
// sums of the dy values it stores (colsum[slice][c]; k_gtail_colsum_final adds the slices in order): the bias gradient of the
// convolution that produced y (center_head.py:339: bias=True in front of the BatchNorm), which otherwise is one more pass
// over the 1.39 GB gradient (ATen's reduce_kernel ran it on 11 workgroups: 2.1 ms per step).
struct GTailBn {
  const float *y, *scale, *shift, *mean, *k0, *k2;
  float* partial;
  float* colsum;
};

template <int KM, int MODE>
__global__ __launch_bounds__(256) void k_gtail_dgrad(const float* __restrict__ dz, const float* __restrict__ w,
                                                     float* __restrict__ da, GTail t, int strip_rows, int strips,
                                                     GTailBn bn) {
  // Lane map and strip walk of k_gtail_fwd / k_gtail_wgrad: lane = (pixel column of a 4-wide strip, channel quad); the
  // lane's 9 * KM weight quads stay in registers, the 3 x 3 window of dz values slides down the strip (dz halo of the strip
  // staged once in LDS), one 16-byte store per pixel and lane.  (First version: lane = channel, a pixel per wave step, nine
  // broadcast LDS reads + 36 FMAs + the pixel's address arithmetic for a 4-byte store: 530 us.)
  __shared__ float4 s_z[(kStripMax + 2) * kHW];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // groups fastest (see k_gtail_fwd)
  const int ps = lane >> 4, cq = lane & 15;
  int r = blockIdx.y;
  const int tx = r % t.tiles_x;
  r /= t.tiles_x;
  const int strip = r % strips, b = r / strips;
  const int tx0 = tx * kTW, ty0 = strip * strip_rows, y_end = min(t.H, ty0 + strip_rows);
  const int Zt = t.G * KM, Ct = t.G * kHC;
  const int nz = ((y_end - ty0 + kTH - 1) / kTH * kTH + 2) * kHW;
  for (int i = tid; i < nz; i += 256) {
    const int zr = i / kHW, zc = i - zr * kHW;
    const int gy = ty0 - 1 + zr, gx = tx0 - 1 + zc;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gy >= 0 && gy < t.H && gx >= 0 && gx < t.W) {
      const float* src = dz + ((size_t)(b * t.H + gy) * t.W + gx) * Zt + g * KM;
#pragma unroll
      for (int k = 0; k < KM; ++k) v[k] = src[k];
    }
    s_z[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
  float4 wr[KM][9];
#pragma unroll
  for (int k = 0; k < KM; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      wr[k][tap] = *reinterpret_cast<const float4*>(w + ((size_t)(g * KM + k) * 9 + tap) * kHC + 4 * cq);
  __syncthreads();
  const int px = 4 * wave + ps, gx = tx0 + px;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, bs, bt, bm, b0, b2;
  if (MODE != 0) {
    const int c0 = g * kHC + 4 * cq;
    bs = *reinterpret_cast<const float4*>(bn.scale + c0);
    bt = *reinterpret_cast<const float4*>(bn.shift + c0);
    if (MODE == 1) bm = *reinterpret_cast<const float4*>(bn.mean + c0);
    if (MODE >= 2) {
      b0 = *reinterpret_cast<const float4*>(bn.k0 + c0);
      b2 = *reinterpret_cast<const float4*>(bn.k2 + c0);
    }
  }
  if (tx0 + 4 * wave < t.W) {
  const size_t o0 = ((size_t)(b * t.H + ty0) * t.W + gx) * Ct + g * kHC + 4 * cq;
  float* po = da + o0;
  const size_t rstride = (size_t)t.W * Ct;
  // y rows are fetched kYD rows ahead of their use through a register ring (a load issued where its row is used exposes the
  // full HBM latency per row at two waves per SIMD: 2.7 TB/s); rows past the image are clamped, never used
  constexpr int kYD = 4;
  const float* ybase = MODE != 0 ? bn.y + ((size_t)b * t.H * t.W + min(gx, t.W - 1)) * Ct + g * kHC + 4 * cq : nullptr;
  float4 yq[kYD];
  if (MODE != 0) {
#pragma unroll
    for (int i = 0; i < kYD; ++i) yq[i] = *reinterpret_cast<const float4*>(ybase + (size_t)min(ty0 + i, t.H - 1) * rstride);
  }
  float4 zw[kTH + 2][3];                              // zw[r][d]: dz of output pixel (block row - 1 + r, px - 1 + d)
  const float4* zp = s_z + px;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int d = 0; d < 3; ++d) zw[i][d] = zp[i * kHW + d];
  zp += 2 * kHW;
  for (int y0 = ty0; y0 < y_end; y0 += kTH) {
#pragma unroll
    for (int y = 0; y < kTH; ++y) {
#pragma unroll
      for (int d = 0; d < 3; ++d) zw[y + 2][d] = zp[d];
      zp += kHW;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // the output pixel that read this input through tap (ty, tx) is (row - (ty - 1), column - (tx - 1))
        const float4 zv = zw[y + 2 - tap / 3][2 - tap % 3];
        const float zk[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
        for (int k = 0; k < KM; ++k) {
          o.x = fmaf(zk[k], wr[k][tap].x, o.x);
          o.y = fmaf(zk[k], wr[k][tap].y, o.y);
          o.z = fmaf(zk[k], wr[k][tap].z, o.z);
          o.w = fmaf(zk[k], wr[k][tap].w, o.w);
        }
      }
      if (y0 + y < y_end && gx < t.W) {
        if (MODE == 0) {
          *reinterpret_cast<float4*>(po) = o;
        } else {
          const float4 yv = yq[y % kYD];
          o.x = fmaf(yv.x, bs.x, bt.x) > 0.f ? o.x : 0.f;
          o.y = fmaf(yv.y, bs.y, bt.y) > 0.f ? o.y : 0.f;
          o.z = fmaf(yv.z, bs.z, bt.z) > 0.f ? o.z : 0.f;
          o.w = fmaf(yv.w, bs.w, bt.w) > 0.f ? o.w : 0.f;
          if (MODE == 1) {
            s1.x += o.x; s1.y += o.y; s1.z += o.z; s1.w += o.w;
            s2.x = fmaf(o.x, yv.x - bm.x, s2.x);
            s2.y = fmaf(o.y, yv.y - bm.y, s2.y);
            s2.z = fmaf(o.z, yv.z - bm.z, s2.z);
            s2.w = fmaf(o.w, yv.w - bm.w, s2.w);
          } else {
            float4 r;
            r.x = fmaf(bs.x, o.x, fmaf(b2.x, yv.x, b0.x));
            r.y = fmaf(bs.y, o.y, fmaf(b2.y, yv.y, b0.y));
            r.z = fmaf(bs.z, o.z, fmaf(b2.z, yv.z, b0.z));
            r.w = fmaf(bs.w, o.w, fmaf(b2.w, yv.w, b0.w));
            *reinterpret_cast<float4*>(po) = r;
            if (MODE == 3) { s1.x += r.x; s1.y += r.y; s1.z += r.z; s1.w += r.w; }
          }
        }
      }
      po += rstride;
      if (MODE != 0)
        yq[y % kYD] = *reinterpret_cast<const float4*>(ybase + (size_t)min(y0 + y + kYD, t.H - 1) * rstride);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int d = 0; d < 3; ++d) zw[i][d] = zw[i + kTH][d];
  }
  }
  if (MODE == 1) {
    // fixed-order sum over the workgroup's 16 pixel columns (4 per wave x 4 waves), one (sum, sum * centred) pair per channel
    __shared__ float4 red[2][16][16];
    red[0][4 * wave + ps][cq] = s1;
    red[1][4 * wave + ps][cq] = s2;
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float* rp = reinterpret_cast<const float*>(&red[which][0][0]) + c;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) a += rp[i * 64];
      bn.partial[((size_t)blockIdx.y * Ct + g * kHC + c) * 2 + which] = a;
    }
  }
  if (MODE == 3) {
    __shared__ float4 red3[16][16];
    red3[4 * wave + ps][cq] = s1;
    __syncthreads();
    if (tid < 64) {
      const float* rp = reinterpret_cast<const float*>(&red3[0][0]) + tid;
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) a += rp[i * 64];
      bn.colsum[(size_t)blockIdx.y * Ct + g * kHC + tid] = a;
    }
  }
}

// out[c] = sum over the slices of colsum[slice][c]: T threads per channel, four loads in flight per thread, fixed tree
template <int T>
__global__ __launch_bounds__(T) void k_gtail_colsum_final(const float* __restrict__ colsum, int slices, int C, float* __restrict__ out) {
  __shared__ float red[T / 64];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = tid;
  for (; s + 3 * T < slices; s += 4 * T) {
    const float v0 = colsum[(size_t)s * C + c], v1 = colsum[(size_t)(s + T) * C + c], v2 = colsum[(size_t)(s + 2 * T) * C + c],
                v3 = colsum[(size_t)(s + 3 * T) * C + c];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; s < slices; s += T) a0 += colsum[(size_t)s * C + c];
  float a = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (T > 64) {
    if (lane == 0) red[wv] = a;
    __syncthreads();
    if (tid == 0) {
      a = red[0];
#pragma unroll
      for (int w = 1; w < T / 64; ++w) a += red[w];
    }
  }
  if (tid == 0) out[c] = a;
}

// dgamma / dbeta and the two per-channel constants of the dy pass (the k_bn_bwd_final of bn_act.hip on this layout)
template <int T>
__global__ __launch_bounds__(T) void k_gtail_bn_final(const float* __restrict__ partial, int slices, int C, long long P,
                                                      const float* __restrict__ scale, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, float* __restrict__ k0, float* __restrict__ k2) {
  // T threads per channel, four independent row loads in flight per thread, fixed tree (see k_bn_bwd_final of bn_act.hip)
  __shared__ float red[T / 64][2];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto ld = [&](int s) { return *reinterpret_cast<const float2*>(partial + ((size_t)s * C + c) * 2); };
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  int s = tid;
  for (; s + 3 * T < slices; s += 4 * T) {
    const float2 v0 = ld(s), v1 = ld(s + T), v2 = ld(s + 2 * T), v3 = ld(s + 3 * T);
    a0 += v0.x; q0 += v0.y;
    a1 += v1.x; q1 += v1.y;
    a2 += v2.x; q2 += v2.y;
    a3 += v3.x; q3 += v3.y;
  }
  for (; s < slices; s += T) {
    const float2 v = ld(s);
    a0 += v.x; q0 += v.y;
  }
  float a = (a0 + a1) + (a2 + a3), q = (q0 + q1) + (q2 + q3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    q += __shfl_xor(q, o);
  }
  if (T > 64) {
    if (lane == 0) { red[wv][0] = a; red[wv][1] = q; }
    __syncthreads();
    if (tid == 0) {
      a = red[0][0], q = red[0][1];
#pragma unroll
      for (int w = 1; w < T / 64; ++w) { a += red[w][0]; q += red[w][1]; }
    }
  }
  if (tid != 0) return;
  const float is = invstd[c], dg = q * is, inv_p = 1.0f / (float)P;
  dbeta[c] = a;
  dgamma[c] = dg;
  const float kk2 = -scale[c] * (dg * inv_p) * is;
  k2[c] = kk2;
  k0[c] = -scale[c] * (a * inv_p) - kk2 * mean[c];
}

// partial[slice][g][k][tap][c] = sum over the slice's tiles of a[q, c] * dz[q - (tap - 1), k]
template <int KM, bool BN>
__global__ __launch_bounds__(256) void k_gtail_wgrad(const float* __restrict__ a, const float* __restrict__ dz,
                                                     float* __restrict__ partial, GTail t, int strip_rows, int strips,
                                                     int nstrips, int S, const float* __restrict__ bn_scale,
                                                     const float* __restrict__ bn_shift) {
  // Same lane map and strip walk as k_gtail_fwd: lane = (pixel column ps of a 4-wide strip, channel quad cq), the wave walks
  // down its strip with the 3 x 3 window of dz values (KM <= 4 floats per pixel, staged once per strip in LDS) in
  // registers; per row one 16-byte load of the hidden tensor and 9 * KM packed FMAs on 4-channel accumulators.  The first
  // version (lane = channel, one pixel per wave step: a 4-byte load, nine broadcast LDS reads, 36 FMAs and the pixel's
  // address arithmetic per step) issued ~3.4 x the instructions: 1.05 ms for the 1.39 GB tensor.
  __shared__ float4 s_z[(kStripMax + 2) * kHW];
  __shared__ float s_red[4][KM * 9][kHC];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ps = lane >> 4, cq = lane & 15;
  const int Zt = t.G * KM, Ct = t.G * kHC;
  const float* zero = reinterpret_cast<const float*>(g_zero_f4);
  float4 acc[KM][9];
#pragma unroll
  for (int k = 0; k < KM; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) acc[k][tap] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = make_float4(0.f, 0.f, 0.f, 0.f);   // see k_gtail_fwd
  if constexpr (BN) {
    bsc = *reinterpret_cast<const float4*>(bn_scale + g * kHC + 4 * cq);
    bsh = *reinterpret_cast<const float4*>(bn_shift + g * kHC + 4 * cq);
  }
  for (int st = blockIdx.y; st < nstrips; st += S) {
    int r = st;
    const int tx = r % t.tiles_x;
    r /= t.tiles_x;
    const int strip = r % strips, b = r / strips;
    const int tx0 = tx * kTW, ty0 = strip * strip_rows, y_end = min(t.H, ty0 + strip_rows);
    __syncthreads();                                    // the previous strip's window reads are done
    // whole 8-row blocks (+ halo): rows past the strip only meet zero inputs, but must hold finite values
    const int nz = ((y_end - ty0 + kTH - 1) / kTH * kTH + 2) * kHW;
    for (int i = tid; i < nz; i += 256) {
      const int zr = i / kHW, zc = i - zr * kHW;
      const int gy = ty0 - 1 + zr, gx = tx0 - 1 + zc;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (gy >= 0 && gy < t.H && gx >= 0 && gx < t.W) {
        const float* src = dz + ((size_t)(b * t.H + gy) * t.W + gx) * Zt + g * KM;
#pragma unroll
        for (int k = 0; k < KM; ++k) v[k] = src[k];
      }
      s_z[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    if (tx0 + 4 * wave >= t.W) continue;                // wave-uniform (the barriers above are reached by every wave)
    const int px = 4 * wave + ps, gx = tx0 + px;
    const bool xok = gx < t.W;
    const float* pa = a + ((size_t)(b * t.H + ty0) * t.W + (xok ? gx : 0)) * Ct + g * kHC + 4 * cq;
    const size_t rstride = (size_t)t.W * Ct;
    constexpr int PD = 3;                               // rows of the hidden tensor in flight ahead of their use
    float4 av[kTH + PD];
    int yn = ty0;
    auto load_next = [&](float4& dst) {
      dst = *reinterpret_cast<const float4*>((xok && yn < y_end) ? pa : zero);
      pa += rstride;
      ++yn;
    };
#pragma unroll
    for (int i = 0; i < PD; ++i) load_next(av[i]);
    float4 zw[kTH + 2][3];                              // zw[r][d]: dz of output pixel (block row - 1 + r, px - 1 + d)
    const float4* zp = s_z + px;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int d = 0; d < 3; ++d) zw[i][d] = zp[i * kHW + d];
    zp += 2 * kHW;
    for (int y0 = ty0; y0 < y_end; y0 += kTH) {
#pragma unroll
      for (int y = 0; y < kTH; ++y) {
        load_next(av[y + PD]);
#pragma unroll
        for (int d = 0; d < 3; ++d) zw[y + 2][d] = zp[d];
        zp += kHW;
        float4 x4 = av[y];                               // 0 outside the image / strip: contributes nothing
        if constexpr (BN) {                              // BatchNorm + ReLU at the row's (single) use: the fetch stays PD rows ahead
          const bool ok = xok && y0 + y < y_end;
          x4.x = ok ? fmaxf(fmaf(x4.x, bsc.x, bsh.x), 0.f) : 0.f;
          x4.y = ok ? fmaxf(fmaf(x4.y, bsc.y, bsh.y), 0.f) : 0.f;
          x4.z = ok ? fmaxf(fmaf(x4.z, bsc.z, bsh.z), 0.f) : 0.f;
          x4.w = ok ? fmaxf(fmaf(x4.w, bsc.w, bsh.w), 0.f) : 0.f;
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          // the output pixel that read this input through tap (ty, tx) is (row - (ty - 1), column - (tx - 1))
          const float4 zv = zw[y + 2 - tap / 3][2 - tap % 3];
          const float zk[4] = {zv.x, zv.y, zv.z, zv.w};
#pragma unroll
          for (int k = 0; k < KM; ++k) {
            acc[k][tap].x = fmaf(x4.x, zk[k], acc[k][tap].x);
            acc[k][tap].y = fmaf(x4.y, zk[k], acc[k][tap].y);
            acc[k][tap].z = fmaf(x4.z, zk[k], acc[k][tap].z);
            acc[k][tap].w = fmaf(x4.w, zk[k], acc[k][tap].w);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < PD; ++i) av[i] = av[i + kTH];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int d = 0; d < 3; ++d) zw[i][d] = zw[i + kTH][d];
    }
  }
  // sum over the four pixel columns of the wave (fixed order), then over the four waves
#pragma unroll
  for (int k = 0; k < KM; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float v[4] = {acc[k][tap].x, acc[k][tap].y, acc[k][tap].z, acc[k][tap].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += __shfl_xor(v[e], 16);
        v[e] += __shfl_xor(v[e], 32);
      }
      if (ps == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s_red[wave][k * 9 + tap][4 * cq + e] = v[e];
      }
    }
  __syncthreads();
  float* out = partial + ((size_t)blockIdx.y * t.G + g) * KM * 9 * kHC;
  for (int i = tid; i < KM * 9 * kHC; i += 256) {
    const int kt = i / kHC, c = i - kt * kHC;
    out[i] = ((s_red[0][kt][c] + s_red[1][kt][c]) + s_red[2][kt][c]) + s_red[3][kt][c];
  }
}

__global__ void k_gtail_wsum(const float* __restrict__ partial, int S, long long n, float* __restrict__ dw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += partial[(size_t)k * n + i];
  dw[i] = s;
}

bool gtail_ok(int B, int H, int W, int G, int KM) {
  return B > 0 && H > 0 && W > 0 && G > 0 && KM >= 1 && KM <= kKMax &&
         (long long)B * ud_div_up(W, kTW) * ud_div_up(H, kTH) <= 65535;     // tiles ride on grid.y
}
int gtail_slices(int ntiles) { return ntiles < 32 ? ntiles : 32; }

}  // namespace

static int gtail_fwd_impl(const float* a, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                          float* z, int B, int H, int W, int G, int KM, ud_stream_t stream_) {
  if (!a || !w || !z || !gtail_ok(B, H, W, G, KM) || (bn_scale == nullptr) != (bn_shift == nullptr)) return UD_ERR_INVALID_ARG;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("head_tail.k_gtail_fwd", stream);
  // strips of up to 48 rows, shorter when that leaves fewer than ~8 workgroups per CU
  int strip_rows = kStripMax;
  while (strip_rows > kTH && (long long)G * B * t.tiles_x * ud_div_up(H, strip_rows) < 2048) strip_rows -= kTH;
  const int strips = ud_div_up(H, strip_rows);
  const dim3 grid(G, B * t.tiles_x * strips);
  switch (KM) {
    case 1: if (bn_scale) k_gtail_fwd<1, true><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); else k_gtail_fwd<1, false><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); break;
    case 2: if (bn_scale) k_gtail_fwd<2, true><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); else k_gtail_fwd<2, false><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); break;
    case 3: if (bn_scale) k_gtail_fwd<3, true><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); else k_gtail_fwd<3, false><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); break;
    default: if (bn_scale) k_gtail_fwd<4, true><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); else k_gtail_fwd<4, false><<<grid, 256, 0, stream>>>(a, w, bias, z, t, strip_rows, strips, bn_scale, bn_shift); break;
  }
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_head_tail_f32_fwd(const float* a, const float* w, const float* bias, float* z, int B, int H,
                                    int W, int G, int KM, ud_stream_t stream_) {
  return gtail_fwd_impl(a, nullptr, nullptr, w, bias, z, B, H, W, G, KM, stream_);
}

// a = the first convolution's raw output; relu(a * scale + shift) (training- or eval-mode BatchNorm folded per channel) is applied
// as the pieces are loaded
extern "C" int ud_head_tail_f32_bn_fwd(const float* a, const float* bn_scale, const float* bn_shift, const float* w,
                                       const float* bias, float* z, int B, int H, int W, int G, int KM, ud_stream_t stream_) {
  if (!bn_scale || !bn_shift) return UD_ERR_INVALID_ARG;
  return gtail_fwd_impl(a, bn_scale, bn_shift, w, bias, z, B, H, W, G, KM, stream_);
}

extern "C" int ud_head_tail_f32_dgrad(const float* dz, const float* w, float* da, int B, int H, int W, int G,
                                      int KM, ud_stream_t stream_) {
  if (!dz || !w || !da || !gtail_ok(B, H, W, G, KM)) return UD_ERR_INVALID_ARG;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("head_tail.k_gtail_dgrad", stream);
  int strip_rows = kStripMax;
  while (strip_rows > kTH && (long long)G * B * t.tiles_x * ud_div_up(H, strip_rows) < 2048) strip_rows -= kTH;
  const int strips = ud_div_up(H, strip_rows);
  const dim3 grid(G, B * t.tiles_x * strips);
  const GTailBn none{};
  switch (KM) {
    case 1: k_gtail_dgrad<1, 0><<<grid, 256, 0, stream>>>(dz, w, da, t, strip_rows, strips, none); break;
    case 2: k_gtail_dgrad<2, 0><<<grid, 256, 0, stream>>>(dz, w, da, t, strip_rows, strips, none); break;
    case 3: k_gtail_dgrad<3, 0><<<grid, 256, 0, stream>>>(dz, w, da, t, strip_rows, strips, none); break;
    default: k_gtail_dgrad<4, 0><<<grid, 256, 0, stream>>>(dz, w, da, t, strip_rows, strips, none); break;
  }
  UD_LAUNCH_CHECK();
  return UD_OK;
}

static void gtail_bn_grid(const GTail& t, int* strip_rows, int* strips) {
  int sr = kStripMax;
  while (sr > kTH && (long long)t.G * t.B * t.tiles_x * ud_div_up(t.H, sr) < 2048) sr -= kTH;
  *strip_rows = sr;
  *strips = ud_div_up(t.H, sr);
}

extern "C" size_t ud_head_tail_f32_bn_bwd_workspace_bytes(int B, int H, int W, int G, int KM) {
  if (!gtail_ok(B, H, W, G, KM)) return 0;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  int strip_rows, strips;
  gtail_bn_grid(t, &strip_rows, &strips);
  const size_t C = (size_t)G * kHC;
  return ud_align_up(((size_t)B * t.tiles_x * strips * C * 3 + 2 * C) * sizeof(float));
}

extern "C" int ud_head_tail_f32_bn_bwd(const float* dz, const float* w, const float* y, const float* bn_scale,
                                       const float* bn_shift, const float* mean, const float* invstd, float* dy,
                                       float* dgamma, float* dbeta, float* dy_colsum, int B, int H, int W, int G, int KM,
                                       void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!dz || !w || !y || !bn_scale || !bn_shift || !mean || !invstd || !dy || !dgamma || !dbeta || !gtail_ok(B, H, W, G, KM))
    return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_head_tail_f32_bn_bwd_workspace_bytes(B, H, W, G, KM)) return UD_ERR_WORKSPACE;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  hipStream_t stream = (hipStream_t)stream_;
  int strip_rows, strips;
  gtail_bn_grid(t, &strip_rows, &strips);
  const int slices = B * t.tiles_x * strips, C = G * kHC;
  const dim3 grid(G, slices);
  float* partial = (float*)workspace;
  float* k0 = partial + (size_t)slices * C * 2;
  float* k2 = k0 + C;
  float* colsum = k2 + C;
  GTailBn bn{y, bn_scale, bn_shift, mean, k0, k2, partial, colsum};
  {
    UdProfScope prof("head_tail.k_gtail_bn_bwd_reduce", stream);
    switch (KM) {
      case 1: k_gtail_dgrad<1, 1><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
      case 2: k_gtail_dgrad<2, 1><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
      case 3: k_gtail_dgrad<3, 1><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
      default: k_gtail_dgrad<4, 1><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
    }
    UD_LAUNCH_CHECK();
    if (slices > 128)
      k_gtail_bn_final<256><<<C, 256, 0, stream>>>(partial, slices, C, (long long)B * H * W, bn_scale, mean, invstd, dgamma, dbeta,
                                                   k0, k2);
    else
      k_gtail_bn_final<64><<<C, 64, 0, stream>>>(partial, slices, C, (long long)B * H * W, bn_scale, mean, invstd, dgamma, dbeta,
                                                 k0, k2);
    UD_LAUNCH_CHECK();
  }
  UdProfScope prof("head_tail.k_gtail_bn_bwd_dx", stream);
  if (dy_colsum) {
    switch (KM) {
      case 1: k_gtail_dgrad<1, 3><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
      case 2: k_gtail_dgrad<2, 3><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
      case 3: k_gtail_dgrad<3, 3><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
      default: k_gtail_dgrad<4, 3><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
    }
    UD_LAUNCH_CHECK();
    if (slices > 128) k_gtail_colsum_final<256><<<C, 256, 0, stream>>>(colsum, slices, C, dy_colsum);
    else k_gtail_colsum_final<64><<<C, 64, 0, stream>>>(colsum, slices, C, dy_colsum);
    UD_LAUNCH_CHECK();
    return UD_OK;
  }
  switch (KM) {
    case 1: k_gtail_dgrad<1, 2><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
    case 2: k_gtail_dgrad<2, 2><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
    case 3: k_gtail_dgrad<3, 2><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
    default: k_gtail_dgrad<4, 2><<<grid, 256, 0, stream>>>(dz, w, dy, t, strip_rows, strips, bn); break;
  }
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" size_t ud_head_tail_f32_wgrad_workspace_bytes(int B, int H, int W, int G, int KM) {
  if (!gtail_ok(B, H, W, G, KM)) return 0;
  const int ntiles = B * ud_div_up(W, kTW) * ud_div_up(H, kTH);
  return ud_align_up((size_t)gtail_slices(ntiles) * G * KM * 9 * kHC * sizeof(float));
}

static int gtail_wgrad_impl(const float* a, const float* bn_scale, const float* bn_shift, const float* dz, float* dw, int B,
                            int H, int W, int G, int KM, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!a || !dz || !dw || !gtail_ok(B, H, W, G, KM) || (bn_scale == nullptr) != (bn_shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_head_tail_f32_wgrad_workspace_bytes(B, H, W, G, KM)) return UD_ERR_WORKSPACE;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  hipStream_t stream = (hipStream_t)stream_;
  int strip_rows = kStripMax;
  while (strip_rows > kTH && B * t.tiles_x * ud_div_up(H, strip_rows) < 32) strip_rows -= kTH;
  const int strips = ud_div_up(H, strip_rows), nstrips = B * t.tiles_x * strips;
  const int S = gtail_slices(nstrips);          // <= the slice count the workspace was sized for (strips <= tiles)
  float* partial = reinterpret_cast<float*>(workspace);
  UdProfScope prof("head_tail.k_gtail_wgrad", stream);
  const dim3 grid(G, S);
  switch (KM) {
    case 1: if (bn_scale) k_gtail_wgrad<1, true><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); else k_gtail_wgrad<1, false><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); break;
    case 2: if (bn_scale) k_gtail_wgrad<2, true><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); else k_gtail_wgrad<2, false><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); break;
    case 3: if (bn_scale) k_gtail_wgrad<3, true><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); else k_gtail_wgrad<3, false><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); break;
    default: if (bn_scale) k_gtail_wgrad<4, true><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); else k_gtail_wgrad<4, false><<<grid, 256, 0, stream>>>(a, dz, partial, t, strip_rows, strips, nstrips, S, bn_scale, bn_shift); break;
  }
  UD_LAUNCH_CHECK();
  const long long n = (long long)G * KM * 9 * kHC;
  k_gtail_wsum<<<ud_div_up(n, 256), 256, 0, stream>>>(partial, S, n, dw);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_head_tail_f32_wgrad(const float* a, const float* dz, float* dw, int B, int H, int W, int G,
                                      int KM, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  return gtail_wgrad_impl(a, nullptr, nullptr, dz, dw, B, H, W, G, KM, workspace, workspace_bytes, stream_);
}

extern "C" int ud_head_tail_f32_bn_wgrad(const float* a, const float* bn_scale, const float* bn_shift, const float* dz,
                                         float* dw, int B, int H, int W, int G, int KM, void* workspace,
                                         size_t workspace_bytes, ud_stream_t stream_) {
  if (!bn_scale || !bn_shift) return UD_ERR_INVALID_ARG;
  return gtail_wgrad_impl(a, bn_scale, bn_shift, dz, dw, B, H, W, G, KM, workspace, workspace_bytes, stream_);
}
