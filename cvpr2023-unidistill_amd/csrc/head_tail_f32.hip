// The 42 second convolutions of the CenterPoint head (SepHead: conv3x3 64->64, BN, ReLU, conv3x3 64->k with
// k <= 3; reference layers/head/det3d/center_head.py:311-362) in FP32 -- the reference's arithmetic.
//
// All SepHeads share their input, so the hidden tensor is a[B,H,W,G*64] (G = 42 stacks, 1.39 GB in fp32 at
// B = 4) and the second layer is a GROUPED 3x3 convolution with 64 inputs and k <= 4 outputs per group: 19
// GFLOP that the libraries run either as a block-diagonal dense conv (42x the FLOPs: 6.2 ms forward, 19 ms
// forward+backward) or as a 42-group conv (slower still).  It is HBM-bound by construction, so these are
// streaming VALU kernels (exact fp32 FMAs), one workgroup per (8 x 16 pixel tile, group):
//   k_gtail_fwd    the 10 x 18 x 64 halo of the group in LDS, thread = pixel: 9 x 64 x k FMAs
//   k_gtail_dgrad  da[q, c] = sum_{tap,k} dz[q - tap, k] * w[k, tap, c]: dz halo in LDS, lane = channel, the
//                  group's 27-36 weights of a lane's channel in registers, coalesced 256-byte row stores
//   k_gtail_wgrad  dW[k, tap, c] = sum_q a[q, c] * dz[q - tap, k]: lane = channel, 36 accumulators in registers,
//                  a workgroup walks a slice of the tiles; slices are summed in a fixed order (deterministic)
#include "ud_common.h"
#include "ud_prof.h"

namespace {

constexpr int kHC = 64;                       // channels per group
constexpr int kTW = 16, kTH = 8, kHW = kTW + 2, kHH = kTH + 2, kHQ = kHW * kHH;   // tile, halo
constexpr int kLD = 68;                       // LDS row stride of the staged halo (floats)
constexpr int kKMax = 4;                      // outputs per group

struct GTail {
  int B, H, W, G, KM, tiles_x, tiles_y;
};

__device__ __forceinline__ void tile_origin(const GTail& t, int tile, int& b, int& ty0, int& tx0) {
  const int per = t.tiles_x * t.tiles_y;
  b = tile / per;
  tile -= b * per;
  ty0 = (tile / t.tiles_x) * kTH;
  tx0 = (tile % t.tiles_x) * kTW;
}

// z[pix, g*KM + k] = bias + sum_{tap, c} a[pix + tap - 1, g*64 + c] * w[g][k][tap][c]
// 256 threads: two threads per output pixel, each over one half of the channels (combined through LDS);
// the halo is staged as independent 16-byte loads (23 in flight per thread) -- a row-per-iteration loop
// serialised one memory latency per halo row and made the kernel 15x slower.
__global__ __launch_bounds__(256) void k_gtail_fwd(const float* __restrict__ a, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ z, GTail t) {
  __shared__ __attribute__((aligned(16))) float s_a[kHQ * kLD];
  __shared__ __attribute__((aligned(16))) float s_w[kKMax * 9 * kHC];
  __shared__ float s_p[kTW * kTH][kKMax];
  const int g = blockIdx.y, tid = threadIdx.x;
  int b, ty0, tx0;
  tile_origin(t, blockIdx.x, b, ty0, tx0);
  const int Ct = t.G * kHC;
  for (int i = tid; i < t.KM * 9 * kHC; i += 256) s_w[i] = w[(size_t)g * t.KM * 9 * kHC + i];
  constexpr int kPieces = kHQ * (kHC / 4);                      // 16-byte pieces of the halo: 2880
  float4 v[(kPieces + 255) / 256];
#pragma unroll
  for (int j = 0; j < (kPieces + 255) / 256; ++j) {
    const int i = tid + j * 256;
    const int q = i >> 4, c4 = i & 15;
    const int gy = ty0 + q / kHW - 1, gx = tx0 + q % kHW - 1;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < kPieces && gy >= 0 && gy < t.H && gx >= 0 && gx < t.W)
      v[j] = *reinterpret_cast<const float4*>(a + ((size_t)(b * t.H + gy) * t.W + gx) * Ct + g * kHC + 4 * c4);
  }
#pragma unroll
  for (int j = 0; j < (kPieces + 255) / 256; ++j) {
    const int i = tid + j * 256;
    if (i < kPieces) *reinterpret_cast<float4*>(s_a + (i >> 4) * kLD + 4 * (i & 15)) = v[j];
  }
  __syncthreads();
  const int pix = tid & 127, half = tid >> 7;
  const int py = pix >> 4, px = pix & 15;
  float acc[kKMax] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const float* ar = s_a + ((py + tap / 3) * kHW + px + tap % 3) * kLD + 32 * half;
    const float* wr = s_w + tap * kHC + 32 * half;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 av = *reinterpret_cast<const float4*>(ar + 4 * c4);
#pragma unroll
      for (int k = 0; k < kKMax; ++k) {
        if (k < t.KM) {
          const float4 wv = *reinterpret_cast<const float4*>(wr + k * 9 * kHC + 4 * c4);
          acc[k] = fmaf(av.x, wv.x, acc[k]);
          acc[k] = fmaf(av.y, wv.y, acc[k]);
          acc[k] = fmaf(av.z, wv.z, acc[k]);
          acc[k] = fmaf(av.w, wv.w, acc[k]);
        }
      }
    }
  }
  if (half) {
#pragma unroll
    for (int k = 0; k < kKMax; ++k) s_p[pix][k] = acc[k];
  }
  __syncthreads();
  const int gy = ty0 + py, gx = tx0 + px;
  if (!half && gy < t.H && gx < t.W) {
    float* o = z + ((size_t)(b * t.H + gy) * t.W + gx) * (t.G * t.KM) + g * t.KM;
    for (int k = 0; k < t.KM; ++k) o[k] = (acc[k] + s_p[pix][k]) + (bias ? bias[g * t.KM + k] : 0.f);
  }
}

// da[q, g*64 + c] = sum_{tap, k} dz[q - (tap - 1), g*KM + k] * w[g][k][tap][c]
__global__ __launch_bounds__(256) void k_gtail_dgrad(const float* __restrict__ dz, const float* __restrict__ w,
                                                     float* __restrict__ da, GTail t) {
  __shared__ float s_z[kHQ][kKMax];
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int b, ty0, tx0;
  tile_origin(t, blockIdx.x, b, ty0, tx0);
  const int Zt = t.G * t.KM, Ct = t.G * kHC;
  for (int i = tid; i < kHQ * kKMax; i += 256) {
    const int q = i / kKMax, k = i - q * kKMax;
    const int gy = ty0 + q / kHW - 1, gx = tx0 + q % kHW - 1;
    float v = 0.f;
    if (k < t.KM && gy >= 0 && gy < t.H && gx >= 0 && gx < t.W)
      v = dz[((size_t)(b * t.H + gy) * t.W + gx) * Zt + g * t.KM + k];
    s_z[q][k] = v;
  }
  float wr[kKMax][9];                                           // this lane's channel: w[g][k][tap][lane]
#pragma unroll
  for (int k = 0; k < kKMax; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
      wr[k][tap] = (k < t.KM) ? w[((size_t)(g * t.KM + k) * 9 + tap) * kHC + lane] : 0.f;
  __syncthreads();
  for (int p = wave; p < kTW * kTH; p += 4) {
    const int py = p >> 4, px = p & 15;
    const int gy = ty0 + py, gx = tx0 + px;
    if (gy >= t.H || gx >= t.W) continue;
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // output pixel that used input q through this tap: q - (tap - 1)  -> halo index (py + 2 - ty, px + 2 - tx)
      const float4 zv = *reinterpret_cast<const float4*>(&s_z[(py + 2 - tap / 3) * kHW + px + 2 - tap % 3][0]);
      acc = fmaf(zv.x, wr[0][tap], acc);
      acc = fmaf(zv.y, wr[1][tap], acc);
      acc = fmaf(zv.z, wr[2][tap], acc);
      acc = fmaf(zv.w, wr[3][tap], acc);
    }
    da[((size_t)(b * t.H + gy) * t.W + gx) * Ct + g * kHC + lane] = acc;
  }
}

// partial[slice][g][k][tap][c] = sum over the slice's tiles of a[q, c] * dz[q - (tap - 1), k]
__global__ __launch_bounds__(256) void k_gtail_wgrad(const float* __restrict__ a, const float* __restrict__ dz,
                                                     float* __restrict__ partial, GTail t, int ntiles,
                                                     int tiles_per_slice) {
  __shared__ float s_z[kHQ][kKMax];
  __shared__ float s_red[4][kKMax * 9][kHC];
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Zt = t.G * t.KM, Ct = t.G * kHC;
  float acc[kKMax][9];
#pragma unroll
  for (int k = 0; k < kKMax; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) acc[k][tap] = 0.f;
  const int t0 = blockIdx.x * tiles_per_slice, t1 = min(ntiles, t0 + tiles_per_slice);
  for (int tile = t0; tile < t1; ++tile) {
    int b, ty0, tx0;
    tile_origin(t, tile, b, ty0, tx0);
    __syncthreads();
    for (int i = tid; i < kHQ * kKMax; i += 256) {
      const int q = i / kKMax, k = i - q * kKMax;
      const int gy = ty0 + q / kHW - 1, gx = tx0 + q % kHW - 1;
      float v = 0.f;
      if (k < t.KM && gy >= 0 && gy < t.H && gx >= 0 && gx < t.W)
        v = dz[((size_t)(b * t.H + gy) * t.W + gx) * Zt + g * t.KM + k];
      s_z[q][k] = v;
    }
    __syncthreads();
    float avs[kTW * kTH / 4];                                   // this wave's 32 pixels: all loads in flight at once
#pragma unroll
    for (int j = 0; j < kTW * kTH / 4; ++j) {
      const int p = wave + 4 * j, gy = ty0 + (p >> 4), gx = tx0 + (p & 15);
      avs[j] = (gy < t.H && gx < t.W) ? a[((size_t)(b * t.H + gy) * t.W + gx) * Ct + g * kHC + lane] : 0.f;
    }
#pragma unroll 4
    for (int j = 0; j < kTW * kTH / 4; ++j) {
      const int p = wave + 4 * j, py = p >> 4, px = p & 15;
      const float av = avs[j];                                   // 0 outside the image: contributes nothing
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float4 zv = *reinterpret_cast<const float4*>(&s_z[(py + 2 - tap / 3) * kHW + px + 2 - tap % 3][0]);
        acc[0][tap] = fmaf(av, zv.x, acc[0][tap]);
        acc[1][tap] = fmaf(av, zv.y, acc[1][tap]);
        acc[2][tap] = fmaf(av, zv.z, acc[2][tap]);
        acc[3][tap] = fmaf(av, zv.w, acc[3][tap]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kKMax; ++k)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) s_red[wave][k * 9 + tap][lane] = acc[k][tap];
  __syncthreads();
  float* out = partial + ((size_t)blockIdx.x * t.G + g) * t.KM * 9 * kHC;
  for (int i = tid; i < t.KM * 9 * kHC; i += 256) {
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
  return B > 0 && H > 0 && W > 0 && G > 0 && G <= 65535 && KM >= 1 && KM <= kKMax;
}
int gtail_slices(int ntiles) { return ntiles < 32 ? ntiles : 32; }

}  // namespace

extern "C" int ud_head_tail_f32_fwd(const float* a, const float* w, const float* bias, float* z, int B, int H,
                                    int W, int G, int KM, ud_stream_t stream_) {
  if (!a || !w || !z || !gtail_ok(B, H, W, G, KM)) return UD_ERR_INVALID_ARG;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("head_tail.k_gtail_fwd", stream);
  k_gtail_fwd<<<dim3(B * t.tiles_x * t.tiles_y, G), 256, 0, stream>>>(a, w, bias, z, t);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_head_tail_f32_dgrad(const float* dz, const float* w, float* da, int B, int H, int W, int G,
                                      int KM, ud_stream_t stream_) {
  if (!dz || !w || !da || !gtail_ok(B, H, W, G, KM)) return UD_ERR_INVALID_ARG;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("head_tail.k_gtail_dgrad", stream);
  k_gtail_dgrad<<<dim3(B * t.tiles_x * t.tiles_y, G), 256, 0, stream>>>(dz, w, da, t);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" size_t ud_head_tail_f32_wgrad_workspace_bytes(int B, int H, int W, int G, int KM) {
  if (!gtail_ok(B, H, W, G, KM)) return 0;
  const int ntiles = B * ud_div_up(W, kTW) * ud_div_up(H, kTH);
  return ud_align_up((size_t)gtail_slices(ntiles) * G * KM * 9 * kHC * sizeof(float));
}

extern "C" int ud_head_tail_f32_wgrad(const float* a, const float* dz, float* dw, int B, int H, int W, int G,
                                      int KM, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!a || !dz || !dw || !gtail_ok(B, H, W, G, KM)) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_head_tail_f32_wgrad_workspace_bytes(B, H, W, G, KM)) return UD_ERR_WORKSPACE;
  GTail t{B, H, W, G, KM, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  hipStream_t stream = (hipStream_t)stream_;
  const int ntiles = B * t.tiles_x * t.tiles_y, S = gtail_slices(ntiles), per = ud_div_up(ntiles, S);
  float* partial = reinterpret_cast<float*>(workspace);
  UdProfScope prof("head_tail.k_gtail_wgrad", stream);
  k_gtail_wgrad<<<dim3(S, G), 256, 0, stream>>>(a, dz, partial, t, ntiles, per);
  UD_LAUNCH_CHECK();
  const long long n = (long long)G * KM * 9 * kHC;
  k_gtail_wsum<<<ud_div_up(n, 256), 256, 0, stream>>>(partial, S, n, dw);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
