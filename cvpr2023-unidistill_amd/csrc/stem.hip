// Frozen ResNet stem for MI355X / gfx950: conv 7x7 / stride 2 / pad 3 (3 -> 64 channels) + eval-mode BatchNorm + ReLU in one
// kernel, max-pool 3x3 / stride 2 / pad 1 in a second one.
//
// Replaces conv1 / bn1 / relu / maxpool of the mmdet ResNet-50 the reference builds as its image backbone
// (unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:143-149, configured with frozen_stages = 0 in
// exps/multisensor_fusion/nuscenes/BEVFusion/BEVFusion_nuscenes_centerhead_fusion_exp.py:24-31: the stem has no gradient and bn1
// runs on its running statistics).  Forward only -- nothing upstream of the stem takes a gradient.
//
// The convolution is an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation: the reference's
// arithmetic).  GEMM row = output pixel, column = output channel, reduction index k = (ky, kx, c).  With three input channels
// the 21 values (kx, c) of one kernel row are CONTIGUOUS in an interleaved (x, c) image row, so the input patch of a workgroup
// is staged once in LDS as rows of (x, c) floats and the A operand needs no im2col: lane (pixel li, k-group g) reads the six
// consecutive floats j = 6 g .. 6 g + 5 of its pixel's window (a kernel row padded from 21 to 24 values: the three pad values
// meet zero weights), i.e. MFMA step s multiplies the four reduction indices {6 g + s}.  The filters are packed on the host
// in exactly that order ([ky][s][g][channel li][tile t]: one ds_read_b128 per lane and step) and stay in LDS for the lifetime of
// the (persistent) workgroup.  Any input layout is read through its strides (NCHW planes or channels-last): the 52 MB layout
// copy the library path needed in front of the stem is gone.
#include "ud_common.h"
#include "ud_prof.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kTR = 8, kTC = 32;                 // conv-output tile of a workgroup: 8 rows x 32 columns (4 waves x 2 rows)
constexpr int kPR = 2 * kTR + 5;                 // 21 input rows
constexpr int kPC = 2 * kTC + 5;                 // 69 input columns
constexpr int kRowF = 216;                       // floats per staged row: 69 x 3 = 207 used, reads reach 2 * 31 * 3 + 23 = 209
constexpr int kWFloats = 7 * 6 * 4 * 16 * 4;     // packed filters: [ky][s][g][li][t]
constexpr int kStemLds = (kPR * kRowF + kWFloats) * 4;

struct StemGeom {
  int B, H, W, OH, OW, tiles_y, tiles_x;
  long long sb, sc, sy, sx;                      // input strides in floats
};

template <bool BF16_OUT>
__global__ __launch_bounds__(256) void k_stem_conv(const float* __restrict__ x, const float* __restrict__ wpk,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   void* __restrict__ y, StemGeom gm) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* s_w = smem_f;                           // [42][4][16][4]
  float* s_p = smem_f + kWFloats;                // [21][216]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  for (int i = tid; i < kWFloats / 4; i += 256)
    reinterpret_cast<float4*>(s_w)[i] = reinterpret_cast<const float4*>(wpk)[i];
  float sc[4], sh[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    sc[t] = scale[16 * t + li];
    sh[t] = shift[16 * t + li];
  }
  const int ntiles = gm.B * gm.tiles_y * gm.tiles_x;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (gm.tiles_y * gm.tiles_x);
    const int tr = tile - b * gm.tiles_y * gm.tiles_x;
    const int oy0 = (tr / gm.tiles_x) * kTR, ox0 = (tr % gm.tiles_x) * kTC;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    __syncthreads();                             // everybody is done with the previous patch (and the filters are in place)
    for (int e = tid; e < kPR * kRowF; e += 256) {
      const int r = e / kRowF, j = e - r * kRowF;
      const int xl = j / 3, c = j - 3 * xl;
      const int iy = iy0 + r, ix = ix0 + xl;
      float v = 0.f;                             // zero padding; the pad floats must be finite too (they meet zero weights)
      if (j < kPC * 3 && iy >= 0 && iy < gm.H && ix >= 0 && ix < gm.W)
        v = x[(long long)b * gm.sb + (long long)c * gm.sc + (long long)iy * gm.sy + (long long)ix * gm.sx];
      s_p[e] = v;
    }
    __syncthreads();
    f32x4 acc[4][4];                             // [pixel tile m = (row, column half)][channel tile t]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
      f32x2 a[4][3];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int row = 2 * (2 * wave + (m >> 1)) + ky, col = 16 * (m & 1) + li;
        const float* p = s_p + row * kRowF + 6 * col + 6 * g;
#pragma unroll
        for (int q = 0; q < 3; ++q) a[m][q] = *reinterpret_cast<const f32x2*>(p + 2 * q);
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const f32x4 bw = *reinterpret_cast<const f32x4*>(s_w + (((ky * 6 + s) * 4 + g) * 16 + li) * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s >> 1][s & 1], bw[t], acc[m][t], 0, 0, 0);
      }
    }
    // acc[m][t][r]: pixel (row 2 wave + (m >> 1), column 16 (m & 1) + 4 g + r), channel 16 t + li
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int oy = oy0 + 2 * wave + (m >> 1);
      if (oy >= gm.OH) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ox = ox0 + 16 * (m & 1) + 4 * g + r;
        if (ox >= gm.OW) continue;
        const size_t o = (((size_t)b * gm.OH + oy) * gm.OW + ox) * 64 + li;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float u = acc[m][t][r] * sc[t] + sh[t];
          const float v = u < 0.f ? 0.f : u;          // ReLU that keeps a NaN, as torch.relu does
          if (BF16_OUT) reinterpret_cast<unsigned short*>(y)[o + 16 * t] = (unsigned short)(ud_pack_bf16x2(v, 0.f) & 0xFFFFu);
          else reinterpret_cast<float*>(y)[o + 16 * t] = v;
        }
      }
    }
  }
}

__device__ __forceinline__ float pmax(float best, float v) { return (v > best || v != v) ? v : best; }   // NaN wins, as in PyTorch

// max-pool 3x3 / stride 2 / pad 1 over a channels-last map with C a multiple of 8 / 4: a thread per (output pixel, 16-byte
// channel piece).  Out-of-image taps are skipped (PyTorch pads with -inf).
template <typename T>
__global__ __launch_bounds__(256) void k_maxpool3s2(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C,
                                                    int PH, int PW) {
  constexpr int kV = 16 / sizeof(T);             // channels per 16-byte piece
  const int pieces = C / kV;
  const long long total = (long long)B * PH * PW * pieces;
  const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
  if (u >= total) return;
  const int pc = (int)(u % pieces);
  long long p = u / pieces;
  const int px = (int)(p % PW);
  p /= PW;
  const int py = (int)(p % PH), b = (int)(p / PH);
  float best[kV];
#pragma unroll
  for (int e = 0; e < kV; ++e) best[e] = -INFINITY;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int iy = 2 * py + dy;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int ix = 2 * px + dx;
      if (ix < 0 || ix >= W) continue;
      const uint4 q = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + pc * kV);
      const unsigned qw[4] = {q.x, q.y, q.z, q.w};
      if (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) best[e] = pmax(best[e], __uint_as_float(qw[e]));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          best[2 * e] = pmax(best[2 * e], __uint_as_float(qw[e] << 16));
          best[2 * e + 1] = pmax(best[2 * e + 1], __uint_as_float(qw[e] & 0xFFFF0000u));
        }
      }
    }
  }
  uint4 o;
  if (sizeof(T) == 4) {
    o = make_uint4(__float_as_uint(best[0]), __float_as_uint(best[1]), __float_as_uint(best[2]), __float_as_uint(best[3]));
  } else {   // the inputs are bf16 values: the maxima are exactly representable
    o = make_uint4((__float_as_uint(best[0]) >> 16) | (__float_as_uint(best[1]) & 0xFFFF0000u),
                   (__float_as_uint(best[2]) >> 16) | (__float_as_uint(best[3]) & 0xFFFF0000u),
                   (__float_as_uint(best[4]) >> 16) | (__float_as_uint(best[5]) & 0xFFFF0000u),
                   (__float_as_uint(best[6]) >> 16) | (__float_as_uint(best[7]) & 0xFFFF0000u));
  }
  *reinterpret_cast<uint4*>(y + (((size_t)b * PH + py) * PW + px) * C + pc * kV) = o;
}

}  // namespace

extern "C" int ud_stem_pack_weights(const float* w, int64_t sn, int64_t sc, int64_t sky, int64_t skx, float* packed) {
  // HOST helper (w: host pointer to the 64 x 3 x 7 x 7 filters with the given strides in floats)
  if (!w || !packed) return UD_ERR_INVALID_ARG;
  for (int ky = 0; ky < 7; ++ky)
    for (int s = 0; s < 6; ++s)
      for (int g = 0; g < 4; ++g)
        for (int li = 0; li < 16; ++li)
          for (int t = 0; t < 4; ++t) {
            const int j = 6 * g + s, kx = j / 3, c = j % 3, n = 16 * t + li;
            packed[((((ky * 6 + s) * 4 + g) * 16 + li) * 4) + t] = j < 21 ? w[n * sn + c * sc + ky * sky + kx * skx] : 0.f;
          }
  return UD_OK;
}

extern "C" int ud_stem_conv7x7_bn_relu(const float* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int H, int W,
                                       const float* packed_w, const float* scale, const float* shift, void* y, int out_bf16,
                                       ud_stream_t stream_) {
  if (!x || !packed_w || !scale || !shift || !y || B <= 0 || H <= 0 || W <= 0) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  StemGeom gm;
  gm.B = B;
  gm.H = H;
  gm.W = W;
  gm.OH = (H + 6 - 7) / 2 + 1;
  gm.OW = (W + 6 - 7) / 2 + 1;
  gm.tiles_y = ud_div_up(gm.OH, kTR);
  gm.tiles_x = ud_div_up(gm.OW, kTC);
  gm.sb = sb;
  gm.sc = sc;
  gm.sy = sy;
  gm.sx = sx;
  const long long ntiles = (long long)B * gm.tiles_y * gm.tiles_x;
  if (ntiles >= (1ll << 31)) return UD_ERR_INVALID_ARG;
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_stem_conv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemLds));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_stem_conv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kStemLds));
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("stem.k_stem_conv", stream);
  const int grid = (int)(ntiles < 512 ? ntiles : 512);          // persistent: two workgroups per CU keep the filters in LDS
  if (out_bf16) k_stem_conv<true><<<grid, 256, kStemLds, stream>>>(x, packed_w, scale, shift, y, gm);
  else k_stem_conv<false><<<grid, 256, kStemLds, stream>>>(x, packed_w, scale, shift, y, gm);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_maxpool3x3s2_nhwc(const void* x, void* y, int B, int H, int W, int C, int is_bf16, ud_stream_t stream_) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % (is_bf16 ? 8 : 4)) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int PH = (H + 2 - 3) / 2 + 1, PW = (W + 2 - 3) / 2 + 1;
  const long long total = (long long)B * PH * PW * (C / (is_bf16 ? 8 : 4));
  if ((total + 255) / 256 >= (1ll << 31)) return UD_ERR_INVALID_ARG;
  UdProfScope prof("stem.k_maxpool", stream);
  if (is_bf16)
    k_maxpool3s2<unsigned short><<<(unsigned)ud_div_up(total, 256), 256, 0, stream>>>(
        (const unsigned short*)x, (unsigned short*)y, B, H, W, C, PH, PW);
  else
    k_maxpool3s2<float><<<(unsigned)ud_div_up(total, 256), 256, 0, stream>>>((const float*)x, (float*)y, B, H, W, C, PH, PW);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
