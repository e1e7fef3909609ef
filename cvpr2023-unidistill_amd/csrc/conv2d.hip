// Dense 3x3 / stride 1 / pad 1 convolution on channels-last bf16 tensors: the BEV trunk + head convs
// of the reference (BaseBEVBackbone, unidistill/layers/blocks_2d/det3d/base_bev_backbone.py:30-110;
// CenterHead.shared_conv, unidistill/layers/head/det3d/center_head.py:408-420) as a hand-written
// implicit GEMM on v_mfma_f32_16x16x32_bf16.
//
//   y[b,oy,ox,n] = epilogue( sum_{tap,c} x[b, oy+ty-1, ox+tx-1, c] * w[n, tap, c] )
//
// One workgroup (4 waves, each a 64 x 64 sub-tile) owns 8 x 16 output pixels x 128 output channels.
// Per 64-channel slice of Cin the 10 x 18 input halo is staged ONCE in LDS and serves all nine
// taps (the gather a library implicit GEMM repeats per tap); the [128 x 64] weight slice of the next
// (tap, slice) is prefetched into registers while the current one is multiplied and lands in the
// other half of a double-buffered LDS tile, so there is one barrier per 32 MFMAs per wave.
// 144-byte LDS rows keep the ds_read_b128 fragment loads conflict-free; the fp32 result tile is
// staged through LDS so that bias / folded BatchNorm / residual / ReLU are applied on, and stored as,
// 16-byte channel pieces.  The data gradient is the same kernel on the flipped, transposed weights.
#include "ud_common.h"
#include "ud_prof.h"

namespace {

constexpr int kTW = 16, kTH = 8;             // output pixel tile
constexpr int kTM = kTW * kTH;               // 128 GEMM rows
constexpr int kTN = 128;                     // output channels per workgroup
constexpr int kKC = 64;                      // input channels per staged slice
constexpr int kHW = kTW + 2, kHH = kTH + 2;  // halo
constexpr int kHQ = kHW * kHH;               // 180 staged pixels
constexpr int kLD = kKC + 8;                 // bf16 elements per LDS row (144 B)
constexpr int kLDO = kTN + 4;                // fp32 elements per staged output row
constexpr int kAIters = (kHQ * 8 + 255) / 256;   // 16-byte units of the halo per thread (6)
constexpr int kBIters = kTN * 8 / 256;           // 16-byte units of a weight slice per thread (4)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGeom {
  int B, H, W, Cin, Cout, tiles_x, tiles_y;
};
struct ConvEp {
  const float* bias;
  const float* scale;
  const float* shift;
  const unsigned short* residual;
  int relu;
};

constexpr size_t kOperandBytes = (size_t)(kHQ + 2 * kTN) * kLD * 2;
constexpr size_t kOutBytes = (size_t)kTM * kLDO * 4;
constexpr size_t kSmemBytes = kOperandBytes > kOutBytes ? kOperandBytes : kOutBytes;

__global__ __launch_bounds__(256) void k_conv3x3_bf16(const unsigned short* __restrict__ x,
                                                      const unsigned short* __restrict__ w,
                                                      unsigned short* __restrict__ y, ConvGeom gm,
                                                      ConvEp ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);          // [kHQ][kLD]
  unsigned short* Bs = As + kHQ * kLD;                                     // [2][kTN][kLD]
  float* Os = reinterpret_cast<float*>(smem);                              // [kTM][kLDO] after the K loop
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: the 8 XCDs (workgroups are dealt round-robin) each walk a contiguous band
  // of tiles, so halo rows shared by neighbouring tiles meet in the same L2.
  const int ntiles = gm.B * gm.tiles_x * gm.tiles_y;
  const int per = (ntiles + 7) / 8;
  int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int b = tile / (gm.tiles_x * gm.tiles_y);
  tile -= b * gm.tiles_x * gm.tiles_y;
  const int ty0 = (tile / gm.tiles_x) * kTH, tx0 = (tile % gm.tiles_x) * kTW;
  const int n0 = blockIdx.y * kTN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 ra[kAIters], rb[kBIters];
  auto fetch_a = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < kAIters; ++j) {
      const int u = tid + 256 * j;
      const int q = u >> 3, c8 = (u & 7) * 8;
      const int qy = q / kHW, qx = q - qy * kHW;
      const int gy = ty0 + qy - 1, gx = tx0 + qx - 1;
      ra[j] = make_uint4(0u, 0u, 0u, 0u);
      if (q < kHQ && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W)
        ra[j] = *reinterpret_cast<const uint4*>(x + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cin +
                                                chunk * kKC + c8);
    }
  };
  auto commit_a = [&]() {
#pragma unroll
    for (int j = 0; j < kAIters; ++j) {
      const int u = tid + 256 * j;
      if (u < kHQ * 8) *reinterpret_cast<uint4*>(As + (u >> 3) * kLD + (u & 7) * 8) = ra[j];
    }
  };
  auto fetch_b = [&](int chunk, int tap) {
#pragma unroll
    for (int j = 0; j < kBIters; ++j) {
      const int u = tid + 256 * j;
      const int n = u >> 3, c8 = (u & 7) * 8;
      rb[j] = make_uint4(0u, 0u, 0u, 0u);
      if (n0 + n < gm.Cout)
        rb[j] = *reinterpret_cast<const uint4*>(w + ((size_t)(n0 + n) * 9 + tap) * gm.Cin + chunk * kKC + c8);
    }
  };
  auto commit_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kBIters; ++j) {
      const int u = tid + 256 * j;
      *reinterpret_cast<uint4*>(Bs + (buf * kTN + (u >> 3)) * kLD + (u & 7) * 8) = rb[j];
    }
  };

  const int nchunks = gm.Cin / kKC, total = nchunks * 9;
  fetch_a(0);
  fetch_b(0, 0);
  commit_a();
  commit_b(0);
  __syncthreads();
  for (int it = 0; it < total; ++it) {
    const int chunk = it / 9, tap = it - chunk * 9;
    const bool more = it + 1 < total;
    const bool new_chunk = more && tap == 8;
    if (more) fetch_b(new_chunk ? chunk + 1 : chunk, new_chunk ? 0 : tap + 1);
    if (tap == 0 && chunk + 1 < nchunks) fetch_a(chunk + 1);
    {
      const unsigned short* bbase = Bs + ((it & 1) * kTN + 64 * wn + li) * kLD + 8 * g;
      const unsigned short* abase = As + ((4 * wm + tap / 3) * kHW + li + tap % 3) * kLD + 8 * g;
#pragma unroll
      for (int ks = 0; ks < kKC / 32; ++ks) {
        bf16x8 a[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          a[ti] = *reinterpret_cast<const bf16x8*>(abase + ti * kHW * kLD + 32 * ks);
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const bf16x8 bb = *reinterpret_cast<const bf16x8*>(bbase + tj * 16 * kLD + 32 * ks);
#pragma unroll
          for (int ti = 0; ti < 4; ++ti)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti], bb, acc[ti][tj], 0, 0, 0);
        }
      }
    }
    if (more) {
      if (new_chunk) {
        __syncthreads();           // every wave is done with this slice's halo
        commit_a();
      }
      commit_b((it + 1) & 1);
    }
    __syncthreads();
  }
  // epilogue 1: accumulators -> fp32 tile in LDS (aliases the operand tiles; the loop ended on a barrier)
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Os[(64 * wm + 16 * ti + 4 * g + r) * kLDO + 64 * wn + 16 * tj + li] = acc[ti][tj][r];
  __syncthreads();
  // epilogue 2: 16-byte channel pieces: bias, folded BN, residual, ReLU, bf16 store
  for (int u = tid; u < kTM * (kTN / 8); u += 256) {
    const int r = u >> 4, c8 = (u & 15) * 8;
    const int gy = ty0 + (r >> 4), gx = tx0 + (r & 15);
    const int n = n0 + c8;
    if (gy >= gm.H || gx >= gm.W || n >= gm.Cout) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(Os + r * kLDO + c8);
    const float4 v1 = *reinterpret_cast<const float4*>(Os + r * kLDO + c8 + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if (ep.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += ep.bias[n + e];
    }
    if (ep.scale) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * ep.scale[n + e] + ep.shift[n + e];
    }
    const size_t off = ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cout + n;
    if (ep.residual) {
      const uint4 h = *reinterpret_cast<const uint4*>(ep.residual + off);
      const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] += __uint_as_float(hw[e] << 16);
        v[2 * e + 1] += __uint_as_float(hw[e] & 0xFFFF0000u);
      }
    }
    if (ep.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<uint4*>(y + off) =
        make_uint4(ud_pack_bf16x2(v[0], v[1]), ud_pack_bf16x2(v[2], v[3]), ud_pack_bf16x2(v[4], v[5]),
                   ud_pack_bf16x2(v[6], v[7]));
  }
}

}  // namespace

extern "C" int ud_conv3x3_nhwc_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cin,
                                    int Cout, const float* bias, const float* scale,
                                    const float* shift, const void* residual, int relu,
                                    ud_stream_t stream_) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 8 != 0) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom gm{B, H, W, Cin, Cout, ud_div_up(W, kTW), ud_div_up(H, kTH)};
  ConvEp ep{bias, scale, shift, reinterpret_cast<const unsigned short*>(residual), relu};
  static bool attr_set = false;
  if (!attr_set) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_bf16, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kSmemBytes));
    attr_set = true;
  }
  const int ntiles = B * gm.tiles_x * gm.tiles_y;
  const int gx = (ntiles + 7) / 8 * 8;
  UdProfScope prof("conv2d.k_conv3x3", stream);
  k_conv3x3_bf16<<<dim3(gx, ud_div_up(Cout, kTN)), 256, kSmemBytes, stream>>>(
      reinterpret_cast<const unsigned short*>(x), reinterpret_cast<const unsigned short*>(w),
      reinterpret_cast<unsigned short*>(y), gm, ep);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
