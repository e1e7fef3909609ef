// fp32 twin of conv2d.hip: 3x3 / stride 1 / pad 1 and 1x1 convolutions on channels-last FP32 tensors for the
// reference's own arithmetic (it trains in fp32: base_cli.py:40-45 has no precision flag) -- the BEV trunk
// (base_bev_backbone.py:30-110), the head (center_head.py:311-420), the fusion conv (base_exp.py:107-135) and
// the ResNet / neck convolutions (lss_fpn.py:143-149) -- on v_mfma_f32_16x16x4_f32 (exact fp32 products,
// fp32 accumulation: 157 TFLOP/s matrix peak, 1/16 of the bf16 pipe).
//
//   y[b,oy,ox,n] = epilogue( sum_{tap,c} x[b, oy+ty-1, ox+tx-1, c] * w[n, tap, c] )
//
// Same skeleton as the bf16 kernel: a workgroup (4 waves) owns 8 x 16 output pixels x TN output channels;
// per 32-channel slice (one 128-byte LDS row per pixel) the 10 x 18 input halo is staged once and serves all
// nine taps; halo and weight slices travel L2 -> LDS by LDS-DMA into unpadded, source-swizzled, double-buffered
// tiles.  With a 32-cycle MFMA the kernel is MFMA-bound by a wide margin (16 KB of fragments per 4096 MFMA
// cycles per wave), so the lever here is simply to keep the MFMA pipe issuing back to back.
// Fragment trick: a lane reads ONE 16-byte piece (4 consecutive channels) per operand and feeds four MFMAs
// with its elements 0..3; MFMA e then reduces over the channels {4g + e : g = lane group 0..3} -- a
// permutation of the slice's channels that A and B share, so the sum is the same.
#include "ud_common.h"
#include "ud_prof.h"
#include "conv_pixmap.h"
#include <cstdlib>

namespace {

constexpr int kTW = 16, kTH = 8, kTM = kTW * kTH;
constexpr int kKC = 32;                      // fp32 input channels per staged slice (one 128-byte LDS row)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGeomF {
  int B, H, W, Cin, Cout, tiles_x, tiles_y;
  long long npix;
  PixMap imap, omap;   // mapped 1x1 launches only (mode 0 everywhere else)
};
struct ConvEpF {
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  int relu;
  int reverse_taps;
  float* stats;       // [tiles][Cout][2] per-tile (sum, sum of squares) of the stored outputs, or nullptr (BatchNorm statistics)
};

// LDS of the 1x1 kernel: two pixel-tile slices, two weight slices; the output tile reuses the space after the K loop
constexpr size_t conv_smem_bytes_f(int tn) {
  const size_t operands = 2 * (size_t)kTM * 128 + 2 * (size_t)tn * 128;
  const size_t out = (size_t)kTM * (tn + 4) * 4;
  return operands > out ? operands : out;
}

__device__ __attribute__((aligned(16))) unsigned int g_zero16f[4];

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Shared epilogue: accumulators -> fp32 tile in LDS (aliases the operand tiles; the K loop ended on a barrier), then 16-byte
// channel pieces: bias, folded BN, residual, ReLU.
template <int TN, int RW, int TM = kTM>
__device__ __forceinline__ void store_tile_f32(const f32x4 (&acc)[RW][4], float* Os, int tid, int wm, int wn, int g, int li,
                                               int b, int ty0, int tx0, int n0, float* __restrict__ y,
                                               const ConvGeomF& gm, const ConvEpF& ep, int tile_lin = 0) {
  constexpr int kTN = TN, kLDO = TN + 4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;   // ep.stats: see conv_store_tile in conv2d.hip
#pragma unroll
  for (int ti = 0; ti < RW; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Os[(16 * RW * wm + 16 * ti + 4 * g + r) * kLDO + 64 * wn + 16 * tj + li] = acc[ti][tj][r];
  __syncthreads();
  for (int u = tid; u < TM * (kTN / 4); u += 256) {
    const int r = u / (kTN / 4), c4 = (u - r * (kTN / 4)) * 4;
    const int gy = ty0 + (r >> 4), gx = tx0 + (r & 15);
    const int n = n0 + c4;
    if (gy >= gm.H || gx >= gm.W || n >= gm.Cout || (long long)(b * gm.H + gy) * gm.W + gx >= gm.npix) continue;
    float4 v = *reinterpret_cast<const float4*>(Os + r * kLDO + c4);
    if (ep.bias) {
      const float4 bv = *reinterpret_cast<const float4*>(ep.bias + n);
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (ep.scale) {
      const float4 sc = *reinterpret_cast<const float4*>(ep.scale + n);
      const float4 sh = *reinterpret_cast<const float4*>(ep.shift + n);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    }
    const size_t off = gm.omap.mode ? gm.omap.off((long long)(b * gm.H + gy) * gm.W + gx, n, gm.Cout)
                                    : ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cout + n;
    if (ep.residual) {
      const float4 h = *reinterpret_cast<const float4*>(ep.residual + off);
      v.x += h.x; v.y += h.y; v.z += h.z; v.w += h.w;
    }
    if (ep.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4*>(y + off) = v;
    if (ep.stats) {
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
    }
  }
  if (ep.stats) {
    // a thread keeps ONE 4-channel piece over all its rows (256 % (TN / 4) == 0): reduce the row groups through LDS
    constexpr int kGroups = 256 / (kTN / 4);
    __syncthreads();                                   // every thread is done reading the output tile
    const int grp = tid / (kTN / 4), c4 = (tid % (kTN / 4)) * 4;
    const float a4[4] = {s1.x, s1.y, s1.z, s1.w}, q4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Os[(grp * kTN + c4 + e) * 2] = a4[e];
      Os[(grp * kTN + c4 + e) * 2 + 1] = q4[e];
    }
    __syncthreads();
    if (tid < kTN && n0 + tid < gm.Cout) {
      double ad = 0.0, qd = 0.0;          // row-group sums combine in double
      for (int k = 0; k < kGroups; ++k) {
        ad += (double)Os[(k * kTN + tid) * 2];
        qd += (double)Os[(k * kTN + tid) * 2 + 1];
      }
      const float a = (float)ad, q = (float)qd;
      ep.stats[((size_t)tile_lin * gm.Cout + n0 + tid) * 2] = a;
      ep.stats[((size_t)tile_lin * gm.Cout + n0 + tid) * 2 + 1] = q;
    }
  }
}

// 3x3: the nine taps of a 32-channel slice unrolled so that every per-tap quantity is an instruction immediate (see
// k_conv3x3_taps in conv2d.hip, whose SQ counters motivated it: a runtime (slice, tap) loop spends ~220 address / control
// instructions per (tap, slice) next to the MFMAs).
template <int TN, int RW>
__global__ __launch_bounds__(256) void k_conv_f32_taps(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, ConvGeomF gm, ConvEpF ep) {
  constexpr int WM = TN == 128 ? 2 : 4, TH = WM * RW, NB = TN / 32;   // tile: TH x 16 output pixels, RW rows per wave
  constexpr int kHW = kTW + 2, kHQ = kHW * (TH + 2), kHQP = (kHQ + 7) / 8 * 8, kAInstr = kHQP / 8;
  constexpr int kBBytes = TN * 128, kABytes = kHQP * 128, kAOff = 2 * kBBytes, kAPer = (kAInstr + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Os = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wm = TN == 128 ? (wave >> 1) : wave, wn = TN == 128 ? (wave & 1) : 0;
  const int ntiles = gm.B * gm.tiles_x * gm.tiles_y;
  const int per = (ntiles + 7) / 8;
  int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int tile_lin = tile;
  const int b = tile / (gm.tiles_x * gm.tiles_y);
  tile -= b * gm.tiles_x * gm.tiles_y;
  const int ty0 = (tile / gm.tiles_x) * TH, tx0 = (tile % gm.tiles_x) * kTW;
  const int n0 = blockIdx.y * TN;
  const float* zero = reinterpret_cast<const float*>(g_zero16f);

  f32x4 acc[RW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int r8 = lane >> 3, slot = lane & 7;
  const float* pa[kAPer];       // halo pieces of this wave: source of slice 0, stepped by one slice (0 on the zero page)
  int inca[kAPer];
#pragma unroll
  for (int i = 0; i < kAPer; ++i) {
    const int q = (wave + 4 * i) * 8 + r8;
    const int qy = q / kHW, qx = q - qy * kHW;
    const int gy = ty0 + qy - 1, gx = tx0 + qx - 1;
    const bool ok = q < kHQ && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
    pa[i] = ok ? x + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cin + ((slot ^ (q & 7)) << 2) : zero;
    inca[i] = ok ? kKC : 0;
  }
  auto stage_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kAPer; ++i) {
      if (wave + 4 * i < kAInstr)
        dma16(pa[i], reinterpret_cast<float*>(smem + kAOff + buf * kABytes + (wave + 4 * i) * 1024));
      pa[i] += inca[i];
    }
  };
  unsigned voffb[NB];           // weight pieces: byte offset from the (tap, slice) base; channels past Cout re-read the last one
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave + 4 * j) * 8 + r8;
    const int nn = min(n0 + n, gm.Cout - 1);
    voffb[j] = (unsigned)(((size_t)nn * 9 * gm.Cin + ((slot ^ (n & 7)) << 2)) * 4);
  }
  auto stage_b = [&](int chunk, int tap, int buf) {
    const int te = ep.reverse_taps ? 8 - tap : tap;
    const char* wb = reinterpret_cast<const char*>(w) + ((size_t)te * gm.Cin + (size_t)chunk * kKC) * 4;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      dma16(reinterpret_cast<const float*>(wb + voffb[j]),
            reinterpret_cast<float*>(smem + buf * kBBytes + (wave + 4 * j) * 1024));
  };
  const int q0 = RW * wm * kHW + li;
  unsigned sa[2][8], sb[2][2];  // swizzled fragment addresses: halo row q0 + d is sa[ks][d & 7] + 128 d
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int c = 0; c < 8; ++c) sa[ks][c] = kAOff + q0 * 128 + (((4 * ks + g) ^ ((q0 + c) & 7)) << 4);
#pragma unroll
    for (int p = 0; p < 2; ++p) sb[ks][p] = p * kBBytes + (64 * wn + li) * 128 + (((4 * ks + g) ^ (li & 7)) << 4);
  }

  const int nchunks = gm.Cin / kKC;
  stage_a(0);
  stage_b(0, 0, 0);
  __syncthreads();
  int adelta = kABytes;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int cpar = chunk & 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) stage_b(chunk, tap + 1, ((tap + 1) & 1) ^ cpar);
      else if (chunk + 1 < nchunks) stage_b(chunk + 1, 0, cpar ^ 1);
      if (tap == 0 && chunk + 1 < nchunks) stage_a(cpar ^ 1);
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f32x4 a[RW];
#pragma unroll
        for (int ti = 0; ti < RW; ++ti) {
          const int d = (ti + dy) * kHW + dx;
          a[ti] = *reinterpret_cast<const f32x4*>(smem + sa[ks][d & 7] + d * 128);
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const f32x4 bb = *reinterpret_cast<const f32x4*>(smem + sb[ks][tap & 1] + tj * 2048);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ti = 0; ti < RW; ++ti)
              acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti][e], bb[e], acc[ti][tj], 0, 0, 0);
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {   // nine taps: the weight buffers end a slice on the other parity; the halo alternates
      const unsigned t = sb[ks][0];
      sb[ks][0] = sb[ks][1];
      sb[ks][1] = t;
#pragma unroll
      for (int c = 0; c < 8; ++c) sa[ks][c] += adelta;
    }
    adelta = -adelta;
  }
  store_tile_f32<TN, RW, TH * kTW>(acc, Os, tid, wm, wn, g, li, b, ty0, tx0, n0, y, gm, ep, tile_lin);
}

constexpr size_t conv_taps_smem_bytes_f(int tn, int rw) {
  const int th = (tn == 128 ? 2 : 4) * rw;
  const size_t operands = 2 * (size_t)(((kTW + 2) * (th + 2) + 7) / 8 * 8) * 128 + 2 * (size_t)tn * 128;
  const size_t out = (size_t)th * kTW * (tn + 4) * 4;
  return operands > out ? operands : out;
}

// Plain 1x1 twin of k_conv_f32_taps: everything is affine in the 32-channel slice index (see k_conv1x1_line in conv2d.hip).
template <int TN>
__global__ __launch_bounds__(256) void k_conv1x1_f32_line(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, ConvGeomF gm, ConvEpF ep) {
  constexpr int WM = TN == 128 ? 2 : 4, RW = 8 / WM, NB = TN / 32;
  constexpr int kABytes = kTM * 128, kBBytes = TN * 128, kBOff = 2 * kABytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wm = TN == 128 ? (wave >> 1) : wave, wn = TN == 128 ? (wave & 1) : 0;
  const int ntiles = gm.tiles_y;
  const int per = (ntiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int n0 = blockIdx.y * TN;

  f32x4 acc[RW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int r8 = lane >> 3, slot = lane & 7;
  const float* pa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 4 * i) * 8 + r8;
    long long p = (long long)tile * kTM + r;
    if (p >= gm.npix) p = gm.npix - 1;            // rows past the last pixel re-read it (never stored)
    pa[i] = x + (size_t)p * gm.Cin + ((slot ^ (r & 7)) << 2);
  }
  const float* pb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave + 4 * j) * 8 + r8;
    pb[j] = w + (size_t)min(n0 + n, gm.Cout - 1) * gm.Cin + ((slot ^ (n & 7)) << 2);
  }
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(pa[i], reinterpret_cast<float*>(smem + buf * kABytes + (wave + 4 * i) * 1024));
      pa[i] += kKC;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      dma16(pb[j], reinterpret_cast<float*>(smem + kBOff + buf * kBBytes + (wave + 4 * j) * 1024));
      pb[j] += kKC;
    }
  };
  unsigned sa[2], sb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    sa[ks] = (RW * wm * 16 + li) * 128 + (((4 * ks + g) ^ (li & 7)) << 4);
    sb[ks] = kBOff + (64 * wn + li) * 128 + (((4 * ks + g) ^ (li & 7)) << 4);
  }
  auto mma = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 a[RW];
#pragma unroll
      for (int ti = 0; ti < RW; ++ti) a[ti] = *reinterpret_cast<const f32x4*>(smem + sa[ks] + buf * kABytes + ti * 2048);
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(smem + sb[ks] + buf * kBBytes + tj * 2048);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ti = 0; ti < RW; ++ti)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti][e], bb[e], acc[ti][tj], 0, 0, 0);
      }
    }
  };
  const int nchunks = gm.Cin / kKC;
  stage(0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; chunk += 2) {
    if (chunk + 1 < nchunks) stage(1);
    mma(0);
    __syncthreads();
    if (chunk + 1 < nchunks) {
      if (chunk + 2 < nchunks) stage(0);
      mma(1);
      __syncthreads();
    }
  }
  store_tile_f32<TN, RW>(acc, reinterpret_cast<float*>(smem), tid, wm, wn, g, li, 0, tile * kTH, 0, n0, y, gm, ep, tile);
}

// 1x1 over a pixel map, fp32 twin of k_conv1x1_mapped (conv2d.hip): strided / transposed / im2col launches of the fp32 mode.
// Source of (pixel p, 32-channel slice k0) = x + base(p) + koff(k0) under a range test, base / py / px per lane (once),
// koff / dy / dx uniform and advanced incrementally per slice.
template <int TN>
__global__ __launch_bounds__(256) void k_conv1x1_mapped_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, ConvGeomF gm, ConvEpF ep) {
  constexpr int WM = TN == 128 ? 2 : 4, RW = 8 / WM, NB = TN / 32;
  constexpr int kABytes = kTM * 128, kBBytes = TN * 128, kBOff = 2 * kABytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wm = TN == 128 ? (wave >> 1) : wave, wn = TN == 128 ? (wave & 1) : 0;
  const int ntiles = gm.tiles_y;
  const int per = (ntiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int n0 = blockIdx.y * TN;
  const float* zero = reinterpret_cast<const float*>(g_zero16f);
  const PixMap& im = gm.imap;
  const int mode = im.mode;

  f32x4 acc[RW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int r8 = lane >> 3, slot = lane & 7;
  const float* pa[4];
  int py[4], px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 4 * i) * 8 + r8;
    long long p = (long long)tile * kTM + r;
    if (p >= gm.npix) p = gm.npix - 1;           // rows past the last pixel re-read it (never stored)
    py[i] = px[i] = 0;
    if (mode == 0) {                             // plain input, mapped output (transposed convolution)
      pa[i] = x + (size_t)p * gm.Cin + ((slot ^ (r & 7)) << 2);
      continue;
    }
    const unsigned pu = (unsigned)p, t = pu / (unsigned)im.Wo;        // rows < 2^30 (launcher): 32-bit divisions
    const int ox = (int)(pu - t * (unsigned)im.Wo);
    const int b = (int)(t / (unsigned)im.Ho), oy = (int)(t - (unsigned)b * (unsigned)im.Ho);
    int y0, x0;
    if (mode == 4) { y0 = oy; x0 = ox; }
    else if (mode == 2) { y0 = im.s * oy + im.a; x0 = im.s * ox + im.b; }
    else { y0 = im.s * oy; x0 = im.s * ox; }
    pa[i] = x + ((size_t)(b * im.H + y0) * im.W + x0) * im.C + ((slot ^ (r & 7)) << 2);
    if (mode >= 3) { py[i] = y0; px[i] = x0; }
  }
  const float* pb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave + 4 * j) * 8 + r8;
    const size_t row = (size_t)min(n0 + n, gm.Cout - 1);     // channels past Cout re-read the last one (never stored)
    pb[j] = w + (mode == 4 ? row * 9 * im.C : row * gm.Cin) + ((slot ^ (n & 7)) << 2);
  }
  const int seg_len = mode == 1 ? im.s * im.C : (mode >= 3 ? im.C : gm.Cin);
  int seg = 0, c = 0;
  auto stage = [&](int buf) {
    long long koff;
    int dy = 0, dx = 0, wk;
    if (mode == 1) {
      koff = (long long)seg * im.W * im.C + c;
      wk = seg * seg_len + c;
    } else if (mode == 3) {
      const int ty = seg / 3, tx = seg - 3 * ty;
      dy = ty - 1; dx = tx - 1;
      koff = ((long long)dy * im.W + dx) * im.C + c;
      wk = seg * seg_len + c;
    } else if (mode == 4) {
      const int nx = 1 + im.b, jy = seg / nx, jx = seg - jy * nx;
      dy = im.a * (1 - jy); dx = im.b * (1 - jx);
      koff = ((long long)dy * im.W + dx) * im.C + c;
      const int ty = im.a ? 2 * jy : 1, tx = im.b ? 2 * jx : 1;
      wk = (ty * 3 + tx) * im.C + c;
    } else {
      koff = c;
      wk = c;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = mode < 3 || ((unsigned)(py[i] + dy) < (unsigned)im.H && (unsigned)(px[i] + dx) < (unsigned)im.W);
      dma16(ok ? pa[i] + koff : zero, reinterpret_cast<float*>(smem + buf * kABytes + (wave + 4 * i) * 1024));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      dma16(pb[j] + wk, reinterpret_cast<float*>(smem + kBOff + buf * kBBytes + (wave + 4 * j) * 1024));
    c += kKC;
    if (c == seg_len) { c = 0; ++seg; }
  };
  unsigned sa[2], sb[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    sa[ks] = (RW * wm * 16 + li) * 128 + (((4 * ks + g) ^ (li & 7)) << 4);
    sb[ks] = kBOff + (64 * wn + li) * 128 + (((4 * ks + g) ^ (li & 7)) << 4);
  }
  auto mma = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 a[RW];
#pragma unroll
      for (int ti = 0; ti < RW; ++ti) a[ti] = *reinterpret_cast<const f32x4*>(smem + sa[ks] + buf * kABytes + ti * 2048);
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) {
        const f32x4 bb = *reinterpret_cast<const f32x4*>(smem + sb[ks] + buf * kBBytes + tj * 2048);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int ti = 0; ti < RW; ++ti)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti][e], bb[e], acc[ti][tj], 0, 0, 0);
      }
    }
  };
  const int nchunks = gm.Cin / kKC;
  stage(0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; chunk += 2) {
    if (chunk + 1 < nchunks) stage(1);
    mma(0);
    __syncthreads();
    if (chunk + 1 < nchunks) {
      if (chunk + 2 < nchunks) stage(0);
      mma(1);
      __syncthreads();
    }
  }
  store_tile_f32<TN, RW>(acc, reinterpret_cast<float*>(smem), tid, wm, wn, g, li, 0, tile * kTH, 0, n0, y, gm, ep, tile);
}

template <int KS>
int launch_f32(const float* x, const float* w, float* y, const ConvGeomF& gm, const ConvEpF& ep, int ntiles,
               const char* name, hipStream_t stream, size_t stats_bytes = 0, int* slices_out = nullptr) {
  UdProfScope prof(name, stream);
  static const int force64 = getenv("UD_F32_TN64") ? atoi(getenv("UD_F32_TN64")) : 0;
  // 1x1: 64-wide output-channel tiles (49 KB of LDS: three workgroups per CU) measured faster than or equal to the 128-wide ones on
  // 19 of the 21 plain 1x1 shapes of the distillation step (tools/time_f32_1x1.py: 8.65 -> 8.17 ms per step); UD_F32_TN64=-1: wide
  const bool narrow = force64 > 0 || gm.Cout <= 64 || (KS == 1 && force64 == 0) || ntiles * ud_div_up(gm.Cout, 128) <= 256;
  const int ntn = ud_div_up(gm.Cout, narrow ? 64 : 128);
  if constexpr (KS == 1) {
    static UdDeviceOnce line_set;
    if (const unsigned long long line_set_bit = line_set.pending()) {
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_f32_line<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)conv_smem_bytes_f(128)));
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_f32_line<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)conv_smem_bytes_f(64)));
      line_set.mark(line_set_bit);
    }
    if (ep.stats) {
      if (stats_bytes < (size_t)ntiles * gm.Cout * 2 * sizeof(float) || !slices_out) return UD_ERR_WORKSPACE;
      *slices_out = ntiles;
    }
    const dim3 grid((ntiles + 7) / 8 * 8, ntn);
    if (narrow) k_conv1x1_f32_line<64><<<grid, 256, conv_smem_bytes_f(64), stream>>>(x, w, y, gm, ep);
    else k_conv1x1_f32_line<128><<<grid, 256, conv_smem_bytes_f(128), stream>>>(x, w, y, gm, ep);
    UD_LAUNCH_CHECK();
    return UD_OK;
  } else {
    static UdDeviceOnce taps_set;
    static int force_rw = 0;     // UD_CONV_RW=n: pixel rows per wave (timing experiments only)
    if (const unsigned long long taps_set_bit = taps_set.pending()) {
#define UD_TAPS_ATTR(TN, RW)                                                                                          \
  UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_f32_taps<TN, RW>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                 (int)conv_taps_smem_bytes_f(TN, RW)))
      UD_TAPS_ATTR(128, 4); UD_TAPS_ATTR(128, 3); UD_TAPS_ATTR(128, 2); UD_TAPS_ATTR(64, 2); UD_TAPS_ATTR(64, 1);
#undef UD_TAPS_ATTR
      if (const char* r = getenv("UD_CONV_RW")) force_rw = atoi(r);
      taps_set.mark(taps_set_bit);
    }
    // Tile height = waves x rows per wave.  A CU works through ceil(workgroups / 256) tiles (two at a time, sharing its
    // MFMA pipes: with a 32-cycle fp32 MFMA a tap is MFMA-bound whatever the height), so the launch takes about
    // ceil(WGs / 256) x rows: 180 x 180 x 4 at 8 rows = 1 104 tiles -> 5 x 4, at 6 rows 1 440 -> 6 x 3 (measured 104 -> 110
    // TFLOP/s; 256 -> 256 @90 x 90: 87 -> 113).
    ConvGeomF g2 = gm;
    const int wmv = narrow ? 4 : 2;
    int rw = narrow ? 2 : 4;
    {
      double best = -1;
      for (int r = narrow ? 2 : 4; r >= (narrow ? 1 : 2); --r) {
        const long long wgs = (long long)gm.B * gm.tiles_x * ud_div_up(gm.H, wmv * r) * ntn;
        const double cost = (double)((wgs + 255) / 256) * (r + 0.1);
        if (best < 0 || cost < best) best = cost, rw = r;
      }
      if (force_rw) rw = force_rw < (narrow ? 1 : 2) ? (narrow ? 1 : 2) : force_rw > (narrow ? 2 : 4) ? (narrow ? 2 : 4) : force_rw;
    }
    g2.tiles_y = ud_div_up(gm.H, wmv * rw);
    if (ep.stats) {
      const int nt = gm.B * g2.tiles_x * g2.tiles_y;
      if (stats_bytes < (size_t)nt * gm.Cout * 2 * sizeof(float) || !slices_out) return UD_ERR_WORKSPACE;
      *slices_out = nt;
    }
    const dim3 grid2((gm.B * g2.tiles_x * g2.tiles_y + 7) / 8 * 8, ntn);
#define UD_TAPS_LAUNCH(TN, RW) k_conv_f32_taps<TN, RW><<<grid2, 256, conv_taps_smem_bytes_f(TN, RW), stream>>>(x, w, y, g2, ep)
    if (narrow) {
      if (rw == 1) UD_TAPS_LAUNCH(64, 1); else UD_TAPS_LAUNCH(64, 2);
    } else {
      if (rw == 2) UD_TAPS_LAUNCH(128, 2); else if (rw == 3) UD_TAPS_LAUNCH(128, 3); else UD_TAPS_LAUNCH(128, 4);
    }
#undef UD_TAPS_LAUNCH
    UD_LAUNCH_CHECK();
    return UD_OK;
  }
}

}  // namespace

extern "C" int ud_conv3x3_nhwc_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin,
                                   int Cout, const float* bias, const float* scale, const float* shift,
                                   const float* residual, int flags, ud_stream_t stream_) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 4 != 0) return UD_ERR_UNSUPPORTED;
  ConvGeomF gm{B, H, W, Cin, Cout, ud_div_up(W, kTW), ud_div_up(H, kTH), (long long)B * H * W};
  ConvEpF ep{bias, scale, shift, residual, flags & 1, (flags >> 1) & 1};
  return launch_f32<3>(x, w, y, gm, ep, B * gm.tiles_x * gm.tiles_y, "conv2d.k_conv3x3_f32", (hipStream_t)stream_);
}

extern "C" int ud_conv3x3_bnstats_nhwc_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin,
                                           int Cout, const float* bias, float* partial, size_t partial_bytes,
                                           int* slices, ud_stream_t stream_) {
  if (!x || !w || !y || !partial || !slices || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 4 != 0) return UD_ERR_UNSUPPORTED;
  ConvGeomF gm{B, H, W, Cin, Cout, ud_div_up(W, kTW), ud_div_up(H, kTH), (long long)B * H * W};
  ConvEpF ep{bias, nullptr, nullptr, nullptr, 0, 0, partial};
  return launch_f32<3>(x, w, y, gm, ep, B * gm.tiles_x * gm.tiles_y, "conv2d.k_conv3x3_f32", (hipStream_t)stream_,
                       partial_bytes, slices);
}

extern "C" int ud_conv1x1_bnstats_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout,
                                           const float* bias, float* partial, size_t partial_bytes, int* slices,
                                           ud_stream_t stream_) {
  if (!x || !w || !y || !partial || !slices || P <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 4 != 0 || P > (int64_t)1 << 30) return UD_ERR_UNSUPPORTED;
  const int H = (int)((P + kTW - 1) / kTW);
  ConvGeomF gm{1, H, kTW, Cin, Cout, 1, ud_div_up(H, kTH), (long long)P};
  ConvEpF ep{bias, nullptr, nullptr, nullptr, 0, 0, partial};
  return launch_f32<1>(x, w, y, gm, ep, gm.tiles_y, "conv2d.k_conv1x1_f32", (hipStream_t)stream_, partial_bytes, slices);
}

extern "C" int ud_conv1x1_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout,
                                   const float* bias, const float* scale, const float* shift,
                                   const float* residual, int flags, ud_stream_t stream_) {
  if (!x || !w || !y || P <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 4 != 0 || P > (int64_t)1 << 30) return UD_ERR_UNSUPPORTED;
  const int H = (int)((P + kTW - 1) / kTW);
  ConvGeomF gm{1, H, kTW, Cin, Cout, 1, ud_div_up(H, kTH), (long long)P};
  ConvEpF ep{bias, scale, shift, residual, flags & 1, 0};
  return launch_f32<1>(x, w, y, gm, ep, gm.tiles_y, "conv2d.k_conv1x1_f32", (hipStream_t)stream_);
}

// fp32 twin of ud_conv1x1_mapped_nhwc_bf16: 1x1 kernel over a mapped input and / or output (strided, transposed and
// im2col launches of the fp32 mode; forward and data gradient).  Map channels % 4 == 0, Cin % 32 == 0, Cout % 4 == 0; a
// 32-channel slice must not straddle a tap / block row.
extern "C" int ud_conv1x1_mapped_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout,
                                          const int* in_map, const int* out_map, ud_stream_t stream_) {
  if (!x || !w || !y || P <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  PixMap im, om;
  if (!map_from_ints(in_map, &im, kKC) || !map_from_ints(out_map, &om, kKC)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 4 != 0 || P > (int64_t)1 << 30) return UD_ERR_UNSUPPORTED;
  if (im.mode == 1 && (im.s * im.C) % kKC != 0) return UD_ERR_UNSUPPORTED;
  if (im.mode == 1 && im.s * im.s * im.C != Cin) return UD_ERR_INVALID_ARG;
  if (im.mode == 2 && im.C != Cin) return UD_ERR_INVALID_ARG;
  if (im.mode == 3 && 9 * im.C != Cin) return UD_ERR_INVALID_ARG;
  if (im.mode == 4 && (1 + im.a) * (1 + im.b) * im.C != Cin) return UD_ERR_INVALID_ARG;
  if (om.mode >= 3) return UD_ERR_UNSUPPORTED;
  if (om.mode == 1 && om.s * om.s * om.C != Cout) return UD_ERR_INVALID_ARG;
  if (om.mode == 2 && om.C != Cout) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int H = (int)((P + kTW - 1) / kTW);
  ConvGeomF gm{1, H, kTW, Cin, Cout, 1, ud_div_up(H, kTH), (long long)P, im, om};
  ConvEpF ep{nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr};
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_mapped_f32<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)conv_smem_bytes_f(128)));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_mapped_f32<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)conv_smem_bytes_f(64)));
    attr_set.mark(attr_set_bit);
  }
  const int ntiles = gm.tiles_y;
  UdProfScope prof("conv2d.k_conv1x1_f32", stream);
  static const int force64 = getenv("UD_F32_TN64") ? atoi(getenv("UD_F32_TN64")) : 0;
  const bool narrow = force64 > 0 || Cout <= 64 || ntiles * ud_div_up(Cout, 128) <= 256;
  const dim3 grid((ntiles + 7) / 8 * 8, ud_div_up(Cout, narrow ? 64 : 128));
  if (narrow) k_conv1x1_mapped_f32<64><<<grid, 256, conv_smem_bytes_f(64), stream>>>(x, w, y, gm, ep);
  else k_conv1x1_mapped_f32<128><<<grid, 256, conv_smem_bytes_f(128), stream>>>(x, w, y, gm, ep);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
