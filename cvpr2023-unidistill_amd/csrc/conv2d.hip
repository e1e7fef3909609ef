// Dense convolutions on channels-last bf16 tensors as hand-written implicit GEMMs on v_mfma_f32_16x16x32_bf16: the BEV
// trunk + head convs of the reference (BaseBEVBackbone, unidistill/layers/blocks_2d/det3d/base_bev_backbone.py:30-110;
// CenterHead.shared_conv, unidistill/layers/head/det3d/center_head.py:408-420), the fusion conv and the ResNet / neck convs
// (unidistill/layers/blocks_2d/mmdet3d/lss_fpn.py:143-149).
//
//   y[b,oy,ox,n] = epilogue( sum_{tap,c} x[b, oy+ty-1, ox+tx-1, c] * w[n, tap, c] )
//
// Kernels in this file
//   k_conv3x3_taps<TN, RW>   3x3 / stride 1 / pad 1, forward and (on transposed weights, taps reversed) data gradient
//   k_conv1x1_line<TN>       plain 1x1 (pixels as a [P][Cin] matrix)
//   k_conv1x1_mapped<TN>     1x1 over a PixMap: k = s / stride s, transposed k = s / stride s, 1x1 / stride s, 3x3 / stride 2
//   k_conv3x3_wgrad_taps, k_conv3x3_wgrad<CT>, k_conv1x1_wgrad_dma<NT, CT>, k_wgrad_sum   weight gradients
// Common design: one workgroup (4 waves, each a (16 RW) x 64 sub-tile) owns a tile of output pixels x TN output channels.
// For the 3x3 kernel the (2 RW + 2) x 18 input halo of a 64-channel slice of Cin is staged ONCE in LDS and serves all nine
// taps (the gather a library implicit GEMM repeats per tap).  Halo and weight slices travel HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write -- the register staged version spent 31 % of its time in the
// weight ds_writes); both are double-buffered, the next (tap, slice) weight tile flies while the current one is
// multiplied: one barrier per 32 MFMAs per wave.  Tiles are unpadded 128-byte rows whose 16-byte slots are XOR-swizzled
// with the row index on the SOURCE address, which keeps every ds_read_b128 fragment load conflict-free; the fp32 result
// tile is staged through LDS so that bias / folded BatchNorm / residual / ReLU are applied on, and stored as, 16-byte
// channel pieces.  Loops are straight line: every per-tap / per-slice address is a register computed once plus an
// instruction immediate (see k_conv3x3_taps for the counters that motivated it).
#include "ud_common.h"
#include "ud_prof.h"
#include "conv_pixmap.h"
#include "wgrad_sum.h"

namespace {

constexpr int kTW = 16, kTH = 8;             // output pixel tile
constexpr int kTM = kTW * kTH;               // 128 GEMM rows
constexpr int kKC = 64;                      // input channels per staged slice (one 128-byte LDS row)
constexpr int kHW = kTW + 2, kHH = kTH + 2;  // halo
constexpr int kHQ = kHW * kHH;               // 180 staged pixels
constexpr int kHQP = (kHQ + 7) / 8 * 8;      // 184: rows are staged 8 at a time
constexpr int kAInstr = kHQP / 8;            // 1-KiB direct-to-LDS pieces of a halo slice (23)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGeom {
  int B, H, W, Cin, Cout, tiles_x, tiles_y;
  long long npix;     // pixels of the tensor (1x1 convs are launched on a [P/16][16] view: the last row may be short)
  PixMap imap, omap;  // 1x1 kernels only (mode 0 everywhere else)
};
struct ConvEp {
  const float* bias;
  const float* scale;
  const float* shift;
  const unsigned short* residual;
  int relu;
  int reverse_taps;   // weights are addressed with tap 8 - t (data gradient on un-flipped weights)
  float* stats;       // [tiles][Cout][2] per-tile (sum, sum of squares) of the stored outputs, or nullptr (BatchNorm statistics)
};

// LDS of the 1x1 kernels: two pixel-tile slices, two weight slices; the fp32 output tile reuses the space after the K loop.
constexpr size_t conv_smem_bytes(int tn) {
  const size_t operands = 2 * (size_t)kTM * kKC * 2 + 2 * (size_t)tn * kKC * 2;
  const size_t out = (size_t)kTM * (tn + 4) * 4;
  return operands > out ? operands : out;
}

constexpr size_t conv_taps_smem_bytes(int tn, int rw) {
  const int th = (tn == 128 ? 2 : 4) * rw;
  const size_t operands = 2 * (size_t)((kHW * (th + 2) + 7) / 8 * 8) * 128 + 2 * (size_t)tn * 128;
  const size_t out = (size_t)th * kTW * (tn + 4) * 4;
  return operands > out ? operands : out;
}

// Out-of-image halo pixels and output channels past Cout are loaded from here (LDS-DMA cannot
// write a constant).
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];

// One wave-wide 1 KiB piece: lane l lands at lds + 16*l (the DMA's fixed pattern) = row l>>3, 16-byte
// slot l&7 of an unpadded 8 x 128 B block.  Slot s of row r holds channel group s ^ (r & 7): the XOR
// swizzle is applied on the SOURCE address, which keeps every later ds_read_b128 conflict-free.
__device__ __forceinline__ void dma16(const unsigned short* src, unsigned short* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Shared epilogue: accumulators -> fp32 tile in LDS (aliases the operand tiles; the K loop ended on a barrier), then
// 16-byte channel pieces: bias, folded BN, residual, ReLU, bf16 store.
// With ep.stats the workgroup also emits, per output channel of its tile, the sum and the sum of squares of the values it
// stored (the bf16-rounded ones: what a separate statistics pass would read back) -- the first pass of a training-mode
// BatchNorm that follows the convolution; ud_bn_stats_from_partials reduces the tiles in a fixed order.
template <int TN, int KS, int TM>
__device__ __forceinline__ void conv_store_rows(float* Os, int tid, int b, int ty0, int tx0, int n0,
                                                unsigned short* __restrict__ y, const ConvGeom& gm, const ConvEp& ep,
                                                int tile_lin);

template <int TN, int KS, int RW, int TM = kTM>
__device__ __forceinline__ void conv_store_tile(const f32x4 (&acc)[RW][4], float* Os, int tid, int wm, int wn, int g,
                                                int li, int b, int ty0, int tx0, int n0, unsigned short* __restrict__ y,
                                                const ConvGeom& gm, const ConvEp& ep, int tile_lin = 0) {
  constexpr int kLDO = TN + 4;
#pragma unroll
  for (int ti = 0; ti < RW; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Os[(16 * RW * wm + 16 * ti + 4 * g + r) * kLDO + 64 * wn + 16 * tj + li] = acc[ti][tj][r];
  __syncthreads();
  conv_store_rows<TN, KS, TM>(Os, tid, b, ty0, tx0, n0, y, gm, ep, tile_lin);
}

// Second half of the epilogue (the fp32 tile is in LDS, published): 16-byte channel pieces -> bias, folded BN, residual, ReLU,
// bf16 store, optional BatchNorm partial sums.
template <int TN, int KS, int TM>
__device__ __forceinline__ void conv_store_rows(float* Os, int tid, int b, int ty0, int tx0, int n0,
                                                unsigned short* __restrict__ y, const ConvGeom& gm, const ConvEp& ep,
                                                int tile_lin) {
  constexpr int kTN = TN, kLDO = TN + 4;
  float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int u = tid; u < TM * (kTN / 8); u += 256) {
    const int r = u / (kTN / 8), c8 = (u - r * (kTN / 8)) * 8;
    const int gy = ty0 + (r >> 4), gx = tx0 + (r & 15);
    const int n = n0 + c8;
    if (gy >= gm.H || gx >= gm.W || n >= gm.Cout || (long long)(b * gm.H + gy) * gm.W + gx >= gm.npix) continue;
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
    const size_t off = KS == 1 ? gm.omap.off((long long)(b * gm.H + gy) * gm.W + gx, n, gm.Cout)
                               : ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cout + n;
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
    const uint4 pk = make_uint4(ud_pack_bf16x2(v[0], v[1]), ud_pack_bf16x2(v[2], v[3]), ud_pack_bf16x2(v[4], v[5]),
                                ud_pack_bf16x2(v[6], v[7]));
    *reinterpret_cast<uint4*>(y + off) = pk;
    if (ep.stats) {
      const unsigned pw[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = __uint_as_float(pw[e] << 16), hi = __uint_as_float(pw[e] & 0xFFFF0000u);
        s1[2 * e] += lo; s2[2 * e] += lo * lo;
        s1[2 * e + 1] += hi; s2[2 * e + 1] += hi * hi;
      }
    }
  }
  if (ep.stats) {
    // a thread keeps ONE 8-channel piece over all its rows (256 % (TN / 8) == 0): reduce the row groups through LDS
    constexpr int kGroups = 256 / (kTN / 8);
    __syncthreads();                                   // every thread is done reading the output tile
    const int grp = tid / (kTN / 8), c8 = (tid % (kTN / 8)) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      Os[(grp * kTN + c8 + e) * 2] = s1[e];
      Os[(grp * kTN + c8 + e) * 2 + 1] = s2[e];
    }
    __syncthreads();
    if (tid < kTN && n0 + tid < gm.Cout) {
      float a = 0.f, q = 0.f;
      for (int k = 0; k < kGroups; ++k) {
        a += Os[(k * kTN + tid) * 2];
        q += Os[(k * kTN + tid) * 2 + 1];
      }
      ep.stats[((size_t)tile_lin * gm.Cout + n0 + tid) * 2] = a;
      ep.stats[((size_t)tile_lin * gm.Cout + n0 + tid) * 2 + 1] = q;
    }
  }
}

// ---- 1x1 over a pixel map (PixMap: strided / transposed / im2col launches) ------------------------------------------------
// TN = output channels per workgroup: 128 (2 x 2 waves of 64 x 64) or 64 (4 x 1 waves of 32 x 64).  The staged tile is the
// 128-pixel tile itself ([P/16][16] view of the virtual pixel matrix), one 64-channel slice of the virtual K' per step, both
// operands double-buffered through LDS-DMA.  For every map the source of (pixel p, slice k0) is
//     x + base(p) + koff(k0)      valid iff  0 <= py(p) + dy(k0) < H  and  0 <= px(p) + dx(k0) < W   (else the zero page)
// with base / py / px per lane (computed once) and koff / dy / dx uniform over the workgroup (advanced incrementally per
// slice: no division in the loop):
//   mode 1 (space to depth)   base = pixel (s oy, s ox);            koff = row * W C + r      (k0 = row * s C + r); always valid
//   mode 2 (subsample)        base = pixel (s oy + a, s ox + b);    koff = k0;                                       always valid
//   mode 3 (3x3 im2col)       base = pixel (s oy, s ox), py / px = its coordinates;  koff = ((ty-1) W + (tx-1)) C + c
//   mode 4 (parity class)     base = pixel (i, j), py / px = (i, j);  koff = (a (1-jy) W + b (1-jx)) C + c
// Fragment addresses are registers computed once plus immediates (slice loop unrolled by two: the buffer index is an
// immediate).  The unmapped launches use k_conv1x1_line below.
template <int TN>
__global__ __launch_bounds__(256) void k_conv1x1_mapped(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ w,
                                                        unsigned short* __restrict__ y, ConvGeom gm,
                                                        ConvEp ep) {
  constexpr int WM = TN == 128 ? 2 : 4, RW = 8 / WM, NB = TN / 32;
  constexpr int kABytes = kTM * 128, kBBytes = TN * 128, kBOff = 2 * kABytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wm = TN == 128 ? (wave >> 1) : wave, wn = TN == 128 ? (wave & 1) : 0;
  const int ntiles = gm.tiles_y;
  const int per = (ntiles + 7) / 8;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);     // XCD-aware tile order
  if (tile >= ntiles) return;
  const int n0 = blockIdx.y * TN;
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero16);
  const PixMap& im = gm.imap;
  const int mode = im.mode;

  f32x4 acc[RW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int r8 = lane >> 3, slot = lane & 7;
  const unsigned short* pa[4];
  int py[4], px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 4 * i) * 8 + r8;
    long long p = (long long)tile * kTM + r;
    if (p >= gm.npix) p = gm.npix - 1;           // rows past the last pixel re-read it (never stored)
    py[i] = px[i] = 0;
    if (mode == 0) {                             // plain input, mapped output (transposed convolution)
      pa[i] = x + (size_t)p * gm.Cin + ((slot ^ (r & 7)) << 3);
      continue;
    }
    const unsigned pu = (unsigned)p, t = pu / (unsigned)im.Wo;        // rows < 2^30 (launcher): 32-bit divisions
    const int ox = (int)(pu - t * (unsigned)im.Wo);
    const int b = (int)(t / (unsigned)im.Ho), oy = (int)(t - (unsigned)b * (unsigned)im.Ho);
    int y0, x0;
    if (mode == 4) { y0 = oy; x0 = ox; }
    else if (mode == 2) { y0 = im.s * oy + im.a; x0 = im.s * ox + im.b; }
    else { y0 = im.s * oy; x0 = im.s * ox; }
    pa[i] = x + ((size_t)(b * im.H + y0) * im.W + x0) * im.C + ((slot ^ (r & 7)) << 3);
    if (mode >= 3) { py[i] = y0; px[i] = x0; }
  }
  const unsigned short* pb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave + 4 * j) * 8 + r8;
    const size_t row = (size_t)min(n0 + n, gm.Cout - 1);     // channels past Cout re-read the last one (never stored)
    pb[j] = w + (mode == 4 ? row * 9 * im.C : row * gm.Cin) + ((slot ^ (n & 7)) << 3);
  }
  // slice state: (seg, c) = (row of the s x s block | tap | class tap, channel offset inside it)
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
      const int ty = im.a ? 2 * jy : 1, tx = im.b ? 2 * jx : 1;    // even coordinate -> tap 1, odd -> 0 then 2
      wk = (ty * 3 + tx) * im.C + c;
    } else {
      koff = c;
      wk = c;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = mode < 3 || ((unsigned)(py[i] + dy) < (unsigned)im.H && (unsigned)(px[i] + dx) < (unsigned)im.W);
      dma16(ok ? pa[i] + koff : zero, reinterpret_cast<unsigned short*>(smem + buf * kABytes + (wave + 4 * i) * 1024));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
      dma16(pb[j] + wk, reinterpret_cast<unsigned short*>(smem + kBOff + buf * kBBytes + (wave + 4 * j) * 1024));
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
    bf16x8 a[2][RW], bb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ti = 0; ti < RW; ++ti)
        a[ks][ti] = *reinterpret_cast<const bf16x8*>(smem + sa[ks] + buf * kABytes + ti * 2048);
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
        bb[ks][tj] = *reinterpret_cast<const bf16x8*>(smem + sb[ks] + buf * kBBytes + tj * 2048);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < RW; ++ti)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][ti], bb[ks][tj], acc[ti][tj], 0, 0, 0);
  };

  const int nchunks = gm.Cin / kKC;
  stage(0);
  __syncthreads();                 // drains the DMAs (vmcnt(0)) and publishes the tiles
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
  conv_store_tile<TN, 1, RW>(acc, reinterpret_cast<float*>(smem), tid, wm, wn, g, li, 0, tile * kTH, 0, n0, y, gm, ep, tile);
}

// ---- 3x3 forward / data gradient: straight-line tap loop --------------------------------------------------------------
// The nine taps of a 64-channel slice are unrolled so that every per-tap quantity is an instruction immediate.  SQ counters
// on the round-1 kernel, which looped over (slice, tap) at run time (tools/pmc_kernel.sh, 128 -> 128
// @180 x 180 x 4): per (tap, slice) a wave issued 32 MFMAs next to 123 other VALU and 93 scalar instructions -- 64-bit
// multiply-adds and exec-masked branches rebuilding the four weight-piece addresses, the 16 swizzled fragment addresses and
// the (slice, tap) split of the loop counter -- and MFMA busy was 23 % of the SIMD cycles.  Here
//   * the swizzled fragment address of halo row q0 + d is  sa[ks][d & 7] + 128 d  (16 registers computed once; the shift d
//     of (tile row, tap) goes into the ds_read offset field),
//   * a weight piece is  (uniform slice base) + (per-lane 32-bit offset computed once): output channels past Cout re-read
//     channel Cout - 1 (never stored), which removes the zero-page select from the loop,
//   * the halo pointers are computed once and advanced by 128 B per slice,
//   * the double buffers alternate through two register sets swapped per slice (nine taps: the parity flips).
template <int TN, int RW>
__global__ __launch_bounds__(256) void k_conv3x3_taps(const unsigned short* __restrict__ x,
                                                      const unsigned short* __restrict__ w,
                                                      unsigned short* __restrict__ y, ConvGeom gm, ConvEp ep) {
  constexpr int WM = TN == 128 ? 2 : 4;             // waves along the pixel rows; a wave owns RW rows of 16 pixels
  constexpr int TH = WM * RW, TM = TH * kTW;        // tile: TH x 16 output pixels
  constexpr int kHQ = kHW * (TH + 2), kHQP = (kHQ + 7) / 8 * 8, kAInstr = kHQP / 8;
  constexpr int NB = TN / 32;                       // weight pieces (8 channels x 128 B) per wave and slice
  constexpr int kBBytes = TN * 128, kABytes = kHQP * 128, kAOff = 2 * kBBytes;
  constexpr int kAPer = (kAInstr + 3) / 4;          // halo pieces per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero16);

  f32x4 acc[RW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int r8 = lane >> 3, slot = lane & 7;
  // halo pieces of this wave: source pointer of slice 0 and its per-slice step (0 for the out-of-image zero page)
  const unsigned short* pa[kAPer];
  int inca[kAPer];
#pragma unroll
  for (int i = 0; i < kAPer; ++i) {
    const int q = (wave + 4 * i) * 8 + r8;
    const int qy = q / kHW, qx = q - qy * kHW;
    const int gy = ty0 + qy - 1, gx = tx0 + qx - 1;
    const bool ok = q < kHQ && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
    pa[i] = ok ? x + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cin + ((slot ^ (q & 7)) << 3) : zero;
    inca[i] = ok ? kKC : 0;
  }
  auto stage_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kAPer; ++i) {
      if (wave + 4 * i < kAInstr)
        dma16(pa[i], reinterpret_cast<unsigned short*>(smem + kAOff + buf * kABytes + (wave + 4 * i) * 1024));
      pa[i] += inca[i];
    }
  };
  // weight pieces: byte offset of (channel row, swizzled slot) from the slice base
  unsigned voffb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave + 4 * j) * 8 + r8;
    const int nn = min(n0 + n, gm.Cout - 1);
    voffb[j] = (unsigned)(((size_t)nn * 9 * gm.Cin + ((slot ^ (n & 7)) << 3)) * 2);
  }
  auto stage_b = [&](int chunk, int tap, int buf) {
    const int te = ep.reverse_taps ? 8 - tap : tap;
    const char* wb = reinterpret_cast<const char*>(w) + ((size_t)te * gm.Cin + (size_t)chunk * kKC) * 2;
#pragma unroll
    for (int j = 0; j < NB; ++j)
      dma16(reinterpret_cast<const unsigned short*>(wb + voffb[j]),
            reinterpret_cast<unsigned short*>(smem + buf * kBBytes + (wave + 4 * j) * 1024));
  };
  // fragment addresses (bytes in LDS)
  const int q0 = RW * wm * kHW + li;
  unsigned sa[2][8], sb[2][2];
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
  __syncthreads();                 // drains the DMAs (vmcnt(0)) and publishes the tiles
  int adelta = kABytes;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int cpar = chunk & 1;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // next weight slice (and, at the start of a slice, the next halo) fly while this one is multiplied
      if (tap < 8) stage_b(chunk, tap + 1, ((tap + 1) & 1) ^ cpar);
      else if (chunk + 1 < nchunks) stage_b(chunk + 1, 0, cpar ^ 1);
      if (tap == 0 && chunk + 1 < nchunks) stage_a(cpar ^ 1);
      const int dy = tap / 3, dx = tap % 3;
      bf16x8 a[2][RW], bb[2][4];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int ti = 0; ti < RW; ++ti) {
          const int d = (ti + dy) * kHW + dx;
          a[ks][ti] = *reinterpret_cast<const bf16x8*>(smem + sa[ks][d & 7] + d * 128);
        }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
          bb[ks][tj] = *reinterpret_cast<const bf16x8*>(smem + sb[ks][tap & 1] + tj * 2048);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
          for (int ti = 0; ti < RW; ++ti)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][ti], bb[ks][tj], acc[ti][tj], 0, 0, 0);
      __syncthreads();
    }
    // nine taps: the weight double buffer ends a slice on the other parity; the halo buffer alternates per slice
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned t = sb[ks][0];
      sb[ks][0] = sb[ks][1];
      sb[ks][1] = t;
#pragma unroll
      for (int c = 0; c < 8; ++c) sa[ks][c] += adelta;
    }
    adelta = -adelta;
  }
  conv_store_tile<TN, 3, RW, TM>(acc, reinterpret_cast<float*>(smem), tid, wm, wn, g, li, b, ty0, tx0, n0, y, gm, ep,
                                 tile_lin);
}

// ---- the same tap kernel on v_mfma_f32_32x32x16_bf16 (round 4) ------------------------------------------------------------
// An MFMA-only loop sustains 1.33 PFLOP/s with the 16x16x32 instruction on this machine and 1.8 (random operands) - 2.4
// (constant) with 32x32x16 (profiles/r04_conv_bf16.md), so the tiles whose height allows 32-pixel blocks (RW even: a block =
// two image rows x 16 pixels) use it: the wave's (16 RW) x 64 sub-tile is (RW / 2) x 2 blocks of 32 x 32, D[pixel][channel].
// Same staging, barriers and epilogue as k_conv3x3_taps; what changes is the fragment geometry -- lane l reads 16 bytes of
// row (l & 31) at k-group 2 ks + (l >> 5) -- and with it the XOR swizzle: the 16 lanes a ds_read_b128 services together now
// share their k-group, so the slot is XORed with (halo column >> 1) & 7 / (channel >> 1) & 7 instead of the row index & 7
// (16 different rows then fall into 16 different 16-byte slots of the 256-byte bank row: 0 conflicts in SQ_LDS_BANK_CONFLICT).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- plain 1x1 (no pixel map): straight-line slice loop -------------------------------------------------------------------
// k_conv1x1_mapped rebuilds every DMA source through PixMap::off and every fragment address per 64-channel slice; for
// the unmapped 1x1 convolutions (the ResNet bottleneck convs: most 1x1 launches of the step) everything is affine in the
// slice index: pointers advance by 128 B per slice, fragment addresses are two registers per operand plus immediates, the
// slice loop is unrolled by two so the double-buffer index is an immediate as well.  Tile = 128 consecutive pixels; rows past
// the last pixel re-read pixel P - 1 and channels past Cout re-read channel Cout - 1 (neither is stored).
template <int TN>
__global__ __launch_bounds__(256) void k_conv1x1_line(const unsigned short* __restrict__ x,
                                                      const unsigned short* __restrict__ w,
                                                      unsigned short* __restrict__ y, ConvGeom gm, ConvEp ep) {
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
  const unsigned short* pa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 4 * i) * 8 + r8;
    long long p = (long long)tile * kTM + r;
    if (p >= gm.npix) p = gm.npix - 1;
    pa[i] = x + (size_t)p * gm.Cin + ((slot ^ (r & 7)) << 3);
  }
  const unsigned short* pb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = (wave + 4 * j) * 8 + r8;
    pb[j] = w + (size_t)min(n0 + n, gm.Cout - 1) * gm.Cin + ((slot ^ (n & 7)) << 3);
  }
  auto stage = [&](int buf) {      // next 64-channel slice of both operands
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(pa[i], reinterpret_cast<unsigned short*>(smem + buf * kABytes + (wave + 4 * i) * 1024));
      pa[i] += kKC;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      dma16(pb[j], reinterpret_cast<unsigned short*>(smem + kBOff + buf * kBBytes + (wave + 4 * j) * 1024));
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
    bf16x8 a[2][RW], bb[2][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int ti = 0; ti < RW; ++ti)
        a[ks][ti] = *reinterpret_cast<const bf16x8*>(smem + sa[ks] + buf * kABytes + ti * 2048);
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
        bb[ks][tj] = *reinterpret_cast<const bf16x8*>(smem + sb[ks] + buf * kBBytes + tj * 2048);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < RW; ++ti)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][ti], bb[ks][tj], acc[ti][tj], 0, 0, 0);
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
  conv_store_tile<TN, 1, RW>(acc, reinterpret_cast<float*>(smem), tid, wm, wn, g, li, 0, tile * kTH, 0, n0, y, gm, ep, tile);
}

// ---- weight gradient -----------------------------------------------------------------------------
//   dW[n][tap][c] = sum_p dy[p][n] * x[p + tap][c]           (p over all B*H*W output pixels)
// A GEMM whose reduction index is the pixel, i.e. both operands are K-STRIDED in their channels-last
// tensors.  Tiles are staged row-major ([pixel][channel], rows padded by 32 B) and the MFMA operand
// fragments are read with ds_read_b64_tr_b16: inside each 16-lane group, lane j points at the 8-byte
// piece [pixel k0 + (j>>2)][channel c0 + 4*(j&3)], and the hardware hands lane i the four values
// [k0 .. k0+3][c0 + i] -- the transpose costs no extra LDS pass.  A workgroup owns one
// (tap, 128 x CT output tile) and a slice of the pixels (64 per step, next step prefetched in
// registers, double-buffered LDS); the slices' partial sums are reduced in a fixed order by
// k_wgrad_sum (deterministic, no atomics).
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int kWP = 64;                      // pixels per step

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* p_lo, const unsigned short* p_hi) {
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p_lo);
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p_hi);
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int CT>   // channels of x per workgroup: 64 or 128
__global__ __launch_bounds__(256) void k_conv3x3_wgrad(const unsigned short* __restrict__ x,
                                                       const unsigned short* __restrict__ dy,
                                                       float* __restrict__ partial, ConvGeom gm,
                                                       int c_tiles, int n_tiles) {
  constexpr int LDN = 128 + 16, LDC = CT + 16;        // bf16 elements per LDS row (+32 B pad)
  constexpr int NU = kWP * 16 / 256;                  // 16-byte units per thread, dy tile (4)
  constexpr int CU = kWP * (CT / 8) / 256;            // ... x tile (2 or 4)
  constexpr int TJ = CT / 32;                         // 16-wide c tiles per wave (2 or 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* Ns = reinterpret_cast<unsigned short*>(smem);        // [2][kWP][LDN]  dy
  unsigned short* Cs = Ns + 2 * kWP * LDN;                               // [2][kWP][LDC]  shifted x
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  int combo = blockIdx.y;
  const int ct = combo % c_tiles; combo /= c_tiles;
  const int nt = combo % n_tiles;
  const int tap = combo / n_tiles;
  const int sy = tap / 3 - 1, sx = tap % 3 - 1;
  const int n0 = nt * 128, c0 = ct * CT;
  const long long P = (long long)gm.B * gm.H * gm.W;
  const int steps = (int)((P + kWP - 1) / kWP);

  f32x4 acc[4][TJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 rn[NU], rc[CU];
  auto fetch = [&](int step) {
    const long long p0 = (long long)step * kWP;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int u = tid + 256 * j, r = u >> 4, n8 = (u & 15) * 8;
      const long long p = p0 + r;
      rn[j] = make_uint4(0u, 0u, 0u, 0u);
      if (p < P && n0 + n8 < gm.Cout) rn[j] = *reinterpret_cast<const uint4*>(dy + (size_t)p * gm.Cout + n0 + n8);
    }
#pragma unroll
    for (int j = 0; j < CU; ++j) {
      const int u = tid + 256 * j, r = u / (CT / 8), c8 = (u - r * (CT / 8)) * 8;
      const long long p = p0 + r;
      rc[j] = make_uint4(0u, 0u, 0u, 0u);
      if (p < P && c0 + c8 < gm.Cin) {
        const unsigned pu = (unsigned)p, tq = pu / (unsigned)gm.W;       // B * H * W fits an int
        const int px = (int)(pu - tq * (unsigned)gm.W), py = (int)(tq % (unsigned)gm.H);
        if (px + sx >= 0 && px + sx < gm.W && py + sy >= 0 && py + sy < gm.H)
          rc[j] = *reinterpret_cast<const uint4*>(x + (size_t)(p + (long long)sy * gm.W + sx) * gm.Cin + c0 + c8);
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const int u = tid + 256 * j;
      *reinterpret_cast<uint4*>(Ns + (buf * kWP + (u >> 4)) * LDN + (u & 15) * 8) = rn[j];
    }
#pragma unroll
    for (int j = 0; j < CU; ++j) {
      const int u = tid + 256 * j, r = u / (CT / 8);
      *reinterpret_cast<uint4*>(Cs + (buf * kWP + r) * LDC + (u - r * (CT / 8)) * 8) = rc[j];
    }
  };

  int step = blockIdx.x;
  if (step < steps) {
    fetch(step);
    commit(0);
  }
  __syncthreads();
  int buf = 0;
  for (; step < steps; step += gridDim.x) {
    const bool more = step + (int)gridDim.x < steps;
    if (more) fetch(step + gridDim.x);
    {
      // lane j of a 16-lane group addresses pixel row (j>>2), 8-byte piece 4*(j&3) of a 16-wide tile
      const unsigned short* nb = Ns + (buf * kWP + 8 * g + (li >> 2)) * LDN + 64 * wm + 4 * (li & 3);
      const unsigned short* cb = Cs + (buf * kWP + 8 * g + (li >> 2)) * LDC + (CT / 2) * wn + 4 * (li & 3);
#pragma unroll
      for (int ks = 0; ks < kWP / 32; ++ks) {
        bf16x8 a[4];
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          a[ti] = tr_frag(nb + 32 * ks * LDN + 16 * ti, nb + (32 * ks + 4) * LDN + 16 * ti);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
          const bf16x8 bb = tr_frag(cb + 32 * ks * LDC + 16 * tj, cb + (32 * ks + 4) * LDC + 16 * tj);
#pragma unroll
          for (int ti = 0; ti < 4; ++ti)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti], bb, acc[ti][tj], 0, 0, 0);
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial[slice][n][tap][c]; D layout: lane holds column c = li, rows n = 4g + r
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int c = c0 + (CT / 2) * wn + 16 * tj + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 64 * wm + 16 * ti + 4 * g + r;
        if (n < gm.Cout && c < gm.Cin)
          partial[(((size_t)blockIdx.x * gm.Cout + n) * 9 + tap) * gm.Cin + c] = acc[ti][tj][r];
      }
    }
}

// Large feature maps: the tap-shared, LDS-DMA staged variant.  A workgroup owns a 64 (dy channels) x 64
// (x channels) tile of dW for ALL nine taps and walks a slice of the 8 x 16 pixel tiles: per pixel tile
// the x halo (10 x 18 pixels) and the dy tile are staged ONCE (global_load_lds_dwordx4, double-buffered,
// one barrier per tile) and serve the nine shifted GEMMs -- the per-tap kernel above re-stages both
// operands for every tap through registers.  Rows are unpadded 128 B; their 32-byte pieces are
// XOR-swizzled with f(row) = bit1 | bit3 << 1 on the SOURCE address, which keeps the transposing fragment
// reads (8 rows x 32 B per half-wave, rows {a..a+3, a+8..a+11} for ANY a, i.e. any tap shift) conflict
// free.  Wave w owns x channels 16w..16w+15: per 32-pixel step 4 dy fragments + 9 x fragments feed 36
// MFMAs (0.36 fragment reads per MFMA; 144 accumulator registers).
__device__ __forceinline__ int wg_fsw(int r) { return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1; }

// The compiler makes every ds_read_tr builtin wait for ALL outstanding LDS-DMA (vmcnt(0)), so the next
// tile's DMAs would never overlap this tile's MFMAs.  The fragment reads are therefore issued as
// inline asm (invisible to the waitcnt pass) and completed by an explicit lgkmcnt wait that is tied to
// the fragment registers.
__device__ __forceinline__ v4s tr_issue(unsigned lds_byte_addr) {
  v4s r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr));
  return r;
}
__device__ __forceinline__ bf16x8 cat8(v4s lo, v4s hi) {
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Straight-line addressing.  SQ counters on the first version of this kernel, which rebuilt the 26 swizzled fragment
// addresses of every 32-pixel step and the ten DMA source addresses of every tile from scratch: 6.1 VALU + 1.4 scalar
// instructions per MFMA, MFMA busy 33 %.  Here
//   * a dy fragment address is  sd[ti] + 4096 ks (+ 512 for the upper half): the swizzle f(p) uses bits 1 and 3 of the pixel
//     index, which the 32-pixel step does not touch;
//   * an x fragment of halo row L + C (L = the lane's row, C = 36 ks + 18 dy + dx (+ 4)) is  sx[C & 15] + 128 C: sixteen
//     registers cover every (step, tap), the shift goes into the ds_read offset field;
//   * a DMA source is (tile base: scalar) + (per-lane offset computed once) under four range compares.
__global__ __launch_bounds__(256, 2) void k_conv3x3_wgrad_taps(const unsigned short* __restrict__ x,
                                                               const unsigned short* __restrict__ dy,
                                                               float* __restrict__ partial, ConvGeom gm,
                                                               int c_tiles, int tiles_per_slice) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kXBytes = kHQP * kKC * 2, kDBytes = kTM * kKC * 2, kDOff = 2 * kXBytes;
  constexpr int kXPer = (kAInstr + 3) / 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int r8 = lane >> 3, slot = lane & 7;
  const int ct = blockIdx.y % c_tiles, nt = blockIdx.y / c_tiles;
  const int n0 = nt * 64, c0 = ct * 64;
  const int per_img = gm.tiles_x * gm.tiles_y, ntiles = gm.B * per_img;
  const int t_begin = blockIdx.x * tiles_per_slice, t_end = min(ntiles, t_begin + tiles_per_slice);
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero16);

  // DMA pieces of this wave: halo pixel (qy, qx) / tile pixel (py, px) and the element offset from the tile's first pixel
  int xqy[kXPer], xqx[kXPer], xoff[kXPer], dpy[4], dpx[4], doff[4];
#pragma unroll
  for (int i = 0; i < kXPer; ++i) {
    const int q = (wave + 4 * i) * 8 + r8, qy = q / kHW;
    xqx[i] = q - qy * kHW;
    xoff[i] = ((qy - 1) * gm.W + (xqx[i] - 1)) * gm.Cin + c0 + ((slot ^ wg_fsw(q)) << 3);
    xqy[i] = q < kHQ ? qy : 1 << 20;               // rows past the halo: never valid
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = (wave + 4 * j) * 8 + r8;
    const int n = n0 + ((slot ^ wg_fsw(p)) << 3);
    dpy[j] = n < gm.Cout ? p >> 4 : 1 << 20;      // channel pieces past Cout: never valid
    dpx[j] = p & 15;
    doff[j] = ((p >> 4) * gm.W + (p & 15)) * gm.Cout + n;
  }
  auto stage = [&](int tile, int buf) {
    const int b = tile / per_img;
    const int rem = tile - b * per_img;
    const int ty0 = (rem / gm.tiles_x) * kTH, tx0 = (rem % gm.tiles_x) * kTW;
    const size_t pix0 = (size_t)(b * gm.H + ty0) * gm.W + tx0;
    const unsigned short* xb = x + pix0 * gm.Cin;
    const unsigned short* db = dy + pix0 * gm.Cout;
#pragma unroll
    for (int i = 0; i < kXPer; ++i) {
      if (wave + 4 * i >= kAInstr) continue;
      const int gy = ty0 + xqy[i] - 1, gx = tx0 + xqx[i] - 1;
      const unsigned short* src = (gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W) ? xb + xoff[i] : zero;
      dma16(src, reinterpret_cast<unsigned short*>(smem + buf * kXBytes + (wave + 4 * i) * 1024));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned short* src = (ty0 + dpy[j] < gm.H && tx0 + dpx[j] < gm.W) ? db + doff[j] : zero;
      dma16(src, reinterpret_cast<unsigned short*>(smem + kDOff + buf * kDBytes + (wave + 4 * j) * 1024));
    }
  };

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses (LDS bytes) of buffer 0
  const int sub = (li & 3) >> 1, half = (li & 1) << 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned sd[4], sx[16];
  {
    const int p0 = 8 * g + (li >> 2);
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) sd[ti] = lds0 + kDOff + 2 * (p0 * kKC + (((2 * ti + sub) ^ wg_fsw(p0)) << 3) + half);
    const int L = (g >> 1) * kHW + 8 * (g & 1) + (li >> 2);
#pragma unroll
    for (int c = 0; c < 16; ++c) sx[c] = lds0 + 2 * (L * kKC + (((2 * wave + sub) ^ wg_fsw(L + c)) << 3) + half);
  }

  if (t_begin < t_end) stage(t_begin, 0);
  __syncthreads();
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    if (tile + 1 < t_end) stage(tile + 1, buf ^ 1);
#pragma unroll
    for (int ks = 0; ks < kTM / 32; ++ks) {
      v4s al[4], ah[4], bl[9], bh[9];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(al[ti]) : "v"(sd[ti]), "n"(ks * 4096));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ah[ti]) : "v"(sd[ti]), "n"(ks * 4096 + 512));
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int C0 = 2 * ks * kHW + (tap / 3) * kHW + tap % 3, C1 = C0 + 4;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bl[tap]) : "v"(sx[C0 & 15]), "n"(C0 * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bh[tap]) : "v"(sx[C1 & 15]), "n"(C1 * 128));
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(al[3]), "+v"(ah[3]));
      asm volatile("" : "+v"(bl[0]), "+v"(bh[0]), "+v"(bl[1]), "+v"(bh[1]), "+v"(bl[2]), "+v"(bh[2]), "+v"(bl[3]),
                        "+v"(bh[3]), "+v"(bl[4]), "+v"(bh[4]));
      asm volatile("" : "+v"(bl[5]), "+v"(bh[5]), "+v"(bl[6]), "+v"(bh[6]), "+v"(bl[7]), "+v"(bh[7]), "+v"(bl[8]),
                        "+v"(bh[8]));
      bf16x8 a[4];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) a[ti] = cat8(al[ti], ah[ti]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bf16x8 bb = cat8(bl[tap], bh[tap]);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          acc[tap][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti], bb, acc[tap][ti], 0, 0, 0);
      }
    }
    __syncthreads();
    // the other buffer: toggle the 20 base addresses
    const int dxb = buf ? -kXBytes : kXBytes, ddb = buf ? -kDBytes : kDBytes;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) sd[ti] += ddb;
#pragma unroll
    for (int c = 0; c < 16; ++c) sx[c] += dxb;
    buf ^= 1;
  }
  // partial[slice][n][tap][c]; D layout: lane holds column c = li, rows n = 4g + r
  const int c = c0 + 16 * wave + li;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * ti + 4 * g + r;
        if (n < gm.Cout) partial[(((size_t)blockIdx.x * gm.Cout + n) * 9 + tap) * gm.Cin + c] = acc[tap][ti][r];
      }
}
constexpr size_t kWgradDmaLds = 2 * ((size_t)kHQP + kTM) * kKC * 2;

// pixel-tile slices of the DMA kernel: <= 2 workgroups per CU over (slices x 64x64 output tiles)
int wgrad_dma_slices(int B, int H, int W, int Cin, int Cout, int* tiles_per_slice) {
  const int ntiles = B * ud_div_up(W, kTW) * ud_div_up(H, kTH);
  const int combos = ud_div_up(Cout, 64) * (Cin / 64);
  int s = 512 / combos;         // all workgroups resident at once (2 per CU): no second round
  if (s > ntiles) s = ntiles;
  if (s < 1) s = 1;
  const int per = (ntiles + s - 1) / s;
  *tiles_per_slice = per;
  return (ntiles + per - 1) / per;
}
// the per-tap kernel keeps the tiny maps (<= 4096 pixels in total: too few pixel tiles to slice)
bool wgrad_use_dma(int B, int H, int W) { return (long long)B * H * W > 4096; }

// ---- 1x1 convolutions: weight gradient ---------------------------------------------------------------
//   dW[n][c] = sum_p dy[p][n] * x[p][c]        (p over all B*H*W pixels; ResNet bottleneck 1x1 convs,
//   unidistill/layers/blocks_2d/mmdet3d/resnet.py via mmcv's ResNet in the reference)
// The pixel-reduced GEMM that a BLAS library runs on 16 CUs (M x N = Cout x Cin has few tiles and the
// library does not split K): here (pixel slice) x (NT x CT output tile) workgroups, 64-pixel steps
// staged by LDS-DMA (double-buffered, one barrier per step), transposing fragment reads issued as
// inline asm (see tr_issue), 2 x 2 waves.  Unpadded rows; 32-byte pieces swizzled on the source
// address: f(r) = (r & 3) | ((r >> 1) & 4) for 256-byte rows, bit1 | bit3 << 1 for 128-byte rows.
template <int NT, int CT>
__global__ __launch_bounds__(256) void k_conv1x1_wgrad_dma(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ dy,
                                                           float* __restrict__ partial, long long P, int Cin,
                                                           int Cout, int c_tiles, int steps_per_slice,
                                                           PixMap xmap, PixMap ymap) {
  constexpr int SN = NT / 8, SC = CT / 8;                 // 16-byte slots per row
  constexpr int RN = 64 / SN, RC = 64 / SC;               // rows per 1-KiB piece
  constexpr int PN = 64 / RN, PC = 64 / RC;               // pieces per 64-row tile
  constexpr int TI = NT / 32, TJ = CT / 32;               // 16-wide tiles per wave (2 x 2 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* Ns = reinterpret_cast<unsigned short*>(smem);        // [2][64][NT]  dy
  unsigned short* Cs = Ns + 2 * 64 * NT;                                 // [2][64][CT]  x
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int ct = blockIdx.y % c_tiles, nt = blockIdx.y / c_tiles;
  const int n0 = nt * NT, c0 = ct * CT;
  const int steps = (int)((P + 63) / 64);
  const int s_begin = blockIdx.x * steps_per_slice, s_end = min(steps, s_begin + steps_per_slice);
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_zero16);
  auto fsw = [](int r, int slots) -> int {
    return slots == 16 ? (((r & 3) | ((r >> 1) & 4)) << 1) : ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1);
  };
  auto stage = [&](int step, int buf) {
#pragma unroll
    for (int j = 0; j < PN / 4; ++j) {
      const int piece = wave + 4 * j, r = piece * RN + lane / SN, slot = lane % SN;
      const long long p = (long long)step * 64 + r;
      const int n = n0 + ((slot ^ fsw(r, SN)) << 3);
      const unsigned short* src = zero;
      if (p < P && n < Cout) src = dy + ymap.off(p, n, Cout);
      dma16(src, Ns + (buf * 64 + piece * RN) * NT);
    }
#pragma unroll
    for (int j = 0; j < PC / 4; ++j) {
      const int piece = wave + 4 * j, r = piece * RC + lane / SC, slot = lane % SC;
      const long long p = (long long)step * 64 + r;
      const unsigned short* src = zero;
      if (p < P) {
        const size_t o = xmap.off(p, c0 + ((slot ^ fsw(r, SC)) << 3), Cin);
        if (o != kNoPixel) src = x + o;
      }
      dma16(src, Cs + (buf * 64 + piece * RC) * CT);
    }
  };
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (s_begin < s_end) stage(s_begin, 0);
  __syncthreads();
  int buf = 0;
  const int sub = (li & 3) >> 1, half = (li & 1) << 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  for (int step = s_begin; step < s_end; ++step) {
    if (step + 1 < s_end) stage(step + 1, buf ^ 1);
    const unsigned nbuf = lds0 + buf * (64 * NT * 2);
    const unsigned cbuf = lds0 + (2 * 64 * NT + buf * 64 * CT) * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r0 = 32 * ks + 8 * g + (li >> 2), r1 = r0 + 4;
      v4s al[TI], ah[TI], bl[TJ], bh[TJ];
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        const int slot = 2 * (TI * wm + ti) + sub;
        al[ti] = tr_issue(nbuf + 2 * (r0 * NT + ((slot ^ fsw(r0, SN)) << 3) + half));
        ah[ti] = tr_issue(nbuf + 2 * (r1 * NT + ((slot ^ fsw(r1, SN)) << 3) + half));
      }
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const int slot = 2 * (TJ * wn + tj) + sub;
        bl[tj] = tr_issue(cbuf + 2 * (r0 * CT + ((slot ^ fsw(r0, SC)) << 3) + half));
        bh[tj] = tr_issue(cbuf + 2 * (r1 * CT + ((slot ^ fsw(r1, SC)) << 3) + half));
      }
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) asm volatile("" : "+v"(al[ti]), "+v"(ah[ti]));
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) asm volatile("" : "+v"(bl[tj]), "+v"(bh[tj]));
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const bf16x8 bb = cat8(bl[tj], bh[tj]);
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(al[ti], ah[ti]), bb, acc[ti][tj], 0, 0, 0);
      }
    }
    __syncthreads();
    buf ^= 1;
  }
  // partial[slice][n][c]; D layout: lane holds column c = li, rows n = 4g + r
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      const int c = c0 + 16 * (TJ * wn + tj) + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * (TI * wm + ti) + 4 * g + r;
        if (n < Cout) partial[((size_t)blockIdx.x * Cout + n) * Cin + c] = acc[ti][tj][r];
      }
    }
}

struct Wgrad1x1Plan {
  int nt, ct, n_tiles, c_tiles, slices, steps_per_slice;
};
Wgrad1x1Plan wgrad1x1_plan(long long P, int Cin, int Cout) {
  Wgrad1x1Plan pl;
  pl.nt = Cout >= 128 ? 128 : 64;
  pl.ct = Cin % 128 == 0 ? 128 : 64;
  pl.n_tiles = ud_div_up(Cout, pl.nt);
  pl.c_tiles = Cin / pl.ct;
  const int steps = (int)((P + 63) / 64);
  int s = 512 / (pl.n_tiles * pl.c_tiles);          // all workgroups resident at once (2 per CU)
  if (s > steps) s = steps;
  if (s < 1) s = 1;
  pl.steps_per_slice = (steps + s - 1) / s;
  pl.slices = (steps + pl.steps_per_slice - 1) / pl.steps_per_slice;
  return pl;
}
template <int NT, int CT>
int launch_wgrad1x1(const void* x, const void* dy, float* partial, long long P, int Cin, int Cout,
                    const Wgrad1x1Plan& pl, const PixMap& xmap, const PixMap& ymap, hipStream_t stream) {
  constexpr size_t lds = (size_t)2 * 64 * (NT + CT) * 2;
  static UdDeviceOnce set;
  if (const unsigned long long set_bit = set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_wgrad_dma<NT, CT>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    set.mark(set_bit);
  }
  k_conv1x1_wgrad_dma<NT, CT><<<dim3(pl.slices, pl.n_tiles * pl.c_tiles), 256, lds, stream>>>(
      (const unsigned short*)x, (const unsigned short*)dy, partial, P, Cin, Cout, pl.c_tiles, pl.steps_per_slice,
      xmap, ymap);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// pixel slices so that (slices x taps x output tiles) is ~600 workgroups
int wgrad_slices(int Cin, int Cout, long long P, int* ct_width) {
  const int CT = (Cin % 128 == 0) ? 128 : 64;
  *ct_width = CT;
  const int combos = 9 * ud_div_up(Cout, 128) * ud_div_up(Cin, CT);
  long long s = (600 + combos - 1) / combos;
  const long long steps = (P + kWP - 1) / kWP;
  if (s > steps) s = steps;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

// stats / stats_bytes / slices_out: optional per-tile BatchNorm partials (see conv_store_tile); the tile count depends on the
// tile height picked below, so it is reported back
static int conv3x3_impl(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout,
                        const float* bias, const float* scale, const float* shift, const void* residual, int relu,
                        float* stats, size_t stats_bytes, int* slices_out, ud_stream_t stream_) {
  if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 8 != 0) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom gm{B, H, W, Cin, Cout, ud_div_up(W, kTW), ud_div_up(H, kTH), (long long)B * H * W, PixMap{}, PixMap{}};
  ConvEp ep{bias, scale, shift, reinterpret_cast<const unsigned short*>(residual), relu & 1, (relu >> 1) & 1, stats};
  static UdDeviceOnce attr_set;
  static int force_rw = 0;         // UD_CONV_RW=n: pixel rows per wave (timing experiments only)
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
#define UD_TAPS_ATTR(TN, RW)                                                                                          \
  UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_taps<TN, RW>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                 (int)conv_taps_smem_bytes(TN, RW)))
    UD_TAPS_ATTR(128, 4); UD_TAPS_ATTR(128, 3); UD_TAPS_ATTR(128, 2);
    UD_TAPS_ATTR(64, 2); UD_TAPS_ATTR(64, 1);
#undef UD_TAPS_ATTR
    if (const char* r = getenv("UD_CONV_RW")) force_rw = atoi(r);
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("conv2d.k_conv3x3", stream);
  const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
  const unsigned short* ws = reinterpret_cast<const unsigned short*>(w);
  unsigned short* ys = reinterpret_cast<unsigned short*>(y);
  // 64-wide output tiles when Cout <= 64, and on small maps where 128-wide tiles would leave CUs idle
  const bool narrow = Cout <= 64 || B * gm.tiles_x * gm.tiles_y * ud_div_up(Cout, 128) <= 256;
  const int ntn = ud_div_up(Cout, narrow ? 64 : 128);
  // Pixel rows per wave (tile height = waves x rows).  A CU works through ceil(workgroups / 256) tiles, two at a time
  // sharing its MFMA pipes, so a launch takes about ceil(WGs / 256) x (rows + 1) -- the +1 is what a tile costs whatever
  // its height (prologue, first DMA round trip, epilogue: ~10 k of the 28 k cycles of an 8-row tile at Cin = 128).  Checked
  // against the timings of tools/time_conv2d.py with UD_CONV_RW = 2 / 3 / 4 on all eleven shapes.
  const int wm = narrow ? 4 : 2;
  int rw = narrow ? 2 : 4;
  {
    long long best = -1;
    for (int r = narrow ? 2 : 4; r >= (narrow ? 1 : 2); --r) {
      const long long wgs = (long long)B * gm.tiles_x * ud_div_up(H, wm * r) * ntn;
      const long long cost = ((wgs + 255) / 256) * (r + 1);
      if (best < 0 || cost < best) best = cost, rw = r;
    }
    if (force_rw) rw = force_rw < (narrow ? 1 : 2) ? (narrow ? 1 : 2) : force_rw > (narrow ? 2 : 4) ? (narrow ? 2 : 4) : force_rw;
  }
  gm.tiles_y = ud_div_up(H, wm * rw);
  if (stats) {
    const size_t need = (size_t)B * gm.tiles_x * gm.tiles_y * Cout * 2 * sizeof(float);
    if (stats_bytes < need || !slices_out) return UD_ERR_WORKSPACE;
    *slices_out = B * gm.tiles_x * gm.tiles_y;
  }
  const dim3 grid((B * gm.tiles_x * gm.tiles_y + 7) / 8 * 8, ntn);
#define UD_TAPS_LAUNCH(TN, RW)                                                                                        \
  k_conv3x3_taps<TN, RW><<<grid, 256, conv_taps_smem_bytes(TN, RW), stream>>>(xs, ws, ys, gm, ep)
  if (narrow) {
    if (rw == 1) UD_TAPS_LAUNCH(64, 1);
    else UD_TAPS_LAUNCH(64, 2);
  } else {
    if (rw == 3) UD_TAPS_LAUNCH(128, 3);
    else if (rw == 2) UD_TAPS_LAUNCH(128, 2);
    else UD_TAPS_LAUNCH(128, 4);
  }
#undef UD_TAPS_LAUNCH
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_conv3x3_nhwc_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cin,
                                    int Cout, const float* bias, const float* scale,
                                    const float* shift, const void* residual, int relu,
                                    ud_stream_t stream) {
  return conv3x3_impl(x, w, y, B, H, W, Cin, Cout, bias, scale, shift, residual, relu, nullptr, 0, nullptr, stream);
}

// Upper bound of the partial-statistics buffer of ud_conv3x3_bnstats_nhwc_* (the smallest tile is 4 rows x 16 pixels).
extern "C" size_t ud_conv3x3_bnstats_bytes(int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  return ud_align_up((size_t)B * ud_div_up(W, kTW) * ud_div_up(H, 4) * Cout * 2 * sizeof(float));
}
extern "C" size_t ud_conv1x1_bnstats_bytes(int64_t P, int Cout) {
  if (P <= 0 || Cout <= 0) return 0;
  return ud_align_up((size_t)((P + kTM - 1) / kTM) * Cout * 2 * sizeof(float));
}

// The convolution (+ bias) AND the first pass of the training-mode BatchNorm that follows it: partial[slice][Cout][2] =
// per-tile (sum, sum of squares) of the stored bf16 values, *slices = number of tiles written (host int).  Finish with
// ud_bn_stats_from_partials.
extern "C" int ud_conv3x3_bnstats_nhwc_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cin,
                                            int Cout, const float* bias, float* partial, size_t partial_bytes,
                                            int* slices, ud_stream_t stream) {
  if (!partial || !slices) return UD_ERR_INVALID_ARG;
  return conv3x3_impl(x, w, y, B, H, W, Cin, Cout, bias, nullptr, nullptr, nullptr, 0, partial, partial_bytes, slices,
                      stream);
}

static int conv1x1_impl(const void* x, const void* w, void* y, int64_t P, int Cin, int Cout,
                        const float* bias, const float* scale, const float* shift,
                        const void* residual, int relu, const PixMap& imap, const PixMap& omap,
                        ud_stream_t stream_, float* stats = nullptr, size_t stats_bytes = 0, int* slices_out = nullptr) {
  if (!x || !w || !y || P <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 8 != 0 || P > (int64_t)1 << 30) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  // pixels as a [ceil(P/16)][16] image: 8 x 16 tiles of 128 consecutive pixels, no halo
  const int H = (int)((P + kTW - 1) / kTW);
  ConvGeom gm{1, H, kTW, Cin, Cout, 1, ud_div_up(H, kTH), (long long)P, imap, omap};
  ConvEp ep{bias, scale, shift, reinterpret_cast<const unsigned short*>(residual), relu & 1, 0, stats};
  if (stats) {
    if (stats_bytes < (size_t)gm.tiles_y * Cout * 2 * sizeof(float) || !slices_out) return UD_ERR_WORKSPACE;
    *slices_out = gm.tiles_y;
  }
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_mapped<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)conv_smem_bytes(128)));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_mapped<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)conv_smem_bytes(64)));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_line<128>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)conv_smem_bytes(128)));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1_line<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)conv_smem_bytes(64)));
    attr_set.mark(attr_set_bit);
  }
  const int ntiles = gm.tiles_y;
  const int gx = (ntiles + 7) / 8 * 8;
  UdProfScope prof("conv2d.k_conv1x1", stream);
  const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
  const unsigned short* ws = reinterpret_cast<const unsigned short*>(w);
  unsigned short* ys = reinterpret_cast<unsigned short*>(y);
  // 64-wide output tiles when Cout <= 64, and on small maps where 128-wide tiles would leave CUs idle
  const bool narrow = Cout <= 64 || ntiles * ud_div_up(Cout, 128) <= 256;
  const dim3 grid(gx, ud_div_up(Cout, narrow ? 64 : 128));
  const size_t lds = conv_smem_bytes(narrow ? 64 : 128);
  if (imap.mode == 0 && omap.mode == 0) {
    if (narrow) k_conv1x1_line<64><<<grid, 256, lds, stream>>>(xs, ws, ys, gm, ep);
    else k_conv1x1_line<128><<<grid, 256, lds, stream>>>(xs, ws, ys, gm, ep);
  } else {
    if (narrow) k_conv1x1_mapped<64><<<grid, 256, lds, stream>>>(xs, ws, ys, gm, ep);
    else k_conv1x1_mapped<128><<<grid, 256, lds, stream>>>(xs, ws, ys, gm, ep);
  }
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_conv1x1_nhwc_bf16(const void* x, const void* w, void* y, int64_t P, int Cin, int Cout,
                                    const float* bias, const float* scale, const float* shift,
                                    const void* residual, int relu, ud_stream_t stream) {
  return conv1x1_impl(x, w, y, P, Cin, Cout, bias, scale, shift, residual, relu, PixMap{}, PixMap{}, stream);
}

extern "C" int ud_conv1x1_bnstats_nhwc_bf16(const void* x, const void* w, void* y, int64_t P, int Cin, int Cout,
                                            const float* bias, float* partial, size_t partial_bytes, int* slices,
                                            ud_stream_t stream) {
  if (!partial || !slices) return UD_ERR_INVALID_ARG;
  return conv1x1_impl(x, w, y, P, Cin, Cout, bias, nullptr, nullptr, nullptr, 0, PixMap{}, PixMap{}, stream, partial,
                      partial_bytes, slices);
}

// 1x1 kernel over a mapped input and / or output (see PixMap): conv k = s / stride s, transposed conv k = s / stride s,
// 1x1 / stride s -- forward and data gradient of all three.  in_map / out_map: 9 ints {mode, s, Ho, Wo, H, W, C, a, b} or
// NULL for a plain [P][K] matrix (9 ints: {mode, s, Ho, Wo, H, W, C, a, b}); no bias / BatchNorm / residual epilogue.
extern "C" int ud_conv1x1_mapped_nhwc_bf16(const void* x, const void* w, void* y, int64_t P, int Cin, int Cout,
                                           const int* in_map, const int* out_map, ud_stream_t stream) {
  PixMap im, om;
  if (!map_from_ints(in_map, &im, 64) || !map_from_ints(out_map, &om, 64)) return UD_ERR_INVALID_ARG;
  if ((im.mode && im.C % 8) || (om.mode && om.C % 8)) return UD_ERR_INVALID_ARG;
  if (im.mode == 1 && (im.s * im.C) % 64 != 0) return UD_ERR_UNSUPPORTED;   // a 64-channel slice must not straddle dy
  if (im.mode == 1 && im.s * im.s * im.C != Cin) return UD_ERR_INVALID_ARG;
  if (im.mode == 2 && im.C != Cin) return UD_ERR_INVALID_ARG;
  if (im.mode == 3 && 9 * im.C != Cin) return UD_ERR_INVALID_ARG;
  if (im.mode == 4 && (1 + im.a) * (1 + im.b) * im.C != Cin) return UD_ERR_INVALID_ARG;
  if (om.mode >= 3) return UD_ERR_UNSUPPORTED;                               // gather only
  if (om.mode == 1 && om.s * om.s * om.C != Cout) return UD_ERR_INVALID_ARG;
  if (om.mode == 2 && om.C != Cout) return UD_ERR_INVALID_ARG;
  return conv1x1_impl(x, w, y, P, Cin, Cout, nullptr, nullptr, nullptr, nullptr, 0, im, om, stream);
}

extern "C" size_t ud_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  int ctw;
  const int S = (wgrad_use_dma(B, H, W) && Cin % 64 == 0) ? wgrad_dma_slices(B, H, W, Cin, Cout, &ctw)
                                                          : wgrad_slices(Cin, Cout, (long long)B * H * W, &ctw);
  return ud_align_up((size_t)S * Cout * 9 * Cin * sizeof(float));
}

extern "C" int ud_conv3x3_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int B, int H, int W,
                                          int Cin, int Cout, void* workspace, size_t workspace_bytes,
                                          ud_stream_t stream_) {
  if (!x || !dy || !dw || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Cin % 64 != 0 || Cout % 8 != 0) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_conv3x3_wgrad_workspace_bytes(B, H, W, Cin, Cout)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = reinterpret_cast<float*>(workspace);
  const size_t n = (size_t)Cout * 9 * Cin;
  if (wgrad_use_dma(B, H, W)) {
    ConvGeom gd{B, H, W, Cin, Cout, ud_div_up(W, kTW), ud_div_up(H, kTH), (long long)B * H * W};
    int per;
    const int S = wgrad_dma_slices(B, H, W, Cin, Cout, &per);
    static UdDeviceOnce set_dma;
    if (const unsigned long long set_dma_bit = set_dma.pending()) {
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_taps, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)kWgradDmaLds));
      set_dma.mark(set_dma_bit);
    }
    UdProfScope prof("conv2d.k_wgrad_dma", stream);
    k_conv3x3_wgrad_taps<<<dim3(S, ud_div_up(Cout, 64) * (Cin / 64)), 256, kWgradDmaLds, stream>>>(
        (const unsigned short*)x, (const unsigned short*)dy, partial, gd, Cin / 64, per);
    UD_LAUNCH_CHECK();
    k_wgrad_sum<<<ud_div_up((long long)(n / 4), 64), 256, 0, stream>>>(partial, S, n, dw);
    UD_LAUNCH_CHECK();
    return UD_OK;
  }
  ConvGeom gm{B, H, W, Cin, Cout, 0, 0, (long long)B * H * W};
  int CT;
  const int S = wgrad_slices(Cin, Cout, (long long)B * H * W, &CT);
  const int c_tiles = ud_div_up(Cin, CT), n_tiles = ud_div_up(Cout, 128);
  UdProfScope prof("conv2d.k_wgrad", stream);
  const dim3 grid(S, 9 * n_tiles * c_tiles);
  if (CT == 128) {
    const size_t lds = (size_t)2 * kWP * (144 + 144) * 2;
    static UdDeviceOnce set128;
    if (const unsigned long long set128_bit = set128.pending()) {
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wgrad<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      set128.mark(set128_bit);
    }
    k_conv3x3_wgrad<128><<<grid, 256, lds, stream>>>((const unsigned short*)x, (const unsigned short*)dy, partial, gm, c_tiles, n_tiles);
  } else {
    const size_t lds = (size_t)2 * kWP * (144 + 80) * 2;
    k_conv3x3_wgrad<64><<<grid, 256, lds, stream>>>((const unsigned short*)x, (const unsigned short*)dy, partial, gm, c_tiles, n_tiles);
  }
  UD_LAUNCH_CHECK();
  k_wgrad_sum<<<ud_div_up((long long)(n / 4), 64), 256, 0, stream>>>(partial, S, n, dw);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" size_t ud_conv1x1_wgrad_workspace_bytes(int64_t P, int Cin, int Cout) {
  if (P <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 != 0) return 0;
  const Wgrad1x1Plan pl = wgrad1x1_plan(P, Cin, Cout);
  return ud_align_up((size_t)pl.slices * Cout * Cin * sizeof(float));
}

static int wgrad1x1_impl(const void* x, const void* dy, float* dw, int64_t P, int Cin, int Cout, void* workspace,
                         size_t workspace_bytes, const PixMap& xmap, const PixMap& ymap, ud_stream_t stream_) {
  if (!x || !dy || !dw || P <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Cin % 64 != 0 || Cout % 8 != 0) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_conv1x1_wgrad_workspace_bytes(P, Cin, Cout)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const Wgrad1x1Plan pl = wgrad1x1_plan(P, Cin, Cout);
  float* partial = reinterpret_cast<float*>(workspace);
  UdProfScope prof("conv2d.k_wgrad_1x1", stream);
  int rc;
  if (pl.nt == 128 && pl.ct == 128) rc = launch_wgrad1x1<128, 128>(x, dy, partial, P, Cin, Cout, pl, xmap, ymap, stream);
  else if (pl.nt == 128) rc = launch_wgrad1x1<128, 64>(x, dy, partial, P, Cin, Cout, pl, xmap, ymap, stream);
  else if (pl.ct == 128) rc = launch_wgrad1x1<64, 128>(x, dy, partial, P, Cin, Cout, pl, xmap, ymap, stream);
  else rc = launch_wgrad1x1<64, 64>(x, dy, partial, P, Cin, Cout, pl, xmap, ymap, stream);
  if (rc != UD_OK) return rc;
  const size_t n = (size_t)Cout * Cin;
  k_wgrad_sum<<<ud_div_up((long long)(n / 4), 64), 256, 0, stream>>>(partial, pl.slices, n, dw);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_conv1x1_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int64_t P, int Cin, int Cout,
                                          void* workspace, size_t workspace_bytes, ud_stream_t stream) {
  return wgrad1x1_impl(x, dy, dw, P, Cin, Cout, workspace, workspace_bytes, PixMap{}, PixMap{}, stream);
}

// dW[n][k] = sum_p dy'[p][n] * x'[p][k] with either operand read through a PixMap (weight gradients of the
// convolutions ud_conv1x1_mapped_nhwc_bf16 runs).  Workspace: ud_conv1x1_wgrad_workspace_bytes(P, Cin, Cout).
extern "C" int ud_conv1x1_wgrad_mapped_nhwc_bf16(const void* x, const void* dy, float* dw, int64_t P, int Cin,
                                                 int Cout, const int* x_map, const int* dy_map, void* workspace,
                                                 size_t workspace_bytes, ud_stream_t stream) {
  PixMap xm, ym;
  if (!map_from_ints(x_map, &xm, 64) || !map_from_ints(dy_map, &ym, 64)) return UD_ERR_INVALID_ARG;
  if ((xm.mode && xm.C % 8) || (ym.mode && ym.C % 8)) return UD_ERR_INVALID_ARG;
  if (xm.mode == 1 && ((xm.s * xm.C) % 64 != 0 || xm.s * xm.s * xm.C != Cin)) return UD_ERR_UNSUPPORTED;
  if (xm.mode == 2 && xm.C != Cin) return UD_ERR_INVALID_ARG;
  if (xm.mode == 3 && 9 * xm.C != Cin) return UD_ERR_INVALID_ARG;
  if (xm.mode == 4 || ym.mode >= 3) return UD_ERR_UNSUPPORTED;
  if (ym.mode == 1 && ym.s * ym.s * ym.C != Cout) return UD_ERR_INVALID_ARG;
  if (ym.mode == 2 && ym.C != Cout) return UD_ERR_INVALID_ARG;
  if (P > (int64_t)1 << 30) return UD_ERR_UNSUPPORTED;                       // PixMap::off divides in 32 bits
  return wgrad1x1_impl(x, dy, dw, P, Cin, Cout, workspace, workspace_bytes, xm, ym, stream);
}
