// Persistent fp32 1x1 convolution (plain and pixel-mapped: strided / transposed / im2col launches) on v_mfma_f32_16x16x4_f32
// for the layers of the reference's image branch and BEV trunk that are 1x1 GEMMs in channels-last form (mmdet ResNet-50
// bottlenecks + FPN through lss_fpn.py:143-149, the strided / transposed blocks of base_bev_backbone.py:38-115; forward and
// data gradient).
//
//   y[p, n] = epilogue( sum_k x[map(p, k)] * w[n, k] )
//
// What the grid-per-tile kernel of conv2d_f32.hip left on the table (tools/time_f32_1x1.py, SQ counters): the launches of one
// step are 264 ... 8 448 workgroups of 128 pixels x 64 channels on 768 slots -- 2.06 workgroups per CU where three fit run at
// the speed of the CUs that got three -- and every workgroup pays its own prologue (first slices from L2 / HBM with nothing to
// overlap) and epilogue.  Here 2 x 256 workgroups stay resident and walk a balanced schedule:
//   * units (pixel tile, 64-channel block) [0, n_dp) whole, unit = logical workgroup, + grid, ... (data parallel); the 32-channel
//     slices of the remaining units form one sequence that is cut into equal ranges (stream-K); a unit that a range covers only
//     partly leaves its raw accumulators in the workspace and k_conv1x1p_fixup adds the pieces in slice order (deterministic)
//     and runs the epilogue;
//   * the slices of a workgroup's whole schedule flow through ONE three-stage LDS ring (24 KB per stage, LDS-DMA through buffer
//     descriptors): the load runs two slices ahead of the MFMAs across unit boundaries, so only the first slice of a
//     workgroup is exposed; one barrier per slice, counted vmcnt (the stores of an epilogue in between only make the wait
//     longer, never shorter);
//   * D = W * X^T (channels are the MFMA rows): a lane ends up with four consecutive channels of one pixel, so the epilogue --
//     bias, folded BN, residual, ReLU, BatchNorm partial sums -- runs on the accumulators and stores 16 bytes per lane
//     straight from registers: no LDS staging, nothing aliasing the ring.
#include "ud_common.h"
#include "ud_prof.h"
#include "conv_pixmap.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace {

constexpr int kTM = 128, kTN = 64, kKC = 32, kNS = 3;
constexpr int kXBytes = kTM * 128, kWBytes = kTN * 128, kStage = kXBytes + kWBytes;
constexpr int kRed = kNS * kStage;                 // BatchNorm partial sums of the four waves: [4][64][2] doubles
constexpr int kSmem = kRed + 4 * 64 * 2 * 8;       // 77 824 bytes: two workgroups per CU
constexpr int kGridP = 512;                        // persistent workgroups (2 per CU on 256 CUs)
constexpr unsigned kOob = 0xFFF00000u;             // a byte offset no tensor reaches (the launcher checks): reads zeros

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PGeom {
  long long npix;
  int Cin, Cout, ngroups, nchunks;
  int n_dp, n_units, sk_len, grid;
  int dq, dr;                // grid = dq * ngroups + dr
  int fast_out;              // plain output, Cout % 64 == 0: one offset per lane + immediates, range-checked by the descriptor
  float* partial;            // stream-K pieces: [2 * grid][128][64]
  PixMap imap, omap;
};
struct PEp {
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  int relu;
  float* stats;              // [tiles][Cout][2] or nullptr
};

#ifndef UD_P_ABL
#define UD_P_ABL 0     // development (tools/_exp/p1): 1 no LDS-DMA, 2 no ring barrier, 4 no fragment reads, 8 no MFMAs, 16 no stores
#endif
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
  if (UD_P_ABL & 1) return;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(size_t)lds_wave_base, 16, voff, soff, 0, 0);
}

// The pieces of one workgroup in schedule order: whole units first (unit = logical workgroup, + grid, ...: (tile, block) stepped
// by the launcher's divmod of the grid -- no division per unit), then its stream-K range unit by unit.
struct PieceIt {
  int tile, ng, dp_left, sk_lo, sk_hi, sk_first;
  __device__ __forceinline__ void init(const PGeom& gm, int L, int J) {
    dp_left = L < gm.n_dp ? (gm.n_dp - 1 - L) / gm.grid + 1 : 0;
    tile = L / gm.ngroups;
    ng = L - tile * gm.ngroups;
    sk_first = sk_lo = gm.n_dp * gm.nchunks + J * gm.sk_len;
    sk_hi = min(sk_lo + gm.sk_len, gm.n_units * gm.nchunks);
  }
  // -> pixel tile, 64-channel block, slices [c0, c1), partial slot (-1: the whole unit)
  __device__ __forceinline__ bool next(const PGeom& gm, int J, int& t, int& n, int& c0, int& c1, int& slot) {
    if (dp_left > 0) {
      --dp_left;
      t = tile, n = ng, c0 = 0, c1 = gm.nchunks, slot = -1;
      ng += gm.dr;
      tile += gm.dq;
      if (ng >= gm.ngroups) ng -= gm.ngroups, ++tile;
      return true;
    }
    if (sk_lo >= sk_hi) return false;
    const int unit = sk_lo / gm.nchunks;
    c0 = sk_lo - unit * gm.nchunks;
    c1 = min(gm.nchunks, c0 + sk_hi - sk_lo);
    slot = (c0 == 0 && c1 == gm.nchunks) ? -1 : 2 * J + (sk_lo == sk_first ? 0 : 1);
    sk_lo += c1 - c0;
    t = unit / gm.ngroups;
    n = unit - t * gm.ngroups;
    return true;
  }
};

// pixel part of a mapped output offset (the channel part is added per 16-channel block): see PixMap::off
__device__ __forceinline__ long long out_pix_off(const PixMap& om, unsigned p, int Cout) {
  if (om.mode == 0) return (long long)p * Cout;
  const unsigned t = p / (unsigned)om.Wo;
  const int ox = (int)(p - t * (unsigned)om.Wo);
  const int b = (int)(t / (unsigned)om.Ho), oy = (int)(t - (unsigned)b * (unsigned)om.Ho);
  if (om.mode == 1) return ((long long)(b * om.H + om.s * oy) * om.W + (long long)om.s * ox) * om.C;
  return ((long long)(b * om.H + om.s * oy + om.a) * om.W + (long long)om.s * ox + om.b) * om.C;
}
__device__ __forceinline__ int out_ch_off(const PixMap& om, int k) {
  if (om.mode != 1) return k;
  const int sc = om.s * om.C, dy = k / sc;
  return dy * om.W * om.C + (k - dy * sc);
}

// sum over the 16 lanes of a row (li) of 16 values per lane, the value with index (b3 b2 b1 b0) ending in the lane with those li
// bits: four exchange steps that halve the values a lane carries (v[i] <-> partner's v[i ^ half]); 15 DPP adds instead of 64.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_transpose_sum(const float (&in)[16], int li) {
  float a[8], b[4], c[2];
  const bool h3 = li & 8, h2 = li & 4, h1 = li & 2, h0 = li & 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {          // partner li ^ 15 (row_mirror 0x140): keep the half with bit 3 == mine
    const float keep = h3 ? in[i + 8] : in[i], send = h3 ? in[i] : in[i + 8];
    a[i] = keep + dpp_f<0x140>(send);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {          // partner li ^ 7 (row_half_mirror 0x141)
    const float keep = h2 ? a[i + 4] : a[i], send = h2 ? a[i] : a[i + 4];
    b[i] = keep + dpp_f<0x141>(send);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {          // partner li ^ 2 (quad_perm [2,3,0,1] = 0x4E)
    const float keep = h1 ? b[i + 2] : b[i], send = h1 ? b[i] : b[i + 2];
    c[i] = keep + dpp_f<0x4E>(send);
  }
  const float keep = h0 ? c[1] : c[0], send = h0 ? c[0] : c[1];      // partner li ^ 1 (quad_perm [1,0,3,2] = 0xB1)
  return keep + dpp_f<0xB1>(send);
}

// The same exchange on doubles (two DPP moves per value): the BatchNorm partial sums leave the tile accurate to the final
// rounding to float -- the statistics kernels downstream combine tiles in double, and a training-mode BatchNorm backward
// amplifies an inconsistency between its statistics and its input by (mean / std)^2 (tests/test_image_branch_f32_gpu.py).
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double row16_transpose_sum(const double (&in)[16], int li) {
  double a[8], b[4], c[2];
  const bool h3 = li & 8, h2 = li & 4, h1 = li & 2, h0 = li & 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double keep = h3 ? in[i + 8] : in[i], send = h3 ? in[i] : in[i + 8];
    a[i] = keep + dpp_d<0x140>(send);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double keep = h2 ? a[i + 4] : a[i], send = h2 ? a[i] : a[i + 4];
    b[i] = keep + dpp_d<0x141>(send);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double keep = h1 ? b[i + 2] : b[i], send = h1 ? b[i] : b[i + 2];
    c[i] = keep + dpp_d<0x4E>(send);
  }
  const double keep = h0 ? c[1] : c[0], send = h0 ? c[0] : c[1];
  return keep + dpp_d<0xB1>(send);
}

template <bool MAPPED>
__global__ __launch_bounds__(256, 2) void k_conv1x1p_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, PGeom gm, PEp ep, unsigned x_bytes,
                                                         unsigned w_bytes, unsigned y_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int r8 = lane >> 3, slot8 = lane & 7;
  const int L = (blockIdx.x & 7) * (gm.grid >> 3) + (blockIdx.x >> 3);      // an XCD walks consecutive units (shared pixel tiles)
  const int J = blockIdx.x;              // stream-K ranges in launch order: a short tail spreads over all XCDs, one range per CU first
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)w_bytes, 0x00020000);
  // outputs / the residual through descriptors too: a pixel or channel past the end becomes an out-of-range offset (stores dropped,
  // loads return zeros) -- no branch around a memory instruction, so a piece issues a FIXED number of them and the ring's
  // vmcnt can count them
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(ep.residual ? ep.residual : y), 0, (int)y_bytes, 0x00020000);
  const PixMap im = gm.imap, om = gm.omap;          // copies: a reference into the by-value argument keeps it on the stack
  const int imode = MAPPED ? im.mode : 0;
  const bool fast_out = !MAPPED && gm.fast_out, want_res = ep.residual != nullptr;

  // ---- load side: three slices ahead of the MFMAs ----
  PieceIt lp;
  lp.init(gm, L, J);
  int l_left = 0;                        // slices left in the piece being loaded
  unsigned l_so = 0;                     // byte offset of its next slice within a row
  bool l_more = true;
  unsigned xv[4], wv[2];                 // this lane's byte offsets (slice 0) of its 16-byte pieces; past the end: zeros
  unsigned xrow[4], wrow[2];             // plain input: the lane's offsets within a pixel tile / a 64-channel block
  const unsigned sw4 = (unsigned)((slot8 ^ r8) << 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) xrow[i] = (unsigned)((wave + 4 * i) * 8 + r8) * (unsigned)gm.Cin * 4u + sw4;
#pragma unroll
  for (int j = 0; j < 2; ++j) wrow[j] = (unsigned)((wave + 4 * j) * 8 + r8) * (unsigned)gm.Cin * 4u + sw4;
  int py[4], px[4];                      // im2col maps: the lane's source pixel (range test per tap)
  int seg = 0, cseg = 0;                 // mapped: tap / block-row index and channel within it of the next slice
  const int seg_len = imode == 1 ? im.s * im.C : (imode >= 3 ? im.C : gm.Cin);
  // Within a segment (a tap / a block row of the map; the whole reduction for plain and strided inputs) a slice is affine: the
  // lane's offsets xo[] are fixed and the slice index rides in the scalar offsets.  Recomputed per segment, not per slice.
  unsigned xo[4];
  unsigned so_x = 0, so_w = 0;
  int seg_left = 0x7fffffff;
  auto seg_setup = [&]() __attribute__((always_inline)) {
    if (!MAPPED || imode == 0 || imode == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xo[i] = xv[i];
      so_x = so_w = l_so;
      seg_left = 0x7fffffff;
      return;
    }
    int koff, dy = 0, dx = 0, wk;
    if (imode == 1) {
      koff = seg * im.W * im.C;
      wk = seg * seg_len;
    } else if (imode == 3) {
      const int ty = seg / 3, tx = seg - 3 * ty;
      dy = ty - 1; dx = tx - 1;
      koff = (dy * im.W + dx) * im.C;
      wk = seg * seg_len;
    } else {
      const int nx = 1 + im.b, jy = seg / nx, jx = seg - jy * nx;
      dy = im.a * (1 - jy); dx = im.b * (1 - jx);
      koff = (dy * im.W + dx) * im.C;
      const int ty = im.a ? 2 * jy : 1, tx = im.b ? 2 * jx : 1;
      wk = (ty * 3 + tx) * im.C;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = xv[i] != kOob &&
                      (imode < 3 || ((unsigned)(py[i] + dy) < (unsigned)im.H && (unsigned)(px[i] + dx) < (unsigned)im.W));
      xo[i] = ok ? xv[i] + (unsigned)(koff * 4) : kOob;
    }
    so_x = (unsigned)cseg * 4u;
    so_w = (unsigned)(wk + cseg) * 4u;
    seg_left = (seg_len - cseg) / kKC;
  };
  auto load_setup = [&]() __attribute__((always_inline)) {
    int tile, ng, c0, c1, slot;
    l_more = lp.next(gm, J, tile, ng, c0, c1, slot);
    if (!l_more) return;
    l_left = c1 - c0;
    l_so = (unsigned)c0 * (kKC * 4);
    if (imode == 0 && gm.fast_out) {     // (Cout % 64 == 0: no channel row past the end)
#pragma unroll
      for (int i = 0; i < 4; ++i) xv[i] = (unsigned)tile * (unsigned)(kTM * gm.Cin * 4) + xrow[i];   // rows past P: past the descriptor
#pragma unroll
      for (int j = 0; j < 2; ++j) wv[j] = (unsigned)ng * (unsigned)(kTN * gm.Cin * 4) + wrow[j];
      seg_setup();
      return;
    }
    const int n0 = ng * kTN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned p = (unsigned)tile * kTM + (wave + 4 * i) * 8 + r8;      // P <= 2^30 (launcher)
      py[i] = px[i] = 0;
      if ((long long)p >= gm.npix) { xv[i] = kOob; continue; }
      if (imode == 0) {
        xv[i] = p * (unsigned)gm.Cin * 4u + sw4;
        continue;
      }
      const unsigned t = p / (unsigned)im.Wo;
      const int ox = (int)(p - t * (unsigned)im.Wo);
      const int b = (int)(t / (unsigned)im.Ho), oy = (int)(t - (unsigned)b * (unsigned)im.Ho);
      int y0, x0;
      if (imode == 4) { y0 = oy; x0 = ox; }
      else if (imode == 2) { y0 = im.s * oy + im.a; x0 = im.s * ox + im.b; }
      else { y0 = im.s * oy; x0 = im.s * ox; }
      xv[i] = (unsigned)((b * im.H + y0) * im.W + x0) * (unsigned)im.C * 4u + sw4;
      if (imode >= 3) { py[i] = y0; px[i] = x0; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = min(n0 + (wave + 4 * j) * 8 + r8, gm.Cout - 1);          // channels past Cout re-read the last one (never stored)
      wv[j] = (unsigned)n * (unsigned)(imode == 4 ? 9 * im.C : gm.Cin) * 4u + sw4;
    }
    if (MAPPED && imode != 0 && imode != 2) { seg = (c0 * kKC) / seg_len; cseg = c0 * kKC - seg * seg_len; }
    seg_setup();
  };
  int in_flight = 0;                     // slices issued and not yet consumed
  auto issue_slice = [&](unsigned buf_off) __attribute__((always_inline)) {
    const unsigned xb = sbase + buf_off + wave * 1024, wb = xb + kXBytes;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rx, xo[i], so_x, xb + i * 4096);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(rw, wv[j], so_w, wb + j * 4096);
    so_x += kKC * 4;
    so_w += kKC * 4;
    l_so += kKC * 4;
    if (MAPPED && --seg_left == 0) {
      ++seg;
      cseg = 0;
      seg_setup();
    }
  };
  auto load_slice = [&](unsigned buf_off) __attribute__((always_inline)) {
    if (!l_more) return;
    if (l_left == 0) {
      load_setup();
      if (!l_more) return;
    }
    issue_slice(buf_off);
    --l_left;
    ++in_flight;
  };

  // ---- MFMA side ----
  PieceIt cp;
  cp.init(gm, L, J);
  int c_tile, c_ng, c_left, c_slot;
  {
    int c0, c1;
    if (!cp.next(gm, J, c_tile, c_ng, c0, c1, c_slot)) return;
    c_left = c1 - c0;
  }
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment addresses in the ring buffer being read (k-steps 0-3 / 4-7 of the slice), stepped with the ring
  unsigned aw0, aw1, ax0, ax1;
  {
    const unsigned p0 = (unsigned)((g ^ (li & 7)) << 4), p1 = (unsigned)(((4 + g) ^ (li & 7)) << 4);
    aw0 = sbase + kXBytes + li * 128 + p0, aw1 = sbase + kXBytes + li * 128 + p1;
    ax0 = sbase + (32 * wave + li) * 128 + p0, ax1 = sbase + (32 * wave + li) * 128 + p1;
  }
  unsigned rd_off = 0;                   // byte offset of that buffer

  // output offsets (bytes) of the lane's 4 x 2 pieces of the unit being multiplied
  unsigned ov[4][2];                     // past the end (kOob, or a row past P on the fast path): not stored
  u32x4 rres[4][2];                      // residual pieces, requested one slice before the epilogue
  auto piece_offsets = [&]() __attribute__((always_inline)) {
    const unsigned p0 = (unsigned)c_tile * kTM + 32 * wave + li;
    if (fast_out) {                      // rows past P lie past the descriptor's end
      const unsigned ob = (p0 * (unsigned)gm.Cout + (unsigned)(c_ng * kTN + 4 * g)) * 4u;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) ov[ci][pj] = ob + 64 * ci + pj * ((unsigned)gm.Cout * 64u);
      return;
    }
    long long po[2];
#pragma unroll
    for (int pj = 0; pj < 2; ++pj) {
      const bool pok = (long long)(p0 + 16 * pj) < gm.npix;
      po[pj] = !pok ? -1 : MAPPED ? out_pix_off(om, p0 + 16 * pj, gm.Cout) : (long long)(p0 + 16 * pj) * gm.Cout;
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int n = c_ng * kTN + 16 * ci + 4 * g;
      const int no = n >= gm.Cout ? -1 : MAPPED ? out_ch_off(om, n) : n;
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) ov[ci][pj] = (no < 0 || po[pj] < 0) ? kOob : (unsigned)(po[pj] + no) * 4u;
    }
  };
  auto request_residual = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj)
        rres[ci][pj] = __builtin_amdgcn_raw_buffer_load_b128(rr, ov[ci][pj], 0, 0);
  };
  auto epilogue = [&]() __attribute__((always_inline)) {
    if (c_slot >= 0) {                   // a stream-K piece: raw sums, [128][64]
      float* dst = gm.partial + ((size_t)c_slot * kTM + 32 * wave + li) * kTN + 4 * g;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj)
          *reinterpret_cast<f32x4*>(dst + (size_t)pj * 16 * kTN + 16 * ci) = acc[ci][pj];
      return;
    }
    const int n0 = c_ng * kTN;
    double s1[16], s2[16];
    float vm[2] = {1.f, 1.f};            // BatchNorm sums: rows past the last pixel do not count
    if (ep.stats) {
#pragma unroll
      for (int pj = 0; pj < 2; ++pj)
        vm[pj] = (long long)((unsigned)c_tile * kTM + 32 * wave + li + 16 * pj) < gm.npix ? 1.f : 0.f;
    }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int n = min(n0 + 16 * ci + 4 * g, gm.Cout - 4);
      f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (ep.bias) bv = *reinterpret_cast<const f32x4*>(ep.bias + n);
      if (ep.scale) {
        sc = *reinterpret_cast<const f32x4*>(ep.scale + n);
        sh = *reinterpret_cast<const f32x4*>(ep.shift + n);
      }
      double t1[4] = {0.0, 0.0, 0.0, 0.0}, t2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) {
        f32x4 v = acc[ci][pj];
        if (ep.bias) v += bv;
        if (ep.scale) v = v * sc + sh;
        if (ep.residual) v += __builtin_bit_cast(f32x4, rres[ci][pj]);
        if (ep.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (!(UD_P_ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, ov[ci][pj], 0, 0);
        if (ep.stats) {
          const float m = fast_out ? vm[pj] : (ov[ci][pj] == kOob ? 0.f : 1.f);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double d = (double)(v[r] * m);
            t1[r] += d;
            t2[r] += d * d;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) s1[4 * ci + r] = t1[r], s2[4 * ci + r] = t2[r];
    }
    if (ep.stats) {                      // per-channel (sum, sum of squares) of the tile's stored outputs
      double* red = reinterpret_cast<double*>(smem + kRed);
      const double a = row16_transpose_sum(s1, li), q = row16_transpose_sum(s2, li);
      // lane li holds value index li = 4 ci + r of its group g: channel 16 ci + 4 g + r
      const int ch = 16 * (li >> 2) + 4 * g + (li & 3);
      red[(wave * 64 + ch) * 2] = a;
      red[(wave * 64 + ch) * 2 + 1] = q;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // raw: the slices in flight stay in flight; `red` is next written after a ring barrier
      if (tid < 128) {
        const int chn = tid >> 1, which = tid & 1;
        const double v = ((red[(0 * 64 + chn) * 2 + which] + red[(1 * 64 + chn) * 2 + which]) +
                          red[(2 * 64 + chn) * 2 + which]) + red[(3 * 64 + chn) * 2 + which];
        if (n0 + chn < gm.Cout) ep.stats[((size_t)c_tile * gm.Cout + n0 + chn) * 2 + which] = (float)v;
      }
    }
  };

  // fragment reads as inline asm: hipcc otherwise sinks them below the MFMAs they are meant to run under; the matching waits
  // name the registers so that no use moves above them
#define UD_P_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define UD_P_RD6(wf, xf, aw, ax)                                                                       \
  do {                                                                                                 \
    if (UD_P_ABL & 4) {                                                                                \
      for (int ci_ = 0; ci_ < 4; ++ci_) wf[ci_] = (f32x4){(float)c_left, (float)ci_, 1.f, 1.f};        \
      for (int pj_ = 0; pj_ < 2; ++pj_) xf[pj_] = (f32x4){(float)c_tile, (float)pj_, 1.f, 1.f};        \
    } else {                                                                                           \
      UD_P_RD(wf[0], aw, 0); UD_P_RD(wf[1], aw, 2048); UD_P_RD(wf[2], aw, 4096); UD_P_RD(wf[3], aw, 6144); \
      UD_P_RD(xf[0], ax, 0); UD_P_RD(xf[1], ax, 2048);                                                  \
    }                                                                                                  \
  } while (0)
  auto mma = [&](const f32x4 (&wf)[4], const f32x4 (&xf)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int pj = 0; pj < 2; ++pj)
          if (!(UD_P_ABL & 8)) acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ci][e], xf[pj][e], acc[ci][pj], 0, 0, 0);
          else acc[ci][pj][e] += wf[ci][e] * xf[pj][0];
  };

  load_slice(0);
  load_slice(kStage);
  load_slice(2 * kStage);
  if (in_flight >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!(UD_P_ABL & 2)) __builtin_amdgcn_s_barrier();
  if ((UD_P_ABL & 32) && (blockIdx.x & 256)) __builtin_amdgcn_s_sleep(8);
  if ((UD_P_ABL & 64) && (blockIdx.x & 256)) __builtin_amdgcn_s_sleep(16);
  if ((UD_P_ABL & 128) && (blockIdx.x & 256)) __builtin_amdgcn_s_setprio(1);
  f32x4 wa[4], xa[2];                    // fragments of k-steps 0-3 of the slice about to be multiplied (read one half-step ahead)
  UD_P_RD6(wa, xa, aw0, ax0);
  piece_offsets();
  // Vector-memory operations retire in order (loads and stores share vmcnt on gfx9): when slice q + 1 is needed, everything
  // issued after its six loads may still be in flight -- the six loads of slice q + 2 and the eight stores of the epilogues that
  // ran in the two steps before (each issued after slice q + 1's loads).
  int eh = 0;                            // bit 0 / 1: an epilogue (8 stores) ran in the previous step / the one before
  // One slice: [fragments of k-steps 4-7] MFMAs 0-3 | slice q + 1 landed + barrier | [load slice q + 3 into this buffer,
  // fragments 0-3 of slice q + 1] MFMAs 4-7 | epilogue at the end of a piece.  Every fragment read has 32 MFMAs to arrive under;
  // at the barrier every wave holds the whole of slice q in registers, so its buffer is free for the load three slices ahead.
  for (;;) {
    // Steady state inside a piece (no epilogue in the last two steps, this one not the piece's last, three slices in flight, the
    // load side inside its own piece): the same step with every decision taken out -- the general step below spends ~60 scalar
    // instructions per slice on them (SQ counters, profiles/r05_conv_f32_1x1.md), and each is paid in MFMA issue time.
    while (c_left >= 2 && eh == 0 && in_flight >= 3 && l_left > 0 && !(UD_P_ABL & 256)) {
      f32x4 wb[4], xb[2];
      UD_P_RD6(wb, xb, aw1, ax1);
      if (!(UD_P_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]), "+v"(wa[3]), "+v"(xa[0]), "+v"(xa[1]) :: "memory");
      mma(wa, xa);
      if (!(UD_P_ABL & 4))
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)"
                     : "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]), "+v"(wb[3]), "+v"(xb[0]), "+v"(xb[1]), "+v"(acc[0][0]), "+v"(acc[0][1]),
                       "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1])
                     :: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      if (!(UD_P_ABL & 2)) __builtin_amdgcn_s_barrier();
      const unsigned freed_ = rd_off;
      const int step_b = rd_off == 2 * kStage ? -2 * kStage : kStage;
      rd_off += step_b;
      aw0 += step_b, aw1 += step_b, ax0 += step_b, ax1 += step_b;
      issue_slice(freed_);
      --l_left;
      UD_P_RD6(wa, xa, aw0, ax0);
      mma(wb, xb);
      --c_left;
    }
    f32x4 wb[4], xb[2];
    UD_P_RD6(wb, xb, aw1, ax1);
    if (!(UD_P_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wa[0]), "+v"(wa[1]), "+v"(wa[2]), "+v"(wa[3]), "+v"(xa[0]), "+v"(xa[1]) :: "memory");
    mma(wa, xa);
    // slice q is in registers: its buffer is free after the barrier.  The accumulators are named so that the MFMAs above stay
    // above (they are what the reads run under).
    if (!(UD_P_ABL & 4))
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]), "+v"(wb[3]), "+v"(xb[0]), "+v"(xb[1]), "+v"(acc[0][0]), "+v"(acc[0][1]),
                     "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1])
                   :: "memory");
    const bool last = c_left == 1, has_next = in_flight >= 2;
    if (has_next) {
      if (in_flight >= 3) {
        if (eh == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (eh == 3) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (!(UD_P_ABL & 2)) __builtin_amdgcn_s_barrier();      // raw: __syncthreads() carries a fence = vmcnt(0), draining the ring
    }
    if (last && want_res && c_slot < 0) request_residual();   // consumed in this step's epilogue: never outstanding at a later wait
    if (has_next) {
      const unsigned freed = rd_off;
      const int step_b = rd_off == 2 * kStage ? -2 * kStage : kStage;
      rd_off += step_b;
      aw0 += step_b, aw1 += step_b, ax0 += step_b, ax1 += step_b;
      load_slice(freed);
      UD_P_RD6(wa, xa, aw0, ax0);
    }
    mma(wb, xb);
    --in_flight;
    eh = (eh << 1) & 2;
    if (last) {
      epilogue();
      eh |= 1;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int c0, c1;
      if (!cp.next(gm, J, c_tile, c_ng, c0, c1, c_slot)) break;
      c_left = c1 - c0;
      piece_offsets();
    } else {
      --c_left;
    }
  }
#undef UD_P_RD6
#undef UD_P_RD
}

// Short reductions (fewer than kShortK slices: the 64 / 128 / 256-channel ResNet layers over 67 k - 270 k pixels) -- a unit is a
// few hundred MFMAs, so what a workgroup spends around them decides: one workgroup per (pixel tile, 64-channel block) with no
// schedule to walk, two LDS stages (three workgroups per CU cover each other's fill), and the same epilogue from registers
// as above (the grid-per-tile kernel of conv2d_f32.hip stages its tile through LDS: ~1 300 instructions per wave against the
// 128 - 512 MFMAs of such a unit).  Plain input and output only.
constexpr int kShortK = 12;
constexpr int kSmemT = 2 * kStage + 4 * 64 * 2 * 8;
__global__ __launch_bounds__(256, 3) void k_conv1x1t_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, PGeom gm, PEp ep, unsigned x_bytes,
                                                         unsigned w_bytes, unsigned y_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int r8 = lane >> 3, slot8 = lane & 7;
  const int ntiles = (int)((gm.npix + kTM - 1) / kTM), per = (ntiles + 7) >> 3;
  const int tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);        // an XCD walks consecutive pixel tiles
  if (tile >= ntiles) return;
  const int ng = blockIdx.y, n0 = ng * kTN;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(ep.residual ? ep.residual : y), 0, (int)y_bytes, 0x00020000);
  const unsigned sw4 = (unsigned)((slot8 ^ r8) << 4);
  unsigned xv[4], wv[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)          // rows past P lie past the descriptor's end: zeros
    xv[i] = ((unsigned)tile * kTM + (wave + 4 * i) * 8 + r8) * (unsigned)gm.Cin * 4u + sw4;
#pragma unroll
  for (int j = 0; j < 2; ++j)          // channels past Cout re-read the last one (never stored)
    wv[j] = (unsigned)min(n0 + (wave + 4 * j) * 8 + r8, gm.Cout - 1) * (unsigned)gm.Cin * 4u + sw4;
  const unsigned lb = sbase + wave * 1024;
  auto stage = [&](int c, int buf) __attribute__((always_inline)) {
    const unsigned so = (unsigned)c * (kKC * 4), xb = lb + buf * kStage;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16(rx, xv[i], so, xb + i * 4096);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(rw, wv[j], so, xb + kXBytes + j * 4096);
  };
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned fwo[2], fxo[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned piece = (unsigned)(((4 * ks + g) ^ (li & 7)) << 4);
    fwo[ks] = kXBytes + li * 128 + piece;
    fxo[ks] = (32 * wave + li) * 128 + piece;
  }
  auto mma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4 wf[4], xf[2];
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) wf[ci] = *reinterpret_cast<const f32x4*>(smem + fwo[ks] + buf * kStage + ci * 2048);
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) xf[pj] = *reinterpret_cast<const f32x4*>(smem + fxo[ks] + buf * kStage + pj * 2048);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
            acc[ci][pj] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ci][e], xf[pj][e], acc[ci][pj], 0, 0, 0);
    }
  };
  // output offsets of the lane's 4 x 2 pieces (kOob, or a row past P on the fast path: not stored) and the residual, requested
  // before the multiplications it hides under
  unsigned ov[4][2];
  {
    const unsigned p0 = (unsigned)tile * kTM + 32 * wave + li;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int n = n0 + 16 * ci + 4 * g;
#pragma unroll
      for (int pj = 0; pj < 2; ++pj)
        ov[ci][pj] = (n >= gm.Cout || (long long)(p0 + 16 * pj) >= gm.npix) ? kOob : ((p0 + 16 * pj) * (unsigned)gm.Cout + n) * 4u;
    }
  }
  u32x4 rres[4][2];
  if (ep.residual) {
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) rres[ci][pj] = __builtin_amdgcn_raw_buffer_load_b128(rr, ov[ci][pj], 0, 0);
  }
  const int nchunks = gm.nchunks;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int c = 0; c < nchunks; c += 2) {
    if (c + 1 < nchunks) stage(c + 1, 1);
    mma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 1 < nchunks) {
      if (c + 2 < nchunks) stage(c + 2, 0);
      mma(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  double s1[16], s2[16];
#pragma unroll
  for (int ci = 0; ci < 4; ++ci) {
    const int n = min(n0 + 16 * ci + 4 * g, gm.Cout - 4);
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (ep.bias) bv = *reinterpret_cast<const f32x4*>(ep.bias + n);
    if (ep.scale) {
      sc = *reinterpret_cast<const f32x4*>(ep.scale + n);
      sh = *reinterpret_cast<const f32x4*>(ep.shift + n);
    }
    double t1[4] = {0.0, 0.0, 0.0, 0.0}, t2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pj = 0; pj < 2; ++pj) {
      f32x4 v = acc[ci][pj];
      if (ep.bias) v += bv;
      if (ep.scale) v = v * sc + sh;
      if (ep.residual) v += __builtin_bit_cast(f32x4, rres[ci][pj]);
      if (ep.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, ov[ci][pj], 0, 0);
      if (ep.stats) {
        const float m = ov[ci][pj] == kOob ? 0.f : 1.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double d = (double)(v[r] * m);
          t1[r] += d;
          t2[r] += d * d;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[4 * ci + r] = t1[r], s2[4 * ci + r] = t2[r];
  }
  if (ep.stats) {
    double* red = reinterpret_cast<double*>(smem + 2 * kStage);
    const double a = row16_transpose_sum(s1, li), q = row16_transpose_sum(s2, li);
    const int ch = 16 * (li >> 2) + 4 * g + (li & 3);
    red[(wave * 64 + ch) * 2] = a;
    red[(wave * 64 + ch) * 2 + 1] = q;
    __syncthreads();
    if (tid < 128) {
      const int chn = tid >> 1, which = tid & 1;
      const double v = ((red[(0 * 64 + chn) * 2 + which] + red[(1 * 64 + chn) * 2 + which]) +
                        red[(2 * 64 + chn) * 2 + which]) + red[(3 * 64 + chn) * 2 + which];
      if (n0 + chn < gm.Cout) ep.stats[((size_t)tile * gm.Cout + n0 + chn) * 2 + which] = (float)v;
    }
  }
}

// Sum of the pieces of the units the stream-K schedule cut, in slice order, + the epilogue.  One workgroup per cut unit; a
// thread owns one 4-channel piece over 8 pixel rows (rows tid / 16 + 16 k).
template <bool MAPPED>
__global__ __launch_bounds__(256) void k_conv1x1p_fixup(float* __restrict__ y, PGeom gm, PEp ep) {
  __shared__ double red[16 * 64 * 2];
  const int tid = threadIdx.x;
  const int u = gm.n_dp + blockIdx.x;
  const int base = gm.n_dp * gm.nchunks;
  const int a = u * gm.nchunks - base, e = a + gm.nchunks - 1;          // the unit's slices within the stream-K sequence
  const int j0 = a / gm.sk_len, j1 = e / gm.sk_len;
  if (j0 == j1) return;                                                 // one workgroup covered it: stored already
  const int tile = u / gm.ngroups, n0 = (u - tile * gm.ngroups) * kTN;
  const int c4 = (tid & 15) * 4, row0 = tid >> 4;
  f32x4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int j = j0; j <= j1; ++j) {
    const int first_unit = (base + j * gm.sk_len) / gm.nchunks;         // the first unit workgroup j touched
    const int slot = 2 * j + (first_unit == u ? 0 : 1);
    const float* src = gm.partial + ((size_t)slot * kTM + row0) * kTN + c4;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += *reinterpret_cast<const f32x4*>(src + (size_t)k * 16 * kTN);
  }
  const int n = n0 + c4;
  const bool nok = n < gm.Cout;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (ep.bias && nok) bv = *reinterpret_cast<const f32x4*>(ep.bias + n);
  if (ep.scale && nok) {
    sc = *reinterpret_cast<const f32x4*>(ep.scale + n);
    sh = *reinterpret_cast<const f32x4*>(ep.shift + n);
  }
  const PixMap om = gm.omap;
  const int no = MAPPED ? out_ch_off(om, nok ? n : 0) : n;
  double t1[4] = {0.0, 0.0, 0.0, 0.0}, t2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const unsigned p = (unsigned)tile * kTM + row0 + 16 * k;
    if (!nok || (long long)p >= gm.npix) continue;
    const long long po = MAPPED ? out_pix_off(om, p, gm.Cout) : (long long)p * gm.Cout;
    f32x4 t = v[k];
    if (ep.bias) t += bv;
    if (ep.scale) t = t * sc + sh;
    if (ep.residual) t += *reinterpret_cast<const f32x4*>(ep.residual + po + no);
    if (ep.relu) {
      t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f);
    }
    *reinterpret_cast<f32x4*>(y + po + no) = t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double d = (double)t[r];
      t1[r] += d;
      t2[r] += d * d;
    }
  }
  if (ep.stats) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[(row0 * 64 + c4 + r) * 2] = t1[r];
      red[(row0 * 64 + c4 + r) * 2 + 1] = t2[r];
    }
    __syncthreads();
    if (tid < 128) {
      const int chn = tid >> 1, which = tid & 1;
      double s = 0.0;
      for (int k = 0; k < 16; ++k) s += red[(k * 64 + chn) * 2 + which];
      if (n0 + chn < gm.Cout) ep.stats[((size_t)tile * gm.Cout + n0 + chn) * 2 + which] = (float)s;
    }
  }
}

std::atomic<int> g_p_sk{-1};        // ud_conv1x1p_stream_k: -1 default (UD_F32_1X1P_SK or 1: where the cost model says so), 0 never, 2 whenever a tail can be cut
std::atomic<int> g_p_mode{-1};      // ud_conv1x1_f32_persistent: -1 default (UD_F32_1X1P or 1), 0 grid-per-tile kernels, 1 persistent

struct PSched {
  int grid, n_dp, sk_len;
};
// Whole rounds data parallel + a stream-K tail where that beats one more, partly filled, round.  The tail's slices are cut into
// ranges of >= 4 slices and a unit into <= 8 pieces (a piece costs a fill / drain and a 32 KB partial tile; the fix-up pass reads
// every piece); ranges go to workgroups in launch order, i.e. one per CU while there are fewer than 256.  Both tails are priced
// with the measured slice times (tools/_exp/p1: 2.3 us per slice for the two workgroups of a CU together, 1.25 us for one alone)
// and the fix-up pass at 6 us + its bytes at 3 TB/s.
PSched p_schedule(long long units, int nchunks, bool have_ws) {
  static const int env_mode = getenv("UD_F32_1X1P_SK") ? atoi(getenv("UD_F32_1X1P_SK")) : 1;
  const int forced = g_p_sk.load(std::memory_order_relaxed);
  const int env_sk = forced >= 0 ? forced : env_mode;
  const int G = kGridP;
  PSched sc{G, (int)units, 0};
  if (units <= 0) return sc;
  const int r = (int)(units % G);
  if (r == 0 || !have_ws || !env_sk || nchunks < 8) return sc;
  int len = (int)(((long long)r * nchunks + G - 1) / G);
  len = std::max(len, std::max(4, (nchunks + 7) / 8));
  if (len >= nchunks) return sc;
  const long long pieces = ((long long)r * nchunks + len - 1) / len;
  const double t_dp = nchunks * (r <= G / 2 ? 1.25 : 2.3);
  const int fan = (nchunks + len - 1) / len + 1;
  const double t_sk = len * (pieces <= G / 2 ? 1.25 : 2.3) + 6.0 + (double)r * (fan + 1) * 32768.0 / 3.0e6;
  if (env_sk < 2 && t_sk >= t_dp) return sc;
  sc.n_dp = (int)(units - r), sc.sk_len = len;
  return sc;
}

}  // namespace

extern "C" void ud_conv1x1_f32_persistent(int mode) { g_p_mode.store(mode, std::memory_order_relaxed); }
extern "C" void ud_conv1x1p_stream_k(int mode) { g_p_sk.store(mode, std::memory_order_relaxed); }

extern "C" int ud_conv1x1_f32_persistent_enabled(void) {
  static const int env_mode = getenv("UD_F32_1X1P") ? atoi(getenv("UD_F32_1X1P")) : 1;
  const int forced = g_p_mode.load(std::memory_order_relaxed);
  return forced >= 0 ? forced : env_mode;
}

extern "C" size_t ud_conv1x1p_f32_workspace_bytes(void) { return (size_t)2 * kGridP * kTM * kTN * sizeof(float); }

// y = conv1x1(x) over pixel maps (in_map / out_map: nine ints as in ud_conv1x1_mapped_nhwc_f32, nullptr = plain), (+ bias)
// (* scale + shift) (+ residual) (ReLU if flags & 1); partial != nullptr: BatchNorm partial sums [*slices = ceil(P / 128)][Cout][2].
// workspace (ud_conv1x1p_f32_workspace_bytes) enables the stream-K tail; without it the schedule is whole units only.
extern "C" int ud_conv1x1p_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout, const float* bias,
                                    const float* scale, const float* shift, const float* residual, int flags, float* partial,
                                    size_t partial_bytes, int* slices, const int* in_map, const int* out_map,
                                    size_t x_elems, size_t y_elems, void* workspace, size_t workspace_bytes,
                                    ud_stream_t stream_) {
  if (!x || !w || !y || P <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
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
  const bool mapped = im.mode != 0 || om.mode != 0;
  if (mapped && (bias || scale || residual || partial)) return UD_ERR_UNSUPPORTED;
  // 32-bit byte offsets through the buffer descriptors
  const size_t xe = im.mode == 0 ? (size_t)P * Cin : x_elems;
  const size_t we = (size_t)Cout * (im.mode == 4 ? 9 * (size_t)im.C : (size_t)Cin);
  const size_t ye = om.mode == 0 ? (size_t)P * Cout : y_elems;
  if (xe == 0 || ye == 0 || xe * 4 >= (size_t)kOob || we * 4 >= (size_t)kOob || ye * 4 >= (size_t)kOob) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int ntiles = (int)((P + kTM - 1) / kTM), ngroups = ud_div_up(Cout, kTN);
  const long long units = (long long)ntiles * ngroups;
  if (units > 0x3fffffffll / (Cin / kKC)) return UD_ERR_UNSUPPORTED;
  const bool have_ws = workspace && workspace_bytes >= ud_conv1x1p_f32_workspace_bytes();
  const PSched sc = p_schedule(units, Cin / kKC, have_ws);
  PGeom gm{(long long)P, Cin, Cout, ngroups, Cin / kKC, sc.n_dp, (int)units, sc.sk_len > 0 ? sc.sk_len : 1, sc.grid,
           sc.grid / ngroups, sc.grid % ngroups, om.mode == 0 && Cout % kTN == 0, (float*)workspace, im, om};
  PEp ep{bias, scale, shift, residual, flags & 1, partial};
  if (partial) {
    if (!slices || partial_bytes < (size_t)ntiles * Cout * 2 * sizeof(float)) return UD_ERR_WORKSPACE;
    *slices = ntiles;
  }
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1t_f32, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemT));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1p_f32<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv1x1p_f32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("conv2d.k_conv1x1_f32", stream);
  const int n_sk = (int)units - sc.n_dp;
  const unsigned xb = (unsigned)(xe * 4), wb = (unsigned)(we * 4), yb = (unsigned)(ye * 4);
  static const int short_k = getenv("UD_F32_1X1T_K") ? atoi(getenv("UD_F32_1X1T_K")) : kShortK;
  if (!mapped && Cin / kKC < short_k) {
    const dim3 grid((ntiles + 7) / 8 * 8, ngroups);
    k_conv1x1t_f32<<<grid, 256, kSmemT, stream>>>(x, w, y, gm, ep, xb, wb, yb);
  } else if (mapped) {
    k_conv1x1p_f32<true><<<sc.grid, 256, kSmem, stream>>>(x, w, y, gm, ep, xb, wb, yb);
    if (n_sk > 0) k_conv1x1p_fixup<true><<<n_sk, 256, 0, stream>>>(y, gm, ep);
  } else {
    k_conv1x1p_f32<false><<<sc.grid, 256, kSmem, stream>>>(x, w, y, gm, ep, xb, wb, yb);
    if (n_sk > 0) k_conv1x1p_fixup<false><<<n_sk, 256, 0, stream>>>(y, gm, ep);
  }
  UD_LAUNCH_CHECK();
  return UD_OK;
}
