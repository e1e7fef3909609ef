// fp32 3x3 / stride 1 / pad 1 convolution as Winograd F(4x4, 3x3) on v_mfma_f32_16x16x4_f32: 36 multiplications per 4 x 4
// outputs per (cin, cout) instead of 144 -- 4x fewer MFMA flops than the direct form, 1.78x fewer than F(2x2, 3x3)
// (csrc/conv2d_f32_wino.hip) -- for the trunk / head / ResNet 3x3 layers of the fp32 (headline) step
// (base_bev_backbone.py:30-110, center_head.py:311-420, lss_fpn.py:143-149).  Forward and data gradient (the weight transform
// takes the transposed, tap-reversed view for the latter).  Rounding: 3-5e-6 of the output's max against fp64 (the direct fp32
// kernel: 0.5-1e-6), tested per layer shape.
//
//   V[f][tile][c] = (B^T d B)[f]     d = 6 x 6 input patch of a 4 x 4 output tile          (in-kernel, per 4-channel stage)
//   U[f][n][c]    = (G g G^T)[f]     g = 3 x 3 filter                                      (k_wino4_weights, once per weight version)
//   M[f][tile][n] = sum_c V[f][tile][c] U[f][n][c]                                         (36 independent GEMMs on the MFMA pipe)
//   y(4 x 4)      = A^T M A                                                                (in registers, then the usual epilogue)
//
// A workgroup (8 waves) owns 32 tiles (TWB x THB, e.g. 8 x 4 -> 32 x 16 output pixels) x 64 output channels: wave (wq, wh) holds
// the 36 frequencies of 16 tiles x 16 channels (144 accumulator registers), so the output transform never leaves the lane.
// Frequencies are stored four to a 16-byte word ([f / 4][row / 16][c][row % 16][f % 4]): ONE ds_read_b128 of V and one of U feed four MFMAs.
// Per 4-input-channel stage the raw (4 THB + 2) x (4 TWB + 2) x 4-channel patch and the 36 KB U stage arrive by LDS-DMA (U is
// stored in exactly the LDS order), every wave transforms its share of the (tile, channel) patches -- four threads per patch, one
// per row group of B^T d: row 0, row 5, rows (1, 2), rows (3, 4); all four run the SAME instruction sequence on wave-uniform
// row offsets / coefficients (a branch per role made hipcc copy the 144 accumulators at every merge: 350 spilled registers) --
// and the MFMA loop reads 16-byte fragments one step ahead.  Patch, V and U are double-buffered: 133 KB of LDS, one workgroup
// per CU, two waves per SIMD.
#include "ud_common.h"
#include "ud_prof.h"
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <algorithm>

#ifndef UD_W4_ABL
#define UD_W4_ABL 0      // development ablations (tools/_exp/w4): 1 no transform, 2 no stage DMA, 4 no epilogue, 8 no MFMA, 16 no stage barrier
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 32, kTN = 64, kKC = 4, kFQ = 9;
constexpr int kUBytes = kFQ * kTN * kKC * 16;      // 36 864
constexpr int kVBytes = kFQ * kT * kKC * 16;       // 18 432
constexpr int kPBytes = 21504;                     // a patch buffer holds EIGHT channels (two stages): 21 DMA pieces of 1 KB
// LDS map: [P0 | P1 | U0 | V0 | V1 | U1 | dump]: a unit's first stages (P0, P1, U0) sit in the low 78 KB, which the epilogue's
// second staging pass does not touch (the next unit's first DMA overlaps it)
constexpr int kP0 = 0, kU0 = 2 * kPBytes, kV0 = kU0 + kUBytes, kV1 = kV0 + kVBytes, kU1 = kV1 + kVBytes;
constexpr int kDump = kU1 + kUBytes;               // 1 KB that swallows the V words a one-row role does not produce and padding DMA pieces
constexpr int kSmem = kDump + 1024;                // 154 624
constexpr int kStageA = 0, kStageB = kDump - 65536;   // output staging: 256 pixel rows x 64 channels per pass
static_assert(kStageB >= kV0 && kV1 == kV0 + kVBytes && 65536 <= kV0 + kVBytes, "LDS map");

struct W4Geom {
  int B, H, W, Cin, Cout, bx, by;   // bx x by tile blocks per image
  unsigned u_bytes;                 // size of the transformed filters
  // Schedule: units (tile block, cout block) [0, n_dp) are walked whole, unit = blockIdx.x, + gridDim.x, ... (data parallel); the
  // stages of units [n_dp, n_units) form ONE sequence of (n_units - n_dp) * Cin / 4 stages that is cut into gridDim.x equal
  // ranges of sk_len stages (stream-K): a workgroup runs its range unit by unit; a unit it covers completely is stored as usual,
  // a piece of a unit goes to the workspace as a partial tile (slot 2 * workgroup + (0: first unit of its range, 1: last)) and
  // k_wino4_fixup adds the pieces in ascending stage order.  A layer of 276 units on 256 CUs takes 1.08 unit times this way
  // instead of 2, a 552-unit layer 2.4 instead of 3.
  int n_dp, n_units, sk_len;
  float* partial;                   // [2 * gridDim.x][512 rows][64] fp32 partial tiles (before bias / BatchNorm / ReLU)
};
struct W4Ep {
  const float* bias;
  const float* scale;               // folded eval-mode BatchNorm: v * scale + shift after the bias (both or neither)
  const float* shift;
  const float* residual;
  int relu;
  float* stats;                     // [blocks][Cout][2] per-workgroup (sum, sum of squares) of the stored outputs, or nullptr
};

// LDS-DMA of 16 bytes per lane through a raw buffer descriptor: LDS[lds_wave_base + lane * 16] = buffer[voff + soff]; a byte
// offset past the descriptor's size reads zeros (image borders, padding pieces) -- no pointer select, no per-lane 64-bit address
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(size_t)lds_wave_base, 16, voff, soff, 0, 0);
}
constexpr unsigned kOob = 0xFFF00000u;                // a voffset no tensor reaches (the launcher checks): reads zeros

// Packed fp32 arithmetic by hand: hipcc splits about half of the f32x2 expressions of the transforms into scalar pairs (47 VALU
// instructions per stage instead of 30), and beside fp32 MFMAs every VALU instruction costs its issue time.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(b), "v"(c));      // a: a wave-uniform coefficient pair (SGPRs)
  return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "s"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// c * (w.lo, w.lo) + (z.lo, z.lo) / the same on the high halves of w and z
__device__ __forceinline__ f32x2 pk_fma_lo(f32x2 c, f32x2 w, f32x2 z) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "s"(c), "v"(w), "v"(z));
  return d;
}
__device__ __forceinline__ f32x2 pk_fma_hi(f32x2 c, f32x2 w, f32x2 z) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(c), "v"(w), "v"(z));
  return d;
}
// c * (w.hi, w.hi) + z
__device__ __forceinline__ f32x2 pk_fma_whi(f32x2 c, f32x2 w, f32x2 z) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(c), "v"(w), "v"(z));
  return d;
}
__device__ __forceinline__ f32x2 pk_mul_whi(f32x2 c, f32x2 w) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "s"(c), "v"(w));
  return d;
}

// frequency order: f' = 6 * slot(i) + j for the element (row i, column j) of the 6 x 6 transform, slot = position of i in
// (0, 5, 1, 2, 3, 4) -- the three row pairs that share their arithmetic are 12 consecutive frequencies = three 16-byte words
__host__ __device__ constexpr int w4_slot(int i) { return i == 0 ? 0 : i == 5 ? 1 : i + 1; }
// ... and within a row the columns in the order (0, 5, 1, 3, 2, 4): the row pass produces the pairs (v0, v5), (v1, v3), (v2, v4) with
// packed fp32 instructions (fp32 MFMAs run on the vector ALUs -- every VALU instruction beside them costs its full issue time)
__host__ __device__ constexpr int w4_cpos(int j) { return j == 0 ? 0 : j == 5 ? 1 : j == 1 ? 2 : j == 3 ? 3 : j == 2 ? 4 : 5; }
__host__ __device__ constexpr int w4_f(int i, int j) { return 6 * w4_slot(i) + w4_cpos(j); }

// U[nb][cc][fq][nl][cl][fp] from g'[n][ky][kx][c] = w[n * s_n + c * s_c + ky' * s_y + kx' * s_x] (ky' = flip ? 2 - ky : ky): the forward
// transform takes (n, c) = (Cout, Cin) of the parameter, the data gradient (n, c) = (Cin, Cout) with flip = 1.  Rows / channels
// past N / C are zero.
__global__ void k_wino4_weights(const float* __restrict__ w, long long s_n, long long s_c, long long s_y, long long s_x, int N,
                                int C, int flip, float* __restrict__ U, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cl = (int)(i & 3), nl = (int)((i >> 2) & 63);
  const long long blk = i >> 8;                        // nb * nch + cc
  const int nch = (C + 3) / 4;
  const int nb = (int)(blk / nch), cc = (int)(blk - (long long)nb * nch);
  const int n = nb * 64 + nl, c = cc * 4 + cl;
  float g[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      g[ky][kx] = (n < N && c < C) ? w[n * s_n + c * s_c + (flip ? 2 - ky : ky) * s_y + (flip ? 2 - kx : kx) * s_x] : 0.f;
  // G = [[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6], [1/24, -1/12, 1/6], [0, 0, 1]]
  float t[6][3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const float a = g[0][kx], b = g[1][kx], cc2 = g[2][kx];
    t[0][kx] = 0.25f * a;
    t[1][kx] = (-1.f / 6.f) * (a + b + cc2);
    t[2][kx] = (-1.f / 6.f) * (a - b + cc2);
    t[3][kx] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * cc2;
    t[4][kx] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * cc2;
    t[5][kx] = cc2;
  }
  // [fq][wq = nl / 16][c][li = nl % 16][f % 4]: the 64 lanes (li, c) of an MFMA B fragment read 1 KB contiguous (lane * 16 bytes:
  // ds_read_b128 is bank-conflict-free on linear addresses, 4-way conflicted on a [row][c] order)
  float* out = U + blk * (kFQ * 64 * 16) + (nl >> 4) * 256 + cl * 64 + (nl & 15) * 4;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const float a = t[r][0], b = t[r][1], cc2 = t[r][2];
    float u[6];
    u[0] = 0.25f * a;
    u[1] = (-1.f / 6.f) * (a + b + cc2);
    u[2] = (-1.f / 6.f) * (a - b + cc2);
    u[3] = (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * cc2;
    u[4] = (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * cc2;
    u[5] = cc2;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int f = w4_f(r, j);
      out[(f >> 2) * (64 * 16) + (f & 3)] = u[j];
    }
  }
}

template <int N>
struct IC {
  static constexpr int value = N;
};
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(IC<Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::forward<F>(f), std::make_integer_sequence<int, N>{});
}

// The fused tail of an output row (4 channels of one pixel): bias, folded BatchNorm, residual, ReLU, store, BatchNorm partial sums.
// Row numbering of a pass (256 rows): row = tl * 16 + py * 4 + px, tl = wh * 8 + g * 2 + (r & 1) <-> tile slot 16 wh + 4 g + 2 pass + (r & 1).
template <int TWB, int THB>
struct W4Tail {
  const W4Geom& gm;
  const W4Ep& ep;
  int b, ty0, tx0, n0, tid, c4, n;
  float4 bv, scv, shv, s1, s2;
  __device__ W4Tail(const W4Geom& gm_, const W4Ep& ep_, int b_, int ty0_, int tx0_, int n0_, int tid_)
      : gm(gm_), ep(ep_), b(b_), ty0(ty0_), tx0(tx0_), n0(n0_), tid(tid_) {
    c4 = (tid & 15) * 4, n = n0 + c4;
    bv = make_float4(0.f, 0.f, 0.f, 0.f), scv = make_float4(1.f, 1.f, 1.f, 1.f), shv = bv, s1 = bv, s2 = bv;
    if (ep.bias && n < gm.Cout) bv = *reinterpret_cast<const float4*>(ep.bias + n);
    if (ep.scale && n < gm.Cout) {
      scv = *reinterpret_cast<const float4*>(ep.scale + n);
      shv = *reinterpret_cast<const float4*>(ep.shift + n);
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ y, int pass, int row, float4 v) {
    static_assert((TWB & (TWB - 1)) == 0, "tile-block width: a power of two (shift / mask below)");
    const int tl = row >> 4, p = row & 15;
    const int slot = 16 * (tl >> 3) + 4 * ((tl >> 1) & 3) + 2 * pass + (tl & 1);      // = 4 k + 2 pass + (tid >> 8) for row = (tid >> 4) + 32 k
    const int sy = slot / TWB, sx = slot & (TWB - 1);
    const int gy = 4 * (ty0 + sy) + (p >> 2), gx = 4 * (tx0 + sx) + (p & 3);
    if (slot >= TWB * THB || gy >= gm.H || gx >= gm.W || n >= gm.Cout) return;
    if (ep.bias) { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
    if (ep.scale) {
      v.x = v.x * scv.x + shv.x; v.y = v.y * scv.y + shv.y; v.z = v.z * scv.z + shv.z; v.w = v.w * scv.w + shv.w;
    }
    const size_t off = ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cout + n;
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
  // a thread keeps ONE 4-channel piece over its 16 rows: reduce the 32 row groups through LDS (red: 16 KB), fixed order
  __device__ void reduce_stats(float* red, int blk_lin) {
    const int grp = tid >> 4;
    const float a4[4] = {s1.x, s1.y, s1.z, s1.w}, q4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[(grp * 64 + c4 + e) * 2] = a4[e];
      red[(grp * 64 + c4 + e) * 2 + 1] = q4[e];
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < gm.Cout) {
      double ad = 0.0, qd = 0.0;          // the 32 row-group sums combine in double (sum of squares minus mean^2 comes next)
      for (int k = 0; k < 32; ++k) {
        ad += (double)red[(k * 64 + tid) * 2];
        qd += (double)red[(k * 64 + tid) * 2 + 1];
      }
      ep.stats[((size_t)blk_lin * gm.Cout + n0 + tid) * 2] = (float)ad;
      ep.stats[((size_t)blk_lin * gm.Cout + n0 + tid) * 2 + 1] = (float)qd;
    }
  }
};

// LDS operations of the input transform that ride in MFMA step fq (after the step's fragment loads): four patch-row reads
// (3 ds_read2_b32 each) in steps 0-3, the four V writes in step 7
__device__ constexpr int kStepOps[9] = {3, 3, 3, 3, 0, 0, 0, 4, 0};
__host__ __device__ constexpr int w4_wait(int fq) {   // LDS operations issued after the fragment loads of step fq (fq < 8)
  return (fq > 0 ? kStepOps[fq - 1] : 0) + 2 + kStepOps[fq];
}

template <int TWB, int THB>
__global__ __launch_bounds__(512) void k_conv3x3_wino4_f32(const float* __restrict__ x, const float* __restrict__ U,
                                                           float* __restrict__ y, W4Geom gm, W4Ep ep) {
  constexpr int PW = 4 * TWB + 2, PH = 4 * THB + 2;
  // A patch row in LDS = [channel half h][RP slots of 16 bytes]: the DMA fetches 32 bytes per pixel (two lanes RP apart), so a
  // 128-byte line of x is requested four times per unit instead of eight (with 16 bytes per pixel and stage the patch DMA alone
  // cost 13 % of the kernel: the L2 -> CU path, not the MFMA pipe, set the pace).  Rows of tiles are 4 patch rows apart:
  // 4 * 2 RP * 16 bytes must step the 256-byte bank period by half of it for the two tile rows of an 8 x 4 block -> RP odd.
  constexpr int RP = (TWB == 16) ? PW : (PW | 1);
  constexpr int RB = 2 * RP * 16;                                   // bytes per patch row
  constexpr int PP = PH * 2 * RP, kPInstr = (PP + 63) / 64;
  static_assert(TWB * THB <= kT && kPInstr * 1024 <= kPBytes && kPInstr <= 24, "tile block");
  constexpr int O0 = 0, O1 = TWB + 1, O2 = 2 * TWB + 2, O3 = 3 * TWB + 2;    // column planes x % 4 = 0 / 1 / 2 / 3 of a patch row
  static_assert((O3 + 0) * 4 <= 255 && (O1 + 1) * 4 <= 255, "ds_read2 offsets");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = (unsigned)(size_t)smem;      // LDS byte address of the dynamic segment (0 in practice)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wq = wave & 3, wh = wave >> 2;
  const int nblocks = gm.B * gm.bx * gm.by;
  const int nchunks = gm.Cin / kKC;                    // even (Cin % 8 == 0)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)gm.B * gm.H * gm.W * gm.Cin * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, (int)gm.u_bytes, 0x00020000);

  int unit, cbi, blk_lin, b, ty0, tx0, n0;
  unsigned po[3];                  // byte offset of this lane's 16 bytes of patch piece wave + 8 i within x (channels 0-3 / 4-7), kOob: zeros
  unsigned uo;                     // wave-uniform byte offset of this wave's first piece of the next U stage
  unsigned pair_off;               // byte offset of the next patch pair's channels
  int nst;                         // stages of the current piece (even)
  auto setup = [&](int unit_, int c0, int c1) {
    unit = unit_;
    nst = c1 - c0;
    cbi = unit / nblocks;
    const int ru_ = unit - cbi * nblocks;
    int blk;
    {   // consecutive items go round the 8 XCDs: XCD k walks its own contiguous range of tile blocks (shared halos stay in its L2)
      const int base = nblocks >> 3, extra = nblocks & 7, k = ru_ & 7;
      blk = k * base + min(k, extra) + (ru_ >> 3);
    }
    blk_lin = blk;
    b = blk / (gm.bx * gm.by);
    blk -= b * gm.bx * gm.by;
    ty0 = (blk / gm.bx) * THB, tx0 = (blk % gm.bx) * TWB;      // in tiles
    n0 = cbi * kTN;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // patch DMA piece pi fills 16-byte slots [64 pi, 64 pi + 64): slot -> (row, channel half, plane position) -> pixel
      const int pi = wave + 8 * i;
      const int qq = pi * 64 + lane;
      const int qy = qq / (2 * RP), qr = qq - qy * (2 * RP);
      const int qh = qr >= RP ? 1 : 0, qs = qr - qh * RP;
      const int qx = qs < O1 ? 4 * qs : qs < O2 ? 4 * (qs - O1) + 1 : qs < O3 ? 4 * (qs - O2) + 2 : 4 * (qs - O3) + 3;
      const int gy = 4 * ty0 + qy - 1, gx = 4 * tx0 + qx - 1;
      const bool ok = qq < PP && qs < PW && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W && !(UD_W4_ABL & (2 | 128));
      po[i] = ok ? (unsigned)(((b * gm.H + gy) * gm.W + gx) * gm.Cin + qh * 4) * 4u : kOob;
    }
    uo = (unsigned)(cbi * nchunks + c0) * (unsigned)kUBytes + wave * 1024;
    pair_off = c0 * 16;            // 4 channels = 16 bytes per stage
  };
  // patch pair (8 channels) -> patch buffer `buf`: ALWAYS three pieces per wave (a constant count lets the stage wait with
  // vmcnt(3) for everything older); pieces past the patch land in the dump
  auto patch_piece = [&](int i, int buf) {
    const int pi = wave + 8 * i;
    dma16(rx, po[i], pair_off, pi < kPInstr ? sbase + kP0 + buf * kPBytes + pi * 1024 : sbase + kDump);
  };
  const unsigned lane16 = lane * 16;
  auto stage_u = [&](int buf) {
    const unsigned ub = sbase + (buf ? kU1 : kU0) + wave * 1024;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i < 4 || wave < 4) dma16(ru, lane16, uo + i * 8192, ub + i * 8192);
  };

  // ---- input transform: thread = (row group of B^T d, tile, channel); wave w: group w >> 1, tiles 16 (w & 1) .. + 15.
  // Every group evaluates  X = a1 r1 + a2 r2,  Y = b3 r3 + b4 r4,  out_a = X + Y,  out_b = X - Y  over patch rows r1..r4:
  //   group 0 (row 0 of B^T d):   X = 4 d0 - 5 d2, Y = d4          -> out_a (out_b is not stored)
  //   group 1 (row 5):            X = 4 d1 - 5 d3, Y = -d5         -> out_b
  //   group 2 (rows 1, 2):        X = d4 - 4 d2,   Y = d3 - 4 d1   -> out_a = row 1, out_b = row 2
  //   group 3 (rows 3, 4):        X = d4 - d2,     Y = 2 d3 - 2 d1 -> out_a = row 3, out_b = row 4
  // then the row pass on out_a / out_b.  Frequencies: group 0 -> f' 0..5, group 1 -> 6..11, group 2 -> 12..23, group 3 -> 24..35.
  const int grp = wave >> 1;
  const int ttile = (wave & 1) * 16 + (lane & 15), tc = lane >> 4;       // the MFMA A-fragment lane order: V words land lane-linear
  const int ttc = min(ttile, TWB * THB - 1);
  const int tyl = ttc / TWB, txl = ttc - tyl * TWB;
  const unsigned tprow = sbase + kP0 + (4 * tyl) * RB + txl * 16 + tc * 4;     // patch row 0 of this thread's tile, buffer 0, half 0
  const int rows4 = grp == 0 ? 0x4420 : grp == 1 ? 0x5531 : 0x1324;            // r1 | r2 << 4 | r3 << 8 | r4 << 12
  const unsigned ro1 = (rows4 & 15) * RB, ro2 = ((rows4 >> 4) & 15) * RB, ro3 = ((rows4 >> 8) & 15) * RB, ro4 = ((rows4 >> 12) & 15) * RB;
  const float ca1 = grp < 2 ? 4.f : 1.f, ca2 = grp < 2 ? -5.f : grp == 2 ? -4.f : -1.f;
  const float cb3 = grp == 0 ? 1.f : grp == 1 ? -1.f : grp == 2 ? 1.f : 2.f, cb4 = grp < 2 ? 0.f : grp == 2 ? -4.f : -2.f;
  // V words [fq][tile / 16][c][tile % 16][4]: A = (a0..a3), B = (a4, a5) low half, C = (b0, b1) high half, D = (b2..b5)
  const unsigned tvbase = sbase + kV0 + (wave & 1) * 1024 + lane * 16;
  const unsigned tdump = sbase + kDump + lane * 16;
  const int q0 = grp < 2 ? 0 : grp == 2 ? 3 : 6;
  const unsigned wA = grp == 1 ? tdump : tvbase + q0 * (kT * kKC * 16);
  const unsigned wB = grp == 1 ? tdump : tvbase + (q0 + 1) * (kT * kKC * 16);
  const unsigned wC = grp == 0 ? tdump : tvbase + (q0 + 1) * (kT * kKC * 16) + 8;
  const unsigned wD = grp == 0 ? tdump : tvbase + (q0 + 2) * (kT * kKC * 16);
  const unsigned vdelta01 = (grp == 1 ? 0 : kVBytes), vdelta23 = (grp == 0 ? 0 : kVBytes);   // V buffer 1 - buffer 0 (0 for the dump)

#define W4_ROWREAD(D, RO)                                                                                               \
  do {                                                                                                                  \
    const unsigned a_ = tprow + (RO);                                                                                   \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[0]) : "v"(a_), "n"(O0 * 4), "n"(O1 * 4));              /* j = 0, 1 */ \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[1]) : "v"(a_), "n"(O2 * 4), "n"(O3 * 4));              /* j = 2, 3 */ \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[2]) : "v"(a_), "n"((O0 + 1) * 4), "n"((O1 + 1) * 4));  /* j = 4, 5 */ \
  } while (0)

  // 1-D transform along a row, v = w B (B^T of F(4, 3): [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0;
  // 0 4 0 -5 0 1]) in six packed instructions: w = pairs (w0, w1), (w2, w3), (w4, w5) -> (v0, v5), (v1, v3), (v2, v4)
  auto row_pass = [](const f32x2 (&w)[3], f32x2 (&v)[3]) {
    const f32x2 c4 = {4.f, 4.f}, c5 = {-5.f, -5.f}, c41 = {-4.f, -1.f}, c12 = {1.f, 2.f}, cn12 = {-1.f, -2.f};
    v[0] = pk_fma(c4, w[0], pk_fma(c5, w[1], w[2]));
    const f32x2 e = pk_fma_lo(c41, w[1], w[2]);       // (w4 - 4 w2, w4 - w2)
    const f32x2 o = pk_fma_hi(c41, w[0], w[1]);       // (w3 - 4 w1, w3 - w1)
    v[1] = pk_fma(c12, o, e);
    v[2] = pk_fma(cn12, o, e);
  };
  const f32x2 ca1v = {ca1, ca1}, ca2v = {ca2, ca2}, cb3v = {cb3, cb3}, cb4v = {cb4, cb4};

  // plain (not interleaved) transform of the unit's first stage: patch buffer 0, half 0 -> V buffer 0 (same arithmetic as in the stage)
  auto transform_first = [&]() {
    auto px2 = [&](unsigned ro, int k) {      // columns (2 k, 2 k + 1) of a patch row
      const int oa = k == 0 ? O0 : k == 1 ? O2 : O0 + 1, ob = k == 0 ? O1 : k == 1 ? O3 : O1 + 1;
      const char* q = smem + (tprow - sbase) + ro;
      return (f32x2){*reinterpret_cast<const float*>(q + oa * 16), *reinterpret_cast<const float*>(q + ob * 16)};
    };
    f32x2 wa[3], wb[3], va[3], vb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f32x2 xx = __builtin_elementwise_fma(ca1v, px2(ro1, k), ca2v * px2(ro2, k));
      const f32x2 yy = __builtin_elementwise_fma(cb4v, px2(ro4, k), cb3v * px2(ro3, k));
      wa[k] = xx + yy;
      wb[k] = xx - yy;
    }
    row_pass(wa, va);
    row_pass(wb, vb);
    *reinterpret_cast<f32x4*>(smem + (wA - sbase)) = (f32x4){va[0].x, va[0].y, va[1].x, va[1].y};
    *reinterpret_cast<f32x2*>(smem + (wB - sbase)) = va[2];
    *reinterpret_cast<f32x2*>(smem + (wC - sbase)) = vb[0];
    *reinterpret_cast<f32x4*>(smem + (wD - sbase)) = (f32x4){vb[1].x, vb[1].y, vb[2].x, vb[2].y};
  };

  f32x4 acc[36];
  f32x4 qa[2], qb[2];                                  // MFMA fragments of step fq of a stage of parity CB: buffer (fq + CB) & 1
  const unsigned fa = sbase + kV0 + wh * 1024 + lane * 16, fb0 = sbase + wq * 1024 + lane * 16;

  auto first_stages = [&]() {      // the DMA an item needs before its first stages: patch pairs 0 and 1, U(0)
#pragma unroll
    for (int i = 0; i < 3; ++i) patch_piece(i, 0);
    stage_u(0);
    uo += kUBytes;
    pair_off += nst > 2 ? 32 : 0;             // (a one-pair piece fetches its pair again: never read)
#pragma unroll
    for (int i = 0; i < 3; ++i) patch_piece(i, 1);
    pair_off += nst > 4 ? 32 : 0;
  };

  // One stage = 9 steps of (2 fragment loads one step ahead, 4 MFMAs) on V / U of parity CB = chunk & 1.  Riding along:
  //   * the input transform of chunk + 1 (patch pair (chunk + 1) / 2, channel half CB ^ 1) -> V[CB ^ 1]: patch-row reads in steps
  //     0-3, arithmetic from two steps after a row was requested (LDS returns in order, so the step's s_waitcnt counts exactly
  //     the operations issued after the fragments it needs), V writes in step 7;
  //   * this wave's LDS-DMA pieces, one per step (a piece blocks the wave's issue for 60+ cycles: in a row at the top of the
  //     stage they idle the MFMA pipe): U(chunk + 1) in steps 0-4; odd stages: patch pair (chunk + 3) / 2 in steps 5-7.  The
  //     piece's last stages issue them too (no branch in the loop): U of the last stage again, the last patch pair again -- into
  //     buffers nobody reads any more (`chunk` counts from the piece's first stage);
  //   * the stage barrier BEFORE the last step's MFMAs: by then the wave holds the step-8 fragments in registers, has written its
  //     share of V[CB ^ 1] and waits for its DMA; right after the barrier the next stage's first fragments are requested and
  //     arrive under the four MFMAs still to issue -- no bubble at the stage boundary, no barrier at the top of a stage.
  auto stage = [&](auto cb_c, int chunk) {
    constexpr int CB = decltype(cb_c)::value;
    constexpr bool TRF = !(UD_W4_ABL & 1);
    const unsigned pa = fa + CB * kVBytes, pb = fb0 + (CB ? kU1 : kU0);
    const unsigned pan = fa + (CB ^ 1) * kVBytes, pbn = fb0 + (CB ? kU0 : kU1);        // the next stage's
    const unsigned poff = (((chunk + 1) >> 1) & 1) * kPBytes + (CB ^ 1) * (RP * 16);   // patch pair buffer + channel half of chunk + 1
    f32x2 r1[3], r2[3], r3[3], r4[3];
    f32x2 wa[3], wb[3], va[3], vb[3];
    const unsigned ub = sbase + (CB ? kU0 : kU1) + wave * 1024;
    const int pbuf = ((chunk + 3) >> 1) & 1;
    static_for<9>([&](auto fq_c) {
      constexpr int fq = decltype(fq_c)::value;
      constexpr int cur = (fq + CB) & 1, nxt = cur ^ 1;
      if constexpr (fq < 8) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qa[nxt]) : "v"(pa), "n"((fq + 1) * (kT * kKC * 16)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qb[nxt]) : "v"(pb), "n"((fq + 1) * (kTN * kKC * 16)));
      }
      // ---- LDS operations of the transform in this step (kStepOps counts them)
      if constexpr (TRF) {
        if constexpr (fq == 0) W4_ROWREAD(r1, ro1 + poff);
        if constexpr (fq == 1) W4_ROWREAD(r2, ro2 + poff);
        if constexpr (fq == 2) W4_ROWREAD(r3, ro3 + poff);
        if constexpr (fq == 3) W4_ROWREAD(r4, ro4 + poff);
        if constexpr (fq == 7) {
          asm volatile("ds_write_b128 %0, %1" ::"v"(wA + (CB ? 0 : vdelta01)), "v"((f32x4){va[0].x, va[0].y, va[1].x, va[1].y}) : "memory");
          asm volatile("ds_write_b64 %0, %1" ::"v"(wB + (CB ? 0 : vdelta01)), "v"(va[2]) : "memory");
          asm volatile("ds_write_b64 %0, %1" ::"v"(wC + (CB ? 0 : vdelta23)), "v"(vb[0]) : "memory");
          asm volatile("ds_write_b128 %0, %1" ::"v"(wD + (CB ? 0 : vdelta23)), "v"((f32x4){vb[1].x, vb[1].y, vb[2].x, vb[2].y}) : "memory");
        }
      }
      // ---- wait for this step's fragments; a patch row requested in step i has arrived by the wait of step i + 2 and is bound
      // to that wait (an in/out operand), from which on the compiler may read it
      if constexpr (fq == 8) {
        // fragments of step 8 and the V writes of step 7 are done; DMA: an even stage needs everything it issued (U(chunk + 1)),
        // an odd one may leave its three patch pieces (needed two stages on) in flight
        if constexpr (CB == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(qa[cur]), "+v"(qb[cur])::"memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" : "+v"(qa[cur]), "+v"(qb[cur])::"memory");
        if (!(UD_W4_ABL & 16)) __builtin_amdgcn_s_barrier();
        asm volatile("ds_read_b128 %0, %1" : "=v"(qa[nxt]) : "v"(pan));
        asm volatile("ds_read_b128 %0, %1" : "=v"(qb[nxt]) : "v"(pbn));
      } else {
        constexpr int WN = TRF ? w4_wait(fq) : 2;
        if constexpr (TRF && fq == 3)
          asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r2[0]), "+v"(r2[1]), "+v"(r2[2]) : "n"(WN));
        else if constexpr (TRF && fq == 4)
          asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(r3[0]), "+v"(r3[1]), "+v"(r3[2]) : "n"(WN));
        else if constexpr (TRF && fq == 5)
          asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]) : "n"(WN));
        else
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(qa[cur]), "+v"(qb[cur]) : "n"(WN));
      }
      // ---- the step's four MFMAs
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (!(UD_W4_ABL & 8) || fq == 0)
          acc[4 * fq + p] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[cur][p], qb[cur][p], acc[4 * fq + p], 0, 0, 0);
      if (!(UD_W4_ABL & (2 | 64))) {
        if constexpr (fq < 4) dma16(ru, lane16, uo + fq * 8192, ub + fq * 8192);
        else if constexpr (fq == 4) {
          if (wave < 4) dma16(ru, lane16, uo + fq * 8192, ub + fq * 8192);
        }
      }
      if constexpr (fq >= 5 && fq < 8 && CB == 1) patch_piece(fq - 5, pbuf);
      // ---- transform arithmetic that became possible with this step's wait (packed fp32: two columns per instruction)
      if constexpr (TRF) {
        if constexpr (fq == 3) {
#pragma unroll
          for (int k = 0; k < 3; ++k) wa[k] = pk_fma(ca1v, r1[k], pk_mul(ca2v, r2[k]));          // X
        }
        if constexpr (fq == 4) {
#pragma unroll
          for (int k = 0; k < 3; ++k) wb[k] = pk_mul(cb3v, r3[k]);
        }
        if constexpr (fq == 5) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const f32x2 yy = pk_fma(cb4v, r4[k], wb[k]);                                  // Y
            wb[k] = pk_sub(wa[k], yy);
            wa[k] = pk_add(wa[k], yy);
          }
        }
        if constexpr (fq == 6) {
          row_pass(wa, va);
          row_pass(wb, vb);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    // next U stage (the unit's last two stages stay on the last one); odd stages: next patch pair (clamped to the last one)
    uo += chunk + 2 < nst ? kUBytes : 0;
    if constexpr (CB == 1) pair_off += chunk + 5 < nst ? 32 : 0;
  };

  // ---- the workgroup's pieces: whole units first (data parallel), then its stream-K range
  const int G = gridDim.x;
  int dp_next = blockIdx.x;                                   // next data-parallel unit
  int sk_lo = gm.n_dp * nchunks + blockIdx.x * gm.sk_len;     // rest of this workgroup's stream-K range, in global stage numbers
  const int sk_hi = min(sk_lo + gm.sk_len, gm.n_units * nchunks);
  int p_unit, p_c0, p_c1, p_slot;                             // piece: unit, stages [c0, c1), partial slot (-1: the whole unit)
  auto next_piece = [&]() -> bool {
    if (dp_next < gm.n_dp) {
      p_unit = dp_next, p_c0 = 0, p_c1 = nchunks, p_slot = -1;
      dp_next += G;
      return true;
    }
    if (sk_lo >= sk_hi) return false;
    p_unit = sk_lo / nchunks;
    p_c0 = sk_lo - p_unit * nchunks;
    p_c1 = min(nchunks, p_c0 + sk_hi - sk_lo);
    const bool first = sk_lo == gm.n_dp * nchunks + (int)blockIdx.x * gm.sk_len;
    p_slot = (p_c0 == 0 && p_c1 == nchunks) ? -1 : 2 * blockIdx.x + (first ? 0 : 1);
    sk_lo += p_c1 - p_c0;
    return true;
  };
  if (!next_piece()) return;
  setup(p_unit, p_c0, p_c1);
  first_stages();
#pragma unroll 1
  for (;;) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA (and the previous piece's stores) are done
    __syncthreads();                                       // first stages in LDS; everybody has left the previous piece's epilogue
    transform_first();
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                       // V(0) complete
    asm volatile("ds_read_b128 %0, %1" : "=v"(qa[0]) : "v"(fa));
    asm volatile("ds_read_b128 %0, %1" : "=v"(qb[0]) : "v"(fb0 + kU0));
#pragma unroll 1
    for (int chunk = 0; chunk < nst; chunk += 2) {
      stage(IC<0>{}, chunk);
      stage(IC<1>{}, chunk + 1);
    }
    // (the last stage's barrier came before its last MFMAs: nobody reads U / V any more, except for the fragments that stage
    // requested for a stage that does not exist -- wait for them before their registers are reused)
    // -- and for the DMA pieces the last stages issued into buffers nobody reads (they must not land in the output staging)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(qa[0]), "+v"(qb[0]), "+v"(qa[1]), "+v"(qb[1])::"memory");
    __syncthreads();
    const int e_b = b, e_ty0 = ty0, e_tx0 = tx0, e_n0 = n0, e_blk = blk_lin, e_slot = p_slot;
    const bool more = next_piece();
    // ---- epilogue: output transform in registers -> fp32 pixel rows in LDS (two passes of 16 tiles: accumulator rows r = 2 pass,
    // 2 pass + 1 of every wave), then coalesced 16-byte stores with the fused tail.  Row = tl * 16 + py * 4 + px with
    // tl = wh * 8 + g * 2 + (r & 1); the 16-channel group is XOR-ed with g (the four lane groups write rows 8 KB apart otherwise).
    W4Tail<TWB, THB> tail(gm, ep, e_b, e_ty0, e_tx0, e_n0, tid);
#pragma unroll
    for (int pass = (UD_W4_ABL & 4) ? 1 : 0; pass < 2; ++pass) {
      float* Os = reinterpret_cast<float*>(smem + (pass ? kStageB : kStageA));
      {
        // A^T M A for the accumulator rows r = 2 pass, 2 pass + 1 together (a register pair of every accumulator): packed fp32
        auto m2 = [&](int i, int j) { return (f32x2){acc[w4_f(i, j)][2 * pass], acc[w4_f(i, j)][2 * pass + 1]}; };
        const f32x2 c2 = {2.f, 2.f}, c4 = {4.f, 4.f}, c8 = {8.f, 8.f};
        f32x2 t[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const f32x2 m1 = m2(1, j), mm2 = m2(2, j), m3 = m2(3, j), m4 = m2(4, j);
          const f32x2 sa = m1 + mm2, da = m1 - mm2, sb = m3 + m4, db = m3 - m4;
          t[0][j] = m2(0, j) + sa + sb;
          t[1][j] = __builtin_elementwise_fma(c2, db, da);
          t[2][j] = __builtin_elementwise_fma(c4, sb, sa);
          t[3][j] = __builtin_elementwise_fma(c8, db, da) + m2(5, j);
        }
        // rows of tile tl = wh * 8 + g * 2 (+ 1 for the pair's second half: 16 rows = 4 096 bytes on)
        const unsigned o = sbase + (pass ? kStageB : kStageA) + ((wh * 8 + g * 2) * 16 * 64 + (16 * (wq ^ g) + li)) * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const f32x2 sa = t[a][1] + t[a][2], da = t[a][1] - t[a][2], sb = t[a][3] + t[a][4], db = t[a][3] - t[a][4];
          const f32x2 y0 = t[a][0] + sa + sb, y1 = __builtin_elementwise_fma(c2, db, da), y2 = __builtin_elementwise_fma(c4, sb, sa),
                      y3 = __builtin_elementwise_fma(c8, db, da) + t[a][5];
          asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(o), "v"(y0.x), "v"(y0.y), "n"(a * 4 + 0), "n"(a * 4 + 16) : "memory");
          asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(o), "v"(y1.x), "v"(y1.y), "n"(a * 4 + 1), "n"(a * 4 + 17) : "memory");
          asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(o), "v"(y2.x), "v"(y2.y), "n"(a * 4 + 2), "n"(a * 4 + 18) : "memory");
          asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(o), "v"(y3.x), "v"(y3.y), "n"(a * 4 + 3), "n"(a * 4 + 19) : "memory");
        }
      }
      __syncthreads();
      if (e_slot < 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int row = (tid >> 4) + 32 * k;
          tail.store(y, pass, row, *reinterpret_cast<const float4*>(Os + row * 64 + (tail.c4 ^ (16 * ((row >> 5) & 3)))));
        }
      } else {      // a piece of a unit: the raw rows go to the workspace, k_wino4_fixup adds the pieces and applies the tail
        float* dst = gm.partial + ((size_t)e_slot * 512 + pass * 256) * 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int row = (tid >> 4) + 32 * k;
          *reinterpret_cast<float4*>(dst + row * 64 + tail.c4) =
              *reinterpret_cast<const float4*>(Os + row * 64 + (tail.c4 ^ (16 * ((row >> 5) & 3))));
        }
      }
      if (pass == 0) {
        __syncthreads();               // staging A has been read: the next piece's first stages may land there
        if (more) {
          setup(p_unit, p_c0, p_c1);
          first_stages();
        }
      }
    }
    if (ep.stats && e_slot < 0) {
      __syncthreads();
      tail.reduce_stats(reinterpret_cast<float*>(smem + kStageB), e_blk);
    }
    if (!more) break;
  }
}

// Sum of the pieces of the units the stream-K schedule cut (ascending stage order: deterministic) + the fused tail.  Two
// workgroups per stream-K unit (one per staging pass = 16 of its 32 tiles), eight independent rows in flight per thread; a unit
// that one workgroup covered completely was stored by k_conv3x3_wino4_f32 itself.  BatchNorm partial sums of a cut unit go to
// the extra rows nblocks + 2 * (unit - n_dp) + pass (zeroed by the launcher); its own row is cleared here.
template <int TWB, int THB>
__global__ __launch_bounds__(512) void k_wino4_fixup(float* __restrict__ y, W4Geom gm, W4Ep ep) {
  __shared__ float red[32 * 64 * 2];
  const int tid = threadIdx.x;
  const int nblocks = gm.B * gm.bx * gm.by, nchunks = gm.Cin / kKC;
  const int u = gm.n_dp + (blockIdx.x >> 1), pass = blockIdx.x & 1;
  const int base = gm.n_dp * nchunks;
  const int a = u * nchunks - base, e = a + nchunks - 1;              // the unit's stages within the stream-K sequence
  const int j0 = a / gm.sk_len, j1 = e / gm.sk_len;
  if (j0 == j1) return;
  const int cbi = u / nblocks, ru = u - cbi * nblocks;
  int blk;
  {
    const int bs = nblocks >> 3, extra = nblocks & 7, k = ru & 7;
    blk = k * bs + min(k, extra) + (ru >> 3);
  }
  const int blk_lin = blk;
  const int b = blk / (gm.bx * gm.by);
  blk -= b * gm.bx * gm.by;
  W4Tail<TWB, THB> tail(gm, ep, b, (blk / gm.bx) * THB, (blk % gm.bx) * TWB, cbi * kTN, tid);
  float4 v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
  for (int j = j0; j <= j1; ++j) {
    const int first_unit = (base + j * gm.sk_len) / nchunks;     // the first unit workgroup j touched
    const int slot = 2 * j + (first_unit == u ? 0 : 1);
    const float* src = gm.partial + ((size_t)slot * 512 + pass * 256 + (tid >> 4)) * 64 + tail.c4;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 p = ud_ldg_stream(src + (size_t)k * 32 * 64);
      v[k].x += p.x; v[k].y += p.y; v[k].z += p.z; v[k].w += p.w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) tail.store(y, pass, (tid >> 4) + 32 * k, v[k]);
  if (ep.stats) {
    tail.reduce_stats(red, nblocks + 2 * (u - gm.n_dp) + pass);
    if (pass == 0 && tid < 64 && tail.n0 + tid < gm.Cout)
      ep.stats[((size_t)blk_lin * gm.Cout + tail.n0 + tid) * 2] = ep.stats[((size_t)blk_lin * gm.Cout + tail.n0 + tid) * 2 + 1] = 0.f;
  }
}

#undef W4_ROWREAD
#undef W4_COL

struct W4Plan {
  int twb, thb, bx, by;
};
// tile-block shape (<= 32 tiles of 4 x 4 outputs) with the fewest blocks for this map
W4Plan w4_plan(int H, int W) {
  static const int shapes[][2] = {{8, 4}, {4, 8}, {16, 2}};
  static const int force = getenv("UD_WINO4_SHAPE") ? atoi(getenv("UD_WINO4_SHAPE")) : -1;
  const int TX = (W + 3) / 4, TY = (H + 3) / 4;
  W4Plan best{};
  long long cost = -1;
  for (int i = 0; i < 3; ++i) {
    if (force >= 0 && i != force) continue;
    const int bx = ud_div_up(TX, shapes[i][0]), by = ud_div_up(TY, shapes[i][1]);
    const long long cst = (long long)bx * by;
    if (cost < 0 || cst < cost) cost = cst, best = W4Plan{shapes[i][0], shapes[i][1], bx, by};
  }
  return best;
}

}  // namespace

namespace {
std::atomic<int> g_sk_mode{-1};        // ud_conv3x3_wino4_stream_k: -1 default (UD_WINO4_SK or 2), 0 never, 1 deep reductions only, 2 whenever a tail exists and beats a whole round
constexpr int kGrid = 256;             // persistent workgroups: one per CU
struct W4Sched {
  int grid, n_dp, sk_len;
};
// data-parallel rounds + a stream-K tail where that beats one more (mostly idle) round; a piece costs ~8 stages of prologue + epilogue
W4Sched w4_schedule(long long units, int nchunks, bool have_ws) {
  static const int env_mode = getenv("UD_WINO4_SK") ? atoi(getenv("UD_WINO4_SK")) : 2;
  const int forced = g_sk_mode.load(std::memory_order_relaxed);
  const int mode = forced >= 0 ? forced : env_mode;
  static const int persist = getenv("UD_WINO4_GRID") ? atoi(getenv("UD_WINO4_GRID")) : kGrid;
  const int G = (int)std::min<long long>(units, persist);
  const int r = (int)(units % G);
  W4Sched sc{G, (int)units, 0};
  // measured (tools/time_wino4.py, fix-up with two workgroups per unit): the tail pays on every layer shape of the step -- 512 -> 64
  // @180^2 407 -> 262 us, 2688 -> 64 2121 -> 1236 us, 128 -> 128 @180^2 179 -> 161 us, 128 -> 128 @32 x 88 x 24 113 -> 88 us
  // (mode 1 keeps it to Cin >= 256)
  if (r == 0 || !have_ws || !mode || (nchunks < 64 && mode < 2)) return sc;
  int len = (int)(((long long)r * nchunks + G - 1) / G);
  len += len & 1;
  if (len + 16 >= (nchunks + 8) * 9 / 10) return sc;
  sc.n_dp = (int)(units - r), sc.sk_len = len;
  return sc;
}
}  // namespace

extern "C" size_t ud_conv3x3_wino4_f32_weight_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)ud_div_up(Cout, 64) * ud_div_up(Cin, 4) * kUBytes;
}

// tile blocks per image of the plan for an H x W map (each 32 tile slots of 4 x 4 outputs): callers compare with
// ceil(H / 4) * ceil(W / 4) to decide whether the map fills the blocks well enough
extern "C" int ud_conv3x3_wino4_f32_blocks(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const W4Plan p = w4_plan(H, W);
  return p.bx * p.by;
}

extern "C" size_t ud_conv3x3_wino4_bnstats_bytes(int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const W4Plan p = w4_plan(H, W);
  return ((size_t)B * p.bx * p.by + 2 * kGrid) * Cout * 2 * sizeof(float);      // + two rows per stream-K unit (< kGrid of them)
}

extern "C" int ud_conv3x3_wino4_f32_weights(const float* w, int64_t s_n, int64_t s_c, int64_t s_y, int64_t s_x, int N, int C,
                                            int flip, float* U, ud_stream_t stream_) {
  if (!w || !U || N <= 0 || C <= 0) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("conv2d.k_wino4_weights", stream);
  const long long total = (long long)ud_div_up(N, 64) * ud_div_up(C, 4) * 256;
  k_wino4_weights<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(w, s_n, s_c, s_y, s_x, N, C, flip, U, total);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// tests / tuning: the stream-K rule (-1: default = UD_WINO4_SK or 2; 0: whole units only; 1: tails of layers with Cin >= 256; 2: every tail that beats one more round)
extern "C" void ud_conv3x3_wino4_stream_k(int mode) { g_sk_mode.store(mode, std::memory_order_relaxed); }

// workspace for the stream-K partial tiles (2 per persistent workgroup)
extern "C" size_t ud_conv3x3_wino4_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  (void)B; (void)H; (void)W; (void)Cin; (void)Cout;
  return (size_t)2 * kGrid * 512 * 64 * sizeof(float);
}

// y = conv3x3(x) (+ bias) (* scale + shift: a folded eval-mode BatchNorm) (+ residual) (ReLU if flags & 1) with U from
// ud_conv3x3_wino4_f32_weights(N = Cout, C = Cin); partial != nullptr: also the per-workgroup BatchNorm partial sums
// ([*slices][Cout][2], same contract as ud_conv3x3_bnstats_nhwc_f32).
extern "C" int ud_conv3x3_wino4_nhwc_f32(const float* x, const float* U, float* y, int B, int H, int W, int Cin, int Cout,
                                         const float* bias, const float* scale, const float* shift, const float* residual,
                                         int flags, float* partial, size_t partial_bytes, int* slices, void* workspace,
                                         size_t workspace_bytes, ud_stream_t stream_) {
  if (!x || !U || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % (2 * kKC) != 0 || Cout % 4 != 0) return UD_ERR_UNSUPPORTED;     // stages come in pairs (8-channel patch fetches)
  hipStream_t stream = (hipStream_t)stream_;
  const W4Plan p = w4_plan(H, W);
  const int nblocks = B * p.bx * p.by;
  const long long units = (long long)nblocks * ud_div_up(Cout, kTN);
  if (units > 0x7fffffffll || (long long)B * H * W * Cin * 4 >= (long long)kOob ||
      ud_conv3x3_wino4_f32_weight_bytes(Cin, Cout) >= (size_t)kOob)
    return UD_ERR_UNSUPPORTED;    // 32-bit byte offsets into x / U (buffer descriptors)
  const bool have_ws = workspace && workspace_bytes >= ud_conv3x3_wino4_f32_workspace_bytes(B, H, W, Cin, Cout);
  const W4Sched sc = w4_schedule(units, Cin / kKC, have_ws);
  W4Geom gm{B, H, W, Cin, Cout, p.bx, p.by, (unsigned)ud_conv3x3_wino4_f32_weight_bytes(Cin, Cout), sc.n_dp, (int)units,
            sc.sk_len > 0 ? sc.sk_len : 2, (float*)workspace};
  W4Ep ep{bias, scale, shift, residual, flags & 1, partial};
  if (partial) {
    const size_t rows = (size_t)nblocks + 2 * (size_t)(units - sc.n_dp);
    if (!slices || partial_bytes < rows * Cout * 2 * sizeof(float)) return UD_ERR_WORKSPACE;
    *slices = (int)rows;
    if (rows > (size_t)nblocks)
      ud_zero_f32_async(partial + (size_t)nblocks * Cout * 2, (rows - nblocks) * (size_t)Cout * 2, stream);
  }
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
#define UD_W4_ATTR(A, Bq) \
  UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wino4_f32<A, Bq>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem))
    UD_W4_ATTR(8, 4); UD_W4_ATTR(4, 8); UD_W4_ATTR(16, 2);
#undef UD_W4_ATTR
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("conv2d.k_conv3x3_wino4_f32", stream);
  const dim3 grid((unsigned)sc.grid);
  const int n_sk = (int)units - sc.n_dp;
#define UD_W4_LAUNCH(A, Bq)                                                                    \
  do {                                                                                         \
    k_conv3x3_wino4_f32<A, Bq><<<grid, 512, kSmem, stream>>>(x, U, y, gm, ep);                 \
    if (n_sk > 0) k_wino4_fixup<A, Bq><<<2 * n_sk, 512, 0, stream>>>(y, gm, ep);               \
  } while (0)
  if (p.twb == 8) UD_W4_LAUNCH(8, 4);
  else if (p.twb == 4) UD_W4_LAUNCH(4, 8);
  else UD_W4_LAUNCH(16, 2);
#undef UD_W4_LAUNCH
  UD_LAUNCH_CHECK();
  return UD_OK;
}
