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
// Frequencies are stored four to a 16-byte word ([f / 4][row][c][f % 4]): ONE ds_read_b128 of V and one of U feed four MFMAs.
// Per 4-input-channel stage the raw (4 THB + 2) x (4 TWB + 2) x 4-channel patch and the 36 KB U stage arrive by LDS-DMA (U is
// stored in exactly the LDS order), waves 0-5 transform the (tile, channel) patches -- three waves' worth of threads per patch:
// rows (0, 5), (1, 2), (3, 4) of B^T d, which share their column pass -- and the MFMA loop reads 16-byte fragments one step ahead.
// Patch, V and U are double-buffered: 132 KB of LDS, one workgroup per CU, two waves per SIMD.
#include "ud_common.h"
#include "ud_prof.h"
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 32, kTN = 64, kKC = 4, kFQ = 9;
constexpr int kUBytes = kFQ * kTN * kKC * 16;      // 36 864
constexpr int kVBytes = kFQ * kT * kKC * 16;       // 18 432
constexpr int kPBytes = 12288;                     // 768 pixel slots of 16 bytes
// LDS map: [P0 | P1 | U0 | V0 | V1 | U1]: a unit's first stages (P0, U0, P1) sit in the low 60 KB, which the epilogue's second
// staging pass does not touch (the next unit's first DMA overlaps it)
constexpr int kP0 = 0, kU0 = 2 * kPBytes, kV0 = kU0 + kUBytes, kV1 = kV0 + kVBytes, kU1 = kV1 + kVBytes;
constexpr int kSmem = kU1 + kUBytes;               // 135 168
constexpr int kStageA = 0, kStageB = kSmem - 65536;   // output staging: 256 pixel rows x 64 channels per pass
static_assert(kStageB >= kV0 && kV1 == kV0 + kVBytes, "LDS map");

struct W4Geom {
  int B, H, W, Cin, Cout, bx, by;   // bx x by tile blocks per image
  int n_items;                      // units (tile block, cout block): a workgroup walks items blockIdx.x, + gridDim.x, ...
};
struct W4Ep {
  const float* bias;
  const float* scale;               // folded eval-mode BatchNorm: v * scale + shift after the bias (both or neither)
  const float* shift;
  const float* residual;
  int relu;
  float* stats;                     // [blocks][Cout][2] per-workgroup (sum, sum of squares) of the stored outputs, or nullptr
};

__device__ __attribute__((aligned(16))) unsigned int g_zero16w4[4];

__device__ __forceinline__ void dma16(const float* src, unsigned lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)(size_t)lds_wave_base, 16, 0, 0);
}

// frequency order: f' = 6 * slot(i) + j for the element (row i, column j) of the 6 x 6 transform, slot = position of i in
// (0, 5, 1, 2, 3, 4) -- the three row pairs that share their arithmetic are 12 consecutive frequencies = three 16-byte words
__host__ __device__ constexpr int w4_slot(int i) { return i == 0 ? 0 : i == 5 ? 1 : i + 1; }

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
  float* out = U + blk * (kFQ * 64 * 16) + nl * 16 + cl * 4;
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
      const int f = 6 * w4_slot(r) + j;
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

// LDS operations a transform wave issues in MFMA step fq (after the step's fragment loads): patch-row reads (3 ds_read2_b32
// each) and V writes (ds_write_b128).  Role 0 = rows (0, 5) of B^T d, role 1 = rows (1, 2) and (3, 4), role 2 = no transform.
__device__ constexpr int kStepOps[3][9] = {{3, 3, 3, 3, 3, 3, 1, 0, 0}, {3, 3, 3, 3, 0, 0, 0, 2, 1}, {0, 0, 0, 0, 0, 0, 0, 0, 0}};
__host__ __device__ constexpr int w4_wait(int role, int fq) {   // LDS operations issued after the fragment loads of step fq
  return (fq > 0 ? kStepOps[role][fq - 1] : 0) + (fq < 8 ? 2 : 0) + kStepOps[role][fq];
}

template <int TWB, int THB>
__global__ __launch_bounds__(512) void k_conv3x3_wino4_f32(const float* __restrict__ x, const float* __restrict__ U,
                                                           float* __restrict__ y, W4Geom gm, W4Ep ep) {
  constexpr int PW = 4 * TWB + 2, PH = 4 * THB + 2;
  // row pitch in 16-byte slots: rows of tiles are 4 patch rows apart; 4 * RP * 16 bytes should step the 256-byte bank period by
  // the width of a tile row's 16-byte slots (TWB * 16 bytes), so that the 16 tiles of a wave read 16 different slots
  constexpr int RP = (TWB == 4) ? PW + 1 : PW;
  constexpr int PP = PH * RP, kPInstr = (PP + 63) / 64;
  static_assert(TWB * THB <= kT && PP * 16 <= kPBytes && kPInstr <= 16, "tile block");
  constexpr int O0 = 0, O1 = TWB + 1, O2 = 2 * TWB + 2, O3 = 3 * TWB + 2;    // column planes x % 4 = 0 / 1 / 2 / 3 of a patch row
  static_assert((O3 + 0) * 4 <= 255 && (O1 + 1) * 4 <= 255, "ds_read2 offsets");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = (unsigned)(size_t)smem;      // LDS byte address of the dynamic segment (0 in practice)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wq = wave & 3, wh = wave >> 2;
  const int nblocks = gm.B * gm.bx * gm.by;
  const int nchunks = gm.Cin / kKC;
  const float* zero = reinterpret_cast<const float*>(g_zero16w4);

  int unit, cbi, blk_lin, b, ty0, tx0, n0;
  const float* pp[2];
  int pinc[2];
  const float* up;
  auto setup = [&](int item) {
    unit = item;
    cbi = unit / nblocks;
    const int ru = unit - cbi * nblocks;
    int blk;
    {   // consecutive items go round the 8 XCDs: XCD k walks its own contiguous range of tile blocks (shared halos stay in its L2)
      const int base = nblocks >> 3, extra = nblocks & 7, k = ru & 7;
      blk = k * base + min(k, extra) + (ru >> 3);
    }
    blk_lin = blk;
    b = blk / (gm.bx * gm.by);
    blk -= b * gm.bx * gm.by;
    ty0 = (blk / gm.bx) * THB, tx0 = (blk % gm.bx) * TWB;      // in tiles
    n0 = cbi * kTN;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // patch DMA instruction pi fills slots [64 pi, 64 pi + 64): slot -> (row, plane position) -> pixel
      const int pi = ((wave + 4) & 7) + 8 * i;
      const int qq = pi * 64 + lane;
      const int qy = qq / RP, qs = qq - qy * RP;
      const int qx = qs < O1 ? 4 * qs : qs < O2 ? 4 * (qs - O1) + 1 : qs < O3 ? 4 * (qs - O2) + 2 : 4 * (qs - O3) + 3;
      const int gy = 4 * ty0 + qy - 1, gx = 4 * tx0 + qx - 1;
      const bool ok = qq < PP && qs < PW && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
      pp[i] = ok ? x + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cin : zero;
      pinc[i] = ok ? kKC : 0;
    }
    up = U + (size_t)cbi * nchunks * (kUBytes / 4) + wave * 256 + lane * 4;
  };
  auto stage_p = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pi = ((wave + 4) & 7) + 8 * i;
      if (pi < kPInstr) dma16(pp[i], sbase + kP0 + buf * kPBytes + pi * 1024);
      pp[i] += pinc[i];
    }
  };
  auto stage_u = [&](int buf) {
    const unsigned ub = sbase + (buf ? kU1 : kU0) + wave * 1024;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (wave + 8 * i < 36) dma16(up + i * 2048, ub + i * 8192);
    up += kUBytes / 4;
  };

  // ---- input transform: waves 0-5, thread = (row-pair role, tile, channel)
  const int role = wave >> 1;                          // 0: rows (0, 5); 1: rows (1, 2); 2: rows (3, 4); 3: none
  const int ttile = (wave & 1) * 16 + (lane >> 2), tc = lane & 3;
  const int ttc = min(ttile, TWB * THB - 1);
  const int tyl = ttc / TWB, txl = ttc - tyl * TWB;
  // byte address of patch row k of this thread's tile, buffer 0: tprow + k * RP * 16
  const unsigned tprow = sbase + kP0 + ((4 * tyl) * RP + txl) * 16 + tc * 4;
  // V word of this thread: quads 3 role .. 3 role + 2, [fq][tile][c][4]
  const unsigned tvw = sbase + kV0 + ((3 * min(role, 2)) * kT * kKC + ttile * kKC + tc) * 16;
  const float kap = role == 1 ? 4.f : 1.f, lam = role == 1 ? 1.f : 2.f;

#define W4_ROWREAD(D, K, POFF)                                                                                          \
  do {                                                                                                                  \
    const unsigned a_ = tprow + (K) * (RP * 16) + (POFF);                                                               \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[0]) : "v"(a_), "n"(O0 * 4), "n"((O0 + 1) * 4));  /* j = 0, 4 */ \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[1]) : "v"(a_), "n"(O1 * 4), "n"((O1 + 1) * 4));  /* j = 1, 5 */ \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[2]) : "v"(a_), "n"(O2 * 4), "n"(O3 * 4));        /* j = 2, 3 */ \
  } while (0)
  // columns j = 0..5 of a row held as read above
#define W4_COL(D, J) ((J) == 0 ? D[0][0] : (J) == 4 ? D[0][1] : (J) == 1 ? D[1][0] : (J) == 5 ? D[1][1] : (J) == 2 ? D[2][0] : D[2][1])

  // 1-D transform along a row: v = w B (B^T of F(4, 3): [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1])
  auto row_pass = [](const float (&w)[6], float (&v)[6]) {
    v[0] = fmaf(4.f, w[0], fmaf(-5.f, w[2], w[4]));
    v[5] = fmaf(4.f, w[1], fmaf(-5.f, w[3], w[5]));
    const float e1 = fmaf(-4.f, w[2], w[4]), o1 = fmaf(-4.f, w[1], w[3]);
    v[1] = e1 + o1;
    v[2] = e1 - o1;
    const float e2 = w[4] - w[2], o2 = w[3] - w[1];
    v[3] = fmaf(2.f, o2, e2);
    v[4] = fmaf(-2.f, o2, e2);
  };

  // plain (not interleaved) transform of the unit's first stage: patch buffer 0 -> V buffer 0
  auto transform_first = [&]() {
    if (role > 2) return;
    const float* P = reinterpret_cast<const float*>(smem + (tprow - sbase));
    auto px = [&](int k, int j) {
      const int o = (j & 3) == 0 ? O0 : (j & 3) == 1 ? O1 : (j & 3) == 2 ? O2 : O3;
      return P[(k * RP + o + (j >> 2)) * 4];
    };
    float wa[6], wb[6], va[6], vb[6];
    if (role == 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        wa[j] = fmaf(4.f, px(0, j), fmaf(-5.f, px(2, j), px(4, j)));
        wb[j] = fmaf(4.f, px(1, j), fmaf(-5.f, px(3, j), px(5, j)));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float e = fmaf(-kap, px(2, j), px(4, j)), o = lam * fmaf(-kap, px(1, j), px(3, j));
        wa[j] = e + o;
        wb[j] = e - o;
      }
    }
    row_pass(wa, va);
    row_pass(wb, vb);
    f32x4* V = reinterpret_cast<f32x4*>(smem + (tvw - sbase));
    V[0] = (f32x4){va[0], va[1], va[2], va[3]};
    V[kT * kKC] = (f32x4){va[4], va[5], vb[0], vb[1]};
    V[2 * kT * kKC] = (f32x4){vb[2], vb[3], vb[4], vb[5]};
  };

  f32x4 acc[36];
  const unsigned fa = sbase + kV0 + ((16 * wh + li) * kKC + g) * 16, fb0 = sbase + ((16 * wq + li) * kKC + g) * 16;

  auto first_stages = [&]() {      // the DMA an item needs before its first stage (pointers advance as in the stage loop)
    stage_p(0);
    stage_u(0);
    if (nchunks > 1) stage_p(1);
  };

  // One stage = 9 steps of (2 fragment loads one step ahead, 4 MFMAs); the transform of the NEXT stage's patch rides along in
  // waves 0-5: patch-row reads in the early steps, arithmetic two steps after the rows were requested (LDS returns in order, so
  // the step's s_waitcnt counts exactly the operations issued after the fragments it needs), V writes at the end.
  auto stage = [&](auto more_c, auto role_c, int chunk) {
    constexpr bool MORE = decltype(more_c)::value;
    constexpr int ROLE = MORE ? decltype(role_c)::value : 2;       // role class: 0, 1 (rows (1,2) / (3,4)), 2 (no transform)
    const int cb = chunk & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();      // V(chunk) written, U(chunk) and patch(chunk + 1) landed
    const unsigned pa = fa + cb * kVBytes, pb = fb0 + (cb ? kU1 : kU0);
    const unsigned poff = (cb ^ 1) * kPBytes;                      // the patch of chunk + 1
    const unsigned vw = tvw + (cb ^ 1) * kVBytes;
    f32x4 qa[2], qb[2];
    f32x2 d0[3], d1[3], d2[3], d3[3], d4[3], d5[3];
    float wa[6], wb[6], va[6], vb[6], ee[6];
    asm volatile("ds_read_b128 %0, %1" : "=v"(qa[0]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1" : "=v"(qb[0]) : "v"(pb));
    if (MORE) stage_u(cb ^ 1);
    if (chunk + 2 < nchunks) stage_p(cb);
    static_for<9>([&](auto fq_c) {
      constexpr int fq = decltype(fq_c)::value;
      constexpr int cur = fq & 1, nxt = cur ^ 1;
      if constexpr (fq < 8) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qa[nxt]) : "v"(pa), "n"((fq + 1) * (kT * kKC * 16)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qb[nxt]) : "v"(pb), "n"((fq + 1) * (kTN * kKC * 16)));
      }
      // ---- LDS operations of the transform in this step
      if constexpr (ROLE == 0) {
        if constexpr (fq == 0) W4_ROWREAD(d0, 0, poff);
        if constexpr (fq == 1) W4_ROWREAD(d2, 2, poff);
        if constexpr (fq == 2) W4_ROWREAD(d4, 4, poff);
        if constexpr (fq == 3) W4_ROWREAD(d1, 1, poff);
        if constexpr (fq == 4) W4_ROWREAD(d3, 3, poff);
        if constexpr (fq == 5) W4_ROWREAD(d5, 5, poff);
        if constexpr (fq == 6)
          asm volatile("ds_write_b128 %0, %1" ::"v"(vw), "v"((f32x4){va[0], va[1], va[2], va[3]}) : "memory");
      } else if constexpr (ROLE == 1) {
        if constexpr (fq == 0) W4_ROWREAD(d2, 2, poff);
        if constexpr (fq == 1) W4_ROWREAD(d4, 4, poff);
        if constexpr (fq == 2) W4_ROWREAD(d1, 1, poff);
        if constexpr (fq == 3) W4_ROWREAD(d3, 3, poff);
        if constexpr (fq == 7) {
          asm volatile("ds_write_b128 %0, %1" ::"v"(vw), "v"((f32x4){va[0], va[1], va[2], va[3]}) : "memory");
          asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(vw), "v"((f32x4){va[4], va[5], vb[0], vb[1]}), "n"(kT * kKC * 16) : "memory");
        }
        if constexpr (fq == 8)
          asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(vw), "v"((f32x4){vb[2], vb[3], vb[4], vb[5]}), "n"(2 * kT * kKC * 16) : "memory");
      }
      // ---- wait for this step's fragments (and with them everything requested two steps ago)
      constexpr int WN = w4_wait(ROLE, fq);
      if constexpr (ROLE == 0 && fq == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(d0[0]), "+v"(d0[1]), "+v"(d0[2]), "+v"(d2[0]), "+v"(d2[1]), "+v"(d2[2]) : "n"(WN));
      else if constexpr (ROLE == 0 && fq == 5)      // row 4 (requested in step 2) arrived with step 4's fragments; bind it here
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(d4[0]), "+v"(d4[1]), "+v"(d4[2]) : "n"(WN));
      else if constexpr (ROLE == 0 && fq == 7)
        asm volatile("s_waitcnt lgkmcnt(%11)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(d1[0]), "+v"(d1[1]), "+v"(d1[2]), "+v"(d3[0]), "+v"(d3[1]), "+v"(d3[2]), "+v"(d5[0]), "+v"(d5[1]), "+v"(d5[2]) : "n"(WN));
      else if constexpr (ROLE == 1 && fq == 3)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(d2[0]), "+v"(d2[1]), "+v"(d2[2]), "+v"(d4[0]), "+v"(d4[1]), "+v"(d4[2]) : "n"(WN));
      else if constexpr (ROLE == 1 && fq == 5)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(d1[0]), "+v"(d1[1]), "+v"(d1[2]), "+v"(d3[0]), "+v"(d3[1]), "+v"(d3[2]) : "n"(WN));
      else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(qa[cur]), "+v"(qb[cur]) : "n"(WN));
      // ---- the step's four MFMAs
#pragma unroll
      for (int p = 0; p < 4; ++p)
        acc[4 * fq + p] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[cur][p], qb[cur][p], acc[4 * fq + p], 0, 0, 0);
      // ---- transform arithmetic that became possible with this step's wait
      if constexpr (ROLE == 0) {
        if constexpr (fq == 5) {     // rows 0, 2, 4 are in: column pass of row 0 of B^T d
#pragma unroll
          for (int j = 0; j < 6; ++j) wa[j] = fmaf(4.f, W4_COL(d0, j), fmaf(-5.f, W4_COL(d2, j), W4_COL(d4, j)));
          row_pass(wa, va);
        }
        if constexpr (fq == 7) {     // rows 1, 3, 5: row 5 of B^T d
#pragma unroll
          for (int j = 0; j < 6; ++j) wb[j] = fmaf(4.f, W4_COL(d1, j), fmaf(-5.f, W4_COL(d3, j), W4_COL(d5, j)));
        }
        if constexpr (fq == 8) row_pass(wb, vb);
      } else if constexpr (ROLE == 1) {
        if constexpr (fq == 3) {
#pragma unroll
          for (int j = 0; j < 6; ++j) ee[j] = fmaf(-kap, W4_COL(d2, j), W4_COL(d4, j));
        }
        if constexpr (fq == 5) {
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const float o = lam * fmaf(-kap, W4_COL(d1, j), W4_COL(d3, j));
            wa[j] = ee[j] + o;
            wb[j] = ee[j] - o;
          }
          row_pass(wa, va);
        }
        if constexpr (fq == 6) row_pass(wb, vb);
      }
    });
    if constexpr (ROLE == 0) {
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(vw), "v"((f32x4){va[4], va[5], vb[0], vb[1]}), "n"(kT * kKC * 16) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(vw), "v"((f32x4){vb[2], vb[3], vb[4], vb[5]}), "n"(2 * kT * kKC * 16) : "memory");
    }
  };

  int item = blockIdx.x;
  setup(item);
  first_stages();
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 1
  for (;;) {
#pragma unroll
    for (int f = 0; f < 36; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA (and the previous item's stores) are done
    __syncthreads();                                       // first stages in LDS; everybody has left the previous item's epilogue
    transform_first();
#pragma unroll 1
    for (int chunk = 0; chunk + 1 < nchunks; ++chunk) {
      if (role == 0) stage(std::true_type{}, IC<0>{}, chunk);
      else if (role < 3) stage(std::true_type{}, IC<1>{}, chunk);
      else stage(std::true_type{}, IC<2>{}, chunk);
    }
    stage(std::false_type{}, IC<2>{}, nchunks - 1);
    __syncthreads();                  // everybody is done with U / V / the patches of this item
    const int e_b = b, e_ty0 = ty0, e_tx0 = tx0, e_n0 = n0, e_blk = blk_lin;
    const int next = item + gridDim.x;
    // ---- epilogue: output transform in registers -> fp32 pixel rows in LDS (two passes of 16 tiles: accumulator rows r = 2 pass,
    // 2 pass + 1 of every wave), then coalesced 16-byte stores with the fused tail.  Row = tl * 16 + py * 4 + px with
    // tl = wh * 8 + g * 2 + (r & 1); the 16-channel group is XOR-ed with g (the four lane groups write rows 8 KB apart otherwise).
    const int c4 = (tid & 15) * 4, n = e_n0 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), scv = make_float4(1.f, 1.f, 1.f, 1.f), shv = bv;
    if (ep.bias && n < gm.Cout) bv = *reinterpret_cast<const float4*>(ep.bias + n);
    if (ep.scale && n < gm.Cout) {
      scv = *reinterpret_cast<const float4*>(ep.scale + n);
      shv = *reinterpret_cast<const float4*>(ep.shift + n);
    }
    s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      float* Os = reinterpret_cast<float*>(smem + (pass ? kStageB : kStageA));
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * pass + rr;
        float t[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {       // A^T M: rows of M in frequency slots (0, 5, 1, 2, 3, 4) -> m0 = slot 0, m5 = slot 1, m1..m4 = slots 2..5
          const float m0 = acc[0 + j][r], m5 = acc[6 + j][r], m1 = acc[12 + j][r], m2 = acc[18 + j][r], m3 = acc[24 + j][r],
                      m4 = acc[30 + j][r];
          const float sa = m1 + m2, da = m1 - m2, sb = m3 + m4, db = m3 - m4;
          t[0][j] = m0 + sa + sb;
          t[1][j] = fmaf(2.f, db, da);
          t[2][j] = fmaf(4.f, sb, sa);
          t[3][j] = fmaf(8.f, db, da) + m5;
        }
        const int tl = wh * 8 + g * 2 + rr;
        float* o = Os + (tl * 16) * 64 + (16 * (wq ^ g) + li);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float sa = t[a][1] + t[a][2], da = t[a][1] - t[a][2], sb = t[a][3] + t[a][4], db = t[a][3] - t[a][4];
          o[(a * 4 + 0) * 64] = t[a][0] + sa + sb;
          o[(a * 4 + 1) * 64] = fmaf(2.f, db, da);
          o[(a * 4 + 2) * 64] = fmaf(4.f, sb, sa);
          o[(a * 4 + 3) * 64] = fmaf(8.f, db, da) + t[a][5];
        }
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int row = (tid >> 4) + 32 * k;
        const int tl = row >> 4, p = row & 15;
        const int slot = 16 * (tl >> 3) + 4 * ((tl >> 1) & 3) + 2 * pass + (tl & 1);
        const int sy = slot / TWB, sx = slot - sy * TWB;
        const int gy = 4 * (e_ty0 + sy) + (p >> 2), gx = 4 * (e_tx0 + sx) + (p & 3);
        if (slot >= TWB * THB || gy >= gm.H || gx >= gm.W || n >= gm.Cout) continue;
        float4 v = *reinterpret_cast<const float4*>(Os + row * 64 + (c4 ^ (16 * ((tl >> 1) & 3))));
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (ep.scale) {
          v.x = v.x * scv.x + shv.x; v.y = v.y * scv.y + shv.y; v.z = v.z * scv.z + shv.z; v.w = v.w * scv.w + shv.w;
        }
        const size_t off = ((size_t)(e_b * gm.H + gy) * gm.W + gx) * gm.Cout + n;
        if (ep.residual) {
          const float4 h = *reinterpret_cast<const float4*>(ep.residual + off);
          v.x += h.x; v.y += h.y; v.z += h.z; v.w += h.w;
        }
        if (ep.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(y + off) = v;
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
      }
      if (pass == 0) {
        __syncthreads();               // staging A has been read: the next item's first stages may land there
        if (next < gm.n_items) {
          setup(next);
          first_stages();
        }
      }
    }
    if (ep.stats) {     // a thread keeps ONE 4-channel piece over its 16 rows: reduce the 32 row groups through LDS, fixed order
      __syncthreads();
      float* Os = reinterpret_cast<float*>(smem + kStageB);
      const int grp = tid >> 4;
      const float a4[4] = {s1.x, s1.y, s1.z, s1.w}, q4[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Os[(grp * 64 + c4 + e) * 2] = a4[e];
        Os[(grp * 64 + c4 + e) * 2 + 1] = q4[e];
      }
      __syncthreads();
      if (tid < 64 && e_n0 + tid < gm.Cout) {
        double ad = 0.0, qd = 0.0;          // the 32 row-group sums combine in double (sum of squares minus mean^2 comes next)
        for (int k = 0; k < 32; ++k) {
          ad += (double)Os[(k * 64 + tid) * 2];
          qd += (double)Os[(k * 64 + tid) * 2 + 1];
        }
        ep.stats[((size_t)e_blk * gm.Cout + e_n0 + tid) * 2] = (float)ad;
        ep.stats[((size_t)e_blk * gm.Cout + e_n0 + tid) * 2 + 1] = (float)qd;
      }
    }
    if (next >= gm.n_items) break;
    item = next;
  }
}

#undef W4_ROWREAD
#undef W4_COL

struct W4Plan {
  int twb, thb, bx, by;
};
// tile-block shape (<= 32 tiles of 4 x 4 outputs) with the fewest blocks for this map
W4Plan w4_plan(int H, int W) {
  static const int shapes[][2] = {{8, 4}, {4, 8}, {16, 2}, {11, 2}};
  static const int force = getenv("UD_WINO4_SHAPE") ? atoi(getenv("UD_WINO4_SHAPE")) : -1;
  const int TX = (W + 3) / 4, TY = (H + 3) / 4;
  W4Plan best{};
  long long cost = -1;
  for (int i = 0; i < 4; ++i) {
    if (force >= 0 && i != force) continue;
    const int bx = ud_div_up(TX, shapes[i][0]), by = ud_div_up(TY, shapes[i][1]);
    const long long cst = (long long)bx * by;
    if (cost < 0 || cst < cost) cost = cst, best = W4Plan{shapes[i][0], shapes[i][1], bx, by};
  }
  return best;
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
  return (size_t)B * p.bx * p.by * Cout * 2 * sizeof(float);
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

// y = conv3x3(x) (+ bias) (* scale + shift: a folded eval-mode BatchNorm) (+ residual) (ReLU if flags & 1) with U from
// ud_conv3x3_wino4_f32_weights(N = Cout, C = Cin); partial != nullptr: also the per-workgroup BatchNorm partial sums
// ([*slices][Cout][2], same contract as ud_conv3x3_bnstats_nhwc_f32).
extern "C" int ud_conv3x3_wino4_nhwc_f32(const float* x, const float* U, float* y, int B, int H, int W, int Cin, int Cout,
                                         const float* bias, const float* scale, const float* shift, const float* residual,
                                         int flags, float* partial, size_t partial_bytes, int* slices, ud_stream_t stream_) {
  if (!x || !U || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 4 != 0) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const W4Plan p = w4_plan(H, W);
  const int nblocks = B * p.bx * p.by;
  const long long units = (long long)nblocks * ud_div_up(Cout, kTN);
  if (units > 0x7fffffffll) return UD_ERR_UNSUPPORTED;
  W4Geom gm{B, H, W, Cin, Cout, p.bx, p.by, (int)units};
  W4Ep ep{bias, scale, shift, residual, flags & 1, partial};
  if (partial) {
    if (!slices || partial_bytes < (size_t)nblocks * Cout * 2 * sizeof(float)) return UD_ERR_WORKSPACE;
    *slices = nblocks;
  }
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
#define UD_W4_ATTR(A, Bq) \
  UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wino4_f32<A, Bq>, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem))
    UD_W4_ATTR(8, 4); UD_W4_ATTR(4, 8); UD_W4_ATTR(16, 2); UD_W4_ATTR(11, 2);
#undef UD_W4_ATTR
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("conv2d.k_conv3x3_wino4_f32", stream);
  static const int persist = getenv("UD_WINO4_GRID") ? atoi(getenv("UD_WINO4_GRID")) : 256;      // one workgroup per CU
  const dim3 grid((unsigned)std::min<long long>(units, persist > 0 ? persist : units));
#define UD_W4_LAUNCH(A, Bq) k_conv3x3_wino4_f32<A, Bq><<<grid, 512, kSmem, stream>>>(x, U, y, gm, ep)
  if (p.twb == 8) UD_W4_LAUNCH(8, 4);
  else if (p.twb == 4) UD_W4_LAUNCH(4, 8);
  else if (p.twb == 16) UD_W4_LAUNCH(16, 2);
  else UD_W4_LAUNCH(11, 2);
#undef UD_W4_LAUNCH
  UD_LAUNCH_CHECK();
  return UD_OK;
}
