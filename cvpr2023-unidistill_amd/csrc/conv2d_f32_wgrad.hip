// fp32 weight gradients of the dense convolutions (the reference's arithmetic: it trains in fp32, base_cli.py:40-45) --
// BEV trunk (base_bev_backbone.py:38-115), CenterHead shared / first SepHead convs (center_head.py:58-99,311-355), the
// fusion conv and the ResNet / neck convolutions (lss_fpn.py:143-149) -- on v_mfma_f32_16x16x4_f32 (exact fp32
// products, fp32 accumulation).  Deterministic: pixel slices keep private partial sums that k_wgrad_sum adds in a fixed
// order (the library's split-K kernels use atomics).
//
//   3x3 / stride 1 / pad 1:   dW[n][tap][c] = sum_p dy[p][n] * x[p + tap][c]
//   1x1 over pixel maps:      dW[n][k]      = sum_p dy'[p][n] * x'[p][k]     (plain, strided, transposed, patch, im2col)
//
// The reduction index of the MFMA is the PIXEL: A[i][k] = dy[pixel k][channel i], B[k][j] = x[pixel k][channel j], one
// fp32 register per operand, so channels-last rows are MFMA fragments as they lie in LDS -- no transposition:
//   A: ds_read_b32   lane (li, g) reads dy[p0 + g][n0 + 16 wave + li]
//   B: ds_read_b128  lane (li, g) reads x[p0 + g (+ tap shift)][4 li .. 4 li + 3]; element e feeds MFMA e, whose column
//      j = li then stands for channel 4 li + e (a permutation of the 64 channels that only the epilogue has to know).
// 3x3: a workgroup owns a 64 (dy channels) x 64 (x channels) tile of dW for ALL nine taps and walks a slice of the
// 8 x 16 pixel tiles; the x halo (10 x 18 pixels) and the dy tile are staged once per pixel tile by LDS-DMA and serve
// the nine shifted GEMMs: per 4-pixel step a wave issues 1 + 9 LDS reads for 36 MFMAs (1 152 MFMA cycles) -- the kernel
// is MFMA-bound by construction.  LDS is double-buffered (2 x 77 KB), one workgroup per CU.
#include "ud_common.h"
#include <cstdlib>
#include <utility>
#include "ud_prof.h"
#include "conv_pixmap.h"
#include "wgrad_sum.h"

namespace {

constexpr int kRowB = 256;                                                       // 64 fp32 channels per LDS row
// pixel tile TW x TH (16 x 8 or 20 x 6: the second one tiles the 180 x 180 BEV maps without a wasted pixel)
template <int TW, int TH>
struct WgTile {
  static constexpr int kTM = TW * TH, kHW = TW + 2, kHQ = kHW * (TH + 2), kHQP = (kHQ + 3) / 4 * 4;   // halo staged 4 rows per DMA
  static constexpr int kXBytes = kHQP * kRowB, kDBytes = kTM * kRowB;
  static constexpr int kXPieces = kHQP / 4, kDPieces = kTM / 4;                // 1-KiB DMA pieces
  static constexpr int kBufBytes = kXBytes + kDBytes;                          // 78 848 / 75 776 per buffer, two buffers
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __attribute__((aligned(16))) unsigned int g_zero16w[4];
constexpr unsigned kOobW = 0xFFF00000u;     // a buffer offset no tensor reaches (the launchers check): reads zeros

template <class F, int... Is>
__device__ __forceinline__ void wg_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wg_static_for(F&& f) {
  wg_static_for_impl(std::forward<F>(f), std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void dma16(const float* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// (a __device__ wrapper: the builtin called directly in a kernel template makes hipcc 7.2 drop the kernel's host stub)
__device__ __forceinline__ void dma16b(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(size_t)lds_wave_base, 16, voff, 0, 0, 0);
}

struct WgGeom {
  int B, H, W, Cin, Cout, tiles_x, tiles_y;
};

template <int kTW, int kTH>
__global__ __launch_bounds__(256) void k_conv3x3_wgrad_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ partial, WgGeom gm, int c_tiles,
                                                           int tiles_per_slice) {
  using T = WgTile<kTW, kTH>;
  constexpr int kHW = T::kHW, kHQ = T::kHQ, kXBytes = T::kXBytes, kXPieces = T::kXPieces, kDPieces = T::kDPieces,
                kBufBytes = T::kBufBytes;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int ct = blockIdx.y % c_tiles, nt = blockIdx.y / c_tiles;
  const int n0 = nt * 64, c0 = ct * 64;
  const int per_img = gm.tiles_x * gm.tiles_y, ntiles = gm.B * per_img;
  const int t_begin = blockIdx.x * tiles_per_slice, t_end = min(ntiles, t_begin + tiles_per_slice);
  const float* zero = reinterpret_cast<const float*>(g_zero16w);

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[t][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // DMA: a piece = 4 LDS rows x 256 B; lane -> (row g, 16-byte slot li).  x rows are stored as they are; dy rows have
  // their slots XOR-ed with 4 on odd pixels (on the SOURCE address: the DMA's LDS pattern is fixed), which puts the
  // two pixel rows a half-wave's ds_read_b32 touches on different bank halves.
  const bool xc_ok = c0 + 4 * li < gm.Cin;
  const int dslot = li ^ ((g & 1) << 2);
  const bool dn_ok = n0 + 4 * dslot < gm.Cout;
  auto stage = [&](int tile, int buf) {
    char* sb = smem + buf * kBufBytes;
    const int b = tile / per_img;
    const int rem = tile - b * per_img;
    const int ty0 = (rem / gm.tiles_x) * kTH, tx0 = (rem % gm.tiles_x) * kTW;
    for (int piece = wave; piece < kXPieces; piece += 4) {
      const int q = piece * 4 + g, qy = q / kHW, qx = q - qy * kHW;
      const int gy = ty0 + qy - 1, gx = tx0 + qx - 1;
      const bool ok = xc_ok && q < kHQ && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
      dma16(ok ? x + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cin + c0 + 4 * li : zero, sb + piece * 1024);
    }
    for (int piece = wave; piece < kDPieces; piece += 4) {
      const int p = piece * 4 + g, gy = ty0 + p / kTW, gx = tx0 + p % kTW;
      const bool ok = dn_ok && gy < gm.H && gx < gm.W;
      dma16(ok ? dy + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cout + n0 + 4 * dslot : zero,
            sb + kXBytes + piece * 1024);
    }
  };
  // Fragment addresses of step (row 0, columns 0..3); a step adds an immediate.  The fragment reads are inline asm with
  // explicit lgkmcnt waits: left to itself the compiler re-used ONE register quad for all nine B fragments of a step
  // (ds_read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, nine times: an exposed LDS round trip per 128 MFMA cycles, 99-105 TFLOP/s);
  // here the ten reads of step s + 1 are in flight while the 36 MFMAs of step s issue.
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int aslot = 4 * wave + (li >> 2);
  unsigned pa = lds0 + kXBytes + g * kRowB + ((aslot ^ ((g & 1) << 2)) << 4) + (li & 3) * 4;
  unsigned pb = lds0 + g * kRowB + li * 16;
  float fa[2];
  f32x4 fb[2][9];
  constexpr int kQ = kTW / 4, kSteps = kTH * kQ;      // 4-pixel steps per tile row / per tile
#define UD_WG_LOADS(BUF, R, Q)                                                                                          \
  do {                                                                                                                  \
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[BUF]) : "v"(pa), "n"(((R) * kTW + 4 * (Q)) * kRowB) : "memory"); \
    _Pragma("unroll") for (int tap = 0; tap < 9; ++tap)                                                                 \
      asm volatile("ds_read_b128 %0, %1 offset:%2"                                                                      \
                   : "=v"(fb[BUF][tap])                                                                                 \
                   : "v"(pb), "n"((((R) + tap / 3) * kHW + 4 * (Q) + tap % 3) * kRowB)                                 \
                   : "memory");                                                                                         \
  } while (0)
#define UD_WG_WAIT(BUF, N)                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                              \
               : "+v"(fa[BUF]), "+v"(fb[BUF][0]), "+v"(fb[BUF][1]), "+v"(fb[BUF][2]), "+v"(fb[BUF][3]), "+v"(fb[BUF][4]), \
                 "+v"(fb[BUF][5]), "+v"(fb[BUF][6]), "+v"(fb[BUF][7]), "+v"(fb[BUF][8]))

  // two LDS buffers, ONE workgroup per CU (one wave per SIMD: the 36 MFMAs of a step are independent, and the fragment
  // reads run a step ahead): tile t + 1 lands while tile t is multiplied, one barrier per tile.  (First version: one
  // buffer and two workgroups per CU to overlap each other's DMA -- but identical workgroups that start together stay in
  // phase, both wait for their DMA at the same time: 99-109 TFLOP/s.)
  if (t_begin < t_end) stage(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    if (tile + 1 < t_end) stage(tile + 1, buf ^ 1);
    UD_WG_LOADS(0, 0, 0);
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const int cur = st & 1;
      if (st + 1 < kSteps) {
        if (cur == 0) UD_WG_LOADS(1, (st + 1) / kQ, (st + 1) % kQ); else UD_WG_LOADS(0, (st + 1) / kQ, (st + 1) % kQ);
        if (cur == 0) UD_WG_WAIT(0, 10); else UD_WG_WAIT(1, 10);
      } else {
        if (cur == 0) UD_WG_WAIT(0, 0); else UD_WG_WAIT(1, 0);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[tap][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur], fb[cur][tap][e], acc[tap][e], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // the next tile has landed; everybody is done reading this one
    const unsigned d = buf ? (unsigned)-kBufBytes : (unsigned)kBufBytes;
    pa += d;
    pb += d;
    buf ^= 1;
  }
#undef UD_WG_LOADS
#undef UD_WG_WAIT
  // partial[slice][n][tap][c]; D layout: lane holds rows n = 4 g + r, column li = channels 4 li + e of MFMA e
  const int c = c0 + 4 * li;
  if (c < gm.Cin) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * wave + 4 * g + r;
        if (n < gm.Cout)
          *reinterpret_cast<float4*>(partial + (((size_t)blockIdx.x * gm.Cout + n) * 9 + tap) * gm.Cin + c) =
              make_float4(acc[tap][0][r], acc[tap][1][r], acc[tap][2][r], acc[tap][3][r]);
      }
  }
}

// tile shape with the smaller padded area (ties: 16 x 8)
bool wg3_wide(int H, int W) {
  return (long long)ud_div_up(W, 20) * 20 * ud_div_up(H, 6) * 6 < (long long)ud_div_up(W, 16) * 16 * ud_div_up(H, 8) * 8;
}
// pixel-tile slices: (slices x 64x64 output tiles) <= 256 workgroups = one per CU, all resident
int wg3_slices(int B, int H, int W, int Cin, int Cout, int* tiles_per_slice) {
  const bool wide = wg3_wide(H, W);
  const int ntiles = B * ud_div_up(W, wide ? 20 : 16) * ud_div_up(H, wide ? 6 : 8);
  const int combos = ud_div_up(Cout, 64) * ud_div_up(Cin, 64);
  int s = 256 / combos;
  if (s > ntiles) s = ntiles;
  if (s < 1) s = 1;
  const int per = (ntiles + s - 1) / s;
  *tiles_per_slice = per;
  return (ntiles + per - 1) / per;
}

// ---- 1x1 over pixel maps ---------------------------------------------------------------------------------------------
// (pixel slice) x (NT x 64 output tile) workgroups, 32-pixel steps staged by LDS-DMA into two buffers (2 x 24 / 16 KB:
// three workgroups per CU), wave w owns NT / 4 dy channels x 64 x channels.  These layers have few channels (64 <-> 256
// ...), i.e. 21-32 flops per byte streamed: the kernel runs at the HBM / L2 rate as much as at the MFMA rate.
constexpr int kPS = 32;                          // pixels per step
// PLAIN (both operands plain [P][C] rows, tensors < 4 GB -- the ResNet / FPN 1x1 layers, most launches of a step): staged through
// buffer descriptors whose base moves with the pixel step, so a lane's offsets are fixed for the whole slice (no address
// arithmetic per piece: that was ~90 VALU instructions per 64 MFMAs, and on this chip every issued instruction is paid for in
// MFMA issue time) and rows past the last pixel read zeros through the range check.
template <int NT, int PLAIN>     // bit 0: x is plain, bit 1: dy is plain (the other operand goes through its pixel map per piece)
__global__ __launch_bounds__(256) void k_conv1x1_wgrad_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ partial, long long P, int Cin, int Cout,
                                                           int c_tiles, int steps_per_slice, PixMap xmap, PixMap ymap) {
  constexpr int TI = NT / 64;                    // 16-channel dy fragments per wave
  constexpr int kDRow = NT * 4;                  // bytes per dy row (512 / 256)
  constexpr int kDSlots = NT / 4;                // 16-byte slots per dy row
  constexpr int kDRowsPer = 1024 / kDRow;        // dy rows per DMA piece (2 / 4)
  constexpr int kDTile = kPS * kDRow, kBuf = kDTile + kPS * kRowB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int ct = blockIdx.y % c_tiles, nt = blockIdx.y / c_tiles;
  const int n0 = nt * NT, c0 = ct * 64;
  const int steps = (int)((P + kPS - 1) / kPS);
  const int s_begin = blockIdx.x * steps_per_slice, s_end = min(steps, s_begin + steps_per_slice);
  const float* zero = reinterpret_cast<const float*>(g_zero16w);

  f32x4 acc[TI][4];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // PLAIN: the lane's fixed byte offsets within a 32-pixel step (kOobW: a channel past the end -> zeros)
  constexpr int kDP = kPS / kDRowsPer / 4, kXP = kPS / 4 / 4;       // dy / x pieces per wave and step
  unsigned vd[kDP], vx[kXP];
  if (PLAIN) {
#pragma unroll
    for (int k = 0; k < kDP; ++k) {
      const int piece = wave + 4 * k, r = piece * kDRowsPer + lane / kDSlots, slot = lane % kDSlots;
      const int n = n0 + 4 * (slot ^ ((r & 1) << 2));
      vd[k] = n < Cout ? (unsigned)(r * Cout + n) * 4u : kOobW;
    }
#pragma unroll
    for (int k = 0; k < kXP; ++k) {
      const int r = (wave + 4 * k) * 4 + g, c = c0 + 4 * li;
      vx[k] = c < Cin ? (unsigned)(r * Cin + c) * 4u : kOobW;
    }
  }
#define UD_WG_STAGE_PLAIN(STEP, BUF)                                                                                          \
  do {                                                                                                                        \
    const long long p0_ = (long long)(STEP) * kPS;                                                                            \
    const unsigned lb = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + (BUF) * kBuf) + wave * 1024;        \
    const unsigned long long left = (unsigned long long)(P - p0_); /* > 0: the step exists */                                 \
    if (PLAIN & 2) {                                                                                                          \
      const __amdgpu_buffer_rsrc_t rd =                                                                                       \
          __builtin_amdgcn_make_buffer_rsrc((void*)(dy + p0_ * Cout), 0, (int)(unsigned)(left * Cout * 4), 0x00020000);        \
      _Pragma("unroll") for (int k = 0; k < kDP; ++k) dma16b(rd, vd[k], lb + k * 4096);                                       \
    }                                                                                                                         \
    if (PLAIN & 1) {                                                                                                          \
      const __amdgpu_buffer_rsrc_t rxs =                                                                                      \
          __builtin_amdgcn_make_buffer_rsrc((void*)(x + p0_ * Cin), 0, (int)(unsigned)(left * Cin * 4), 0x00020000);          \
      _Pragma("unroll") for (int k = 0; k < kXP; ++k) dma16b(rxs, vx[k], lb + kDTile + k * 4096);                             \
    }                                                                                                                         \
  } while (0)
  auto stage = [&](int step, int buf) {
    char* sb = smem + buf * kBuf;
    const long long p0 = (long long)step * kPS;
    if (!(PLAIN & 2))
    for (int piece = wave; piece < kPS / kDRowsPer; piece += 4) {      // dy: slots XOR 4 on odd pixels (see the 3x3 kernel)
      const int r = piece * kDRowsPer + lane / kDSlots, slot = lane % kDSlots;
      const int n = n0 + 4 * (slot ^ ((r & 1) << 2));
      const long long p = p0 + r;
      dma16((p < P && n < Cout) ? dy + ymap.off(p, n, Cout) : zero, sb + piece * 1024);
    }
    if (!(PLAIN & 1))
    for (int piece = wave; piece < kPS / 4; piece += 4) {
      const int r = piece * 4 + g;
      const long long p = p0 + r;
      const float* src = zero;
      if (p < P && c0 + 4 * li < Cin) {
        const size_t o = xmap.off(p, c0 + 4 * li, Cin);
        if (o != kNoPixel) src = x + o;
      }
      dma16(src, sb + kDTile + piece * 1024);
    }
  };
  // fragment reads as inline asm, one 4-pixel step ahead of the MFMAs (see the 3x3 kernel)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned pa[TI];
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) {
    const int slot = 4 * (TI * wave + ti) + (li >> 2);
    pa[ti] = lds0 + g * kDRow + ((slot ^ ((g & 1) << 2)) << 4) + (li & 3) * 4;
  }
  unsigned pb = lds0 + kDTile + g * kRowB + li * 16;
  float fa[2][TI];
  f32x4 fb[2];
  // (macros, not generic lambdas: with the step offset as an asm immediate hipcc 7.2 loses the host stub of a kernel whose body
  // holds a generic lambda around it)
#define UD_WG_LOADS(B2, KS)                                                                                          \
  do {                                                                                                               \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[B2]) : "v"(pb), "n"(4 * (KS) * kRowB) : "memory");       \
    _Pragma("unroll") for (int ti = 0; ti < TI; ++ti)                                                                \
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[B2][ti]) : "v"(pa[ti]), "n"(4 * (KS) * kDRow) : "memory"); \
  } while (0)
// The waits name NO registers: an in/out operand lets hipcc put a register copy of a fragment in front of the wait, i.e. while the
// asm-issued read is still in flight (seen in the <64, plain> variant: `v_mov v8, v17` one instruction before lgkmcnt(0) -- one
// wave in ~40 launches multiplied a stale fragment).  Instead a scheduling barrier right after the wait keeps the MFMAs below it.
#define UD_WG_WAIT(N)                                    \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#define UD_WG_STEP(KS)                                                                                               \
  do {                                                                                                               \
    constexpr int cur = (KS) & 1;                                                                                    \
    if constexpr ((KS) + 1 < kPS / 4) {                                                                              \
      UD_WG_LOADS(cur ^ 1, ((KS) + 1) % (kPS / 4));                                                                  \
      if (TI == 2) UD_WG_WAIT(3); else UD_WG_WAIT(2);                                                                \
    } else {                                                                                                         \
      UD_WG_WAIT(0);                                                                                                 \
    }                                                                                                                \
    _Pragma("unroll") for (int ti = 0; ti < TI; ++ti)                                                                \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                  \
        acc[ti][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][ti], fb[cur][e], acc[ti][e], 0, 0, 0);            \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
  static_assert(kPS / 4 == 8, "eight 4-pixel steps per staged step");

  if (s_begin < s_end) {
    if (PLAIN) UD_WG_STAGE_PLAIN(s_begin, 0);
    if (PLAIN != 3) stage(s_begin, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int step = s_begin; step < s_end; ++step) {
    if (step + 1 < s_end) {
      if (PLAIN) UD_WG_STAGE_PLAIN(step + 1, buf ^ 1);
      if (PLAIN != 3) stage(step + 1, buf ^ 1);
    }
    UD_WG_LOADS(0, 0);
    UD_WG_STEP(0); UD_WG_STEP(1); UD_WG_STEP(2); UD_WG_STEP(3); UD_WG_STEP(4); UD_WG_STEP(5); UD_WG_STEP(6); UD_WG_STEP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();            // the next step has landed; everybody is done reading this one
    const unsigned d = buf ? (unsigned)-kBuf : (unsigned)kBuf;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti) pa[ti] += d;
    pb += d;
    buf ^= 1;
  }
#undef UD_WG_STEP
#undef UD_WG_WAIT
#undef UD_WG_LOADS
#undef UD_WG_STAGE_PLAIN
  const int c = c0 + 4 * li;
  if (c < Cin) {
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 16 * (TI * wave + ti) + 4 * g + r;
        if (n < Cout)
          *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * Cout + n) * Cin + c) =
              make_float4(acc[ti][0][r], acc[ti][1][r], acc[ti][2][r], acc[ti][3][r]);
      }
  }
}

struct Wg1Plan {
  int nt, n_tiles, c_tiles, slices, steps_per_slice;
};
Wg1Plan wg1_plan(long long P, int Cin, int Cout) {
  Wg1Plan pl;
  pl.nt = Cout > 64 ? 128 : 64;
  pl.n_tiles = ud_div_up(Cout, pl.nt);
  pl.c_tiles = ud_div_up(Cin, 64);
  const int steps = (int)((P + kPS - 1) / kPS);
  int s = 768 / (pl.n_tiles * pl.c_tiles);           // three workgroups per CU, all resident
  if (s > steps) s = steps;
  if (s < 1) s = 1;
  pl.steps_per_slice = (steps + s - 1) / s;
  pl.slices = (steps + pl.steps_per_slice - 1) / pl.steps_per_slice;
  return pl;
}

}  // namespace

extern "C" size_t ud_conv3x3_wgrad_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  int per;
  return ud_align_up((size_t)wg3_slices(B, H, W, Cin, Cout, &per) * Cout * 9 * Cin * sizeof(float));
}

// x [B][H][W][Cin], dy [B][H][W][Cout] fp32 channels-last -> dw [Cout][3][3][Cin] fp32.  Cin % 4 == 0, Cout % 4 == 0.
extern "C" int ud_conv3x3_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin,
                                         int Cout, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!x || !dy || !dw || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Cin % 4 != 0 || Cout % 4 != 0) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_conv3x3_wgrad_f32_workspace_bytes(B, H, W, Cin, Cout)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = reinterpret_cast<float*>(workspace);
  const bool wide = wg3_wide(H, W);
  WgGeom gm{B, H, W, Cin, Cout, ud_div_up(W, wide ? 20 : 16), ud_div_up(H, wide ? 6 : 8)};
  int per;
  const int S = wg3_slices(B, H, W, Cin, Cout, &per);
  constexpr int lds16 = 2 * WgTile<16, 8>::kBufBytes, lds20 = 2 * WgTile<20, 6>::kBufBytes;
  static UdDeviceOnce set;
  if (const unsigned long long set_bit = set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_f32<16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds16));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wgrad_f32<20, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds20));
    set.mark(set_bit);
  }
  const int c_tiles = ud_div_up(Cin, 64);
  {
    UdProfScope prof("conv2d.k_wgrad_f32", stream);
    const dim3 grid(S, ud_div_up(Cout, 64) * c_tiles);
    if (wide) k_conv3x3_wgrad_f32<20, 6><<<grid, 256, lds20, stream>>>(x, dy, partial, gm, c_tiles, per);
    else k_conv3x3_wgrad_f32<16, 8><<<grid, 256, lds16, stream>>>(x, dy, partial, gm, c_tiles, per);
    UD_LAUNCH_CHECK();
  }
  return ud_wgrad_sum(partial, S, (size_t)Cout * 9 * Cin, dw, stream);
}

extern "C" size_t ud_conv1x1_wgrad_f32_workspace_bytes(int64_t P, int Cin, int Cout) {
  if (P <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const Wg1Plan pl = wg1_plan(P, Cin, Cout);
  return ud_align_up((size_t)pl.slices * Cout * Cin * sizeof(float));
}

// dW[n][k] = sum_p dy'[p][n] * x'[p][k], either operand optionally read through a PixMap (NULL = plain [P][C] rows):
// the weight gradients of every convolution ud_conv1x1_nhwc_f32 / ud_conv1x1_mapped_nhwc_f32 run.
extern "C" int ud_conv1x1_wgrad_mapped_nhwc_f32(const float* x, const float* dy, float* dw, int64_t P, int Cin, int Cout,
                                                const int* x_map, const int* dy_map, void* workspace,
                                                size_t workspace_bytes, ud_stream_t stream_) {
  if (!x || !dy || !dw || P <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  PixMap xm, ym;
  if (!map_from_ints(x_map, &xm, 4) || !map_from_ints(dy_map, &ym, 4)) return UD_ERR_INVALID_ARG;
  if (Cin % 4 != 0 || Cout % 4 != 0 || P > (int64_t)1 << 30) return UD_ERR_UNSUPPORTED;
  if (xm.mode == 1 && xm.s * xm.s * xm.C != Cin) return UD_ERR_INVALID_ARG;
  if (xm.mode == 2 && xm.C != Cin) return UD_ERR_INVALID_ARG;
  if (xm.mode == 3 && 9 * xm.C != Cin) return UD_ERR_INVALID_ARG;
  if (xm.mode == 4 || ym.mode >= 3) return UD_ERR_UNSUPPORTED;
  if (ym.mode == 1 && ym.s * ym.s * ym.C != Cout) return UD_ERR_INVALID_ARG;
  if (ym.mode == 2 && ym.C != Cout) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_conv1x1_wgrad_f32_workspace_bytes(P, Cin, Cout)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const Wg1Plan pl = wg1_plan(P, Cin, Cout);
  float* partial = reinterpret_cast<float*>(workspace);
  {
    UdProfScope prof("conv2d.k_wgrad_1x1_f32", stream);
    const dim3 grid(pl.slices, pl.n_tiles * pl.c_tiles);
    static const int no_plain = getenv("UD_F32_WGRAD_PLAIN") ? !atoi(getenv("UD_F32_WGRAD_PLAIN")) : 0;
    const int plain = no_plain ? 0 : ((xm.mode == 0 && (size_t)P * Cin * 4 < (size_t)kOobW) ? 1 : 0) |
                                     ((ym.mode == 0 && (size_t)P * Cout * 4 < (size_t)kOobW) ? 2 : 0);
#define UD_WG1(NTv, PL, LDSB) \
  k_conv1x1_wgrad_f32<NTv, PL><<<grid, 256, LDSB, stream>>>(x, dy, partial, P, Cin, Cout, pl.c_tiles, pl.steps_per_slice, xm, ym)
#define UD_WG1P(NTv, LDSB)                                                                    \
  do {                                                                                        \
    if (plain == 3) UD_WG1(NTv, 3, LDSB); else if (plain == 2) UD_WG1(NTv, 2, LDSB);          \
    else if (plain == 1) UD_WG1(NTv, 1, LDSB); else UD_WG1(NTv, 0, LDSB);                     \
  } while (0)
    if (pl.nt == 128) UD_WG1P(128, 2 * kPS * (512 + 256));
    else UD_WG1P(64, 2 * kPS * (256 + 256));
#undef UD_WG1P
#undef UD_WG1
    UD_LAUNCH_CHECK();
  }
  return ud_wgrad_sum(partial, pl.slices, (size_t)Cout * Cin, dw, stream);
}
