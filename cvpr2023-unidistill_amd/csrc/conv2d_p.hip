// 3x3 / stride 1 / pad 1 convolution of channels-last bf16 tensors on v_mfma_f32_32x32x16_bf16 -- the PERSISTENT kernel of
// the BEV trunk / head / image branch (forward and, on transposed weights with the taps reversed, data gradient).
// Reference layers: BaseBEVBackbone (unidistill/layers/blocks_2d/det3d/base_bev_backbone.py:30-110), CenterHead.shared_conv /
// SepHead (unidistill/layers/head/det3d/center_head.py:58-99), the mmdet ResNet built in lss_fpn.py:143-149.
//
// Why a second 3x3 kernel (k_conv3x3_taps in conv2d.hip stays for small maps): measured on MI355X (tools/mfma_peak.hip,
// tools/dma_rate.hip) v_mfma_f32_16x16x32_bf16 sustains 1.33 PFLOP/s in an MFMA-only loop, v_mfma_f32_32x32x16_bf16 2.4;
// LDS-DMA from L2-resident weights sustains 29 TB/s chip-wide, so staging is a latency / synchronisation problem, not a
// bandwidth one; and a tile of the old kernel spent ~10 k of its 28 k cycles in prologue + epilogue.  Hence:
//   * one 512-thread workgroup per CU, PERSISTENT: it walks a static, balanced range of work units (half tiles of 8 rows x 16
//     pixels x TN channels; two vertically adjacent units are computed as one 16 x 16 tile) -- the last round of a launch is
//     made of half tiles instead of idle CUs;
//   * the pipeline never drains between tiles: a "stage" is (tile, 64-channel slice of Cin) = nine taps; during a stage the
//     NEXT stage's 18 x 18 halo (waves 0..3: two 1-KiB LDS-DMA pieces per tap, awaited at tap 7) and the weights THREE taps
//     ahead (waves 4..7, 4-slot ring, counted vmcnt) are in flight, next tile included; one raw s_barrier per tap, and the
//     fragments of a tap's first k-step are read during the previous tap, so a tap opens with MFMAs.  The two
//     streams are issued by different waves because vmcnt retires in order: a weight slice queued behind an HBM halo piece
//     could not be waited for on its own;
//   * waves: 4 (pixel rows) x 2 (channels); a wave owns 2 x NBW blocks of 32 pixels x 32 channels, D[channel][pixel] =
//     W[channel][k] X[k][pixel]: the accumulator lane holds ONE pixel and 4-channel runs, so after a v_permlane32_swap a lane
//     owns 8 consecutive channels of its pixel -- the epilogue (bias, folded BN, residual, ReLU, bf16) stores 16 bytes
//     straight from registers, no LDS staging, no barrier;
//   * 128-byte LDS rows (64 channels) with the 16-byte slots XOR-swizzled by (column >> 1) & 7 on the DMA SOURCE address:
//     the 16 lanes that a ds_read_b128 services together (same k-group, 16 different pixels / channels) hit 16 different slots;
//   * BatchNorm partial sums (per work unit: sum and sum of squares of the stored bf16 values) by a 16-step butterfly
//     transpose-reduction across the lanes, then a fixed-order sum over the four pixel-row waves.
#include "ud_common.h"
#include "ud_prof.h"
#include "conv3x3_p.h"
#include <algorithm>
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kHaloW = 18;
constexpr int kHaloPieces = 41;                  // ceil(18 * 18 / 8) 1-KiB pieces (8 pixels x 128 B)
constexpr int kHaloBytes = kHaloPieces * 1024;
constexpr int kPiecesPerWave = 11;               // the four HALO waves: 4 x 11 >= 41 pieces; the surplus lands in a dump slot
constexpr int kThreads = 512;

struct PGeom {
  int B, H, W, Cin, Cout;
  int tiles_x, nb8;          // 16-pixel column tiles, 8-row half bands per image
  int units;                 // spatial half units = B * tiles_x * nb8
  long long total;           // units * n tiles
};
struct PEp {
  const float* bias;
  const float* scale;
  const float* shift;
  const unsigned short* residual;
  int relu, reverse_taps;
  float* stats;              // [units][Cout][2] or nullptr
};

__device__ __attribute__((aligned(16))) unsigned int g_zero_p[4];

__device__ __forceinline__ void dma16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ bf16x8 lds_frag(const char* smem, unsigned addr) {
  return *reinterpret_cast<const bf16x8*>(smem + addr);
}
// Fragment read the compiler does not track: it would wait lgkmcnt(0) before the first use of ANY outstanding LDS read,
// i.e. for the reads of the next k-step it has just issued; here the waits are counted by hand (UD_WAITSET).
__device__ __forceinline__ void lds_read_asm(bf16x8& dst, unsigned addr, int imm) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm));
}

// One work item of a workgroup's range: a full tile (two half units) or a half tile.
struct Tile {
  int b, ty0, tx0, n0, full, unit;   // unit = spatial half-unit index (statistics slice)
};

template <int NBW, int ABL = 0>   // ABL: timing ablations (wrong results), see ud_conv3x3_p_launch
__global__ __launch_bounds__(kThreads) void k_conv3x3_p(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ w,
                                                        unsigned short* __restrict__ y, PGeom gm, PEp ep) {
  constexpr int TN = 64 * NBW;
  constexpr int kWBytes = TN * 128;                  // one tap's weight slice (TN channels x 64 k)
  constexpr int kHaloOff = 4 * kWBytes;            // 4-slot weight ring (three taps of look-ahead)
  constexpr int kDumpOff = kHaloOff + 2 * kHaloBytes;
  constexpr int kStatOff = kDumpOff + 1024;          // [4 pixel-row waves][TN][2] floats
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, kh = lane >> 5, col = lane & 15, rsel = (lane >> 4) & 1;
  const int r8 = lane >> 3, slot = lane & 7;
  const char* zero = reinterpret_cast<const char*>(g_zero_p);
  const int nchunks = gm.Cin >> 6;

  // ---- this workgroup's range of work units ------------------------------------------------------------------------
  const long long u_begin = gm.total * blockIdx.x / gridDim.x, u_end = gm.total * (blockIdx.x + 1) / gridDim.x;
  if (u_begin >= u_end) return;
  auto decode = [&](long long v, Tile& t) -> int {   // -> units consumed (1 or 2); everything here is wave-uniform
    const int nt = (int)(v / gm.units);
    int s = (int)(v - (long long)nt * gm.units);
    t.unit = __builtin_amdgcn_readfirstlane(s);
    t.n0 = __builtin_amdgcn_readfirstlane(nt * TN);
    const int hb = s % gm.nb8;
    s /= gm.nb8;
    const int tx = s % gm.tiles_x;
    t.b = __builtin_amdgcn_readfirstlane(s / gm.tiles_x);
    t.ty0 = __builtin_amdgcn_readfirstlane(hb * 8);
    t.tx0 = __builtin_amdgcn_readfirstlane(tx * 16);
    t.full = __builtin_amdgcn_readfirstlane(((hb & 1) == 0 && hb + 1 < gm.nb8 && v + 1 < u_end) ? 1 : 0);
    return 1 + t.full;
  };

  // ---- per-lane constants ---------------------------------------------------------------------------------------------
  // X fragment addresses (pixel operand): [dx][ks] for pixel (row rsel of the block, column col + dx), k-group 2 ks + kh
  unsigned xa[3][4];        // + the stage's offset (halo buffer, block rows), updated in place per stage
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      xa[dx][ks] = kHaloOff + (unsigned)((rsel * kHaloW + col + dx) * 128 + (((2 * ks + kh) ^ (((col + dx) >> 1) & 7)) << 4));
  unsigned xoff_prev = 0;
  // W fragment addresses (channel operand): row = channel 32 NBW wn + l31 of the slice (+ 32 per block)
  unsigned wa[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    wa[ks] = (unsigned)((32 * NBW * wn + l31) * 128 + (((2 * ks + kh) ^ ((l31 >> 1) & 7)) << 4));

  // ---- stage descriptors --------------------------------------------------------------------------------------------
  Tile cur, nxt;
  nxt.b = nxt.ty0 = nxt.tx0 = nxt.n0 = nxt.full = nxt.unit = 0;
  long long v = u_begin;
  v += decode(v, cur);
  int chunk = 0;
  // halo sources of a (tile, chunk): pointer per piece (zero page for out-of-image / out-of-tile pixels)
  // (32-bit byte offset from x; ~0u = the zero page: out-of-image / out-of-tile pixels)
  auto halo_src = [&](const Tile& t, int ch, int i) -> unsigned {      // i is a literal at every call site
    int q = (wave + 4 * i) * 8 + r8;                // halo waves are waves 0..3
    asm volatile("" : "+v"(q));                     // recompute per use: hoisted out of the stage loop, the (qy, qx, swizzle)
    const int qy = q / kHaloW, qx = q - qy * kHaloW;   // triples of the 11 pieces would pin ~35 registers for the whole kernel
    const int gy = t.ty0 + qy - 1, gx = t.tx0 + qx - 1;
    const bool ok = qy < (t.full ? 18 : 10) && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
    const size_t off = (((size_t)(t.b * gm.H + gy) * gm.W + gx) * gm.Cin + ch * 64 + ((slot ^ ((qx >> 1) & 7)) << 3)) * 2;
    return ok ? (unsigned)off : ~0u;
  };
  auto halo_ptr = [&](unsigned off) -> const char* {
    return off == ~0u ? zero : reinterpret_cast<const char*>(x) + off;
  };
  auto halo_dst = [&](int buf, int i) -> char* {
    const int pi = wave + 4 * i;
    return smem + (pi < kHaloPieces ? kHaloOff + buf * kHaloBytes + pi * 1024 : kDumpOff);
  };
  // weight piece jj of (n0, chunk, tap): rows n = 8 (wave + 8 jj) + r8 of the slice; per-lane byte offset of (row, swizzled
  // slot) for an n tile (channels past Cout re-read the last one: never stored), plus a uniform (tap, slice) term
  constexpr int kWPer = 2 * NBW;                     // weight pieces per WEIGHT wave (waves 4..7) and tap
  struct WOff { unsigned o[kWPer]; };
  const int wq = wave & 3;
  auto w_rows = [&](int n0) {
    WOff r;
#pragma unroll
    for (int jj = 0; jj < kWPer; ++jj) {
      const int n = (wq + 4 * jj) * 8 + r8;
      r.o[jj] = (unsigned)(((size_t)min(n0 + n, gm.Cout - 1) * 9 * gm.Cin + ((slot ^ ((n >> 1) & 7)) << 3)) * 2);
    }
    return r;
  };
  auto w_issue = [&](const WOff& ro, int ch, int tap, int ring) {
    const int te = ep.reverse_taps ? 8 - tap : tap;
    const char* base = reinterpret_cast<const char*>(w) + ((size_t)te * gm.Cin + ch * 64) * 2;
#pragma unroll
    for (int jj = 0; jj < kWPer; ++jj) dma16(base + ro.o[jj], smem + ring * kWBytes + (wq + 4 * jj) * 1024);
  };

  f32x16 acc[NBW][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nb][mb][e] = 0.f;
  };
  zero_acc();

  // ---- prologue: halo of the first stage (waves 0..3), weights of its taps 0 and 1 (waves 4..7) -------------------------
  // The two DMA streams live in DIFFERENT waves: vmcnt retires in issue order, so a weight fetch queued behind a halo
  // piece (HBM latency) could not be waited for before that piece had landed; split, the weight waves count only L2 hits.
  const bool w_wave = wave >= 4;
  WOff wo_cur = w_rows(cur.n0), wo_nxt = wo_cur;
  if (w_wave) {
    w_issue(wo_cur, 0, 0, 0);
    w_issue(wo_cur, 0, 1, 1);
    w_issue(wo_cur, 0, 2, 2);
    wait_vm<kWPer>();                   // W(0), W(1) landed, W(2) in flight
  } else {
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) dma16(halo_ptr(halo_src(cur, 0, i)), halo_dst(0, i));
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  // Software pipeline across the tap barriers: the fragments of a tap's FIRST k-step are read during the previous tap (its
  // weights were confirmed one barrier earlier: three taps of look-ahead), so a tap opens with MFMAs instead of an LDS round
  // trip, and the DMA issue sits behind them.
  bf16x8 wf[2][NBW], xf[2][2];

  int buf = 0, phase = 0;                // ring slot of tap t of this stage = (t + phase) & 3  (9 = 1 mod 4)
  {
    const unsigned x0 = (unsigned)((cur.full ? wm * 4 : wm * 2) * kHaloW * 128);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) wf[0][nb] = lds_frag(smem, wa[0] + nb * 4096);
    xf[0][0] = lds_frag(smem, xa[0][0] + x0);
    xf[0][1] = lds_frag(smem, xa[0][0] + x0 + 2 * kHaloW * 128);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) wf[1][nb] = wf[0][nb];
    xf[1][0] = xf[1][1] = xf[0][0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  bool more = true;
  while (more) {
    // the stage after this one: next slice of the tile, or slice 0 of the next tile, or nothing (dummy fetches)
    const bool last_chunk = chunk + 1 == nchunks;
    bool have_next = true;
    int nchunk = chunk + 1;
    if (last_chunk) {
      nchunk = 0;
      if (v < u_end) v += decode(v, nxt);
      else have_next = false;
    }
    Tile nt;                  // by value, field by field: a reference selected at run time would put both structs in scratch
    nt.b = last_chunk ? nxt.b : cur.b; nt.ty0 = last_chunk ? nxt.ty0 : cur.ty0; nt.tx0 = last_chunk ? nxt.tx0 : cur.tx0;
    nt.n0 = last_chunk ? nxt.n0 : cur.n0; nt.full = last_chunk ? nxt.full : cur.full; nt.unit = last_chunk ? nxt.unit : cur.unit;
    if (last_chunk && have_next) wo_nxt = w_rows(nxt.n0);
    else if (!last_chunk) wo_nxt = wo_cur;
    // (no next stage: the two look-ahead weight fetches of taps 7 and 8 re-read slice 0 of the current n tile -- harmless
    //  traffic from L2 that keeps the per-tap vmcnt arithmetic uniform)
    // (the halo sources of the next stage are computed by the halo waves two per tap, right where they are issued)
    // fragment bases of this stage: halo buffer, and the block rows of a full (4 per wave) or half (2 per wave) tile
    const unsigned xoff = (unsigned)(buf * kHaloBytes + (cur.full ? wm * 4 : wm * 2) * kHaloW * 128);
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xa[dx][ks] += xoff - xoff_prev;
    xoff_prev = xoff;
    const int full = cur.full;
    // fragment offset of the NEXT stage relative to this one (other halo buffer, its tile kind)
    const unsigned xnext = (unsigned)((buf ^ 1) * kHaloBytes + (nt.full ? wm * 4 : wm * 2) * kHaloW * 128) - xoff;

  /* One tap = four k-steps of 16 channels, software-pipelined over two fragment sets: the reads of step k + 1 are issued  \
     before the MFMAs of step k (set 0 enters the tap already loaded -- read during the previous tap -- and leaves it    \
     holding the first step of the next one); the DMA issue of the wave's role comes last, under the draining MFMAs. */     \
#define UD_LOADSET(S, WBASE, XBASE, DY, FULL)                                                                           \
  if (!(ABL & 2)) {                                                                                                     \
    _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb) lds_read_asm(wf[S][nb], (WBASE), nb * 4096);                     \
    lds_read_asm(xf[S][0], (XBASE), (DY) * kHaloW * 128);                                                               \
    if (FULL) lds_read_asm(xf[S][1], (XBASE), (2 + (DY)) * kHaloW * 128);                                               \
  }
/* wait until at most N LDS reads are outstanding; the set's registers are in / out operands so that no MFMA on them can   \
   be scheduled above the wait (the compiler knows nothing about the asm loads' latency) */                               \
#define UD_WAITSET(S, N)                                                                                                \
  if (!(ABL & 2)) {                                                                                                     \
    if (NBW == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(wf[S][0]), "+v"(wf[S][NBW - 1]), "+v"(xf[S][0]), "+v"(xf[S][1]) : "n"(N)); \
    else asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(wf[S][0]), "+v"(xf[S][0]), "+v"(xf[S][1]) : "n"(N));               \
  }
#define UD_MMASET(S, FULL)                                                                                              \
  if (!(ABL & 1)) {                                                                                                     \
    _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                  \
      acc[nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[S][nb], xf[S][0], acc[nb][0], 0, 0, 0);                   \
    if (FULL) {                                                                                                         \
      _Pragma("unroll") for (int nb = 0; nb < NBW; ++nb)                                                                \
        acc[nb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[S][nb], xf[S][1], acc[nb][1], 0, 0, 0);                 \
    }                                                                                                                   \
  }
#define UD_TAP(T, FULL)                                                                                                 \
  {                                                                                                                     \
    constexpr int dy = (T) / 3, dx = (T) % 3;                                                                           \
    constexpr int ndy = (T) < 8 ? ((T) + 1) / 3 : 0, ndx = (T) < 8 ? ((T) + 1) % 3 : 0;                                 \
    constexpr int kReads = NBW + 1 + ((FULL) ? 1 : 0);                                                                  \
    constexpr int kReadsNext = NBW + 1 + (((FULL) || (T) == 8) ? 1 : 0);                                                \
    unsigned wslot = (unsigned)((((T) + phase) & 3) * kWBytes);                                                         \
    unsigned nslot = (unsigned)((((T) + 1 + phase) & 3) * kWBytes);                                                     \
    /* opaque: keeps the slot-relative fragment addresses of all nine taps from being precomputed per stage (~30 VGPRs) */ \
    asm volatile("" : "+s"(wslot), "+s"(nslot));                                                                        \
    const unsigned nx = xa[ndx][0] + ((T) < 8 ? 0u : xnext);                                                            \
    /* set 0 holds k-step 0 (read and awaited during the previous tap) */                                               \
    UD_LOADSET(1, wa[1] + wslot, xa[dx][1], dy, FULL)                                                                   \
    UD_MMASET(0, FULL)                                                                                                  \
    UD_LOADSET(0, wa[2] + wslot, xa[dx][2], dy, FULL)                                                                   \
    UD_WAITSET(1, kReads)                                                                                               \
    UD_MMASET(1, FULL)                                                                                                  \
    UD_LOADSET(1, wa[3] + wslot, xa[dx][3], dy, FULL)                                                                   \
    UD_WAITSET(0, kReads)                                                                                               \
    UD_MMASET(0, FULL)                                                                                                  \
    /* first k-step of the next tap (T = 8: tap 0 of the next stage, whose halo was awaited before barrier 7; it may be \
       a full tile while this one is not: always read both blocks then) */                                              \
    UD_LOADSET(0, wa[0] + nslot, nx, ndy, (FULL) || (T) == 8)                                                           \
    UD_WAITSET(1, kReadsNext)                                                                                           \
    UD_MMASET(1, FULL)                                                                                                  \
    /* weights three taps ahead (next stage for T >= 6) / two halo pieces of the next stage */                          \
    if (w_wave) {                                                                                                       \
      if (!(ABL & 8)) {                                                                                                 \
      if ((T) < 6) w_issue(wo_cur, chunk, (T) + 3, ((T) + 3 + phase) & 3);                                              \
      else w_issue(wo_nxt, nchunk, (T) - 6, ((T) + 3 + phase) & 3);                                                     \
      wait_vm<kWPer>();                      /* W(T + 2) landed; W(T + 3) may fly */                                   \
      }                                                                                                                 \
    } else {                                                                                                            \
      if ((T) < 6 && !(ABL & 4)) {                                                                                      \
        dma16(halo_ptr(have_next ? halo_src(nt, nchunk, 2 * (T)) : ~0u), halo_dst(buf ^ 1, 2 * (T)));                   \
        if (2 * (T) + 1 < kPiecesPerWave)                                                                               \
          dma16(halo_ptr(have_next ? halo_src(nt, nchunk, 2 * (T) + 1) : ~0u), halo_dst(buf ^ 1, 2 * (T) + 1));         \
      }                                                                                                                 \
      if ((T) == 7) wait_vm<0>();            /* the next stage's halo: visible after this barrier */                   \
    }                                                                                                                   \
    UD_WAITSET(0, 0)                         /* the next tap's first fragments */                                       \
    if (!(ABL & 16)) __builtin_amdgcn_s_barrier();                                                                      \
  }
#define UD_TAPS(FULL) UD_TAP(0, FULL) UD_TAP(1, FULL) UD_TAP(2, FULL) UD_TAP(3, FULL) UD_TAP(4, FULL) UD_TAP(5, FULL) \
                      UD_TAP(6, FULL) UD_TAP(7, FULL) UD_TAP(8, FULL)
    if (full) { UD_TAPS(true) } else { UD_TAPS(false) }
#undef UD_TAPS
#undef UD_TAP
#undef UD_MMASET
#undef UD_WAITSET
#undef UD_LOADSET

    buf ^= 1;
    phase = (phase + 1) & 3;
    if (!last_chunk) {
      ++chunk;
      continue;
    }
    // ---- epilogue of the tile: registers -> bias / folded BN / residual / ReLU -> bf16, 16-byte stores ---------------------
    float* sred = reinterpret_cast<float*>(smem + kStatOff);
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
      float st1[16], st2[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) st1[e] = st2[e] = 0.f;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        if (mb == 1 && !cur.full) break;
        const int gy = cur.ty0 + (cur.full ? wm * 4 + mb * 2 : wm * 2) + rsel, gx = cur.tx0 + col;
        const bool pix_ok = gy < gm.H && gx < gm.W;
        const size_t pix = ((size_t)(cur.b * gm.H + gy) * gm.W + gx) * gm.Cout;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float vv[8];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[nb][mb][8 * p + k]),
                                                             __float_as_uint(acc[nb][mb][8 * p + 4 + k]), false, false);
            vv[k] = __uint_as_float(sw[0]);
            vv[4 + k] = __uint_as_float(sw[1]);
          }
          const int n = cur.n0 + 32 * NBW * wn + 32 * nb + 16 * p + 8 * kh;
          const bool ok = pix_ok && n < gm.Cout;
          if (ep.bias && n < gm.Cout) {
            const float4 b0 = *reinterpret_cast<const float4*>(ep.bias + n), b1 = *reinterpret_cast<const float4*>(ep.bias + n + 4);
            vv[0] += b0.x; vv[1] += b0.y; vv[2] += b0.z; vv[3] += b0.w;
            vv[4] += b1.x; vv[5] += b1.y; vv[6] += b1.z; vv[7] += b1.w;
          }
          if (ep.scale && n < gm.Cout) {
            const float4 s0 = *reinterpret_cast<const float4*>(ep.scale + n), s1 = *reinterpret_cast<const float4*>(ep.scale + n + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(ep.shift + n), h1 = *reinterpret_cast<const float4*>(ep.shift + n + 4);
            vv[0] = vv[0] * s0.x + h0.x; vv[1] = vv[1] * s0.y + h0.y; vv[2] = vv[2] * s0.z + h0.z; vv[3] = vv[3] * s0.w + h0.w;
            vv[4] = vv[4] * s1.x + h1.x; vv[5] = vv[5] * s1.y + h1.y; vv[6] = vv[6] * s1.z + h1.z; vv[7] = vv[7] * s1.w + h1.w;
          }
          if (ep.residual && ok) {
            const uint4 h = *reinterpret_cast<const uint4*>(ep.residual + pix + n);
            const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              vv[2 * e] += __uint_as_float(hw[e] << 16);
              vv[2 * e + 1] += __uint_as_float(hw[e] & 0xFFFF0000u);
            }
          }
          if (ep.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] = fmaxf(vv[e], 0.f);
          }
          const uint4 pk = make_uint4(ud_pack_bf16x2(vv[0], vv[1]), ud_pack_bf16x2(vv[2], vv[3]),
                                      ud_pack_bf16x2(vv[4], vv[5]), ud_pack_bf16x2(vv[6], vv[7]));
          if (ok && !(ABL & 32)) *reinterpret_cast<uint4*>(y + pix + n) = pk;
          if (ep.stats) {
            const unsigned pw[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float lo = ok ? __uint_as_float(pw[e] << 16) : 0.f, hi = ok ? __uint_as_float(pw[e] & 0xFFFF0000u) : 0.f;
              st1[8 * p + 2 * e] += lo; st2[8 * p + 2 * e] += lo * lo;
              st1[8 * p + 2 * e + 1] += hi; st2[8 * p + 2 * e + 1] += hi * hi;
            }
          }
        }
      }
      if (ep.stats) {
        // butterfly transpose-reduction over the 32 lanes of a half wave: 16 values -> each lane ends with the total of value
        // index e = 8 b4 + 4 b3 + 2 b2 + b1 (bits of the lane), duplicated over b0
        const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          float a8[8], a4[4], a2[2];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float lo = which ? st2[e] : st1[e], hi = which ? st2[8 + e] : st1[8 + e];
            a8[e] = (b4 ? hi : lo) + __shfl_xor(b4 ? lo : hi, 16);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) a4[e] = (b3 ? a8[4 + e] : a8[e]) + __shfl_xor(b3 ? a8[e] : a8[4 + e], 8);
#pragma unroll
          for (int e = 0; e < 2; ++e) a2[e] = (b2 ? a4[2 + e] : a4[e]) + __shfl_xor(b2 ? a4[e] : a4[2 + e], 4);
          float a1 = (b1 ? a2[1] : a2[0]) + __shfl_xor(b1 ? a2[0] : a2[1], 2);
          a1 += __shfl_xor(a1, 1);
          if (!(lane & 1)) {
            const int e = (b4 ? 8 : 0) + (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
            const int c = 32 * NBW * wn + 32 * nb + 16 * (e >> 3) + 8 * kh + (e & 7);
            sred[(wm * TN + c) * 2 + which] = a1;
          }
        }
      }
    }
    if (ep.stats) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // sred is written again only after the nine tap barriers of the next stage
      if (tid < 2 * TN) {
        const int c = tid >> 1, which = tid & 1;
        if (cur.n0 + c < gm.Cout) {
          float sum = 0.f;
#pragma unroll
          for (int m = 0; m < 4; ++m) sum += sred[(m * TN + c) * 2 + which];
          ep.stats[((size_t)cur.unit * gm.Cout + cur.n0 + c) * 2 + which] = sum;
          if (cur.full) ep.stats[((size_t)(cur.unit + 1) * gm.Cout + cur.n0 + c) * 2 + which] = 0.f;
        }
      }
    }
    zero_acc();
    chunk = 0;
    if (have_next) { cur = nxt; wo_cur = wo_nxt; }
    else more = false;
  }
  wait_vm<0>();        // the dummy fetches of the last stage
}

}  // namespace

// Dispatch switch (process-wide): 0 = off (default: on the step's shapes this kernel measures 0.95-1.07x the tap kernel of
// conv2d.hip -- profiles/r04_conv_bf16.md), 1 = on wherever it applies.  UD_CONV_P in the environment sets the initial value.
static int g_conv_p_mode = getenv("UD_CONV_P") ? atoi(getenv("UD_CONV_P")) : 0;

extern "C" int ud_conv3x3_persistent(int mode) {
  const int old = g_conv_p_mode;
  if (mode >= 0) g_conv_p_mode = mode;
  return old;
}

bool ud_conv3x3_p_supported(int B, int H, int W, int Cin, int Cout) {
  if (!g_conv_p_mode) return false;
  if (Cin % 64 != 0 || Cout % 8 != 0) return false;
  // maps that fill 8 x 16 half tiles reasonably, and enough work units for the 256 workgroups
  if (H < 8 || W < 12) return false;
  const long long units = (long long)B * ud_div_up(W, 16) * ud_div_up(H, 8) * ud_div_up(Cout, Cout <= 64 ? 64 : 128);
  return units >= 384;
}

int ud_conv3x3_p_slices(int B, int H, int W) { return B * ud_div_up(W, 16) * ud_div_up(H, 8); }

int ud_conv3x3_p_launch(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, const float* bias,
                        const float* scale, const float* shift, const void* residual, int relu, int reverse_taps,
                        float* stats, hipStream_t stream) {
  PGeom gm;
  gm.B = B; gm.H = H; gm.W = W; gm.Cin = Cin; gm.Cout = Cout;
  gm.tiles_x = ud_div_up(W, 16);
  gm.nb8 = ud_div_up(H, 8);
  gm.units = B * gm.tiles_x * gm.nb8;
  const bool narrow = Cout <= 64;
  const int ntn = ud_div_up(Cout, narrow ? 64 : 128);
  gm.total = (long long)gm.units * ntn;
  PEp ep{bias, scale, shift, reinterpret_cast<const unsigned short*>(residual), relu, reverse_taps, stats};
  static UdDeviceOnce attr_set;
  constexpr size_t lds128 = 4 * 128 * 128 + 2 * kHaloBytes + 1024 + 4 * 128 * 2 * 4;
  constexpr size_t lds64 = 4 * 64 * 128 + 2 * kHaloBytes + 1024 + 4 * 64 * 2 * 4;
  if (const unsigned long long bit = attr_set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_p<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_p<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64));
    attr_set.mark(bit);
  }
  static const int grid_cap = getenv("UD_CONV_P_GRID") ? atoi(getenv("UD_CONV_P_GRID")) : 256;      // one workgroup per CU
  const int grid = (int)std::min<long long>(gm.total, grid_cap > 0 ? grid_cap : 256);
  const unsigned short* xs = reinterpret_cast<const unsigned short*>(x);
  const unsigned short* ws = reinterpret_cast<const unsigned short*>(w);
  unsigned short* ys = reinterpret_cast<unsigned short*>(y);
  static const int abl = getenv("UD_CONV_P_ABL") ? atoi(getenv("UD_CONV_P_ABL")) : 0;
  if (abl && !narrow) {       // timing ablations (results are wrong): 1 no MFMA, 2 no LDS reads, 4 no halo DMA, 8 no weight DMA, 16 no barrier
#define UD_ABL(A) case A: UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_p<2, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128)); \
                          k_conv3x3_p<2, A><<<grid, kThreads, lds128, stream>>>(xs, ws, ys, gm, ep); break;
    switch (abl) { UD_ABL(1) UD_ABL(2) UD_ABL(3) UD_ABL(4) UD_ABL(8) UD_ABL(12) UD_ABL(16) UD_ABL(28) UD_ABL(30) UD_ABL(60) UD_ABL(62) default: break; }
#undef UD_ABL
    UD_LAUNCH_CHECK();
    return UD_OK;
  }
  if (narrow) k_conv3x3_p<1><<<grid, kThreads, lds64, stream>>>(xs, ws, ys, gm, ep);
  else k_conv3x3_p<2><<<grid, kThreads, lds128, stream>>>(xs, ws, ys, gm, ep);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
