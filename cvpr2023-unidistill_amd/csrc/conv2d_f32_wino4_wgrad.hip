// fp32 weight gradient of the 3x3 / stride 1 / pad 1 convolutions as the gradient of the Winograd F(4x4, 3x3) form
// (conv2d_f32_wino4.hip): with y = A^T [sum_c U . V] A per 4 x 4 tile,
//
//   dU[f][n][c] = sum_tiles (A dy A^T)[f][tile][n] * (B^T x B)[f][tile][c]        36 GEMMs, reduction index = tile
//   dW[n][.][c] = G^T dU[.][n][c] G                                              (k_wino4_wgrad_finish, after the ordered slice sum)
//
// 36 multiplications per 16 output pixels per (n, c) instead of 144 (direct) / 64 (F(2x2), conv2d_f32_wino_wgrad.hip).  Replaces
// the backward-weights pass of the same layers (base_bev_backbone.py:38-115, center_head.py:58-99,311-355, lss_fpn.py:143-149).
//
// A workgroup (8 waves) owns a 64 (n) x 32 (c) block of dU for ALL 36 frequencies over a slice of the tile stages: wave
// (wn, wc) holds the 36 frequencies of a 16 x 16 block (144 accumulator registers) -- the MFMA loop is the forward kernel's, with
// DY' = A dy A^T in the place of the transformed filters: per stage of four tiles (one MFMA K step) ONE ds_read_b128 of DY' and
// one of V feed four MFMAs (frequencies four to a 16-byte word: [f / 4][16-row group][tile][row % 16][f % 4]).  Both operands
// are made in the kernel, one stage ahead, by all eight waves with packed fp32 arithmetic (fp32 MFMAs run on the vector ALUs:
// every other VALU instruction costs its issue time):
//   * V: the 6 x 18 x 32-channel x patch of the stage's four tiles (one tile row, x = 4 tx0 .. 4 tx0 + 15) arrives by LDS-DMA
//     (128 bytes per pixel: whole lines); thread = (row group of B^T d, tile, channel), the forward kernel's transform;
//   * DY': the four 4 x 4 x 64-channel dy tiles are loaded straight into registers a stage ahead (16 coalesced dwords per
//     thread; a raw LDS copy would not fit beside the double-buffered operands); thread = (rows (0, 1, 2) | (3, 4, 5) of A dy,
//     tile, channel).
// Tile stages are cut into slices (one workgroup per (slice, n block, c block)); the slices' partial dU are added in a fixed
// order by k_wino4_wgrad_finish (deterministic; no atomics).
#include "ud_common.h"
#include "ud_prof.h"
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kNB = 64, kCB = 32, kTS = 4;             // dU block and tiles per stage
constexpr int kDBytes = 9 * 4 * kTS * 16 * 16;          // DY' of a stage: [9 fq][4 n groups][4 tiles][16 n][4 f]   36 864
constexpr int kVBytes = 9 * 2 * kTS * 16 * 16;          // V of a stage:   [9 fq][2 c groups][4 tiles][16 c][4 f]   18 432
constexpr int kXRow = 2 * 18 * 64;                      // a patch row: [channel half][18 columns in x % 4 planes][16 channels]
constexpr int kXBytes = 14336;                          // 6 rows (13 824 bytes) in 14 DMA pieces of 1 KB
constexpr int kX0 = 0, kD0 = 2 * kXBytes, kV0 = kD0 + 2 * kDBytes, kDump = kV0 + 2 * kVBytes, kSmem = kDump + 1024;   // 140 288
constexpr unsigned kInv = 0x80000000u;                  // voffset of a lane that must read zeros (descriptors hold < 2 GB)

struct G4Geom {
  int B, H, W, Cin, Cout, TY, SX;   // tile rows per image, stages (4 tiles) per tile row
  int nstages, sps;                 // stages in total / per slice (even)
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(size_t)lds_wave_base, 16, voff, soff, 0, 0);
}

// frequency order shared with the forward kernel: f' = 6 * slot(i) + cpos(j)
__host__ __device__ constexpr int g4_slot(int i) { return i == 0 ? 0 : i == 5 ? 1 : i + 1; }
__host__ __device__ constexpr int g4_cpos(int j) { return j == 0 ? 0 : j == 5 ? 1 : j == 1 ? 2 : j == 3 ? 3 : j == 2 ? 4 : 5; }
__host__ __device__ constexpr int g4_f(int i, int j) { return 6 * g4_slot(i) + g4_cpos(j); }

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

// LDS operations of the two transforms that ride in MFMA step fq (after the step's fragment loads): x patch-row reads (3
// ds_read2_b32 each) in steps 0-3, DY' writes (3 b128 in step 2, 3 b64 in step 4), V writes (4) in step 7
__device__ constexpr int kStepOps[9] = {3, 3, 6, 3, 3, 0, 0, 4, 0};
__host__ __device__ constexpr int g4_wait(int fq) { return (fq > 0 ? kStepOps[fq - 1] : 0) + 2 + kStepOps[fq]; }

__global__ __launch_bounds__(512) void k_wino4_wgrad_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ partial, G4Geom gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned sbase = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wn = wave & 3, wc = wave >> 2;                         // MFMA block of this wave
  const int n0 = blockIdx.y * kNB, c0 = blockIdx.z * kCB;
  const int s_begin = blockIdx.x * gm.sps;
  const int nst = gm.sps;                                          // (stages past gm.nstages contribute zeros)
  constexpr int O0 = 0, O1 = 5, O2 = 10, O3 = 14;                  // column planes x % 4 = 0 / 1 / 2 / 3 of a patch row (5, 5, 4, 4 columns)
  const long long xbytes = (long long)gm.B * gm.H * gm.W * gm.Cin * 4, dbytes = (long long)gm.B * gm.H * gm.W * gm.Cout * 4;

  // ---- stage geometry (wave-uniform): stage s = (image b, tile row ty, four tiles from tile column 4 sx).  The byte offsets of
  // its first pixel in x (channel c0) and dy travel with it and are stepped by constants (the scalar unit is not free beside
  // fp32 MFMAs: re-deriving them with 64-bit multiplies cost ~140 scalar instructions per wave and stage).
  struct StageAt {
    int b, ty, sx;
    bool rowin;                     // b < B and the patch rows 4 ty - 1 .. 4 ty + 4 lie inside the image (changes with ty only)
    bool inner;                     // every pixel the stage touches lies inside the image (x patch rows 4 ty - 1 .. + 4, columns 16 sx - 1 .. + 16)
    long long xoff, doff;           // bytes: x pixel (b, 4 ty - 1, 16 sx - 1) channel c0; dy pixel (b, 4 ty, 16 sx) channel 0 (images past B: image 0)
  };
  const long long x_sx = 16ll * gm.Cin * 4, d_sx = 16ll * gm.Cout * 4;                       // next stage in the tile row
  const long long x_ty = (4ll * gm.W - 16ll * gm.SX) * gm.Cin * 4, d_ty = (4ll * gm.W - 16ll * gm.SX) * gm.Cout * 4;   // + row wrap
  const long long x_b = ((long long)gm.H - 4ll * gm.TY) * gm.W * gm.Cin * 4, d_b = ((long long)gm.H - 4ll * gm.TY) * gm.W * gm.Cout * 4;
  // interior tile rows are 1 .. ty_in, interior stages of a row 1 .. sx_in (one unsigned compare each; the row's half is
  // re-evaluated only where ty changes: the classification used to cost ~20 scalar instructions per stage, twice)
  const unsigned ty_in = (unsigned)max((gm.H - 4 + 3) / 4 - 1, 0), sx_in = (unsigned)max((gm.W - 16 + 15) / 16 - 1, 0);
  auto classify_row = [&](StageAt& a) { a.rowin = a.b < gm.B && (unsigned)(a.ty - 1) < ty_in; };
  auto classify = [&](StageAt& a) { a.inner = a.rowin && (unsigned)(a.sx - 1) < sx_in; };
  auto decode = [&](int s) {
    StageAt a;
    const int per = gm.TY * gm.SX;
    a.b = s / per;
    const int r = s - a.b * per;
    a.ty = r / gm.SX;
    a.sx = r - a.ty * gm.SX;
    const long long bb = a.b < gm.B ? a.b : 0;
    a.xoff = (((bb * gm.H + 4 * a.ty - 1) * gm.W + 16 * a.sx - 1) * gm.Cin + c0) * 4;
    a.doff = (((bb * gm.H + 4 * a.ty) * gm.W + 16 * a.sx) * gm.Cout) * 4;
    classify_row(a);
    classify(a);
    return a;
  };
  auto advance = [&](StageAt& a) {
    a.xoff += x_sx, a.doff += d_sx;
    if (++a.sx == gm.SX) {
      a.sx = 0;
      a.xoff += x_ty, a.doff += d_ty;
      if (++a.ty == gm.TY) {
        a.ty = 0, ++a.b;
        a.xoff += x_b, a.doff += d_b;
        if (a.b >= gm.B) a.xoff = ((-(long long)gm.W - 1) * gm.Cin + c0) * 4, a.doff = 0;   // stages past the last image: image 0 again (all masked)
      }
      classify_row(a);
    }
    classify(a);
  };
  auto interior = [&](const StageAt& a) { return a.inner; };

  // ---- x patch DMA: piece p (two per wave: wave, wave + 8 < 14) fills 16-byte slots [64 p, 64 p + 64) of the patch buffer;
  // slot -> (row, channel half, column position in its plane, 4-channel group); lane constants: the byte offset from the patch's
  // top-left pixel (patch row qy, column qx) and (qy, qx) themselves for the masked path of edge stages
  int xq[2];                        // qy << 8 | qx, -1: padding slot
  unsigned xo[2];                   // byte offset of the lane's 16 bytes relative to pixel (4 ty - 1, 16 sx - 1), channel c0
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = (wave + 8 * i) * 64 + lane;
    const int row = q / 144, rem = q - row * 144;
    const int ch = rem / 72, rem2 = rem - ch * 72;
    const int pos = rem2 >> 2, c4 = rem2 & 3;
    const int qx = pos < O1 ? 4 * pos : pos < O2 ? 4 * (pos - O1) + 1 : pos < O3 ? 4 * (pos - O2) + 2 : 4 * (pos - O3) + 3;
    const bool ok = row < 6 && wave + 8 * i < 14;
    xq[i] = ok ? (row << 8 | qx) : -1;
    xo[i] = ok ? (unsigned)(((row * gm.W + qx) * gm.Cin + 16 * ch + 4 * c4) * 4) : kInv;
  }
  auto issue_x = [&](const StageAt& a, int buf) {
    // descriptor based at the patch's top-left pixel (clamped to the tensor for the masked path)
    if (interior(a)) {
      const __amdgpu_buffer_rsrc_t r =
          __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(x) + a.xoff), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i)      // always two pieces per wave (a constant count for the vmcnt waits); pieces 14, 15: zeros into the dump
        dma16(r, xo[i], 0, wave + 8 * i < 14 ? sbase + kX0 + buf * kXBytes + (wave + 8 * i) * 1024 : sbase + kDump);
    } else {
      // edge stage: per-lane range test; offsets from the tensor base (they may not be negative)
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)xbytes, 0x00020000);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int qy = xq[i] >> 8, qx = xq[i] & 255;
        const int gy = 4 * a.ty - 1 + qy, gx = 16 * a.sx - 1 + qx;
        const bool ok = xq[i] >= 0 && a.b < gm.B && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
        const unsigned vo = ok ? (unsigned)a.xoff + xo[i] : kInv;
        dma16(r, vo, 0, wave + 8 * i < 14 ? sbase + kX0 + buf * kXBytes + (wave + 8 * i) * 1024 : sbase + kDump);
      }
    }
  };

  // ---- dy tiles: thread = (rows of A dy: wave >> 2, n group wave & 3, tile lane >> 4, channel lane & 15); 16 dwords a stage ahead
  const int dng = wave & 3, drole = wave >> 2, dt = lane >> 4;
  const unsigned dlane = (unsigned)((4 * dt * gm.Cout + n0 + 16 * dng + li) * 4);    // tile dt starts 4 pixels right of the stage's first
  // the NEXT stage's raw dy: pixel rows 0..3, column pairs (0, 1) / (2, 3) -- named registers (an array indexed through the
  // lambdas below ended up in scratch memory: its loads then drained the whole vmcnt queue)
  f32x2 rd00, rd01, rd10, rd11, rd20, rd21, rd30, rd31;
  __amdgpu_buffer_rsrc_t rdy_cur;
  auto dy_rsrc = [&](const StageAt& a) {
    const long long left = dbytes - a.doff;             // bytes from there to the end of the tensor: later pixels read zeros
    rdy_cur = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(dy) + a.doff), 0,
                                                (int)(left > 0x7fffffffll ? 0x7fffffffll : left), 0x00020000);
  };
#define G4_LOAD_DY(PY, A, B_)                                                                                                  \
  do {                                                                                                                         \
    const unsigned so_ = (unsigned)((PY) * gm.W * gm.Cout * 4);                                                                \
    A.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy_cur, dlane, so_, 0));                             \
    A.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy_cur, dlane, so_ + gm.Cout * 4, 0));               \
    B_.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy_cur, dlane, so_ + gm.Cout * 8, 0));              \
    B_.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy_cur, dlane, so_ + gm.Cout * 12, 0));             \
  } while (0)
#define G4_LOAD_DY_ALL()            \
  do {                              \
    G4_LOAD_DY(0, rd00, rd01);      \
    G4_LOAD_DY(1, rd10, rd11);      \
    G4_LOAD_DY(2, rd20, rd21);      \
    G4_LOAD_DY(3, rd30, rd31);      \
  } while (0)
  // pixels of an edge stage outside the image hold whatever lies there (the next row / image): zero them before the transform
  auto mask_dy = [&](const StageAt& a) {
    if (interior(a)) return;
    const int xr = gm.W - 16 * a.sx - 4 * dt;            // columns of this lane's tile inside the image
    const bool y0 = a.b < gm.B && 4 * a.ty + 0 < gm.H, y1 = a.b < gm.B && 4 * a.ty + 1 < gm.H, y2 = a.b < gm.B && 4 * a.ty + 2 < gm.H,
               y3 = a.b < gm.B && 4 * a.ty + 3 < gm.H;
#define G4_MASK(Y, A, B_)                     \
  A.x = (Y && xr > 0) ? A.x : 0.f;            \
  A.y = (Y && xr > 1) ? A.y : 0.f;            \
  B_.x = (Y && xr > 2) ? B_.x : 0.f;          \
  B_.y = (Y && xr > 3) ? B_.y : 0.f;
    G4_MASK(y0, rd00, rd01) G4_MASK(y1, rd10, rd11) G4_MASK(y2, rd20, rd21) G4_MASK(y3, rd30, rd31)
#undef G4_MASK
  };

  // ---- x transform (the forward kernel's): thread = (row group vg, channel half, tile, channel)
  const int vg = wave >> 1, chalf = wave & 1;
  const unsigned tprow = sbase + kX0 + chalf * (18 * 64) + (lane >> 4) * 64 + li * 4;      // patch row 0, buffer 0
  const int rows4 = vg == 0 ? 0x4420 : vg == 1 ? 0x5531 : 0x1324;                          // r1 | r2 << 4 | r3 << 8 | r4 << 12
  const unsigned ro1 = (rows4 & 15) * kXRow, ro2 = ((rows4 >> 4) & 15) * kXRow, ro3 = ((rows4 >> 8) & 15) * kXRow,
                 ro4 = ((rows4 >> 12) & 15) * kXRow;
  const float ca1 = vg < 2 ? 4.f : 1.f, ca2 = vg < 2 ? -5.f : vg == 2 ? -4.f : -1.f;
  const float cb3 = vg == 0 ? 1.f : vg == 1 ? -1.f : vg == 2 ? 1.f : 2.f, cb4 = vg < 2 ? 0.f : vg == 2 ? -4.f : -2.f;
  const f32x2 ca1v = {ca1, ca1}, ca2v = {ca2, ca2}, cb3v = {cb3, cb3}, cb4v = {cb4, cb4};
  const unsigned tvbase = sbase + kV0 + chalf * 1024 + lane * 16;                          // [fq][c half][tile][c % 16][4]
  const unsigned tdump = sbase + kDump + lane * 16;
  const int vq0 = vg < 2 ? 0 : vg == 2 ? 3 : 6;
  const unsigned wA = vg == 1 ? tdump : tvbase + vq0 * 2048;
  const unsigned wB = vg == 1 ? tdump : tvbase + (vq0 + 1) * 2048;
  const unsigned wC = vg == 0 ? tdump : tvbase + (vq0 + 1) * 2048 + 8;
  const unsigned wD = vg == 0 ? tdump : tvbase + (vq0 + 2) * 2048;
  const unsigned vdelta01 = vg == 1 ? 0 : kVBytes, vdelta23 = vg == 0 ? 0 : kVBytes;      // V buffer 1 - buffer 0 (0 for the dump)

#define G4_ROWREAD(D, RO)                                                                                               \
  do {                                                                                                                  \
    const unsigned a_ = tprow + (RO);                                                                                   \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[0]) : "v"(a_), "n"(O0 * 16), "n"(O1 * 16));              /* j = 0, 1 */ \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[1]) : "v"(a_), "n"(O2 * 16), "n"(O3 * 16));              /* j = 2, 3 */ \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(D[2]) : "v"(a_), "n"((O0 + 1) * 16), "n"((O1 + 1) * 16));  /* j = 4, 5 */ \
  } while (0)

  // v = w B, B^T of F(4, 3): pairs (w0, w1), (w2, w3), (w4, w5) -> (v0, v5), (v1, v3), (v2, v4)
  auto row_pass_x = [](const f32x2 (&w)[3], f32x2 (&v)[3]) {
    const f32x2 c4 = {4.f, 4.f}, c5 = {-5.f, -5.f}, c41 = {-4.f, -1.f}, c12 = {1.f, 2.f}, cn12 = {-1.f, -2.f};
    v[0] = __builtin_elementwise_fma(c4, w[0], __builtin_elementwise_fma(c5, w[1], w[2]));
    const f32x2 e = __builtin_elementwise_fma(c41, (f32x2){w[1].x, w[1].x}, (f32x2){w[2].x, w[2].x});
    const f32x2 o = __builtin_elementwise_fma(c41, (f32x2){w[0].y, w[0].y}, (f32x2){w[1].y, w[1].y});
    v[1] = __builtin_elementwise_fma(c12, o, e);
    v[2] = __builtin_elementwise_fma(cn12, o, e);
  };
  // v = w A^T, A of F(4, 3) = [1 0 0 0; 1 1 1 1; 1 -1 1 -1; 1 2 4 8; 1 -2 4 -8; 0 0 0 1]: pairs (w0, w1), (w2, w3) -> (v0, v5), (v1, v3), (v2, v4)
  auto row_pass_d = [](const f32x2 (&w)[2], f32x2 (&v)[3]) {
    const f32x2 c14 = {1.f, 4.f}, c12 = {1.f, 2.f}, c18 = {1.f, 8.f};
    v[0] = (f32x2){w[0].x, w[1].y};
    const f32x2 e = __builtin_elementwise_fma(c14, (f32x2){w[1].x, w[1].x}, (f32x2){w[0].x, w[0].x});     // (w0 + w2, w0 + 4 w2)
    const f32x2 o = __builtin_elementwise_fma(c18, (f32x2){w[1].y, w[1].y}, c12 * (f32x2){w[0].y, w[0].y});   // (w1 + w3, 2 w1 + 8 w3)
    v[1] = e + o;
    v[2] = e - o;
  };
  // DY' roles: role 0 = rows (1, 2) of A dy as the +- pair and row 0 (= d0) as the single row; role 1 = rows (3, 4) and row 5 (= d3)
  const float dcx = drole ? 4.f : 1.f, dcy1 = drole ? 2.f : 1.f, dcy3 = drole ? 8.f : 1.f;
  const f32x2 dcxv = {dcx, dcx}, dcy1v = {dcy1, dcy1}, dcy3v = {dcy3, dcy3};
  const unsigned tdbase = sbase + kD0 + dng * 1024 + lane * 16;                            // [fq][n group][tile][n % 16][4]
  const unsigned dP = tdbase + (drole ? 6 : 3) * 4096;                                     // three whole words: the +- pair's 12 frequencies
  // the single row's six frequencies: role 0 -> f' 0..5 (word 0, word 1 low half), role 1 -> f' 6..11 (word 1 high half, word 2)
  const unsigned dS0 = tdbase + (drole ? 4096 + 8 : 0), dS1 = tdbase + (drole ? 2 * 4096 : 8), dS2 = tdbase + (drole ? 2 * 4096 + 8 : 4096);
  f32x2 dv[3][3];                   // DY' rows (+, -, single) of this thread, each as (v0, v5), (v1, v3), (v2, v4)
  auto dy_transform = [&]() {
    f32x2 wp[2], wm[2], ws[2];
    {
      const f32x2 X = __builtin_elementwise_fma(dcxv, rd20, rd00), Y = __builtin_elementwise_fma(dcy3v, rd30, dcy1v * rd10);
      wp[0] = X + Y, wm[0] = X - Y, ws[0] = drole ? rd30 : rd00;
    }
    {
      const f32x2 X = __builtin_elementwise_fma(dcxv, rd21, rd01), Y = __builtin_elementwise_fma(dcy3v, rd31, dcy1v * rd11);
      wp[1] = X + Y, wm[1] = X - Y, ws[1] = drole ? rd31 : rd01;
    }
    row_pass_d(wp, dv[0]);
    row_pass_d(wm, dv[1]);
    row_pass_d(ws, dv[2]);
  };

  f32x4 acc[36];
#pragma unroll
  for (int f = 0; f < 36; ++f) acc[f] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 qa[2], qb[2];
  const unsigned fa = sbase + kD0 + wn * 1024 + lane * 16, fb = sbase + kV0 + wc * 1024 + lane * 16;

  // ---- prologue: raw data of stages 0 and 1, operands of stage 0
  StageAt s1 = decode(s_begin);     // the stage whose dy is in rd / whose patch the x transform reads next
  StageAt s2 = s1;                  // the stage whose raw data is requested next
  issue_x(s2, 0);
  dy_rsrc(s2);
  G4_LOAD_DY_ALL();
  advance(s2);
  issue_x(s2, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  {
    mask_dy(s1);
    dy_transform();
    const f32x2* P = nullptr;
    (void)P;
    auto px2 = [&](unsigned ro, int k) {      // columns (2 k, 2 k + 1) of a patch row, buffer 0
      const int oa = k == 0 ? O0 : k == 1 ? O2 : O0 + 1, ob = k == 0 ? O1 : k == 1 ? O3 : O1 + 1;
      const char* q = smem + (tprow - sbase) + ro;
      return (f32x2){*reinterpret_cast<const float*>(q + oa * 64), *reinterpret_cast<const float*>(q + ob * 64)};
    };
    f32x2 wa[3], wb[3], va[3], vb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const f32x2 xx = __builtin_elementwise_fma(ca1v, px2(ro1, k), ca2v * px2(ro2, k));
      const f32x2 yy = __builtin_elementwise_fma(cb4v, px2(ro4, k), cb3v * px2(ro3, k));
      wa[k] = xx + yy;
      wb[k] = xx - yy;
    }
    row_pass_x(wa, va);
    row_pass_x(wb, vb);
    *reinterpret_cast<f32x4*>(smem + (wA - sbase)) = (f32x4){va[0].x, va[0].y, va[1].x, va[1].y};
    *reinterpret_cast<f32x2*>(smem + (wB - sbase)) = va[2];
    *reinterpret_cast<f32x2*>(smem + (wC - sbase)) = vb[0];
    *reinterpret_cast<f32x4*>(smem + (wD - sbase)) = (f32x4){vb[1].x, vb[1].y, vb[2].x, vb[2].y};
    *reinterpret_cast<f32x4*>(smem + (dP - sbase)) = (f32x4){dv[0][0].x, dv[0][0].y, dv[0][1].x, dv[0][1].y};
    *reinterpret_cast<f32x4*>(smem + (dP - sbase) + 4096) = (f32x4){dv[0][2].x, dv[0][2].y, dv[1][0].x, dv[1][0].y};
    *reinterpret_cast<f32x4*>(smem + (dP - sbase) + 8192) = (f32x4){dv[1][1].x, dv[1][1].y, dv[1][2].x, dv[1][2].y};
    *reinterpret_cast<f32x2*>(smem + (dS0 - sbase)) = dv[2][0];
    *reinterpret_cast<f32x2*>(smem + (dS1 - sbase)) = dv[2][1];
    *reinterpret_cast<f32x2*>(smem + (dS2 - sbase)) = dv[2][2];
  }
  advance(s1);                      // s1 = stage s_begin + 1: its patch is in X[1], its dy comes now
  dy_rsrc(s1);
  G4_LOAD_DY_ALL();
  advance(s2);                      // s2 = stage s_begin + 2
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  asm volatile("ds_read_b128 %0, %1" : "=v"(qa[0]) : "v"(fa));
  asm volatile("ds_read_b128 %0, %1" : "=v"(qb[0]) : "v"(fb));

  // One stage = 9 steps of (2 fragment loads one step ahead, 4 MFMAs) on DY' / V of parity CB.  Riding along: the two transforms
  // of stage + 1 -> buffers CB ^ 1 (its patch sits in X[CB ^ 1], its dy in rd), the raw data of stage + 2 (patch -> X[CB], dy ->
  // rd once the transform has read it), the stage barrier before the last step's MFMAs (see conv2d_f32_wino4.hip).
  auto stage = [&](auto cb_c) {
    constexpr int CB = decltype(cb_c)::value;
    const unsigned pa = fa + CB * kDBytes, pb = fb + CB * kVBytes;
    const unsigned pan = fa + (CB ^ 1) * kDBytes, pbn = fb + (CB ^ 1) * kVBytes;
    const unsigned poff = (CB ^ 1) * kXBytes;
    const unsigned dd = (CB ^ 1) * kDBytes;
    f32x2 r1[3], r2[3], r3[3], r4[3];
    f32x2 wa[3], wb[3], va[3], vb[3];
    static_for<9>([&](auto fq_c) {
      constexpr int fq = decltype(fq_c)::value;
      constexpr int cur = (fq + CB) & 1, nxt = cur ^ 1;
      if constexpr (fq < 8) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qa[nxt]) : "v"(pa), "n"((fq + 1) * 4096));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qb[nxt]) : "v"(pb), "n"((fq + 1) * 2048));
      }
      // ---- LDS operations of the transforms in this step (kStepOps counts them)
      if constexpr (fq == 0) G4_ROWREAD(r1, ro1 + poff);
      if constexpr (fq == 1) G4_ROWREAD(r2, ro2 + poff);
      if constexpr (fq == 2) {
        G4_ROWREAD(r3, ro3 + poff);
        asm volatile("ds_write_b128 %0, %1" ::"v"(dP + dd), "v"((f32x4){dv[0][0].x, dv[0][0].y, dv[0][1].x, dv[0][1].y}) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(dP + dd), "v"((f32x4){dv[0][2].x, dv[0][2].y, dv[1][0].x, dv[1][0].y}) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(dP + dd), "v"((f32x4){dv[1][1].x, dv[1][1].y, dv[1][2].x, dv[1][2].y}) : "memory");
      }
      if constexpr (fq == 3) G4_ROWREAD(r4, ro4 + poff);
      if constexpr (fq == 4) {
        asm volatile("ds_write_b64 %0, %1" ::"v"(dS0 + dd), "v"(dv[2][0]) : "memory");
        asm volatile("ds_write_b64 %0, %1" ::"v"(dS1 + dd), "v"(dv[2][1]) : "memory");
        asm volatile("ds_write_b64 %0, %1" ::"v"(dS2 + dd), "v"(dv[2][2]) : "memory");
      }
      if constexpr (fq == 7) {
        asm volatile("ds_write_b128 %0, %1" ::"v"(wA + (CB ? 0 : vdelta01)), "v"((f32x4){va[0].x, va[0].y, va[1].x, va[1].y}) : "memory");
        asm volatile("ds_write_b64 %0, %1" ::"v"(wB + (CB ? 0 : vdelta01)), "v"(va[2]) : "memory");
        asm volatile("ds_write_b64 %0, %1" ::"v"(wC + (CB ? 0 : vdelta23)), "v"(vb[0]) : "memory");
        asm volatile("ds_write_b128 %0, %1" ::"v"(wD + (CB ? 0 : vdelta23)), "v"((f32x4){vb[1].x, vb[1].y, vb[2].x, vb[2].y}) : "memory");
      }
      // ---- wait for this step's fragments (a patch row requested in step i has arrived by the wait of step i + 2)
      if constexpr (fq == 8) {
        // fragments of step 8, the DY' / V writes and this wave's patch pieces are done; the 16 dy dwords issued after the
        // pieces (steps 1-4) may stay in flight
        asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" : "+v"(qa[cur]), "+v"(qb[cur])::"memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("ds_read_b128 %0, %1" : "=v"(qa[nxt]) : "v"(pan));
        asm volatile("ds_read_b128 %0, %1" : "=v"(qb[nxt]) : "v"(pbn));
      } else {
        constexpr int WN = g4_wait(fq);
        if constexpr (fq == 3)
          asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r2[0]), "+v"(r2[1]), "+v"(r2[2]) : "n"(WN));
        else if constexpr (fq == 4)
          asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(r3[0]), "+v"(r3[1]), "+v"(r3[2]) : "n"(WN));
        else if constexpr (fq == 5)
          asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(qa[cur]), "+v"(qb[cur]), "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]) : "n"(WN));
        else
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(qa[cur]), "+v"(qb[cur]) : "n"(WN));
      }
      // ---- the step's four MFMAs
#pragma unroll
      for (int p = 0; p < 4; ++p)
        acc[4 * fq + p] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[cur][p], qb[cur][p], acc[4 * fq + p], 0, 0, 0);
      // ---- raw data and arithmetic
      if constexpr (fq == 0) {
        issue_x(s2, CB);            // patch of stage + 2 (X[CB] was read during the previous stage)
        // the dy of stage + 1 (requested a stage ago; only the two patch pieces are younger) -> DY' rows in registers
        asm volatile("s_waitcnt vmcnt(2)"
                     : "+v"(rd00), "+v"(rd01), "+v"(rd10), "+v"(rd11), "+v"(rd20), "+v"(rd21), "+v"(rd30), "+v"(rd31)
                     :: "memory");
        mask_dy(s1);
        dy_transform();
        s1 = s2;                    // (s2 is one stage ahead of s1 until step 6 moves it on)
        dy_rsrc(s1);
      }
      // dy of stage + 2 (the registers are free: the transform has read them)
      if constexpr (fq == 1) {
        G4_LOAD_DY(0, rd00, rd01);
        G4_LOAD_DY(1, rd10, rd11);
      }
      if constexpr (fq == 2) {
        G4_LOAD_DY(2, rd20, rd21);
        G4_LOAD_DY(3, rd30, rd31);
      }
      if constexpr (fq == 3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) wa[k] = __builtin_elementwise_fma(ca1v, r1[k], ca2v * r2[k]);          // X
      }
      if constexpr (fq == 4) {
#pragma unroll
        for (int k = 0; k < 3; ++k) wb[k] = cb3v * r3[k];
      }
      if constexpr (fq == 5) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const f32x2 yy = __builtin_elementwise_fma(cb4v, r4[k], wb[k]);                                  // Y
          wb[k] = wa[k] - yy;
          wa[k] = wa[k] + yy;
        }
      }
      if constexpr (fq == 6) {
        row_pass_x(wa, va);
        row_pass_x(wb, vb);
        advance(s2);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
#pragma unroll 1
  for (int k = 0; k < nst; k += 2) {
    stage(IC<0>{});
    stage(IC<1>{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(qa[0]), "+v"(qb[0]), "+v"(qa[1]), "+v"(qb[1])::"memory");
#undef G4_ROWREAD
#undef G4_LOAD_DY
#undef G4_LOAD_DY_ALL
  // partial[slice][n block][c block][f'][c 32][n 64]: lane holds n = 16 wn + 4 g + r (r = 0..3: one 16-byte store), c = 16 wc + li
  float* dst = partial + ((((size_t)blockIdx.x * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * 36) * (kCB * kNB) +
               (16 * wc + li) * kNB + 16 * wn + 4 * g;
#pragma unroll
  for (int f = 0; f < 36; ++f) *reinterpret_cast<f32x4*>(dst + (size_t)f * (kCB * kNB)) = acc[f];
}

// partial[0] += partial[1] + ... + partial[nslices - 1], in that order (deterministic), one 16-byte piece per thread
__global__ __launch_bounds__(256) void k_wino4_wgrad_sum(float* __restrict__ partial, int nslices, size_t sstride4, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4* p = reinterpret_cast<float4*>(partial) + i;
  float4 s = p[0];
  int sl = 1;
  for (; sl + 3 < nslices; sl += 4) {      // four independent loads in flight, added in slice order
    const float4 a = p[(size_t)sl * sstride4], b = p[(size_t)(sl + 1) * sstride4], c = p[(size_t)(sl + 2) * sstride4],
                 d = p[(size_t)(sl + 3) * sstride4];
    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
    s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
  }
  for (; sl < nslices; ++sl) {
    const float4 a = p[(size_t)sl * sstride4];
    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
  }
  p[0] = s;
}

// dw[n][ky][kx][c] = sum_{i, j} G[i][ky] G[j][kx] * (sum over slices, in order, of partial[slice][.][.][f'(i, j)][c][n]);
// block = (n block, c block), thread = (c, 4 consecutive n)
__global__ __launch_bounds__(512) void k_wino4_wgrad_finish(const float* __restrict__ partial, int nslices, int nb, int cb, int Cout,
                                                            int Cin, float* __restrict__ dw) {
  const int n4 = (threadIdx.x & 15) * 4, c = threadIdx.x >> 4;
  const size_t blk = (size_t)blockIdx.x * cb + blockIdx.y;
  const size_t sstride = (size_t)nb * cb * 36 * (kCB * kNB);
  const float* src = partial + blk * 36 * (kCB * kNB) + c * kNB + n4;
  const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f}, {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                         {1.f / 24.f, 1.f / 12.f, 1.f / 6.f}, {1.f / 24.f, -1.f / 12.f, 1.f / 6.f}, {0.f, 0.f, 1.f}};
  float4 t[6][3];                    // sum_j dU[i][j] G[j][kx]
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) t[i][kx] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int sl = 0; sl < nslices; ++sl) {
        const float4 v = ud_ldg_stream(src + sl * sstride + (size_t)g4_f(i, j) * (kCB * kNB));
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        t[i][kx].x += G[j][kx] * s.x; t[i][kx].y += G[j][kx] * s.y; t[i][kx].z += G[j][kx] * s.z; t[i][kx].w += G[j][kx] * s.w;
      }
    }
  }
  const int n = blockIdx.x * kNB + n4, cc = blockIdx.y * kCB + c;
  if (cc >= Cin) return;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        o.x += G[i][ky] * t[i][kx].x; o.y += G[i][ky] * t[i][kx].y; o.z += G[i][ky] * t[i][kx].z; o.w += G[i][ky] * t[i][kx].w;
      }
      const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < Cout) dw[(((size_t)(n + e) * 3 + ky) * 3 + kx) * Cin + cc] = ov[e];
    }
}

struct G4Plan {
  int TY, SX, nstages, sps, nslices, nb, cb;
};
G4Plan g4_plan(int B, int H, int W, int Cin, int Cout) {
  G4Plan p;
  p.TY = (H + 3) / 4;
  p.SX = ((W + 3) / 4 + kTS - 1) / kTS;
  p.nstages = B * p.TY * p.SX;
  p.nb = ud_div_up(Cout, kNB), p.cb = ud_div_up(Cin, kCB);
  int slices = 256 / (p.nb * p.cb);
  if (slices < 1) slices = 1;
  if (slices > (p.nstages + 1) / 2) slices = (p.nstages + 1) / 2;
  p.sps = ud_div_up(p.nstages, slices);
  p.sps += p.sps & 1;                                  // stages come in pairs
  p.nslices = ud_div_up(p.nstages, p.sps);
  return p;
}

}  // namespace

extern "C" size_t ud_conv3x3_wino4_wgrad_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const G4Plan p = g4_plan(B, H, W, Cin, Cout);
  return (size_t)p.nslices * p.nb * p.cb * 36 * kCB * kNB * sizeof(float);
}

// x [B][H][W][Cin], dy [B][H][W][Cout] -> dw [Cout][3][3][Cin]  (same contract as ud_conv3x3_wgrad_nhwc_f32).
// Cin % 32 == 0, Cout % 64 == 0, tensors below 2 GB -- else UD_ERR_UNSUPPORTED (the caller uses the F(2x2) / direct kernels).
extern "C" int ud_conv3x3_wino4_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin,
                                               int Cout, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Cin % kCB != 0 || Cout % kNB != 0) return UD_ERR_UNSUPPORTED;
  if ((long long)B * H * W * Cin * 4 >= 0x7fffffffll || (long long)B * H * W * Cout * 4 >= 0x7fffffffll) return UD_ERR_UNSUPPORTED;
  const G4Plan p = g4_plan(B, H, W, Cin, Cout);
  if (workspace_bytes < ud_conv3x3_wino4_wgrad_f32_workspace_bytes(B, H, W, Cin, Cout)) return UD_ERR_WORKSPACE;
  if (p.nb > 65535 || p.cb > 65535) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wino4_wgrad_f32, hipFuncAttributeMaxDynamicSharedMemorySize, kSmem));
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("conv2d.k_wgrad_wino4_f32", stream);
  float* partial = static_cast<float*>(workspace);
  G4Geom gm{B, H, W, Cin, Cout, p.TY, p.SX, p.nstages, p.sps};
  k_wino4_wgrad_f32<<<dim3(p.nslices, p.nb, p.cb), 512, kSmem, stream>>>(x, dy, partial, gm);
  UD_LAUNCH_CHECK();
  if (p.nslices > 1) {
    const size_t n4 = (size_t)p.nb * p.cb * 36 * kCB * kNB / 4;
    k_wino4_wgrad_sum<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>(partial, p.nslices, n4, n4);
    UD_LAUNCH_CHECK();
  }
  k_wino4_wgrad_finish<<<dim3(p.nb, p.cb), 512, 0, stream>>>(partial, 1, p.nb, p.cb, Cout, Cin, dw);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
