// fp32 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) on v_mfma_f32_16x16x4_f32: 16 multiplications per
// 2 x 2 outputs per (cin, cout) instead of 36 -- 2.25x fewer MFMA flops for the trunk / head / ResNet 3x3 layers of the
// fp32 (headline) step (base_bev_backbone.py:30-110, center_head.py:311-420, lss_fpn.py:143-149), which sit at the fp32
// MFMA ceiling as direct convolutions (profiles/r03_conv_f32.md).  Forward and data gradient (the weight transform takes
// the transposed, tap-reversed view for the latter).
//
//   V[f][tile][c] = (B^T d B)[f]     d = 4 x 4 input patch of a 2 x 2 output tile          (in-kernel, per 8-channel stage)
//   U[f][n][c]    = (G g G^T)[f]     g = 3 x 3 filter                                      (k_wino_weights, once per weight version)
//   M[f][tile][n] = sum_c V[f][tile][c] U[f][n][c]                                         (16 independent GEMMs on the MFMA pipe)
//   y(2 x 2)      = A^T M A                                                                (in registers, then the usual epilogue)
//
// A workgroup (8 waves) owns 64 tiles (TWB x THB, e.g. 9 x 7 -> 18 x 14 output pixels) x 64 output channels: wave (wq, wh)
// holds the 16 frequencies of 32 tiles x 16 channels (128 accumulator registers), so the output transform never leaves the
// lane.  Per 8-input-channel stage: the raw (2 THB + 2) x (2 TWB + 2) x 8-channel patch and the 32 KB U stage arrive by
// LDS-DMA (U is stored in exactly the LDS order: [cout block][stage][f][64][8]), every thread transforms one (tile, channel)
// patch into V, and the MFMA loop reads 8-byte fragments (2 reduction steps per read).  Patch, V and U are double-buffered:
// 150 KB of LDS, one workgroup per CU, two waves per SIMD.  The whole stage is hand-scheduled: fragment reads one step ahead of
// their MFMAs, the next stage's input transform spread over the steps, the stage barrier one step before the stage's end.
#include "ud_common.h"
#include "ud_prof.h"
#include <cstdlib>
#include <type_traits>
#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kNT = 64, kTN = 64, kKC = 8;
constexpr int kUBytes = 16 * kTN * kKC * 4, kVBytes = 16 * kNT * kKC * 4, kPBytes = 11264;
constexpr int kUOff = 0, kVOff = 2 * kUBytes, kPOff = kVOff + 2 * kVBytes, kWinoSmem = kPOff + 2 * kPBytes;

struct WinoGeom {
  int B, H, W, Cin, Cout, bx, by;   // bx x by tile blocks per image
  int n_full;                       // work items [0, n_full) are whole units (tile block, cout block); the rest quarter units
  int n_items;                      // n_full + 4 x (quarter-split units): a workgroup walks items blockIdx.x, + gridDim.x, ...
};
struct WinoEp {
  const float* bias;
  const float* scale;               // folded eval-mode BatchNorm: v * scale + shift after the bias (both or neither)
  const float* shift;
  const float* residual;
  int relu;
  float* stats;                     // [blocks][Cout][2] per-workgroup (sum, sum of squares) of the stored outputs, or nullptr
};

__device__ __attribute__((aligned(16))) unsigned int g_zero16w[4];

__device__ __forceinline__ void dma16w(const float* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// U[nb][cc][f][nl][cl] from g'[n][ky][kx][c] = w[n * s_n + c * s_c + ky' * s_y + kx' * s_x] (ky' = flip ? 2 - ky : ky): the forward
// transform takes (n, c) = (Cout, Cin) of the parameter, the data gradient (n, c) = (Cin, Cout) with flip = 1.  Rows / channels
// past N / C are zero.
__global__ void k_wino_weights(const float* __restrict__ w, long long s_n, long long s_c, long long s_y, long long s_x, int N,
                               int C, int flip, float* __restrict__ U, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cl = (int)(i & 7), nl = (int)((i >> 3) & 63);
  const long long blk = i >> 9;                        // nb * nch + cc
  const int nch = (C + 7) / 8;
  const int nb = (int)(blk / nch), cc = (int)(blk - (long long)nb * nch);
  const int n = nb * 64 + nl, c = cc * 8 + cl;
  float g[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      g[ky][kx] = (n < N && c < C) ? w[n * s_n + c * s_c + (flip ? 2 - ky : ky) * s_y + (flip ? 2 - kx : kx) * s_x] : 0.f;
  float t[4][3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    t[0][kx] = g[0][kx];
    t[1][kx] = 0.5f * (g[0][kx] + g[1][kx] + g[2][kx]);
    t[2][kx] = 0.5f * (g[0][kx] - g[1][kx] + g[2][kx]);
    t[3][kx] = g[2][kx];
  }
  float* out = U + blk * (16 * 512) + nl * 8 + cl;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    out[(4 * a + 0) * 512] = t[a][0];
    out[(4 * a + 1) * 512] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
    out[(4 * a + 2) * 512] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
    out[(4 * a + 3) * 512] = t[a][2];
  }
}

template <int TWB, int THB>
__global__ __launch_bounds__(512) void k_conv3x3_wino_f32(const float* __restrict__ x, const float* __restrict__ U,
                                                          float* __restrict__ y, WinoGeom gm, WinoEp ep) {
  constexpr int PW = 2 * TWB + 2, PH = 2 * THB + 2, PP = PW * PH, kPInstr = (PP + 31) / 32;
  static_assert(TWB * THB <= kNT && kPInstr * 1024 <= kPBytes && kPInstr <= 16, "tile block");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wq = wave & 3, wh = wave >> 2;
  const int nblocks = gm.B * gm.bx * gm.by;
  // Unit = (tile block, 64-channel cout block), cout block major.  When the unit count leaves a short last round on the 256 CUs
  // (e.g. 1040 units: 4 rounds + 16 units alone on the chip for a fifth), the launcher turns that remainder into QUARTER units:
  // four workgroups per unit, each multiplying one 16-tile M block (q >= 0) -- a quarter unit costs ~0.4 of a whole one.
  // PERSISTENT workgroups (one per CU): item = blockIdx.x, + gridDim.x, ...; the next item's first stages are DMA-ed while the
  // current one runs its epilogue (a unit of a Cin = 64 layer is 8 stages: launch + first-stage latency + epilogue were 26 % of it).
  int unit, q, cbi, blk_lin, b, ty0, tx0, n0;
  const int nchunks = gm.Cin / kKC;
  const float* zero = reinterpret_cast<const float*>(g_zero16w);
  const float* pp[2];
  int pinc[2];
  const float* up;
  auto setup = [&](int item) {
    unit = item < gm.n_full ? item : gm.n_full + ((item - gm.n_full) >> 2);
    q = item < gm.n_full ? -1 : (item - gm.n_full) & 3;
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
      // patch rows are stored even columns first, then odd ones: the transform's lanes (consecutive tiles, fixed patch column)
      // then read consecutive 32-byte slots instead of every other one (4-way LDS bank conflicts: 18 M conflict cycles per launch)
      const int qq = (wave + 8 * i) * 32 + (lane >> 1);
      const int qy = qq / PW, qs = qq - qy * PW;
      const int qx = qs < PW / 2 ? 2 * qs : 2 * (qs - PW / 2) + 1;
      const int gy = 2 * ty0 + qy - 1, gx = 2 * tx0 + qx - 1;
      const bool ok = qq < PP && gy >= 0 && gy < gm.H && gx >= 0 && gx < gm.W;
      pp[i] = ok ? x + ((size_t)(b * gm.H + gy) * gm.W + gx) * gm.Cin + (lane & 1) * 4 : zero;
      pinc[i] = ok ? kKC : 0;
    }
    up = U + (size_t)cbi * nchunks * (16 * 512) + wave * 1024 + lane * 4;
  };
  auto stage_p = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (wave + 8 * i < kPInstr) dma16w(pp[i], smem + kPOff + buf * kPBytes + (wave + 8 * i) * 1024);
      pp[i] += pinc[i];
    }
  };
  auto stage_u = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16w(up + i * 256, smem + kUOff + buf * kUBytes + (wave * 4 + i) * 1024);
    up += 16 * 512;
  };
  // input transform: thread (tile slot t, channel c)
  const int t = tid >> 3, c = tid & 7;
  const int tyl = min(t / TWB, THB - 1), txl = t % TWB;
  const int pbase = ((2 * tyl) * PW + txl) * 8 + c;      // slot of patch column 2 txl (even columns first, see stage_p)
  auto transform = [&](int buf) {
    const float* P = reinterpret_cast<const float*>(smem + kPOff + buf * kPBytes) + pbase;
    float* V = reinterpret_cast<float*>(smem + kVOff + buf * kVBytes) + t * 8 + c;
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = P[(i * PW + (j & 1) * (PW / 2) + (j >> 1)) * 8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float r0 = d[0][j] - d[2][j], r1 = d[1][j] + d[2][j], r2 = d[2][j] - d[1][j], r3 = d[1][j] - d[3][j];
      d[0][j] = r0; d[1][j] = r1; d[2][j] = r2; d[3][j] = r3;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V[(4 * i + 0) * 512] = d[i][0] - d[i][2];
      V[(4 * i + 1) * 512] = d[i][1] + d[i][2];
      V[(4 * i + 2) * 512] = d[i][2] - d[i][1];
      V[(4 * i + 3) * 512] = d[i][1] - d[i][3];
    }
  };

  f32x4 acc[16][2];
  const unsigned fa = kVOff + ((32 * wh + li) * 8 + 2 * g) * 4, fb = kUOff + ((16 * wq + li) * 8 + 2 * g) * 4;

  // MFMA loop fragments: hand-placed ds_read_b64 (two frequencies = 6 reads per step, one step ahead of the 8 MFMAs that consume
  // them) -- hipcc would otherwise wait for the LDS-DMA issued at the top of the stage (vmcnt(0)) before the first fragment read
  // and serialise read -> wait -> MFMA.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 qa0[2][2], qa1[2][2], qb[2][2];
#define UD_WN_LOADS(BUF, F)                                                                                             \
  do {                                                                                                                  \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                     \
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qa0[BUF][h]) : "v"(pa), "n"(((F) + h) * 2048));               \
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qb[BUF][h]) : "v"(pb), "n"(((F) + h) * 2048));                \
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qa1[BUF][h]) : "v"(pa), "n"(((F) + h) * 2048 + 512));         \
    }                                                                                                                   \
  } while (0)
#define UD_WN_WAIT(BUF, N)                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                              \
               : "+v"(qa0[BUF][0]), "+v"(qa0[BUF][1]), "+v"(qa1[BUF][0]), "+v"(qa1[BUF][1]), "+v"(qb[BUF][0]), "+v"(qb[BUF][1]))

  // The input transform of stage chunk + 1 rides inside the MFMA loop of stage chunk: its 8 patch reads go out with the step-0
  // fragments, the 32 additions run under the MFMAs of steps 2-3, its 8 V writes go out in steps 4-7 (LDS returns in order, so
  // every s_waitcnt below counts exactly the operations issued after the fragments it needs).
  f32x2 td[4][2];
  unsigned tp[4];                        // patch row i of this thread's tile; V column of this thread
#pragma unroll
  for (int i = 0; i < 4; ++i) tp[i] = kPOff + (pbase + i * PW * 8) * 4;
  const unsigned tv = kVOff + (t * 8 + c) * 4;
#define UD_WN_TREADS(POFF)                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                       \
    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:8" : "=v"(td[i][0]) : "v"(tp[i] + (POFF)));   /* columns 0, 2 */ \
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(td[i][1]) : "v"(tp[i] + (POFF)), "n"((PW / 2) * 8), "n"((PW / 2) * 8 + 8)); /* 1, 3 */ \
  }
#define UD_WN_TWAIT(BUF, N)                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                              \
               : "+v"(qa0[BUF][0]), "+v"(qa0[BUF][1]), "+v"(qa1[BUF][0]), "+v"(qa1[BUF][1]), "+v"(qb[BUF][0]), "+v"(qb[BUF][1]), \
                 "+v"(td[0][0]), "+v"(td[0][1]), "+v"(td[1][0]), "+v"(td[1][1]), "+v"(td[2][0]), "+v"(td[2][1]),         \
                 "+v"(td[3][0]), "+v"(td[3][1]))

  auto first_stages = [&]() {      // the DMA an item needs before its first stage (pointers advance as in the stage loop)
    stage_p(0);
    stage_u(0);
    if (nchunks > 1) stage_p(1);
  };
  // One barrier per stage, placed BEFORE the stage's last MFMA step: by then every wave holds its step-7 fragments in registers
  // and has written its share of V(chunk + 1), so right after the barrier the next stage's DMA, first fragment reads and patch
  // reads go out and complete under the 8 MFMAs of step 7 -- no bubble at the stage boundary.
  auto step_mma = [&](int cur, int st) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = 2 * st + h;
      acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa0[cur][h][0], qb[cur][h][0], acc[f][0], 0, 0, 0);
      acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa1[cur][h][0], qb[cur][h][0], acc[f][1], 0, 0, 0);
      acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa0[cur][h][1], qb[cur][h][1], acc[f][0], 0, 0, 0);
      acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa1[cur][h][1], qb[cur][h][1], acc[f][1], 0, 0, 0);
    }
  };
  auto stage_mma = [&](auto more_c, auto first_c, int chunk) {
    constexpr bool more = decltype(more_c)::value, first = decltype(first_c)::value;
    const int cb = chunk & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();      // V(chunk) written, U(chunk) and patch(chunk + 1) landed; everybody holds the last fragments of chunk - 1
    const unsigned pa = fa + cb * kVBytes, pb = fb + cb * kUBytes;
    const unsigned tvw = tv + (cb ^ 1) * kVBytes;
    float tw[16];
    UD_WN_LOADS(0, 0);
    if constexpr (more) { UD_WN_TREADS((cb ^ 1) * kPBytes); }
    // the next stage's DMA issue is woven between the 8 MFMAs of the previous stage's last step (fragments already in registers)
    __builtin_amdgcn_sched_barrier(0);
    if (more) stage_u(cb ^ 1);
    if (chunk + 2 < nchunks) stage_p(cb);
    if constexpr (!first) {
      step_mma(1, 7);
#pragma unroll
      for (int i_ = 0; i_ < 8; ++i_) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 7; ++st) {
      const int cur = st & 1;
      if (more && st >= 3) {              // two V rows (four frequencies) per step
        const int f0 = 4 * (st - 3);
        asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(tvw), "v"(tw[f0]), "v"(tw[f0 + 1]), "n"(8 * f0), "n"(8 * f0 + 8));
        asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(tvw), "v"(tw[f0 + 2]), "v"(tw[f0 + 3]), "n"(8 * f0 + 16), "n"(8 * f0 + 24));
      }
      if (cur == 0) UD_WN_LOADS(1, 2 * (st + 1)); else UD_WN_LOADS(0, 2 * (st + 1));
      // operations issued after this step's fragments: the next step's 6 reads, plus the transform's reads (step 0) / writes (3-6)
      if (more) {
        if (st == 0) UD_WN_WAIT(0, 14);
        else if (st == 1) UD_WN_WAIT(1, 6);
        else if (st == 2) UD_WN_TWAIT(0, 6);
        else if (cur == 0) UD_WN_WAIT(0, 8);
        else UD_WN_WAIT(1, 8);
      } else {
        if (cur == 0) UD_WN_WAIT(0, 6); else UD_WN_WAIT(1, 6);
      }
      if (st == 2 && more) {
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[i][0] = td[i][0][0]; d[i][2] = td[i][0][1]; d[i][1] = td[i][1][0]; d[i][3] = td[i][1][1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float r0 = d[0][j] - d[2][j], r1 = d[1][j] + d[2][j], r2 = d[2][j] - d[1][j], r3 = d[1][j] - d[3][j];
          d[0][j] = r0; d[1][j] = r1; d[2][j] = r2; d[3][j] = r3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          tw[4 * i + 0] = d[i][0] - d[i][2];
          tw[4 * i + 1] = d[i][1] + d[i][2];
          tw[4 * i + 2] = d[i][2] - d[i][1];
          tw[4 * i + 3] = d[i][1] - d[i][3];
        }
      }
      step_mma(cur, st);
      if (st == 2 && more) {              // weave the transform's 32 additions between this step's 8 MFMAs
#pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
        }
      }
    }
  };
  int item = blockIdx.x;
  setup(item);
  first_stages();
#pragma unroll 1
  for (;;) {
#pragma unroll
  for (int f = 0; f < 16; ++f) acc[f][0] = acc[f][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA (and the previous item's stores) are done
  __syncthreads();                                       // first stages in LDS; everybody has left the previous item's epilogue
  if (q < 0) {
    transform(0);
    if (nchunks == 1) {
      stage_mma(std::false_type{}, std::true_type{}, 0);
    } else {
      stage_mma(std::true_type{}, std::true_type{}, 0);
#pragma unroll 1
      for (int chunk = 1; chunk + 1 < nchunks; ++chunk) stage_mma(std::true_type{}, std::false_type{}, chunk);
      stage_mma(std::false_type{}, std::false_type{}, nchunks - 1);
    }
    UD_WN_WAIT(1, 0);
    step_mma(1, 7);
  } else {
    // quarter unit: tile slots [16 q, 16 q + 16).  Waves 2q, 2q + 1 own those slots' input transform, the four waves of half
    // q >> 1 multiply M block q & 1 into acc[.][0]; everybody stages U / the patch and meets at the stage barrier.  Plain
    // sequence per stage (transform, then the MFMA steps with their fragments one step ahead).
    const bool do_tr = (wave >> 1) == q, do_mma = wh == (q >> 1);
    if (do_tr) transform(0);
    const unsigned faq = fa + (q & 1) * 512;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const int cb = chunk & 1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
      const bool more = chunk + 1 < nchunks;
      if (more) stage_u(cb ^ 1);
      if (chunk + 2 < nchunks) stage_p(cb);
      if (do_tr && more) {
        const unsigned tvw = tv + (cb ^ 1) * kVBytes;
        UD_WN_TREADS((cb ^ 1) * kPBytes);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(td[0][0]), "+v"(td[0][1]), "+v"(td[1][0]), "+v"(td[1][1]), "+v"(td[2][0]), "+v"(td[2][1]),
                       "+v"(td[3][0]), "+v"(td[3][1]));
        float d[4][4], tw[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[i][0] = td[i][0][0]; d[i][2] = td[i][0][1]; d[i][1] = td[i][1][0]; d[i][3] = td[i][1][1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float r0 = d[0][j] - d[2][j], r1 = d[1][j] + d[2][j], r2 = d[2][j] - d[1][j], r3 = d[1][j] - d[3][j];
          d[0][j] = r0; d[1][j] = r1; d[2][j] = r2; d[3][j] = r3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          tw[4 * i + 0] = d[i][0] - d[i][2];
          tw[4 * i + 1] = d[i][1] + d[i][2];
          tw[4 * i + 2] = d[i][2] - d[i][1];
          tw[4 * i + 3] = d[i][1] - d[i][3];
        }
#pragma unroll
        for (int f0 = 0; f0 < 16; f0 += 2)
          asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(tvw), "v"(tw[f0]), "v"(tw[f0 + 1]), "n"(8 * f0), "n"(8 * f0 + 8));
      }
      if (do_mma) {
        const unsigned pa = faq + cb * kVBytes, pb = fb + cb * kUBytes;
#define UD_WQ_LOADS(BUF, F)                                                                                             \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                                       \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qa0[BUF][h]) : "v"(pa), "n"(((F) + h) * 2048));                 \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qb[BUF][h]) : "v"(pb), "n"(((F) + h) * 2048));                  \
  }
#define UD_WQ_WAIT(BUF, N) \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(qa0[BUF][0]), "+v"(qa0[BUF][1]), "+v"(qb[BUF][0]), "+v"(qb[BUF][1]))
        UD_WQ_LOADS(0, 0);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const int cur = st & 1;
          if (st + 1 < 8) {
            if (cur == 0) { UD_WQ_LOADS(1, 2 * (st + 1)); UD_WQ_WAIT(0, 4); } else { UD_WQ_LOADS(0, 2 * (st + 1)); UD_WQ_WAIT(1, 4); }
          } else {
            UD_WQ_WAIT(1, 0);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int f = 2 * st + h;
            acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa0[cur][h][0], qb[cur][h][0], acc[f][0], 0, 0, 0);
            acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa0[cur][h][1], qb[cur][h][1], acc[f][0], 0, 0, 0);
          }
        }
#undef UD_WQ_LOADS
#undef UD_WQ_WAIT
      }
    }
  }
  __syncthreads();                  // everybody is done with U / V / the patches of this item
  // this item's epilogue identity, then the NEXT item's first stages go out (patch buffers and U0 are free; the output tile
  // below lives in the V region) and land during the epilogue
  const int e_q = q, e_b = b, e_ty0 = ty0, e_tx0 = tx0, e_n0 = n0, e_unit = unit, e_blk = blk_lin;
  const int next = item + gridDim.x;
  if (next < gm.n_items) {
    setup(next);
    first_stages();
  }
  // output transform in registers -> fp32 tile in LDS: row = slot * 4 + a * 2 + b, 64 channels per row, the 16-channel group
  // XOR-ed with the row's lane group (four lane groups write rows 16 apart = the same banks otherwise)
  float* Os = reinterpret_cast<float*>(smem + kVOff);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (e_q >= 0 && (mb == 1 || wh != (e_q >> 1))) continue;      // quarter unit: one M block, held in acc[.][0]
      float s[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[0][j] = acc[0 + j][mb][r] + acc[4 + j][mb][r] + acc[8 + j][mb][r];
        s[1][j] = acc[4 + j][mb][r] - acc[8 + j][mb][r] - acc[12 + j][mb][r];
      }
      const int slot = 32 * wh + 16 * (e_q >= 0 ? (e_q & 1) : mb) + 4 * g + r;
      float* o = Os + (slot * 4) * 64 + (16 * (wq ^ g) + li);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        o[(a * 2 + 0) * 64] = s[a][0] + s[a][1] + s[a][2];
        o[(a * 2 + 1) * 64] = s[a][1] - s[a][2] - s[a][3];
      }
    }
  __syncthreads();
  const int c4 = (tid & 15) * 4, n = e_n0 + c4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), scv = make_float4(1.f, 1.f, 1.f, 1.f), shv = bv;
  if (ep.bias && n < gm.Cout) bv = *reinterpret_cast<const float4*>(ep.bias + n);
  if (ep.scale && n < gm.Cout) {
    scv = *reinterpret_cast<const float4*>(ep.scale + n);
    shv = *reinterpret_cast<const float4*>(ep.shift + n);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int row = (tid >> 4) + 32 * k;
    const int slot = row >> 2;
    const int sy = slot / TWB, sx = slot - sy * TWB;
    const int gy = 2 * (e_ty0 + sy) + ((row >> 1) & 1), gx = 2 * (e_tx0 + sx) + (row & 1);
    if (slot >= TWB * THB || gy >= gm.H || gx >= gm.W || n >= gm.Cout || (e_q >= 0 && (slot >> 4) != e_q)) continue;
    float4 v = *reinterpret_cast<const float4*>(Os + row * 64 + (c4 ^ (16 * ((row >> 4) & 3))));
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
  if (ep.stats) {     // a thread keeps ONE 4-channel piece over its 8 rows: reduce the 32 row groups through LDS, fixed order
    __syncthreads();
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
      const float a = (float)ad, qq = (float)qd;
      // whole units fill row `tile block`; quarter units rows past the tile blocks (zeroed by the launcher for the other cout
      // blocks), and quarter 0 clears the unit's own row
      const size_t row = e_q < 0 ? (size_t)e_blk : (size_t)nblocks + 4 * (size_t)(e_unit - gm.n_full) + e_q;
      ep.stats[(row * gm.Cout + e_n0 + tid) * 2] = a;
      ep.stats[(row * gm.Cout + e_n0 + tid) * 2 + 1] = qq;
      if (e_q == 0) ep.stats[((size_t)e_blk * gm.Cout + e_n0 + tid) * 2] = ep.stats[((size_t)e_blk * gm.Cout + e_n0 + tid) * 2 + 1] = 0.f;
    }
  }
  if (next >= gm.n_items) break;
  item = next;
  }
}

#undef UD_WN_TREADS
#undef UD_WN_TWAIT
#undef UD_WN_LOADS
#undef UD_WN_WAIT

struct WinoPlan {
  int twb, thb, bx, by;
};
// units of the short last round that are issued as quarter units (0: none)
int wino_split(long long units) {
  static const int off = getenv("UD_WINO_NO_SPLIT") ? 1 : 0;
  const int rem = (int)(units % 256);
  return (!off && units > 256 && rem > 0 && rem <= 64) ? rem : 0;
}
// tile-block shape with the fewest padded slots for this map
WinoPlan wino_plan(int H, int W) {
  static const int shapes[][2] = {{8, 8}, {9, 7}, {10, 6}, {4, 16}, {16, 4}, {11, 4}};
  static const int force = getenv("UD_WINO_SHAPE") ? atoi(getenv("UD_WINO_SHAPE")) : -1;
  const int TX = (W + 1) / 2, TY = (H + 1) / 2;
  WinoPlan best{};
  long long cost = -1;
  for (int i = 0; i < 6; ++i) {
    if (force >= 0 && i != force) continue;
    const int bx = ud_div_up(TX, shapes[i][0]), by = ud_div_up(TY, shapes[i][1]);
    const long long cst = (long long)bx * by;
    if (cost < 0 || cst < cost) cost = cst, best = WinoPlan{shapes[i][0], shapes[i][1], bx, by};
  }
  return best;
}

}  // namespace

extern "C" size_t ud_conv3x3_wino_f32_weight_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0) return 0;
  return (size_t)ud_div_up(Cout, 64) * ud_div_up(Cin, 8) * 16 * 512 * sizeof(float);
}

// tile blocks per image of the plan for an H x W map (each 64 tile slots of 2 x 2 outputs): callers compare with
// ceil(H / 2) * ceil(W / 2) to decide whether the map fills the blocks well enough
extern "C" int ud_conv3x3_wino_f32_blocks(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const WinoPlan p = wino_plan(H, W);
  return p.bx * p.by;
}

extern "C" size_t ud_conv3x3_wino_bnstats_bytes(int B, int H, int W, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const WinoPlan p = wino_plan(H, W);
  return ((size_t)B * p.bx * p.by + 4 * 64) * Cout * 2 * sizeof(float);     // + the rows of up to 64 quarter-split units
}

extern "C" int ud_conv3x3_wino_f32_weights(const float* w, int64_t s_n, int64_t s_c, int64_t s_y, int64_t s_x, int N, int C,
                                           int flip, float* U, ud_stream_t stream_) {
  if (!w || !U || N <= 0 || C <= 0) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("conv2d.k_wino_weights", stream);
  const long long total = (long long)ud_div_up(N, 64) * ud_div_up(C, 8) * 512;
  k_wino_weights<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(w, s_n, s_c, s_y, s_x, N, C, flip, U, total);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// y = conv3x3(x) (+ bias) (* scale + shift: a folded eval-mode BatchNorm) (+ residual) (ReLU if flags & 1) with U from ud_conv3x3_wino_f32_weights(N = Cout, C = Cin); partial != nullptr:
// also the per-workgroup BatchNorm partial sums ([*slices][Cout][2], same contract as ud_conv3x3_bnstats_nhwc_f32).
extern "C" int ud_conv3x3_wino_nhwc_f32(const float* x, const float* U, float* y, int B, int H, int W, int Cin, int Cout,
                                        const float* bias, const float* scale, const float* shift, const float* residual,
                                        int flags, float* partial, size_t partial_bytes, int* slices, ud_stream_t stream_) {
  if (!x || !U || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return UD_ERR_INVALID_ARG;
  if (Cin % kKC != 0 || Cout % 4 != 0) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const WinoPlan p = wino_plan(H, W);
  const int nblocks = B * p.bx * p.by;
  const long long units = (long long)nblocks * ud_div_up(Cout, kTN);
  if (units + 3 * 64 > 0x7fffffffll) return UD_ERR_UNSUPPORTED;
  const int split = wino_split(units);
  WinoGeom gm{B, H, W, Cin, Cout, p.bx, p.by, (int)units - split, (int)units + 3 * split};
  WinoEp ep{bias, scale, shift, residual, flags & 1, partial};
  if (partial) {
    const size_t rows = (size_t)nblocks + 4 * (size_t)split;
    if (!slices || partial_bytes < rows * Cout * 2 * sizeof(float)) return UD_ERR_WORKSPACE;
    *slices = (int)rows;
    if (split) ud_zero_f32_async(partial + (size_t)nblocks * Cout * 2, 4 * (size_t)split * Cout * 2, stream);
  }
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
#define UD_WINO_ATTR(A, Bq) \
  UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv3x3_wino_f32<A, Bq>, hipFuncAttributeMaxDynamicSharedMemorySize, kWinoSmem))
    UD_WINO_ATTR(8, 8); UD_WINO_ATTR(9, 7); UD_WINO_ATTR(10, 6); UD_WINO_ATTR(4, 16); UD_WINO_ATTR(16, 4); UD_WINO_ATTR(11, 4);
#undef UD_WINO_ATTR
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("conv2d.k_conv3x3_wino_f32", stream);
  static const int persist = getenv("UD_WINO_GRID") ? atoi(getenv("UD_WINO_GRID")) : 256;      // one workgroup per CU
  const dim3 grid((unsigned)std::min<long long>(units + 3 * split, persist > 0 ? persist : units + 3 * split));
#define UD_WINO_LAUNCH(A, Bq) k_conv3x3_wino_f32<A, Bq><<<grid, 512, kWinoSmem, stream>>>(x, U, y, gm, ep)
  if (p.twb == 8) UD_WINO_LAUNCH(8, 8);
  else if (p.twb == 9) UD_WINO_LAUNCH(9, 7);
  else if (p.twb == 10) UD_WINO_LAUNCH(10, 6);
  else if (p.twb == 4) UD_WINO_LAUNCH(4, 16);
  else if (p.twb == 11) UD_WINO_LAUNCH(11, 4);
  else UD_WINO_LAUNCH(16, 4);
#undef UD_WINO_LAUNCH
  UD_LAUNCH_CHECK();
  return UD_OK;
}
