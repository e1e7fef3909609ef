// fp32 weight gradient of the 3x3 / stride 1 / pad 1 convolutions as the gradient of the Winograd F(2x2, 3x3) form
// (conv2d_f32_wino.hip): with y = A^T [sum_c U . V] A per 2 x 2 tile,
//
//   dU[f][n][c] = sum_tiles (A dy A^T)[f][tile][n] * (B^T x B)[f][tile][c]        16 GEMMs, reduction index = tile
//   dW[n][.][c] = G^T dU[.][n][c] G                                              (k_wino_wgrad_finish, after the ordered slice sum)
//
// 16 multiplications per tile per (n, c) instead of 36 (2.25x fewer MFMA flops than k_conv3x3_wgrad_f32).  Replaces the
// backward-weights pass of the same layers (base_bev_backbone.py:38-115, center_head.py:58-99,311-355, lss_fpn.py:143-149).
//
// A workgroup (8 waves) owns a 64 (n) x 64 (c) block of dU for all 16 frequencies over a slice of the tiles; wave w holds
// frequencies 2w, 2w + 1 (2 x 16 blocks x 4 = 128 accumulator registers).  Per stage of 8 tiles: wave w loads tile w's raw
// 4 x 4 x-patch (lane = channel: 16 coalesced 256-byte loads) and 2 x 2 dy block straight into registers one stage ahead, transforms
// them and writes V / DY [f][tile][64] to LDS (double-buffered, 2 x 64 KB); the MFMA loop reads 16-byte fragments
// (lane group g = tile 4k + g, lane li = rows / columns 4 li .. 4 li + 3): 2 reads feed 16 MFMAs.  Slices are reduced in a
// fixed order (deterministic; no atomics).
// SQ counters (tools/pmc_wino.py): MFMA pipe 50 % busy, no LDS bank conflicts.  Tried and measured slower or equal: raw rows staged once
// per 8-tile stage by LDS-DMA with every wave building its two frequencies' fragments on the fly (26 KB per stage instead of 40 KB of
// L2 -> L1 traffic, no V / DY tensors: 0.234 ms vs 0.224 at 128 -> 128 @180^2 x 4; software-pipelined under the previous group's
// MFMAs with sched_group_barrier: 0.269 ms, register-bound at 256); prefetch one vs two stages ahead (equal).
#include "ud_common.h"
#include "ud_prof.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kST = 8;                               // tiles per stage (one per wave)
constexpr int kBufBytes = 2 * 16 * kST * 64 * 4;     // DY + V of one stage: 64 KB
constexpr int kVHalf = 16 * kST * 64 * 4;

struct WwGeom {
  int B, H, W, Cin, Cout, TX, TY;
  long long ntiles;
  int nstages, sps;                                  // stages in total / per slice
};

__global__ __launch_bounds__(512) void k_wino_wgrad_f32(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ partial, WwGeom gm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int n0 = blockIdx.y * 64, c0 = blockIdx.z * 64;
  const int s_begin = blockIdx.x * gm.sps, s_end = min(gm.nstages, s_begin + gm.sps);
  const bool cok = c0 + lane < gm.Cin, nok = n0 + lane < gm.Cout;

  float rx[2][16], rd[2][4];             // two sets: the loads of stage k + 2 fly while stage k + 1's are transformed
  // Raw loads, branch-free and cheap to address: the wave walks its tiles (tile id += 8 per stage) with a running (image, row,
  // column); every load is issued with coordinates clamped into the image (32-bit element offsets from the tensor base: the
  // launcher refuses tensors >= 2^31 elements) and invalid values are replaced in the transform -- 20 independent loads in flight.
  const unsigned xlane = (unsigned)min(c0 + lane, gm.Cin - 1), dlane = (unsigned)min(n0 + lane, gm.Cout - 1);
  unsigned xmask[2] = {0, 0}, dmask[2] = {0, 0};   // wave-uniform validity bits of the raw values in flight
  int tb, tty, ttx;                        // tile of this wave in the NEXT stage to load
  {
    const long long t = (long long)s_begin * kST + wave;
    const int per = gm.TX * gm.TY;
    tb = (int)(t / per);
    const int rem = (int)(t - (long long)tb * per);
    tty = rem / gm.TX, ttx = rem - tty * gm.TX;
  }
  // Addressing state of the wave's current tile (wave-uniform): element offsets of its 4 patch rows / 4 patch columns in x and of the
  // 2 x 2 block in dy, and the validity masks.  Interior tiles of a row (9 stages in 11 on the 180-wide maps) only step the column
  // offsets by 8 tiles; the full recomputation (clamps, masks: ~200 scalar instructions) runs at row ends and image borders.
  unsigned rowx[4], colx[4], rowd[2], cold[2], xm_cur = 0, dm_cur = 0;
  bool st_interior = false, wrapped = true;
  const bool xfull = c0 + 64 <= gm.Cin, dfull = n0 + 64 <= gm.Cout;
  auto load_part = [&](auto set_c, int part) {   // four parts: the stage loop spreads them between its first MFMAs
    constexpr int S = decltype(set_c)::value;
    if (part == 0) {
      const bool inter = tb < gm.B && ttx >= 1 && 2 * ttx + 2 < gm.W;
      if (st_interior && inter && !wrapped) {
#pragma unroll
        for (int j = 0; j < 4; ++j) colx[j] += 2 * kST * (unsigned)gm.Cin;
        cold[0] += 2 * kST * (unsigned)gm.Cout;
        cold[1] += 2 * kST * (unsigned)gm.Cout;
      } else {
        const int tv = tb < gm.B ? 1 : 0;
        const unsigned img = (unsigned)(tv ? tb : 0) * gm.H * gm.W;
        unsigned rmask = 0, cmask = 0, rowoff[4], coloff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gy = 2 * tty - 1 + i, gx = 2 * ttx - 1 + i;
          rmask |= (unsigned)((gy >= 0) & (gy < gm.H) & tv) << i;
          cmask |= (unsigned)((gx >= 0) & (gx < gm.W)) << i;
          rowoff[i] = img + (unsigned)min(max(gy, 0), gm.H - 1) * gm.W;
          coloff[i] = (unsigned)min(max(gx, 0), gm.W - 1);
          rowx[i] = rowoff[i] * (unsigned)gm.Cin;
          colx[i] = coloff[i] * (unsigned)gm.Cin;
        }
        rowd[0] = rowoff[1] * (unsigned)gm.Cout, rowd[1] = rowoff[2] * (unsigned)gm.Cout;
        cold[0] = coloff[1] * (unsigned)gm.Cout, cold[1] = coloff[2] * (unsigned)gm.Cout;
        xm_cur = dm_cur = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) xm_cur |= ((rmask >> i) & 1) ? cmask << (4 * i) : 0u;
#pragma unroll
        for (int a = 0; a < 2; ++a) dm_cur |= ((rmask >> (a + 1)) & 1) ? ((cmask >> 1) & 3u) << (2 * a) : 0u;
      }
      st_interior = inter;
      xmask[S] = xm_cur, dmask[S] = dm_cur;
    } else if (part < 3) {
#pragma unroll
      for (int i = 2 * (part - 1); i < 2 * (part - 1) + 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) rx[S][4 * i + j] = x[rowx[i] + colx[j] + xlane];
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) rd[S][2 * a + q] = dy[rowd[a] + cold[q] + dlane];
      ttx += kST;                          // next stage
      wrapped = false;
      while (ttx >= gm.TX) {
        ttx -= gm.TX;
        wrapped = true;
        if (++tty == gm.TY) tty = 0, ++tb;
      }
    }
  };
  auto load_raw = [&](auto set_c) {
#pragma unroll
    for (int part = 0; part < 4; ++part) load_part(set_c, part);
  };
  // transform of the raw values in flight, in four parts (part 0-3) so that the stage loop can spread it between its last MFMAs
  float td[4][4];
  auto transform_part = [&](auto set_c, int buf, int part) {
    constexpr int S = decltype(set_c)::value;
    float* D = reinterpret_cast<float*>(smem + buf * kBufBytes) + wave * 64 + lane;
    float* V = reinterpret_cast<float*>(smem + buf * kBufBytes + kVHalf) + wave * 64 + lane;
    if (part == 0) {          // DY = A dy A^T, A = [[1, 0], [1, 1], [1, -1], [0, -1]]
      if (!(dfull && dmask[S] == 0xFu)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rd[S][k] = (nok && ((dmask[S] >> k) & 1)) ? rd[S][k] : 0.f;
      }
      const float* r = rd[S];
      const float z[4][2] = {{r[0], r[1]}, {r[0] + r[2], r[1] + r[3]}, {r[0] - r[2], r[1] - r[3]}, {-r[2], -r[3]}};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        D[(4 * i + 0) * 512] = z[i][0];
        D[(4 * i + 1) * 512] = z[i][0] + z[i][1];
        D[(4 * i + 2) * 512] = z[i][0] - z[i][1];
        D[(4 * i + 3) * 512] = -z[i][1];
      }
    } else if (part == 1) {   // V = B^T x B: column pass
      if (!(xfull && xmask[S] == 0xFFFFu)) {
#pragma unroll
        for (int k = 0; k < 16; ++k) rx[S][k] = (cok && ((xmask[S] >> k) & 1)) ? rx[S][k] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        td[0][j] = rx[S][j] - rx[S][8 + j];
        td[1][j] = rx[S][4 + j] + rx[S][8 + j];
        td[2][j] = rx[S][8 + j] - rx[S][4 + j];
        td[3][j] = rx[S][4 + j] - rx[S][12 + j];
      }
    } else {                  // row pass + writes, two rows per part
#pragma unroll
      for (int i = 2 * (part - 2); i < 2 * (part - 2) + 2; ++i) {
        V[(4 * i + 0) * 512] = td[i][0] - td[i][2];
        V[(4 * i + 1) * 512] = td[i][1] + td[i][2];
        V[(4 * i + 2) * 512] = td[i][2] - td[i][1];
        V[(4 * i + 3) * 512] = td[i][1] - td[i][3];
      }
    }
  };
  auto transform = [&](auto set_c, int buf) {
#pragma unroll
    for (int part = 0; part < 4; ++part) transform_part(set_c, buf, part);
  };

  f32x4 acc[2][4][4];
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[fi][a][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads: hand-placed, one (frequency, tile group) ahead of its 16 MFMAs
  f32x4 qa[2], qb[2];
  const unsigned fbase = (2 * wave) * 2048 + g * 256 + li * 16;
#define UD_WW_LOADS(BUF, FI, K)                                                                                        \
  do {                                                                                                                 \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qa[BUF]) : "v"(pf), "n"((FI) * 2048 + (K) * 1024));            \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qb[BUF]) : "v"(pf), "n"(kVHalf + (FI) * 2048 + (K) * 1024));   \
  } while (0)
#define UD_WW_WAIT(BUF, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(qa[BUF]), "+v"(qb[BUF]))

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  const int nst = s_end - s_begin;
  if (nst > 0) {
    load_raw(Set0{});
    transform(Set0{}, 0);
    if (nst > 1) load_raw(Set1{});
  }
  // stage k (buffer k & 1): the raw loads of stage k + 2 go out between the first 16 MFMAs (into the register set stage k used),
  // stage k + 1's raw values -- issued a whole stage ago: ~2 us, global latency measured at ~1.2 us here -- are transformed between
  // the last 16.
  auto stage = [&](auto par_c, int k) {
    constexpr int P = decltype(par_c)::value;
    using SetP = std::integral_constant<int, P>;
    using SetQ = std::integral_constant<int, P ^ 1>;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();              // stage k is in LDS; everybody is done reading the other buffer
    const bool more = k + 1 < nst, more2 = k + 2 < nst;
    const unsigned pf = fbase + P * kBufBytes;
    UD_WW_LOADS(0, 0, 0);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int cur = st & 1;
      if (st + 1 < 4) {
        if (cur == 0) UD_WW_LOADS(1, (st + 1) >> 1, (st + 1) & 1); else UD_WW_LOADS(0, (st + 1) >> 1, (st + 1) & 1);
        if (cur == 0) UD_WW_WAIT(0, 2); else UD_WW_WAIT(1, 2);
      } else {
        UD_WW_WAIT(1, 0);
      }
      const int fi = st >> 1;
      // the raw loads of stage k + 2 are woven between the 16 MFMAs of step 0, the transform of stage k + 1 between those of step 3:
      // one MFMA, then a few of the other instructions (sched_group_barrier) -- chunks of them between groups of four MFMAs left the
      // pipe idle while both waves of the SIMD, in lockstep behind the stage barrier, worked through the same chunk (10.8 -> 10.4 ms
      // per step; spreading each over two steps was slower again)
      const bool ride = (st == 0 && more2) || (st == 3 && more);
      if (ride) {
        __builtin_amdgcn_sched_barrier(0);
        if (st == 0) load_raw(SetP{}); else transform(SetQ{}, P ^ 1);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[fi][a][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[cur][a], qb[cur][q], acc[fi][a][q], 0, 0, 0);
      if (ride) {
#pragma unroll
        for (int i_ = 0; i_ < 16; ++i_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (st == 0) {
            __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          } else {
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  for (int k = 0; k < nst; k += 2) {
    stage(Set0{}, k);
    if (k + 1 < nst) stage(Set1{}, k + 1);
  }
#undef UD_WW_LOADS
#undef UD_WW_WAIT
  // partial[slice][f][n][c]; D layout: lane holds rows 4 g + r <-> n = 4 (4 g + r) + a, column li <-> c = 4 li + q
  const int c = c0 + 4 * li;
  if (c < gm.Cin) {
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + 4 * (4 * g + r) + a;
          if (n < gm.Cout)
            *reinterpret_cast<float4*>(partial + (((size_t)blockIdx.x * 16 + 2 * wave + fi) * gm.Cout + n) * gm.Cin + c) =
                make_float4(acc[fi][a][0][r], acc[fi][a][1][r], acc[fi][a][2][r], acc[fi][a][3][r]);
        }
  }
}

// dw[n][tap][c] = sum_f coef[tap][f] * (sum over slices, in order, of partial[slice][f][n][c]); block = (n, 64 channels), thread = (f, 4 channels)
__global__ __launch_bounds__(256) void k_wino_wgrad_finish(const float* __restrict__ partial, int nslices, int Cout, int Cin,
                                                           float* __restrict__ dw) {
  __shared__ float4 M[16][16];
  const int cx = threadIdx.x & 15, f = threadIdx.x >> 4;
  const int n = blockIdx.x, c = blockIdx.y * 64 + cx * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < Cin)
    for (int sl = 0; sl < nslices; ++sl) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (((size_t)sl * 16 + f) * Cout + n) * Cin + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  M[f][cx] = s;
  __syncthreads();
  if (f < 9 && c < Cin) {
    const float GT[3][4] = {{1.f, 0.5f, 0.5f, 0.f}, {0.f, 0.5f, -0.5f, 0.f}, {0.f, 0.5f, 0.5f, 1.f}};
    const int ky = f / 3, kx = f - 3 * ky;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float k = GT[ky][i] * GT[kx][j];
        const float4 m = M[4 * i + j][cx];
        o.x += k * m.x; o.y += k * m.y; o.z += k * m.z; o.w += k * m.w;
      }
    *reinterpret_cast<float4*>(dw + ((size_t)n * 9 + f) * Cin + c) = o;
  }
}

struct WwPlan {
  int TX, TY, nstages, sps, nslices, nb, cb;
  long long ntiles;
};
WwPlan ww_plan(int B, int H, int W, int Cin, int Cout) {
  WwPlan p;
  p.TX = (W + 1) / 2, p.TY = (H + 1) / 2;
  p.ntiles = (long long)B * p.TX * p.TY;
  p.nstages = (int)((p.ntiles + kST - 1) / kST);
  p.nb = ud_div_up(Cout, 64), p.cb = ud_div_up(Cin, 64);
  int slices = 256 / (p.nb * p.cb);
  if (slices < 1) slices = 1;
  if (slices > p.nstages) slices = p.nstages;
  p.sps = ud_div_up(p.nstages, slices);
  p.nslices = ud_div_up(p.nstages, p.sps);
  return p;
}

}  // namespace

extern "C" size_t ud_conv3x3_wino_wgrad_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const WwPlan p = ww_plan(B, H, W, Cin, Cout);
  return (size_t)p.nslices * 16 * Cout * Cin * sizeof(float);
}

// x [B][H][W][Cin], dy [B][H][W][Cout] -> dw [Cout][3][3][Cin]  (same contract as ud_conv3x3_wgrad_nhwc_f32)
extern "C" int ud_conv3x3_wino_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin,
                                              int Cout, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Cin % 4 != 0 || Cout % 4 != 0) return UD_ERR_UNSUPPORTED;
  const WwPlan p = ww_plan(B, H, W, Cin, Cout);
  if (workspace_bytes < (size_t)p.nslices * 16 * Cout * Cin * sizeof(float)) return UD_ERR_WORKSPACE;
  if (p.nb > 65535 || p.cb > 65535 || (long long)B * H * W * Cin >= (1ll << 31) || (long long)B * H * W * Cout >= (1ll << 31))
    return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wino_wgrad_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBufBytes));
    attr_set.mark(attr_set_bit);
  }
  UdProfScope prof("conv2d.k_wgrad_wino_f32", stream);
  float* partial = static_cast<float*>(workspace);
  WwGeom gm{B, H, W, Cin, Cout, p.TX, p.TY, p.ntiles, p.nstages, p.sps};
  k_wino_wgrad_f32<<<dim3(p.nslices, p.nb, p.cb), 512, 2 * kBufBytes, stream>>>(x, dy, partial, gm);
  UD_LAUNCH_CHECK();
  k_wino_wgrad_finish<<<dim3(Cout, ud_div_up(Cin, 64)), 256, 0, stream>>>(partial, p.nslices, Cout, Cin, dw);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
