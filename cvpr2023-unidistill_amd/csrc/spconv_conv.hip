// Sparse 3-D convolution compute kernels for MI355X / gfx950 (fp32, exact-f32 MFMA).
//
// Replaces spconv's gather-GEMM-scatter (ConvAlgo.Native) behind SubMConv3d / SparseConv3d
// (unidistill/layers/blocks_3d/det3d/spconv_backbone.py:21-48,71-92,259-340).
//
// Output-stationary implicit GEMM over the dense rulebook nbr[site][K] of spconv_index.hip:
//   out[o, :] = bias + sum_k sum_c in[nbr[o][k], c] * W[:, k, c]
// A 256-thread workgroup owns 64 output rows; per kernel offset k (skipped when no row of the
// tile has that neighbour) the gathered input rows and W[:,k,:] are staged in LDS and multiplied
// with v_mfma_f32_16x16x4_f32 (exact fp32, f32 accumulate).  No scatter-add, no fp atomics: the
// result is deterministic.  dgrad is the same kernel on the transposed rulebook with W's n/c
// strides swapped; wgrad reduces over row chunks into ordered partials.
#include "ud_common.h"
#include "ud_prof.h"
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTM = 64;  // output rows per workgroup

// W element (n, k, c) lives at W[n*sn + k*sk + c*sc]
struct WStrides {
  long long sn, sk, sc;
};

// Optional fused epilogue: y = relu?((conv + bias) * scale + shift + residual)
struct ConvEpilogue {
  const float* scale;     // [Cout] or nullptr
  const float* shift;     // [Cout] or nullptr
  const float* residual;  // [Mout, Cout] or nullptr
  int relu;
};

template <int CIN_P, int COUT_P>
__global__ __launch_bounds__(256) void k_conv_mfma(const float* __restrict__ in, int cin,
                                                   const int32_t* __restrict__ nbr, int K,
                                                   int mirror, const float* __restrict__ W,
                                                   WStrides ws, const float* __restrict__ bias,
                                                   float* __restrict__ out, int cout, int Mout) {
  constexpr int LDA = CIN_P + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);       // [kTM][LDA]
  float* Bs = As + kTM * LDA;                        // [COUT_P][LDA]
  int* s_nbr = reinterpret_cast<int*>(Bs + COUT_P * LDA);  // [kTM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int row0 = blockIdx.x * kTM;
  constexpr int NT = COUT_P / 16;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int k = 0; k < K; ++k) {
    const int kk = mirror ? (K - 1 - k) : k;  // rulebook column to read for weight offset k
    int r = -1;
    if (tid < kTM && row0 + tid < Mout) r = nbr[(size_t)(row0 + tid) * K + kk];
    if (tid < kTM) s_nbr[tid] = r;
    if (!__syncthreads_or(r >= 0)) continue;  // nobody in this tile has neighbour k
    // ---- stage gathered input rows (zeros where the neighbour is missing / padded channels)
    if ((cin & 3) == 0) {
      for (int idx = tid; idx < kTM * (CIN_P / 4); idx += 256) {
        const int row = idx / (CIN_P / 4), c4 = (idx - row * (CIN_P / 4)) * 4;
        const int rr = s_nbr[row];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr >= 0 && c4 < cin) v = *reinterpret_cast<const float4*>(in + (size_t)rr * cin + c4);
        *reinterpret_cast<float4*>(As + row * LDA + c4) = v;
      }
    } else {
      for (int idx = tid; idx < kTM * CIN_P; idx += 256) {
        const int row = idx / CIN_P, c = idx - row * CIN_P;
        const int rr = s_nbr[row];
        As[row * LDA + c] = (rr >= 0 && c < cin) ? in[(size_t)rr * cin + c] : 0.f;
      }
    }
    // ---- stage W[:, k, :] as Bs[n][c]
    if (ws.sc == 1 && (cin & 3) == 0) {
      for (int idx = tid; idx < COUT_P * (CIN_P / 4); idx += 256) {
        const int n = idx / (CIN_P / 4), c4 = (idx - n * (CIN_P / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < cout && c4 < cin)
          v = *reinterpret_cast<const float4*>(W + n * ws.sn + k * ws.sk + c4);
        *reinterpret_cast<float4*>(Bs + n * LDA + c4) = v;
      }
    } else {
      // strided weights (dgrad reads W with n/c swapped): make n the fast thread index when sn == 1
      for (int idx = tid; idx < COUT_P * CIN_P; idx += 256) {
        int n, c;
        if (ws.sn == 1) {
          c = idx / COUT_P;
          n = idx - c * COUT_P;
        } else {
          n = idx / CIN_P;
          c = idx - n * CIN_P;
        }
        Bs[n * LDA + c] = (n < cout && c < cin) ? W[n * ws.sn + k * ws.sk + c * ws.sc] : 0.f;
      }
    }
    __syncthreads();
    // ---- MFMA: wave owns rows [16*wave, 16*wave+16) x all COUT_P columns
    const float* arow = As + (wave * 16 + li) * LDA + 4 * g;
#pragma unroll 2
    for (int cb = 0; cb < CIN_P / 16; ++cb) {
      const float4 a = *reinterpret_cast<const float4*>(arow + cb * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(Bs + (t * 16 + li) * LDA + cb * 16 + 4 * g);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- epilogue: acc[t][r] = out[row0 + 16*wave + 4*g + r][16*t + li]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 16 + li;
    if (col >= cout) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + wave * 16 + 4 * g + r;
      if (row < Mout) out[(size_t)row * cout + col] = acc[t][r] + bv;
    }
  }
}

// ---- v2: 128-row tiles, 8 waves, register-prefetched staging (wide channels) --------------------
// Used when both padded channel counts are >= 64.  Differences to k_conv_mfma:
//   * the tile's whole rulebook slice (K x 128 neighbour ids) is read once into LDS and turned
//     into a block-level and per-wave activity mask -> no global read / block reduction per offset;
//   * the gathered rows and W[:,k,:] of the NEXT active offset are fetched into registers before
//     the MFMA phase of the current one (global latency hidden under the matrix work) and written
//     to LDS after it; a wave skips the MFMAs of offsets none of its 16 rows uses;
//   * MFMAs are issued round-robin over the accumulators (no back-to-back dependent pairs).
constexpr int kTM2 = 128;

template <int CIN_P, int COUT_P, bool VA, bool VB>   // VA: cin % 4 == 0 (16-byte row pieces); VB: VA and channel-contiguous weights
__global__ __launch_bounds__(512) void k_conv_mfma_v2(const float* __restrict__ in, int cin,
                                                      const int32_t* __restrict__ nbr, int K,
                                                      int mirror, const float* __restrict__ W,
                                                      WStrides ws, const float* __restrict__ bias,
                                                      float* __restrict__ out, int cout, int Mout,
                                                      const int32_t* __restrict__ order,
                                                      ConvEpilogue ep) {
  // +8 floats per row: row stride = 8 (mod 64) banks, so the 16 lanes a ds_read_b128 serves together -- rows {0-3, 12-15}
  // of one 16-byte column and rows {4-11} of the next -- land on 16 distinct 16-byte slots (with +4, rows 11 and 12 of
  // neighbouring columns shared a slot: SQ_LDS_BANK_CONFLICT was 40 % of the LDS cycles)
  constexpr int LDA = CIN_P + 8;
  constexpr int NT = COUT_P / 16;
  constexpr int A4 = kTM2 * (CIN_P / 4) / 512;                  // float4 per thread for the A tile
  constexpr int B4 = (COUT_P * (CIN_P / 4) + 511) / 512;        // float4 per thread for the B tile
  constexpr int BU = COUT_P * (CIN_P / 4);                      // float4 units in the B tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);          // [kTM2][LDA]
  float* Bs = As + kTM2 * LDA;                          // [COUT_P][LDA]
  int* s_nbr = reinterpret_cast<int*>(Bs + COUT_P * LDA);  // [K][kTM2]
  // all LDS lives in the dynamic region (a static __shared__ in front would shift its base off
  // the 16-byte alignment the b128 reads need)
  unsigned& s_active = *reinterpret_cast<unsigned*>(s_nbr + K * kTM2);
  int* s_row = s_nbr + K * kTM2 + 4;                    // [kTM2] output row of each tile row (-1: none)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int row0 = (int)(gridDim.x - 1u - blockIdx.x) * kTM2;   // mask-sorted rows: the tiles with the most active offsets sit at the end -> dispatch them first (longest first)
  if (tid == 0) s_active = 0u;
  __syncthreads();
  // tile row -> output row: identity, or the caller's mask-sorted order (rows with similar
  // neighbour masks share a tile, so far fewer offsets are active per tile)
  if (tid < kTM2) {
    const int p = row0 + tid;
    s_row[tid] = (p < Mout) ? (order ? order[p] : p) : -1;
  }
  __syncthreads();
  // rulebook slice -> LDS (column kk of the rulebook serves weight offset k)
  unsigned mine = 0u;
  for (int idx = tid; idx < kTM2 * K; idx += 512) {
    const int r = idx / K, k = idx - r * K;
    int v = -1;
    const int orow = s_row[r];
    if (orow >= 0) v = nbr[(size_t)orow * K + (mirror ? K - 1 - k : k)];
    s_nbr[k * kTM2 + r] = v;
    if (v >= 0) mine |= 1u << k;
  }
  // block-wide OR of the per-thread masks
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor((int)mine, o);
  if (lane == 0 && mine) atomicOr(&s_active, mine);
  __syncthreads();
  const unsigned active = s_active;
  // per-wave mask: which offsets do my 16 rows use
  unsigned wmask = 0u;
  for (int k = 0; k < K; ++k) {
    const int v = (lane < 16) ? s_nbr[k * kTM2 + wave * 16 + lane] : -1;
    if (__any(v >= 0)) wmask |= 1u << k;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[A4], rb[VB ? B4 : 1];
  // thread <-> (row or weight row r0 + RPP j, 16-byte channel piece c4): the same piece for every j
  constexpr int PER = CIN_P / 4, RPP = 512 / PER;
  const int r0 = tid / PER, c4 = (tid % PER) * 4;
  const bool cok = c4 < cin;
  const float* in_c = in + (cok ? c4 : 0);
  const int wc4 = cok ? c4 : 0;

  // The gathered rows AND W[:,k,:] of the next active offset are fetched into registers during the MFMA phase of the
  // current one and written to LDS after it.  Every load is unconditional (rows without a neighbour read row 0, weight
  // rows past cout re-read the last one; the zeros go in at commit time), and the vector / scalar variants are template
  // parameters: chosen at run time (round 1) the 128 x 128 instance needed 256 VGPRs and spilled 95 SGPRs, and its
  // weights were loaded synchronously inside the commit.
  auto fetch_a = [&](int k, int j) {
    const int rr = s_nbr[k * kTM2 + r0 + RPP * j];
    if constexpr (VA) {
      ra[j] = *reinterpret_cast<const float4*>(in_c + (size_t)(rr < 0 ? 0 : rr) * cin);
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr >= 0) {
        const float* src = in + (size_t)rr * cin + c4;
        if (c4 + 0 < cin) v.x = src[0];
        if (c4 + 1 < cin) v.y = src[1];
        if (c4 + 2 < cin) v.z = src[2];
        if (c4 + 3 < cin) v.w = src[3];
      }
      ra[j] = v;
    }
  };
  auto fetch_b = [&](int k, int j) {
    if constexpr (VB) {
      const int n = r0 + RPP * j;
      // 32-bit offset from the (uniform) weight pointer: eight 64-bit row pointers would be hoisted out of the offset loop
      // and spilled
      if (n < COUT_P)
        rb[j] = *reinterpret_cast<const float4*>(W + (unsigned)(k * (int)ws.sk + (n < cout ? n : cout - 1) * (int)ws.sn + wc4));
    }
  };
  auto fetch = [&](int k) {
#pragma unroll
    for (int j = 0; j < A4; ++j) fetch_a(k, j);
#pragma unroll
    for (int j = 0; j < B4; ++j) fetch_b(k, j);
  };
  auto commit = [&](int k) {
    // component-wise selects: `ok ? ra[j] : zero4` on the float4 structs becomes a select of two ADDRESSES, which sends the
    // register arrays to scratch memory
    auto keep = [](bool ok, const float4& v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); };
#pragma unroll
    for (int j = 0; j < A4; ++j) {
      const int row = r0 + RPP * j;
      const bool ok = !VA || (cok && s_nbr[k * kTM2 + row] >= 0);
      *reinterpret_cast<float4*>(As + row * LDA + c4) = keep(ok, ra[j]);
    }
    if constexpr (VB) {
#pragma unroll
      for (int j = 0; j < B4; ++j) {
        const int n = r0 + RPP * j;
        if (n < COUT_P) *reinterpret_cast<float4*>(Bs + n * LDA + c4) = keep(cok && n < cout, rb[j]);
      }
    } else {
      // strided weights (dgrad: n is the fast axis in memory): thread <-> (c, 4 consecutive n)
#pragma unroll
      for (int j = 0; j < B4; ++j) {
        const int u = tid + 512 * j;
        if (u >= BU) break;
        const int c = u / (COUT_P / 4), n4 = (u - c * (COUT_P / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < cin) {
          const float* src = W + k * ws.sk + c * ws.sc;
          if (n4 + 0 < cout) v.x = src[(n4 + 0) * ws.sn];
          if (n4 + 1 < cout) v.y = src[(n4 + 1) * ws.sn];
          if (n4 + 2 < cout) v.z = src[(n4 + 2) * ws.sn];
          if (n4 + 3 < cout) v.w = src[(n4 + 3) * ws.sn];
        }
        Bs[(n4 + 0) * LDA + c] = v.x;
        Bs[(n4 + 1) * LDA + c] = v.y;
        Bs[(n4 + 2) * LDA + c] = v.z;
        Bs[(n4 + 3) * LDA + c] = v.w;
      }
    }
  };

  unsigned todo = active;
  int k = todo ? (__ffs((int)todo) - 1) : -1;
  if (k >= 0) fetch(k);
  while (k >= 0) {
    todo &= todo - 1;
    __builtin_amdgcn_sched_barrier(0);     // the phases stay apart too (same reason as between the matrix steps below)
    commit(k);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    const int knext = todo ? (__ffs((int)todo) - 1) : -1;
    // The next offset's loads are issued BETWEEN the matrix steps, one row piece (and at most one weight piece) per
    // 16-channel step: issued in one burst after the barrier, the eight waves' 128 KB of requests took 4 600 cycles to
    // get into the memory pipeline with the MFMA pipes idle (in-kernel cycle stamps).
    constexpr int CB = CIN_P / 16;
    static_assert(A4 == CB, "one gathered-row piece per 16-channel step");
    if ((wmask >> k) & 1u) {
      const float* arow = As + (wave * 16 + li) * LDA + 4 * g;
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        if (knext >= 0) {
          fetch_a(knext, cb);
#pragma unroll
          for (int j = 0; j < B4; ++j)
            if (j * CB / B4 == cb) fetch_b(knext, j);
        }
        const float4 a = *reinterpret_cast<const float4*>(arow + cb * 16);
        float4 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
          b[t] = *reinterpret_cast<const float4*>(Bs + (t * 16 + li) * LDA + cb * 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
        // keep the steps apart: with the loop unrolled the scheduler hoists the fragment reads of several steps to the top
        // (256 VGPRs, 24 of them spilled in the 128 x 128 instance)
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (knext >= 0) {
      fetch(knext);
    }
    __syncthreads();
    k = knext;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 16 + li;
    if (col >= cout) continue;
    const float bv = bias ? bias[col] : 0.f;
    const float sc = ep.scale ? ep.scale[col] : 1.f;
    const float sh = ep.shift ? ep.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = s_row[wave * 16 + 4 * g + r];
      if (row >= 0) {
        float v = acc[t][r] + bv;
        if (ep.scale) v = v * sc + sh;                       // folded eval-mode BatchNorm
        if (ep.residual) v += ep.residual[(size_t)row * cout + col];
        if (ep.relu) v = fmaxf(v, 0.f);
        out[(size_t)row * cout + col] = v;
      }
    }
  }
}

// ---- bf16-MFMA variant (algo 3, mixed-precision mode) -------------------------------------------
// Same tiling / masks / prefetch as k_conv_mfma_v2, but the gathered rows and the weights are
// rounded to bf16 when they are written to LDS and multiplied with v_mfma_f32_16x16x32_bf16
// (fp32 accumulate): 8x fewer matrix instructions at ~2x the issue rate, so the wide layers stop
// being bound by the fp32 matrix pipe (157 TF) and become staging bound.  Inputs/outputs in HBM
// stay fp32.  Used when the caller runs under bf16 autocast; NOT bit-compatible with the fp32 path.
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) { return ud_pack_bf16x2(a, b); }

template <int CIN_P, int COUT_P>   // CIN_P multiple of 32
__global__ __launch_bounds__(512) void k_conv_mfma_bf16(const float* __restrict__ in, int cin,
                                                        const int32_t* __restrict__ nbr, int K,
                                                        int mirror, const float* __restrict__ W,
                                                        WStrides ws, const float* __restrict__ bias,
                                                        float* __restrict__ out, int cout, int Mout,
                                                        const int32_t* __restrict__ order,
                                                        ConvEpilogue ep, int io) {
  // io bit 0: `in` holds bf16 rows; bit 1: `out` and ep.residual hold bf16; bit 2: W holds bf16
  const bool in_bf = io & 1, out_bf = io & 2, w_bf = io & 4;
  const unsigned short* in_h = reinterpret_cast<const unsigned short*>(in);
  const unsigned short* W_h = reinterpret_cast<const unsigned short*>(W);
  constexpr int LDB = CIN_P + 8;                                 // bf16 elements per LDS row
  constexpr int NT = COUT_P / 16;
  constexpr int A4 = kTM2 * (CIN_P / 4) / 512;
  constexpr int B4 = (COUT_P * (CIN_P / 4) + 511) / 512;
  constexpr int BU = COUT_P * (CIN_P / 4);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);  // [kTM2][LDB]
  unsigned short* Bs = As + kTM2 * LDB;                           // [COUT_P][LDB]
  int* s_nbr = reinterpret_cast<int*>(Bs + COUT_P * LDB);         // [K][kTM2]
  unsigned& s_active = *reinterpret_cast<unsigned*>(s_nbr + K * kTM2);
  int* s_row = s_nbr + K * kTM2 + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int row0 = (int)(gridDim.x - 1u - blockIdx.x) * kTM2;   // mask-sorted rows: the tiles with the most active offsets sit at the end -> dispatch them first (longest first)
  if (tid == 0) s_active = 0u;
  if (tid < kTM2) {
    const int p = row0 + tid;
    s_row[tid] = (p < Mout) ? (order ? order[p] : p) : -1;
  }
  __syncthreads();
  unsigned mine = 0u;
  for (int idx = tid; idx < kTM2 * K; idx += 512) {
    const int r = idx / K, k = idx - r * K;
    int v = -1;
    const int orow = s_row[r];
    if (orow >= 0) v = nbr[(size_t)orow * K + (mirror ? K - 1 - k : k)];
    s_nbr[k * kTM2 + r] = v;
    if (v >= 0) mine |= 1u << k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor((int)mine, o);
  if (lane == 0 && mine) atomicOr(&s_active, mine);
  __syncthreads();
  const unsigned active = s_active;
  unsigned wmask = 0u;
  for (int k = 0; k < K; ++k) {
    const int v = (lane < 16) ? s_nbr[k * kTM2 + wave * 16 + lane] : -1;
    if (__any(v >= 0)) wmask |= 1u << k;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool vecA = (cin & 3) == 0;
  const bool vecB = (ws.sc == 1) && vecA;
  float4 ra[A4];
  auto fetch = [&](int k) {
#pragma unroll
    for (int j = 0; j < A4; ++j) {
      const int u = tid + 512 * j;
      const int row = u / (CIN_P / 4), c4 = (u - row * (CIN_P / 4)) * 4;
      const int rr = s_nbr[k * kTM2 + row];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr >= 0 && in_bf) {          // bf16 rows (cin % 4 == 0): 8 bytes carry the 4 channels
        if (c4 < cin) {
          const uint2 h = *reinterpret_cast<const uint2*>(in_h + (size_t)rr * cin + c4);
          v.x = __uint_as_float(h.x);
          v.y = __uint_as_float(h.y);
        }
      } else if (rr >= 0) {
        const float* src = in + (size_t)rr * cin + c4;
        if (vecA) {
          if (c4 < cin) v = *reinterpret_cast<const float4*>(src);
        } else {
          if (c4 + 0 < cin) v.x = src[0];
          if (c4 + 1 < cin) v.y = src[1];
          if (c4 + 2 < cin) v.z = src[2];
          if (c4 + 3 < cin) v.w = src[3];
        }
      }
      ra[j] = v;
    }
  };
  auto commit = [&](int k) {
#pragma unroll
    for (int j = 0; j < A4; ++j) {
      const int u = tid + 512 * j;
      const int row = u / (CIN_P / 4), c4 = (u - row * (CIN_P / 4)) * 4;
      *reinterpret_cast<uint2*>(As + row * LDB + c4) =
          in_bf ? make_uint2(__float_as_uint(ra[j].x), __float_as_uint(ra[j].y))
                : make_uint2(pack_bf16x2(ra[j].x, ra[j].y), pack_bf16x2(ra[j].z, ra[j].w));
    }
    if (vecB) {
#pragma unroll
      for (int j = 0; j < B4; ++j) {
        const int u = tid + 512 * j;
        if (u >= BU) break;
        const int n = u / (CIN_P / 4), c4 = (u - n * (CIN_P / 4)) * 4;
        uint2 o = make_uint2(0u, 0u);
        if (n < cout && c4 < cin) {
          if (w_bf) {
            o = *reinterpret_cast<const uint2*>(W_h + n * ws.sn + k * ws.sk + c4);
          } else {
            const float4 v = *reinterpret_cast<const float4*>(W + n * ws.sn + k * ws.sk + c4);
            o = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
          }
        }
        *reinterpret_cast<uint2*>(Bs + n * LDB + c4) = o;
      }
    } else {
#pragma unroll
      for (int j = 0; j < B4; ++j) {
        const int u = tid + 512 * j;
        if (u >= BU) break;
        const int c = u / (COUT_P / 4), n4 = (u - c * (COUT_P / 4)) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < cin) {
          const float* src = W + k * ws.sk + c * ws.sc;
          for (int q = 0; q < 4; ++q)
            if (n4 + q < cout) v[q] = src[(n4 + q) * ws.sn];
        }
        for (int q = 0; q < 4; ++q)
          Bs[(n4 + q) * LDB + c] = (unsigned short)(ud_pack_bf16x2(v[q], 0.f) & 0xFFFFu);
      }
    }
  };
  unsigned todo = active;
  int k = todo ? (__ffs((int)todo) - 1) : -1;
  if (k >= 0) fetch(k);
  while (k >= 0) {
    todo &= todo - 1;
    commit(k);
    __syncthreads();
    const int knext = todo ? (__ffs((int)todo) - 1) : -1;
    if (knext >= 0) fetch(knext);
    if ((wmask >> k) & 1u) {
      const unsigned short* arow = As + (wave * 16 + li) * LDB + 8 * g;
#pragma unroll
      for (int cb = 0; cb < CIN_P / 32; ++cb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + cb * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (t * 16 + li) * LDB + cb * 32 + 8 * g);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    k = knext;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 16 + li;
    if (col >= cout) continue;
    const float bv = bias ? bias[col] : 0.f;
    const float sc = ep.scale ? ep.scale[col] : 1.f;
    const float sh = ep.shift ? ep.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = s_row[wave * 16 + 4 * g + r];
      if (row >= 0) {
        float v = acc[t][r] + bv;
        if (ep.scale) v = v * sc + sh;
        if (ep.residual)
          v += out_bf ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(
                                            ep.residual)[(size_t)row * cout + col] << 16)
                      : ep.residual[(size_t)row * cout + col];
        if (ep.relu) v = fmaxf(v, 0.f);
        if (out_bf)
          reinterpret_cast<unsigned short*>(out)[(size_t)row * cout + col] =
              (unsigned short)(ud_pack_bf16x2(v, 0.f) & 0xFFFFu);
        else
          out[(size_t)row * cout + col] = v;
      }
    }
  }
}

// ---- all-bf16 inference kernel (bf16 rows, bf16 weights, bf16 output) ------------------------------
// Same tile / mask scheme as k_conv_mfma_bf16, with everything the mixed-precision teacher path
// allows: 16-byte gathers of bf16 rows, the next offset's A rows AND weight slice prefetched in
// registers while the current one is multiplied (the weight slice was a synchronous L2 round trip
// per offset before -- the dominant cost once the products run on the bf16 pipe), and the output tile
// staged through LDS as fp32 so residual rows are read and results written as 16-byte pieces.
// bytes of the region shared by the operand tiles (K loop) and the staged fp32 output tile (epilogue)
__host__ __device__ constexpr size_t fast_front_bytes(int cin_p, int cout_p) {
  const size_t a = (size_t)(kTM2 + cout_p) * (cin_p + 8) * 2, o = (size_t)kTM2 * (cout_p + 4) * 4;
  return ((a > o ? a : o) + 15) / 16 * 16;
}

template <int CIN_P, int COUT_P>   // CIN_P multiple of 32; cin, cout multiples of 8
__global__ __launch_bounds__(512) void k_conv_mfma_bf16_fast(
    const unsigned short* __restrict__ in, int cin, const int32_t* __restrict__ nbr, int K, int mirror,
    const unsigned short* __restrict__ W, WStrides ws, const float* __restrict__ bias,
    unsigned short* __restrict__ out, int cout, int Mout, const int32_t* __restrict__ order,
    ConvEpilogue ep) {
  constexpr int LDB = CIN_P + 8;
  constexpr int NT = COUT_P / 16;
  constexpr int AU = kTM2 * (CIN_P / 8);            // 16-byte units of the A tile
  constexpr int BU = COUT_P * (CIN_P / 8);
  constexpr int A8 = (AU + 511) / 512, B8 = (BU + 511) / 512;
  constexpr int LDO = COUT_P + 4;                   // fp32 elements per staged output row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Bs = As + kTM2 * LDB;
  int* s_nbr = reinterpret_cast<int*>(smem + fast_front_bytes(CIN_P, COUT_P));
  unsigned& s_active = *reinterpret_cast<unsigned*>(s_nbr + K * kTM2);
  int* s_row = s_nbr + K * kTM2 + 4;
  float* Os = reinterpret_cast<float*>(smem);       // [kTM2][LDO], aliases As/Bs after the K loop
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int row0 = (int)(gridDim.x - 1u - blockIdx.x) * kTM2;   // mask-sorted rows: the tiles with the most active offsets sit at the end -> dispatch them first (longest first)
  if (tid == 0) s_active = 0u;
  if (tid < kTM2) {
    const int p = row0 + tid;
    s_row[tid] = (p < Mout) ? (order ? order[p] : p) : -1;
  }
  __syncthreads();
  unsigned mine = 0u;
  for (int idx = tid; idx < kTM2 * K; idx += 512) {
    const int r = idx / K, k = idx - r * K;
    int v = -1;
    const int orow = s_row[r];
    if (orow >= 0) v = nbr[(size_t)orow * K + (mirror ? K - 1 - k : k)];
    s_nbr[k * kTM2 + r] = v;
    if (v >= 0) mine |= 1u << k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor((int)mine, o);
  if (lane == 0 && mine) atomicOr(&s_active, mine);
  __syncthreads();
  const unsigned active = s_active;
  unsigned wmask = 0u;
  for (int k = 0; k < K; ++k) {
    const int v = (lane < 16) ? s_nbr[k * kTM2 + wave * 16 + lane] : -1;
    if (__any(v >= 0)) wmask |= 1u << k;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 ra[A8], rb[B8];
  auto fetch = [&](int k) {
#pragma unroll
    for (int j = 0; j < A8; ++j) {
      const int u = tid + 512 * j;
      ra[j] = make_uint4(0u, 0u, 0u, 0u);
      if (u < AU) {
        const int row = u / (CIN_P / 8), c8 = (u - row * (CIN_P / 8)) * 8;
        const int rr = s_nbr[k * kTM2 + row];
        if (rr >= 0 && c8 < cin) ra[j] = *reinterpret_cast<const uint4*>(in + (size_t)rr * cin + c8);
      }
    }
#pragma unroll
    for (int j = 0; j < B8; ++j) {
      const int u = tid + 512 * j;
      rb[j] = make_uint4(0u, 0u, 0u, 0u);
      if (u < BU) {
        const int n = u / (CIN_P / 8), c8 = (u - n * (CIN_P / 8)) * 8;
        if (n < cout && c8 < cin) rb[j] = *reinterpret_cast<const uint4*>(W + n * ws.sn + k * ws.sk + c8);
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < A8; ++j) {
      const int u = tid + 512 * j;
      if (u < AU) {
        const int row = u / (CIN_P / 8), c8 = (u - row * (CIN_P / 8)) * 8;
        *reinterpret_cast<uint4*>(As + row * LDB + c8) = ra[j];
      }
    }
#pragma unroll
    for (int j = 0; j < B8; ++j) {
      const int u = tid + 512 * j;
      if (u < BU) {
        const int n = u / (CIN_P / 8), c8 = (u - n * (CIN_P / 8)) * 8;
        *reinterpret_cast<uint4*>(Bs + n * LDB + c8) = rb[j];
      }
    }
  };
  unsigned todo = active;
  int k = todo ? (__ffs((int)todo) - 1) : -1;
  if (k >= 0) fetch(k);
  while (k >= 0) {
    todo &= todo - 1;
    commit();
    __syncthreads();
    const int knext = todo ? (__ffs((int)todo) - 1) : -1;
    if (knext >= 0) fetch(knext);
    if ((wmask >> k) & 1u) {
      const unsigned short* arow = As + (wave * 16 + li) * LDB + 8 * g;
#pragma unroll
      for (int cb = 0; cb < CIN_P / 32; ++cb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + cb * 32);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (t * 16 + li) * LDB + cb * 32 + 8 * g);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    k = knext;
  }
  // epilogue 1: bias + folded BatchNorm in registers, fp32 tile -> LDS
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 16 + li;
    const bool cv = col < cout;
    const float bv = (cv && bias) ? bias[col] : 0.f;
    const float sc = (cv && ep.scale) ? ep.scale[col] : 1.f;
    const float sh = (cv && ep.shift) ? ep.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[(wave * 16 + 4 * g + r) * LDO + col] = (acc[t][r] + bv) * sc + sh;
  }
  __syncthreads();
  // epilogue 2: + residual, ReLU, bf16, 16-byte rows
  const unsigned short* res = reinterpret_cast<const unsigned short*>(ep.residual);
  for (int u = tid; u < kTM2 * (COUT_P / 8); u += 512) {
    const int r = u / (COUT_P / 8), c8 = (u - r * (COUT_P / 8)) * 8;
    const int row = s_row[r];
    if (row < 0 || c8 >= cout) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(Os + r * LDO + c8);
    const float4 v1 = *reinterpret_cast<const float4*>(Os + r * LDO + c8 + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if (res) {
      const uint4 h = *reinterpret_cast<const uint4*>(res + (size_t)row * cout + c8);
      const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[2 * q] += __uint_as_float(hw[q] << 16);
        v[2 * q + 1] += __uint_as_float(hw[q] & 0xFFFF0000u);
      }
    }
    if (ep.relu) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
    }
    *reinterpret_cast<uint4*>(out + (size_t)row * cout + c8) =
        make_uint4(ud_pack_bf16x2(v[0], v[1]), ud_pack_bf16x2(v[2], v[3]), ud_pack_bf16x2(v[4], v[5]),
                   ud_pack_bf16x2(v[6], v[7]));
  }
}

// Out-of-rulebook rows and output channels past cout are DMA'd from here (LDS-DMA cannot write a constant).
__device__ __attribute__((aligned(16))) unsigned int g_sp_zero16[4];

// bytes of the LDS region shared by the double-buffered operand tiles and the staged fp32 output tile
__host__ __device__ constexpr size_t dma_front_bytes(int cin_p, int cout_p) {
  const int rpp = 64 / (cin_p / 8);
  const size_t a = (size_t)2 * (kTM2 + (cout_p + rpp - 1) / rpp * rpp) * cin_p * 2;
  const size_t o = (size_t)kTM2 * (cout_p + 4) * 4;
  return ((a > o ? a : o) + 15) / 16 * 16;
}

// ---- all-bf16 kernel with LDS-DMA staging -----------------------------------------------------------
// k_conv_mfma_bf16_fast with the gathered rows and the weight slice moved L2 -> LDS by
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), double-buffered so the next offset's
// operands fly while the current offset is multiplied: one barrier per offset instead of two.
template <int CIN_P, int COUT_P>   // cin == CIN_P in {32, 64, 128}; cout multiple of 8
__global__ __launch_bounds__(512) void k_conv_mfma_bf16_dma(
    const unsigned short* __restrict__ in, int cin, const int32_t* __restrict__ nbr, int K, int mirror,
    const unsigned short* __restrict__ W, WStrides ws, const float* __restrict__ bias,
    unsigned short* __restrict__ out, int cout, int Mout, const int32_t* __restrict__ order,
    ConvEpilogue ep) {
  constexpr int NT = COUT_P / 16;
  constexpr int S = CIN_P / 8;                      // 16-byte slots per (unpadded) LDS row
  constexpr int RPP = 64 / S;                       // rows per 1-KiB DMA piece
  constexpr int APIECES = kTM2 / RPP, BPIECES = (COUT_P + RPP - 1) / RPP;
  constexpr int LDO = COUT_P + 4;                   // fp32 elements per staged output row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* As = reinterpret_cast<unsigned short*>(smem);            // [2][kTM2][CIN_P]
  unsigned short* Bs = As + 2 * kTM2 * CIN_P;                                // [2][BPIECES*RPP][CIN_P]
  int* s_nbr = reinterpret_cast<int*>(smem + dma_front_bytes(CIN_P, COUT_P));
  unsigned& s_active = *reinterpret_cast<unsigned*>(s_nbr + K * kTM2);
  int* s_row = s_nbr + K * kTM2 + 4;
  float* Os = reinterpret_cast<float*>(smem);       // [kTM2][LDO], aliases As/Bs after the K loop
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int row0 = (int)(gridDim.x - 1u - blockIdx.x) * kTM2;   // mask-sorted rows: the tiles with the most active offsets sit at the end -> dispatch them first (longest first)
  if (tid == 0) s_active = 0u;
  if (tid < kTM2) {
    const int p = row0 + tid;
    s_row[tid] = (p < Mout) ? (order ? order[p] : p) : -1;
  }
  __syncthreads();
  unsigned mine = 0u;
  for (int idx = tid; idx < kTM2 * K; idx += 512) {
    const int r = idx / K, k = idx - r * K;
    int v = -1;
    const int orow = s_row[r];
    if (orow >= 0) v = nbr[(size_t)orow * K + (mirror ? K - 1 - k : k)];
    s_nbr[k * kTM2 + r] = v;
    if (v >= 0) mine |= 1u << k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor((int)mine, o);
  if (lane == 0 && mine) atomicOr(&s_active, mine);
  __syncthreads();
  const unsigned active = s_active;
  unsigned wmask = 0u;
  for (int k = 0; k < K; ++k) {
    const int v = (lane < 16) ? s_nbr[k * kTM2 + wave * 16 + lane] : -1;
    if (__any(v >= 0)) wmask |= 1u << k;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Row r of a tile keeps its 16-byte slot t at position t ^ swz(r): the swizzle is applied to the SOURCE
  // address (the DMA's LDS pattern is fixed: lane l lands at piece base + 16 l) and makes every
  // ds_read_b128 fragment load below conflict-free on the unpadded rows.
  auto swz = [](int r) -> int {
    if (S == 16) return r & 15;
    if (S == 8) return (r >> 1) & 7;
    const int q = (r >> 2) & 3;                      // S == 4
    return q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 3 : 1));
  };
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_sp_zero16);
  const int prow = lane / S, pslot = lane % S;
  auto stage = [&](int k, int buf) {
#pragma unroll
    for (int j = 0; j < (APIECES + 7) / 8; ++j) {
      const int piece = wave + 8 * j;
      if (piece < APIECES) {
        const int r = piece * RPP + prow;
        const int rr = s_nbr[k * kTM2 + r];
        const unsigned short* src = rr >= 0 ? in + (size_t)rr * cin + ((pslot ^ swz(r)) << 3) : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(As + (buf * kTM2 + piece * RPP) * CIN_P),
                                         16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < (BPIECES + 7) / 8; ++j) {
      const int piece = wave + 8 * j;
      if (piece < BPIECES) {
        const int n = piece * RPP + prow;
        const unsigned short* src = n < cout ? W + n * ws.sn + k * ws.sk + ((pslot ^ swz(n)) << 3) : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(Bs + (buf * BPIECES * RPP + piece * RPP) * CIN_P),
                                         16, 0, 0);
      }
    }
  };
  unsigned todo = active;
  int k = todo ? (__ffs((int)todo) - 1) : -1;
  int buf = 0;
  if (k >= 0) stage(k, 0);
  __syncthreads();                                   // drains the DMAs (vmcnt(0)) and publishes the tiles
  while (k >= 0) {
    todo &= todo - 1;
    const int knext = todo ? (__ffs((int)todo) - 1) : -1;
    if (knext >= 0) stage(knext, buf ^ 1);           // flies while this offset is multiplied
    if ((wmask >> k) & 1u) {
      const int ar = wave * 16 + li;
      const unsigned short* arow = As + (buf * kTM2 + ar) * CIN_P;
      const int asw = swz(ar);
#pragma unroll
      for (int cb = 0; cb < CIN_P / 32; ++cb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + (((4 * cb + g) ^ asw) << 3));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int n = t * 16 + li;
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (buf * BPIECES * RPP + n) * CIN_P +
                                                             (((4 * cb + g) ^ swz(n)) << 3));
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    buf ^= 1;
    k = knext;
  }
  // epilogue 1: bias + folded BatchNorm in registers, fp32 tile -> LDS
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = t * 16 + li;
    const bool cv = col < cout;
    const float bv = (cv && bias) ? bias[col] : 0.f;
    const float sc = (cv && ep.scale) ? ep.scale[col] : 1.f;
    const float sh = (cv && ep.shift) ? ep.shift[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) Os[(wave * 16 + 4 * g + r) * LDO + col] = (acc[t][r] + bv) * sc + sh;
  }
  __syncthreads();
  // epilogue 2: + residual, ReLU, bf16, 16-byte rows
  const unsigned short* res = reinterpret_cast<const unsigned short*>(ep.residual);
  for (int u = tid; u < kTM2 * (COUT_P / 8); u += 512) {
    const int r = u / (COUT_P / 8), c8 = (u - r * (COUT_P / 8)) * 8;
    const int row = s_row[r];
    if (row < 0 || c8 >= cout) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(Os + r * LDO + c8);
    const float4 v1 = *reinterpret_cast<const float4*>(Os + r * LDO + c8 + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if (res) {
      const uint4 h = *reinterpret_cast<const uint4*>(res + (size_t)row * cout + c8);
      const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[2 * q] += __uint_as_float(hw[q] << 16);
        v[2 * q + 1] += __uint_as_float(hw[q] & 0xFFFF0000u);
      }
    }
    if (ep.relu) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
    }
    *reinterpret_cast<uint4*>(out + (size_t)row * cout + c8) =
        make_uint4(ud_pack_bf16x2(v[0], v[1]), ud_pack_bf16x2(v[2], v[3]), ud_pack_bf16x2(v[4], v[5]),
                   ud_pack_bf16x2(v[6], v[7]));
  }
}

// Any-size fallback (and the cross-check in tests): one thread per (row, n), sequential k, c.
__global__ __launch_bounds__(256) void k_conv_generic(const float* __restrict__ in, int cin,
                                                      const int32_t* __restrict__ nbr, int K,
                                                      int mirror, const float* __restrict__ W,
                                                      WStrides ws, const float* __restrict__ bias,
                                                      float* __restrict__ out, int cout, int Mout) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)Mout * cout) return;
  const int row = (int)(t / cout), n = (int)(t - (long long)row * cout);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int r = nbr[(size_t)row * K + (mirror ? K - 1 - k : k)];
    if (r < 0) continue;
    const float* x = in + (size_t)r * cin;
    const float* w = W + n * ws.sn + k * ws.sk;
    for (int c = 0; c < cin; ++c) acc = fmaf(x[c], w[c * ws.sc], acc);
  }
  out[t] = acc + (bias ? bias[n] : 0.f);
}

// ---- weight gradient ------------------------------------------------------------------------
// partial[g][k][n][c] = sum over the rows o of chunk g of gout[o][n] * in[nbr[o][k]][c]
template <int CIN_P, int COUT_P>
__global__ __launch_bounds__(256) void k_wgrad_mfma(const float* __restrict__ in, int cin,
                                                    const int32_t* __restrict__ nbr, int K,
                                                    const float* __restrict__ gout, int cout,
                                                    float* __restrict__ partial, int Mout,
                                                    int rows_per_chunk) {
  constexpr int LDO = kTM + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Gs = reinterpret_cast<float*>(smem);  // [COUT_P][LDO]   gout tile, transposed
  float* Is = Gs + COUT_P * LDO;                // [CIN_P][LDO]    gathered input tile, transposed
  int* s_nbr = reinterpret_cast<int*>(Is + CIN_P * LDO);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int k = blockIdx.x;
  const int chunk = blockIdx.y;
  constexpr int NTILES = (COUT_P / 16) * (CIN_P / 16);
  constexpr int TPW = (NTILES + 3) / 4;  // tiles per wave
  f32x4 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int r_begin = chunk * rows_per_chunk;
  const int r_end = min(r_begin + rows_per_chunk, Mout);
  for (int row0 = r_begin; row0 < r_end; row0 += kTM) {
    int r = -1;
    if (tid < kTM && row0 + tid < r_end) r = nbr[(size_t)(row0 + tid) * K + k];
    if (tid < kTM) s_nbr[tid] = r;
    if (!__syncthreads_or(r >= 0)) continue;
    // gout tile transposed: Gs[n][o]; rows whose neighbour is missing contribute zero anyway
    // because the matching Is column is zero.
    for (int idx = tid; idx < kTM * COUT_P; idx += 256) {
      const int o = idx / COUT_P, n = idx - o * COUT_P;
      const int row = row0 + o;
      Gs[n * LDO + o] = (row < r_end && n < cout) ? gout[(size_t)row * cout + n] : 0.f;
    }
    for (int idx = tid; idx < kTM * CIN_P; idx += 256) {
      const int o = idx / CIN_P, c = idx - o * CIN_P;
      const int rr = s_nbr[o];
      Is[c * LDO + o] = (rr >= 0 && c < cin) ? in[(size_t)rr * cin + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tile = wave + 4 * t;
      if (tile < NTILES) {
        const int nt = tile / (CIN_P / 16), ct = tile - nt * (CIN_P / 16);
        const float* ga = Gs + (nt * 16 + li) * LDO + 4 * g;
        const float* ib = Is + (ct * 16 + li) * LDO + 4 * g;
#pragma unroll
        for (int ob = 0; ob < kTM / 16; ++ob) {
          const float4 a = *reinterpret_cast<const float4*>(ga + ob * 16);
          const float4 b = *reinterpret_cast<const float4*>(ib + ob * 16);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  float* p = partial + ((size_t)chunk * K + k) * COUT_P * CIN_P;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wave + 4 * t;
    if (tile < NTILES) {
      const int nt = tile / (CIN_P / 16), ct = tile - nt * (CIN_P / 16);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        p[(size_t)(nt * 16 + 4 * g + r) * CIN_P + ct * 16 + li] = acc[t][r];
    }
  }
}

// ---- weight gradient, fp32, row-major staging ------------------------------------------------------------------------
// k_wgrad_mfma above transposes both tiles into LDS element by element (one div / mod, one scalar load and one 8-way
// bank-conflicted ds_write_b32 per element, 64 per thread and tile) and gives every wave single 16 x 16 tiles with no
// fragment reuse: 4.2 ms per 128-channel layer of the LiDAR detector's fp32 step (198 k rows: 42 TFLOP/s; this kernel:
// 2.03 ms = 86 TFLOP/s = 55 % of the fp32 matrix peak).  Here the 64 gout rows and
// the 64 gathered input rows are copied as they lie in memory (16-byte loads and ds_write_b128, rows padded to a stride of
// 16 mod 32 banks) and the fp32 MFMA's operands are read straight out of the row-major tiles: lane (g, li) of
// v_mfma_f32_16x16x4_f32 wants A[i = li][k = g] = gout[row 4s + g][n0 + li] -- 16 consecutive floats of each of two rows per
// half wave, conflict-free at that stride.  Waves form a WN x WC grid over the Cout x Cin tile and reuse their fragments
// ((TN + TC) reads per TN x TC MFMAs).  Same decomposition (offset k, row chunk) and fixed-order chunk reduction as above.
template <int P>
struct WgLd { static constexpr int v = (P % 32 == 0) ? P + 16 : P + 32; };

template <int CIN_P, int COUT_P>
__global__ __launch_bounds__(256) void k_wgrad_rows(const float* __restrict__ in, int cin,
                                                    const int32_t* __restrict__ nbr, int K,
                                                    const float* __restrict__ gout, int cout,
                                                    float* __restrict__ partial, int Mout,
                                                    int rows_per_chunk) {
  constexpr int LDG = WgLd<COUT_P>::v, LDI = WgLd<CIN_P>::v;
  constexpr int NT = COUT_P / 16, CT = CIN_P / 16;
  constexpr int WN = NT >= 2 ? 2 : 1, WC = (4 / WN) < CT ? (4 / WN) : CT;   // wave grid (idle waves on the tiny layers)
  constexpr int TN = NT / WN, TC = CT / WC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Gs = reinterpret_cast<float*>(smem);  // [kTM][LDG]  gout rows
  float* Is = Gs + kTM * LDG;                   // [kTM][LDI]  gathered input rows (zero where the neighbour is missing)
  int* s_nbr = reinterpret_cast<int*>(Is + kTM * LDI);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int k = blockIdx.x, chunk = blockIdx.y;
  const bool active_wave = wave < WN * WC;
  const int wn = wave / WC, wc = wave % WC;
  f32x4 acc[TN][TC];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TC; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int r_begin = chunk * rows_per_chunk;
  const int r_end = min(r_begin + rows_per_chunk, Mout);
  for (int row0 = r_begin; row0 < r_end; row0 += kTM) {
    int r = -1;
    if (tid < kTM && row0 + tid < r_end) r = nbr[(size_t)(row0 + tid) * K + k];
    if (tid < kTM) s_nbr[tid] = r;
    if (!__syncthreads_or(r >= 0)) continue;
    for (int u = tid; u < kTM * (COUT_P / 4); u += 256) {
      const int o = u / (COUT_P / 4), n4 = (u - o * (COUT_P / 4)) * 4;
      const int row = row0 + o;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < r_end && n4 < cout) v = *reinterpret_cast<const float4*>(gout + (size_t)row * cout + n4);
      *reinterpret_cast<float4*>(Gs + o * LDG + n4) = v;
    }
    for (int u = tid; u < kTM * (CIN_P / 4); u += 256) {
      const int o = u / (CIN_P / 4), c4 = (u - o * (CIN_P / 4)) * 4;
      const int rr = s_nbr[o];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rr >= 0 && c4 < cin) v = *reinterpret_cast<const float4*>(in + (size_t)rr * cin + c4);
      *reinterpret_cast<float4*>(Is + o * LDI + c4) = v;
    }
    __syncthreads();
    if (active_wave) {
      const float* ga = Gs + g * LDG + wn * TN * 16 + li;
      const float* ib = Is + g * LDI + wc * TC * 16 + li;
#pragma unroll 4
      for (int s4 = 0; s4 < kTM / 4; ++s4) {      // four rows per MFMA
        float a[TN], b[TC];
#pragma unroll
        for (int t = 0; t < TN; ++t) a[t] = ga[s4 * 4 * LDG + t * 16];
#pragma unroll
        for (int t = 0; t < TC; ++t) b[t] = ib[s4 * 4 * LDI + t * 16];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tc = 0; tc < TC; ++tc)
            acc[tn][tc] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tn], b[tc], acc[tn][tc], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (!active_wave) return;
  float* p = partial + ((size_t)chunk * K + k) * COUT_P * CIN_P;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        p[(size_t)((wn * TN + tn) * 16 + 4 * g + r) * CIN_P + (wc * TC + tc) * 16 + li] = acc[tn][tc][r];
}

// ---- weight gradient, bf16 operands (mixed-precision training) --------------------------------------
// Same decomposition (one workgroup per (offset k, row chunk g), ordered reduction of the chunks), but
// the products run on v_mfma_f32_16x16x32_bf16 -- 16x the rate of the fp32 matrix instruction the exact
// path uses.  The reduction index of this GEMM is the sparse ROW, i.e. both operands (gout[row][n] and
// the gathered in[nbr[row][k]][c]) are K-strided in memory: tiles of 64 rows are staged row-major as
// bf16 and the fragments are read with ds_read_b64_tr_b16 (see csrc/conv2d.hip: inside a 16-lane group
// lane j points at [row k0 + (j>>2)][channel 4*(j&3)..+3] and lane i receives [k0..k0+3][channel i]).
// Row tiles follow the mask-sorted row order of the forward kernel; a per-tile 27-bit activity mask
// (k_tile_masks) lets a workgroup skip the tiles that have no pair for its offset.
typedef short v4s_t __attribute__((ext_vector_type(4)));
constexpr int kWR = 64;                              // rows per step

__global__ __launch_bounds__(256) void k_tile_masks(const int32_t* __restrict__ nbr, int K,
                                                    const int32_t* __restrict__ order, int Mout,
                                                    unsigned* __restrict__ masks, int ntiles) {
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (tile >= ntiles) return;
  const int p = tile * kWR + lane;
  unsigned m = 0u;
  if (p < Mout) {
    const int row = order ? order[p] : p;
    for (int k = 0; k < K; ++k)
      if (nbr[(size_t)row * K + k] >= 0) m |= 1u << k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor((int)m, o);
  if (lane == 0) masks[tile] = m;
}

__device__ __forceinline__ v4s_t tr_issue(unsigned lds_byte_addr) {
  v4s_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(lds_byte_addr));
  return r;
}
__device__ __forceinline__ bf16x8 cat8(v4s_t lo, v4s_t hi) {
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 tr_frag8(const unsigned short* lo_p, const unsigned short* hi_p) {
  const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)lo_p);
  const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)hi_p);
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// Compact the ids of this workgroup's tiles that have a pair for offset k into LDS (one wave, ordered).
// Scanning the tile masks in global memory one dependent load at a time cost more than the MFMAs.
constexpr int kMaxTilesPerChunk = 2048;
__device__ __forceinline__ int build_active_list(const unsigned* __restrict__ masks, int k, int t_begin,
                                                 int t_end, short* list) {
  __shared__ int s_count;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int count = 0;
    for (int base = t_begin; base < t_end; base += 64) {
      const int t = base + lane;
      const bool on = t < t_end && ((masks[t] >> k) & 1u);
      const unsigned long long b = __ballot(on);
      if (on) list[count + __popcll(b & ((1ull << lane) - 1ull))] = (short)(t - t_begin);
      count += __popcll(b);
    }
    if (lane == 0) s_count = count;
  }
  __syncthreads();
  return s_count;
}

template <int CIN_P, int COUT_P, bool BF_IO>   // BF_IO: `in` and `gout` already hold bf16 (cin, cout % 8 == 0)
__global__ __launch_bounds__(256) void k_wgrad_bf16(const float* __restrict__ in, int cin,
                                                    const int32_t* __restrict__ nbr, int K,
                                                    const float* __restrict__ gout, int cout,
                                                    float* __restrict__ partial, int Mout,
                                                    const int32_t* __restrict__ order,
                                                    const unsigned* __restrict__ masks, int ntiles,
                                                    int tiles_per_chunk) {
  constexpr int LDN = COUT_P + 16, LDC = CIN_P + 16;          // bf16 elements per LDS row
  constexpr int NT = COUT_P / 16, CTT = CIN_P / 16;
  constexpr int WM = NT >= 2 ? 2 : 1, WN = 4 / WM;
  constexpr int TI = NT / WM, TJ = (CTT / WN) > 0 ? (CTT / WN) : 1;
  constexpr int NU = (kWR * (COUT_P / 4) + 255) / 256;        // float4 units per thread
  constexpr int CU = (kWR * (CIN_P / 4) + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* Ns = reinterpret_cast<unsigned short*>(smem);      // [2][kWR][LDN]
  unsigned short* Cs = Ns + 2 * kWR * LDN;                             // [2][kWR][LDC]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  const bool wave_on = wn * TJ < CTT;
  const int k = blockIdx.x, chunk = blockIdx.y;
  const int t_begin = chunk * tiles_per_chunk, t_end = min(ntiles, (chunk + 1) * tiles_per_chunk);
  __shared__ short s_list[kMaxTilesPerChunk];
  const int n_active = build_active_list(masks, k, t_begin, t_end, s_list);
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NU8 = (kWR * (COUT_P / 8) + 255) / 256, CU8 = (kWR * (CIN_P / 8) + 255) / 256;
  float4 rn[BF_IO ? 1 : NU], rc[BF_IO ? 1 : CU];
  uint4 hn[BF_IO ? NU8 : 1], hc[BF_IO ? CU8 : 1];
  const unsigned short* in_h = reinterpret_cast<const unsigned short*>(in);
  const unsigned short* gout_h = reinterpret_cast<const unsigned short*>(gout);
  const bool vec_n = (cout & 3) == 0, vec_c = (cin & 3) == 0;
  auto fetch = [&](int t) {
    if constexpr (BF_IO) {
#pragma unroll
      for (int j = 0; j < NU8; ++j) {
        const int u = tid + 256 * j, r = u / (COUT_P / 8), n8 = (u - r * (COUT_P / 8)) * 8;
        const int p = t * kWR + r;
        hn[j] = make_uint4(0u, 0u, 0u, 0u);
        if (r < kWR && p < Mout && n8 < cout)
          hn[j] = *reinterpret_cast<const uint4*>(gout_h + (size_t)(order ? order[p] : p) * cout + n8);
      }
#pragma unroll
      for (int j = 0; j < CU8; ++j) {
        const int u = tid + 256 * j, r = u / (CIN_P / 8), c8 = (u - r * (CIN_P / 8)) * 8;
        const int p = t * kWR + r;
        hc[j] = make_uint4(0u, 0u, 0u, 0u);
        if (r < kWR && p < Mout && c8 < cin) {
          const int rr = nbr[(size_t)(order ? order[p] : p) * K + k];
          if (rr >= 0) hc[j] = *reinterpret_cast<const uint4*>(in_h + (size_t)rr * cin + c8);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        const int u = tid + 256 * j, r = u / (COUT_P / 4), n4 = (u - r * (COUT_P / 4)) * 4;
        const int p = t * kWR + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < kWR && p < Mout) {
          const float* src = gout + (size_t)(order ? order[p] : p) * cout + n4;
          if (vec_n) {
            if (n4 < cout) v = *reinterpret_cast<const float4*>(src);
          } else {
            if (n4 + 0 < cout) v.x = src[0];
            if (n4 + 1 < cout) v.y = src[1];
            if (n4 + 2 < cout) v.z = src[2];
            if (n4 + 3 < cout) v.w = src[3];
          }
        }
        rn[j] = v;
      }
#pragma unroll
      for (int j = 0; j < CU; ++j) {
        const int u = tid + 256 * j, r = u / (CIN_P / 4), c4 = (u - r * (CIN_P / 4)) * 4;
        const int p = t * kWR + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < kWR && p < Mout) {
          const int rr = nbr[(size_t)(order ? order[p] : p) * K + k];
          if (rr >= 0) {
            const float* src = in + (size_t)rr * cin + c4;
            if (vec_c) {
              if (c4 < cin) v = *reinterpret_cast<const float4*>(src);
            } else {
              if (c4 + 0 < cin) v.x = src[0];
              if (c4 + 1 < cin) v.y = src[1];
              if (c4 + 2 < cin) v.z = src[2];
              if (c4 + 3 < cin) v.w = src[3];
            }
          }
        }
        rc[j] = v;
      }
    }
  };
  auto commit = [&](int buf) {
    if constexpr (BF_IO) {
#pragma unroll
      for (int j = 0; j < NU8; ++j) {
        const int u = tid + 256 * j, r = u / (COUT_P / 8), n8 = (u - r * (COUT_P / 8)) * 8;
        if (r < kWR) *reinterpret_cast<uint4*>(Ns + (buf * kWR + r) * LDN + n8) = hn[j];
      }
#pragma unroll
      for (int j = 0; j < CU8; ++j) {
        const int u = tid + 256 * j, r = u / (CIN_P / 8), c8 = (u - r * (CIN_P / 8)) * 8;
        if (r < kWR) *reinterpret_cast<uint4*>(Cs + (buf * kWR + r) * LDC + c8) = hc[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        const int u = tid + 256 * j, r = u / (COUT_P / 4), n4 = (u - r * (COUT_P / 4)) * 4;
        if (r < kWR)
          *reinterpret_cast<uint2*>(Ns + (buf * kWR + r) * LDN + n4) =
              make_uint2(ud_pack_bf16x2(rn[j].x, rn[j].y), ud_pack_bf16x2(rn[j].z, rn[j].w));
      }
#pragma unroll
      for (int j = 0; j < CU; ++j) {
        const int u = tid + 256 * j, r = u / (CIN_P / 4), c4 = (u - r * (CIN_P / 4)) * 4;
        if (r < kWR)
          *reinterpret_cast<uint2*>(Cs + (buf * kWR + r) * LDC + c4) =
              make_uint2(ud_pack_bf16x2(rc[j].x, rc[j].y), ud_pack_bf16x2(rc[j].z, rc[j].w));
      }
    }
  };
  if (n_active > 0) {
    fetch(t_begin + s_list[0]);
    commit(0);
  }
  __syncthreads();
  int buf = 0;
  for (int ai = 0; ai < n_active; ++ai) {
    const bool more = ai + 1 < n_active;
    if (more) fetch(t_begin + s_list[ai + 1]);
    if (wave_on) {
      const unsigned short* nb = Ns + (buf * kWR + 8 * g + (li >> 2)) * LDN + 16 * TI * wm + 4 * (li & 3);
      const unsigned short* cb = Cs + (buf * kWR + 8 * g + (li >> 2)) * LDC + 16 * TJ * wn + 4 * (li & 3);
#pragma unroll
      for (int ks = 0; ks < kWR / 32; ++ks) {
        bf16x8 a[TI];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
          a[ti] = tr_frag8(nb + 32 * ks * LDN + 16 * ti, nb + (32 * ks + 4) * LDN + 16 * ti);
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
          const bf16x8 bb = tr_frag8(cb + 32 * ks * LDC + 16 * tj, cb + (32 * ks + 4) * LDC + 16 * tj);
#pragma unroll
          for (int ti = 0; ti < TI; ++ti)
            acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti], bb, acc[ti][tj], 0, 0, 0);
        }
      }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // partial[chunk][k][n][c] (padded sizes); D layout: column c = li, rows n = 4g + r
  if (wave_on) {
    float* pbase = partial + ((size_t)chunk * K + k) * COUT_P * CIN_P;
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pbase[(size_t)(16 * (TI * wm + ti) + 4 * g + r) * CIN_P + 16 * (TJ * wn + tj) + li] = acc[ti][tj][r];
  }
}

// bf16-IO weight gradient with LDS-DMA staging (cin == CIN_P, cout == COUT_P in {64, 128}): the 64-row
// gout / gathered-input tiles go L2 -> LDS by global_load_lds_dwordx4 into unpadded rows whose 32-byte
// pieces are XOR-swizzled on the source address so that the transposing fragment reads
// (ds_read_b64_tr_b16: 8 rows x 32 B per 32 lanes) hit 8 different pieces; double-buffered, one barrier
// per tile.  Removes the ds_write pass that made the register-staged kernel LDS-bound.
template <int CIN_P, int COUT_P, bool GATHER_G>   // GATHER_G: gout rows are fetched through gorder
__global__ __launch_bounds__(256) void k_wgrad_bf16_dma(const unsigned short* __restrict__ in,
                                                        const int32_t* __restrict__ nbr, int K,
                                                        const unsigned short* __restrict__ gout,
                                                        float* __restrict__ partial, int Mout,
                                                        const int32_t* __restrict__ gorder,
                                                        const unsigned* __restrict__ masks, int ntiles,
                                                        int tiles_per_chunk) {
  constexpr int SN = COUT_P / 8, SC = CIN_P / 8;             // 16-byte slots per row
  constexpr int RN = 64 / SN, RC = 64 / SC;                   // rows per 1-KiB piece
  constexpr int PN = kWR / RN, PC = kWR / RC;                 // pieces per tile
  constexpr int NT = COUT_P / 16, CTT = CIN_P / 16;
  constexpr int TI = NT / 2, TJ = CTT / 2;                    // 2 x 2 waves
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* Ns = reinterpret_cast<unsigned short*>(smem);       // [2][kWR][COUT_P]
  unsigned short* Cs = Ns + 2 * kWR * COUT_P;                           // [2][kWR][CIN_P]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int k = blockIdx.x, chunk = blockIdx.y;
  const int t_begin = chunk * tiles_per_chunk, t_end = min(ntiles, (chunk + 1) * tiles_per_chunk);
  __shared__ short s_list[kMaxTilesPerChunk];
  const int n_active = build_active_list(masks, k, t_begin, t_end, s_list);
  const unsigned short* zero = reinterpret_cast<const unsigned short*>(g_sp_zero16);
  // swizzle of the 32-byte piece index by the row (see the header comment); in 16-byte slot units << 1
  auto fsw = [](int r, int slots) -> int {
    // 256-, 128- and 64-byte rows: the 8 rows x 32 B of a half-wave fragment read land on distinct banks
    return slots == 16 ? (((r & 3) | ((r >> 1) & 4)) << 1)
                       : (slots == 8 ? ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1) : (((r >> 3) & 1) << 1));
  };
  // The rows a tile gathers are known only through an index load (nbr; the host passes rulebook and gout rows
  // already in tile order).  Issued inside the staging it would put a memory latency in front of every
  // tile's DMA; it is issued one tile AHEAD instead, before this iteration's DMAs (so that nothing waits on
  // them), and lands while the tile is multiplied.
  struct TileIdx {
    int t;              // tile
    int rr[PC / 4];     // gathered input rows of this lane's pieces as loaded (validity is re-derived from t:
                        // nothing may consume the loaded values before the next iteration)
    int og[GATHER_G ? PN / 4 : 1];   // gout rows (GATHER_G)
  };
  auto load_idx = [&](int t, TileIdx& ix) {
    ix.t = t;
    if (GATHER_G) {
#pragma unroll
      for (int j = 0; j < PN / 4; ++j)
        ix.og[j] = gorder[min(t * kWR + (wave + 4 * j) * RN + lane / SN, Mout - 1)];
    }
#pragma unroll
    for (int j = 0; j < PC / 4; ++j) {
      const int p = t * kWR + (wave + 4 * j) * RC + lane / SC;
      ix.rr[j] = nbr[(size_t)min(p, Mout - 1) * K + k];         // branch-free: all loads issue back to back
    }
  };
  auto stage = [&](const TileIdx& ix, int buf) {
#pragma unroll
    for (int j = 0; j < PN / 4; ++j) {
      const int piece = wave + 4 * j, r = piece * RN + lane / SN, slot = lane % SN;
      const int p = ix.t * kWR + r;
      const unsigned short* src = zero;
      if (p < Mout) src = gout + (size_t)(GATHER_G ? ix.og[j] : p) * COUT_P + ((slot ^ fsw(r, SN)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Ns + (buf * kWR + piece * RN) * COUT_P),
                                       16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PC / 4; ++j) {
      const int piece = wave + 4 * j, r = piece * RC + lane / SC, slot = lane % SC;
      const int p = ix.t * kWR + r;
      const unsigned short* src = zero;
      if (p < Mout && ix.rr[j] >= 0) src = in + (size_t)ix.rr[j] * CIN_P + ((slot ^ fsw(r, SC)) << 3);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(Cs + (buf * kWR + piece * RC) * CIN_P),
                                       16, 0, 0);
    }
  };
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  TileIdx ix, nxt;
  if (n_active > 0) {
    load_idx(t_begin + s_list[0], ix);
    stage(ix, 0);
  }
  if (n_active > 1) load_idx(t_begin + s_list[1], ix);
  __syncthreads();
  int buf = 0;
  const int sub = (li & 3) >> 1, half = (li & 1) << 2;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  for (int ai = 0; ai < n_active; ++ai) {
    if (ai + 2 < n_active) load_idx(t_begin + s_list[ai + 2], nxt);  // lands while this tile is multiplied
    if (ai + 1 < n_active) stage(ix, buf ^ 1);
    {
      // transposing fragment read: lane li of a 16-lane group points at [row k0 + (li>>2)][channels c0 + 4*(li&3) ..+3].
      // Issued as inline asm + an explicit lgkmcnt wait tied to the fragment registers: the compiler makes
      // the ds_read_tr builtin wait for ALL outstanding LDS-DMA (vmcnt(0)), which would serialise the next
      // tile's DMAs behind this tile's MFMAs.
      const unsigned nbuf = lds0 + buf * (kWR * COUT_P * 2);
      const unsigned cbuf = lds0 + (2 * kWR * COUT_P + buf * kWR * CIN_P) * 2;
#pragma unroll
      for (int ks = 0; ks < kWR / 32; ++ks) {
        const int r0 = 32 * ks + 8 * g + (li >> 2), r1 = r0 + 4;
        v4s_t al[TI], ah[TI], bl[TJ], bh[TJ];
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
          const int slot = 2 * (TI * wm + ti) + sub;
          al[ti] = tr_issue(nbuf + 2 * (r0 * COUT_P + ((slot ^ fsw(r0, SN)) << 3) + half));
          ah[ti] = tr_issue(nbuf + 2 * (r1 * COUT_P + ((slot ^ fsw(r1, SN)) << 3) + half));
        }
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
          const int slot = 2 * (TJ * wn + tj) + sub;
          bl[tj] = tr_issue(cbuf + 2 * (r0 * CIN_P + ((slot ^ fsw(r0, SC)) << 3) + half));
          bh[tj] = tr_issue(cbuf + 2 * (r1 * CIN_P + ((slot ^ fsw(r1, SC)) << 3) + half));
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
    }
    __syncthreads();
    buf ^= 1;
    ix = nxt;
  }
  float* pbase = partial + ((size_t)chunk * K + k) * COUT_P * CIN_P;
#pragma unroll
  for (int ti = 0; ti < TI; ++ti)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pbase[(size_t)(16 * (TI * wm + ti) + 4 * g + r) * CIN_P + 16 * (TJ * wn + tj) + li] = acc[ti][tj][r];
}

// gW[n][k][c] (KRSC, dense) = sum over chunks in order of partial[g][k][n][c]
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ partial, int G,
                                                      int K, int cinp, int coutp, int cin, int cout,
                                                      float* __restrict__ gW) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)cout * K * cin) return;
  const int c = (int)(t % cin);
  const int k = (int)((t / cin) % K);
  const int n = (int)(t / ((long long)cin * K));
  float acc = 0.f;
  for (int gi = 0; gi < G; ++gi)
    acc += partial[(((size_t)gi * K + k) * coutp + n) * cinp + c];
  gW[t] = acc;
}

__global__ __launch_bounds__(256) void k_wgrad_generic(const float* __restrict__ in, int cin,
                                                       const int32_t* __restrict__ nbr, int K,
                                                       const float* __restrict__ gout, int cout,
                                                       float* __restrict__ gW, int Mout) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)cout * K * cin) return;
  const int c = (int)(t % cin);
  const int k = (int)((t / cin) % K);
  const int n = (int)(t / ((long long)cin * K));
  float acc = 0.f;
  for (int o = 0; o < Mout; ++o) {
    const int r = nbr[(size_t)o * K + k];
    if (r >= 0) acc = fmaf(gout[(size_t)o * cout + n], in[(size_t)r * cin + c], acc);
  }
  gW[t] = acc;
}


// ---- v3 (fp32, 64 / 128 channels, channel-contiguous weights): LDS-DMA, double-buffered 64-channel stages ---------------------
// k_conv_mfma_v2 above commits the gathered rows and the offset's weights through registers into ONE 153 KB tile: its waves
// spend 41-45 % of their time in the commit -> barrier -> MFMA -> barrier hand-over (81 TFLOP/s on the 128-channel layers).
// Here a stage = (active offset, 64-channel half of Cin): 128 gathered row pieces (256 B each) + Cout weight row pieces,
// moved L2 -> LDS by global_load_lds (per-lane source addresses: the gather costs no VGPR round trip, no ds_write), into one of
// TWO stage buffers: stage s + 1 flies while stage s is multiplied, one barrier per stage.  Rows are unpadded 256 B = 16
// 16-byte slots, XOR-swizzled with (row & 15) on the SOURCE address, so the b128 fragment reads (16 lanes = 16 rows, same
// logical slot) touch 16 distinct slots.  Waves: 4 (32 rows) x 2 (Cout / 2 columns); per 16-channel group a wave reads 2 A + NT B
// fragments for 8 NT MFMAs (a lane's 16-byte piece feeds four MFMAs, see conv2d_f32.hip).
constexpr unsigned kDmaOob = 0xFFFF0000u;      // size of the buffer descriptors of k_conv_dma_f32 = the byte offset that reads zeros
// LDS-DMA of 16 bytes per lane through a raw buffer descriptor: LDS[lds_wave_base + lane * 16] = buffer[voff + soff]; an offset past
// the descriptor's size reads zeros.  (A __device__ function: the builtin does not exist in the host pass, and a lambda of a
// kernel is compiled for both sides -- called from the lambda directly it silently drops the kernel's host stub.)
__device__ __forceinline__ void sp_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(size_t)lds_wave_base, 16, voff, soff, 0, 0);
}

// (A three-buffer ring for the 64 -> 64 layers -- two stages in flight under a counted vmcnt -- was measured in round 4 and
// removed in round 5: 670 us against k_conv_mfma_v2's 620 us on the step's four layers.)
template <int CIN, int COUT, int NBUF>
__global__ __launch_bounds__(512) void k_conv_dma_f32(const float* __restrict__ in, const int32_t* __restrict__ nbr, int K,
                                                      int mirror, const float* __restrict__ W, WStrides ws,
                                                      const float* __restrict__ bias, float* __restrict__ out, int Mout,
                                                      const int32_t* __restrict__ order, ConvEpilogue ep) {
  constexpr int NT = COUT / 32;                      // 16-column tiles per wave (2 column halves)
  constexpr int kAB = kTM2 * 256, kWB = COUT * 256, kStage = kAB + kWB;
  constexpr int kHalves = CIN / 64;
  constexpr int kAPieces = kTM2 / 4, kWPieces = COUT / 4;      // 1-KiB DMA pieces: 4 rows x 256 B
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_nbr = reinterpret_cast<int*>(smem + NBUF * kStage);   // [K][kTM2]
  unsigned& s_active = *reinterpret_cast<unsigned*>(s_nbr + K * kTM2);
  int* s_row = s_nbr + K * kTM2 + 4;                            // [kTM2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, li = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = (int)(gridDim.x - 1u - blockIdx.x) * kTM2;   // mask-sorted rows: the tiles with the most active offsets sit at the end -> dispatch them first (longest first)
  if (tid == 0) s_active = 0u;
  if (tid < kTM2) {
    const int p = row0 + tid;
    s_row[tid] = (p < Mout) ? (order ? order[p] : p) : -1;
  }
  __syncthreads();
  unsigned mine = 0u;
  for (int idx = tid; idx < kTM2 * K; idx += 512) {
    const int r = idx / K, k = idx - r * K;
    int v = -1;
    const int orow = s_row[r];
    if (orow >= 0) v = nbr[(size_t)orow * K + (mirror ? K - 1 - k : k)];
    s_nbr[k * kTM2 + r] = v;
    if (v >= 0) mine |= 1u << k;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine |= __shfl_xor((int)mine, o);
  if (lane == 0 && mine) atomicOr(&s_active, mine);
  __syncthreads();
  const unsigned active = s_active;
  unsigned wmask = 0u;                                 // offsets any of this wave's 32 rows uses
  for (int k = 0; k < K; ++k) {                        // (skipping per 16-row tile instead measured no faster)
    const int v = (lane < 32) ? s_nbr[k * kTM2 + wm * 32 + lane] : -1;
    if (__any(v >= 0)) wmask |= 1u << k;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // DMA of one stage: pieces are dealt to the 8 waves round-robin; lane -> (row 4 p + g, LDS slot li <- source slot li ^ (row & 15)).
  // Sources go through raw buffer descriptors (round 6): a lane's byte offset is ONE shift-add of the gathered row id (rows are
  // CIN * 4 = 256 / 512 bytes) -- or an out-of-range offset, which reads zeros, for a missing neighbour -- and the (offset, half)
  // part of a weight address rides in the scalar offset; the per-lane weight offsets are fixed for the whole launch.  With
  // 64-bit pointer arithmetic a stage cost every wave ~90 vector + ~70 scalar instructions beside its 128 MFMAs, and on this
  // chip every issued instruction of an fp32-MFMA kernel is paid in MFMA issue time (SQ counters, profiles/r06_spconv_encoder.md).
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)kDmaOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)kDmaOob, 0x00020000);
  const unsigned swz = (unsigned)((li ^ ((4 * wave + g) & 15)) << 4);      // (4 (wave + 8 i) + g) & 15 does not depend on i
  unsigned wvo[kWPieces / 8];
#pragma unroll
  for (int i = 0; i < kWPieces / 8; ++i) wvo[i] = (unsigned)(4 * (wave + 8 * i) + g) * (unsigned)ws.sn * 4u + swz;
  auto stage = [&](int k, int half, int buf) {
    const unsigned sb = (unsigned)(size_t)smem + buf * kStage;
    int rr[kAPieces / 8];
    {
      const unsigned na = NBUF * kStage + (unsigned)(k * kTM2 + 4 * wave + g) * 4u;              // one LDS round trip, not four
#pragma unroll
      for (int i = 0; i < kAPieces / 8; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rr[i]) : "v"(na), "n"(i * 128));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < kAPieces / 8; ++i) asm volatile("" : "+v"(rr[i]));
    }
    const unsigned sa = (unsigned)half * 256u;
    const unsigned sw = __builtin_amdgcn_readfirstlane((unsigned)k * (unsigned)ws.sk * 4u + (unsigned)half * 256u);
#pragma unroll
    for (int i = 0; i < kAPieces / 8; ++i) {
      const int p = wave + 8 * i;
      const unsigned vo = rr[i] >= 0 ? (unsigned)rr[i] * (unsigned)(CIN * 4) + swz : kDmaOob;
      sp_dma16(rin, vo, sa, sb + p * 1024);
    }
#pragma unroll
    for (int i = 0; i < kWPieces / 8; ++i) {
      const int p = wave + 8 * i;
      sp_dma16(rw, wvo[i], sw, sb + kAB + p * 1024);
    }
  };
  // fragment byte offsets inside a stage buffer, per 16-channel group cb: slot (4 cb + g) ^ li of row (tile * 16 + li)
  int fo[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) fo[cb] = li * 256 + (((4 * cb + g) ^ li) << 4);

  // stage list: (active offset, half) in order; m walks the offsets (its lowest set bit = the stage's offset)
  auto succ = [&](int& k, int& half, unsigned& m) {
    if (++half == kHalves) {
      half = 0;
      m &= m - 1;
      k = m ? (__ffs((int)m) - 1) : -1;
    }
  };
  unsigned m0 = active;
  int k = m0 ? (__ffs((int)m0) - 1) : -1, half = 0, buf = 0;
  int k1 = k, h1 = half;
  unsigned m1 = m0;
  if (k >= 0) {
    stage(k, 0, 0);
    succ(k1, h1, m1);
  }
  while (k >= 0) {
    // the stage after the next one
    int k2 = k1, h2 = h1;
    unsigned m2 = m1;
    if (k1 >= 0) succ(k2, h2, m2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // raw barrier: __syncthreads() is lowered to `s_waitcnt vmcnt(0) lgkmcnt(0)` + s_barrier, which would drain the stage in
    // flight.  The waves' own LDS reads (inline asm) were awaited with lgkmcnt(0) before their last MFMA group.
    __builtin_amdgcn_s_barrier();          // this stage has landed for everybody; everybody is done with the previous stage's buffer
    asm volatile("" ::: "memory");
    if (k1 >= 0) stage(k1, h1, buf ^ 1);
    if ((wmask >> k) & 1u) {
      // fragment reads by hand, one 16-channel group ahead of its MFMAs: hipcc puts `s_waitcnt vmcnt(0)` in front of a compiled LDS
      // read that follows an LDS-DMA issue -- i.e. it waited here for the NEXT stage's DMA issued four lines up, and the double
      // buffer bought nothing
      const unsigned ab = buf * kStage + (wm * 32) * 256, wb = buf * kStage + kAB + (wn * (COUT / 2)) * 256;
      f32x4 fa[2][2], fbv[2][NT];
#define UD_SD_LOADS(BUF, CB)                                                                                            \
  do {                                                                                                                  \
    const unsigned va_ = ab + fo[CB], vb_ = wb + fo[CB];                                                                \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                       \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[BUF][i]) : "v"(va_), "n"(i * 4096));                       \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                                      \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbv[BUF][t]) : "v"(vb_), "n"(t * 4096));                      \
  } while (0)
      UD_SD_LOADS(0, 0);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const int cur = cb & 1;
        if (cb + 1 < 4) {
          if (cur == 0) UD_SD_LOADS(1, cb + 1); else UD_SD_LOADS(0, cb + 1);
          if (NT == 4) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(fa[cur][i]));       // the values are valid from here on
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(fbv[cur][t]));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[cur][i][e], fbv[cur][t][e], acc[i][t], 0, 0, 0);
      }
#undef UD_SD_LOADS
    }
    k = k1, half = h1, m0 = m1;
    k1 = k2, h1 = h2, m1 = m2;
    buf = buf + 1 == NBUF ? 0 : buf + 1;
  }
  // epilogue: acc[i][t][r] = out[row(wm * 32 + 16 i + 4 g + r)][wn * COUT / 2 + 16 t + li]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = wn * (COUT / 2) + t * 16 + li;
    const float bv = bias ? bias[col] : 0.f;
    const float sc = ep.scale ? ep.scale[col] : 1.f;
    const float sh = ep.shift ? ep.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = s_row[wm * 32 + 16 * i + 4 * g + r];
        if (row >= 0) {
          float v = acc[i][t][r] + bv;
          if (ep.scale) v = v * sc + sh;
          if (ep.residual) v += ep.residual[(size_t)row * COUT + col];
          if (ep.relu) v = fmaxf(v, 0.f);
          out[(size_t)row * COUT + col] = v;
        }
      }
  }
}

template <int CIN, int COUT>
int launch_conv_dma_f32(const float* in, const int32_t* nbr, int K, int mirror, const float* W, WStrides ws,
                        const float* bias, float* out, int Mout, const int32_t* order, ConvEpilogue ep,
                        hipStream_t stream) {
  constexpr int NBUF = 2;
  const size_t lds = NBUF * (size_t)(kTM2 * 256 + COUT * 256) + (size_t)K * kTM2 * sizeof(int) + 16 + kTM2 * sizeof(int);
  if (lds > 160 * 1024) return UD_ERR_UNSUPPORTED;
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = attr_set.pending()) {
    // the request depends on the runtime K (K * kTM2 ints): reserve the chip's 160 KiB once, so a K = 3 layer launched first
    // does not pin the limit below what a later K = 27 layer needs
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_dma_f32<CIN, COUT, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
    attr_set.mark(attr_set_bit);
  }
  k_conv_dma_f32<CIN, COUT, NBUF><<<ud_div_up(Mout, kTM2), 512, lds, stream>>>(in, nbr, K, mirror, W, ws, bias, out, Mout, order, ep);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

inline int pad16(int c) { return (c + 15) / 16 * 16; }

template <int CIN_P, int COUT_P>
int launch_conv_v2(const float* in, int cin, const int32_t* nbr, int K, int mirror, const float* W,
                   WStrides ws, const float* bias, float* out, int cout, int Mout,
                   const int32_t* order, ConvEpilogue ep, hipStream_t stream) {
  const size_t lds = (size_t)(kTM2 + COUT_P) * (CIN_P + 8) * sizeof(float) + (size_t)K * kTM2 * sizeof(int) + 16 + kTM2 * sizeof(int);
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = (lds > 64 * 1024) ? attr_set.pending() : 0ull) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_mfma_v2<CIN_P, COUT_P, true, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_mfma_v2<CIN_P, COUT_P, true, false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_mfma_v2<CIN_P, COUT_P, false, false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_set_bit);
  }
  const bool va = (cin & 3) == 0, vb = va && ws.sc == 1;
  const dim3 grid(ud_div_up(Mout, kTM2));
  if (vb)
    k_conv_mfma_v2<CIN_P, COUT_P, true, true><<<grid, 512, lds, stream>>>(in, cin, nbr, K, mirror, W, ws, bias, out, cout,
                                                                          Mout, order, ep);
  else if (va)
    k_conv_mfma_v2<CIN_P, COUT_P, true, false><<<grid, 512, lds, stream>>>(in, cin, nbr, K, mirror, W, ws, bias, out, cout,
                                                                           Mout, order, ep);
  else
    k_conv_mfma_v2<CIN_P, COUT_P, false, false><<<grid, 512, lds, stream>>>(in, cin, nbr, K, mirror, W, ws, bias, out,
                                                                            cout, Mout, order, ep);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

template <int CIN_P, int COUT_P>
int launch_conv_bf16(const float* in, int cin, const int32_t* nbr, int K, int mirror, const float* W,
                     WStrides ws, const float* bias, float* out, int cout, int Mout,
                     const int32_t* order, ConvEpilogue ep, int io, hipStream_t stream) {
  constexpr int CP = CIN_P < 32 ? 32 : CIN_P;
  if ((io & 1) && (cin & 3)) return UD_ERR_UNSUPPORTED;          // bf16 rows are read 4 channels at a time
  if ((io & 4) && !(ws.sc == 1 && (cin & 3) == 0)) return UD_ERR_UNSUPPORTED;
  if (io == 7 && cin == CP && (cout & 7) == 0 && (ws.sn & 7) == 0 && (ws.sk & 7) == 0) {
    const size_t lds_d = dma_front_bytes(CP, COUT_P) + (size_t)K * kTM2 * sizeof(int) + 16 + kTM2 * sizeof(int);
    if (lds_d <= 160 * 1024) {
      static UdDeviceOnce dma_attr_set;
      if (const unsigned long long dma_attr_set_bit = (lds_d > 64 * 1024) ? dma_attr_set.pending() : 0ull) {
        UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_mfma_bf16_dma<CP, COUT_P>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
        dma_attr_set.mark(dma_attr_set_bit);
      }
      k_conv_mfma_bf16_dma<CP, COUT_P><<<ud_div_up(Mout, kTM2), 512, lds_d, stream>>>(
          reinterpret_cast<const unsigned short*>(in), cin, nbr, K, mirror,
          reinterpret_cast<const unsigned short*>(W), ws, bias, reinterpret_cast<unsigned short*>(out), cout,
          Mout, order, ep);
      UD_LAUNCH_CHECK();
      return UD_OK;
    }
  }
  if (io == 7 && (cin & 7) == 0 && (cout & 7) == 0 && (ws.sn & 7) == 0 && (ws.sk & 7) == 0) {
    const size_t lds_f = fast_front_bytes(CP, COUT_P) + (size_t)K * kTM2 * sizeof(int) + 16 +
                         kTM2 * sizeof(int);
    static UdDeviceOnce fast_attr_set;
    if (const unsigned long long fast_attr_set_bit = (lds_f > 64 * 1024) ? fast_attr_set.pending() : 0ull) {
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_mfma_bf16_fast<CP, COUT_P>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
      fast_attr_set.mark(fast_attr_set_bit);
    }
    k_conv_mfma_bf16_fast<CP, COUT_P><<<ud_div_up(Mout, kTM2), 512, lds_f, stream>>>(
        reinterpret_cast<const unsigned short*>(in), cin, nbr, K, mirror,
        reinterpret_cast<const unsigned short*>(W), ws, bias, reinterpret_cast<unsigned short*>(out), cout,
        Mout, order, ep);
    UD_LAUNCH_CHECK();
    return UD_OK;
  }
  const size_t lds = (size_t)(kTM2 + COUT_P) * (CP + 8) * sizeof(unsigned short) +
                     (size_t)K * kTM2 * sizeof(int) + 16 + kTM2 * sizeof(int);
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = (lds > 64 * 1024) ? attr_set.pending() : 0ull) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_mfma_bf16<CP, COUT_P>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_set_bit);
  }
  k_conv_mfma_bf16<CP, COUT_P><<<ud_div_up(Mout, kTM2), 512, lds, stream>>>(
      in, cin, nbr, K, mirror, W, ws, bias, out, cout, Mout, order, ep, io);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

template <int CIN_P, int COUT_P>
int launch_conv(const float* in, int cin, const int32_t* nbr, int K, int mirror, const float* W,
                WStrides ws, const float* bias, float* out, int cout, int Mout,
                const int32_t* order, ConvEpilogue ep, int algo, hipStream_t stream) {
  if (K <= 32 && algo == 3)
    return launch_conv_bf16<CIN_P, COUT_P>(in, cin, nbr, K, mirror, W, ws, bias, out, cout, Mout, order, ep, 0, stream);
  // fp32, 64 / 128 channels exactly, channel-contiguous 16-byte-aligned weights: the LDS-DMA kernel
  // 64 -> 64 stays on k_conv_mfma_v2: a stage there is 64 MFMAs per wave, shorter than the DMA round trip and than the skew of
  // an 8-wave barrier.  Measured on the step's four 64 -> 64 layers (394 k rows, 5.45 M pairs): v2 620 us, this kernel with two
  // stage buffers 700 us (round 3), with a three-stage ring 670 us (round 4; removed), with the round-6 descriptor addressing 660 us.
  if constexpr ((CIN_P == 64 || CIN_P == 128) && (COUT_P == 64 || COUT_P == 128) && !(CIN_P == 64 && COUT_P == 64)) {
    static const bool no_dma = getenv("UD_SPCONV_NO_DMA") != nullptr;       // A/B timing
    // (32-bit byte offsets into W through a buffer descriptor; the INPUT tensor must be smaller than 4 GiB - 64 KiB as well --
    // the entry point does not know its row count: include/unidistill_hip.h states the limit, ops/spconv.py enforces it)
    const long long w_span = ((long long)(cout - 1) * ws.sn + (long long)(K - 1) * ws.sk + cin) * 4;
    if (!no_dma && K <= 32 && algo == 0 && cin == CIN_P && cout == COUT_P && ws.sc == 1 && (ws.sn & 3) == 0 && (ws.sk & 3) == 0 &&
        ws.sn >= 0 && ws.sk >= 0 && w_span < (long long)kDmaOob && ((size_t)W & 15) == 0 && ((size_t)in & 15) == 0)
      return launch_conv_dma_f32<CIN_P, COUT_P>(in, nbr, K, mirror, W, ws, bias, out, Mout, order, ep, stream);
  }
  // activity masks are 32-bit; the LDS rulebook slice must fit next to the tiles
  if (K <= 32 && algo != 2)
    return launch_conv_v2<CIN_P, COUT_P>(in, cin, nbr, K, mirror, W, ws, bias, out, cout, Mout, order, ep, stream);
  if (ep.scale || ep.residual || ep.relu) return UD_ERR_UNSUPPORTED;
  const size_t lds = (size_t)(kTM + COUT_P) * (CIN_P + 4) * sizeof(float) + kTM * sizeof(int);
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = (lds > 64 * 1024) ? attr_set.pending() : 0ull) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_conv_mfma<CIN_P, COUT_P>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_set_bit);
  }
  k_conv_mfma<CIN_P, COUT_P><<<ud_div_up(Mout, kTM), 256, lds, stream>>>(
      in, cin, nbr, K, mirror, W, ws, bias, out, cout, Mout);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

template <int CIN_P, int COUT_P>
int launch_wgrad(const float* in, int cin, const int32_t* nbr, int K, const float* gout, int cout,
                 float* gW, int Mout, float* partial, int G, int rows_per_chunk,
                 hipStream_t stream) {
  dim3 grid(K, G);
  if (cin % 4 == 0 && cout % 4 == 0) {        // 16-byte row copies (every layer but the 5-channel input conv)
    const size_t lds = (size_t)kTM * (WgLd<CIN_P>::v + WgLd<COUT_P>::v) * sizeof(float) + kTM * sizeof(int);
    static UdDeviceOnce attr_set;
    if (const unsigned long long attr_set_bit = (lds > 64 * 1024) ? attr_set.pending() : 0ull) {
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wgrad_rows<CIN_P, COUT_P>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set.mark(attr_set_bit);
    }
    k_wgrad_rows<CIN_P, COUT_P><<<grid, 256, lds, stream>>>(in, cin, nbr, K, gout, cout, partial, Mout, rows_per_chunk);
  } else {
    const size_t lds = (size_t)(CIN_P + COUT_P) * (kTM + 4) * sizeof(float) + kTM * sizeof(int);
    static UdDeviceOnce attr_set;
    if (const unsigned long long attr_set_bit = (lds > 64 * 1024) ? attr_set.pending() : 0ull) {
      UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wgrad_mfma<CIN_P, COUT_P>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set.mark(attr_set_bit);
    }
    k_wgrad_mfma<CIN_P, COUT_P><<<grid, 256, lds, stream>>>(in, cin, nbr, K, gout, cout, partial,
                                                            Mout, rows_per_chunk);
  }
  UD_LAUNCH_CHECK();
  k_wgrad_reduce<<<ud_div_up((long long)cout * K * cin, 256), 256, 0, stream>>>(
      partial, G, K, CIN_P, COUT_P, cin, cout, gW);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

#define UD_CONV_CASES(X) \
  X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) X(64, 64) X(64, 128) X(128, 64) X(128, 128)

int wgrad_chunks(int Mout, int* rows_per_chunk) {
  // K x G workgroups: with K = 27 offsets, ~40 row chunks give ~1 000 workgroups (two rounds of two per CU); at least
  // 512 rows (8 tiles) per chunk.  (The first version used >= 4 096 rows per chunk: 216 workgroups for a 30 k-row layer.)
  int G = (Mout + 511) / 512;
  if (G > 40) G = 40;
  if (G < 1) G = 1;
  int rpc = (Mout + G - 1) / G;
  rpc = (rpc + kTM - 1) / kTM * kTM;
  *rows_per_chunk = rpc;
  return (Mout + rpc - 1) / rpc > 0 ? (Mout + rpc - 1) / rpc : 1;
}

}  // namespace

// out f32[Mout,Cout] = bias + implicit-GEMM over nbr i32[Mout,K] of in f32[Min,Cin] with weights
// addressed as W[n*w_sn + k*w_sk + c*w_sc] (n < Cout, k < K, c < Cin).  mirror != 0 reads rulebook
// column K-1-k for weight offset k (submanifold dgrad).  algo: 0 = auto (MFMA when the padded
// channel pair is instantiated), 1 = force the generic VALU kernel.
extern "C" int ud_spconv_conv(const float* in, const int32_t* nbr, const float* W, int64_t w_sn,
                              int64_t w_sk, int64_t w_sc, int mirror, const float* bias,
                              float* out, int Mout, int K, int Cin, int Cout, int algo,
                              const int32_t* row_order, const float* ep_scale,
                              const float* ep_shift, const float* ep_residual, int ep_relu,
                              ud_stream_t stream_) {
  if (Mout < 0 || K <= 0 || Cin <= 0 || Cout <= 0) return UD_ERR_INVALID_ARG;
  if (Mout == 0) return UD_OK;
  if (!in || !nbr || !W || !out) return UD_ERR_INVALID_ARG;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return UD_ERR_INVALID_ARG;
  ConvEpilogue ep{ep_scale, ep_shift, ep_residual, ep_relu};
  hipStream_t stream = (hipStream_t)stream_;
  WStrides ws{w_sn, w_sk, w_sc};
  const int cp = pad16(Cin), np = pad16(Cout);
  UdProfScope prof("spconv.k_conv", stream);
  if (algo == 0 || algo == 2 || algo == 3) {   // 0: fp32 MFMA, 2: first-generation kernel, 3: bf16 MFMA
#define X(A, B) \
  if (cp == A && np == B) \
    return launch_conv<A, B>(in, Cin, nbr, K, mirror, W, ws, bias, out, Cout, Mout, row_order, ep, algo, stream);
    UD_CONV_CASES(X)
#undef X
  }
  if (ep.scale || ep.residual || ep.relu) return UD_ERR_UNSUPPORTED;
  k_conv_generic<<<ud_div_up((long long)Mout * Cout, 256), 256, 0, stream>>>(
      in, Cin, nbr, K, mirror, W, ws, bias, out, Cout, Mout);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// Mixed-precision inference variant of ud_spconv_conv: the bf16 MFMA kernel with bf16 tensors in HBM.
// io_flags bit 0: in is bf16 [*, Cin]; bit 1: out and ep_residual are bf16 [Mout, Cout]; bit 2: W is bf16.
extern "C" int ud_spconv_conv_bf16io(const void* in, const int32_t* nbr, const void* W, int64_t w_sn,
                                     int64_t w_sk, int64_t w_sc, int mirror, const float* bias,
                                     void* out, int Mout, int K, int Cin, int Cout, int io_flags,
                                     const int32_t* row_order, const float* ep_scale,
                                     const float* ep_shift, const void* ep_residual, int ep_relu,
                                     ud_stream_t stream_) {
  if (Mout < 0 || K <= 0 || K > 32 || Cin <= 0 || Cout <= 0 || (io_flags & ~7)) return UD_ERR_INVALID_ARG;
  if (Mout == 0) return UD_OK;
  if (!in || !nbr || !W || !out) return UD_ERR_INVALID_ARG;
  if ((ep_scale == nullptr) != (ep_shift == nullptr)) return UD_ERR_INVALID_ARG;
  ConvEpilogue ep{ep_scale, ep_shift, reinterpret_cast<const float*>(ep_residual), ep_relu};
  hipStream_t stream = (hipStream_t)stream_;
  WStrides ws{w_sn, w_sk, w_sc};
  const int cp = pad16(Cin), np = pad16(Cout);
  UdProfScope prof("spconv.k_conv", stream);
#define X(A, B)             \
  if (cp == A && np == B)   \
    return launch_conv_bf16<A, B>(reinterpret_cast<const float*>(in), Cin, nbr, K, mirror,           \
                                  reinterpret_cast<const float*>(W), ws, bias,                        \
                                  reinterpret_cast<float*>(out), Cout, Mout, row_order, ep, io_flags, \
                                  stream);
  UD_CONV_CASES(X)
#undef X
  return UD_ERR_UNSUPPORTED;
}

extern "C" size_t ud_spconv_wgrad_workspace_bytes(int Mout, int K, int Cin, int Cout) {
  if (Mout < 0 || K <= 0 || Cin <= 0 || Cout <= 0) return 0;
  int rpc;
  const int G = wgrad_chunks(Mout > 0 ? Mout : 1, &rpc);
  return ud_align_up((size_t)G * K * pad16(Cin) * pad16(Cout) * sizeof(float));
}

namespace {
// Row chunks per offset: many small (offset, chunk) units balance the very different number of active
// tiles per offset (the centre offset touches every tile, corner offsets a third of them); bounded by
// the size of the ordered partial-sum buffer (<= 128 MiB).
int wgrad_bf16_max_chunks(int K, int cin_p, int cout_p) {
  const long long per = (long long)K * cin_p * cout_p * (long long)sizeof(float);
  long long g = (128ll << 20) / (per > 0 ? per : 1);
  if (g > 64) g = 64;
  if (g < 8) g = 8;
  return (int)g;
}
int wgrad_bf16_chunks(int ntiles, int K, int cin_p, int cout_p, int* tiles_per_chunk) {
  int G = wgrad_bf16_max_chunks(K, cin_p, cout_p);
  if (G > ntiles) G = ntiles;
  if (G < 1) G = 1;
  *tiles_per_chunk = (ntiles + G - 1) / G;
  if (*tiles_per_chunk > kMaxTilesPerChunk) return -1;      // caller falls back (never at real sizes)
  return (ntiles + *tiles_per_chunk - 1) / *tiles_per_chunk;
}

template <int CIN_P, int COUT_P>
int launch_wgrad_bf16(const float* in, int cin, const int32_t* nbr, int K, const float* gout, int cout,
                      float* gW, int Mout, const int32_t* order, float* partial, unsigned* masks,
                      const unsigned* given_masks, int io_bf16, hipStream_t stream) {
  const int ntiles = ud_div_up(Mout, kWR);
  int tpc;
  const int G = wgrad_bf16_chunks(ntiles, K, CIN_P, COUT_P, &tpc);
  if (G < 0) return UD_ERR_UNSUPPORTED;
  if (given_masks) {
    masks = const_cast<unsigned*>(given_masks);
  } else {
    k_tile_masks<<<ud_div_up(ntiles, 4), 256, 0, stream>>>(nbr, K, order, Mout, masks, ntiles);
    UD_LAUNCH_CHECK();
  }
  const size_t lds = (size_t)2 * kWR * (CIN_P + COUT_P + 32) * sizeof(unsigned short);
  static UdDeviceOnce attr_set;
  if (const unsigned long long attr_set_bit = (lds > 64 * 1024) ? attr_set.pending() : 0ull) {
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wgrad_bf16<CIN_P, COUT_P, false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wgrad_bf16<CIN_P, COUT_P, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set.mark(attr_set_bit);
  }
  if constexpr ((CIN_P == 32 || CIN_P == 64 || CIN_P == 128) && (COUT_P == 32 || COUT_P == 64 || COUT_P == 128)) {
    // the DMA kernel walks nbr in tile order: either nothing is permuted (order NULL) or the caller passes a
    // rulebook already in row_order (io bit 1) and row_order only locates the gout rows
    const bool presorted = (io_bf16 & 2) != 0;
    if ((io_bf16 & 1) && cin == CIN_P && cout == COUT_P && (order == nullptr || presorted)) {
      const size_t lds_d = (size_t)2 * kWR * (CIN_P + COUT_P) * sizeof(unsigned short);
      static UdDeviceOnce dma_set;
      if (const unsigned long long dma_set_bit = (lds_d > 64 * 1024) ? dma_set.pending() : 0ull) {
        UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wgrad_bf16_dma<CIN_P, COUT_P, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
        UD_HIP_TRY(hipFuncSetAttribute((const void*)k_wgrad_bf16_dma<CIN_P, COUT_P, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_d));
        dma_set.mark(dma_set_bit);
      }
      const unsigned short* in16 = reinterpret_cast<const unsigned short*>(in);
      const unsigned short* g16 = reinterpret_cast<const unsigned short*>(gout);
      if (order)
        k_wgrad_bf16_dma<CIN_P, COUT_P, true><<<dim3(K, G), 256, lds_d, stream>>>(in16, nbr, K, g16, partial, Mout,
                                                                                   order, masks, ntiles, tpc);
      else
        k_wgrad_bf16_dma<CIN_P, COUT_P, false><<<dim3(K, G), 256, lds_d, stream>>>(in16, nbr, K, g16, partial, Mout,
                                                                                    nullptr, masks, ntiles, tpc);
      UD_LAUNCH_CHECK();
      k_wgrad_reduce<<<ud_div_up((long long)cout * K * cin, 256), 256, 0, stream>>>(
          partial, G, K, CIN_P, COUT_P, cin, cout, gW);
      UD_LAUNCH_CHECK();
      return UD_OK;
    }
  }
  if (io_bf16 & 2) return UD_ERR_UNSUPPORTED;     // the pre-sorted-rulebook mode exists for the DMA kernel only
  if (io_bf16 & 1)
    k_wgrad_bf16<CIN_P, COUT_P, true><<<dim3(K, G), 256, lds, stream>>>(in, cin, nbr, K, gout, cout, partial,
                                                                        Mout, order, masks, ntiles, tpc);
  else
    k_wgrad_bf16<CIN_P, COUT_P, false><<<dim3(K, G), 256, lds, stream>>>(in, cin, nbr, K, gout, cout, partial,
                                                                         Mout, order, masks, ntiles, tpc);
  UD_LAUNCH_CHECK();
  k_wgrad_reduce<<<ud_div_up((long long)cout * K * cin, 256), 256, 0, stream>>>(
      partial, G, K, CIN_P, COUT_P, cin, cout, gW);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
}  // namespace

extern "C" size_t ud_spconv_wgrad_bf16_workspace_bytes(int Mout, int K, int Cin, int Cout) {
  if (Mout < 0 || K <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const int ntiles = ud_div_up(Mout > 0 ? Mout : 1, kWR);
  const size_t G = (size_t)wgrad_bf16_max_chunks(K, pad16(Cin), pad16(Cout));
  return ud_align_up(G * K * pad16(Cin) * pad16(Cout) * sizeof(float)) +
         ud_align_up((size_t)ntiles * sizeof(unsigned));
}

// Per-64-row-tile activity masks (bit k: some row of the tile has a pair at offset k) of a rulebook in
// the given row order; depends on the rulebook only, so callers may compute it once and reuse it.
extern "C" int ud_spconv_tile_masks(const int32_t* nbr, int Mout, int K, const int32_t* row_order,
                                    unsigned* masks, ud_stream_t stream_) {
  if (Mout < 0 || K <= 0 || K > 32) return UD_ERR_INVALID_ARG;
  if (Mout == 0) return UD_OK;
  if (!nbr || !masks) return UD_ERR_INVALID_ARG;
  const int ntiles = ud_div_up(Mout, kWR);
  k_tile_masks<<<ud_div_up(ntiles, 4), 256, 0, (hipStream_t)stream_>>>(nbr, K, row_order, Mout, masks, ntiles);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// Mixed-precision weight gradient: bf16 operands (rounded when the row tiles are staged), fp32
// accumulation, ordered reduction.  row_order (optional) = the forward's mask-sorted row permutation.
extern "C" int ud_spconv_wgrad_bf16(const void* in_, const int32_t* nbr, const void* gout_, float* gW,
                                    int Mout, int K, int Cin, int Cout, int io_bf16,
                                    const int32_t* row_order, const unsigned* tile_masks,
                                    void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  const float* in = reinterpret_cast<const float*>(in_);
  const float* gout = reinterpret_cast<const float*>(gout_);
  if (Mout < 0 || K <= 0 || K > 32 || Cin <= 0 || Cout <= 0 || !gW) return UD_ERR_INVALID_ARG;
  if ((io_bf16 & 1) && ((Cin & 7) || (Cout & 7))) return UD_ERR_UNSUPPORTED;
  if ((io_bf16 & 2) && (!(io_bf16 & 1) || !row_order || !tile_masks)) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (Mout == 0) {
    UD_HIP_TRY(hipMemsetAsync(gW, 0, (size_t)Cout * K * Cin * sizeof(float), stream));
    return UD_OK;
  }
  if (!in || !nbr || !gout) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_spconv_wgrad_bf16_workspace_bytes(Mout, K, Cin, Cout))
    return UD_ERR_WORKSPACE;
  const int cp = pad16(Cin), np = pad16(Cout);
  float* partial = reinterpret_cast<float*>(workspace);
  unsigned* masks = reinterpret_cast<unsigned*>(
      reinterpret_cast<char*>(workspace) +
      ud_align_up((size_t)wgrad_bf16_max_chunks(K, cp, np) * K * cp * np * sizeof(float)));
  UdProfScope prof("spconv.k_wgrad", stream);
#define X(A, B) \
  if (cp == A && np == B) \
    return launch_wgrad_bf16<A, B>(in, Cin, nbr, K, gout, Cout, gW, Mout, row_order, partial, masks, tile_masks, io_bf16, stream);
  UD_CONV_CASES(X)
#undef X
  return UD_ERR_UNSUPPORTED;
}

// gW f32[Cout,K,Cin] (dense KRSC) = sum_o gout[o,:]^T (x) in[nbr[o][k], :]
extern "C" int ud_spconv_wgrad(const float* in, const int32_t* nbr, const float* gout, float* gW,
                               int Mout, int K, int Cin, int Cout, int algo, void* workspace,
                               size_t workspace_bytes, ud_stream_t stream_) {
  if (Mout < 0 || K <= 0 || Cin <= 0 || Cout <= 0 || !gW) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (Mout == 0) {
    UD_HIP_TRY(hipMemsetAsync(gW, 0, (size_t)Cout * K * Cin * sizeof(float), stream));
    return UD_OK;
  }
  if (!in || !nbr || !gout) return UD_ERR_INVALID_ARG;
  const int cp = pad16(Cin), np = pad16(Cout);
  UdProfScope prof("spconv.k_wgrad", stream);
  if (algo != 1) {   // weight gradients accumulate in fp32 MFMA for every MFMA algo (0, 2, 3)
    int rpc;
    const int G = wgrad_chunks(Mout, &rpc);
    if (!workspace || workspace_bytes < ud_spconv_wgrad_workspace_bytes(Mout, K, Cin, Cout))
      return UD_ERR_WORKSPACE;
#define X(A, B) \
  if (cp == A && np == B) \
    return launch_wgrad<A, B>(in, Cin, nbr, K, gout, Cout, gW, Mout, (float*)workspace, G, rpc, stream);
    UD_CONV_CASES(X)
#undef X
  }
  k_wgrad_generic<<<ud_div_up((long long)Cout * K * Cin, 256), 256, 0, stream>>>(
      in, Cin, nbr, K, gout, Cout, gW, Mout);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
