// LiDAR hard voxelization fused with MeanVFE for MI355X / gfx950.
//
// Replaces spconv.pytorch.utils.PointToVoxel.__call__ (third-party CUDA hash voxelizer; call
// site unidistill/data/det3d/preprocess/voxelization.py:31-38,54) and MeanVFE.forward
// (unidistill/layers/blocks_3d/det3d/vfe/mean_vfe.py:14-34).
//
// Canonical (CPU-deterministic) semantics, see oracle/ud_oracle.c:oracle_voxelize:
//   key = floor((p - range_min) / voxel_size) per axis, point kept iff 0 <= key < grid;
//   voxels are numbered in order of first appearance in the point list, per sample, and no new
//   voxel is created once a sample has max_voxels; a voxel keeps its first P points in input
//   order; num = min(#points, P); coords are emitted (b, z, y, x).
// The GPU gets the same result without any order-dependent race, in five launches + one memset:
//   k_insert   open-addressing hash over 64-bit entries (key << 32 | first point id): one word per
//              voxel, so a point touches ONE random cache line (a plain load settles points whose voxel
//              already holds a smaller id without any atomic; otherwise CAS on the empty slot / a 64-bit
//              atomicMin on the matching one)
//   k_first    per point ONE random read (its voxel's final entry) -> first id, kept for k_assign; the
//              "I am the first point" flags become a bitmap over the points by wave ballots (no atomics),
//              with in-block popcount prefixes; k_scan (one workgroup) scans the ~N/1024 block totals -> the
//              first-appearance rank of a voxel is a three-term lookup in cache-resident tables
//   k_assign   per point: rank -> output row; count (wave-aggregated integer atomic); the P smallest
//              point ids of a voxel are kept by an atomicMin insertion chain (slot j always ends up
//              holding the j-th smallest id, whatever the arrival order)
//   k_gather   a wave per 64 output rows, the lanes spread over the (point slot, feature) elements of one
//              row at a time: coalesced 4*P*F-byte row stores, num / count rows as 64-wide vectors, the
//              mean as an ordered sum over the slots (shuffles) -- voxels[M,P,F] is optional, so the fused
//              path writes M*(F+5) words instead of M*(P*F+4).
#include "ud_common.h"
#include "ud_prof.h"
#include <limits.h>
#include <algorithm>

namespace {

constexpr unsigned kEmpty = 0xFFFFFFFFu;
constexpr unsigned long long kEmpty64 = 0xFFFFFFFFFFFFFFFFull;

struct VoxParams {
  float lo[3];
  float vs[3];
  int grid[3];  // x, y, z
  int F;        // floats per point
  int P;        // max points per voxel
  int maxM;     // max voxels per sample
  int B;
  int N;        // points per sample (collate_fn pads clouds to a common length)
  unsigned tmask;
  int tshift;
};

__device__ __forceinline__ unsigned hash_slot(unsigned key, int shift) {
  return (key * 2654435761u) >> shift;
}

__global__ __launch_bounds__(256) void k_insert(const float* __restrict__ pts, VoxParams p,
                                                unsigned long long* __restrict__ table,
                                                int* __restrict__ pslot) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)p.B * p.N;
  if (gid >= total) return;
  const float* q = pts + gid * p.F;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float f = floorf(__fdiv_rn(__fsub_rn(q[a], p.lo[a]), p.vs[a]));
    ok = ok && (f >= 0.0f) && (f < (float)p.grid[a]);  // false for NaN
    c[a] = ok ? (int)f : 0;
  }
  int slot = -1;
  if (ok) {
    const int b = (int)(gid / p.N);
    const unsigned key =
        (unsigned)(((b * p.grid[2] + c[2]) * p.grid[1] + c[1])) * (unsigned)p.grid[0] + (unsigned)c[0];
    const unsigned long long mine = ((unsigned long long)key << 32) | (unsigned long long)(unsigned)gid;
    unsigned h = hash_slot(key, p.tshift);
    while (true) {
      // relaxed peek: an entry of this voxel that already holds a smaller id needs no atomic at all
      unsigned long long cur = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == kEmpty64) {
        cur = atomicCAS(&table[h], kEmpty64, mine);
        if (cur == kEmpty64) break;
      }
      if ((unsigned)(cur >> 32) == key) {
        if (cur > mine) atomicMin(&table[h], mine);
        break;
      }
      h = (h + 1) & p.tmask;
    }
    slot = (int)h;
  }
  pslot[gid] = slot;
}

// Per point: the final first id of its voxel (the one random read of the table after the inserts), kept
// for k_assign; "I am my voxel's first point" flags leave the wave as one ballot word, so the first-point
// bitmap costs no atomics.  A workgroup covers 1024 consecutive points = 16 bitmap words and stores their
// in-block exclusive popcount prefixes + the block total; k_scan (one workgroup) scans the block totals
// (a thousand values) and derives the per-sample voxel ranks / counts.  The first-appearance rank
// of any voxel is then  bprefix[fp >> 10] + wlocal[fp >> 6] + popc(bitmap[fp >> 6] below bit fp & 63).
constexpr int kFirstTile = 1024;

__device__ __forceinline__ int rank_of(unsigned fp, const unsigned long long* bitmap, const int* bprefix,
                                       const unsigned short* wlocal) {
  return bprefix[fp >> 10] + (int)wlocal[fp >> 6] +
         __popcll(bitmap[fp >> 6] & ((1ull << (fp & 63)) - 1ull));
}

__global__ __launch_bounds__(256) void k_first(const unsigned long long* __restrict__ table,
                                               const int* __restrict__ pslot, VoxParams p,
                                               unsigned* __restrict__ fpid,
                                               unsigned long long* __restrict__ bitmap,
                                               unsigned short* __restrict__ wlocal,
                                               int* __restrict__ part) {
  __shared__ int s_cnt[16];
  const long long total = (long long)p.B * p.N;
  const int lane = ud_lane(), wv = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * kFirstTile;
  // all four slot reads, then all four (random) table reads, are in flight together
  int sl[4];
  unsigned fp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long g = base + k * 256 + threadIdx.x;
    sl[k] = (g < total) ? pslot[g] : -1;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    fp[k] = (sl[k] >= 0) ? (unsigned)table[sl[k]] : kEmpty;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long g = base + k * 256 + threadIdx.x;
    if (g < total) fpid[g] = fp[k];
    const unsigned long long bits = __ballot(fp[k] == (unsigned)g && g < total);
    if (lane == 0) {
      const long long w = (base >> 6) + k * 4 + wv;
      if (w * 64 < total) bitmap[w] = bits;
      s_cnt[k * 4 + wv] = __popcll(bits);
    }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    int pre = 0;
    for (int i = 0; i < 16; ++i) pre += (i < (int)threadIdx.x) ? s_cnt[i] : 0;
    const long long w = (base >> 6) + threadIdx.x;
    if (w * 64 < total) wlocal[w] = (unsigned short)pre;
    if (threadIdx.x == 15) part[blockIdx.x] = pre + s_cnt[15];
  }
}

// One workgroup: exclusive scan of the ~N/1024 block totals, the voxel rank at every sample boundary, the
// per-sample voxel counts (capped at max_voxels) and their total.  A separate launch on purpose: folding it
// into k_first as a "last workgroup" tail needs a release fence per workgroup, and on this chip an
// agent-scope release writes the L2 back -- with the L2 full of the hash table's dirty lines that cost
// 110 us per call (measured), against ~3 us for this launch.
__global__ __launch_bounds__(256) void k_scan(const int* __restrict__ part, int nb, VoxParams p,
                                              const unsigned long long* __restrict__ bitmap,
                                              const unsigned short* __restrict__ wlocal,
                                              int* __restrict__ bprefix, int* __restrict__ samp_rank,
                                              int32_t* __restrict__ m_out) {
  __shared__ int s_w[4];
  const int lane = ud_lane(), wv = threadIdx.x >> 6;
  int run = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int i = b0 + threadIdx.x;
    const int v = (i < nb) ? part[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int a = __shfl_up(inc, o);
      if (lane >= o) inc += a;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = s_w[k];
      if (k < wv) pre += t;
      tot += t;
    }
    if (i < nb) bprefix[i] = run + pre + inc - v;
    run += tot;
    __syncthreads();
  }
  __syncthreads();
  // rank at every sample boundary, then the per-sample voxel counts (capped) and their total
  // (bprefix was written by this workgroup: read it back at agent scope, past the L1)
  if (threadIdx.x == 0) {
    int prev = 0, tot = 0;
    for (int b = 0; b <= p.B; ++b) {
      int r = run;
      if (b < p.B) {
        const unsigned g = (unsigned)((long long)b * p.N);
        r = __hip_atomic_load(&bprefix[g >> 10], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
            (int)wlocal[g >> 6] + __popcll(bitmap[g >> 6] & ((1ull << (g & 63)) - 1ull));
      }
      samp_rank[b] = r;
      if (b > 0) {
        const int m = min(r - prev, p.maxM);
        m_out[b - 1] = m;
        tot += m;
      }
      prev = r;
    }
    m_out[p.B] = tot;
  }
}

// Output row base of sample b = sum over earlier samples of min(#voxels, maxM).
__device__ __forceinline__ int sample_row_base(const int* __restrict__ samp_rank, int b, int maxM) {
  int base = 0;
  for (int i = 0; i < b; ++i) base += min(samp_rank[i + 1] - samp_rank[i], maxM);
  return base;
}

__global__ __launch_bounds__(256) void k_assign(const float* __restrict__ pts, VoxParams p,
                                                const unsigned* __restrict__ fpid,
                                                const unsigned long long* __restrict__ bitmap,
                                                const int* __restrict__ bprefix,
                                                const unsigned short* __restrict__ wlocal,
                                                const int* __restrict__ samp_rank,
                                                unsigned* __restrict__ top,
                                                unsigned* __restrict__ cnt,
                                                int32_t* __restrict__ coords) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)p.B * p.N;
  int row = -1;
  if (gid < total) {
    const unsigned first = fpid[gid];
    if (first != kEmpty) {
      const int b = (int)(gid / p.N);
      const int r = rank_of(first, bitmap, bprefix, wlocal) - samp_rank[b];
      if (r < p.maxM) {
        row = sample_row_base(samp_rank, b, p.maxM) + r;
        if (first == (unsigned)gid) {  // first point of the voxel writes its coordinates
          const float* q = pts + gid * p.F;
          int c[3];
#pragma unroll
          for (int a = 0; a < 3; ++a)
            c[a] = (int)floorf(__fdiv_rn(__fsub_rn(q[a], p.lo[a]), p.vs[a]));
          *reinterpret_cast<int4*>(coords + (size_t)row * 4) = make_int4(b, c[2], c[1], c[0]);
        }
      }
    }
  }
  // Point count per voxel: one integer atomic per run of equal rows inside the wave (zero-padded
  // clouds put thousands of consecutive points into one voxel).  cnt starts at 0xFFFFFFFF (the one
  // memset of the workspace), i.e. it holds count - 1.
  const int lane = ud_lane();
  const int prev = __shfl_up(row, 1);
  const bool start = (lane == 0) || (prev != row);
  const unsigned long long starts = __ballot(start);
  const unsigned long long upto = starts & ((2ull << lane) - 1ull);
  const int lead = 63 - __clzll(upto);
  const unsigned long long after = starts & ~((2ull << lead) - 1ull);
  const int end = after ? (__ffsll((long long)after) - 1) : 64;
  if (row >= 0 && lane == lead) atomicAdd(&cnt[row], (unsigned)(end - lead));
  // Keep the P smallest point ids: only the first P lanes of a run can be among them.
  if (row >= 0 && (lane - lead) < p.P) {
    unsigned* t = top + (size_t)row * p.P;
    unsigned carry = (unsigned)gid;
    if (__hip_atomic_load(&t[p.P - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > carry) {
      for (int j = 0; j < p.P; ++j) {
        const unsigned old = atomicMin(&t[j], carry);
        if (old == kEmpty) break;          // landed in a free slot
        carry = max(old, carry);           // the larger id moves on to the next slot
      }
    }
  }
}

// A wave owns 64 consecutive output rows = one contiguous block of 64*P*F output floats, and walks that
// block 64 elements at a time (element -> row r, point slot j, feature f): every store instruction is a
// full 256-byte piece and the iterations are independent, so many gathered point reads are in flight.
// The rows' kept point ids are staged once in LDS (coalesced).  mean[f] = (((v0 + v1) + v2) + ...) /
// max(n, 1) in slot order, like MeanVFE's sum over the slot axis.
constexpr int kGatherPMax = 16;  // ids staged in LDS up to this many slots per voxel

__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ pts, VoxParams p,
                                                const unsigned* __restrict__ top,
                                                const unsigned* __restrict__ cnt,
                                                const int32_t* __restrict__ m_out,
                                                float* __restrict__ voxels,
                                                int32_t* __restrict__ num,
                                                float* __restrict__ mean) {
  __shared__ unsigned s_id[4][64 * kGatherPMax];
  __shared__ int s_n[4][64];
  const int lane = ud_lane(), wv = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wv) * 64;
  const int M = m_out[p.B];
  if (row0 >= M) return;
  const int rows = min(64, M - row0);
  const int P = p.P, F = p.F, E = P * F;
  int n_l = 0;
  if (lane < rows) {
    n_l = min((int)(cnt[row0 + lane] + 1u), P);
    if (num) num[row0 + lane] = n_l;
  }
  s_n[wv][lane] = n_l;
  const bool staged = P <= kGatherPMax;
  const unsigned* trow = top + (size_t)row0 * P;
  if (staged)
    for (int i = lane; i < rows * P; i += 64) s_id[wv][i] = trow[i];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
  const float invF = 1.0f / (float)F, invE = 1.0f / (float)E;
  if (voxels) {
    float* vrow = voxels + (size_t)row0 * E;
    const int total = rows * E;
    for (int e = lane; e < total; e += 64) {
      int r = (int)(((float)e + 0.5f) * invE);
      const int rem = e - r * E;
      const int j = (int)(((float)rem + 0.5f) * invF);
      const int f = rem - j * F;
      float v = 0.0f;
      if (j < s_n[wv][r]) {
        const unsigned pid = staged ? s_id[wv][r * P + j] : trow[r * P + j];
        v = pts[(size_t)pid * F + f];
      }
      __builtin_nontemporal_store(v, &vrow[e]);
    }
  }
  if (mean) {
    float* mrow = mean + (size_t)row0 * F;
    const int total = rows * F;
    for (int e = lane; e < total; e += 64) {
      const int r = (int)(((float)e + 0.5f) * invF);
      const int f = e - r * F;
      const int n = s_n[wv][r];
      float acc = 0.0f;
      for (int j = 0; j < P; ++j) {
        float v = 0.0f;
        if (j < n) {
          const unsigned pid = staged ? s_id[wv][r * P + j] : trow[r * P + j];
          v = pts[(size_t)pid * F + f];
        }
        acc = __fadd_rn(acc, v);
      }
      mrow[e] = __fdiv_rn(acc, (float)max(n, 1));
    }
  }
}

struct VoxWs {
  unsigned long long* table;  // ---- 0xFF-initialised block: hash entries, top lists, counts - 1, ticket
  unsigned* top;
  unsigned* cnt;
  unsigned* ticket;
  size_t ff_bytes;
  unsigned long long* bitmap;  // ---- fully written by k_first (no initialisation)
  unsigned short* wlocal;
  int* part;
  int* bprefix;
  unsigned* fpid;
  int* pslot;
  int* samp_rank;
  size_t total_bytes;
  unsigned T;
  int tshift;
  int nwords;
  int ntile;
  int cap;
};

VoxWs carve(void* ws, int B, int N, int P, int maxM) {
  UdArena a(ws, (size_t)-1);
  VoxWs w;
  const size_t total = (size_t)B * N;
  unsigned T = 1024;
  int lg = 10;
  while ((size_t)T * 3 < total * 4) {  // load factor <= 0.75 even if every point opens a voxel
    T <<= 1;
    ++lg;
  }
  w.T = T;
  w.tshift = 32 - lg;
  w.cap = (int)((size_t)B * maxM < total ? (size_t)B * maxM : total);
  w.nwords = (int)((total + 63) / 64);
  w.ntile = (int)((total + kFirstTile - 1) / kFirstTile);
  w.table = a.take<unsigned long long>(T);
  w.top = a.take<unsigned>((size_t)w.cap * P);
  w.cnt = a.take<unsigned>(w.cap);
  w.ticket = a.take<unsigned>(4);
  w.ff_bytes = a.used;
  w.bitmap = a.take<unsigned long long>(w.nwords);
  w.wlocal = a.take<unsigned short>(w.nwords);
  w.part = a.take<int>(w.ntile);
  w.bprefix = a.take<int>(w.ntile);
  w.fpid = a.take<unsigned>(total);
  w.pslot = a.take<int>(total);
  w.samp_rank = a.take<int>(B + 1);
  w.total_bytes = a.used;
  return w;
}

bool vox_sizes_ok(int B, int N, int F, int P, int maxM, const int* grid) {
  if (B <= 0 || N <= 0 || F < 3 || P <= 0 || P > 64 || maxM <= 0) return false;
  if ((long long)B * N >= (1ll << 30)) return false;
  if (grid) {
    if (grid[0] <= 0 || grid[1] <= 0 || grid[2] <= 0) return false;
    const long long cells = (long long)B * grid[0] * grid[1] * grid[2];
    if (cells >= 0xFFFFFFFFll) return false;  // 32-bit linear key
  }
  return true;
}

}  // namespace

extern "C" size_t ud_voxelize_workspace_bytes(int B, int N, int P, int max_voxels) {
  if (!vox_sizes_ok(B, N, 3, P, max_voxels, nullptr)) return 0;
  return carve(nullptr, B, N, P, max_voxels).total_bytes;
}

extern "C" int ud_voxelize_capacity(int B, int N, int max_voxels) {
  if (B <= 0 || N <= 0 || max_voxels <= 0) return 0;
  const long long a = (long long)B * max_voxels, b = (long long)B * N;
  return (int)(a < b ? a : b);
}

extern "C" int ud_voxelize(const float* points, int B, int N, int F, const float* voxel_size,
                           const float* range, int P, int max_voxels, float* voxels,
                           int32_t* coords, int32_t* num_points, float* mean_feats,
                           int32_t* m_out, void* workspace, size_t workspace_bytes,
                           ud_stream_t stream_) {
  if (!points || !voxel_size || !range || !coords || !m_out) return UD_ERR_INVALID_ARG;
  VoxParams p;
  for (int a = 0; a < 3; ++a) {
    p.lo[a] = range[a];
    p.vs[a] = voxel_size[a];
    if (!(voxel_size[a] > 0.f)) return UD_ERR_INVALID_ARG;
    // grid = round((hi - lo) / voxel_size), as Voxelization.__init__ computes it (voxelization.py:40-43)
    p.grid[a] = (int)llround(((double)range[a + 3] - (double)range[a]) / (double)voxel_size[a]);
  }
  if (!vox_sizes_ok(B, N, F, P, max_voxels, p.grid)) return UD_ERR_INVALID_ARG;
  VoxWs w = carve(workspace, B, N, P, max_voxels);
  if (!workspace || workspace_bytes < w.total_bytes) return UD_ERR_WORKSPACE;
  p.F = F;
  p.P = P;
  p.maxM = max_voxels;
  p.B = B;
  p.N = N;
  p.tmask = w.T - 1;
  p.tshift = w.tshift;
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * N;
  UD_HIP_TRY(hipMemsetAsync(w.table, 0xFF, w.ff_bytes, stream));
  {
    UdProfScope prof("voxelize.k_insert", stream);
    k_insert<<<ud_div_up(total, 256), 256, 0, stream>>>(points, p, w.table, w.pslot);
    UD_LAUNCH_CHECK();
  }
  {
    UdProfScope prof("voxelize.k_first", stream);
    k_first<<<w.ntile, 256, 0, stream>>>(w.table, w.pslot, p, w.fpid, w.bitmap, w.wlocal, w.part);
    UD_LAUNCH_CHECK();
    k_scan<<<1, 256, 0, stream>>>(w.part, w.ntile, p, w.bitmap, w.wlocal, w.bprefix, w.samp_rank, m_out);
    UD_LAUNCH_CHECK();
  }
  {
    UdProfScope prof("voxelize.k_assign", stream);
    k_assign<<<ud_div_up(total, 256), 256, 0, stream>>>(points, p, w.fpid, w.bitmap, w.bprefix, w.wlocal,
                                                        w.samp_rank, w.top, w.cnt, coords);
    UD_LAUNCH_CHECK();
  }
  {
    UdProfScope prof("voxelize.k_gather", stream);
    k_gather<<<ud_div_up(w.cap, 256), 256, 0, stream>>>(points, p, w.top, w.cnt, m_out, voxels,
                                                        num_points, mean_feats);
    UD_LAUNCH_CHECK();
  }
  return UD_OK;
}
