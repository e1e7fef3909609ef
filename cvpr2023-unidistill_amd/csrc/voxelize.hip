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
// Two algorithms behind ud_voxelize, same bits:
//
// algo 0 (default) -- NO global atomics.  Measured on MI355X (tools/atomic_rate.hip): a device-scope atomic on a random
// word costs the same whatever its flavour (returning or not, 32 or 64 bit, 1 MB or 42 MB table): 24.7 G ops/s, i.e.
// 48 us per pass over 1.19 M points -- they execute at the memory side.  Plain random stores / loads run 3-5 x faster,
// LDS atomics are free in comparison.  So the points are PARTITIONED by a hash of their voxel key and every partition is
// voxelized inside one workgroup's LDS:
//   k_vp_partition  a workgroup per 2 048-point chunk: voxel key per point, bucket = hash(key) (all points of a voxel share
//                   a bucket), LDS histogram + ranks, the chunk's (key, point id) pairs leave grouped by bucket
//   k_vp_bucket     a workgroup per bucket, all in LDS with LDS atomics: gathers its ~2 k pairs from the chunks' segments,
//                   hashes the keys (slot = voxel), counting-sorts the point ids by slot, and a thread per voxel selects
//                   its P smallest ids in ascending order (= the kept points; the smallest = the voxel's first appearance);
//                   emits one record + id list per voxel and a byte flag at the first point id
//   k_vp_flags      flags -> bitmap over the points by wave ballots + in-block popcount prefixes; k_scan (one workgroup)
//                   scans the block totals: first-appearance rank = a three-term lookup, per-sample counts and caps
//   k_vp_rows       rank -> output row of every record (dropped beyond max_voxels); coords, count and id list in row order
//   k_gather        (shared with algo 1) a wave per 64 consecutive output rows: num and the point rows gathered through
//                   the id lists (voxels[M,P,F] and / or the MeanVFE mean), contiguous 256-byte stores
// One global atomic per BUCKET (record slots).  A bucket that receives more than 3 072 points (thousands of points in one
// voxel, e.g. long zero-padded tails) sets the overflow word m_out[B + 1]; the caller then repeats with algo 1.
//
// algo 1 -- open-addressing hash with device-scope atomics (round 1-2 design; any input), five launches + one memset:
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
#include <cstdlib>

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

// voxel key of the point at q (global point index gid); false: outside the grid / NaN
template <typename P>
__device__ __forceinline__ bool point_key_at(const float* __restrict__ q, const P& p, long long gid, unsigned& key) {
  int c[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float f = floorf(__fdiv_rn(__fsub_rn(q[a], p.lo[a]), p.vs[a]));
    ok = ok && (f >= 0.0f) && (f < (float)p.grid[a]);  // false for NaN
    c[a] = ok ? (int)f : 0;
  }
  const int b = (int)(gid / p.N);
  key = (unsigned)(((b * p.grid[2] + c[2]) * p.grid[1] + c[1])) * (unsigned)p.grid[0] + (unsigned)c[0];
  return ok;
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

template <bool CLEAN>
__device__ __forceinline__ void gather_rows(const float* __restrict__ pts, const VoxParams& p,
                                            unsigned* __restrict__ top, int TS,
                                            unsigned* __restrict__ cnt,
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
  const bool staged = TS <= kGatherPMax;
  const unsigned* trow = top + (size_t)row0 * TS;      // id lists, TS words per row (TS = P, or P rounded up to 4)
  if (CLEAN && !staged) __builtin_trap();              // (the cleaning variant is only launched with P <= kGatherPMax)
  if (staged)
    for (int i = lane; i < rows * TS; i += 64) s_id[wv][i] = trow[i];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
  const float invF = 1.0f / (float)F, invE = 1.0f / (float)E;
  if (voxels) {
    float* vrow = voxels + (size_t)row0 * E;       // 64 * E * 4 bytes per wave: 16-byte aligned
    const int total = rows * E;
    auto element = [&](int e) -> float {
      const int r = (int)(((float)e + 0.5f) * invE);
      const int rem = e - r * E;
      const int j = (int)(((float)rem + 0.5f) * invF);
      const int f = rem - j * F;
      if (j >= s_n[wv][r]) return 0.0f;
      const unsigned pid = staged ? s_id[wv][r * TS + j] : trow[r * TS + j];
      return pts[(size_t)pid * F + f];
    };
    // 16-byte stores: four gathered elements per lane and store instruction (a quarter of the 4-byte version's store
    // instructions; the walk over the 12.5 KB block stays contiguous)
    const int total4 = total >> 2;
    for (int q = lane; q < total4; q += 64) {
      ud_vf4 v4;
      v4.x = element(4 * q);
      v4.y = element(4 * q + 1);
      v4.z = element(4 * q + 2);
      v4.w = element(4 * q + 3);
      __builtin_nontemporal_store(v4, reinterpret_cast<ud_vf4*>(vrow) + q);
    }
    for (int e = 4 * total4 + lane; e < total; e += 64) __builtin_nontemporal_store(element(e), &vrow[e]);
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
          const unsigned pid = staged ? s_id[wv][r * TS + j] : trow[r * TS + j];
          v = pts[(size_t)pid * F + f];
        }
        acc = __fadd_rn(acc, v);
      }
      mrow[e] = __fdiv_rn(acc, (float)max(n, 1));
    }
  }
  if (CLEAN) {      // algo 2: leave the id lists / counts of the rows this wave consumed as the 0xFF state the next call starts from
    unsigned* wrow = top + (size_t)row0 * TS;
    for (int i = lane; i < rows * TS; i += 64) wrow[i] = kEmpty;
    if (lane < rows) cnt[row0 + lane] = kEmpty;
  }
}

__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ pts, VoxParams p,
                                                const unsigned* __restrict__ top, int TS,
                                                const unsigned* __restrict__ cnt,
                                                const int32_t* __restrict__ m_out,
                                                float* __restrict__ voxels,
                                                int32_t* __restrict__ num,
                                                float* __restrict__ mean) {
  gather_rows<false>(pts, p, const_cast<unsigned*>(top), TS, const_cast<unsigned*>(cnt), m_out, voxels, num, mean);
}


// ---- algo 2: the atomic hash in THREE launches for small clouds (<= 256 tiles of 1 024 points) ------------------------------
// At 30 k points every kernel of algo 1 runs 2-3 us and the op is its chain of launches (2 memsets + 5 kernels: 44 us).  Here:
//   k_insert2        = k_insert (+ the overflow word; checks the caller's claim that the workspace is in its clean state)
//   k_first_assign   = k_first + k_scan + k_assign in one launch: a tile publishes its first-point bitmap words, their in-tile
//                      prefixes and its count with device-scope stores (ordered by completion, not by a release fence: a fence
//                      would write the L2 back), then waits for the counts of the tiles BEFORE it -- a first point always lies
//                      in an earlier (or the same) tile, and workgroups are dispatched in index order, so the wait cannot
//                      deadlock -- scans them in LDS and assigns rows / counts / kept ids exactly like k_assign
//   k_gather_clean   = k_gather; every wave then restores the 0xFF state of the id lists / counts it consumed, the point tiles
//                      reset their hash slots and the header word is stamped: the next call needs no memset (the caller passes
//                      algo 3 = "the workspace still holds what my last algo-2/3 call with these sizes left"; k_insert2
//                      verifies the stamp and reports m_out[B + 1] = 2 otherwise -> repeat with algo 2).
constexpr unsigned kAggValid = 0x80000000u;

__global__ __launch_bounds__(256) void k_insert2(const float* __restrict__ pts, VoxParams p,
                                                 unsigned long long* __restrict__ table, int* __restrict__ pslot,
                                                 const unsigned* __restrict__ hdr, unsigned expect, int check,
                                                 int32_t* __restrict__ m_out) {
  const bool bad = check && __hip_atomic_load(&hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != expect;
  if (blockIdx.x == 0 && threadIdx.x == 0) m_out[p.B + 1] = bad ? 2 : 0;
  if (bad) return;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)p.B * p.N;
  if (gid >= total) return;
  unsigned key;
  const bool ok = point_key_at(pts + gid * p.F, p, gid, key);
  int slot = -1;
  if (ok) {
    const unsigned long long mine = ((unsigned long long)key << 32) | (unsigned long long)(unsigned)gid;
    unsigned h = hash_slot(key, p.tshift);
    while (true) {
      // small clouds: most points open their own voxel -- the CAS goes first (one round trip), no peek
      const unsigned long long cur = atomicCAS(&table[h], kEmpty64, mine);
      if (cur == kEmpty64) break;
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

__global__ __launch_bounds__(256) void k_first_assign(const float* __restrict__ pts, VoxParams p,
                                                      const unsigned long long* __restrict__ table,
                                                      const int* __restrict__ pslot,
                                                      unsigned long long* __restrict__ bitmap, unsigned* __restrict__ wl32,
                                                      unsigned* __restrict__ agg, unsigned* __restrict__ top,
                                                      unsigned* __restrict__ cnt, int32_t* __restrict__ coords,
                                                      int32_t* __restrict__ m_out, unsigned* __restrict__ hdr, int ntile) {
  __shared__ int s_cnt[16];
  __shared__ int s_E[258];          // s_E[i] = first points in tiles < i
  __shared__ int s_w[4];
  __shared__ int s_samp[66];        // first-appearance rank at the start of sample b (b <= B), as far as this tile can know it
  const long long total = (long long)p.B * p.N;
  const int lane = ud_lane(), wv = threadIdx.x >> 6, tid = threadIdx.x;
  const int t = blockIdx.x;
  const long long base = (long long)t * kFirstTile;
  if (t == 0 && tid == 0) __hip_atomic_store(&hdr[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // state: in use
  if (m_out[p.B + 1] != 0) return;                                     // k_insert2 refused the workspace
  int sl[4];
  unsigned fp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long g = base + k * 256 + tid;
    sl[k] = (g < total) ? pslot[g] : -1;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) fp[k] = (sl[k] >= 0) ? (unsigned)table[sl[k]] : kEmpty;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long g = base + k * 256 + tid;
    const unsigned long long bits = __ballot(fp[k] == (unsigned)g && g < total);
    if (lane == 0) {
      const long long w = (base >> 6) + k * 4 + wv;
      if (w * 64 < total) __hip_atomic_store(&bitmap[w], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_cnt[k * 4 + wv] = __popcll(bits);
    }
  }
  __syncthreads();
  int tot = 0;
  if (tid < 16) {
    int pre = 0;
    for (int i = 0; i < 16; ++i) pre += (i < tid) ? s_cnt[i] : 0;
    const long long w = (base >> 6) + tid;
    if (w * 64 < total) __hip_atomic_store(&wl32[w], (unsigned)pre, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 15) s_w[0] = pre + s_cnt[15];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this thread's device-scope stores have completed ...
  __syncthreads();                                       // ... everybody's have
  tot = s_w[0];
  if (tid == 0) __hip_atomic_store(&agg[t], (unsigned)tot | kAggValid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // counts of the tiles before this one
  int a = 0;
  if (tid < t) {
    unsigned v;
    do {
      v = __hip_atomic_load(&agg[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!(v & kAggValid)) __builtin_amdgcn_s_sleep(1);
    } while (!(v & kAggValid));
    a = (int)(v & ~kAggValid);
  } else if (tid == t) {
    a = tot;
  }
  __syncthreads();
  {
    int inc = a;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int pre = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) pre += (k < wv) ? s_w[k] : 0;
    s_E[tid + 1] = pre + inc;
    if (tid == 0) s_E[0] = 0;
  }
  __syncthreads();
  auto rank_at = [&](unsigned g) -> int {       // first points before point g (g in a tile <= t, or g == total in the last tile)
    if ((long long)g >= total) return s_E[ntile];
    const unsigned long long bw = __hip_atomic_load(&bitmap[g >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned wl = __hip_atomic_load(&wl32[g >> 6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s_E[g >> 10] + (int)wl + __popcll(bw & ((1ull << (g & 63)) - 1ull));
  };
  const long long tile_end = min(base + kFirstTile, total);
  if (tid <= p.B) {
    const long long g = (long long)tid * p.N;
    s_samp[tid] = (g < tile_end || (g == total && t == ntile - 1)) ? rank_at((unsigned)g) : 0x3fffffff;
  }
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const long long gid = base + k * 256 + tid;
    int row = -1;
    if (gid < total && fp[k] != kEmpty) {
      const unsigned first = fp[k];
      const int b = (int)(gid / p.N);
      const int r = rank_at(first) - s_samp[b];
      if (r < p.maxM) {
        int rb = 0;
        for (int i = 0; i < b; ++i) rb += min(s_samp[i + 1] - s_samp[i], p.maxM);
        row = rb + r;
        if (first == (unsigned)gid) {  // first point of the voxel writes its coordinates
          const float* q = pts + gid * p.F;
          int c[3];
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) c[ax] = (int)floorf(__fdiv_rn(__fsub_rn(q[ax], p.lo[ax]), p.vs[ax]));
          *reinterpret_cast<int4*>(coords + (size_t)row * 4) = make_int4(b, c[2], c[1], c[0]);
        }
      }
    }
    // counts and kept ids: as k_assign (one integer atomic per run of equal rows inside the wave; atomicMin insertion chain)
    const int prev = __shfl_up(row, 1);
    const bool start = (lane == 0) || (prev != row);
    const unsigned long long starts = __ballot(start);
    const unsigned long long upto = starts & ((2ull << lane) - 1ull);
    const int lead = 63 - __clzll(upto);
    const unsigned long long after = starts & ~((2ull << lead) - 1ull);
    const int end = after ? (__ffsll((long long)after) - 1) : 64;
    if (row >= 0 && lane == lead) atomicAdd(&cnt[row], (unsigned)(end - lead));
    if (row >= 0 && (lane - lead) < p.P) {
      unsigned* tt = top + (size_t)row * p.P;
      unsigned carry = (unsigned)gid;
      for (int j = 0; j < p.P; ++j) {          // (no peek at the last slot first: one round trip for a voxel's only point)
        const unsigned old = atomicMin(&tt[j], carry);
        if (old == kEmpty) break;
        carry = max(old, carry);
      }
    }
  }
  if (t == ntile - 1 && tid == 0) {      // per-sample voxel counts (capped) and their total
    int tot_m = 0;
    for (int b = 0; b < p.B; ++b) {
      const int m = min(s_samp[b + 1] - s_samp[b], p.maxM);
      m_out[b] = m;
      tot_m += m;
    }
    m_out[p.B] = tot_m;
  }
}

__global__ __launch_bounds__(256) void k_gather_clean(const float* __restrict__ pts, VoxParams p, unsigned* __restrict__ top,
                                                      unsigned* __restrict__ cnt, const int32_t* __restrict__ m_out,
                                                      float* __restrict__ voxels, int32_t* __restrict__ num,
                                                      float* __restrict__ mean, unsigned long long* __restrict__ table,
                                                      const int* __restrict__ pslot, unsigned* __restrict__ agg, int ntile,
                                                      unsigned* __restrict__ hdr, unsigned stamp) {
  if (m_out[p.B + 1] != 0) return;
  const long long total = (long long)p.B * p.N;
  for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
    const int sl = pslot[g];
    if (sl >= 0) table[sl] = kEmpty64;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < ntile; i += gridDim.x * 256) agg[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[0] = stamp;      // read by the NEXT call's k_insert2: a launch boundary away
  gather_rows<true>(pts, p, top, p.P, cnt, m_out, voxels, num, mean);
}

// ---- algo 0: hash partition + LDS sort (see the header) -------------------------------------------------------------
constexpr int kCH = 2048;        // points per partition chunk
constexpr int kCap = 3072;       // pairs per bucket that fit the LDS tables (3/4 of the hash slots)
constexpr int kMaxNB = 1024;     // buckets

__device__ __forceinline__ bool point_key(const float* __restrict__ q, const VoxParams& p, long long gid, unsigned& key) {
  int c[3];
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float f = floorf(__fdiv_rn(__fsub_rn(q[a], p.lo[a]), p.vs[a]));
    ok = ok && (f >= 0.0f) && (f < (float)p.grid[a]);  // false for NaN
    c[a] = ok ? (int)f : 0;
  }
  const int b = (int)(gid / p.N);
  key = (unsigned)(((b * p.grid[2] + c[2]) * p.grid[1] + c[1])) * (unsigned)p.grid[0] + (unsigned)c[0];
  return ok;
}

// tab[bucket][chunk] = (offset of the bucket's segment inside the chunk's pair list) << 16 | its length
__global__ __launch_bounds__(256) void k_vp_partition(const float* __restrict__ pts, VoxParams p, int NB, int lgNB,
                                                      int nchunks, unsigned long long* __restrict__ pairs,
                                                      unsigned* __restrict__ tab, unsigned char* __restrict__ flags) {
  __shared__ unsigned s_hist[kMaxNB];
  __shared__ unsigned long long s_pairs[kCH];
  __shared__ unsigned s_w[4];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long total = (long long)p.B * p.N, base = (long long)c * kCH;
  for (int i = tid; i < NB; i += 256) s_hist[i] = 0u;
  *reinterpret_cast<unsigned long long*>(flags + base + tid * 8) = 0ull;     // this chunk's first-point flags (kCH bytes)
  __syncthreads();
  unsigned key[kCH / 256], rk[kCH / 256];
  int bkt[kCH / 256];
#pragma unroll
  for (int i = 0; i < kCH / 256; ++i) {
    const long long gid = base + tid + 256 * i;
    bkt[i] = -1;
    if (gid < total && point_key(pts + gid * p.F, p, gid, key[i])) {
      bkt[i] = lgNB ? (int)((key[i] * 2654435761u) >> (32 - lgNB)) : 0;
      rk[i] = atomicAdd(&s_hist[bkt[i]], 1u);            // LDS atomic: rank inside (chunk, bucket); any order will do
    }
  }
  __syncthreads();
  // exclusive scan of the NB counters: thread t owns counters [t * per, t * per + per)
  const int per = NB > 256 ? NB / 256 : 1;
  unsigned v[kMaxNB / 256], sum = 0u;
#pragma unroll
  for (int k = 0; k < kMaxNB / 256; ++k) {
    const int idx = tid * per + k;
    v[k] = (k < per && idx < NB) ? s_hist[idx] : 0u;
    sum += v[k];
  }
  unsigned inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned a = __shfl_up(inc, o);
    if (lane >= o) inc += a;
  }
  if (lane == 63) s_w[wv] = inc;
  __syncthreads();
  unsigned off = inc - sum;
  for (int k = 0; k < wv; ++k) off += s_w[k];
  const unsigned nvalid = s_w[0] + s_w[1] + s_w[2] + s_w[3];
#pragma unroll
  for (int k = 0; k < kMaxNB / 256; ++k) {
    const int idx = tid * per + k;
    if (k < per && idx < NB) {
      s_hist[idx] = off;
      tab[(size_t)idx * nchunks + c] = (off << 16) | v[k];
      off += v[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kCH / 256; ++i)
    if (bkt[i] >= 0)
      s_pairs[s_hist[bkt[i]] + rk[i]] = ((unsigned long long)key[i] << 32) | (unsigned long long)(unsigned)(base + tid + 256 * i);
  __syncthreads();
  for (unsigned i = tid; i < nvalid; i += 256) pairs[base + i] = s_pairs[i];
}

struct VoxRec {
  unsigned key, first, count, pad;       // voxel key, first point id, #points
};

// A workgroup per bucket, everything in LDS with LDS atomics (which cost nothing next to the 24.7 G/s of a device-scope
// atomic): gather the bucket's pairs; open-addressing hash of the keys (CAS) + per-slot counts and ranks; exclusive scan
// of the counts; counting-sort the point ids into per-voxel segments; a thread per voxel then selects its P smallest ids in
// ascending order (segments average 2-3 ids) and writes the voxel's record, id list and first-point flag.
// (First version: bitonic sort of the (key, id) pairs -- 78 barrier-separated stages, 35 us per workgroup.)
constexpr int kLgTab = 12, kTab = 1 << kLgTab;      // hash slots: load factor <= 0.75 at the cap, ~0.25-0.5 typically
static_assert(kCap * 4 == kTab * 3, "kCap = 3/4 kTab");
constexpr int kScr = kCap + kCap / 2 + 2048;         // words: point ids, 16-bit slots, the gather's offsets; then the compacted slot list

__global__ __launch_bounds__(1024, 8) void k_vp_bucket(const unsigned long long* __restrict__ pairs,
                                                    const unsigned* __restrict__ tab, int nchunks, int P, int TS,
                                                    unsigned* __restrict__ vtop, VoxRec* __restrict__ rec,
                                                    unsigned* __restrict__ nrec, unsigned char* __restrict__ flags,
                                                    int32_t* __restrict__ ovf) {
  __shared__ unsigned s_key[kCap];        // keys of the gathered pairs; later the ids grouped by voxel
  // one block for the arrays that are dead after step 4 (point ids, slots, the gather's offsets): step 5 stages its output there
  __shared__ __attribute__((aligned(16))) unsigned s_scr[kScr];
  unsigned* const s_pid = s_scr;
  unsigned short* const s_slot = reinterpret_cast<unsigned short*>(s_scr + kCap);
  unsigned* const s_cpos = s_scr + kCap + kCap / 2;
  unsigned* const s_csrc = s_cpos + 1024;
  __shared__ unsigned s_tkey[kTab];
  __shared__ unsigned s_tcnt[kTab + 1];    // counts, then exclusive offsets
  __shared__ int s_ws[16], s_wo[16];
  __shared__ unsigned s_nv, s_base, s_nheavy;
  __shared__ unsigned s_heavy[kCap / 5 + 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < kTab; i += 1024) {
    s_tkey[i] = kEmpty;
    s_tcnt[i] = 0u;
  }
  if (tid == 0) s_nheavy = 0u;
  // 1. segment offsets (exclusive scan of the chunk counts of this bucket) + gather into LDS.  The copy is FLAT over the
  //    pairs (pair i -> its chunk by binary search in the offsets): a thread-per-chunk copy loop waits for its longest
  //    segment, one memory round trip per pair (12 of them for a Poisson(4) maximum: most of the first version's 19 us).
  int run = 0;
  for (int c0 = 0; c0 < nchunks; c0 += 1024) {
    const int c = c0 + tid;
    const unsigned meta = c < nchunks ? tab[(size_t)b * nchunks + c] : 0u;
    const int len = (int)(meta & 0xFFFFu), off = (int)(meta >> 16);
    int inc = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int a2 = __shfl_up(inc, o);
      if (lane >= o) inc += a2;
    }
    if (lane == 63) s_ws[wv] = inc;
    __syncthreads();
    int pos = inc - len, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int t = s_ws[k];
      if (k < wv) pos += t;
      tot += t;
    }
    s_cpos[tid] = (unsigned)pos;                                  // exclusive offset inside this pass
    s_csrc[tid] = (unsigned)((size_t)c * kCH + off);              // pair index of the segment (total < 2^30 points)
    __syncthreads();
    for (int i = tid; i < tot && run + i < kCap; i += 1024) {
      int lo = 0, hi = 1023;                                      // last chunk t with cpos[t] <= i
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (s_cpos[mid] <= (unsigned)i) lo = mid; else hi = mid - 1;
      }
      const unsigned long long e = pairs[(size_t)s_csrc[lo] + (i - (int)s_cpos[lo])];
      s_key[run + i] = (unsigned)(e >> 32);
      s_pid[run + i] = (unsigned)e;
    }
    run += tot;
    __syncthreads();
  }
  const int n = run;
  if (n > kCap) {                      // uniform: this bucket does not fit in LDS -> the caller falls back to algo 1
    if (tid == 0) *ovf = 1;
    return;
  }
  if (n == 0) return;
  // 2. hash the keys: slot per pair, rank of the pair inside its voxel (arrival order: any will do)
  unsigned rk[kCap / 1024];
#pragma unroll
  for (int u = 0; u < kCap / 1024; ++u) {
    const int i = tid + 1024 * u;
    if (i < n) {
      const unsigned key = s_key[i];
      // NOT the partition's multiplicative hash again: the keys of a bucket agree in exactly its top bits
      unsigned h = (key ^ (key >> 16)) * 0x7FEB352Du;
      h ^= h >> 15;
      h = (h * 0x846CA68Bu) >> (32 - kLgTab);
      while (true) {
        const unsigned old = atomicCAS(&s_tkey[h], kEmpty, key);
        if (old == kEmpty || old == key) break;
        h = (h + 1) & (kTab - 1);
      }
      s_slot[i] = (unsigned short)h;
      rk[u] = atomicAdd(&s_tcnt[h], 1u);
    }
  }
  __syncthreads();
  // 3. exclusive scan of the slot counts (thread t owns slots [4 t, 4 t + 4)); occupied slots = voxels: their exclusive count in
  //    slot order = the voxel's record index inside the bucket
  unsigned occ_off;
  {
    unsigned v[kTab / 1024], sum = 0u, occ = 0u;
#pragma unroll
    for (int k = 0; k < kTab / 1024; ++k) {
      v[k] = s_tcnt[tid * (kTab / 1024) + k];
      sum += v[k];
      occ += v[k] != 0u;
    }
    unsigned inc = sum, oinc = occ;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned a2 = __shfl_up(inc, o), o2 = __shfl_up(oinc, o);
      if (lane >= o) inc += a2, oinc += o2;
    }
    if (lane == 63) s_ws[wv] = (int)inc, s_wo[wv] = (int)oinc;
    __syncthreads();
    unsigned off = inc - sum;
    occ_off = oinc - occ;
    unsigned tot_occ = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < wv) off += (unsigned)s_ws[k], occ_off += (unsigned)s_wo[k];
      tot_occ += (unsigned)s_wo[k];
    }
#pragma unroll
    for (int k = 0; k < kTab / 1024; ++k) {
      s_tcnt[tid * (kTab / 1024) + k] = off;
      off += v[k];
    }
    if (tid == 1023) s_tcnt[kTab] = off;
    if (tid == 0) {
      s_nv = tot_occ;
      s_base = atomicAdd(nrec, tot_occ);      // the ONE global atomic of this workgroup
    }
  }
  __syncthreads();
  // 4. ids grouped by voxel (s_key is free: the keys live in the table now)
#pragma unroll
  for (int u = 0; u < kCap / 1024; ++u) {
    const int i = tid + 1024 * u;
    if (i < n) s_key[s_tcnt[s_slot[i]] + rk[u]] = s_pid[i];
  }
  __syncthreads();
  // 5. the occupied slots are COMPACTED (slot order = record order inside the bucket), then a thread per voxel: its P smallest ids
  //    in ascending order (ids are unique), the record and the first-point flag.  Rounds 2-5 walked the 4 096 slots four per
  //    thread with ~23 % of them occupied: every wave ran the selection body four times with a quarter of its lanes, and a
  //    voxel's record / id words left as single scattered stores in arrival order -- 2.2 M of them per 480 k voxels, 126 MB of
  //    HBM traffic for 31 MB of payload (profiles/traffic.json).  Now consecutive lanes hold consecutive records: the body runs
  //    once per 1 024 voxels with every lane busy, the record is one coalesced 16-byte store per lane and the sorted ids leave
  //    as 16-byte words of a contiguous row block.  Measured at 1.19 M points (round 6, early exits after each step, ~6 us of
  //    dispatch included): gather 13.4 us, + hash 17.2, + scan 19.8, + counting sort 20.1, + this step 36.1 -- the same 36 us as
  //    the slot walk (the small clouds gained 2 us): what this step costs is its 480 k first-point byte flags (random
  //    single-byte stores: a 64-byte sector read-modify-write each) and draining ~16 MB of records + id words, not instructions.
  const unsigned rbase = s_base, nv = s_nv;
  unsigned short* const s_list = reinterpret_cast<unsigned short*>(s_scr);      // [kTab]: slot of record lj (s_scr is free after step 4)
  {
    unsigned lj = occ_off;
#pragma unroll
    for (int k = 0; k < kTab / 1024; ++k) {
      const int h = tid * (kTab / 1024) + k;
      if (s_tkey[h] != kEmpty) s_list[lj++] = (unsigned short)h;
    }
  }
  __syncthreads();
  for (unsigned lj = tid; lj < nv; lj += 1024) {
    const unsigned h = s_list[lj];
    const unsigned o0 = s_tcnt[h], cnt = s_tcnt[h + 1] - o0;
    const unsigned j_rec = rbase + lj;
    if (cnt > 8u) {
      const unsigned e = atomicAdd(&s_nheavy, 1u);      // <= kCap / 9 = 341 such voxels in a bucket
      s_heavy[e] = h | (lj << 16);
      continue;
    }
    // the common case (2-3 ids per voxel): all reads independent; rank of an id = number of smaller ids; the sorted list is
    // assembled in registers (out[p] = the id of rank p) and stored 16 bytes at a time
    unsigned v[8], out[8];
#pragma unroll
    for (int a2 = 0; a2 < 8; ++a2) v[a2] = (unsigned)a2 < cnt ? s_key[o0 + a2] : kEmpty;
#pragma unroll
    for (int p2 = 0; p2 < 8; ++p2) out[p2] = kEmpty;
#pragma unroll
    for (int a2 = 0; a2 < 8; ++a2) {
      unsigned r2 = 0u;
#pragma unroll
      for (int c2 = 0; c2 < 8; ++c2) r2 += v[c2] < v[a2];
#pragma unroll
      for (int p2 = 0; p2 < 8; ++p2) out[p2] = (r2 == (unsigned)p2 && (unsigned)a2 < cnt) ? v[a2] : out[p2];
    }
    const unsigned keep = min(cnt, (unsigned)P);
    uint4* trow = reinterpret_cast<uint4*>(vtop + (size_t)j_rec * TS);       // TS % 4 == 0: 16-byte aligned rows
    trow[0] = make_uint4(out[0], out[1], out[2], out[3]);
    if (keep > 4u) trow[1] = make_uint4(out[4], out[5], out[6], out[7]);
    VoxRec r;
    r.key = s_tkey[h];
    r.first = out[0];
    r.count = cnt;
    r.pad = 0u;
    *reinterpret_cast<uint4*>(rec + j_rec) = *reinterpret_cast<const uint4*>(&r);
    flags[out[0]] = 1;
  }
  // voxels with more than 8 points (zero-padded tails, coarse grids): a WAVE per voxel -- every lane scans a strided part of
  // the segment, the wave takes the minimum; P rounds.  (Left to one thread, a 400-point voxel kept its whole workgroup --
  // and with it the kernel -- busy for 50 us.)
  __syncthreads();
  const unsigned nheavy = s_nheavy;
  for (unsigned e = wv; e < nheavy; e += 16) {
    const unsigned h = s_heavy[e] & 0xFFFFu, j_rec = rbase + (s_heavy[e] >> 16);
    const unsigned o0 = s_tcnt[h], cnt = s_tcnt[h + 1] - o0, keep = min(cnt, (unsigned)P);
    unsigned* top = vtop + (size_t)j_rec * TS;
    unsigned last = 0u, first = 0u;
    for (unsigned j = 0; j < keep; ++j) {
      unsigned m = kEmpty;
      for (unsigned t = lane; t < cnt; t += 64) {
        const unsigned v = s_key[o0 + t];
        if ((j == 0 || v > last) && v < m) m = v;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = min(m, (unsigned)__shfl_xor((int)m, o));
      if (lane == 0) top[j] = m;
      last = m;
      if (j == 0) first = m;
    }
    if (lane == 0) {
      VoxRec r;
      r.key = s_tkey[h];
      r.first = first;
      r.count = cnt;
      r.pad = 0u;
      *reinterpret_cast<uint4*>(rec + j_rec) = *reinterpret_cast<const uint4*>(&r);
      flags[first] = 1;
    }
  }
}

// first-point byte flags -> bitmap + in-block popcount prefixes (the layout k_scan / rank_of read)
__global__ __launch_bounds__(256) void k_vp_flags(const unsigned char* __restrict__ flags, VoxParams p,
                                                  unsigned long long* __restrict__ bitmap,
                                                  unsigned short* __restrict__ wlocal, int* __restrict__ part) {
  __shared__ int s_cnt[16];
  const long long total = (long long)p.B * p.N;
  const int lane = ud_lane(), wv = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * kFirstTile;
  unsigned char f[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long g = base + k * 256 + threadIdx.x;
    f[k] = (g < total) ? flags[g] : (unsigned char)0;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long bits = __ballot(f[k] != 0);
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

// Records -> row order: rank -> output row of every voxel record (dropped beyond max_voxels); the row's coordinates, its
// point count (as count - 1, k_gather's convention) and its kept ids land in row-indexed arrays, so that the final pass is
// the same k_gather as algo 1's (a wave per 64 consecutive rows, contiguous stores).
__global__ __launch_bounds__(256) void k_vp_rows(VoxParams p, const VoxRec* __restrict__ rec,
                                                 const unsigned* __restrict__ nrec, const unsigned* __restrict__ vtop,
                                                 const unsigned long long* __restrict__ bitmap,
                                                 const int* __restrict__ bprefix, const unsigned short* __restrict__ wlocal,
                                                 const int* __restrict__ samp_rank, int TS, unsigned* __restrict__ top,
                                                 unsigned* __restrict__ cnt, int32_t* __restrict__ coords) {
  const unsigned j = blockIdx.x * 256u + threadIdx.x;
  if (j >= *nrec) return;
  const uint4 q = *reinterpret_cast<const uint4*>(rec + j);
  unsigned t = q.x;
  const int x = (int)(t % (unsigned)p.grid[0]);
  t /= (unsigned)p.grid[0];
  const int y = (int)(t % (unsigned)p.grid[1]);
  t /= (unsigned)p.grid[1];
  const int z = (int)(t % (unsigned)p.grid[2]), b = (int)(t / (unsigned)p.grid[2]);
  const int r = rank_of(q.y, bitmap, bprefix, wlocal) - samp_rank[b];
  if (r >= p.maxM) return;
  const size_t row = (size_t)sample_row_base(samp_rank, b, p.maxM) + r;
  *reinterpret_cast<int4*>(coords + row * 4) = make_int4(b, z, y, x);
  cnt[row] = q.z - 1u;
  const int keep = min((int)q.z, p.P);
  if ((TS & 3) == 0) {
    const uint4* src = reinterpret_cast<const uint4*>(vtop + (size_t)j * TS);
    uint4* dst = reinterpret_cast<uint4*>(top + row * TS);
    for (int k = 0; 4 * k < keep; ++k) dst[k] = src[k];
  } else {
    const unsigned* src = vtop + (size_t)j * TS;
    unsigned* dst = top + row * TS;
    for (int k = 0; k < keep; ++k) dst[k] = src[k];
  }
}

struct VoxWs {
  unsigned long long* table;  // ---- 0xFF-initialised block: hash entries, top lists, counts - 1, ticket
  unsigned* top;
  unsigned* cnt;
  unsigned* ticket;
  unsigned* wl32;              // algo 2: in-tile prefixes of the bitmap words, 32-bit (device-scope stores)
  unsigned* hdr;               // algo 2: state stamp of the workspace
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
  w.wl32 = a.take<unsigned>(w.nwords);
  w.hdr = a.take<unsigned>(4);
  w.total_bytes = a.used;
  return w;
}

struct VpWs {
  unsigned long long* pairs;   // [nchunks * kCH]
  unsigned* tab;               // [NB][nchunks]
  unsigned char* flags;        // [nchunks * kCH]
  unsigned* vtop;              // [B * N][P] kept point ids per voxel record (ascending)
  unsigned* top;               // [cap][P] the same lists in output-row order
  unsigned* cnt;               // [cap]    points per row - 1
  VoxRec* rec;                 // [B * N]
  unsigned* nrec;
  unsigned long long* bitmap;
  unsigned short* wlocal;
  int* part;
  int* bprefix;
  int* samp_rank;
  size_t total_bytes;
  int nchunks, NB, lgNB, nwords, ntile, TS;
};

VpWs carve_vp(void* ws, int B, int N, int P, int maxM) {
  UdArena a(ws, (size_t)-1);
  VpWs w;
  const size_t total = (size_t)B * N;
  w.nchunks = (int)((total + kCH - 1) / kCH);
  // buckets: a power of two with <= ~2.4 k points each on average (capacity 4 096), at most kMaxNB
  int nb = 1, lg = 0;
  while (nb < kMaxNB && (size_t)nb * 2400 < total) {
    nb <<= 1;
    ++lg;
  }
  w.NB = nb;
  w.lgNB = lg;
  w.nwords = (int)((total + 63) / 64);
  w.ntile = (int)((total + kFirstTile - 1) / kFirstTile);
  w.pairs = a.take<unsigned long long>((size_t)w.nchunks * kCH);
  w.tab = a.take<unsigned>((size_t)w.NB * w.nchunks);
  w.flags = a.take<unsigned char>((size_t)w.nchunks * kCH);
  w.TS = (P + 3) / 4 * 4;
  w.vtop = a.take<unsigned>(total * w.TS);
  const size_t cap = (size_t)B * maxM < total ? (size_t)B * maxM : total;
  w.top = a.take<unsigned>(cap * w.TS);
  w.cnt = a.take<unsigned>(cap);
  w.rec = a.take<VoxRec>(total);
  w.nrec = a.take<unsigned>(4);
  w.bitmap = a.take<unsigned long long>(w.nwords);
  w.wlocal = a.take<unsigned short>(w.nwords);
  w.part = a.take<int>(w.ntile);
  w.bprefix = a.take<int>(w.ntile);
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
  return std::max(carve(nullptr, B, N, P, max_voxels).total_bytes, carve_vp(nullptr, B, N, P, max_voxels).total_bytes);
}

extern "C" int ud_voxelize_capacity(int B, int N, int max_voxels) {
  if (B <= 0 || N <= 0 || max_voxels <= 0) return 0;
  const long long a = (long long)B * max_voxels, b = (long long)B * N;
  return (int)(a < b ? a : b);
}

extern "C" int ud_voxelize(const float* points, int B, int N, int F, const float* voxel_size,
                           const float* range, int P, int max_voxels, float* voxels,
                           int32_t* coords, int32_t* num_points, float* mean_feats,
                           int32_t* m_out, void* workspace, size_t workspace_bytes, int algo,
                           ud_stream_t stream_) {
  if (!points || !voxel_size || !range || !coords || !m_out || algo < 0 || algo > 3) return UD_ERR_INVALID_ARG;
  VoxParams p;
  for (int a = 0; a < 3; ++a) {
    p.lo[a] = range[a];
    p.vs[a] = voxel_size[a];
    if (!(voxel_size[a] > 0.f)) return UD_ERR_INVALID_ARG;
    // grid = round((hi - lo) / voxel_size), as Voxelization.__init__ computes it (voxelization.py:40-43)
    p.grid[a] = (int)llround(((double)range[a + 3] - (double)range[a]) / (double)voxel_size[a]);
  }
  if (!vox_sizes_ok(B, N, F, P, max_voxels, p.grid)) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_voxelize_workspace_bytes(B, N, P, max_voxels)) return UD_ERR_WORKSPACE;
  p.F = F;
  p.P = P;
  p.maxM = max_voxels;
  p.B = B;
  p.N = N;
  p.tmask = 0;
  p.tshift = 0;
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * N;
  if (algo == 0) {
    VpWs v = carve_vp(workspace, B, N, P, max_voxels);
    const int cap = ud_voxelize_capacity(B, N, max_voxels);
    // record counter + the overflow word (m_out[B + 1]) start at zero
    UD_HIP_TRY(hipMemsetAsync(v.nrec, 0, 16, stream));
    UD_HIP_TRY(hipMemsetAsync(m_out + B + 1, 0, sizeof(int32_t), stream));
    {
      UdProfScope prof("voxelize.k_partition", stream);
      k_vp_partition<<<v.nchunks, 256, 0, stream>>>(points, p, v.NB, v.lgNB, v.nchunks, v.pairs, v.tab, v.flags);
      UD_LAUNCH_CHECK();
    }
    {
      UdProfScope prof("voxelize.k_bucket", stream);
      k_vp_bucket<<<v.NB, 1024, 0, stream>>>(v.pairs, v.tab, v.nchunks, P, v.TS, v.vtop, v.rec, v.nrec, v.flags, m_out + B + 1);
      UD_LAUNCH_CHECK();
    }
    {
      UdProfScope prof("voxelize.k_flags", stream);
      k_vp_flags<<<v.ntile, 256, 0, stream>>>(v.flags, p, v.bitmap, v.wlocal, v.part);
      UD_LAUNCH_CHECK();
      k_scan<<<1, 256, 0, stream>>>(v.part, v.ntile, p, v.bitmap, v.wlocal, v.bprefix, v.samp_rank, m_out);
      UD_LAUNCH_CHECK();
    }
    {
      UdProfScope prof("voxelize.k_rows", stream);
      k_vp_rows<<<ud_div_up(total, 256), 256, 0, stream>>>(p, v.rec, v.nrec, v.vtop, v.bitmap, v.bprefix, v.wlocal,
                                                           v.samp_rank, v.TS, v.top, v.cnt, coords);
      UD_LAUNCH_CHECK();
    }
    {
      UdProfScope prof("voxelize.k_gather", stream);
      k_gather<<<ud_div_up(cap, 256), 256, 0, stream>>>(points, p, v.top, v.TS, v.cnt, m_out, voxels, num_points, mean_feats);
      UD_LAUNCH_CHECK();
    }
    return UD_OK;
  }
  VoxWs w = carve(workspace, B, N, P, max_voxels);
  p.tmask = w.T - 1;
  p.tshift = w.tshift;
  if (algo >= 2) {
    if (w.ntile > 256 || P > kGatherPMax || B > 64) return UD_ERR_UNSUPPORTED;
    const unsigned stamp = 0xC1EA0000u ^ (unsigned)(B * 0x9E3779B1u) ^ (unsigned)(N * 0x85EBCA77u) ^ (unsigned)(P * 0xC2B2AE3Du) ^
                           (unsigned)max_voxels;
    if (algo == 2) {      // unknown workspace contents: the one memset (hash entries, id lists, counts: 0xFF; tile counts: 0)
      UD_HIP_TRY(hipMemsetAsync(w.table, 0xFF, w.ff_bytes, stream));
      UD_HIP_TRY(hipMemsetAsync(w.part, 0, (size_t)w.ntile * sizeof(int), stream));
    }
    {
      UdProfScope prof("voxelize.k_insert2", stream);
      k_insert2<<<ud_div_up(total, 256), 256, 0, stream>>>(points, p, w.table, w.pslot, w.hdr, stamp, algo == 3, m_out);
      UD_LAUNCH_CHECK();
    }
    {
      UdProfScope prof("voxelize.k_first_assign", stream);
      k_first_assign<<<w.ntile, 256, 0, stream>>>(points, p, w.table, w.pslot, w.bitmap, w.wl32, (unsigned*)w.part, w.top, w.cnt,
                                                  coords, m_out, w.hdr, w.ntile);
      UD_LAUNCH_CHECK();
    }
    {
      UdProfScope prof("voxelize.k_gather", stream);
      const int blocks = std::max(ud_div_up(w.cap, 256), std::min(w.ntile, 64));
      k_gather_clean<<<blocks, 256, 0, stream>>>(points, p, w.top, w.cnt, m_out, voxels, num_points, mean_feats, w.table,
                                                 w.pslot, (unsigned*)w.part, w.ntile, w.hdr, stamp);
      UD_LAUNCH_CHECK();
    }
    return UD_OK;
  }
  UD_HIP_TRY(hipMemsetAsync(m_out + B + 1, 0, sizeof(int32_t), stream));
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
    k_gather<<<ud_div_up(w.cap, 256), 256, 0, stream>>>(points, p, w.top, P, w.cnt, m_out, voxels,
                                                        num_points, mean_feats);
    UD_LAUNCH_CHECK();
  }
  return UD_OK;
}
