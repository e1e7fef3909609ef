// BEV pool (LSS splat) for MI355X / gfx950.
//
// Replaces voxel_pooling_ext.voxel_pooling_forward_wrapper
// (reference: unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:48-59) and the python
// backward gather (lss_fpn.py:64-79).
//
// Design (not the reference CUDA op's per-(point,channel) atomicAdd):
//   k_bin    one thread per frustum point: bounds test, write pos_memo; the points of a cell are counted with one
//            integer atomic per (workgroup, distinct cell), which gives every point an arbitrary in-cell rank, and the
//            point id goes STRAIGHT into slot `rank` of the cell's fixed 256-slot row of the id table (a light cell's
//            list is the first 256 bytes of its row: no scan of the counts, no second scatter pass -- rounds 1-3 ran
//            k_cell_offsets + k_fill here).  Ranks >= 256 (the frustum puts ~200 points into its densest cells) append
//            (cell, id) to an overflow list; the workgroup whose reservation crosses 64 appends the cell to the heavy list.
//   k_pool   light role: one WAVE per cell (<= 64 points): in-register rank sort of the cell's
//            point ids, then each lane owns 4 channels (16 B) of the 1 KiB feature row and adds
//            the rows in ascending point order -> coalesced 1 KiB reads, one 1 KiB write, no fp
//            atomics, bit-reproducible and bit-identical to a sequential CPU loop.
//            heavy role (first kHeavyBlocks workgroups of the same launch, so the long cells
//            start first and overlap the light ones): the cell's row + its entries of the overflow list, LDS bitonic
//            sort of the ids, 4 waves sum contiguous chunks, partials are combined in wave order.
// HBM traffic = every kept feature row once + the output once; that is the algorithmic minimum.
#include "ud_common.h"
#include "ud_prof.h"
#include "lss_geom.h"
#include <limits.h>

namespace {

constexpr int kLightMax = 64;      // cells with more points go to k_heavy
constexpr int kRow = 256;          // slots of a cell's row in the id table (1 KiB); ranks beyond go to the overflow list
constexpr int kHeavySortMax = 4096;  // LDS sort capacity of the heavy role; above: index-range scan
constexpr int kHeavyBlocks = 128;    // persistent heavy-role workgroups at the front of k_pool
constexpr int kLightBlocks = 2048;   // persistent light-role workgroups (256 CUs x 8)

// ----------------------------------------------------------------------------------------
// Where a point's bin comes from: the geom tensor of the reference boundary, or the frustum geometry itself (fused lift-splat of
// the training step: the [B,N,3] bins are never written).
struct BinsLoad {
  const int32_t* __restrict__ geom;
  __device__ __forceinline__ void get(long long gid, int& x, int& y, int& z) const {
    x = geom[gid * 3 + 0];
    y = geom[gid * 3 + 1];
    z = geom[gid * 3 + 2];
  }
};
struct BinsFrustum {
  UdFrustum f;
  __device__ __forceinline__ void get(long long gid, int& x, int& y, int& z) const {
    float q[4];
    ud_frustum_point(f, gid, q, &x, &y, &z);
  }
};

template <class Bins>
__global__ __launch_bounds__(256) void k_bin(Bins bins,
                                             int32_t* __restrict__ pos, int* __restrict__ count,
                                             int* __restrict__ list, int* __restrict__ cellid,
                                             int* __restrict__ heavy_list, int* __restrict__ heavy_cnt,
                                             int2* __restrict__ ovf, int* __restrict__ ovf_cnt,
                                             long long total, int N, int nx, int ny, int nz) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  int cell = -1;
  if (gid < total) {
    const int b = (int)(gid / N);
    int x, y, z;
    bins.get(gid, x, y, z);
    const bool kept = (x >= 0) & (x < nx) & (y >= 0) & (y < ny) & (z >= 0) & (z < nz);
    if (kept) cell = (b * ny + y) * nx + x;
    pos[gid * 3 + 0] = kept ? b : -1;
    pos[gid * 3 + 1] = kept ? y : -1;
    pos[gid * 3 + 2] = kept ? x : -1;
  }
  // One global atomic per DISTINCT cell of the workgroup's 256 points: frustum points are ordered (camera, depth, image row, image
  // column), so a workgroup holds ~6 image rows of one depth plane -- the rows of a column fall into the same BEV cell (only z
  // differs).  Device-scope atomics run at 24.7 G/s on this chip whatever their flavour (tools/atomic_rate.hip); one per run of equal
  // consecutive cells (~280 k at 473 k points) was 11 of this kernel's 17 us.  The cells are counted in an LDS hash (LDS atomics), a
  // thread per occupied slot reserves the cell's range, and every point takes base + its LDS ticket.
  constexpr int kSlots = 512;
  __shared__ int s_key[kSlots], s_cnt[kSlots], s_base[kSlots];
  __shared__ int s_ovf, s_ovf_base;
  for (int i = threadIdx.x; i < kSlots; i += 256) s_key[i] = -1, s_cnt[i] = 0;
  if (threadIdx.x == 0) s_ovf = 0;
  __syncthreads();
  int slot = -1, ticket = 0;
  if (cell >= 0) {
    unsigned h = ((unsigned)cell * 2654435761u) >> 23;            // 9 bits
    for (;;) {
      const int prev = atomicCAS(&s_key[h], -1, cell);
      if (prev == -1 || prev == cell) break;
      h = (h + 1) & (kSlots - 1);
    }
    slot = (int)h;
    ticket = atomicAdd(&s_cnt[slot], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSlots; i += 256)
    if (s_cnt[i] > 0) {
      const int key = s_key[i], c = s_cnt[i];
      const int b = atomicAdd(&count[key], c);
      s_base[i] = b;
      // exactly one reservation per cell crosses the light limit: that workgroup lists the cell as heavy
      if (b <= kLightMax && b + c > kLightMax) heavy_list[atomicAdd(heavy_cnt, 1)] = key;
    }
  __syncthreads();
  int r = -1;
  if (gid < total) {
    cellid[gid] = cell;
    if (cell >= 0) r = s_base[slot] + ticket;
  }
  const bool over = r >= kRow;
  if (r >= 0 && !over) list[(size_t)cell * kRow + r] = (int)gid;
  // ranks past the row (a cell with hundreds of points): one overflow-list reservation per workgroup
  int t = 0;
  if (over) t = atomicAdd(&s_ovf, 1);
  if (__syncthreads_or(over)) {
    if (threadIdx.x == 0) s_ovf_base = atomicAdd(ovf_cnt, s_ovf);
    __syncthreads();
    if (over) ovf[s_ovf_base + t] = make_int2(cell, (int)gid);
  }
}

// ----------------------------------------------------------------------------------------
// Row add helpers: a lane owns VEC consecutive channels.
template <int VEC>
struct RowVec;
template <>
struct RowVec<4> {
  using T = float4;
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ void add(T& a, const T& b) {
    a.x = __fadd_rn(a.x, b.x);
    a.y = __fadd_rn(a.y, b.y);
    a.z = __fadd_rn(a.z, b.z);
    a.w = __fadd_rn(a.w, b.w);
  }
};
template <>
struct RowVec<1> {
  using T = float;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ void add(T& a, const T& b) { a = __fadd_rn(a, b); }
};

// ---- row sources ---------------------------------------------------------------------------
// A source turns a point id into the feature row that is pooled.  `prep` runs once per id with
// one id per lane (vector code); `load` runs with a wave-uniform lane index j and returns the
// slice of that point's row owned by the calling lane.
template <int VEC>
struct SrcFeat {  // materialised feat[B*N, C]  (reference boundary, lss_fpn.py:48-59)
  using V = typename RowVec<VEC>::T;
  const float* __restrict__ feat;
  int C;
  struct Meta {
    int id;
  };
  __device__ __forceinline__ Meta prep(int id) const { return Meta{id}; }
  __device__ __forceinline__ V load(const Meta& m, int j, int ch) const {
    const int p = __builtin_amdgcn_readlane(m.id, j);
    // every feature row is read exactly once: stream it past the caches (nontemporal)
    if constexpr (VEC == 4) {
      typedef float vf4 __attribute__((ext_vector_type(4)));
      const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(feat + (size_t)p * C + ch));
      return make_float4(t.x, t.y, t.z, t.w);
    } else {
      return __builtin_nontemporal_load(feat + (size_t)p * C + ch);
    }
  }
};

template <int VEC>
struct SrcLift {  // fused lift: row = depth_prob[point] * context[pixel(point), :]  (lss_fpn.py:289-292)
  using V = typename RowVec<VEC>::T;
  const float* __restrict__ prob;  // [B*ncam, D, fH*fW]  == point order
  const float* __restrict__ ctx;   // [B*ncam, fH*fW, C]  pixel-major
  int C;
  int DHW;  // D*fH*fW  points per camera
  int HW;   // fH*fW
  struct Meta {
    int pix;
    float p;
  };
  __device__ __forceinline__ Meta prep(int id) const {
    Meta m;
    if (id == INT_MAX) id = 0;
    const int cam = id / DHW;  // global camera index b*ncam + cam
    m.pix = cam * HW + (id - cam * DHW) % HW;
    m.p = prob[id];
    return m;
  }
  __device__ __forceinline__ V load(const Meta& m, int j, int ch) const {
    const int px = __builtin_amdgcn_readlane(m.pix, j);
    const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m.p), j));
    V v = *reinterpret_cast<const V*>(ctx + (size_t)px * C + ch);
    if constexpr (VEC == 4) {
      v.x = __fmul_rn(v.x, w);
      v.y = __fmul_rn(v.y, w);
      v.z = __fmul_rn(v.z, w);
      v.w = __fmul_rn(v.w, w);
    } else {
      v = __fmul_rn(v, w);
    }
    return v;
  }
};

// Add rows j = 0..k-1 (k <= 64; one prepared id per lane, ascending) into acc, in that order.
template <int VEC, class Src>
__device__ __forceinline__ void add_rows_wave(const Src& src, const typename Src::Meta& m, int k,
                                              int ch, typename RowVec<VEC>::T& acc) {
  using V = typename RowVec<VEC>::T;
  int j = 0;
  for (; j + 8 <= k; j += 8) {
    V r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = src.load(m, j + u, ch);
#pragma unroll
    for (int u = 0; u < 8; ++u) RowVec<VEC>::add(acc, r[u]);
  }
  for (; j + 2 <= k; j += 2) {
    const V r0 = src.load(m, j, ch);
    const V r1 = src.load(m, j + 1, ch);
    RowVec<VEC>::add(acc, r0);
    RowVec<VEC>::add(acc, r1);
  }
  if (j < k) {
    const V r0 = src.load(m, j, ch);
    RowVec<VEC>::add(acc, r0);
  }
}

// Heavy role: one 256-thread workgroup per cell with > kLightMax points.
template <int VEC, class Src>
__device__ __forceinline__ void pool_heavy_cell(const Src& src, float* __restrict__ out,
                                                const int* __restrict__ count,
                                                const int* __restrict__ list,
                                                const int2* __restrict__ ovf, int novf,
                                                const int* __restrict__ cellid, int cell, int N,
                                                int nynx, int C, unsigned flags, int* s_ids,
                                                float (*s_part)[64 * VEC], int* s_n) {
  using V = typename RowVec<VEC>::T;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int k = count[cell];
  float* orow = out + (size_t)cell * C;
  const bool sortable = (k <= kHeavySortMax);
  if (sortable) {
    int npow = 1;
    while (npow < k) npow <<= 1;
    // ranks 0..kRow-1 sit in the cell's row of the id table, the others among the overflow entries (any order: sorted next)
    if (tid < min(k, kRow)) s_ids[tid] = list[(size_t)cell * kRow + tid];
    if (k > kRow) {
      if (tid == 0) *s_n = kRow;
      __syncthreads();
      for (int i0 = 0; i0 < novf; i0 += 4 * 256) {
        int2 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * 256 + tid;
          e[u] = i < novf ? ovf[i] : make_int2(-1, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e[u].x == cell) s_ids[atomicAdd(s_n, 1)] = e[u].y;
      }
    }
    for (int i = k + tid; i < npow; i += 256) s_ids[i] = INT_MAX;
    __syncthreads();
    for (int len = 2; len <= npow; len <<= 1) {
      for (int st = len >> 1; st > 0; st >>= 1) {
        for (int i = tid; i < (npow >> 1); i += 256) {
          const int lo = ((i & ~(st - 1)) << 1) | (i & (st - 1));
          const int hi2 = lo | st;
          const bool up = ((lo & len) == 0);
          const int a = s_ids[lo], b2 = s_ids[hi2];
          if ((a > b2) == up) {
            s_ids[lo] = b2;
            s_ids[hi2] = a;
          }
        }
        __syncthreads();
      }
    }
  }
  for (int c0 = 0; c0 < C; c0 += 64 * VEC) {
    const int chr = c0 + lane * VEC;
    const bool act = chr < C;
    const int ch = act ? chr : 0;  // idle lanes read channel 0 and discard
    V acc = RowVec<VEC>::zero();
    if (sortable) {
      // wave w sums the contiguous chunk [w*per, (w+1)*per) of the sorted ids, 64 at a time
      const int per = (((k + 3) >> 2) + 63) & ~63;
      const int lo = wave * per;
      const int hi = min(lo + per, k);
      for (int j0 = lo; j0 < hi; j0 += 64) {
        const int kk = min(64, hi - j0);
        const int id = (lane < kk) ? s_ids[j0 + lane] : INT_MAX;
        const typename Src::Meta m = src.prep(id);
        add_rows_wave<VEC>(src, m, kk, ch, acc);
      }
    } else {
      // More points than the LDS sort holds: every wave walks a contiguous slice of this
      // batch's point ids in ascending order and keeps the ones that belong to the cell.
      const int b = cell / nynx;
      const long long p0 = (long long)b * N;
      const int per = (((N + 3) >> 2) + 63) & ~63;
      const int lo = wave * per;
      const int hi = min(lo + per, N);
      for (int i = lo; i < hi; i += 64) {
        const int p = i + lane;
        const bool mt = (p < hi) && (cellid[p0 + p] == cell);
        const unsigned long long mask = __ballot(mt);
        if (mask == 0ull) continue;
        // compact the matching ids to the low lanes, ascending
        const int kk = __popcll(mask);
        const unsigned long long below = (1ull << lane) - 1ull;
        // matching lanes take slots [0,kk) in lane order, the others the distinct slots [kk,64)
        const int dst = mt ? __popcll(mask & below) : kk + __popcll(~mask & below);
        const int pushed = __builtin_amdgcn_ds_permute(dst << 2, mt ? (int)(p0 + p) : INT_MAX);
        const int id = (lane < kk) ? pushed : INT_MAX;
        const typename Src::Meta m = src.prep(id);
        add_rows_wave<VEC>(src, m, kk, ch, acc);
      }
    }
    if (act) *reinterpret_cast<V*>(&s_part[wave][lane * VEC]) = acc;
    __syncthreads();
    if (wave == 0 && act) {
      V tot = *reinterpret_cast<const V*>(&s_part[0][lane * VEC]);
#pragma unroll
      for (int w = 1; w < 4; ++w)
        RowVec<VEC>::add(tot, *reinterpret_cast<const V*>(&s_part[w][lane * VEC]));
      if (!(flags & UD_POOL_OVERWRITE)) {
        V old = *reinterpret_cast<const V*>(orow + ch);
        RowVec<VEC>::add(old, tot);
        tot = old;
      }
      *reinterpret_cast<V*>(orow + ch) = tot;
    }
    __syncthreads();
  }
}

template <int VEC, class Src>
__global__ __launch_bounds__(256) void k_pool(Src src, float* __restrict__ out,
                                              const int* __restrict__ count,
                                              const int* __restrict__ list,
                                              const int* __restrict__ cellid,
                                              const int* __restrict__ heavy_list,
                                              const int* __restrict__ heavy_cnt,
                                              const int2* __restrict__ ovf,
                                              const int* __restrict__ ovf_cnt, int ncell, int N,
                                              int nynx, int C, unsigned flags) {
  using V = typename RowVec<VEC>::T;
  __shared__ int s_ids[kHeavySortMax];
  __shared__ float s_part[4][64 * VEC];
  __shared__ int s_n;
  if (blockIdx.x < kHeavyBlocks) {
    const int nheavy = *heavy_cnt, novf = *ovf_cnt;
    for (int h = blockIdx.x; h < nheavy; h += kHeavyBlocks)
      pool_heavy_cell<VEC>(src, out, count, list, ovf, novf, cellid, heavy_list[h], N, nynx, C, flags, s_ids,
                           s_part, &s_n);
    return;
  }
  // Light role: persistent waves stride over the cells; the next cell's id row and the count of the one after it are
  // prefetched while the current cell's rows stream in.
  const int lane = ud_lane();
  const int nwaves = (gridDim.x - kHeavyBlocks) * 4;
  int cell =
      (blockIdx.x - kHeavyBlocks) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (cell >= ncell) return;
  int k = __builtin_amdgcn_readfirstlane(count[cell]);
  int mine = (k <= kLightMax && lane < k) ? list[(size_t)cell * kRow + lane] : INT_MAX;
  int k1 = 0;
  if (cell + nwaves < ncell) k1 = count[cell + nwaves];
  while (true) {
    const int cell1 = cell + nwaves, cell2 = cell + 2 * nwaves;
    k1 = __builtin_amdgcn_readfirstlane(k1);
    int mine1 = INT_MAX, k2 = 0;
    if (cell1 < ncell && k1 <= kLightMax && lane < k1) mine1 = list[(size_t)cell1 * kRow + lane];
    if (cell2 < ncell) k2 = count[cell2];
    if (k <= kLightMax) {  // else: heavy role owns this cell
      float* orow = out + (size_t)cell * C;
      if (k == 0) {
        if (flags & UD_POOL_OVERWRITE)
          for (int ch = lane * VEC; ch < C; ch += 64 * VEC)
            *reinterpret_cast<V*>(orow + ch) = RowVec<VEC>::zero();
      } else {
        // rank sort: my position = number of ids smaller than mine (ids are distinct)
        int r = 0;
        for (int j = 0; j < k; ++j) r += (__builtin_amdgcn_readlane(mine, j) < mine);
        // push my id to lane r; lanes >= k all push INT_MAX to lane k (never read)
        const int sorted = __builtin_amdgcn_ds_permute(r << 2, mine);
        const typename Src::Meta m = src.prep(lane < k ? sorted : INT_MAX);
        // Wave-uniform channel loop: the cross-lane reads inside add_rows_wave must run with all
        // lanes active (idle lanes compute on channel 0 and discard), otherwise the compiler may
        // sink the per-lane metadata into a divergent region and readlane would see stale lanes.
        for (int c0 = 0; c0 < C; c0 += 64 * VEC) {  // C = 256, VEC = 4: one trip
          const int chr = c0 + lane * VEC;
          const bool act = chr < C;
          const int ch = act ? chr : 0;
          V acc = RowVec<VEC>::zero();
          add_rows_wave<VEC>(src, m, k, ch, acc);
          if (act) {
            if (!(flags & UD_POOL_OVERWRITE)) {
              V old = *reinterpret_cast<const V*>(orow + ch);
              RowVec<VEC>::add(old, acc);
              acc = old;
            }
            *reinterpret_cast<V*>(orow + ch) = acc;
          }
        }
      }
    }
    if (cell1 >= ncell) break;
    cell = cell1;
    k = k1;
    mine = mine1;
    k1 = k2;
  }
}

// ----------------------------------------------------------------------------------------
// Backward: gfeat[row, :] = gout_nhwc[cell(row), :] or 0.  One wave per point row, 4 rows per trip.
template <int VEC>
__global__ __launch_bounds__(256) void k_bwd(const float* __restrict__ gout,
                                             const int32_t* __restrict__ pos,
                                             float* __restrict__ gfeat, long long total, int C,
                                             int nx, int ny) {
  using V = typename RowVec<VEC>::T;
  const int lane = ud_lane();
  const long long wave0 =
      ((long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long row = wave0 + u;
    if (row >= total) break;
    const int b = pos[row * 3 + 0];
    const int y = pos[row * 3 + 1];
    const int x = pos[row * 3 + 2];
    const bool kept = (b >= 0);
    const float* src = gout + ((size_t)((kept ? b : 0) * ny + (kept ? y : 0)) * nx + (kept ? x : 0)) * C;
    float* dst = gfeat + (size_t)row * C;
    for (int ch = lane * VEC; ch < C; ch += 64 * VEC) {
      V v = RowVec<VEC>::zero();
      if (kept) v = *reinterpret_cast<const V*>(src + ch);   // grad rows are re-read: keep cached
      if constexpr (VEC == 4)
        ud_stg_stream(dst + ch, v);                          // 484 MB written once: stream out
      else
        *reinterpret_cast<V*>(dst + ch) = v;
    }
  }
}

// Strided [B, C, ny, nx] (any strides) -> dense NHWC [B, ny, nx, C] through a 32x33 LDS tile.
__global__ __launch_bounds__(256) void k_to_nhwc(const float* __restrict__ src, long long sb,
                                                 long long sc, long long sy, long long sx,
                                                 float* __restrict__ dst, int C, int ny, int nx) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 32;  // pixel tile (y*nx + x)
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31;
  const int ty = threadIdx.x >> 5;  // 0..7
  const int npix = ny * nx;
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i;
    const int p = p0 + tx;
    float v = 0.f;
    if (c < C && p < npix) {
      const int y = p / nx, x = p - y * nx;
      v = src[b * sb + c * sc + y * sy + x * sx];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int p = p0 + i;
    const int c = c0 + tx;
    if (c < C && p < npix) dst[((size_t)b * npix + p) * C + c] = tile[tx][i];
  }
}

struct PoolWs {
  int* count;       // points per cell
  int* heavy_cnt;   // cells with > kLightMax points
  int* ovf_cnt;     // entries of ovf
  int* cellid;      // cell of every point (-1: out of grid); read by the > kHeavySortMax fallback
  int* list;        // [ncell][kRow] point ids by in-cell rank
  int* heavy_list;
  int2* ovf;        // (cell, point id) of the ranks >= kRow
  size_t zero_bytes;  // count, heavy_cnt and ovf_cnt are contiguous and zeroed together
  size_t total_bytes;
};

PoolWs carve(void* ws, int B, int N, int nx, int ny) {
  UdArena a(ws, (size_t)-1);
  const size_t ncell = (size_t)B * ny * nx;
  const size_t total = (size_t)B * N;
  PoolWs w;
  w.count = a.take<int>(ncell);
  w.heavy_cnt = a.take<int>(1);
  w.ovf_cnt = a.take<int>(1);
  w.zero_bytes = a.used;
  w.cellid = a.take<int>(total);
  w.list = a.take<int>(ncell * kRow);
  w.heavy_list = a.take<int>(total / (kLightMax + 1) + 1);
  w.ovf = a.take<int2>(total);
  w.total_bytes = a.used;
  return w;
}

bool sizes_ok(int B, int N, int C, int nx, int ny, int nz) {
  if (B <= 0 || N <= 0 || C <= 0 || nx <= 0 || ny <= 0 || nz <= 0) return false;
  if ((long long)B * N >= INT_MAX) return false;
  if ((long long)B * nx * ny >= INT_MAX) return false;
  return true;
}

}  // namespace

extern "C" size_t ud_bev_pool_workspace_bytes(int B, int N, int C, int nx, int ny, int nz) {
  if (!sizes_ok(B, N, C, nx, ny, nz)) return 0;
  return carve(nullptr, B, N, nx, ny).total_bytes;
}

// Build the per-cell point lists for geom (bins) and write pos_memo.
template <class Bins>
static int build_lists(const Bins& bins, int32_t* pos, int B, int N, int nx, int ny, int nz,
                       const PoolWs& w, hipStream_t stream) {
  const long long total = (long long)B * N;
  UD_HIP_TRY(hipMemsetAsync(w.count, 0, w.zero_bytes, stream));
  k_bin<Bins><<<ud_div_up(total, 256), 256, 0, stream>>>(bins, pos, w.count, w.list, w.cellid, w.heavy_list, w.heavy_cnt,
                                                         w.ovf, w.ovf_cnt, total, N, nx, ny, nz);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

template <int VEC, class Src>
static int launch_pool(const Src& src, float* out, const PoolWs& w, int B, int N, int C, int nx,
                       int ny, unsigned flags, const char* prof_name, hipStream_t stream) {
  const int ncell = B * ny * nx;
  const int grid = kHeavyBlocks + min(ud_div_up(ncell, 4), kLightBlocks);
  UdProfScope prof(prof_name, stream);
  k_pool<VEC, Src><<<grid, 256, 0, stream>>>(src, out, w.count, w.list, w.cellid, w.heavy_list,
                                             w.heavy_cnt, w.ovf, w.ovf_cnt, ncell, N, ny * nx, C, flags);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_bev_pool_fwd(const int32_t* geom, const float* feat, float* out, int32_t* pos,
                               int B, int N, int C, int nx, int ny, int nz, unsigned flags,
                               void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (!sizes_ok(B, N, C, nx, ny, nz) || !geom || !feat || !out || !pos) return UD_ERR_INVALID_ARG;
  if (flags > 1u) return UD_ERR_INVALID_ARG;
  PoolWs w = carve(workspace, B, N, nx, ny);
  if (!workspace || workspace_bytes < w.total_bytes) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = build_lists(BinsLoad{geom}, pos, B, N, nx, ny, nz, w, stream);
  if (rc != UD_OK) return rc;
  const bool vec4 = (C % 4 == 0) && (((uintptr_t)feat | (uintptr_t)out) % 16 == 0);
  if (vec4) {
    SrcFeat<4> src{feat, C};
    return launch_pool<4>(src, out, w, B, N, C, nx, ny, flags, "bev_pool.k_pool", stream);
  }
  SrcFeat<1> src{feat, C};
  return launch_pool<1>(src, out, w, B, N, C, nx, ny, flags, "bev_pool.k_pool", stream);
}

// Fused lift + splat (a7-a9): the [B,N,C] lifted tensor is never materialised.
extern "C" int ud_lss_splat_fwd(const int32_t* geom, const float* prob, const float* ctx_pm,
                                float* out, int32_t* pos, int B, int ncam, int D, int fH, int fW,
                                int C, int nx, int ny, int nz, void* workspace,
                                size_t workspace_bytes, ud_stream_t stream_) {
  if (ncam <= 0 || D <= 0 || fH <= 0 || fW <= 0) return UD_ERR_INVALID_ARG;
  const long long Nll = (long long)ncam * D * fH * fW;
  if (Nll >= INT_MAX) return UD_ERR_INVALID_ARG;
  const int N = (int)Nll;
  if (!sizes_ok(B, N, C, nx, ny, nz) || !geom || !prob || !ctx_pm || !out || !pos)
    return UD_ERR_INVALID_ARG;
  PoolWs w = carve(workspace, B, N, nx, ny);
  if (!workspace || workspace_bytes < w.total_bytes) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int rc = build_lists(BinsLoad{geom}, pos, B, N, nx, ny, nz, w, stream);
  if (rc != UD_OK) return rc;
  const bool vec4 = (C % 4 == 0) && (((uintptr_t)ctx_pm | (uintptr_t)out) % 16 == 0);
  if (vec4) {
    SrcLift<4> src{prob, ctx_pm, C, D * fH * fW, fH * fW};
    return launch_pool<4>(src, out, w, B, N, C, nx, ny, UD_POOL_OVERWRITE, "lss.k_splat", stream);
  }
  SrcLift<1> src{prob, ctx_pm, C, D * fH * fW, fH * fW};
  return launch_pool<1>(src, out, w, B, N, C, nx, ny, UD_POOL_OVERWRITE, "lss.k_splat", stream);
}

// The same with the binning taken from the frustum geometry itself (ud_lss_geometry's arithmetic, lss_geom.h): the [B,N,3] bins
// of the training step are never written or read back, and the launch that produced them is gone.
extern "C" int ud_lss_splat_geom_fwd(const float* mats, const float* frustum_u, const float* frustum_v,
                                     const float* frustum_d, const float* lo, const float* size, int has_bda,
                                     const float* prob, const float* ctx_pm, float* out, int32_t* pos, int B,
                                     int ncam, int D, int fH, int fW, int C, int nx, int ny, int nz,
                                     void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  if (ncam <= 0 || D <= 0 || fH <= 0 || fW <= 0) return UD_ERR_INVALID_ARG;
  const long long Nll = (long long)ncam * D * fH * fW;
  if (Nll >= INT_MAX) return UD_ERR_INVALID_ARG;
  const int N = (int)Nll;
  if (!sizes_ok(B, N, C, nx, ny, nz) || !mats || !frustum_u || !frustum_v || !frustum_d || !lo || !size || !prob ||
      !ctx_pm || !out || !pos)
    return UD_ERR_INVALID_ARG;
  PoolWs w = carve(workspace, B, N, nx, ny);
  if (!workspace || workspace_bytes < w.total_bytes) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const BinsFrustum bins{UdFrustum{mats, frustum_u, frustum_v, frustum_d, D, fH, fW, has_bda, lo[0], lo[1], lo[2], size[0],
                                   size[1], size[2]}};
  int rc = build_lists(bins, pos, B, N, nx, ny, nz, w, stream);
  if (rc != UD_OK) return rc;
  const bool vec4 = (C % 4 == 0) && (((uintptr_t)ctx_pm | (uintptr_t)out) % 16 == 0);
  if (vec4) {
    SrcLift<4> src{prob, ctx_pm, C, D * fH * fW, fH * fW};
    return launch_pool<4>(src, out, w, B, N, C, nx, ny, UD_POOL_OVERWRITE, "lss.k_splat", stream);
  }
  SrcLift<1> src{prob, ctx_pm, C, D * fH * fW, fH * fW};
  return launch_pool<1>(src, out, w, B, N, C, nx, ny, UD_POOL_OVERWRITE, "lss.k_splat", stream);
}

extern "C" size_t ud_bev_pool_bwd_workspace_bytes(int B, int C, int nx, int ny, int64_t sc) {
  if (B <= 0 || C <= 0 || nx <= 0 || ny <= 0) return 0;
  if (sc == 1) return 256;  // NHWC already: nothing to stage
  return ud_align_up((size_t)B * C * nx * ny * sizeof(float));
}

extern "C" int ud_bev_pool_bwd(const float* gout, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                               const int32_t* pos, float* gfeat, int B, int N, int C, int nx,
                               int ny, void* workspace, size_t workspace_bytes,
                               ud_stream_t stream_) {
  if (!sizes_ok(B, N, C, nx, ny, 1) || !gout || !pos || !gfeat) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const float* g = gout;
  const bool dense_nhwc = (sc == 1) && (sx == C) && (sy == (int64_t)C * nx) &&
                          (sb == (int64_t)C * nx * ny);
  if (!dense_nhwc) {
    const size_t need = ud_align_up((size_t)B * C * nx * ny * sizeof(float));
    if (!workspace || workspace_bytes < need) return UD_ERR_WORKSPACE;
    dim3 grid(ud_div_up((long long)nx * ny, 32), ud_div_up(C, 32), B);
    k_to_nhwc<<<grid, 256, 0, stream>>>(gout, sb, sc, sy, sx, (float*)workspace, C, ny, nx);
    UD_LAUNCH_CHECK();
    g = (const float*)workspace;
  }
  const long long total = (long long)B * N;
  const bool vec4 = (C % 4 == 0) && (((uintptr_t)g | (uintptr_t)gfeat) % 16 == 0);
  const int grid = ud_div_up(total, 16);
  UdProfScope prof("bev_pool.k_bwd", stream);
  if (vec4)
    k_bwd<4><<<grid, 256, 0, stream>>>(g, pos, gfeat, total, C, nx, ny);
  else
    k_bwd<1><<<grid, 256, 0, stream>>>(g, pos, gfeat, total, C, nx, ny);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
