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
// The GPU gets the same result without any order-dependent race:
//   k_insert   open-addressing hash insert of the linear key; atomicMin records the FIRST point id
//   k_flag_*   two-level scan over "I am my voxel's first point" flags -> first-appearance rank
//   k_assign   per point: voxel row = sample base + rank; count (wave-aggregated integer atomic);
//              the P smallest point ids of a voxel are kept by an atomicMin insertion chain
//              (slot j always ends up holding the j-th smallest id, whatever the arrival order)
//   k_gather   per voxel: copy the kept points (zero padded), num, and the mean (sum in slot order
//              / max(num,1)) -- voxels[M,P,F] is optional, so the fused path writes M*(F+5) words
//              instead of M*(P*F+4).
#include "ud_common.h"
#include <limits.h>

namespace {

constexpr unsigned kEmpty = 0xFFFFFFFFu;
constexpr int kTile = 1024;  // points per workgroup in the flag scan

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
                                                unsigned* __restrict__ tkey,
                                                unsigned* __restrict__ tfirst,
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
    unsigned h = hash_slot(key, p.tshift);
    while (true) {
      const unsigned old = atomicCAS(&tkey[h], kEmpty, key);
      if (old == kEmpty || old == key) break;
      h = (h + 1) & p.tmask;
    }
    atomicMin(&tfirst[h], (unsigned)gid);
    slot = (int)h;
  }
  pslot[gid] = slot;
}

// exclusive scan of one int per thread across a 256-thread block
__device__ __forceinline__ int block_excl_scan(int v, int* s_w, int& total) {
  const int lane = ud_lane(), wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int a = __shfl_up(inc, o);
    if (lane >= o) inc += a;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int t = s_w[w];
    if (w < wave) base += t;
    tot += t;
  }
  total = tot;
  __syncthreads();
  return base + inc - v;
}

__device__ __forceinline__ int4 load_flags(const unsigned* __restrict__ tfirst,
                                           const int* __restrict__ pslot, long long g0,
                                           long long total) {
  int f[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long g = g0 + u;
    int v = 0;
    if (g < total) {
      const int s = pslot[g];
      v = (s >= 0) && (tfirst[s] == (unsigned)g);
    }
    f[u] = v;
  }
  return make_int4(f[0], f[1], f[2], f[3]);
}

__global__ __launch_bounds__(256) void k_flag_partials(const unsigned* __restrict__ tfirst,
                                                       const int* __restrict__ pslot,
                                                       int* __restrict__ part, long long total) {
  __shared__ int s_w[4];
  const long long g0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const int4 f = load_flags(tfirst, pslot, g0, total);
  int tot;
  block_excl_scan(f.x + f.y + f.z + f.w, s_w, tot);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// first-appearance rank of every voxel (global over the batch) + rank at each sample boundary
__global__ __launch_bounds__(256) void k_flag_ranks(const unsigned* __restrict__ tfirst,
                                                    const int* __restrict__ pslot,
                                                    const int* __restrict__ part,
                                                    int* __restrict__ tvid,
                                                    int* __restrict__ samp_rank, long long total,
                                                    int N, int B) {
  __shared__ int s_w[4];
  int pre = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) pre += part[i];
  int pre_tot;
  block_excl_scan(pre, s_w, pre_tot);
  const long long g0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const int4 f = load_flags(tfirst, pslot, g0, total);
  int tot;
  int run = pre_tot + block_excl_scan(f.x + f.y + f.z + f.w, s_w, tot);
  const int fl[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long g = g0 + u;
    if (g < total) {
      if (g % N == 0) samp_rank[g / N] = run;
      if (fl[u]) tvid[pslot[g]] = run;
      run += fl[u];
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) samp_rank[B] = pre_tot + tot;
}

// Output row base of sample b = sum over earlier samples of min(#voxels, maxM).
__device__ __forceinline__ int sample_row_base(const int* __restrict__ samp_rank, int b, int maxM) {
  int base = 0;
  for (int i = 0; i < b; ++i) base += min(samp_rank[i + 1] - samp_rank[i], maxM);
  return base;
}

__global__ __launch_bounds__(256) void k_assign(const float* __restrict__ pts, VoxParams p,
                                                const unsigned* __restrict__ tfirst,
                                                const int* __restrict__ pslot,
                                                const int* __restrict__ tvid,
                                                const int* __restrict__ samp_rank,
                                                unsigned* __restrict__ top,
                                                unsigned* __restrict__ cnt,
                                                int32_t* __restrict__ coords,
                                                int32_t* __restrict__ m_out) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)p.B * p.N;
  int row = -1;
  if (gid < total) {
    const int s = pslot[gid];
    if (s >= 0) {
      const int b = (int)(gid / p.N);
      const int r = tvid[s] - samp_rank[b];
      if (r < p.maxM) {
        row = sample_row_base(samp_rank, b, p.maxM) + r;
        if (tfirst[s] == (unsigned)gid) {  // first point of the voxel writes its coordinates
          const float* q = pts + gid * p.F;
          int c[3];
#pragma unroll
          for (int a = 0; a < 3; ++a)
            c[a] = (int)floorf(__fdiv_rn(__fsub_rn(q[a], p.lo[a]), p.vs[a]));
          coords[row * 4 + 0] = b;
          coords[row * 4 + 1] = c[2];
          coords[row * 4 + 2] = c[1];
          coords[row * 4 + 3] = c[0];
        }
      }
    }
  }
  if (gid == 0) {  // per-sample voxel counts + total, for the host / downstream kernels
    int tot = 0;
    for (int b = 0; b < p.B; ++b) {
      const int m = min(samp_rank[b + 1] - samp_rank[b], p.maxM);
      m_out[b] = m;
      tot += m;
    }
    m_out[p.B] = tot;
  }
  // Point count per voxel: one integer atomic per run of equal rows inside the wave (zero-padded
  // clouds put thousands of consecutive points into one voxel).
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

__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ pts, VoxParams p,
                                                const unsigned* __restrict__ top,
                                                const unsigned* __restrict__ cnt,
                                                const int32_t* __restrict__ m_out,
                                                float* __restrict__ voxels,
                                                int32_t* __restrict__ num,
                                                float* __restrict__ mean) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= m_out[p.B]) return;
  const int n = min((int)cnt[row], p.P);
  if (num) num[row] = n;
  const unsigned* t = top + (size_t)row * p.P;
  const float inv_den = (float)max(n, 1);
  for (int f = 0; f < p.F; ++f) {
    float acc = 0.0f;
    for (int j = 0; j < p.P; ++j) {
      float v = 0.0f;
      if (j < n) v = pts[(size_t)t[j] * p.F + f];
      if (voxels) voxels[((size_t)row * p.P + j) * p.F + f] = v;
      acc = __fadd_rn(acc, v);
    }
    if (mean) mean[(size_t)row * p.F + f] = __fdiv_rn(acc, inv_den);
  }
}

struct VoxWs {
  unsigned* tkey;    // ---- 0xFF-initialised block
  unsigned* tfirst;
  unsigned* top;
  size_t ff_bytes;
  unsigned* cnt;     // ---- zero-initialised block
  size_t zero_off, zero_bytes;
  int* pslot;
  int* tvid;
  int* part;
  int* samp_rank;
  size_t total_bytes;
  unsigned T;
  int tshift;
  int ntile;
  int cap;
};

VoxWs carve(void* ws, int B, int N, int P, int maxM) {
  UdArena a(ws, (size_t)-1);
  VoxWs w;
  const size_t total = (size_t)B * N;
  unsigned T = 1024;
  int lg = 10;
  while ((size_t)T < 2 * total) {
    T <<= 1;
    ++lg;
  }
  w.T = T;
  w.tshift = 32 - lg;
  w.cap = (int)((size_t)B * maxM < total ? (size_t)B * maxM : total);
  w.ntile = (int)((total + kTile - 1) / kTile);
  w.tkey = a.take<unsigned>(T);
  w.tfirst = a.take<unsigned>(T);
  w.top = a.take<unsigned>((size_t)w.cap * P);
  w.ff_bytes = a.used;
  w.zero_off = a.used;
  w.cnt = a.take<unsigned>(w.cap);
  w.zero_bytes = a.used - w.zero_off;
  w.pslot = a.take<int>(total);
  w.tvid = a.take<int>(T);
  w.part = a.take<int>(w.ntile);
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
  UD_HIP_TRY(hipMemsetAsync(w.tkey, 0xFF, w.ff_bytes, stream));
  UD_HIP_TRY(hipMemsetAsync(w.cnt, 0, w.zero_bytes, stream));
  k_insert<<<ud_div_up(total, 256), 256, 0, stream>>>(points, p, w.tkey, w.tfirst, w.pslot);
  UD_LAUNCH_CHECK();
  k_flag_partials<<<w.ntile, 256, 0, stream>>>(w.tfirst, w.pslot, w.part, total);
  UD_LAUNCH_CHECK();
  k_flag_ranks<<<w.ntile, 256, 0, stream>>>(w.tfirst, w.pslot, w.part, w.tvid, w.samp_rank, total,
                                            N, B);
  UD_LAUNCH_CHECK();
  k_assign<<<ud_div_up(total, 256), 256, 0, stream>>>(points, p, w.tfirst, w.pslot, w.tvid,
                                                      w.samp_rank, w.top, w.cnt, coords, m_out);
  UD_LAUNCH_CHECK();
  k_gather<<<ud_div_up(w.cap, 256), 256, 0, stream>>>(points, p, w.top, w.cnt, m_out, voxels,
                                                      num_points, mean_feats);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
