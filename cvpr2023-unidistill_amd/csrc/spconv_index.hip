// Sparse 3-D convolution bookkeeping for MI355X / gfx950: site index, rulebooks, densify.
//
// Replaces the indexing half of spconv (third-party CUDA, not in the reference tree) as used by
// unidistill/layers/blocks_3d/det3d/spconv_backbone.py:21-48,71-92,259-340 (SubMConv3d /
// SparseConv3d with indice_key) and SparseConvTensor.dense()
// (unidistill/layers/blocks_2d/det3d/map_to_bev/height_compression.py:19).
//
// Design: instead of spconv's hash table + unordered atomically-numbered outputs, every level
// keeps a RANK BITMAP over its voxel grid: one bit per cell + an exclusive popcount prefix per
// 64-bit word.  rank(cell) = prefix[word] + popc(bits below) is the cell's position in ascending
// (b, z, y, x) order, so
//   * the output sites of a strided conv come out sorted and deterministic with no sort,
//   * a neighbour lookup is two dependent loads with good locality (x-neighbours share a word),
//   * rulebooks are dense per-site tables nbr[site][K] (input row or -1), which makes the
//     convolution output-stationary: no scatter-add, no fp atomics, fused epilogues possible.
// For site sets that arrive in arbitrary row order (the voxelizer's first-appearance order) a
// perm[] maps rank -> row.
#include "ud_common.h"
#include "ud_prof.h"
#include <limits.h>

namespace {

constexpr int kWordTile = 1024;  // bitmap words per workgroup in the prefix scan

struct GridShape {
  int B, Dz, Hy, Wx;
  __host__ __device__ long long cells() const { return (long long)B * Dz * Hy * Wx; }
  __host__ __device__ long long nwords() const { return (cells() + 63) >> 6; }
  __device__ long long lin(int b, int z, int y, int x) const {
    return (((long long)b * Dz + z) * Hy + y) * Wx + x;
  }
  __device__ bool inside(int z, int y, int x) const {
    return (unsigned)z < (unsigned)Dz && (unsigned)y < (unsigned)Hy && (unsigned)x < (unsigned)Wx;
  }
};

struct IndexView {
  unsigned long long* words;
  unsigned* prefix;
  int* perm;  // rank -> row, or nullptr when rows are already in rank order
  long long nwords_padded;
};

__host__ __device__ inline long long padded_words(long long nwords) {
  return (nwords + kWordTile - 1) / kWordTile * kWordTile;
}

IndexView index_view(void* base, const GridShape& g, int M, size_t* bytes) {
  IndexView v;
  const long long nwp = padded_words(g.nwords());
  UdArena a(base, (size_t)-1);
  v.words = a.take<unsigned long long>(nwp);
  v.prefix = a.take<unsigned>(nwp);
  v.perm = a.take<int>(M > 0 ? M : 1);
  v.nwords_padded = nwp;
  if (bytes) *bytes = a.used;
  return v;
}

__device__ __forceinline__ int index_lookup(const unsigned long long* __restrict__ words,
                                            const unsigned* __restrict__ prefix,
                                            const int* __restrict__ perm, long long lin) {
  const unsigned long long w = words[lin >> 6];
  const int bit = (int)(lin & 63);
  if (!((w >> bit) & 1ull)) return -1;
  const int rank = (int)prefix[lin >> 6] + __popcll(w & ((1ull << bit) - 1ull));
  return perm ? perm[rank] : rank;
}

// m_dev (all three per-row kernels): the row count lives in device memory (the launch covers an upper bound M) -- the
// voxel count and the level sizes of a LiDAR encoder pass then need ONE host read instead of five.
__global__ __launch_bounds__(256) void k_set_bits(const int32_t* __restrict__ coords, int M,
                                                  const int32_t* __restrict__ m_dev, GridShape g,
                                                  unsigned long long* __restrict__ words) {
  // grid-stride: with a device-side row count the launch covers an upper bound (millions of rows at the deeper levels); a bounded
  // grid keeps the cost of the rows that do not exist at zero instead of tens of microseconds of empty workgroups
  if (m_dev) M = min(M, *m_dev);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
    const int b = coords[i * 4 + 0], z = coords[i * 4 + 1], y = coords[i * 4 + 2],
              x = coords[i * 4 + 3];
    if ((unsigned)b >= (unsigned)g.B || !g.inside(z, y, x)) continue;  // ignored like spconv does
    const long long lin = g.lin(b, z, y, x);
    atomicOr(&words[lin >> 6], 1ull << (lin & 63));
  }
}

__device__ __forceinline__ int block_excl_scan_i(int v, int* s_w, int& total) {
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

__global__ __launch_bounds__(256) void k_word_partials(const unsigned long long* __restrict__ words,
                                                       int* __restrict__ part) {
  __shared__ int s_w[4];
  const ulonglong4 w = reinterpret_cast<const ulonglong4*>(words)[(size_t)blockIdx.x * 256 + threadIdx.x];
  const int c = __popcll(w.x) + __popcll(w.y) + __popcll(w.z) + __popcll(w.w);
  int tot;
  block_excl_scan_i(c, s_w, tot);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_word_prefix(const unsigned long long* __restrict__ words,
                                                     const int* __restrict__ part,
                                                     unsigned* __restrict__ prefix,
                                                     int* __restrict__ total_out) {
  __shared__ int s_w[4];
  int pre = 0;
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) pre += part[i];
  int pre_tot;
  block_excl_scan_i(pre, s_w, pre_tot);
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  const ulonglong4 w = reinterpret_cast<const ulonglong4*>(words)[q];
  const int c0 = __popcll(w.x), c1 = __popcll(w.y), c2 = __popcll(w.z), c3 = __popcll(w.w);
  int tot;
  const int ex = pre_tot + block_excl_scan_i(c0 + c1 + c2 + c3, s_w, tot);
  reinterpret_cast<uint4*>(prefix)[q] = make_uint4(ex, ex + c0, ex + c0 + c1, ex + c0 + c1 + c2);
  if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *total_out = pre_tot + tot;
}

__global__ __launch_bounds__(256) void k_fill_perm(const int32_t* __restrict__ coords, int M,
                                                   const int32_t* __restrict__ m_dev, GridShape g,
                                                   const unsigned long long* __restrict__ words,
                                                   const unsigned* __restrict__ prefix,
                                                   int* __restrict__ perm) {
  if (m_dev) M = min(M, *m_dev);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {
    const int b = coords[i * 4 + 0], z = coords[i * 4 + 1], y = coords[i * 4 + 2],
              x = coords[i * 4 + 3];
    if ((unsigned)b >= (unsigned)g.B || !g.inside(z, y, x)) continue;
    const long long lin = g.lin(b, z, y, x);
    const unsigned long long w = words[lin >> 6];
    const int bit = (int)(lin & 63);
    perm[(int)prefix[lin >> 6] + __popcll(w & ((1ull << bit) - 1ull))] = i;
  }
}

// Submanifold rulebook: nbr[o][k] = row of the active site at coord(o) + (k - centre), else -1.
__global__ __launch_bounds__(256) void k_subm_rulebook(const int32_t* __restrict__ coords, int M,
                                                       GridShape g, int kz, int ky, int kx,
                                                       const unsigned long long* __restrict__ words,
                                                       const unsigned* __restrict__ prefix,
                                                       const int* __restrict__ perm,
                                                       int32_t* __restrict__ nbr) {
  const int K = kz * ky * kx;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)M * K) return;
  const int o = (int)(t / K), k = (int)(t - (long long)o * K);
  const int dz = k / (ky * kx) - kz / 2, dy = (k / kx) % ky - ky / 2, dx = k % kx - kx / 2;
  const int b = coords[o * 4 + 0];
  const int z = coords[o * 4 + 1] + dz, y = coords[o * 4 + 2] + dy, x = coords[o * 4 + 3] + dx;
  int r = -1;
  if (g.inside(z, y, x)) r = index_lookup(words, prefix, perm, g.lin(b, z, y, x));
  nbr[t] = r;
}

struct ConvGeom {
  int k[3], s[3], p[3];  // z, y, x
};

// Mark every output cell reachable from an active input under (kernel, stride, padding).
__global__ __launch_bounds__(256) void k_mark_outputs(const int32_t* __restrict__ coords, int M,
                                                      const int32_t* __restrict__ m_dev, GridShape gin,
                                                      GridShape gout, ConvGeom c,
                                                      unsigned long long* __restrict__ words) {
  if (m_dev) M = min(M, *m_dev);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < M; i += gridDim.x * 256) {     // grid-stride: see k_set_bits
    const int b = coords[i * 4 + 0];
    const int in[3] = {coords[i * 4 + 1], coords[i * 4 + 2], coords[i * 4 + 3]};
    if ((unsigned)b >= (unsigned)gin.B || !gin.inside(in[0], in[1], in[2])) continue;
    for (int a = 0; a < c.k[0]; ++a) {
      const int tz = in[0] + c.p[0] - a;
      if (tz < 0 || tz % c.s[0]) continue;
      const int oz = tz / c.s[0];
      if (oz >= gout.Dz) continue;
      for (int e = 0; e < c.k[1]; ++e) {
        const int ty = in[1] + c.p[1] - e;
        if (ty < 0 || ty % c.s[1]) continue;
        const int oy = ty / c.s[1];
        if (oy >= gout.Hy) continue;
        for (int f = 0; f < c.k[2]; ++f) {
          const int tx = in[2] + c.p[2] - f;
          if (tx < 0 || tx % c.s[2]) continue;
          const int ox = tx / c.s[2];
          if (ox >= gout.Wx) continue;
          const long long lin = gout.lin(b, oz, oy, ox);
          const unsigned long long bit = 1ull << (lin & 63);
          // an output cell is reached by several inputs: only the first arrival needs the atomic
          if (!(__builtin_nontemporal_load(&words[lin >> 6]) & bit)) atomicOr(&words[lin >> 6], bit);
        }
      }
    }
  }
}

// One thread per bitmap word: emit the coordinates of its set bits at their ranks.
__global__ __launch_bounds__(256) void k_emit_coords(const unsigned long long* __restrict__ words,
                                                     const unsigned* __restrict__ prefix,
                                                     long long nwords, GridShape g, int cap,
                                                     int32_t* __restrict__ coords) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= nwords) return;
  unsigned long long w = words[q];
  int r = (int)prefix[q];
  while (w) {
    const int bit = __ffsll((long long)w) - 1;
    w &= w - 1;
    long long lin = (q << 6) + bit;
    const int x = (int)(lin % g.Wx);
    lin /= g.Wx;
    const int y = (int)(lin % g.Hy);
    lin /= g.Hy;
    const int z = (int)(lin % g.Dz);
    const int b = (int)(lin / g.Dz);
    if (r < cap) {
      coords[r * 4 + 0] = b;
      coords[r * 4 + 1] = z;
      coords[r * 4 + 2] = y;
      coords[r * 4 + 3] = x;
    }
    ++r;
  }
}

// Strided-conv rulebook: out_nbr[o][k] = input row at o*s - p + k (or -1); in_nbr[i][k] = o.
__global__ __launch_bounds__(256) void k_down_rulebook(const int32_t* __restrict__ out_coords,
                                                       int Mout, GridShape gin, ConvGeom c,
                                                       const unsigned long long* __restrict__ words,
                                                       const unsigned* __restrict__ prefix,
                                                       const int* __restrict__ perm,
                                                       int32_t* __restrict__ out_nbr,
                                                       int32_t* __restrict__ in_nbr) {
  const int K = c.k[0] * c.k[1] * c.k[2];
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)Mout * K) return;
  const int o = (int)(t / K), k = (int)(t - (long long)o * K);
  const int a = k / (c.k[1] * c.k[2]), e = (k / c.k[2]) % c.k[1], f = k % c.k[2];
  const int b = out_coords[o * 4 + 0];
  const int z = out_coords[o * 4 + 1] * c.s[0] - c.p[0] + a;
  const int y = out_coords[o * 4 + 2] * c.s[1] - c.p[1] + e;
  const int x = out_coords[o * 4 + 3] * c.s[2] - c.p[2] + f;
  int r = -1;
  if (gin.inside(z, y, x)) r = index_lookup(words, prefix, perm, gin.lin(b, z, y, x));
  out_nbr[t] = r;
  if (r >= 0 && in_nbr) in_nbr[(long long)r * K + k] = o;
}

// dense[b, c, z, y, x] = feat[row(b,z,y,x), c] (0 where no voxel): NCDHW is x-fastest, the feature rows
// are c-fastest, so a (whole BEV row of <= 192 cells) x 64-channel tile is transposed through LDS: the feature
// side moves 256-byte rows, the dense side one contiguous 4*Wx-byte run per channel.  A workgroup owns one
// (b, z, y) line and walks the channel chunks; the dense
// tensor is written exactly once, zeros included (no memset of the 33 MB-per-sample tensor).  BWD reads
// the dense gradient the same way and writes the rows of the occupied cells.
constexpr int kDenseX = 192;   // cells of one (b, z, y) line per workgroup: the whole 180-wide BEV row -> 720-byte runs
template <bool BWD>
__global__ __launch_bounds__(256) void k_dense(float* __restrict__ feat,
                                               const int32_t* __restrict__ rowmap, int C,
                                               GridShape g, float* __restrict__ dense) {
  __shared__ float s_t[kDenseX][65];
  __shared__ int s_row[kDenseX];
  const int lane = ud_lane(), wv = threadIdx.x >> 6;
  const int x0 = blockIdx.x * kDenseX;
  const int line = blockIdx.y;                    // (b * Dz + z) * Hy + y
  const int y = line % g.Hy, z = (line / g.Hy) % g.Dz, b = line / (g.Hy * g.Dz);
  const int nx = min(kDenseX, g.Wx - x0);
  if (threadIdx.x < kDenseX)
    s_row[threadIdx.x] = (threadIdx.x < nx) ? rowmap[(long long)line * g.Wx + x0 + threadIdx.x] : -1;
  __syncthreads();
  const size_t plane = (size_t)g.Dz * g.Hy * g.Wx;
  const size_t base = (size_t)b * C * plane + ((size_t)z * g.Hy + y) * g.Wx + x0;
  // every channel run starts on a 16-byte boundary and is a whole number of 16-byte pieces (<= 64 of them)
  const bool vec4 = !BWD && (g.Wx & 3) == 0 && (nx & 3) == 0 && (x0 & 3) == 0 && nx <= 256 &&
                    (reinterpret_cast<size_t>(dense) & 15) == 0;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int nc = min(64, C - c0);
    if (!BWD) {
      float v[kDenseX / 4];                       // a wave stages whole 256-byte feature rows, lanes along c:
#pragma unroll                                    // all 48 row loads in flight before the first LDS write
      for (int j = 0; j < kDenseX / 4; ++j) {
        const int x = wv + 4 * j;
        const int row = x < nx ? s_row[x] : -1;
        v[j] = (row >= 0 && lane < nc) ? feat[(size_t)row * C + c0 + lane] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < kDenseX / 4; ++j)
        if (wv + 4 * j < nx) s_t[wv + 4 * j][lane] = v[j];
      __syncthreads();
      if (vec4) {                                  // 16-byte stores: a 720-byte run is 45 of them (one per lane)
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
          const int c = wv * 16 + i, x = 4 * lane;
          if (c < nc && x < nx) {
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 q = {s_t[x][c], s_t[x + 1][c], s_t[x + 2][c], s_t[x + 3][c]};
            __builtin_nontemporal_store(q, reinterpret_cast<f32x4*>(&dense[base + (size_t)(c0 + c) * plane + x]));
          }
        }
      } else {
#pragma unroll 2
        for (int i = 0; i < 16; ++i) {            // wave wv stores channels wv*16 .. +15: one 4*nx-byte run each
          const int c = wv * 16 + i;
          if (c < nc)
            for (int x = lane; x < nx; x += 64)
              __builtin_nontemporal_store(s_t[x][c], &dense[base + (size_t)(c0 + c) * plane + x]);
        }
      }
      __syncthreads();
    } else {
      float v[16][kDenseX / 64];                  // 48 independent loads per lane, then the LDS writes
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < kDenseX / 64; ++k) {
          const int c = wv * 16 + i, x = lane + 64 * k;
          v[i][k] = (c < nc && x < nx) ? __builtin_nontemporal_load(&dense[base + (size_t)(c0 + c) * plane + x]) : 0.0f;
        }
#pragma unroll
      for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int k = 0; k < kDenseX / 64; ++k)
          if (lane + 64 * k < nx) s_t[lane + 64 * k][wv * 16 + i] = v[i][k];
      __syncthreads();
#pragma unroll 4
      for (int x = wv; x < nx; x += 4) {
        const int row = s_row[x];
        if (row >= 0 && lane < nc) feat[(size_t)row * C + c0 + lane] = s_t[x][lane];
      }
      __syncthreads();
    }
  }
}

bool shape_ok(const GridShape& g) {
  return g.B > 0 && g.Dz > 0 && g.Hy > 0 && g.Wx > 0 && g.cells() < (1ll << 40);
}

int scan_words(const IndexView& v, const GridShape& g, int* part, int* total_out,
               hipStream_t stream) {
  const int ntile = (int)(v.nwords_padded / kWordTile);
  k_word_partials<<<ntile, 256, 0, stream>>>(v.words, part);
  UD_LAUNCH_CHECK();
  k_word_prefix<<<ntile, 256, 0, stream>>>(v.words, part, v.prefix, total_out);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

}  // namespace

extern "C" size_t ud_spconv_index_bytes(int B, int Dz, int Hy, int Wx, int M) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || M < 0) return 0;
  size_t bytes = 0;
  index_view(nullptr, g, M, &bytes);
  // + the scan partials live behind the index so one buffer carries everything
  return bytes + ud_align_up((size_t)(padded_words(g.nwords()) / kWordTile) * sizeof(int));
}

// Build the rank index of the site set coords i32[M,4] (b,z,y,x) on grid (B,Dz,Hy,Wx).
// rows_sorted != 0: row i already IS rank i (outputs of ud_spconv_down_outputs) -> no perm.
static int build_index_impl(const int32_t* coords, int M, const int32_t* m_dev, int B, int Dz, int Hy, int Wx,
                            int rows_sorted, void* index, size_t index_bytes, ud_stream_t stream_);

extern "C" int ud_spconv_build_index(const int32_t* coords, int M, int B, int Dz, int Hy, int Wx,
                                     int rows_sorted, void* index, size_t index_bytes,
                                     ud_stream_t stream_) {
  return build_index_impl(coords, M, nullptr, B, Dz, Hy, Wx, rows_sorted, index, index_bytes, stream_);
}

// Same, with the row count in device memory: rows [0, min(*m_dev, M_cap)) of coords are indexed (index sized for M_cap).
extern "C" int ud_spconv_build_index_dev(const int32_t* coords, const int32_t* m_dev, int M_cap, int B, int Dz, int Hy,
                                         int Wx, int rows_sorted, void* index, size_t index_bytes,
                                         ud_stream_t stream_) {
  if (!m_dev) return UD_ERR_INVALID_ARG;
  return build_index_impl(coords, M_cap, m_dev, B, Dz, Hy, Wx, rows_sorted, index, index_bytes, stream_);
}

static int build_index_impl(const int32_t* coords, int M, const int32_t* m_dev, int B, int Dz, int Hy, int Wx,
                            int rows_sorted, void* index, size_t index_bytes, ud_stream_t stream_) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || M < 0 || (M > 0 && !coords) || !index) return UD_ERR_INVALID_ARG;
  if (index_bytes < ud_spconv_index_bytes(B, Dz, Hy, Wx, M)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  size_t used = 0;
  IndexView v = index_view(index, g, M, &used);
  int* part = (int*)((char*)index + used);
  UD_HIP_TRY(hipMemsetAsync(v.words, 0, v.nwords_padded * sizeof(unsigned long long), stream));
  if (M > 0) {
    k_set_bits<<<min(ud_div_up(M, 256), 4096), 256, 0, stream>>>(coords, M, m_dev, g, v.words);
    UD_LAUNCH_CHECK();
  }
  int rc = scan_words(v, g, part, nullptr, stream);
  if (rc != UD_OK) return rc;
  if (!rows_sorted && M > 0) {
    k_fill_perm<<<min(ud_div_up(M, 256), 4096), 256, 0, stream>>>(coords, M, m_dev, g, v.words, v.prefix, v.perm);
    UD_LAUNCH_CHECK();
  }
  return UD_OK;
}

extern "C" int ud_spconv_subm_rulebook(const void* index, int rows_sorted, const int32_t* coords,
                                       int M, int B, int Dz, int Hy, int Wx, int kz, int ky, int kx,
                                       int32_t* nbr, ud_stream_t stream_) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || M < 0 || !index || (M > 0 && (!coords || !nbr))) return UD_ERR_INVALID_ARG;
  if (kz <= 0 || ky <= 0 || kx <= 0 || !(kz & ky & kx & 1)) return UD_ERR_INVALID_ARG;  // odd sizes
  if (M == 0) return UD_OK;
  IndexView v = index_view((void*)index, g, M, nullptr);
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)M * kz * ky * kx;
  k_subm_rulebook<<<ud_div_up(total, 256), 256, 0, stream>>>(
      coords, M, g, kz, ky, kx, v.words, v.prefix, rows_sorted ? nullptr : v.perm, nbr);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// Output sites of SparseConv3d(kernel, stride, padding) on the input site set: builds the OUTPUT
// level's rank index (rows sorted), writes out_coords (up to out_cap rows) and the count m_out
// (device int).  Output grid = floor((in + 2p - k) / s) + 1 per axis.
static int down_outputs_impl(const int32_t* in_coords, int Min, const int32_t* min_dev, int B, int Dz, int Hy,
                             int Wx, const int* ksize, const int* stride, const int* pad, void* out_index,
                             size_t out_index_bytes, int32_t* out_coords, int out_cap, int32_t* m_out,
                             ud_stream_t stream_);

extern "C" int ud_spconv_down_outputs(const int32_t* in_coords, int Min, int B, int Dz, int Hy,
                                      int Wx, const int* ksize, const int* stride, const int* pad,
                                      void* out_index, size_t out_index_bytes, int32_t* out_coords,
                                      int out_cap, int32_t* m_out, ud_stream_t stream_) {
  return down_outputs_impl(in_coords, Min, nullptr, B, Dz, Hy, Wx, ksize, stride, pad, out_index, out_index_bytes,
                           out_coords, out_cap, m_out, stream_);
}

// Same, with the input row count in device memory (min(*min_dev, Min_cap) rows of in_coords are used): levels chain on the
// device, m_out of one call is min_dev of the next.
extern "C" int ud_spconv_down_outputs_dev(const int32_t* in_coords, const int32_t* min_dev, int Min_cap, int B, int Dz,
                                          int Hy, int Wx, const int* ksize, const int* stride, const int* pad,
                                          void* out_index, size_t out_index_bytes, int32_t* out_coords,
                                          int out_cap, int32_t* m_out, ud_stream_t stream_) {
  if (!min_dev) return UD_ERR_INVALID_ARG;
  return down_outputs_impl(in_coords, Min_cap, min_dev, B, Dz, Hy, Wx, ksize, stride, pad, out_index, out_index_bytes,
                           out_coords, out_cap, m_out, stream_);
}

static int down_outputs_impl(const int32_t* in_coords, int Min, const int32_t* min_dev, int B, int Dz, int Hy,
                             int Wx, const int* ksize, const int* stride, const int* pad, void* out_index,
                             size_t out_index_bytes, int32_t* out_coords, int out_cap, int32_t* m_out,
                             ud_stream_t stream_) {
  if (!ksize || !stride || !pad || !out_index || !m_out || Min < 0) return UD_ERR_INVALID_ARG;
  GridShape gin{B, Dz, Hy, Wx};
  ConvGeom c;
  int od[3];
  const int in_d[3] = {Dz, Hy, Wx};
  for (int a = 0; a < 3; ++a) {
    c.k[a] = ksize[a];
    c.s[a] = stride[a];
    c.p[a] = pad[a];
    if (c.k[a] <= 0 || c.s[a] <= 0 || c.p[a] < 0) return UD_ERR_INVALID_ARG;
    od[a] = (in_d[a] + 2 * c.p[a] - c.k[a]) / c.s[a] + 1;
    if (od[a] <= 0) return UD_ERR_INVALID_ARG;
  }
  GridShape gout{B, od[0], od[1], od[2]};
  if (!shape_ok(gin) || !shape_ok(gout)) return UD_ERR_INVALID_ARG;
  if (out_index_bytes < ud_spconv_index_bytes(B, od[0], od[1], od[2], out_cap)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  size_t used = 0;
  IndexView v = index_view(out_index, gout, out_cap, &used);
  int* part = (int*)((char*)out_index + used);
  UD_HIP_TRY(hipMemsetAsync(v.words, 0, v.nwords_padded * sizeof(unsigned long long), stream));
  if (Min > 0) {
    k_mark_outputs<<<min(ud_div_up(Min, 256), 4096), 256, 0, stream>>>(in_coords, Min, min_dev, gin, gout, c, v.words);
    UD_LAUNCH_CHECK();
  }
  int rc = scan_words(v, gout, part, m_out, stream);
  if (rc != UD_OK) return rc;
  if (out_coords && out_cap > 0) {
    k_emit_coords<<<ud_div_up(gout.nwords(), 256), 256, 0, stream>>>(v.words, v.prefix,
                                                                     gout.nwords(), gout, out_cap,
                                                                     out_coords);
    UD_LAUNCH_CHECK();
  }
  return UD_OK;
}

// Rulebooks of the strided conv: out_nbr i32[Mout,K] (input row or -1) and, when in_nbr != NULL,
// in_nbr i32[Min,K] (output row fed by input i through kernel offset k, or -1) for dgrad.
extern "C" int ud_spconv_down_rulebook(const void* in_index, int in_rows_sorted, int Min, int B,
                                       int Dz, int Hy, int Wx, const int* ksize, const int* stride,
                                       const int* pad, const int32_t* out_coords, int Mout,
                                       int32_t* out_nbr, int32_t* in_nbr, ud_stream_t stream_) {
  if (!in_index || !ksize || !stride || !pad || Mout < 0 || Min < 0) return UD_ERR_INVALID_ARG;
  GridShape gin{B, Dz, Hy, Wx};
  if (!shape_ok(gin)) return UD_ERR_INVALID_ARG;
  ConvGeom c;
  for (int a = 0; a < 3; ++a) {
    c.k[a] = ksize[a];
    c.s[a] = stride[a];
    c.p[a] = pad[a];
  }
  hipStream_t stream = (hipStream_t)stream_;
  const int K = c.k[0] * c.k[1] * c.k[2];
  if (in_nbr && Min > 0)
    UD_HIP_TRY(hipMemsetAsync(in_nbr, 0xFF, (size_t)Min * K * sizeof(int32_t), stream));
  if (Mout == 0) return UD_OK;
  if (!out_coords || !out_nbr) return UD_ERR_INVALID_ARG;
  IndexView v = index_view((void*)in_index, gin, Min, nullptr);
  k_down_rulebook<<<ud_div_up((long long)Mout * K, 256), 256, 0, stream>>>(
      out_coords, Mout, gin, c, v.words, v.prefix, in_rows_sorted ? nullptr : v.perm, out_nbr,
      in_nbr);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// ---- HeightCompression for the mixed-precision path ----------------------------------------------------
// bev[b][y][x][c * Dz + z] = feat[row(b,z,y,x)][c] (0 where no voxel) as a channels-last bf16 map: what
// `dense()` + `view(N, C*D, H, W)` (reference height_compression.py:19-22) + the trunk's bf16 cast produce,
// written once (no fp32 NCDHW tensor, no memset of it, no cast pass).  A small row map (cell -> row)
// turns the scatter into a per-pixel gather with fully coalesced stores.
__global__ __launch_bounds__(256) void k_fill_rowmap(const int32_t* __restrict__ coords, int M, GridShape g,
                                                     int32_t* __restrict__ rowmap) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int b = coords[i * 4 + 0], z = coords[i * 4 + 1], y = coords[i * 4 + 2], x = coords[i * 4 + 3];
  if ((unsigned)b < (unsigned)g.B && g.inside(z, y, x)) rowmap[g.lin(b, z, y, x)] = i;
}

template <bool BWD>
__global__ __launch_bounds__(256) void k_bev_nhwc(unsigned short* __restrict__ feat,
                                                  const int32_t* __restrict__ rowmap, int C, GridShape g,
                                                  unsigned short* __restrict__ bev) {
  const int CO = C * g.Dz, pieces = CO / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)g.B * g.Hy * g.Wx * pieces;
  if (t >= total) return;
  const int piece = (int)(t % pieces);
  const long long pix = t / pieces;
  const int x = (int)(pix % g.Wx), y = (int)((pix / g.Wx) % g.Hy), b = (int)(pix / ((long long)g.Wx * g.Hy));
  unsigned short v[4] = {0, 0, 0, 0};
  unsigned short* dst = bev + pix * CO + piece * 4;
  if (BWD) {
    const uint2 gpk = *reinterpret_cast<const uint2*>(dst);
    v[0] = gpk.x & 0xFFFFu; v[1] = gpk.x >> 16; v[2] = gpk.y & 0xFFFFu; v[3] = gpk.y >> 16;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int o = piece * 4 + e, c = o / g.Dz, z = o - c * g.Dz;
    const int row = rowmap[g.lin(b, z, y, x)];
    if (row >= 0) {
      if (BWD) feat[(size_t)row * C + c] = v[e];
      else v[e] = feat[(size_t)row * C + c];
    }
  }
  if (!BWD) *reinterpret_cast<uint2*>(dst) = make_uint2(v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16));
}

// fp32 twin (the reference's arithmetic): the channels-last fp32 BEV map the trunk's kernels read, written once -- instead of the
// NCDHW dense() tensor plus an NCHW -> NHWC copy of it (47 + 86 us at B = 4).
template <bool BWD>
__global__ __launch_bounds__(256) void k_bev_nhwc_f32(float* __restrict__ feat, const int32_t* __restrict__ rowmap, int C,
                                                      GridShape g, float* __restrict__ bev) {
  const int CO = C * g.Dz, pieces = CO / 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)g.B * g.Hy * g.Wx * pieces;
  if (t >= total) return;
  const int piece = (int)(t % pieces);
  const long long pix = t / pieces;
  const int x = (int)(pix % g.Wx), y = (int)((pix / g.Wx) % g.Hy), b = (int)(pix / ((long long)g.Wx * g.Hy));
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  float* dst = bev + pix * CO + piece * 4;
  if (BWD) {
    const float4 gq = *reinterpret_cast<const float4*>(dst);
    v[0] = gq.x; v[1] = gq.y; v[2] = gq.z; v[3] = gq.w;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int o = piece * 4 + e, c = o / g.Dz, z = o - c * g.Dz;
    const int row = rowmap[g.lin(b, z, y, x)];
    if (row >= 0) {
      if (BWD) feat[(size_t)row * C + c] = v[e];
      else v[e] = feat[(size_t)row * C + c];
    }
  }
  if (!BWD) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((f32x4v){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4v*>(dst));
  }
}

extern "C" size_t ud_sparse_bev_workspace_bytes(int B, int Dz, int Hy, int Wx) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g)) return 0;
  return ud_align_up((size_t)g.cells() * sizeof(int32_t));
}

// feat bf16[M,C], coords i32[M,4] -> bev bf16[B,Hy,Wx,C*Dz].  (C * Dz) % 4 == 0.
extern "C" int ud_sparse_to_bev_bf16(const void* feat, const int32_t* coords, int M, int C, int B, int Dz,
                                     int Hy, int Wx, void* bev, void* workspace, size_t workspace_bytes,
                                     ud_stream_t stream_) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || C <= 0 || M < 0 || !bev) return UD_ERR_INVALID_ARG;
  if ((C * Dz) % 4) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx)) return UD_ERR_WORKSPACE;
  if (M > 0 && (!feat || !coords)) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  int32_t* rowmap = reinterpret_cast<int32_t*>(workspace);
  UD_HIP_TRY(hipMemsetAsync(rowmap, 0xFF, (size_t)g.cells() * sizeof(int32_t), stream));
  if (M > 0) {
    k_fill_rowmap<<<ud_div_up(M, 256), 256, 0, stream>>>(coords, M, g, rowmap);
    UD_LAUNCH_CHECK();
  }
  const long long total = (long long)B * Hy * Wx * (C * Dz / 4);
  k_bev_nhwc<false><<<ud_div_up(total, 256), 256, 0, stream>>>((unsigned short*)feat, rowmap, C, g,
                                                               (unsigned short*)bev);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// Backward: gfeat bf16[M,C] (every row of an active voxel is written) from gbev bf16[B,Hy,Wx,C*Dz].
extern "C" int ud_bev_to_sparse_bf16(const void* gbev, const int32_t* coords, int M, int C, int B, int Dz,
                                     int Hy, int Wx, void* gfeat, void* workspace, size_t workspace_bytes,
                                     ud_stream_t stream_) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || C <= 0 || M < 0) return UD_ERR_INVALID_ARG;
  if ((C * Dz) % 4) return UD_ERR_UNSUPPORTED;
  if (M == 0) return UD_OK;
  if (!gbev || !coords || !gfeat) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int32_t* rowmap = reinterpret_cast<int32_t*>(workspace);
  UD_HIP_TRY(hipMemsetAsync(rowmap, 0xFF, (size_t)g.cells() * sizeof(int32_t), stream));
  k_fill_rowmap<<<ud_div_up(M, 256), 256, 0, stream>>>(coords, M, g, rowmap);
  UD_LAUNCH_CHECK();
  const long long total = (long long)B * Hy * Wx * (C * Dz / 4);
  k_bev_nhwc<true><<<ud_div_up(total, 256), 256, 0, stream>>>((unsigned short*)gfeat, rowmap, C, g,
                                                              (unsigned short*)gbev);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// fp32: feat f32[M,C] -> bev f32[B,Hy,Wx,C*Dz], and its backward (gfeat rows of active voxels are written; the caller zeroes gfeat).
static int bev_f32_impl(float* feat, const int32_t* coords, int M, int C, int B, int Dz, int Hy, int Wx, float* bev,
                        void* workspace, size_t workspace_bytes, bool bwd, hipStream_t stream) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || C <= 0 || M < 0 || !bev) return UD_ERR_INVALID_ARG;
  if ((C * Dz) % 4) return UD_ERR_UNSUPPORTED;
  if (bwd && M == 0) return UD_OK;
  if (M > 0 && (!feat || !coords)) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx)) return UD_ERR_WORKSPACE;
  int32_t* rowmap = reinterpret_cast<int32_t*>(workspace);
  UD_HIP_TRY(hipMemsetAsync(rowmap, 0xFF, (size_t)g.cells() * sizeof(int32_t), stream));
  if (M > 0) {
    k_fill_rowmap<<<ud_div_up(M, 256), 256, 0, stream>>>(coords, M, g, rowmap);
    UD_LAUNCH_CHECK();
  }
  const long long total = (long long)B * Hy * Wx * (C * Dz / 4);
  UdProfScope prof("spconv.k_bev_f32", stream);
  if (bwd) k_bev_nhwc_f32<true><<<ud_div_up(total, 256), 256, 0, stream>>>(feat, rowmap, C, g, bev);
  else k_bev_nhwc_f32<false><<<ud_div_up(total, 256), 256, 0, stream>>>(feat, rowmap, C, g, bev);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_sparse_to_bev_f32(const float* feat, const int32_t* coords, int M, int C, int B, int Dz, int Hy, int Wx,
                                    float* bev, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  return bev_f32_impl(const_cast<float*>(feat), coords, M, C, B, Dz, Hy, Wx, bev, workspace, workspace_bytes, false,
                      (hipStream_t)stream_);
}

extern "C" int ud_bev_to_sparse_f32(const float* gbev, const int32_t* coords, int M, int C, int B, int Dz, int Hy, int Wx,
                                    float* gfeat, void* workspace, size_t workspace_bytes, ud_stream_t stream_) {
  return bev_f32_impl(gfeat, coords, M, C, B, Dz, Hy, Wx, const_cast<float*>(gbev), workspace, workspace_bytes, true,
                      (hipStream_t)stream_);
}

// SparseConvTensor.dense(): dense f32[B, C, Dz, Hy, Wx] = 0 everywhere, feat[row, :] at coords.
// workspace: ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx) (the cell -> row map).
extern "C" int ud_sparse_to_dense(const float* feat, const int32_t* coords, int M, int C, int B,
                                  int Dz, int Hy, int Wx, float* dense, void* workspace,
                                  size_t workspace_bytes, ud_stream_t stream_) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || C <= 0 || M < 0 || !dense) return UD_ERR_INVALID_ARG;
  if ((long long)B * Dz * Hy > 65535ll) return UD_ERR_UNSUPPORTED;   // grid.y
  if (M > 0 && (!feat || !coords)) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int32_t* rowmap = reinterpret_cast<int32_t*>(workspace);
  UD_HIP_TRY(hipMemsetAsync(rowmap, 0xFF, (size_t)g.cells() * sizeof(int32_t), stream));
  if (M > 0) {
    k_fill_rowmap<<<ud_div_up(M, 256), 256, 0, stream>>>(coords, M, g, rowmap);
    UD_LAUNCH_CHECK();
  }
  UdProfScope prof("spconv.k_dense", stream);
  k_dense<false><<<dim3(ud_div_up(Wx, kDenseX), B * Dz * Hy), 256, 0, stream>>>((float*)feat, rowmap, C, g, dense);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// Backward of dense(): gfeat[row, :] = gdense[b, :, z, y, x].  Same workspace.
extern "C" int ud_dense_to_sparse(const float* gdense, const int32_t* coords, int M, int C, int B,
                                  int Dz, int Hy, int Wx, float* gfeat, void* workspace,
                                  size_t workspace_bytes, ud_stream_t stream_) {
  GridShape g{B, Dz, Hy, Wx};
  if (!shape_ok(g) || C <= 0 || M < 0) return UD_ERR_INVALID_ARG;
  if (M == 0) return UD_OK;
  if (!gdense || !coords || !gfeat) return UD_ERR_INVALID_ARG;
  if ((long long)B * Dz * Hy > 65535ll) return UD_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < ud_sparse_bev_workspace_bytes(B, Dz, Hy, Wx)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  int32_t* rowmap = reinterpret_cast<int32_t*>(workspace);
  UD_HIP_TRY(hipMemsetAsync(rowmap, 0xFF, (size_t)g.cells() * sizeof(int32_t), stream));
  k_fill_rowmap<<<ud_div_up(M, 256), 256, 0, stream>>>(coords, M, g, rowmap);
  UD_LAUNCH_CHECK();
  k_dense<true><<<dim3(ud_div_up(Wx, kDenseX), B * Dz * Hy), 256, 0, stream>>>(gfeat, rowmap, C, g, (float*)gdense);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
