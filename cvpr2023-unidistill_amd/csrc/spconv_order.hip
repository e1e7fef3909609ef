// Row order of a sparse-convolution rulebook for the output-stationary kernels of spconv_conv.hip (the scheduling side of
// spconv's implicit GEMM behind SubMConv3d / SparseConv3d, unidistill/layers/blocks_3d/det3d/spconv_backbone.py:10-113):
// rows sorted by their neighbour bit mask, so that a 128-row tile activates few of the K kernel offsets.
//
//   k_offset_counts offset frequencies from ~4 096 sampled rows (LDS histograms, K integer atomics per workgroup)
//   k_row_masks     bit position of offset k = its rank by descending frequency (ties: lower k first) -- the rarest offsets
//                   (the corners of a 3x3x3 kernel) get the top bits, so rows group by their rare neighbours first (20.2
//                   instead of 20.6 active offsets per tile at the 128-channel level of the LiDAR encoder);
//                   mask[r] = OR over k of (nbr[r][k] >= 0) << bitpos[k]
//   stable LSD radix sort of (mask, row) over the K mask bits (equal masks keep their row order) -> order[]: ceil(K / 9) passes of
//                   k_rs_hist (per-chunk digit counts) -> k_rs_rowscan (a wave per digit: exclusive scan over the chunks + the
//                   digit's total) -> k_rs_scatter (scans the <= 512 digit totals itself; a wave owns 512 consecutive keys: rank
//                   inside the wave by ballots, running per-digit counters in LDS) -- integer counting only, no library primitive (rounds 1-5 called rocprim::radix_sort_pairs here:
//                   the last library kernel on the product path)
// One call from the host side instead of ~16 tensor-library launches per rulebook (nine rulebooks per encoder pass).
#include "ud_common.h"
#include "ud_prof.h"
#include <algorithm>
#include <cstring>

namespace {

// offset frequencies over every step-th row: LDS histogram per workgroup, K device-scope atomics per workgroup
__global__ __launch_bounds__(256) void k_offset_counts(const int32_t* __restrict__ nbr, int M, int K, int step,
                                                       unsigned* __restrict__ cnt) {
  __shared__ unsigned s_cnt[32];
  if (threadIdx.x < 32) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const long long total = (long long)((M + step - 1) / step) * K;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int r = (int)(e / K) * step, k = (int)(e % K);
    if (nbr[(size_t)r * K + k] >= 0) atomicAdd(&s_cnt[k], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < K && s_cnt[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], s_cnt[threadIdx.x]);
}

// bit position of offset k = its rank by descending frequency (ties: lower k first), recomputed by every workgroup from the K
// counters; mask[r] = OR over k of (nbr[r][k] >= 0) << bitpos[k]
__global__ __launch_bounds__(256) void k_row_masks(const int32_t* __restrict__ nbr, int M, int K,
                                                   const unsigned* __restrict__ cnt, unsigned* __restrict__ mask) {
  __shared__ unsigned s_c[32];
  __shared__ int s_pos[32];
  if ((int)threadIdx.x < K) s_c[threadIdx.x] = cnt[threadIdx.x];
  __syncthreads();
  if ((int)threadIdx.x < K) {
    const int k = threadIdx.x;
    const unsigned mine = s_c[k];
    int rank = 0;
    for (int j = 0; j < K; ++j) rank += (s_c[j] > mine) || (s_c[j] == mine && j < k);
    s_pos[k] = rank;
  }
  __syncthreads();
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= M) return;
  const int32_t* row = nbr + (size_t)r * K;
  unsigned m = 0u;
  for (int k = 0; k < K; ++k) m |= (unsigned)(row[k] >= 0) << s_pos[k];
  mask[r] = m;
}

// ---- stable LSD radix sort of (key, row) pairs: digits of at most 9 bits, chunks of 2 048 keys ------------------------------
constexpr int kRsChunk = 2048, kRsBinsMax = 512, kRsWave = kRsChunk / 4;      // a wave owns 512 consecutive keys of its chunk

// hist[digit * nblk + chunk] = number of keys of the chunk with that digit
__global__ __launch_bounds__(256) void k_rs_hist(const unsigned* __restrict__ keys, int M, int shift, int bins, int nblk,
                                                 unsigned* __restrict__ hist) {
  __shared__ unsigned s_h[kRsBinsMax];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < bins; i += 256) s_h[i] = 0u;
  __syncthreads();
  const unsigned dm = (unsigned)bins - 1u;
#pragma unroll
  for (int r = 0; r < kRsChunk / 256; ++r) {
    const int i = b * kRsChunk + r * 256 + tid;
    if (i < M) atomicAdd(&s_h[(keys[i] >> shift) & dm], 1u);
  }
  __syncthreads();
  for (int i = tid; i < bins; i += 256) hist[(size_t)i * nblk + b] = s_h[i];
}

// hist[digit][0 .. nblk) -> its exclusive scan in place + dtot[digit] = the row's total: ONE WAVE per digit row, 64 chunks per trip.
// (First version: one workgroup scanning all bins * nblk counters, 97 uncoalesced counters per thread: 108 us per pass at 394 k
// rows -- 2.4 ms per encoder pass, more than the library sort it replaced.)
__global__ __launch_bounds__(256) void k_rs_rowscan(unsigned* __restrict__ hist, int bins, int nblk, unsigned* __restrict__ dtot) {
  const int lane = threadIdx.x & 63, d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= bins) return;
  unsigned* row = hist + (size_t)d * nblk;
  unsigned carry = 0u;
  for (int b0 = 0; b0 < nblk; b0 += 64) {
    const int b = b0 + lane;
    const unsigned v = b < nblk ? row[b] : 0u;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned a = __shfl_up(inc, o);
      if (lane >= o) inc += a;
    }
    if (b < nblk) row[b] = carry + inc - v;
    carry += __shfl(inc, 63);
  }
  if (lane == 0) dtot[d] = carry;
}

// keys_out / vals_out[destination] = the pair, destinations in (digit, chunk, position inside the chunk) order: stable.
// vals_in == nullptr: the value of key i is i (first pass); keys_out == nullptr: only the values are needed (last pass).
__global__ __launch_bounds__(256) void k_rs_scatter(const unsigned* __restrict__ keys_in, const int32_t* __restrict__ vals_in, int M,
                                                    int shift, int bins, int nblk, const unsigned* __restrict__ hist,
                                                    const unsigned* __restrict__ dtot, unsigned* __restrict__ keys_out,
                                                    int32_t* __restrict__ vals_out) {
  __shared__ unsigned s_run[4][kRsBinsMax];          // per wave: its keys per digit, then the running destination per digit
  __shared__ unsigned s_dpre[kRsBinsMax];            // where digit d's group starts = exclusive scan of the digit totals
  __shared__ unsigned s_wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const unsigned dm = (unsigned)bins - 1u;
  for (int i = tid; i < 4 * bins; i += 256) s_run[i / bins][i % bins] = 0u;
  {   // exclusive scan of the <= 512 digit totals: two consecutive digits per thread
    const int d0 = 2 * tid;
    const unsigned t0 = d0 < bins ? dtot[d0] : 0u, t1 = d0 + 1 < bins ? dtot[d0 + 1] : 0u;
    unsigned inc = t0 + t1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned a = __shfl_up(inc, o);
      if (lane >= o) inc += a;
    }
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    unsigned off = inc - (t0 + t1);
    for (int w = 0; w < wv; ++w) off += s_wsum[w];
    if (d0 < bins) s_dpre[d0] = off;
    if (d0 + 1 < bins) s_dpre[d0 + 1] = off + t0;
  }
  __syncthreads();
  const int w0 = b * kRsChunk + wv * kRsWave;
  unsigned key[kRsWave / 64];
#pragma unroll
  for (int r = 0; r < kRsWave / 64; ++r) {
    const int i = w0 + r * 64 + lane;
    key[r] = i < M ? keys_in[i] : 0u;
    if (i < M) atomicAdd(&s_run[wv][(key[r] >> shift) & dm], 1u);
  }
  __syncthreads();
  for (int d = tid; d < bins; d += 256) {            // wave bases of digit d: the chunk's group start + the earlier waves' keys
    unsigned base = s_dpre[d] + hist[(size_t)d * nblk + b];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned c = s_run[w][d];
      s_run[w][d] = base;
      base += c;
    }
  }
  __syncthreads();
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < kRsWave / 64; ++r) {
    const int i = w0 + r * 64 + lane;
    const bool ok = i < M;
    const unsigned d = (key[r] >> shift) & dm;
    unsigned long long peers = __ballot(ok);          // lanes of this round with my digit
    for (int bit = 0; (1 << bit) < bins; ++bit) {
      const unsigned long long bal = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? bal : ~bal;
    }
    unsigned pos = 0u;
    if (ok) pos = s_run[wv][d] + (unsigned)__popcll(peers & below);
    __builtin_amdgcn_wave_barrier();                  // every lane has read the counter before the group's first lane moves it
    if (ok && (peers & below) == 0ull) s_run[wv][d] += (unsigned)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the update has landed before the next round reads it
    if (ok) {
      if (keys_out) keys_out[pos] = key[r];
      vals_out[pos] = vals_in ? vals_in[i] : i;
    }
  }
}

struct OrderWs {
  unsigned* cnt;
  unsigned* mask;
  unsigned* keys[2];
  int32_t* vals[2];
  unsigned* hist;
  unsigned* dtot;
  int nblk;
  size_t total_bytes;
};

OrderWs carve_order(void* ws, int M, int K) {
  (void)K;
  UdArena a(ws, (size_t)-1);
  OrderWs w;
  w.nblk = ud_div_up(M, kRsChunk);
  w.cnt = a.take<unsigned>(32);
  w.mask = a.take<unsigned>(M);
  w.keys[0] = a.take<unsigned>(M);
  w.keys[1] = a.take<unsigned>(M);
  w.vals[0] = a.take<int32_t>(M);
  w.vals[1] = a.take<int32_t>(M);
  w.hist = a.take<unsigned>((size_t)kRsBinsMax * w.nblk);
  w.dtot = a.take<unsigned>(kRsBinsMax);
  w.total_bytes = a.used;
  return w;
}

// order[] = row indices sorted (stably) by the low `bits` bits of key[]
int radix_order(const OrderWs& w, int M, int bits, int32_t* order, hipStream_t stream) {
  const int passes = std::max(1, ud_div_up(bits, 9)), db = ud_div_up(std::max(bits, 1), passes);
  const unsigned* kin = w.mask;
  const int32_t* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const int shift = p * db, bins = 1 << std::min(db, bits - shift > 0 ? bits - shift : 1);
    const bool last = p + 1 == passes;
    unsigned* kout = last ? nullptr : w.keys[p & 1];
    int32_t* vout = last ? order : w.vals[p & 1];
    k_rs_hist<<<w.nblk, 256, 0, stream>>>(kin, M, shift, bins, w.nblk, w.hist);
    UD_LAUNCH_CHECK();
    k_rs_rowscan<<<ud_div_up(bins, 4), 256, 0, stream>>>(w.hist, bins, w.nblk, w.dtot);
    UD_LAUNCH_CHECK();
    k_rs_scatter<<<w.nblk, 256, 0, stream>>>(kin, vin, M, shift, bins, w.nblk, w.hist, w.dtot, kout, vout);
    UD_LAUNCH_CHECK();
    kin = kout, vin = vout;
  }
  return UD_OK;
}

}  // namespace

extern "C" size_t ud_spconv_mask_order_workspace_bytes(int M, int K) {
  if (M <= 0 || K <= 0 || K > 31) return 0;
  return carve_order(nullptr, M, K).total_bytes;
}

extern "C" int ud_spconv_mask_order(const int32_t* nbr, int M, int K, int32_t* order, void* workspace, size_t workspace_bytes,
                                    ud_stream_t stream_) {
  if (!nbr || !order || M <= 0 || K <= 0 || K > 31) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < ud_spconv_mask_order_workspace_bytes(M, K)) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  OrderWs w = carve_order(workspace, M, K);
  UdProfScope prof("spconv.mask_order", stream);
  const int step = M / 4096 > 1 ? M / 4096 : 1;
  UD_HIP_TRY(hipMemsetAsync(w.cnt, 0, 32 * sizeof(unsigned), stream));
  const long long sampled = (long long)((M + step - 1) / step) * K;
  k_offset_counts<<<(unsigned)std::min<long long>(64, (sampled + 2047) / 2048), 256, 0, stream>>>(nbr, M, K, step, w.cnt);
  UD_LAUNCH_CHECK();
  k_row_masks<<<ud_div_up(M, 256), 256, 0, stream>>>(nbr, M, K, w.cnt, w.mask);
  UD_LAUNCH_CHECK();
  return radix_order(w, M, K, order, stream);
}
