// Row order of a sparse-convolution rulebook for the output-stationary kernels of spconv_conv.hip (the scheduling side of
// spconv's implicit GEMM behind SubMConv3d / SparseConv3d, unidistill/layers/blocks_3d/det3d/spconv_backbone.py:10-113):
// rows sorted by their neighbour bit mask, so that a 128-row tile activates few of the K kernel offsets.
//
//   k_offset_counts offset frequencies from ~4 096 sampled rows (LDS histograms, K integer atomics per workgroup)
//   k_row_masks     bit position of offset k = its rank by descending frequency (ties: lower k first) -- the rarest offsets
//                   (the corners of a 3x3x3 kernel) get the top bits, so rows group by their rare neighbours first (20.2
//                   instead of 20.6 active offsets per tile at the 128-channel level of the LiDAR encoder);
//                   mask[r] = OR over k of (nbr[r][k] >= 0) << bitpos[k];  iota[r] = r
//   rocprim::radix_sort_pairs over the K mask bits (stable: equal masks keep their row order) -> order[]
// One call from the host side instead of ~16 tensor-library launches per rulebook (nine rulebooks per encoder pass).
#include "ud_common.h"
#include "ud_prof.h"
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

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
// counters; mask[r] = OR over k of (nbr[r][k] >= 0) << bitpos[k];  iota[r] = r
__global__ __launch_bounds__(256) void k_row_masks(const int32_t* __restrict__ nbr, int M, int K,
                                                   const unsigned* __restrict__ cnt, unsigned* __restrict__ mask,
                                                   int32_t* __restrict__ iota) {
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
  iota[r] = r;
}

struct OrderWs {
  unsigned* cnt;
  unsigned* mask;
  unsigned* mask_sorted;
  int32_t* iota;
  void* sort_tmp;
  size_t sort_bytes, total_bytes;
};

OrderWs carve_order(void* ws, int M, int K) {
  UdArena a(ws, (size_t)-1);
  OrderWs w;
  w.cnt = a.take<unsigned>(32);
  w.mask = a.take<unsigned>(M);
  w.mask_sorted = a.take<unsigned>(M);
  w.iota = a.take<int32_t>(M);
  w.sort_bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, w.sort_bytes, (const unsigned*)nullptr, (unsigned*)nullptr,
                                  (const int32_t*)nullptr, (int32_t*)nullptr, (size_t)M, 0u, (unsigned)K, (hipStream_t)0);
  w.sort_tmp = a.take<char>(w.sort_bytes);
  w.total_bytes = a.used;
  return w;
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
  k_row_masks<<<ud_div_up(M, 256), 256, 0, stream>>>(nbr, M, K, w.cnt, w.mask, w.iota);
  UD_LAUNCH_CHECK();
  size_t bytes = w.sort_bytes;
  UD_HIP_TRY(rocprim::radix_sort_pairs(w.sort_tmp, bytes, (const unsigned*)w.mask, w.mask_sorted, (const int32_t*)w.iota,
                                       order, (size_t)M, 0u, (unsigned)K, stream));
  return UD_OK;
}
