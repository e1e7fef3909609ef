// BatchNorm2d (+ residual) (+ ReLU) on channels-last bf16 activations as streaming kernels.
//
// The reference's dense layers are Conv2d -> BatchNorm2d -> ReLU chains (BaseBEVBackbone,
// base_bev_backbone.py:48-66; ResNet bottlenecks via mmdet; CenterHead.shared_conv, center_head.py:
// 408-420).  Run as separate library ops each chain link is its own read+write pass over the
// activation (BN statistics, BN normalise, ReLU, residual add; and three more in backward).  These
// kernels are the HBM-bound formulation: statistics are one read (ud_head_tail_stats, any
// [pixels][C % 64 == 0] bf16 tensor), normalise + residual + ReLU one read/write, and the backward is
// one reduction pass + one pass that writes dx -- the ReLU mask is recomputed from x, never stored.
//   y  = act(x * scale + shift (+ residual)),  scale = gamma * invstd, shift = beta - mean * scale
//   dr = dy * [y > 0];  dbeta = sum dr;  dgamma = invstd * sum dr (x - mean)
//   dx = scale * dr + k2 * x + k0,  k2 = -scale * dgamma / P * invstd,  k0 = -scale * dbeta / P - k2 * mean
#include "ud_common.h"
#include "ud_prof.h"

namespace {

constexpr int kMaxSlices = 1024;
// pixel slices per 64-channel group: enough workgroups (~2048) to fill 256 CUs whatever C is
int group_width(int C) { return C % 64 == 0 ? 64 : (C % 32 == 0 ? 32 : 16); }
int slices_for(long long P, int C) {
  const int gw = group_width(C), groups = C / gw;
  long long s = 2048 / (groups > 0 ? groups : 1);
  const long long cap = (P + 256 / (gw / 8) - 1) / (256 / (gw / 8));
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > kMaxSlices) s = kMaxSlices;
  return (int)s;
}

__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

__device__ __forceinline__ void unpack8(const uint4& r, float* v) {
  v[0] = bf_lo(r.x); v[1] = bf_hi(r.x); v[2] = bf_lo(r.y); v[3] = bf_hi(r.y);
  v[4] = bf_lo(r.z); v[5] = bf_hi(r.z); v[6] = bf_lo(r.w); v[7] = bf_hi(r.w);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  return make_uint4(ud_pack_bf16x2(v[0], v[1]), ud_pack_bf16x2(v[2], v[3]),
                    ud_pack_bf16x2(v[4], v[5]), ud_pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void load8(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Thread = (8-channel chunk, pixel lane): the chunk's scale / shift live in registers for the whole
// pixel loop (a flat unit loop re-loaded 4 constant vectors and took a 64-bit modulo per 16 bytes).
// grid (pixel slices, channel blocks); CH = chunks per workgroup (power of two <= 256).
__global__ __launch_bounds__(256) void k_bn_act_fwd(const unsigned short* __restrict__ x,
                                                    const unsigned short* __restrict__ res,
                                                    const float* __restrict__ scale,
                                                    const float* __restrict__ shift,
                                                    unsigned short* __restrict__ y, long long P, int C,
                                                    int CH, int relu) {
  const int chunk = blockIdx.y * CH + threadIdx.x % CH, lanes = 256 / CH, pl = threadIdx.x / CH;
  if (chunk * 8 >= C) return;
  float s[8], t[8];
  load8(scale + chunk * 8, s);
  load8(shift + chunk * 8, t);
  for (long long p = (long long)blockIdx.x * lanes + pl; p < P; p += (long long)gridDim.x * lanes) {
    const size_t off = (size_t)p * C + chunk * 8;
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(x + off), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], s[e], t[e]);
    if (res) {
      float r[8];
      unpack8(*reinterpret_cast<const uint4*>(res + off), r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<uint4*>(y + off) = pack8(v);
  }
}

// mask source: y (the saved output) when a residual took part, else recomputed from x.
__device__ __forceinline__ void masked_grad(const float* xv, const uint4* yraw, const float* dyv,
                                            const float* s, const float* t, int relu, float* dr) {
  if (!relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dr[e] = dyv[e];
    return;
  }
  if (yraw) {
    float yv[8];
    unpack8(*yraw, yv);
#pragma unroll
    for (int e = 0; e < 8; ++e) dr[e] = yv[e] > 0.f ? dyv[e] : 0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) dr[e] = fmaf(xv[e], s[e], t[e]) > 0.f ? dyv[e] : 0.f;
  }
}

// One workgroup = (pixel slice, group of GW channels); thread = (8-channel chunk, pixel lane).
template <int GW>
struct GroupMap {
  static constexpr int kChunks = GW / 8, kLanes = 256 / kChunks;
};

// grid (slices, C/GW): partial[slice][C][2] = sum (x - pivot), sum (x - pivot)^2, pivot = x[row 0][c]
template <int GW>
__global__ __launch_bounds__(256) void k_bn_stats_partial(const unsigned short* __restrict__ x, long long P,
                                                          int C, float* __restrict__ partial) {
  using M = GroupMap<GW>;
  __shared__ float red[M::kLanes][GW + 1][2];
  const int cg = blockIdx.y, tid = threadIdx.x, chunk = tid % M::kChunks, pl = tid / M::kChunks;
  const int c0 = cg * GW + chunk * 8;
  float piv[8], s1[8], s2[8];
  unpack8(*reinterpret_cast<const uint4*>(x + c0), piv);
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  for (long long p = (long long)blockIdx.x * M::kLanes + pl; p < P; p += (long long)gridDim.x * M::kLanes) {
    float v[8];
    unpack8(*reinterpret_cast<const uint4*>(x + (size_t)p * C + c0), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[e] - piv[e];
      s1[e] += d;
      s2[e] += d * d;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[pl][chunk * 8 + e][0] = s1[e]; red[pl][chunk * 8 + e][1] = s2[e]; }
  __syncthreads();
  if (tid < 2 * GW) {
    const int c = tid >> 1, w = tid & 1;
    float a = 0.f;
    for (int i = 0; i < M::kLanes; ++i) a += red[i][c][w];
    partial[((size_t)blockIdx.x * C + cg * GW + c) * 2 + w] = a;
  }
}

// one wave per channel: mean / biased variance / invstd, folded scale+shift, running statistics
__global__ void k_bn_stats_final(const unsigned short* __restrict__ x, const float* __restrict__ partial,
                                 int slices, long long P, int C, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, float* __restrict__ mean,
                                 float* __restrict__ var, float* __restrict__ invstd,
                                 float* __restrict__ scale, float* __restrict__ shift,
                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                 float momentum, long long* __restrict__ batches_tracked) {
  const int c = blockIdx.x, lane = threadIdx.x;
  if (batches_tracked && c == 0 && lane == 0) *batches_tracked += 1;   // nn.BatchNorm's step counter
  double a = 0.0, q = 0.0;
  for (int s = lane; s < slices; s += 64) {
    a += partial[((size_t)s * C + c) * 2];
    q += partial[((size_t)s * C + c) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    q += __shfl_xor(q, o);
  }
  if (lane != 0) return;
  const double piv = __uint_as_float((unsigned)x[c] << 16);
  const double m = a / (double)P;
  double v = q / (double)P - m * m;
  if (v < 0.0) v = 0.0;
  const float mu = (float)(piv + m), is = (float)(1.0 / sqrt(v + (double)eps));
  mean[c] = mu;
  var[c] = (float)v;
  invstd[c] = is;
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - mu * sc;
  if (running_mean) {
    const double unbiased = v * ((double)P / (double)(P > 1 ? P - 1 : 1));
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// grid (slices, C/GW): partial[slice][C][2] = sum dr, sum dr * (x - mean)
template <int GW>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const unsigned short* __restrict__ x,
                                                       const unsigned short* __restrict__ y,
                                                       const unsigned short* __restrict__ dy,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ mean, long long P,
                                                       int C, int relu, float* __restrict__ partial) {
  using M = GroupMap<GW>;
  __shared__ float red[M::kLanes][GW + 1][2];
  const int cg = blockIdx.y, tid = threadIdx.x, chunk = tid % M::kChunks, pl = tid / M::kChunks;
  const int c0 = cg * GW + chunk * 8;
  float s[8], t[8], mu[8], s1[8], s2[8];
  load8(scale + c0, s);
  load8(shift + c0, t);
  load8(mean + c0, mu);
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  for (long long p = (long long)blockIdx.x * M::kLanes + pl; p < P; p += (long long)gridDim.x * M::kLanes) {
    const size_t off = (size_t)p * C + c0;
    float xv[8], dyv[8], dr[8];
    unpack8(*reinterpret_cast<const uint4*>(x + off), xv);
    unpack8(*reinterpret_cast<const uint4*>(dy + off), dyv);
    uint4 yr;
    if (y) yr = *reinterpret_cast<const uint4*>(y + off);
    masked_grad(xv, y ? &yr : nullptr, dyv, s, t, relu, dr);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += dr[e];
      s2[e] += dr[e] * (xv[e] - mu[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[pl][chunk * 8 + e][0] = s1[e]; red[pl][chunk * 8 + e][1] = s2[e]; }
  __syncthreads();
  if (tid < 2 * GW) {
    const int c = tid >> 1, w = tid & 1;
    float a = 0.f;
    for (int i = 0; i < M::kLanes; ++i) a += red[i][c][w];
    partial[((size_t)blockIdx.x * C + cg * GW + c) * 2 + w] = a;
  }
}

__global__ void k_bn_bwd_final(const float* __restrict__ partial, int slices, int C, long long P,
                               const float* __restrict__ scale, const float* __restrict__ mean,
                               const float* __restrict__ invstd, float* __restrict__ dgamma,
                               float* __restrict__ dbeta, float* __restrict__ k0, float* __restrict__ k2) {
  const int c = blockIdx.x, lane = threadIdx.x;      // one wave per channel, fixed-order reduction
  float a = 0.f, q = 0.f;
  for (int s = lane; s < slices; s += 64) {
    a += partial[((size_t)s * C + c) * 2];
    q += partial[((size_t)s * C + c) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    q += __shfl_xor(q, o);
  }
  if (lane != 0) return;
  const float is = invstd[c], dg = q * is, inv_p = 1.0f / (float)P;
  dbeta[c] = a;
  dgamma[c] = dg;
  const float kk2 = -scale[c] * (dg * inv_p) * is;
  k2[c] = kk2;
  k0[c] = -scale[c] * (a * inv_p) - kk2 * mean[c];
}

__global__ __launch_bounds__(256) void k_bn_bwd_dx(const unsigned short* __restrict__ x,
                                                   const unsigned short* __restrict__ y,
                                                   const unsigned short* __restrict__ dy,
                                                   const float* __restrict__ scale,
                                                   const float* __restrict__ shift,
                                                   const float* __restrict__ k0,
                                                   const float* __restrict__ k2,
                                                   unsigned short* __restrict__ dx,
                                                   unsigned short* __restrict__ dres, long long P, int C,
                                                   int CH, int relu) {
  const int chunk = blockIdx.y * CH + threadIdx.x % CH, lanes = 256 / CH, pl = threadIdx.x / CH;
  if (chunk * 8 >= C) return;
  float s[8], t[8], a0[8], a2[8];
  load8(scale + chunk * 8, s);
  load8(shift + chunk * 8, t);
  load8(k0 + chunk * 8, a0);
  load8(k2 + chunk * 8, a2);
  for (long long p = (long long)blockIdx.x * lanes + pl; p < P; p += (long long)gridDim.x * lanes) {
    const size_t off = (size_t)p * C + chunk * 8;
    float xv[8], dyv[8], dr[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(x + off), xv);
    unpack8(*reinterpret_cast<const uint4*>(dy + off), dyv);
    uint4 yr;
    if (y) yr = *reinterpret_cast<const uint4*>(y + off);
    masked_grad(xv, y ? &yr : nullptr, dyv, s, t, relu, dr);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(s[e], dr[e], fmaf(a2[e], xv[e], a0[e]));
    *reinterpret_cast<uint4*>(dx + off) = pack8(o);
    if (dres) *reinterpret_cast<uint4*>(dres + off) = pack8(dr);
  }
}

struct BnWs { float *partial, *k0, *k2; };
size_t carve(UdArena& ar, int C, BnWs* w) {
  w->partial = ar.take<float>((size_t)kMaxSlices * C * 2);
  w->k0 = ar.take<float>(C);
  w->k2 = ar.take<float>(C);
  return ar.used;
}
// grid for the streaming kernels: CH channel chunks (of 8) per workgroup, enough pixel slices for ~4096 groups
void stream_grid(long long P, int C, int* CH, dim3* grid) {
  const int chunks = C / 8;
  int ch = 1;
  while (ch < chunks && ch < 256) ch <<= 1;
  const int lanes = 256 / ch, cblocks = (chunks + ch - 1) / ch;
  long long slices = (P + lanes - 1) / lanes;
  const long long want = 4096 / cblocks > 0 ? 4096 / cblocks : 1;
  if (slices > want) slices = want;
  if (slices < 1) slices = 1;
  *CH = ch;
  *grid = dim3((unsigned)slices, (unsigned)cblocks);
}

}  // namespace

extern "C" {

size_t ud_bn_act_workspace_bytes(int C) {
  if (C <= 0) return 0;
  UdArena ar(nullptr, 0);
  BnWs w;
  return carve(ar, C, &w);
}

int ud_bn_stats(const void* x, long long P, int C, const float* gamma, const float* beta, float eps,
                float* mean, float* var, float* invstd, float* scale, float* shift, float* running_mean,
                float* running_var, float momentum, long long* batches_tracked, void* workspace,
                size_t workspace_bytes, ud_stream_t stream_) {
  if (!x || !gamma || !beta || !mean || !var || !invstd || !scale || !shift || P <= 0 || C <= 0 ||
      ((running_mean == nullptr) != (running_var == nullptr)))
    return UD_ERR_INVALID_ARG;
  if (C % 16) return UD_ERR_UNSUPPORTED;
  UdArena ar(workspace, workspace_bytes);
  BnWs w;
  carve(ar, C, &w);
  if (!ar.ok()) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  UdProfScope prof("bn_act.stats", stream);
  const int slices = slices_for(P, C), gw = group_width(C);
  if (gw == 64)
    k_bn_stats_partial<64><<<dim3(slices, C / 64), 256, 0, stream>>>((const unsigned short*)x, P, C, w.partial);
  else if (gw == 32)
    k_bn_stats_partial<32><<<dim3(slices, C / 32), 256, 0, stream>>>((const unsigned short*)x, P, C, w.partial);
  else
    k_bn_stats_partial<16><<<dim3(slices, C / 16), 256, 0, stream>>>((const unsigned short*)x, P, C, w.partial);
  UD_LAUNCH_CHECK();
  k_bn_stats_final<<<C, 64, 0, stream>>>((const unsigned short*)x, w.partial, slices, P, C, gamma, beta, eps,
                                         mean, var, invstd, scale, shift, running_mean, running_var, momentum,
                                         batches_tracked);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

int ud_bn_act_fwd(const void* x, const void* residual, const float* scale, const float* shift, void* y,
                  long long P, int C, int relu, ud_stream_t stream_) {
  if (!x || !scale || !shift || !y || P <= 0 || C <= 0) return UD_ERR_INVALID_ARG;
  if (C % 8) return UD_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  int CH;
  dim3 grid;
  stream_grid(P, C, &CH, &grid);
  UdProfScope prof("bn_act.k_fwd", stream);
  k_bn_act_fwd<<<grid, 256, 0, stream>>>((const unsigned short*)x, (const unsigned short*)residual, scale, shift,
                                         (unsigned short*)y, P, C, CH, relu);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

int ud_bn_act_bwd(const void* x, const void* y, const void* dy, const float* scale, const float* shift,
                  const float* mean, const float* invstd, void* dx, void* dresidual, float* dgamma,
                  float* dbeta, long long P, int C, int relu, void* workspace, size_t workspace_bytes,
                  ud_stream_t stream_) {
  if (!x || !dy || !scale || !shift || !mean || !invstd || !dx || !dgamma || !dbeta || P <= 0 || C <= 0)
    return UD_ERR_INVALID_ARG;
  if (C % 16) return UD_ERR_UNSUPPORTED;
  UdArena ar(workspace, workspace_bytes);
  BnWs w;
  carve(ar, C, &w);
  if (!ar.ok()) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  {
    UdProfScope prof("bn_act.k_bwd_reduce", stream);
    const int slices = slices_for(P, C), gw = group_width(C);
#define UD_BN_REDUCE(GW)                                                                                   \
  k_bn_bwd_reduce<GW><<<dim3(slices, C / GW), 256, 0, stream>>>(                                           \
      (const unsigned short*)x, (const unsigned short*)y, (const unsigned short*)dy, scale, shift, mean, P, \
      C, relu, w.partial)
    if (gw == 64) UD_BN_REDUCE(64); else if (gw == 32) UD_BN_REDUCE(32); else UD_BN_REDUCE(16);
#undef UD_BN_REDUCE
    UD_LAUNCH_CHECK();
    k_bn_bwd_final<<<C, 64, 0, stream>>>(w.partial, slices, C, P, scale, mean, invstd,
                                                           dgamma, dbeta, w.k0, w.k2);
    UD_LAUNCH_CHECK();
  }
  int CH;
  dim3 grid;
  stream_grid(P, C, &CH, &grid);
  UdProfScope prof("bn_act.k_bwd_dx", stream);
  k_bn_bwd_dx<<<grid, 256, 0, stream>>>((const unsigned short*)x, (const unsigned short*)y,
                                        (const unsigned short*)dy, scale, shift, w.k0, w.k2, (unsigned short*)dx,
                                        (unsigned short*)dresidual, P, C, CH, relu);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

}  // extern "C"
