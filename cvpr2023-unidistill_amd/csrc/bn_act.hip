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
// element access: 8 consecutive channels of a bf16 (16 bytes) or fp32 (2 x 16 bytes) activation row
__device__ __forceinline__ void ld8(const unsigned short* p, float* v) { unpack8(*reinterpret_cast<const uint4*>(p), v); }
__device__ __forceinline__ void st8(unsigned short* p, const float* v) { *reinterpret_cast<uint4*>(p) = pack8(v); }
__device__ __forceinline__ void ld8(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ float elem_f(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
__device__ __forceinline__ float elem_f(float u) { return u; }

__device__ __forceinline__ void load8(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// Thread = (8-channel chunk, pixel lane): the chunk's scale / shift live in registers for the whole
// pixel loop (a flat unit loop re-loaded 4 constant vectors and took a 64-bit modulo per 16 bytes).
// grid (pixel slices, channel blocks); CH = chunks per workgroup (power of two <= 256).
template <typename T>
__global__ __launch_bounds__(256) void k_bn_act_fwd(const T* __restrict__ x, const T* __restrict__ res,
                                                    const float* __restrict__ scale,
                                                    const float* __restrict__ shift, T* __restrict__ y,
                                                    long long P, int C, int CH, int relu, long long y_ld) {
  const int chunk = blockIdx.y * CH + threadIdx.x % CH, lanes = 256 / CH, pl = threadIdx.x / CH;
  if (chunk * 8 >= C) return;
  float s[8], t[8];
  load8(scale + chunk * 8, s);
  load8(shift + chunk * 8, t);
  for (long long p = (long long)blockIdx.x * lanes + pl; p < P; p += (long long)gridDim.x * lanes) {
    const size_t off = (size_t)p * C + chunk * 8;
    float v[8];
    ld8(x + off, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], s[e], t[e]);
    if (res) {
      float r[8];
      ld8(res + off, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    st8(y + (size_t)p * y_ld + chunk * 8, v);       // y_ld > C: the rows of a channel slice of a wider map (a fused concatenation)
  }
}

// mask source: y (the saved output) when a residual took part, else recomputed from x.
// has_y is a flag, not a null yv: `y ? yv : nullptr` at the call site is a select of two addresses, which forces the yv
// register array into scratch memory (48 bytes per lane written and re-read per pixel in two HBM-bound kernels)
__device__ __forceinline__ void masked_grad(const float* xv, const float* yv, bool has_y, const float* dyv,
                                            const float* s, const float* t, int relu, float* dr) {
  if (!relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dr[e] = dyv[e];
    return;
  }
  if (has_y) {
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
template <int GW, typename T>
__global__ __launch_bounds__(256) void k_bn_stats_partial(const T* __restrict__ x, long long P,
                                                          int C, float* __restrict__ partial) {
  using M = GroupMap<GW>;
  __shared__ float red[M::kLanes][GW + 1][2];
  const int cg = blockIdx.y, tid = threadIdx.x, chunk = tid % M::kChunks, pl = tid / M::kChunks;
  const int c0 = cg * GW + chunk * 8;
  float piv[8], s1[8], s2[8];
  ld8(x + c0, piv);
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  for (long long p = (long long)blockIdx.x * M::kLanes + pl; p < P; p += (long long)gridDim.x * M::kLanes) {
    float v[8];
    ld8(x + (size_t)p * C + c0, v);
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

// 16 channels per workgroup, 16 slice lanes per channel (a slice row of 16 channels is one 128-byte read): mean / biased
// variance / invstd, folded scale+shift, running statistics.  Fixed summation order (deterministic); x == nullptr: the
// partials are unshifted (they come from a convolution epilogue), else shifted by the pivot x[c].
template <typename T>
__global__ __launch_bounds__(256) void k_bn_stats_final(const T* __restrict__ x, const float* __restrict__ partial,
                                 int slices, long long P, int C, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, float* __restrict__ mean,
                                 float* __restrict__ var, float* __restrict__ invstd,
                                 float* __restrict__ scale, float* __restrict__ shift,
                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                 float momentum, long long* __restrict__ batches_tracked) {
  __shared__ double red[16][16][2];
  const int tid = threadIdx.x, cl = tid & 15, sl = tid >> 4;
  const int c = blockIdx.x * 16 + cl;
  if (batches_tracked && blockIdx.x == 0 && tid == 0) *batches_tracked += 1;   // nn.BatchNorm's step counter
  // four independent loads per trip (fixed order): with one, the ~90 dependent trips of a 1 440-tile convolution made this
  // kernel 20 us of pure latency on C / 16 workgroups
  auto ld = [&](int s) { return *reinterpret_cast<const float2*>(partial + ((size_t)s * C + c) * 2); };
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  int s = sl;
  for (; s + 48 < slices; s += 64) {
    const float2 v0 = ld(s), v1 = ld(s + 16), v2 = ld(s + 32), v3 = ld(s + 48);
    a0 += v0.x; q0 += v0.y;
    a1 += v1.x; q1 += v1.y;
    a2 += v2.x; q2 += v2.y;
    a3 += v3.x; q3 += v3.y;
  }
  for (; s < slices; s += 16) {
    const float2 v = ld(s);
    a0 += v.x; q0 += v.y;
  }
  double a = (a0 + a1) + (a2 + a3), q = (q0 + q1) + (q2 + q3);
  red[sl][cl][0] = a;
  red[sl][cl][1] = q;
  __syncthreads();
  if (sl != 0) return;
  a = q = 0.0;
  for (int k = 0; k < 16; ++k) {
    a += red[k][cl][0];
    q += red[k][cl][1];
  }
  const double piv = x ? (double)elem_f(x[c]) : 0.0;
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
template <int GW, typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_reduce(const T* __restrict__ x, const T* __restrict__ y,
                                                       const T* __restrict__ dy,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const float* __restrict__ mean, long long P,
                                                       int C, int relu, float* __restrict__ partial, long long dy_ld) {
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
    float xv[8], dyv[8], dr[8], yv[8];
    ld8(x + off, xv);
    ld8(dy + (size_t)p * dy_ld + c0, dyv);
    if (y) ld8(y + off, yv);
    masked_grad(xv, yv, y != nullptr, dyv, s, t, relu, dr);
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

// One workgroup of T threads per channel; every thread adds rows t, t + T, ... with FOUR independent loads in flight, then a fixed
// tree.  (Rounds 2-5: one wave per channel walking up to 1 024 partial rows in 16 dependent trips -- 3 us on an idle chip, 26 us
// on average inside the step, where every trip waits behind the other streams' memory traffic: 71 of them sit on the main
// stream's backward chain between a layer's reduction pass and its dx pass.)
template <int T>
__global__ __launch_bounds__(T) void k_bn_bwd_final(const float* __restrict__ partial, int slices, int C, long long P,
                                                    const float* __restrict__ scale, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, float* __restrict__ k0, float* __restrict__ k2) {
  __shared__ float red[T / 64][2];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto ld = [&](int s) { return *reinterpret_cast<const float2*>(partial + ((size_t)s * C + c) * 2); };
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  int s = tid;
  for (; s + 3 * T < slices; s += 4 * T) {
    const float2 v0 = ld(s), v1 = ld(s + T), v2 = ld(s + 2 * T), v3 = ld(s + 3 * T);
    a0 += v0.x; q0 += v0.y;
    a1 += v1.x; q1 += v1.y;
    a2 += v2.x; q2 += v2.y;
    a3 += v3.x; q3 += v3.y;
  }
  for (; s < slices; s += T) {
    const float2 v = ld(s);
    a0 += v.x; q0 += v.y;
  }
  float a = (a0 + a1) + (a2 + a3), q = (q0 + q1) + (q2 + q3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    q += __shfl_xor(q, o);
  }
  if (T > 64) {
    if (lane == 0) { red[wv][0] = a; red[wv][1] = q; }
    __syncthreads();
    if (tid == 0) {
      a = red[0][0], q = red[0][1];
#pragma unroll
      for (int w = 1; w < T / 64; ++w) { a += red[w][0]; q += red[w][1]; }
    }
  }
  if (tid != 0) return;
  const float is = invstd[c], dg = q * is, inv_p = 1.0f / (float)P;
  dbeta[c] = a;
  dgamma[c] = dg;
  const float kk2 = -scale[c] * (dg * inv_p) * is;
  k2[c] = kk2;
  k0[c] = -scale[c] * (a * inv_p) - kk2 * mean[c];
}

template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_dx(const T* __restrict__ x, const T* __restrict__ y,
                                                   const T* __restrict__ dy,
                                                   const float* __restrict__ scale,
                                                   const float* __restrict__ shift,
                                                   const float* __restrict__ k0,
                                                   const float* __restrict__ k2, T* __restrict__ dx,
                                                   T* __restrict__ dres, long long P, int C,
                                                   int CH, int relu, long long dy_ld) {
  const int chunk = blockIdx.y * CH + threadIdx.x % CH, lanes = 256 / CH, pl = threadIdx.x / CH;
  if (chunk * 8 >= C) return;
  float s[8], t[8], a0[8], a2[8];
  load8(scale + chunk * 8, s);
  load8(shift + chunk * 8, t);
  load8(k0 + chunk * 8, a0);
  load8(k2 + chunk * 8, a2);
  for (long long p = (long long)blockIdx.x * lanes + pl; p < P; p += (long long)gridDim.x * lanes) {
    const size_t off = (size_t)p * C + chunk * 8;
    float xv[8], dyv[8], dr[8], o[8], yv[8];
    ld8(x + off, xv);
    ld8(dy + (size_t)p * dy_ld + chunk * 8, dyv);
    if (y) ld8(y + off, yv);
    masked_grad(xv, yv, y != nullptr, dyv, s, t, relu, dr);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(s[e], dr[e], fmaf(a2[e], xv[e], a0[e]));
    st8(dx + off, o);
    if (dres) st8(dres + off, dr);
  }
}


// ---- column sums: out[c] = sum_p x[p * ld + c] -- the bias gradient of a convolution (sum of dy over batch and pixels:
// center_head.py:64,339,353 bias=True; the depth net's 1x1) as two HBM-rate passes in a fixed order.  (ATen's reduce_kernel
// ran a channels-last [P][64..2688] sum on 11 workgroups: 0.65 TB/s.)
// vector pass: thread = (8-channel chunk, pixel lane), grid (pixel slices, channel blocks)
template <typename T>
__global__ __launch_bounds__(256) void k_colsum_partial(const T* __restrict__ x, long long P, int C, long long ld, int CH,
                                                        float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x, ck = tid % CH, lanes = 256 / CH, pl = tid / CH;
  const int chunk = blockIdx.y * CH + ck;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  if (chunk * 8 < C) {
    const T* xp = x + chunk * 8;
    long long p = (long long)blockIdx.x * lanes + pl;
    const long long step = (long long)gridDim.x * lanes;
    for (; p + step < P; p += 2 * step) {          // two independent rows in flight
      float v[8], u[8];
      ld8(xp + (size_t)p * ld, v);
      ld8(xp + (size_t)(p + step) * ld, u);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += u[e];
    }
    if (p < P) {
      float v[8];
      ld8(xp + (size_t)p * ld, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[pl * (CH * 8) + ck * 8 + e] = s[e];
  __syncthreads();
  for (int c = tid; c < CH * 8; c += 256) {
    const int cc = blockIdx.y * CH * 8 + c;
    if (cc >= C) break;
    float a = 0.f;
    for (int i = 0; i < lanes; ++i) a += red[i * (CH * 8) + c];
    partial[(size_t)blockIdx.x * C + cc] = a;
  }
}

// any C / ld / alignment: thread = (row of the pass, channel), 4-byte (2-byte) loads
template <typename T>
__global__ __launch_bounds__(256) void k_colsum_partial_scalar(const T* __restrict__ x, long long P, int C, long long ld,
                                                               float* __restrict__ partial) {
  __shared__ float red[256];
  const int tid = threadIdx.x, cb0 = blockIdx.y * 256, cw = min(256, C - cb0), R = 256 / cw;
  const int c = tid % cw, r = tid / cw;
  float a0 = 0.f, a1 = 0.f;
  if (r < R) {
    const T* xp = x + cb0 + c;
    long long p = (long long)blockIdx.x * R + r;
    const long long step = (long long)gridDim.x * R;
    for (; p + step < P; p += 2 * step) {
      a0 += elem_f(xp[(size_t)p * ld]);
      a1 += elem_f(xp[(size_t)(p + step) * ld]);
    }
    if (p < P) a0 += elem_f(xp[(size_t)p * ld]);
  }
  red[tid] = a0 + a1;
  __syncthreads();
  if (tid < cw) {
    float a = 0.f;
    for (int i = 0; i < R; ++i) a += red[i * cw + tid];
    partial[(size_t)blockIdx.x * C + cb0 + tid] = a;
  }
}

template <int T>
__global__ __launch_bounds__(T) void k_colsum_final(const float* __restrict__ partial, int slices, int C, float* __restrict__ out) {
  __shared__ float red[T / 64];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;      // T threads per channel, fixed-order reduction
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = tid;
  for (; s + 3 * T < slices; s += 4 * T) {
    const float v0 = partial[(size_t)s * C + c], v1 = partial[(size_t)(s + T) * C + c], v2 = partial[(size_t)(s + 2 * T) * C + c],
                v3 = partial[(size_t)(s + 3 * T) * C + c];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; s < slices; s += T) a0 += partial[(size_t)s * C + c];
  float a = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (T > 64) {
    if (lane == 0) red[wv] = a;
    __syncthreads();
    if (tid == 0) {
      a = red[0];
#pragma unroll
      for (int w = 1; w < T / 64; ++w) a += red[w];
    }
  }
  if (tid == 0) out[c] = a;
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

template <typename T>
int bn_stats_impl(const T* x, long long P, int C, const float* gamma, const float* beta, float eps,
                  float* mean, float* var, float* invstd, float* scale, float* shift, float* running_mean,
                  float* running_var, float momentum, long long* batches_tracked, void* workspace,
                  size_t workspace_bytes, hipStream_t stream) {
  if (!x || !gamma || !beta || !mean || !var || !invstd || !scale || !shift || P <= 0 || C <= 0 ||
      ((running_mean == nullptr) != (running_var == nullptr)))
    return UD_ERR_INVALID_ARG;
  if (C % 16) return UD_ERR_UNSUPPORTED;
  UdArena ar(workspace, workspace_bytes);
  BnWs w;
  carve(ar, C, &w);
  if (!ar.ok()) return UD_ERR_WORKSPACE;
  UdProfScope prof("bn_act.stats", stream);
  const int slices = slices_for(P, C), gw = group_width(C);
  if (gw == 64)
    k_bn_stats_partial<64, T><<<dim3(slices, C / 64), 256, 0, stream>>>(x, P, C, w.partial);
  else if (gw == 32)
    k_bn_stats_partial<32, T><<<dim3(slices, C / 32), 256, 0, stream>>>(x, P, C, w.partial);
  else
    k_bn_stats_partial<16, T><<<dim3(slices, C / 16), 256, 0, stream>>>(x, P, C, w.partial);
  UD_LAUNCH_CHECK();
  k_bn_stats_final<T><<<C / 16, 256, 0, stream>>>(x, w.partial, slices, P, C, gamma, beta, eps, mean, var, invstd, scale,
                                            shift, running_mean, running_var, momentum, batches_tracked);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

template <typename T>
int bn_fwd_impl(const T* x, const T* residual, const float* scale, const float* shift, T* y, long long P, int C,
                int relu, long long y_ld, hipStream_t stream) {
  if (!x || !scale || !shift || !y || P <= 0 || C <= 0 || y_ld < C) return UD_ERR_INVALID_ARG;
  if (C % 8 || y_ld % 8 || ((size_t)y & 15)) return UD_ERR_UNSUPPORTED;
  int CH;
  dim3 grid;
  stream_grid(P, C, &CH, &grid);
  UdProfScope prof("bn_act.k_fwd", stream);
  k_bn_act_fwd<T><<<grid, 256, 0, stream>>>(x, residual, scale, shift, y, P, C, CH, relu, y_ld);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

template <typename T>
int bn_bwd_impl(const T* x, const T* y, const T* dy, const float* scale, const float* shift, const float* mean,
                const float* invstd, T* dx, T* dresidual, float* dgamma, float* dbeta, long long P, int C, int relu,
                long long dy_ld, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!x || !dy || !scale || !shift || !mean || !invstd || !dx || !dgamma || !dbeta || P <= 0 || C <= 0 || dy_ld < C)
    return UD_ERR_INVALID_ARG;
  if (C % 16 || dy_ld % 8 || ((size_t)dy & 15)) return UD_ERR_UNSUPPORTED;
  UdArena ar(workspace, workspace_bytes);
  BnWs w;
  carve(ar, C, &w);
  if (!ar.ok()) return UD_ERR_WORKSPACE;
  {
    UdProfScope prof("bn_act.k_bwd_reduce", stream);
    const int slices = slices_for(P, C), gw = group_width(C);
    if (gw == 64)
      k_bn_bwd_reduce<64, T><<<dim3(slices, C / 64), 256, 0, stream>>>(x, y, dy, scale, shift, mean, P, C, relu, w.partial, dy_ld);
    else if (gw == 32)
      k_bn_bwd_reduce<32, T><<<dim3(slices, C / 32), 256, 0, stream>>>(x, y, dy, scale, shift, mean, P, C, relu, w.partial, dy_ld);
    else
      k_bn_bwd_reduce<16, T><<<dim3(slices, C / 16), 256, 0, stream>>>(x, y, dy, scale, shift, mean, P, C, relu, w.partial, dy_ld);
    UD_LAUNCH_CHECK();
    if (slices > 128)
      k_bn_bwd_final<256><<<C, 256, 0, stream>>>(w.partial, slices, C, P, scale, mean, invstd, dgamma, dbeta, w.k0, w.k2);
    else
      k_bn_bwd_final<64><<<C, 64, 0, stream>>>(w.partial, slices, C, P, scale, mean, invstd, dgamma, dbeta, w.k0, w.k2);
    UD_LAUNCH_CHECK();
  }
  int CH;
  dim3 grid;
  stream_grid(P, C, &CH, &grid);
  UdProfScope prof("bn_act.k_bwd_dx", stream);
  k_bn_bwd_dx<T><<<grid, 256, 0, stream>>>(x, y, dy, scale, shift, w.k0, w.k2, dx, dresidual, P, C, CH, relu, dy_ld);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

template <typename T>
int colsum_impl(const T* x, long long P, int C, long long ld, float* out, void* workspace, size_t workspace_bytes,
                hipStream_t stream) {
  if (!x || !out || P <= 0 || C <= 0 || ld < C) return UD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < (size_t)kMaxSlices * C * sizeof(float)) return UD_ERR_WORKSPACE;
  float* partial = (float*)workspace;
  UdProfScope prof("bn_act.k_colsum", stream);
  int slices;
  const bool vec = C % 8 == 0 && (ld * sizeof(T)) % 16 == 0 && ((size_t)x & 15) == 0;
  if (vec) {
    const int chunks = C / 8;
    int ch = 1;
    while (ch < chunks && ch < 256) ch <<= 1;
    const int lanes = 256 / ch, cblocks = (chunks + ch - 1) / ch;
    long long s = (P + 2 * lanes - 1) / (2 * lanes);
    const long long want = 2048 / cblocks > 0 ? 2048 / cblocks : 1;
    if (s > want) s = want;
    if (s > kMaxSlices) s = kMaxSlices;
    if (s < 1) s = 1;
    slices = (int)s;
    k_colsum_partial<T><<<dim3(slices, cblocks), 256, 0, stream>>>(x, P, C, ld, ch, partial);
  } else {
    const int cblocks = (C + 255) / 256, cw = C < 256 ? C : 256, R = 256 / cw;
    long long s = (P + 2 * R - 1) / (2 * R);
    const long long want = 2048 / cblocks > 0 ? 2048 / cblocks : 1;
    if (s > want) s = want;
    if (s > kMaxSlices) s = kMaxSlices;
    if (s < 1) s = 1;
    slices = (int)s;
    k_colsum_partial_scalar<T><<<dim3(slices, cblocks), 256, 0, stream>>>(x, P, C, ld, partial);
  }
  UD_LAUNCH_CHECK();
  if (slices > 128) k_colsum_final<256><<<C, 256, 0, stream>>>(partial, slices, C, out);
  else k_colsum_final<64><<<C, 64, 0, stream>>>(partial, slices, C, out);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" {

size_t ud_bn_act_workspace_bytes(int C) {
  if (C <= 0) return 0;
  UdArena ar(nullptr, 0);
  BnWs w;
  return carve(ar, C, &w);
}

typedef unsigned short bf16_t;

int ud_bn_stats(const void* x, long long P, int C, const float* gamma, const float* beta, float eps,
                float* mean, float* var, float* invstd, float* scale, float* shift, float* running_mean,
                float* running_var, float momentum, long long* batches_tracked, void* workspace,
                size_t workspace_bytes, ud_stream_t stream) {
  return bn_stats_impl<bf16_t>((const bf16_t*)x, P, C, gamma, beta, eps, mean, var, invstd, scale, shift, running_mean,
                               running_var, momentum, batches_tracked, workspace, workspace_bytes, (hipStream_t)stream);
}
int ud_bn_stats_f32(const float* x, long long P, int C, const float* gamma, const float* beta, float eps,
                    float* mean, float* var, float* invstd, float* scale, float* shift, float* running_mean,
                    float* running_var, float momentum, long long* batches_tracked, void* workspace,
                    size_t workspace_bytes, ud_stream_t stream) {
  return bn_stats_impl<float>(x, P, C, gamma, beta, eps, mean, var, invstd, scale, shift, running_mean, running_var,
                              momentum, batches_tracked, workspace, workspace_bytes, (hipStream_t)stream);
}

// Second pass of the statistics when the first one came out of a convolution epilogue (ud_conv*_bnstats_nhwc_*):
// partial[slice][C][2] = per-tile (sum, sum of squares) of the P x C tensor, reduced in slice order (deterministic).
int ud_bn_stats_from_partials(const float* partial, int slices, long long P, int C, const float* gamma,
                              const float* beta, float eps, float* mean, float* var, float* invstd, float* scale,
                              float* shift, float* running_mean, float* running_var, float momentum,
                              long long* batches_tracked, ud_stream_t stream) {
  if (!partial || slices <= 0 || !gamma || !beta || !mean || !var || !invstd || !scale || !shift || P <= 0 || C <= 0 ||
      ((running_mean == nullptr) != (running_var == nullptr)))
    return UD_ERR_INVALID_ARG;
  UdProfScope prof("bn_act.stats", (hipStream_t)stream);
  if (C % 16) return UD_ERR_UNSUPPORTED;
  k_bn_stats_final<float><<<C / 16, 256, 0, (hipStream_t)stream>>>(nullptr, partial, slices, P, C, gamma, beta, eps, mean, var,
                                                              invstd, scale, shift, running_mean, running_var, momentum,
                                                              batches_tracked);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

int ud_bn_act_fwd(const void* x, const void* residual, const float* scale, const float* shift, void* y,
                  long long P, int C, int relu, ud_stream_t stream) {
  return bn_fwd_impl<bf16_t>((const bf16_t*)x, (const bf16_t*)residual, scale, shift, (bf16_t*)y, P, C, relu, C,
                             (hipStream_t)stream);
}
int ud_bn_act_fwd_f32(const float* x, const float* residual, const float* scale, const float* shift, float* y,
                      long long P, int C, int relu, ud_stream_t stream) {
  return bn_fwd_impl<float>(x, residual, scale, shift, y, P, C, relu, C, (hipStream_t)stream);
}
// y / dy as a channel slice of a wider channels-last map: rows of C elements every `ld` elements (ld >= C, ld % 8 == 0, the
// slice's first element 16-byte aligned).  The upsampling heads of the BEV trunk write their BatchNorm + ReLU outputs straight
// into the concatenated map and read its gradient in place (base_bev_backbone.py:117-141: torch.cat of the deblock outputs).
int ud_bn_act_fwd_ld(const void* x, const void* residual, const float* scale, const float* shift, void* y,
                     long long P, int C, long long y_ld, int relu, ud_stream_t stream) {
  return bn_fwd_impl<bf16_t>((const bf16_t*)x, (const bf16_t*)residual, scale, shift, (bf16_t*)y, P, C, relu, y_ld,
                             (hipStream_t)stream);
}
int ud_bn_act_fwd_ld_f32(const float* x, const float* residual, const float* scale, const float* shift, float* y,
                         long long P, int C, long long y_ld, int relu, ud_stream_t stream) {
  return bn_fwd_impl<float>(x, residual, scale, shift, y, P, C, relu, y_ld, (hipStream_t)stream);
}

int ud_bn_act_bwd(const void* x, const void* y, const void* dy, const float* scale, const float* shift,
                  const float* mean, const float* invstd, void* dx, void* dresidual, float* dgamma,
                  float* dbeta, long long P, int C, int relu, void* workspace, size_t workspace_bytes,
                  ud_stream_t stream) {
  return bn_bwd_impl<bf16_t>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, scale, shift, mean, invstd,
                             (bf16_t*)dx, (bf16_t*)dresidual, dgamma, dbeta, P, C, relu, C, workspace, workspace_bytes,
                             (hipStream_t)stream);
}
int ud_bn_act_bwd_f32(const float* x, const float* y, const float* dy, const float* scale, const float* shift,
                      const float* mean, const float* invstd, float* dx, float* dresidual, float* dgamma,
                      float* dbeta, long long P, int C, int relu, void* workspace, size_t workspace_bytes,
                      ud_stream_t stream) {
  return bn_bwd_impl<float>(x, y, dy, scale, shift, mean, invstd, dx, dresidual, dgamma, dbeta, P, C, relu, C, workspace,
                            workspace_bytes, (hipStream_t)stream);
}
int ud_bn_act_bwd_ld(const void* x, const void* y, const void* dy, long long dy_ld, const float* scale, const float* shift,
                     const float* mean, const float* invstd, void* dx, void* dresidual, float* dgamma,
                     float* dbeta, long long P, int C, int relu, void* workspace, size_t workspace_bytes,
                     ud_stream_t stream) {
  return bn_bwd_impl<bf16_t>((const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, scale, shift, mean, invstd,
                             (bf16_t*)dx, (bf16_t*)dresidual, dgamma, dbeta, P, C, relu, dy_ld, workspace, workspace_bytes,
                             (hipStream_t)stream);
}
int ud_bn_act_bwd_ld_f32(const float* x, const float* y, const float* dy, long long dy_ld, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float* dx, float* dresidual,
                         float* dgamma, float* dbeta, long long P, int C, int relu, void* workspace,
                         size_t workspace_bytes, ud_stream_t stream) {
  return bn_bwd_impl<float>(x, y, dy, scale, shift, mean, invstd, dx, dresidual, dgamma, dbeta, P, C, relu, dy_ld,
                            workspace, workspace_bytes, (hipStream_t)stream);
}

size_t ud_colsum_workspace_bytes(int C) { return C > 0 ? ud_align_up((size_t)kMaxSlices * C * sizeof(float)) : 0; }
int ud_colsum_f32(const float* x, long long P, int C, long long ld, float* out, void* workspace, size_t workspace_bytes,
                  ud_stream_t stream) {
  return colsum_impl<float>(x, P, C, ld, out, workspace, workspace_bytes, (hipStream_t)stream);
}
int ud_colsum_bf16(const void* x, long long P, int C, long long ld, float* out, void* workspace, size_t workspace_bytes,
                   ud_stream_t stream) {
  return colsum_impl<bf16_t>((const bf16_t*)x, P, C, ld, out, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
