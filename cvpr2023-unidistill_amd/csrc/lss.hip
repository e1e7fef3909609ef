// Camera lift-splat (LSS) kernels for MI355X / gfx950: frustum geometry + BEV binning, depth
// softmax, lift (materialised, reference boundary) and the backward of the fused lift+splat.
//
// Reference: unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py
//   :173-198 create_frustum      :200-240 get_geometry      :289-292 softmax (x) context
//   :310 permute(0,1,3,4,5,2)    :311-313 binning (.int() truncates toward zero)
//
// The reference runs get_geometry as ~10 batched 4x4 matmul launches over a [B,6,112,16,44,4,1]
// tensor plus two torch.inverse calls, then materialises the lifted [B,N,C] tensor twice
// (484 MB each at C=256).  Here: one tiny kernel inverts/combines the per-camera matrices, one
// kernel maps every frustum point to its BEV bin, and the fused path (ud_lss_splat_fwd in
// bev_pool.hip + k_splat_bwd below) pools depth_prob * context directly.
#include "ud_common.h"
#include "ud_prof.h"
#include "lss_geom.h"

namespace {

// ---- per-camera matrices --------------------------------------------------------------------
// mats[cam] = { inverse(ida) , sensor2ego @ inverse(intrin) , bda }  (3 x 16 floats, row major)
__device__ bool inv4x4(const float* a, double* o) {
  double m[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      m[i][j] = (double)a[i * 4 + j];
      m[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    double best = fabs(m[c][c]);
    for (int r = c + 1; r < 4; ++r)
      if (fabs(m[r][c]) > best) {
        best = fabs(m[r][c]);
        piv = r;
      }
    if (best == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < 8; ++j) {
        const double t = m[c][j];
        m[c][j] = m[piv][j];
        m[piv][j] = t;
      }
    const double inv = 1.0 / m[c][c];
    for (int j = 0; j < 8; ++j) m[c][j] *= inv;
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        const double f = m[r][c];
        if (f != 0.0)
          for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
      }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) o[i * 4 + j] = m[i][4 + j];
  return true;
}

// Products follow the arithmetic of the reference's CPU path bit for bit: torch's small batched
// matmul (contraction*rows*cols < 400, aten/native/LinearAlgebra.cpp baddbmm_cpu_kernel) is a plain
// loop  acc = 0; acc += a[k]*b[k]  in fp32 with separately rounded product and sum (no FMA).
__device__ __forceinline__ float dot4_seq(float a0, float b0, float a1, float b1, float a2, float b2,
                                          float a3, float b3) {
  float acc = __fadd_rn(0.0f, __fmul_rn(a0, b0));
  acc = __fadd_rn(acc, __fmul_rn(a1, b1));
  acc = __fadd_rn(acc, __fmul_rn(a2, b2));
  acc = __fadd_rn(acc, __fmul_rn(a3, b3));
  return acc;
}

__global__ void k_prepare_mats(const float* __restrict__ s2e, const float* __restrict__ intrin,
                               const float* __restrict__ ida, const float* __restrict__ bda,
                               const float* __restrict__ ida_inv, const float* __restrict__ intrin_inv,
                               float* __restrict__ mats, int B, int ncam) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * ncam) return;
  const int b = i / ncam;
  double inv[16];
  float* o = mats + (size_t)i * 48;
  // inverse(ida): the caller's fp32 inverse (torch.inverse, as the reference computes it), or the
  // correctly rounded one (Gauss-Jordan in fp64, one rounding to fp32)
  if (ida_inv) {
    for (int k = 0; k < 16; ++k) o[k] = ida_inv[(size_t)i * 16 + k];
  } else {
    if (!inv4x4(ida + (size_t)i * 16, inv))
      for (int k = 0; k < 16; ++k) inv[k] = __builtin_nan("");
    for (int k = 0; k < 16; ++k) o[k] = (float)inv[k];
  }
  // combine = sensor2ego @ inverse(intrin) (lss_fpn.py:233): fp32 inverse, then the fp32 product
  float kin[16];
  if (intrin_inv) {
    for (int k = 0; k < 16; ++k) kin[k] = intrin_inv[(size_t)i * 16 + k];
  } else {
    if (!inv4x4(intrin + (size_t)i * 16, inv))
      for (int k = 0; k < 16; ++k) inv[k] = __builtin_nan("");
    for (int k = 0; k < 16; ++k) kin[k] = (float)inv[k];
  }
  const float* s = s2e + (size_t)i * 16;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      o[16 + r * 4 + c] = dot4_seq(s[r * 4 + 0], kin[0 * 4 + c], s[r * 4 + 1], kin[1 * 4 + c],
                                   s[r * 4 + 2], kin[2 * 4 + c], s[r * 4 + 3], kin[3 * 4 + c]);
  for (int k = 0; k < 16; ++k)
    o[32 + k] = bda ? bda[(size_t)b * 16 + k] : ((k % 5 == 0) ? 1.0f : 0.0f);
}

// One thread per frustum point (b, cam, d, h, w): ego coordinates + BEV bin (arithmetic in lss_geom.h).
__global__ __launch_bounds__(256) void k_geometry(UdFrustum f, int ncams_total, float* __restrict__ geom,
                                                  int32_t* __restrict__ bins) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)f.D * f.fH * f.fW * ncams_total) return;
  float q[4];
  int bx, by, bz;
  ud_frustum_point(f, gid, q, &bx, &by, &bz);
  if (geom) {
    geom[gid * 3 + 0] = q[0];
    geom[gid * 3 + 1] = q[1];
    geom[gid * 3 + 2] = q[2];
  }
  bins[gid * 3 + 0] = bx;
  bins[gid * 3 + 1] = by;
  bins[gid * 3 + 2] = bz;
}

// ---- depth softmax --------------------------------------------------------------------------
// One thread per pixel (coalesced across the pixel axis): softmax over the first D channels of
// depth_feature[bn, :, h, w] (any strides) -> prob[bn, D, fH*fW] dense.
__global__ __launch_bounds__(256) void k_depth_softmax(const float* __restrict__ x, long long sn,
                                                       long long sc, long long sh, long long sw,
                                                       float* __restrict__ prob, int BN, int D,
                                                       int fH, int fW) {
  const int HW = fH * fW;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)BN * HW) return;
  const int bn = (int)(gid / HW);
  const int pix = (int)(gid - (long long)bn * HW);
  const int h = pix / fW, w = pix - h * fW;
  const float* px = x + bn * sn + h * sh + w * sw;
  float mx = -INFINITY;
  for (int d = 0; d < D; ++d) mx = fmaxf(mx, px[d * sc]);
  float sum = 0.f;
  float* po = prob + (size_t)bn * D * HW + pix;
  for (int d = 0; d < D; ++d) {
    const float e = expf(px[d * sc] - mx);
    po[(size_t)d * HW] = e;
    sum += e;
  }
  for (int d = 0; d < D; ++d) po[(size_t)d * HW] = __fdiv_rn(po[(size_t)d * HW], sum);
}

// Generic 2-D tile transpose between a strided [Bt, R, K] view and a dense [Bt, K, R] buffer:
//   to_dense:   dst[b, k, r] = src[b*sb + r*sr + k*sk]          (dst dense [Bt, K, R])
//   from_dense: dst[b*sb + r*sr + k*sk] = src[b, k, r]          (src dense [Bt, K, R])
template <bool TO_DENSE>
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ src,
                                                   float* __restrict__ dst, long long sb,
                                                   long long sr, long long sk, int R, int K) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int k0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  if (TO_DENSE) {
    // strided side: K is the fast axis when sk == 1, so read with tx along k
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, k = k0 + tx;
      tile[i][tx] = (r < R && k < K) ? src[b * sb + r * sr + k * sk] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
      const int k = k0 + i, r = r0 + tx;
      if (r < R && k < K) dst[((size_t)b * K + k) * R + r] = tile[tx][i];
    }
  } else {
    for (int i = ty; i < 32; i += 8) {
      const int k = k0 + i, r = r0 + tx;
      tile[i][tx] = (r < R && k < K) ? src[((size_t)b * K + k) * R + r] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
      const int r = r0 + i, k = k0 + tx;
      if (r < R && k < K) dst[b * sb + r * sr + k * sk] = tile[tx][i];
    }
  }
}

// ---- lift, materialised (reference boundary) ------------------------------------------------
// lifted[bn, d, pix, :] = prob[bn, d, pix] * ctx[bn, pix, :]   one wave per (bn, pix, 8 depths)
__global__ __launch_bounds__(256) void k_lift_fwd(const float* __restrict__ prob,
                                                  const float* __restrict__ ctx,
                                                  float* __restrict__ lifted, int BN, int D, int HW,
                                                  int C) {
  const int lane = ud_lane();
  const long long item = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int dgroups = (D + 7) / 8;
  if (item >= (long long)BN * HW * dgroups) return;
  const int dg = (int)(item % dgroups);
  const long long t = item / dgroups;
  const int pix = (int)(t % HW);
  const int bn = (int)(t / HW);
  const float* crow = ctx + ((size_t)bn * HW + pix) * C;
  for (int ch = lane * 4; ch < C; ch += 256) {
    const float4 c = *reinterpret_cast<const float4*>(crow + ch);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int d = dg * 8 + u;
      if (d < D) {
        const float p = prob[((size_t)bn * D + d) * HW + pix];
        float4 o = make_float4(__fmul_rn(p, c.x), __fmul_rn(p, c.y), __fmul_rn(p, c.z),
                               __fmul_rn(p, c.w));
        ud_stg_stream(lifted + (((size_t)bn * D + d) * HW + pix) * C + ch, o);   // written once
      }
    }
  }
}

// Backward of depth softmax (x) context for one pixel per wave.  GRAD_FROM_CELLS = false: the
// upstream grad is the materialised g_lifted[bn, d, pix, :]; true (fused splat): the grad row of
// point (bn,d,pix) is gout[cell(point), :] gathered through pos_memo.
// Emits, pixel-major, g[bn, pix, 0:D] = softmax backward and g[bn, pix, D:D+C] = context grad.
template <bool GRAD_FROM_CELLS>
__global__ __launch_bounds__(256) void k_lift_bwd(const float* __restrict__ gsrc,
                                                  const int32_t* __restrict__ pos,
                                                  const float* __restrict__ prob,
                                                  const float* __restrict__ ctx,
                                                  float* __restrict__ g_pm, int BN, int ncam, int D,
                                                  int HW, int C, int nx, int ny) {
  const int lane = ud_lane();
  const long long item = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (item >= (long long)BN * HW) return;
  const int pix = (int)(item % HW);
  const int bn = (int)(item / HW);
  const float* crow = ctx + ((size_t)bn * HW + pix) * C;
  float* grow = g_pm + ((size_t)bn * HW + pix) * (D + C);
  // this lane's depth slots: d = lane, lane + 64, ...   (D <= 256 supported, checked by the host)
  float pr[4], gp[4];
  int cell[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d = lane + 64 * s;
    pr[s] = 0.f;
    gp[s] = 0.f;
    cell[s] = -1;
    if (d < D) {
      const size_t pt = ((size_t)bn * D + d) * HW + pix;
      pr[s] = prob[pt];
      if (GRAD_FROM_CELLS) {
        const int b = pos[pt * 3 + 0];
        if (b >= 0) cell[s] = (b * ny + pos[pt * 3 + 1]) * nx + pos[pt * 3 + 2];
      } else {
        cell[s] = 0;
      }
    }
  }
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int ch = c0 + lane * 4;
    const bool act = ch < C;
    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) cv = *reinterpret_cast<const float4*>(crow + ch);
    float4 gc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s * 64 >= D) break;
      const int dmax = min(64, D - s * 64);
      for (int j = 0; j < dmax; ++j) {
        const int cl = __builtin_amdgcn_readlane(cell[s], j);
        if (cl < 0) continue;  // point fell outside the BEV grid: zero gradient
        const float p = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[s]), j));
        const int d = s * 64 + j;
        const float* src = GRAD_FROM_CELLS ? gsrc + (size_t)cl * C
                                           : gsrc + (((size_t)bn * D + d) * HW + pix) * C;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) g = GRAD_FROM_CELLS ? *reinterpret_cast<const float4*>(src + ch) : ud_ldg_stream(src + ch);
        gc.x = fmaf(p, g.x, gc.x);
        gc.y = fmaf(p, g.y, gc.y);
        gc.z = fmaf(p, g.z, gc.z);
        gc.w = fmaf(p, g.w, gc.w);
        float dot = g.x * cv.x + g.y * cv.y + g.z * cv.z + g.w * cv.w;
        dot = ud_wave_sum(dot);
        if (lane == j) gp[s] += dot;
      }
    }
    if (act) *reinterpret_cast<float4*>(grow + D + ch) = gc;
  }
  // softmax backward: g_logit[d] = p[d] * (gp[d] - sum_j p[j] gp[j])
  float part = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) part += pr[s] * gp[s];
  const float tot = ud_wave_sum(part);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d = lane + 64 * s;
    if (d < D) grow[d] = pr[s] * (gp[s] - tot);
  }
}

bool lss_ok(int BN, int D, int fH, int fW, int C) {
  return BN > 0 && D > 0 && D <= 256 && fH > 0 && fW > 0 && C > 0 &&
         (long long)BN * D * fH * fW < (1ll << 31);
}

}  // namespace

extern "C" int ud_lss_prepare_mats(const float* sensor2ego, const float* intrin, const float* ida,
                                   const float* bda, const float* ida_inv,
                                   const float* intrin_inv, int B, int ncam, float* mats,
                                   ud_stream_t stream_) {
  if (!sensor2ego || !intrin || !ida || !mats || B <= 0 || ncam <= 0) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  k_prepare_mats<<<ud_div_up((long long)B * ncam, 64), 64, 0, stream>>>(
      sensor2ego, intrin, ida, bda, ida_inv, intrin_inv, mats, B, ncam);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_lss_geometry(const float* mats, const float* frustum_u, const float* frustum_v,
                               const float* frustum_d, int B, int ncam, int D, int fH, int fW,
                               const float* lo, const float* size, int has_bda, float* geom,
                               int32_t* bins, ud_stream_t stream_) {
  if (!mats || !frustum_u || !frustum_v || !frustum_d || !lo || !size || !bins)
    return UD_ERR_INVALID_ARG;
  if (!lss_ok(B * ncam, D, fH, fW, 1)) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const long long total = (long long)B * ncam * D * fH * fW;
  UdProfScope prof("lss.k_geometry", stream);
  const UdFrustum f{mats, frustum_u, frustum_v, frustum_d, D, fH, fW, has_bda, lo[0], lo[1], lo[2], size[0], size[1], size[2]};
  k_geometry<<<ud_div_up(total, 256), 256, 0, stream>>>(f, B * ncam, geom, bins);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

// depth_feature (strided [BN, D+C, fH, fW]) -> prob [BN, D, fH*fW] dense, ctx_pm [BN, fH*fW, C]
extern "C" int ud_lss_depth_ctx(const float* depth_feature, int64_t sn, int64_t sc, int64_t sh,
                                int64_t sw, int BN, int D, int C, int fH, int fW, float* prob,
                                float* ctx_pm, ud_stream_t stream_) {
  if (!depth_feature || !prob || !ctx_pm || !lss_ok(BN, D, fH, fW, C)) return UD_ERR_INVALID_ARG;
  if (sw * fW != sh) return UD_ERR_UNSUPPORTED;  // pixels must be addressable as one axis
  hipStream_t stream = (hipStream_t)stream_;
  const int HW = fH * fW;
  k_depth_softmax<<<ud_div_up((long long)BN * HW, 256), 256, 0, stream>>>(
      depth_feature, sn, sc, sh, sw, prob, BN, D, fH, fW);
  UD_LAUNCH_CHECK();
  // context channels [D, D+C): view [BN, R=C, K=HW] strided (sr = sc, sk = sw) -> dense [BN, HW, C]
  dim3 grid(ud_div_up(HW, 32), ud_div_up(C, 32), BN);
  k_transpose<true><<<grid, 256, 0, stream>>>(depth_feature + (size_t)D * sc, ctx_pm, sn, sc, sw,
                                              C, HW);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" int ud_lss_lift_fwd(const float* prob, const float* ctx_pm, float* lifted, int BN,
                               int D, int C, int fH, int fW, ud_stream_t stream_) {
  if (!prob || !ctx_pm || !lifted || !lss_ok(BN, D, fH, fW, C) || C % 4) return UD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const long long items = (long long)BN * fH * fW * ((D + 7) / 8);
  UdProfScope prof("lss.k_lift_fwd", stream);
  k_lift_fwd<<<ud_div_up(items, 4), 256, 0, stream>>>(prob, ctx_pm, lifted, BN, D, fH * fW, C);
  UD_LAUNCH_CHECK();
  return UD_OK;
}

extern "C" size_t ud_lss_lift_bwd_workspace_bytes(int BN, int D, int C, int fH, int fW) {
  if (!lss_ok(BN, D, fH, fW, C)) return 0;
  return ud_align_up((size_t)BN * fH * fW * (D + C) * sizeof(float));
}

// Backward of the lift.  g_lifted (dense [BN, D, HW, C]) when pos == NULL; otherwise the fused
// splat backward: gout is a dense NHWC BEV grad [B, ny, nx, C] gathered through pos [B*N, 3].
// Writes g_depth_feature through strides (sn, sc, sh, sw) (e.g. an NCHW conv-output grad).
extern "C" int ud_lss_lift_bwd(const float* gsrc, const int32_t* pos, const float* prob,
                               const float* ctx_pm, float* g_depth_feature, int64_t sn, int64_t sc,
                               int64_t sh, int64_t sw, int BN, int ncam, int D, int C, int fH,
                               int fW, int nx, int ny, void* workspace, size_t workspace_bytes,
                               ud_stream_t stream_) {
  if (!gsrc || !prob || !ctx_pm || !g_depth_feature || !lss_ok(BN, D, fH, fW, C) || C % 4)
    return UD_ERR_INVALID_ARG;
  if (sw * fW != sh) return UD_ERR_UNSUPPORTED;
  const size_t need = ud_lss_lift_bwd_workspace_bytes(BN, D, C, fH, fW);
  if (!workspace || workspace_bytes < need) return UD_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int HW = fH * fW;
  float* g_pm = (float*)workspace;
  {
    UdProfScope prof(pos ? "lss.k_splat_bwd" : "lss.k_lift_bwd", stream);
    if (pos)
      k_lift_bwd<true><<<ud_div_up((long long)BN * HW, 4), 256, 0, stream>>>(
          gsrc, pos, prob, ctx_pm, g_pm, BN, ncam, D, HW, C, nx, ny);
    else
      k_lift_bwd<false><<<ud_div_up((long long)BN * HW, 4), 256, 0, stream>>>(
          gsrc, pos, prob, ctx_pm, g_pm, BN, ncam, D, HW, C, nx, ny);
    UD_LAUNCH_CHECK();
  }
  // pixel-major [BN, HW, D+C] -> strided [BN, D+C, fH, fW]:  R = D+C (stride sc), K = HW (stride sw)
  // dense side is [Bt, K=HW, R=D+C]
  dim3 grid(ud_div_up(HW, 32), ud_div_up(D + C, 32), BN);
  k_transpose<false><<<grid, 256, 0, stream>>>(g_pm, g_depth_feature, sn, sc, sw, D + C, HW);
  UD_LAUNCH_CHECK();
  return UD_OK;
}
