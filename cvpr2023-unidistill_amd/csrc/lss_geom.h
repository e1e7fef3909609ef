// Frustum point -> ego coordinates -> BEV bin: the arithmetic of LSSFPN.get_geometry + the binning of
// LSSFPN._forward_single_sweep (reference lss_fpn.py:200-240, 311-313), shared by k_geometry (lss.hip) and the fused
// geometry + binning kernel of the lift-splat (bev_pool.hip) so that both produce the same bits.
#pragma once
#include <stdint.h>

// The reference's CPU matmul for these 4x4 products is  acc = 0; acc += a[k]*b[k]  in fp32 with separately rounded product
// and sum (no FMA); see lss.hip.
__device__ __forceinline__ float ud_dot4_seq(float a0, float b0, float a1, float b1, float a2, float b2, float a3, float b3) {
  float acc = __fadd_rn(0.0f, __fmul_rn(a0, b0));
  acc = __fadd_rn(acc, __fmul_rn(a1, b1));
  acc = __fadd_rn(acc, __fmul_rn(a2, b2));
  acc = __fadd_rn(acc, __fmul_rn(a3, b3));
  return acc;
}

__device__ __forceinline__ void ud_mat4_apply(const float* __restrict__ m, const float* p, float* q) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
    q[r] = ud_dot4_seq(m[r * 4 + 0], p[0], m[r * 4 + 1], p[1], m[r * 4 + 2], p[2], m[r * 4 + 3], p[3]);
}

struct UdFrustum {           // what a frustum point needs: mats f32[B*ncam,3,16] (ud_lss_prepare_mats), the frustum axes, the grid
  const float* mats;
  const float* fu;
  const float* fv;
  const float* fd;
  int D, fH, fW, has_bda;
  float lo0, lo1, lo2, sz0, sz1, sz2;
};

// point gid = ((cam * D + d) * fH + h) * fW + w  (cam = b * ncam + camera): ego coordinates q[0..2] and bins
__device__ __forceinline__ void ud_frustum_point(const UdFrustum& f, long long gid, float* q, int* bx, int* by, int* bz) {
  const long long per_cam = (long long)f.D * f.fH * f.fW;
  const int cam = (int)(gid / per_cam);
  int r = (int)(gid - cam * per_cam);
  const int d = r / (f.fH * f.fW);
  r -= d * f.fH * f.fW;
  const int h = r / f.fW;
  const int w = r - h * f.fW;
  const float* m = f.mats + (size_t)cam * 48;
  float p[4] = {f.fu[w], f.fv[h], f.fd[d], 1.0f};
  ud_mat4_apply(m, p, q);                 // undo image-space augmentation (lss_fpn.py:221-222)
  p[0] = __fmul_rn(q[0], q[2]);           // (u*d, v*d, d, 1)                 (:225-231)
  p[1] = __fmul_rn(q[1], q[2]);
  p[2] = q[2];
  p[3] = q[3];
  ud_mat4_apply(m + 16, p, q);            // camera -> ego                    (:233-234)
  if (f.has_bda) {                        // BEV-space augmentation           (:235-239)
    p[0] = q[0];
    p[1] = q[1];
    p[2] = q[2];
    p[3] = q[3];
    ud_mat4_apply(m + 32, p, q);
  }
  // ((geom - (voxel_coord - voxel_size/2)) / voxel_size).int()              (:311-313)
  *bx = (int)__fdiv_rn(__fsub_rn(q[0], f.lo0), f.sz0);
  *by = (int)__fdiv_rn(__fsub_rn(q[1], f.lo1), f.sz1);
  *bz = (int)__fdiv_rn(__fsub_rn(q[2], f.lo2), f.sz2);
}
