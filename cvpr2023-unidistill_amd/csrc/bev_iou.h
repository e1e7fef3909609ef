// Rotated-BEV IoU of two boxes (x, y, z, dx, dy, dz, heading): shared by nms.hip and proposals.hip.
// Intersection = Sutherland-Hodgman clipping of rectangle A by the four half-planes of rectangle B,
// shoelace area; IoU = inter / max(area_a + area_b - inter, 1e-8) (restates the published behaviour of
// OpenPCDet's iou3d_nms op, which the reference binds as `iou3d_nms_cuda`; binary absent from its tree).
#pragma once
#include "ud_common.h"

namespace ud_iou {

struct P2 { float x, y; };

__device__ __forceinline__ void rect_corners(const float* b, P2* c) {
  const float cs = cosf(b[6]), sn = sinf(b[6]);
  const float hx = 0.5f * b[3], hy = 0.5f * b[4];
  const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};      // counter-clockwise
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c[k].x = b[0] + lx[k] * cs - ly[k] * sn;
    c[k].y = b[1] + lx[k] * sn + ly[k] * cs;
  }
}

// area of (convex polygon A) clipped by the CCW convex polygon B (both rectangles here)
__device__ __forceinline__ float clipped_area(const P2* A, const P2* B) {
  P2 poly[10], tmp[10];
  int n = 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) poly[k] = A[k];
  for (int e = 0; e < 4 && n > 0; ++e) {
    const P2 p = B[e], q = B[(e + 1) & 3];
    const float ex = q.x - p.x, ey = q.y - p.y;
    int m = 0;
    for (int k = 0; k < n; ++k) {
      const P2 s = poly[k], t = poly[(k + 1 == n) ? 0 : k + 1];
      const float ds = ex * (s.y - p.y) - ey * (s.x - p.x);    // >= 0: inside (left of the edge)
      const float dt = ex * (t.y - p.y) - ey * (t.x - p.x);
      if (ds >= 0.f) tmp[m++] = s;
      if ((ds >= 0.f) != (dt >= 0.f)) {
        const float u = ds / (ds - dt);
        tmp[m].x = s.x + u * (t.x - s.x);
        tmp[m].y = s.y + u * (t.y - s.y);
        ++m;
      }
    }
    n = m;
    for (int k = 0; k < n; ++k) poly[k] = tmp[k];
  }
  if (n < 3) return 0.f;
  float a2 = 0.f;
  for (int k = 0; k < n; ++k) {
    const P2 s = poly[k], t = poly[(k + 1 == n) ? 0 : k + 1];
    a2 += s.x * t.y - t.x * s.y;
  }
  return 0.5f * fabsf(a2);
}

__device__ __forceinline__ float iou_bev(const float* a, const float* b) {
  P2 ca[4], cb[4];
  rect_corners(a, ca);
  rect_corners(b, cb);
  const float inter = clipped_area(ca, cb);
  const float sa = a[3] * a[4], sb = b[3] * b[4];
  return inter / fmaxf(sa + sb - inter, 1e-8f);
}

}  // namespace ud_iou
