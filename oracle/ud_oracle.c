/*
 * ud_oracle.c -- CPU restatement (plain C, scalar, single thread) of the UniDistill hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the timed CPU baseline.  The product path (cvpr2023-unidistill_amd/) never
 * imports it and fails loudly when its HIP library is missing.
 *
 * Every function cites the reference file:line (relative to /root/reference/unidistill/) whose
 * behaviour it restates.  Pieces whose reference implementation is a third-party binary that is
 * absent from the reference tree (voxel_pooling_ext, spconv) follow the call-site contract and
 * the published algorithm; see the per-function notes and DESIGN.md ("parity pinning").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------
 * BEV pool forward.
 * Reference: layers/blocks_3d/mmdet3d/lss_fpn.py:43-59 -- the wrapper allocates
 * out = zeros[B, ny, nx, C] and pos_memo = -1[B, N, 3] and calls the (missing) extension
 * voxel_pooling_ext.voxel_pooling_forward_wrapper(B, N, C, nx, ny, nz, geom, feat, out, pos).
 * The extension is BEVDepth's voxel_pooling (binary only, .MISSING_LARGE_BLOBS:1); its contract
 * is fixed by the python around it: out layout [B, ny, nx, C] (lss_fpn.py:43-45,62), pos_memo
 * = (b, y, x) for every point that was pooled, -1 otherwise (lss_fpn.py:66-77).  A point is
 * pooled iff 0<=x<nx, 0<=y<ny, 0<=z<nz.  PARITY UNPINNED by the reference (no test, no binary);
 * pinned here by tests/golden/bev_pool_bwd_*.npz for the backward, which IS reference python.
 * Sums run in ascending point order.
 * ------------------------------------------------------------------------------------- */
void oracle_bev_pool_fwd(const int32_t* geom, const float* feat, float* out, int32_t* pos, int B,
                         int N, int C, int nx, int ny, int nz) {
  for (int b = 0; b < B; ++b) {
    for (int n = 0; n < N; ++n) {
      const size_t p = (size_t)b * N + n;
      const int x = geom[p * 3 + 0], y = geom[p * 3 + 1], z = geom[p * 3 + 2];
      if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz) continue;
      pos[p * 3 + 0] = b;
      pos[p * 3 + 1] = y;
      pos[p * 3 + 2] = x;
      float* o = out + (((size_t)b * ny + y) * nx + x) * C;
      const float* f = feat + p * C;
      for (int c = 0; c < C; ++c) o[c] += f[c];
    }
  }
}

/* BEV pool backward.  Reference: lss_fpn.py:64-79 (VoxelPooling.backward):
 * grad_feat[kept] = grad_out[b, :, y, x]; zero elsewhere.  gout is NCHW [B, C, ny, nx]. */
void oracle_bev_pool_bwd(const float* gout_nchw, const int32_t* pos, float* gfeat, int B, int N,
                         int C, int nx, int ny) {
  memset(gfeat, 0, (size_t)B * N * C * sizeof(float));
  for (size_t p = 0; p < (size_t)B * N; ++p) {
    const int b = pos[p * 3 + 0], y = pos[p * 3 + 1], x = pos[p * 3 + 2];
    if (b == -1) continue;
    for (int c = 0; c < C; ++c)
      gfeat[p * C + c] = gout_nchw[(((size_t)b * C + c) * ny + y) * nx + x];
  }
}
