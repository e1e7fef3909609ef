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
#ifdef _OPENMP
#include <omp.h>
#endif
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
#ifdef _OPENMP
  /* OpenMP build (bench.py's cpu_baseline only; the checker is the scalar build): thread t owns the BEV rows y % T == t, walks
   * ALL points in input order and adds those that land in its rows -- every cell still sums in ascending point order, so the
   * result is bit-identical to the loop below. */
#pragma omp parallel
  {
    const int T = omp_get_num_threads(), t = omp_get_thread_num();
    for (int b = 0; b < B; ++b)
      for (int n = 0; n < N; ++n) {
        const size_t p = (size_t)b * N + n;
        const int x = geom[p * 3 + 0], y = geom[p * 3 + 1], z = geom[p * 3 + 2];
        if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz || y % T != t) continue;
        pos[p * 3 + 0] = b;
        pos[p * 3 + 1] = y;
        pos[p * 3 + 2] = x;
        float* o = out + (((size_t)b * ny + y) * nx + x) * C;
        const float* f = feat + p * C;
        for (int c = 0; c < C; ++c) o[c] += f[c];
      }
  }
  return;
#endif
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

/* The lift of LSSFPN._forward_single_sweep (lss_fpn.py:289-310): prob = softmax over the D depth bins, lifted[n][d][h][w][c] =
 * prob[n][d][h][w] * ctx[n][c][h][w].  x = depth net output [BN][D + C][fH][fW].  (numpy formulation: oracle.lss_lift; this C
 * twin exists for the OpenMP baseline and is checked against it in tests/test_oracle_golden.py.) */
void oracle_lss_lift(const float* x, int BN, int D, int C, int fH, int fW, float* lifted, float* prob) {
  const size_t HW = (size_t)fH * fW;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static)
#endif
  for (int n = 0; n < BN; ++n)
    for (size_t q = 0; q < HW; ++q) {
      const float* z = x + (size_t)n * (D + C) * HW + q;
      float m = z[0];
      for (int d = 1; d < D; ++d) m = z[d * HW] > m ? z[d * HW] : m;
      float sum = 0.f;
      float* pr = prob + (size_t)n * D * HW + q;
      for (int d = 0; d < D; ++d) {
        pr[d * HW] = expf(z[d * HW] - m);
        sum += pr[d * HW];
      }
      for (int d = 0; d < D; ++d) {
        pr[d * HW] = pr[d * HW] / sum;
        float* o = lifted + (((size_t)n * D + d) * HW + q) * C;
        const float pv = pr[d * HW];
        for (int c = 0; c < C; ++c) o[c] = pv * z[(size_t)(D + c) * HW];
      }
    }
}

/* BEV pool backward.  Reference: lss_fpn.py:64-79 (VoxelPooling.backward):
 * grad_feat[kept] = grad_out[b, :, y, x]; zero elsewhere.  gout is NCHW [B, C, ny, nx]. */
void oracle_bev_pool_bwd(const float* gout_nchw, const int32_t* pos, float* gfeat, int B, int N,
                         int C, int nx, int ny) {
  memset(gfeat, 0, (size_t)B * N * C * sizeof(float));
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (size_t p = 0; p < (size_t)B * N; ++p) {
    const int b = pos[p * 3 + 0], y = pos[p * 3 + 1], x = pos[p * 3 + 2];
    if (b == -1) continue;
    for (int c = 0; c < C; ++c)
      gfeat[p * C + c] = gout_nchw[(((size_t)b * C + c) * ny + y) * nx + x];
  }
}

/* ---------------------------------------------------------------------------------------
 * Hard voxelization (PointToVoxel) + MeanVFE.
 * Reference call site: data/det3d/preprocess/voxelization.py:31-38 (ctor), :53-58 (per-sample
 * call, clone), :59-68 (batch index prepended to coords -> (b, z, y, x)).
 * The voxelizer itself is spconv.pytorch.utils.PointToVoxel from the third-party wheel
 * spconv-cu111>=2.1.12 (requirements.txt:16), NOT in the reference tree -> PARITY UNPINNED by
 * the reference.  This restates spconv's published CPU algorithm (the GPU hash version is order
 * non-deterministic): for each point in input order: c = floor((p - range_min) / voxel_size) per
 * axis, skip unless 0 <= c < grid; look the voxel up; if new and the sample already has
 * max_voxels voxels, skip the point, else create it (voxels numbered by first appearance);
 * if the voxel holds fewer than P points, append the point; count it.  num = min(count, P)
 * because appended points stop at P.  grid = round((max - min) / voxel_size)
 * (voxelization.py:40-43).
 * MeanVFE (layers/blocks_3d/det3d/vfe/mean_vfe.py:25-32, pinned by tests/golden/mean_vfe.npz):
 * mean = sum over the P slots / max(num, 1).
 * Outputs are concatenated over the batch like Voxelization.forward does.
 * Returns the total number of voxels; m_out[b] = voxels of sample b, m_out[B] = total.
 * ------------------------------------------------------------------------------------- */
typedef struct {
  int64_t* keys;
  int32_t* vals;
  size_t mask;
} omap_t;

static void omap_init(omap_t* m, size_t n) {
  size_t cap = 1024;
  while (cap < 2 * n) cap <<= 1;
  m->keys = (int64_t*)malloc(cap * sizeof(int64_t));
  m->vals = (int32_t*)malloc(cap * sizeof(int32_t));
  for (size_t i = 0; i < cap; ++i) m->keys[i] = -1;
  m->mask = cap - 1;
}
static void omap_free(omap_t* m) {
  free(m->keys);
  free(m->vals);
}
/* returns pointer to the value slot for key; *found tells whether it existed */
static int32_t* omap_slot(omap_t* m, int64_t key, int* found) {
  size_t h = ((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 20 & m->mask;
  while (m->keys[h] != -1 && m->keys[h] != key) h = (h + 1) & m->mask;
  *found = (m->keys[h] == key);
  m->keys[h] = key;
  return &m->vals[h];
}

int oracle_voxelize(const float* points, int B, int N, int F, const float* voxel_size,
                    const float* range, int P, int max_voxels, float* voxels /*[cap,P,F]*/,
                    int32_t* coords /*[cap,4]*/, int32_t* num /*[cap]*/, float* mean /*[cap,F]*/,
                    int32_t* m_out /*[B+1]*/) {
  int grid[3];
  for (int a = 0; a < 3; ++a)
    grid[a] = (int)llround(((double)range[a + 3] - (double)range[a]) / (double)voxel_size[a]);
  int row0 = 0;
  for (int b = 0; b < B; ++b) {
    omap_t map;
    omap_init(&map, (size_t)N);
    int nvox = 0;
    for (int n = 0; n < N; ++n) {
      const float* q = points + ((size_t)b * N + n) * F;
      int c[3];
      int ok = 1;
      for (int a = 0; a < 3; ++a) {
        const float f = floorf((q[a] - range[a]) / voxel_size[a]);
        if (!(f >= 0.0f && f < (float)grid[a])) {
          ok = 0;
          break;
        }
        c[a] = (int)f;
      }
      if (!ok) continue;
      const int64_t key = ((int64_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
      int found;
      int32_t* slot = omap_slot(&map, key, &found);
      int row;
      if (!found) {
        if (nvox >= max_voxels) {
          *slot = -1; /* remembered as "rejected": later points of this voxel are skipped too */
          continue;
        }
        row = row0 + nvox++;
        *slot = row;
        coords[row * 4 + 0] = b;
        coords[row * 4 + 1] = c[2];
        coords[row * 4 + 2] = c[1];
        coords[row * 4 + 3] = c[0];
        num[row] = 0;
        if (voxels) memset(voxels + (size_t)row * P * F, 0, (size_t)P * F * sizeof(float));
      } else {
        row = *slot;
        if (row < 0) continue;
      }
      if (num[row] < P) {
        if (voxels)
          memcpy(voxels + ((size_t)row * P + num[row]) * F, q, (size_t)F * sizeof(float));
        else if (mean) { /* fused mean without materialised voxels: accumulate in slot order */
          float* mrow = mean + (size_t)row * F;
          for (int f = 0; f < F; ++f) mrow[f] = (num[row] == 0 ? 0.0f : mrow[f]) + q[f];
        }
        num[row] += 1;
      }
    }
    omap_free(&map);
    m_out[b] = nvox;
    row0 += nvox;
  }
  m_out[B] = row0;
  if (mean) {
    for (int r = 0; r < row0; ++r) {
      const float den = (float)(num[r] > 1 ? num[r] : 1);
      for (int f = 0; f < F; ++f) {
        float acc;
        if (voxels) {
          acc = 0.0f;
          for (int j = 0; j < P; ++j) acc += voxels[((size_t)r * P + j) * F + f];
        } else {
          acc = mean[(size_t)r * F + f];
        }
        mean[(size_t)r * F + f] = acc / den;
      }
    }
  }
  return row0;
}

/* MeanVFE alone (mean_vfe.py:25-32): voxels[M,P,F], num[M] -> out[M,F]. */
void oracle_mean_vfe(const float* voxels, const int32_t* num, float* out, int M, int P, int F) {
  for (int r = 0; r < M; ++r) {
    const float den = (float)(num[r] > 1 ? num[r] : 1);
    for (int f = 0; f < F; ++f) {
      float acc = 0.0f;
      for (int j = 0; j < P; ++j) acc += voxels[((size_t)r * P + j) * F + f];
      out[(size_t)r * F + f] = acc / den;
    }
  }
}

/* ---------------------------------------------------------------------------------------
 * Sparse 3-D convolution (spconv, third-party wheel spconv-cu111>=2.1.12, requirements.txt:16;
 * NOT in the reference tree -> PARITY UNPINNED by the reference).  Call sites:
 * layers/blocks_3d/det3d/spconv_backbone.py:21-48 (post_act_block), :71-92 (SparseBasicBlock),
 * :259-340 (VoxelResBackBone8x layers), :354-359 (SparseConvTensor(features, indices(b,z,y,x),
 * spatial_shape, batch_size)).  Textbook semantics restated here:
 *   SubMConv3d(k odd): output sites = input sites; out[o] = bias + sum over offsets d in the
 *     k-cube of W[:, d, :] . in[site at coord(o) + d - k/2] (missing neighbours contribute 0).
 *   SparseConv3d(k, s, p): output grid floor((in + 2p - k)/s) + 1; output sites = every cell
 *     reachable from an active input (o*s - p + d = i for some d); out[o] = bias + sum_d
 *     W[:, d, :] . in[site at o*s - p + d].  Output rows are emitted in ascending (b,z,y,x).
 * Weights are KRSC: W[n][d][c] with d enumerating (dz, dy, dx), dz slowest.
 * ------------------------------------------------------------------------------------- */
static int64_t lin4(int b, int z, int y, int x, int Dz, int Hy, int Wx) {
  return (((int64_t)b * Dz + z) * Hy + y) * Wx + x;
}

void oracle_spconv_subm_rulebook(const int32_t* coords, int M, int B, int Dz, int Hy, int Wx,
                                 int kz, int ky, int kx, int32_t* nbr) {
  (void)B;
  omap_t map;
  omap_init(&map, (size_t)M + 1);
  int found;
  for (int i = 0; i < M; ++i)
    *omap_slot(&map, lin4(coords[i * 4], coords[i * 4 + 1], coords[i * 4 + 2], coords[i * 4 + 3],
                          Dz, Hy, Wx), &found) = i;
  const int K = kz * ky * kx;
  for (int o = 0; o < M; ++o)
    for (int k = 0; k < K; ++k) {
      const int z = coords[o * 4 + 1] + k / (ky * kx) - kz / 2;
      const int y = coords[o * 4 + 2] + (k / kx) % ky - ky / 2;
      const int x = coords[o * 4 + 3] + k % kx - kx / 2;
      int r = -1;
      if (z >= 0 && z < Dz && y >= 0 && y < Hy && x >= 0 && x < Wx) {
        const int64_t key = lin4(coords[o * 4], z, y, x, Dz, Hy, Wx);
        size_t h = ((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 20 & map.mask;
        while (map.keys[h] != -1 && map.keys[h] != key) h = (h + 1) & map.mask;
        if (map.keys[h] == key) r = map.vals[h];
      }
      nbr[(size_t)o * K + k] = r;
    }
  omap_free(&map);
}

static int cmp_i64(const void* a, const void* b) {
  const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* returns Mout; out_coords must hold 8*Min rows for k=3,s=2 (generally k^3 * Min). */
int oracle_spconv_down(const int32_t* in_coords, int Min, int B, int Dz, int Hy, int Wx,
                       const int* ks, const int* st, const int* pd, int32_t* out_coords,
                       int32_t* out_nbr /* [Mout,K] or NULL */, int32_t* in_nbr /* [Min,K] or NULL */) {
  (void)B;
  int od[3];
  const int id[3] = {Dz, Hy, Wx};
  for (int a = 0; a < 3; ++a) od[a] = (id[a] + 2 * pd[a] - ks[a]) / st[a] + 1;
  const int K = ks[0] * ks[1] * ks[2];
  int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * (size_t)(Min > 0 ? Min : 1) * K);
  size_t nk = 0;
  for (int i = 0; i < Min; ++i) {
    const int b = in_coords[i * 4];
    for (int k = 0; k < K; ++k) {
      const int d[3] = {k / (ks[1] * ks[2]), (k / ks[2]) % ks[1], k % ks[2]};
      int o[3], ok = 1;
      for (int a = 0; a < 3; ++a) {
        const int t = in_coords[i * 4 + 1 + a] + pd[a] - d[a];
        if (t < 0 || t % st[a] || t / st[a] >= od[a]) {
          ok = 0;
          break;
        }
        o[a] = t / st[a];
      }
      if (ok) keys[nk++] = lin4(b, o[0], o[1], o[2], od[0], od[1], od[2]);
    }
  }
  qsort(keys, nk, sizeof(int64_t), cmp_i64);
  int Mout = 0;
  for (size_t j = 0; j < nk; ++j)
    if (j == 0 || keys[j] != keys[j - 1]) {
      int64_t l = keys[j];
      out_coords[Mout * 4 + 3] = (int)(l % od[2]);
      l /= od[2];
      out_coords[Mout * 4 + 2] = (int)(l % od[1]);
      l /= od[1];
      out_coords[Mout * 4 + 1] = (int)(l % od[0]);
      out_coords[Mout * 4 + 0] = (int)(l / od[0]);
      ++Mout;
    }
  free(keys);
  if (out_nbr) {
    omap_t map;
    omap_init(&map, (size_t)Min + 1);
    int found;
    for (int i = 0; i < Min; ++i)
      *omap_slot(&map, lin4(in_coords[i * 4], in_coords[i * 4 + 1], in_coords[i * 4 + 2],
                            in_coords[i * 4 + 3], Dz, Hy, Wx), &found) = i;
    if (in_nbr)
      for (size_t j = 0; j < (size_t)Min * K; ++j) in_nbr[j] = -1;
    for (int o = 0; o < Mout; ++o)
      for (int k = 0; k < K; ++k) {
        const int d[3] = {k / (ks[1] * ks[2]), (k / ks[2]) % ks[1], k % ks[2]};
        int c[3], ok = 1;
        for (int a = 0; a < 3; ++a) {
          c[a] = out_coords[o * 4 + 1 + a] * st[a] - pd[a] + d[a];
          if (c[a] < 0 || c[a] >= id[a]) ok = 0;
        }
        int r = -1;
        if (ok) {
          const int64_t key = lin4(out_coords[o * 4], c[0], c[1], c[2], Dz, Hy, Wx);
          size_t h = ((uint64_t)key * 0x9E3779B97F4A7C15ull) >> 20 & map.mask;
          while (map.keys[h] != -1 && map.keys[h] != key) h = (h + 1) & map.mask;
          if (map.keys[h] == key) r = map.vals[h];
        }
        out_nbr[(size_t)o * K + k] = r;
        if (r >= 0 && in_nbr) in_nbr[(size_t)r * K + k] = o;
      }
    omap_free(&map);
  }
  return Mout;
}

/* out[o][n] = bias[n] + sum_k sum_c in[nbr[o][k']][c] * W[n*sn + k*sk + c*sc], k' = mirror ? K-1-k : k */
void oracle_spconv_conv(const float* in, const int32_t* nbr, const float* W, int64_t sn,
                        int64_t sk, int64_t sc, int mirror, const float* bias, float* out, int Mout,
                        int K, int Cin, int Cout) {
  for (int o = 0; o < Mout; ++o)
    for (int n = 0; n < Cout; ++n) {
      double acc = 0.0; /* double accumulator: the oracle is the "true" value, kernels get a tolerance */
      for (int k = 0; k < K; ++k) {
        const int r = nbr[(size_t)o * K + (mirror ? K - 1 - k : k)];
        if (r < 0) continue;
        for (int c = 0; c < Cin; ++c)
          acc += (double)in[(size_t)r * Cin + c] * (double)W[n * sn + k * sk + c * sc];
      }
      out[(size_t)o * Cout + n] = (float)(acc + (bias ? (double)bias[n] : 0.0));
    }
}

/* gW[n][k][c] = sum_o gout[o][n] * in[nbr[o][k]][c] */
void oracle_spconv_wgrad(const float* in, const int32_t* nbr, const float* gout, float* gW,
                         int Mout, int K, int Cin, int Cout) {
  double* acc = (double*)calloc((size_t)Cout * K * Cin, sizeof(double));
  for (int o = 0; o < Mout; ++o)
    for (int k = 0; k < K; ++k) {
      const int r = nbr[(size_t)o * K + k];
      if (r < 0) continue;
      for (int n = 0; n < Cout; ++n) {
        const double g = gout[(size_t)o * Cout + n];
        for (int c = 0; c < Cin; ++c)
          acc[((size_t)n * K + k) * Cin + c] += g * (double)in[(size_t)r * Cin + c];
      }
    }
  for (size_t j = 0; j < (size_t)Cout * K * Cin; ++j) gW[j] = (float)acc[j];
  free(acc);
}

/* SparseConvTensor.dense() (height_compression.py:19): [B, C, Dz, Hy, Wx] */
void oracle_sparse_to_dense(const float* feat, const int32_t* coords, int M, int C, int B, int Dz,
                            int Hy, int Wx, float* dense) {
  memset(dense, 0, sizeof(float) * (size_t)B * C * Dz * Hy * Wx);
  for (int r = 0; r < M; ++r)
    for (int c = 0; c < C; ++c)
      dense[((((size_t)coords[r * 4] * C + c) * Dz + coords[r * 4 + 1]) * Hy + coords[r * 4 + 2]) * Wx +
            coords[r * 4 + 3]] = feat[(size_t)r * C + c];
}

/* ---- rotated-BEV IoU + greedy NMS (test infrastructure) ------------------------------------------------
 * CPU restatement of `iou3d_nms_cuda.nms_gpu` as called by the reference
 * (unidistill/layers/head/det3d/generate_proposals/centerpoint_gen_proposals.py:85-105).  The extension
 * binary is absent from the reference tree (OpenPCDet iou3d_nms lineage): PARITY UNPINNED vs the binary;
 * the published behaviour is restated -- boxes (x,y,z,dx,dy,dz,heading) sorted by descending score, box i
 * suppresses later boxes with BEV IoU > thresh unless suppressed itself.  Same float operations as the
 * HIP kernel (Sutherland-Hodgman clip + shoelace); oracle_iou_bev_f64 is an independent double-precision
 * formulation used to check the restatement itself. */
typedef struct { float x, y; } op2;

static void o_corners(const float* b, op2* c) {
  const float cs = cosf(b[6]), sn = sinf(b[6]);
  const float hx = 0.5f * b[3], hy = 0.5f * b[4];
  const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
  for (int k = 0; k < 4; ++k) {
    c[k].x = b[0] + lx[k] * cs - ly[k] * sn;
    c[k].y = b[1] + lx[k] * sn + ly[k] * cs;
  }
}

static float o_clipped_area(const op2* A, const op2* B) {
  op2 poly[10], tmp[10];
  int n = 4;
  for (int k = 0; k < 4; ++k) poly[k] = A[k];
  for (int e = 0; e < 4 && n > 0; ++e) {
    const op2 p = B[e], q = B[(e + 1) & 3];
    const float ex = q.x - p.x, ey = q.y - p.y;
    int m = 0;
    for (int k = 0; k < n; ++k) {
      const op2 s = poly[k], t = poly[(k + 1 == n) ? 0 : k + 1];
      const float ds = ex * (s.y - p.y) - ey * (s.x - p.x);
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
    const op2 s = poly[k], t = poly[(k + 1 == n) ? 0 : k + 1];
    a2 += s.x * t.y - t.x * s.y;
  }
  return 0.5f * fabsf(a2);
}

float oracle_iou_bev(const float* a, const float* b) {
  op2 ca[4], cb[4];
  o_corners(a, ca);
  o_corners(b, cb);
  const float inter = o_clipped_area(ca, cb);
  const float sa = a[3] * a[4], sb = b[3] * b[4];
  const float den = sa + sb - inter;
  return inter / (den > 1e-8f ? den : 1e-8f);
}

void oracle_iou_bev_matrix(const float* a, int Na, const float* b, int Nb, float* iou) {
  for (int i = 0; i < Na; ++i)
    for (int j = 0; j < Nb; ++j) iou[(size_t)i * Nb + j] = oracle_iou_bev(a + (size_t)i * 7, b + (size_t)j * 7);
}

/* keep[0..ret) = kept indices in order */
int oracle_nms_bev(const float* boxes, int N, float thresh, int64_t* keep) {
  char* removed = (char*)calloc((size_t)(N > 0 ? N : 1), 1);
  int count = 0;
  for (int i = 0; i < N; ++i) {
    if (removed[i]) continue;
    keep[count++] = i;
    for (int j = i + 1; j < N; ++j)
      if (!removed[j] && oracle_iou_bev(boxes + (size_t)i * 7, boxes + (size_t)j * 7) > thresh) removed[j] = 1;
  }
  free(removed);
  return count;
}

/* Independent check of the restatement: overlap area by integrating, over x, the length of the
 * intersection of the two rectangles' vertical chords (double precision, adaptive only in that the
 * integration breakpoints are all corner / edge-crossing abscissae, between which the chord overlap is
 * piecewise linear or zero -> exact with Simpson on each piece up to rounding). */
static void o_chord(const double* c, double x, double* lo, double* hi) {   /* c: 4 corners (x,y) CCW */
  double ylo = 1e300, yhi = -1e300;
  for (int k = 0; k < 4; ++k) {
    const double x0 = c[2 * k], y0 = c[2 * k + 1], x1 = c[2 * ((k + 1) & 3)], y1 = c[2 * ((k + 1) & 3) + 1];
    const double a = x0 < x1 ? x0 : x1, b = x0 < x1 ? x1 : x0;
    if (x < a || x > b) continue;
    if (x1 == x0) {
      if (y0 < ylo) ylo = y0; if (y0 > yhi) yhi = y0;
      if (y1 < ylo) ylo = y1; if (y1 > yhi) yhi = y1;
    } else {
      const double y = y0 + (y1 - y0) * (x - x0) / (x1 - x0);
      if (y < ylo) ylo = y; if (y > yhi) yhi = y;
    }
  }
  *lo = ylo; *hi = yhi;
}

static double o_overlap_len(const double* ca, const double* cb, double x) {
  double a0, a1, b0, b1;
  o_chord(ca, x, &a0, &a1);
  o_chord(cb, x, &b0, &b1);
  if (a0 > a1 || b0 > b1) return 0.0;
  const double lo = a0 > b0 ? a0 : b0, hi = a1 < b1 ? a1 : b1;
  return hi > lo ? hi - lo : 0.0;
}

static int o_cmp_double(const void* p, const void* q) {
  const double a = *(const double*)p, b = *(const double*)q;
  return (a > b) - (a < b);
}

double oracle_iou_bev_f64(const float* a, const float* b) {
  double ca[8], cb[8];
  const float* bx[2] = {a, b};
  double* cc[2] = {ca, cb};
  for (int s = 0; s < 2; ++s) {
    const double cs = cos((double)bx[s][6]), sn = sin((double)bx[s][6]);
    const double hx = 0.5 * bx[s][3], hy = 0.5 * bx[s][4];
    const double lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
    for (int k = 0; k < 4; ++k) {
      cc[s][2 * k] = bx[s][0] + lx[k] * cs - ly[k] * sn;
      cc[s][2 * k + 1] = bx[s][1] + lx[k] * sn + ly[k] * cs;
    }
  }
  /* breakpoints: all corner abscissae + abscissae of edge-edge crossings */
  double xs[8 + 16];
  int n = 0;
  for (int k = 0; k < 4; ++k) { xs[n++] = ca[2 * k]; xs[n++] = cb[2 * k]; }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      const double x1 = ca[2 * i], y1 = ca[2 * i + 1], x2 = ca[2 * ((i + 1) & 3)], y2 = ca[2 * ((i + 1) & 3) + 1];
      const double x3 = cb[2 * j], y3 = cb[2 * j + 1], x4 = cb[2 * ((j + 1) & 3)], y4 = cb[2 * ((j + 1) & 3) + 1];
      const double den = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
      if (fabs(den) < 1e-14) continue;
      const double t = ((x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)) / den;
      const double u = ((x1 - x3) * (y1 - y2) - (y1 - y3) * (x1 - x2)) / den;
      if (t >= 0 && t <= 1 && u >= 0 && u <= 1) xs[n++] = x1 + t * (x2 - x1);
    }
  qsort(xs, (size_t)n, sizeof(double), o_cmp_double);
  double inter = 0.0;
  for (int k = 0; k + 1 < n; ++k) {
    const double x0 = xs[k], x1 = xs[k + 1];
    if (x1 - x0 < 1e-15) continue;
    const double e = (x1 - x0) * 1e-9;
    const double f0 = o_overlap_len(ca, cb, x0 + e), f1 = o_overlap_len(ca, cb, x1 - e);
    const double fm = o_overlap_len(ca, cb, 0.5 * (x0 + x1));
    inter += (x1 - x0) * (f0 + 4.0 * fm + f1) / 6.0;
  }
  const double sa = (double)a[3] * a[4], sb = (double)b[3] * b[4];
  const double den = sa + sb - inter;
  return inter / (den > 1e-8 ? den : 1e-8);
}
