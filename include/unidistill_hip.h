/*
 * unidistill_hip.h -- C ABI of libunidistill_hip.so (MI355X / gfx950).
 *
 * One shared library sits underneath the Python names the UniDistill reference
 * imports for its BEV feature-extraction + distillation hot path.  Every entry
 * point takes plain device pointers + sizes + a hipStream_t, never a torch type,
 * returns 0 on success or a negative UD_ERR_* code, and never throws.
 * All pointers are DEVICE pointers unless a parameter is documented "host".
 * Kernels are enqueued on `stream`; nothing in here synchronises the device.
 *
 * Reference interface each group replaces (paths relative to the reference tree,
 * /root/reference/unidistill/...):
 *
 *   ud_bev_pool_*      voxel_pooling_ext.voxel_pooling_forward_wrapper
 *                      layers/blocks_3d/mmdet3d/lss_fpn.py:48-59 (fwd), :64-79 (bwd)
 *   ud_lss_*           LSSFPN.get_geometry / binning / lift
 *                      layers/blocks_3d/mmdet3d/lss_fpn.py:200-240, :289-316
 *   ud_voxelize*       spconv.pytorch.utils.PointToVoxel.__call__ + MeanVFE.forward
 *                      data/det3d/preprocess/voxelization.py:31-38,54
 *                      layers/blocks_3d/det3d/vfe/mean_vfe.py:14-34
 *   ud_spconv_*        spconv.pytorch.{SubMConv3d,SparseConv3d,SparseConvTensor.dense}
 *                      layers/blocks_3d/det3d/spconv_backbone.py:21-48,71-92,259-340
 *                      layers/blocks_2d/det3d/map_to_bev/height_compression.py:19-22
 *   ud_distill_*       FeatureDistillLoss / BEVDistillLoss / ResponseDistillLoss
 *                      exps/.../BEVFusion_nuscenes_centerhead_camera_exp_distill_lidar.py
 *                      :196-245, :248-323, :326-385, gaussian mask :100-178
 */
#ifndef UNIDISTILL_HIP_H
#define UNIDISTILL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* hipStream_t is an opaque pointer; spelled void* so this header needs no HIP include. */
typedef void* ud_stream_t;

#define UD_OK 0
#define UD_ERR_INVALID_ARG (-1)
#define UD_ERR_WORKSPACE (-2) /* workspace pointer NULL or too small */
#define UD_ERR_HIP (-3)       /* a HIP runtime call / launch failed    */
#define UD_ERR_UNSUPPORTED (-4)

/* Library identity: returns "unidistill_hip <abi-version> gfx950". */
const char* ud_version(void);
/* ABI version integer; bumped whenever a signature in this header changes. */
int ud_abi_version(void);
/* Human-readable text for a UD_ERR_* code. */
const char* ud_error_string(int code);

/*
 * Optional profiler: when enabled, the library brackets its dominant kernels with hipEvents on
 * the launch stream.  ud_prof_read(name) waits for those events and returns the summed kernel
 * time in ms and the number of launches (names: "bev_pool.k_pool", "bev_pool.k_bwd", ...).
 */
void ud_prof_enable(int on);
int ud_prof_read(const char* name, double* total_ms, int* calls, int reset);

/* HBM calibration: mode 0 = streaming read of src (dst only keeps the loads alive), mode 1 = copy
 * src -> dst; n_floats elements.  Timed through ud_prof ("bench.stream_read" / "bench.stream_copy"). */
int ud_bench_stream(const float* src, float* dst, size_t n_floats, int mode, ud_stream_t stream);

/* ------------------------------------------------------------------------- */
/* BEV pool (camera LSS splat)                                               */
/* ------------------------------------------------------------------------- */

/* flags for ud_bev_pool_fwd */
#define UD_POOL_ACCUMULATE 0u /* out += pooled sums; caller pre-zeroed out (reference contract, lss_fpn.py:43-47) */
#define UD_POOL_OVERWRITE 1u  /* every cell of out is written (empty cells get 0); no pre-zero needed             */

/* Bytes of scratch ud_bev_pool_fwd needs for these sizes (0 on invalid sizes). */
size_t ud_bev_pool_workspace_bytes(int B, int N, int C, int nx, int ny, int nz);

/*
 * out[b, y, x, :] (+)= sum over points n of batch b with geom[b,n]=(x,y,z) inside
 * 0<=x<nx, 0<=y<ny, 0<=z<nz of feat[b,n,:];  pos[b,n,:] = (b,y,x) for kept points,
 * (-1,-1,-1) otherwise.  Sums run in ascending point index n (deterministic; the
 * reference CUDA op used unordered atomicAdd).
 *   geom  i32[B,N,3]   feat f32[B,N,C]   out f32[B,ny,nx,C]   pos i32[B,N,3]
 */
int ud_bev_pool_fwd(const int32_t* geom, const float* feat, float* out, int32_t* pos,
                    int B, int N, int C, int nx, int ny, int nz, unsigned flags,
                    void* workspace, size_t workspace_bytes, ud_stream_t stream);

/*
 * gfeat[b,n,:] = gout[b,:,y,x] for pos[b,n]=(b,y,x) != -1, else 0   (lss_fpn.py:64-79).
 * gout is addressed through element strides (sb, sc, sy, sx) so both the NCHW grad the
 * reference produces and an NHWC (channels-last) grad are accepted; NHWC (sc == 1) is the
 * fast path, anything else is transposed into `workspace` first
 * (ud_bev_pool_bwd_workspace_bytes).
 */
size_t ud_bev_pool_bwd_workspace_bytes(int B, int C, int nx, int ny, int64_t sc);
int ud_bev_pool_bwd(const float* gout, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                    const int32_t* pos, float* gfeat, int B, int N, int C, int nx, int ny,
                    void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Camera lift-splat (LSS): geometry, depth softmax, lift, fused lift+splat  */
/* ------------------------------------------------------------------------- */

/*
 * mats[b*ncam+cam] = { inverse(ida), sensor2ego @ inverse(intrin), bda }  f32[B*ncam,3,16]
 * (lss_fpn.py:221-222,233,235-239).  sensor2ego/intrin/ida f32[B,ncam,4,4]; bda f32[B,4,4] or NULL.
 * ida_inv / intrin_inv f32[B,ncam,4,4]: the fp32 inverses as the caller's torch.inverse produced
 * them (the reference's own call; its rounding depends on the LAPACK/solver backend), or NULL for
 * the correctly rounded inverse computed in fp64 in the kernel.  Every product is evaluated as the
 * reference's CPU path does (acc = 0; acc += a[k]*b[k], fp32, no FMA), so with the reference's
 * inverses the ego coordinates and bins of ud_lss_geometry are bit-identical to its CPU output.
 */
int ud_lss_prepare_mats(const float* sensor2ego, const float* intrin, const float* ida,
                        const float* bda, const float* ida_inv, const float* intrin_inv, int B,
                        int ncam, float* mats, ud_stream_t stream);

/*
 * LSSFPN.get_geometry + binning (lss_fpn.py:200-240, :311-313) for every frustum point
 * (b, cam, d, h, w): geom f32[B,ncam,D,fH,fW,3] (optional, NULL to skip) and
 * bins i32[B,ncam,D,fH,fW,3] = ((geom - lo) / size).int()  with host lo/size f32[3]
 * (lo = voxel_coord - voxel_size/2 evaluated in fp32 by the caller).  frustum_u[fW], frustum_v[fH],
 * frustum_d[D] are the device vectors of create_frustum (lss_fpn.py:173-198).
 */
int ud_lss_geometry(const float* mats, const float* frustum_u, const float* frustum_v,
                    const float* frustum_d, int B, int ncam, int D, int fH, int fW,
                    const float* lo, const float* size, int has_bda, float* geom, int32_t* bins,
                    ud_stream_t stream);

/*
 * depth_feature f32[BN, D+C, fH, fW] addressed by element strides (sn, sc, sh, sw) ->
 *   prob   f32[BN, D, fH*fW]   softmax over the first D channels (lss_fpn.py:289)
 *   ctx_pm f32[BN, fH*fW, C]   context channels D..D+C, pixel-major
 */
int ud_lss_depth_ctx(const float* depth_feature, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                     int BN, int D, int C, int fH, int fW, float* prob, float* ctx_pm,
                     ud_stream_t stream);

/* Materialised lift (reference boundary, lss_fpn.py:290-310):
 * lifted f32[BN, D, fH*fW, C] = prob (x) ctx  == img_feat_with_depth.permute(0,1,3,4,5,2). */
int ud_lss_lift_fwd(const float* prob, const float* ctx_pm, float* lifted, int BN, int D, int C,
                    int fH, int fW, ud_stream_t stream);

/*
 * Fused lift + splat forward (a7-a9): out[b,y,x,:] = sum over points of prob[point] *
 * ctx_pm[pixel(point), :], in ascending point order; the [B,N,C] tensor is never formed.
 * geom i32[B,N,3] with N = ncam*D*fH*fW; out f32[B,ny,nx,C] (every cell written);
 * pos i32[B,N,3] as ud_bev_pool_fwd.  Workspace: ud_bev_pool_workspace_bytes(B,N,C,nx,ny,nz).
 */
int ud_lss_splat_fwd(const int32_t* geom, const float* prob, const float* ctx_pm, float* out,
                     int32_t* pos, int B, int ncam, int D, int fH, int fW, int C, int nx, int ny,
                     int nz, void* workspace, size_t workspace_bytes, ud_stream_t stream);

/*
 * ud_lss_splat_fwd with the binning computed from the frustum inside the list-building kernel (the arithmetic of
 * ud_lss_geometry: mats / frustum axes / lo / size / has_bda as there): what LSSFPN._forward_single_sweep
 * (lss_fpn.py:277-319) needs from get_geometry is only the bins, so the training step never materialises them.
 * out, pos and the workspace as ud_lss_splat_fwd; bit-identical to ud_lss_geometry followed by ud_lss_splat_fwd.
 */
int ud_lss_splat_geom_fwd(const float* mats, const float* frustum_u, const float* frustum_v,
                          const float* frustum_d, const float* lo, const float* size, int has_bda,
                          const float* prob, const float* ctx_pm, float* out, int32_t* pos, int B, int ncam,
                          int D, int fH, int fW, int C, int nx, int ny, int nz, void* workspace,
                          size_t workspace_bytes, ud_stream_t stream);

/*
 * Backward of softmax (x) context.  pos == NULL: gsrc is the dense grad of the materialised
 * lifted tensor f32[BN, D, fH*fW, C].  pos != NULL (fused splat backward): gsrc is the dense NHWC
 * BEV grad f32[B, ny, nx, C] and each point gathers its cell's row through pos i32[B*N,3].
 * Writes g_depth_feature f32[BN, D+C, fH, fW] through element strides (sn, sc, sh, sw).
 */
size_t ud_lss_lift_bwd_workspace_bytes(int BN, int D, int C, int fH, int fW);
int ud_lss_lift_bwd(const float* gsrc, const int32_t* pos, const float* prob, const float* ctx_pm,
                    float* g_depth_feature, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int BN,
                    int ncam, int D, int C, int fH, int fW, int nx, int ny, void* workspace,
                    size_t workspace_bytes, ud_stream_t stream);

/* ------------------------------------------------------------------------- */
/* LiDAR voxelization (+ fused MeanVFE)                                      */
/* ------------------------------------------------------------------------- */

/*
 * Replaces spconv.pytorch.utils.PointToVoxel.__call__ (voxelization.py:31-38,54) for a whole
 * collated batch and, optionally fused, MeanVFE.forward (mean_vfe.py:14-34).
 *   points      f32[B,N,F]  (F >= 3: x,y,z first; clouds padded to a common N like collate_fn)
 *   voxel_size  host f32[3] (x,y,z);  range host f32[6] (xmin,ymin,zmin,xmax,ymax,zmax)
 *   P           max points kept per voxel;  max_voxels  cap PER SAMPLE
 * Outputs hold ud_voxelize_capacity(B,N,max_voxels) rows; rows [0, m_out[B]) are valid:
 *   voxels      f32[cap,P,F] zero padded, or NULL to skip materialising it (fused path)
 *   coords      i32[cap,4]  (b, z, y, x)
 *   num_points  i32[cap] or NULL;  mean_feats f32[cap,F] or NULL (sum over slots / max(num,1))
 *   m_out       i32[B+2]    voxels per sample, then the total, then the overflow word (device memory)
 *   algo        0: hash partition + per-partition LDS sort, no global atomics (the fast path; if one partition
 *                  receives more than 3 072 points -- thousands of points in a single voxel -- it sets the overflow
 *                  word m_out[B+1] = 1, the outputs are then invalid and the call must be repeated with algo 1);
 *               1: open-addressing hash with device-scope atomics (any input).  Both give the same bits.
 * Deterministic: voxel order = first appearance in the point list, kept points = the first P in
 * input order, no voxel created beyond max_voxels per sample (oracle/ud_oracle.c:oracle_voxelize).
 */
size_t ud_voxelize_workspace_bytes(int B, int N, int P, int max_voxels);
int ud_voxelize_capacity(int B, int N, int max_voxels);
int ud_voxelize(const float* points, int B, int N, int F, const float* voxel_size,
                const float* range, int P, int max_voxels, float* voxels, int32_t* coords,
                int32_t* num_points, float* mean_feats, int32_t* m_out, void* workspace,
                size_t workspace_bytes, int algo, ud_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Sparse 3-D convolution (spconv boundary) + densify                        */
/* ------------------------------------------------------------------------- */

/*
 * Site index of one level: a rank bitmap over the (B, Dz, Hy, Wx) voxel grid (spconv_index.hip).
 * coords are i32[M,4] = (b, z, y, x) like SparseConvTensor.indices (spconv_backbone.py:354-359).
 * rows_sorted != 0 promises that row i is the i-th site in ascending (b,z,y,x) order (true for
 * the outputs of ud_spconv_down_outputs); otherwise a rank->row permutation is built.
 */
size_t ud_spconv_index_bytes(int B, int Dz, int Hy, int Wx, int M);
int ud_spconv_build_index(const int32_t* coords, int M, int B, int Dz, int Hy, int Wx,
                          int rows_sorted, void* index, size_t index_bytes, ud_stream_t stream);

/* SubMConv3d rulebook (odd kernel, implicit padding k/2): nbr i32[M, kz*ky*kx] = row of the
 * active site at coord + (k - centre) or -1.  Offsets enumerate z slowest, x fastest. */
int ud_spconv_subm_rulebook(const void* index, int rows_sorted, const int32_t* coords, int M,
                            int B, int Dz, int Hy, int Wx, int kz, int ky, int kx, int32_t* nbr,
                            ud_stream_t stream);

/*
 * SparseConv3d(kernel, stride, padding) output sites = every cell reachable from an active
 * input; ksize/stride/pad are host int[3] in (z, y, x) order; the output grid is
 * floor((in + 2p - k)/s) + 1 per axis.  Builds the OUTPUT level's index (rows sorted), writes
 * out_coords i32[out_cap,4] in ascending (b,z,y,x) order and the count into m_out (device int).
 */
int ud_spconv_down_outputs(const int32_t* in_coords, int Min, int B, int Dz, int Hy, int Wx,
                           const int* ksize, const int* stride, const int* pad, void* out_index,
                           size_t out_index_bytes, int32_t* out_coords, int out_cap,
                           int32_t* m_out, ud_stream_t stream);
/* Device-count variants: the row count of the input site set is read from device memory (the launches cover the upper
 * bound M_cap / Min_cap), so ud_voxelize -> level index -> the four down-sampling levels of VoxelResBackBone8x
 * (spconv_backbone.py:279-340) chain without a host read in between; ONE read of all counts sizes the tensors afterwards. */
int ud_spconv_build_index_dev(const int32_t* coords, const int32_t* m_dev, int M_cap, int B, int Dz, int Hy, int Wx,
                              int rows_sorted, void* index, size_t index_bytes, ud_stream_t stream);
int ud_spconv_down_outputs_dev(const int32_t* in_coords, const int32_t* min_dev, int Min_cap, int B, int Dz, int Hy,
                               int Wx, const int* ksize, const int* stride, const int* pad, void* out_index,
                               size_t out_index_bytes, int32_t* out_coords, int out_cap, int32_t* m_out,
                               ud_stream_t stream);

/* out_nbr i32[Mout,K]: input row at o*s - p + k or -1; in_nbr i32[Min,K] (optional, for dgrad):
 * output row that input i feeds through offset k, or -1. */
int ud_spconv_down_rulebook(const void* in_index, int in_rows_sorted, int Min, int B, int Dz,
                            int Hy, int Wx, const int* ksize, const int* stride, const int* pad,
                            const int32_t* out_coords, int Mout, int32_t* out_nbr,
                            int32_t* in_nbr, ud_stream_t stream);

/*
 * out f32[Mout,Cout] = bias + sum_k sum_c in[nbr[o][k], c] * W[n*w_sn + k*w_sk + c*w_sc].
 * Forward with spconv-2.x KRSC weights [Cout,K,Cin]: (w_sn, w_sk, w_sc) = (K*Cin, Cin, 1).
 * Input gradient: in := gout, nbr := in_nbr (strided conv) or the subm rulebook with mirror = 1,
 * (w_sn, w_sk, w_sc) = (1, Cin, K*Cin), Cin/Cout swapped.  Exact-fp32 MFMA, deterministic.
 * algo 0 = auto (128-row MFMA kernel), 1 = generic VALU kernel (any channel counts),
 * 2 = first-generation 64-row MFMA kernel, 3 = bf16-input MFMA with fp32 accumulation
 * (mixed-precision mode: operands are rounded to bf16 in LDS; tensors in HBM stay fp32).
 * Fused epilogue (MFMA kernel, algo 0): y = relu?((conv + bias) * ep_scale + ep_shift + ep_residual)
 * with ep_scale/ep_shift f32[Cout] (a folded eval-mode BatchNorm1d; both or neither),
 * ep_residual f32[Mout,Cout] (a SparseBasicBlock skip), ep_relu 0/1; pass NULL/0 to disable.
 * row_order (optional, may be NULL): a permutation i32[Mout] of the output rows; tiles are formed
 * over row_order so rows with similar neighbour masks share a tile (same results, fewer active
 * kernel offsets per tile; only the wide-channel MFMA kernel uses it).
 * Size limit: `in` must be smaller than 4 GiB - 64 KiB (0xFFFF0000 bytes): the 64 / 128-channel fp32 kernel addresses the gathered
 * rows with 32-bit byte offsets through a buffer descriptor (the entry point is not told the input's row count).
 */
int ud_spconv_conv(const float* in, const int32_t* nbr, const float* W, int64_t w_sn, int64_t w_sk,
                   int64_t w_sc, int mirror, const float* bias, float* out, int Mout, int K,
                   int Cin, int Cout, int algo, const int32_t* row_order, const float* ep_scale,
                   const float* ep_shift, const float* ep_residual, int ep_relu, ud_stream_t stream);

/* gW f32[Cout,K,Cin] = sum_o gout[o,n] * in[nbr[o][k], c]  (ordered partial sums, deterministic). */
/* Mixed-precision weight gradient (training under bf16 autocast): operands rounded to bf16 when the
 * 64-row tiles are staged, fp32 accumulation on v_mfma_f32_16x16x32_bf16, ordered reduction of the row
 * chunks (deterministic).  row_order (optional) is the forward kernel's row permutation; tiles that
 * have no pair for an offset are skipped.  K <= 32.  gW f32[Cout,K,Cin].  io_bf16 bit 0: `in` and `gout`
 * already hold bf16 rows (Cin, Cout % 8 == 0); otherwise they are fp32.  Bit 1 (with bit 0, row_order and
 * tile_masks; Cin, Cout in {32, 64, 128}, else UD_ERR_UNSUPPORTED): `nbr` and `tile_masks` are ALREADY permuted
 * into row_order and row_order only locates the gout rows -- spares the caller a gather pass over gout. */
size_t ud_spconv_wgrad_bf16_workspace_bytes(int Mout, int K, int Cin, int Cout);
int ud_spconv_wgrad_bf16(const void* in, const int32_t* nbr, const void* gout, float* gW, int Mout,
                         int K, int Cin, int Cout, int io_bf16, const int32_t* row_order,
                         const unsigned* tile_masks, void* workspace, size_t workspace_bytes,
                         ud_stream_t stream);
/* tile_masks (optional, u32[ceil(Mout/64)]): bit k set iff some row of the 64-row tile (in row_order) has
 * a pair at offset k; computed internally when NULL.  It depends on the rulebook only. */
int ud_spconv_tile_masks(const int32_t* nbr, int Mout, int K, const int32_t* row_order, unsigned* masks,
                         ud_stream_t stream);
/* Row order of a rulebook for the kernels above (`row_order`): rows sorted (stable) by their neighbour bit mask, the bit of
 * offset k weighted by its rarity among ~4 096 sampled rows (rarest on top).  order: i32[Mout].  K <= 31. */
size_t ud_spconv_mask_order_workspace_bytes(int Mout, int K);
int ud_spconv_mask_order(const int32_t* nbr, int Mout, int K, int32_t* order, void* workspace, size_t workspace_bytes,
                         ud_stream_t stream);
/* Mixed-precision inference variant: the bf16-operand MFMA kernel (algo 3 above) with bf16 tensors in
 * HBM so the gather moves half the bytes.  io_flags bit 0: `in` is bf16 [*, Cin] (Cin % 4 == 0);
 * bit 1: `out` and `ep_residual` are bf16 [Mout, Cout]; bit 2: `W` is bf16 (w_sc == 1).  K <= 32. */
int ud_spconv_conv_bf16io(const void* in, const int32_t* nbr, const void* W, int64_t w_sn,
                          int64_t w_sk, int64_t w_sc, int mirror, const float* bias, void* out,
                          int Mout, int K, int Cin, int Cout, int io_flags, const int32_t* row_order,
                          const float* ep_scale, const float* ep_shift, const void* ep_residual,
                          int ep_relu, ud_stream_t stream);
size_t ud_spconv_wgrad_workspace_bytes(int Mout, int K, int Cin, int Cout);
int ud_spconv_wgrad(const float* in, const int32_t* nbr, const float* gout, float* gW, int Mout,
                    int K, int Cin, int Cout, int algo, void* workspace, size_t workspace_bytes,
                    ud_stream_t stream);

/* SparseConvTensor.dense() (height_compression.py:19): dense f32[B,C,Dz,Hy,Wx], every element written
 * once (feature or zero; the caller need not clear it), and its backward gather
 * gfeat[row,:] = gdense[b,:,z,y,x].  workspace: ud_sparse_bev_workspace_bytes(B,Dz,Hy,Wx) (cell -> row map).
 * B*Dz*Hy <= 65535. */
int ud_sparse_to_dense(const float* feat, const int32_t* coords, int M, int C, int B, int Dz,
                       int Hy, int Wx, float* dense, void* workspace, size_t workspace_bytes,
                       ud_stream_t stream);
int ud_dense_to_sparse(const float* gdense, const int32_t* coords, int M, int C, int B, int Dz,
                       int Hy, int Wx, float* gfeat, void* workspace, size_t workspace_bytes,
                       ud_stream_t stream);
/* HeightCompression in the mixed-precision path (reference layers/blocks_2d/det3d/map_to_bev/
 * height_compression.py:19-22: dense() then view(N, C*D, H, W)): bev bf16[B,Hy,Wx,C*Dz] (channels-last,
 * channel = c*Dz + z) straight from feat bf16[M,C], zeros where no voxel; and its backward. (C*Dz) % 4 == 0. */
size_t ud_sparse_bev_workspace_bytes(int B, int Dz, int Hy, int Wx);
int ud_sparse_to_bev_bf16(const void* feat, const int32_t* coords, int M, int C, int B, int Dz, int Hy,
                          int Wx, void* bev, void* workspace, size_t workspace_bytes, ud_stream_t stream);
int ud_bev_to_sparse_bf16(const void* gbev, const int32_t* coords, int M, int C, int B, int Dz, int Hy,
                          int Wx, void* gfeat, void* workspace, size_t workspace_bytes, ud_stream_t stream);
/* fp32 twin (the reference's arithmetic): the channels-last fp32 BEV map in one pass instead of dense() + an NCHW -> NHWC copy. */
int ud_sparse_to_bev_f32(const float* feat, const int32_t* coords, int M, int C, int B, int Dz, int Hy, int Wx, float* bev,
                         void* workspace, size_t workspace_bytes, ud_stream_t stream);
int ud_bev_to_sparse_f32(const float* gbev, const int32_t* coords, int M, int C, int B, int Dz, int Hy, int Wx, float* gfeat,
                         void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Distillation losses (feature / relation / response) + gaussian box mask   */
/* ------------------------------------------------------------------------- */

/*
 * training_step's box prep on the device (distill_lidar.py:449-455, :466-483, :73-97):
 * gt f32[B,M,S>=7] (x,y,z,dx,dy,dz,yaw,...) -> corners_px f32[B,M,4,2] = rotated BEV corners in
 * BEV pixels ((corner - pc_min) / pixel), valid u8[B,M] = 1 for rows up to the last non-zero row.
 */
int ud_distill_box_corners(const float* gt, int B, int M, int S, double pc_min_x, double pc_min_y,
                           double pixel_x, double pixel_y, float* corners_px, unsigned char* valid,
                           ud_stream_t stream);

/*
 * kind 0 = FeatureDistillLoss (distill_lidar.py:196-245), kind 1 = BEVDistillLoss (:248-323).
 * s / t: student / teacher BEV maps [B,C,H,W] addressed through host element strides
 * int64[4] = (sb, sc, sy, sx).  Forward writes box_loss f32[B,M] (per-box loss, 0 for invalid boxes;
 * the scalar loss is sum(box_loss) / (reduce_mean(n_valid) + 1e-4)).  Backward writes
 * (*gscale) * d sum(box_loss)/ds into the pre-zeroed gs (gscale: DEVICE scalar): per-(box, key point, channel) gradients
 * into `workspace` (>= ud_distill_box_bwd_workspace_bytes), then a sorted, atomic-free scatter of the overlapping bilinear
 * footprints (fixed summation order: bitwise reproducible).  M <= 56.
 */
int ud_distill_box_fwd(int kind, const float* s, const int64_t* s_strides, const float* t,
                       const int64_t* t_strides, const float* corners_px,
                       const unsigned char* valid, int B, int M, int C, int H, int W,
                       float* box_loss, ud_stream_t stream);
size_t ud_distill_box_bwd_workspace_bytes(int B, int M, int C);
int ud_distill_box_bwd(int kind, const float* s, const int64_t* s_strides, const float* t,
                       const int64_t* t_strides, const float* corners_px,
                       const unsigned char* valid, int B, int M, int C, int H, int W,
                       const float* gscale, float* gs, const int64_t* gs_strides,
                       void* workspace, size_t workspace_bytes, ud_stream_t stream);
/* ud_distill_box_bwd ADDED to gs instead of written into a zeroed gs: gs already holds the gradient the feature map received from
 * its other consumer; only the pixels the boxes touch are read and written (deterministic).  Replaces the zero fill + autograd's
 * dense add of the two maps in the student's backward. */
int ud_distill_box_bwd_acc(int kind, const float* s, const int64_t* s_strides, const float* t,
                           const int64_t* t_strides, const float* corners_px,
                           const unsigned char* valid, int B, int M, int C, int H, int W,
                           const float* gscale, float* gs, const int64_t* gs_strides,
                           void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* calculate_box_mask_gaussian (distill_lidar.py:100-178) on the device: mask f32[B,H,W]. */
size_t ud_distill_mask_workspace_bytes(int B, int M);
int ud_distill_gaussian_mask(const float* gt, int B, int M, int S, double pc_min_x,
                             double pc_min_y, double pixel_x, double pixel_y, int H, int W,
                             float* mask, void* workspace, size_t workspace_bytes,
                             ud_stream_t stream);

/*
 * ResponseDistillLoss (distill_lidar.py:326-385).  HOST arrays of DEVICE pointers: n_hm heat-map
 * tensors [B,hm_ch[i],H,W] (student: probabilities as produced by the head's clamped sigmoid;
 * teacher: logits, turned into clamp(sigmoid(x/2), lo, hi)) and n_reg regression tensors
 * [B,reg_ch[i],H,W], all dense NCHW.  Forward: partial f32[B*ceil(H*W/256), 2] per-workgroup sums
 * of (|max_c s - max_c t| * mask, mean_c|s - t| * mask).  Backward writes every element of the
 * student-side grads; gscale_* are device scalars.
 */
int ud_distill_resp_fwd(const float* const* s_hm, const float* const* t_hm, const int* hm_ch,
                        int n_hm, const float* const* s_reg, const float* const* t_reg,
                        const int* reg_ch, int n_reg, const float* mask, int B, int H, int W,
                        float clamp_lo, float clamp_hi, float* partial, ud_stream_t stream);
int ud_distill_resp_bwd(const float* const* s_hm, const float* const* t_hm, float* const* g_hm,
                        const int* hm_ch, int n_hm, const float* const* s_reg,
                        const float* const* t_reg, float* const* g_reg, const int* reg_ch,
                        int n_reg, const float* mask, int B, int H, int W, float clamp_lo,
                        float clamp_hi, const float* gscale_cls, const float* gscale_reg,
                        ud_stream_t stream);
/* The same two passes with every tensor read / written in place through an (sb, sc, sp) element-stride triple (batch, channel,
 * linearised pixel; stride(H) == W * stride(W)): channels-last maps and channel slices of the packed CenterHead output need no dense
 * NCHW copies.  *_st: host int64[n][3]. */
int ud_distill_resp_fwd_strided(const float* const* s_hm, const int64_t* s_hm_st, const float* const* t_hm,
                                const int64_t* t_hm_st, const int* hm_ch, int n_hm, const float* const* s_reg,
                                const int64_t* s_reg_st, const float* const* t_reg, const int64_t* t_reg_st, const int* reg_ch,
                                int n_reg, const float* mask, int B, int H, int W, float clamp_lo, float clamp_hi, float* partial,
                                ud_stream_t stream);
int ud_distill_resp_bwd_strided(const float* const* s_hm, const int64_t* s_hm_st, const float* const* t_hm,
                                const int64_t* t_hm_st, float* const* g_hm, const int64_t* g_hm_st, const int* hm_ch, int n_hm,
                                const float* const* s_reg, const int64_t* s_reg_st, const float* const* t_reg,
                                const int64_t* t_reg_st, float* const* g_reg, const int64_t* g_reg_st, const int* reg_ch,
                                int n_reg, const float* mask, int B, int H, int W, float clamp_lo, float clamp_hi,
                                const float* gscale_cls, const float* gscale_reg, ud_stream_t stream);

/* ---- Detection-head tail: BatchNorm -> ReLU -> per-head 3x3 conv (64 -> k) --------------------------
 * Replaces modules 1..3 of every SepHead stack (reference unidistill/layers/head/det3d/
 * center_head.py:311-362: nn.Sequential(Conv 64->64, BatchNorm2d, ReLU, Conv 64->k)) for all
 * G = tasks x heads packed heads at once, on the hidden tensor produced by the packed first conv.
 *   y, dy : [B][H][W][G*64] bf16, channels-last.          z, dz : [B][G*kmax][H][W] fp32.
 *   w2, dw2 : [G][kmax][9][64] fp32 (tap = ky*3 + kx), rows j >= a head's real k are zero.
 *   per-channel vectors (gamma, beta, mean, var, invstd, scale, shift, dgamma, dbeta): fp32 [G*64].
 * kmax <= 3 (UD_ERR_UNSUPPORTED otherwise).  BN + ReLU are applied while tiles are staged, the conv
 * multiplies bf16 operands with fp32 accumulation (same arithmetic as the bf16 autocast library path).
 *   ud_head_tail_stats : training-mode batch statistics of y -> mean, biased var, invstd and the
 *                        folded scale = gamma*invstd, shift = beta - mean*scale.  When running_mean /
 *                        running_var are given (both or neither) they are updated in place like
 *                        nn.BatchNorm2d does: r = (1-momentum)*r + momentum*{mean, unbiased var};
 *                        batches_tracked (optional, int64 on the device) is incremented by one.
 *   ud_head_tail_fwd   : z = conv(relu(y*scale + shift), w2) + b2   (any scale/shift: batch or running).
 *   ud_head_tail_bwd   : training-mode backward through conv, ReLU and BatchNorm: dy, dw2, dgamma,
 *                        dbeta (db2 = sum of dz is left to the caller).  Deterministic. */
size_t ud_head_tail_workspace_bytes(int G);
int ud_head_tail_stats(const void* y, int B, int H, int W, int G, const float* gamma,
                       const float* beta, float eps, float* mean, float* var, float* invstd,
                       float* scale, float* shift, float* running_mean, float* running_var,
                       float momentum, long long* batches_tracked, void* workspace,
                       size_t workspace_bytes, ud_stream_t stream);
int ud_head_tail_fwd(const void* y, const float* scale, const float* shift, const float* w2,
                     const float* b2, float* z, int B, int H, int W, int G, int kmax,
                     ud_stream_t stream);
int ud_head_tail_bwd(const void* y, const float* dz, const float* w2, const float* scale,
                     const float* shift, const float* mean, const float* invstd, void* dy,
                     float* dw2, float* dgamma, float* dbeta, int B, int H, int W, int G, int kmax,
                     void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* fp32 mode (the reference's arithmetic) of the second SepHead convolutions (center_head.py:311-362): a grouped
 * 3x3 / pad-1 convolution with 64 inputs and KM <= 4 outputs per group on channels-last fp32 tensors, exact fp32
 * FMAs, HBM-bound streaming kernels (the libraries run it as a 42x padded block-diagonal conv).
 *   a  f32 [B][H][W][G*64]   hidden tensor after BatchNorm + ReLU        w  f32 [G][KM][9][64] (tap = ky*3+kx)
 *   z  f32 [B][H][W][G*KM]   = bias[G*KM] + conv                          dz same layout as z
 *   dgrad: da [B][H][W][G*64];  wgrad: dw [G][KM][9][64], fixed-order slice reduction (deterministic). */
int ud_head_tail_f32_fwd(const float* a, const float* w, const float* bias, float* z, int B, int H, int W, int G,
                         int KM, ud_stream_t stream);
int ud_head_tail_f32_dgrad(const float* dz, const float* w, float* da, int B, int H, int W, int G, int KM,
                           ud_stream_t stream);
size_t ud_head_tail_f32_wgrad_workspace_bytes(int B, int H, int W, int G, int KM);
int ud_head_tail_f32_wgrad(const float* a, const float* dz, float* dw, int B, int H, int W, int G, int KM,
                           void* workspace, size_t workspace_bytes, ud_stream_t stream);
/* The same second convolutions reading the first convolution's RAW output: relu(a * scale + shift) -- the BatchNorm of the SepHead
 * (center_head.py:341-350) folded per channel [G*64] -- is applied to every piece as it is loaded, in the forward and in the weight
 * gradient; the normalised 1.39 GB hidden tensor is never written or read back. */
int ud_head_tail_f32_bn_fwd(const float* a, const float* bn_scale, const float* bn_shift, const float* w, const float* bias,
                            float* z, int B, int H, int W, int G, int KM, ud_stream_t stream);
int ud_head_tail_f32_bn_wgrad(const float* a, const float* bn_scale, const float* bn_shift, const float* dz, float* dw, int B,
                              int H, int W, int G, int KM, void* workspace, size_t workspace_bytes, ud_stream_t stream);
/* Backward of relu(bn(y)) -> second convolutions down to y in two passes that RECOMPUTE the tail's data gradient instead of
 * storing it: dgamma / dbeta [G*64] (training-mode statistics: mean / invstd of y, folded scale / shift) and dy [B, H, W, G*64].
 * dy_colsum (optional, [G*64]): the per-channel sums of the dy values stored, in a fixed order -- the bias gradient of the
 * convolution that produced y (center_head.py:339 bias=True), emitted by the pass that writes dy instead of a third pass over it. */
size_t ud_head_tail_f32_bn_bwd_workspace_bytes(int B, int H, int W, int G, int KM);
int ud_head_tail_f32_bn_bwd(const float* dz, const float* w, const float* y, const float* bn_scale, const float* bn_shift,
                            const float* mean, const float* invstd, float* dy, float* dgamma, float* dbeta, float* dy_colsum,
                            int B, int H, int W, int G, int KM, void* workspace, size_t workspace_bytes, ud_stream_t stream);
/* Column sums of a channels-last tensor, out[c] = sum_p x[p * ld + c] (ld >= C elements between rows): the bias gradient of a
 * convolution = the sum of its output gradient over batch and pixels (reference: autograd of nn.Conv2d(bias=True),
 * center_head.py:64,339,353; lss_fpn.py:160 depth net).  Two HBM-rate passes, fixed summation order (deterministic); any C
 * (16-byte vector loads when C % 8 == 0 and the rows are 16-byte aligned). */
size_t ud_colsum_workspace_bytes(int C);
int ud_colsum_f32(const float* x, long long P, int C, long long ld, float* out, void* workspace, size_t workspace_bytes,
                  ud_stream_t stream);
int ud_colsum_bf16(const void* x, long long P, int C, long long ld, float* out, void* workspace, size_t workspace_bytes,
                   ud_stream_t stream);

/* ---- Dense 3x3 / stride 1 / pad 1 convolution, channels-last bf16 (BEV trunk + head convs) ----------
 * Replaces nn.Conv2d(k=3, s=1, p=1) of BaseBEVBackbone (reference unidistill/layers/blocks_2d/det3d/
 * base_bev_backbone.py:30-110) and CenterHead.shared_conv (layers/head/det3d/center_head.py:408-420)
 * when tensors are bf16 channels-last: x [B][H][W][Cin], w [Cout][9][Cin] (tap = ky*3+kx),
 * y [B][H][W][Cout]; fp32 accumulation on the MFMA pipe.  Optional fused epilogue, in this order:
 * + bias[Cout], * scale + shift (folded eval BatchNorm), + residual (bf16, y's layout), ReLU.
 * Cin % 64 == 0 and Cout % 8 == 0, else UD_ERR_UNSUPPORTED.  The data gradient of the convolution is
 * the same call on dy with w' [Cin][9][Cout], w'[c][8 - tap][n] = w[n][tap][c] -- or with the un-flipped
 * w'[c][tap][n] = w[n][tap][c] and bit 1 of `relu` set (walk the taps in reverse).  `relu`: bit 0 = ReLU. */
int ud_conv3x3_nhwc_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout,
                         const float* bias, const float* scale, const float* shift,
                         const void* residual, int relu, ud_stream_t stream);
/* 1x1 / stride-1 convolution (the ResNet bottleneck 1x1 convs of the reference's image branch) on the same
 * kernel family: x [P][Cin] bf16 (P = B*H*W channels-last pixels), w [Cout][Cin] bf16, y [P][Cout] bf16,
 * same fused epilogue (bit 0 of `relu` only).  The data gradient is the same call on dy with w^T
 * [Cin][Cout].  Cin % 64 == 0, Cout % 8 == 0, else UD_ERR_UNSUPPORTED. */
int ud_conv1x1_nhwc_bf16(const void* x, const void* w, void* y, int64_t P, int Cin, int Cout,
                         const float* bias, const float* scale, const float* shift,
                         const void* residual, int relu, ud_stream_t stream);
/* fp32 twins (the reference trains in fp32: exps/base_cli.py:40-45 has no precision flag): channels-last
 * FP32 x / w / y, exact fp32 products and fp32 accumulation on v_mfma_f32_16x16x4_f32.  Same argument
 * meaning and fused epilogue as the bf16 calls (residual is fp32, y's layout); `flags` bit 0 = ReLU, bit 1 =
 * walk the taps in reverse (data gradient on un-flipped transposed weights).  Cin % 32 == 0, Cout % 4 == 0. */
int ud_conv3x3_nhwc_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout,
                        const float* bias, const float* scale, const float* shift, const float* residual,
                        int flags, ud_stream_t stream);
int ud_conv1x1_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout,
                        const float* bias, const float* scale, const float* shift, const float* residual,
                        int flags, ud_stream_t stream);
/* Convolution (+ bias) fused with the FIRST pass of the training-mode BatchNorm that follows it in the reference's
 * Conv2d -> BatchNorm2d -> ReLU links (base_bev_backbone.py:48-66, center_head.py:408-420, the mmdet ResNet
 * bottlenecks): besides y, every workgroup writes partial[tile][Cout][2] = (sum, sum of squares) of the values it
 * stored (for bf16 the rounded ones: what a separate statistics pass would read back).  `partial_bytes` >=
 * ud_conv{3x3,1x1}_bnstats_bytes(...) (an upper bound: the tile height is chosen per launch); *slices (host int) =
 * number of tiles written.  Finish with ud_bn_stats_from_partials.  Same shape rules as the plain calls. */
size_t ud_conv3x3_bnstats_bytes(int B, int H, int W, int Cout);
size_t ud_conv1x1_bnstats_bytes(int64_t P, int Cout);
int ud_conv3x3_bnstats_nhwc_bf16(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout,
                                 const float* bias, float* partial, size_t partial_bytes, int* slices,
                                 ud_stream_t stream);
int ud_conv1x1_bnstats_nhwc_bf16(const void* x, const void* w, void* y, int64_t P, int Cin, int Cout,
                                 const float* bias, float* partial, size_t partial_bytes, int* slices,
                                 ud_stream_t stream);
int ud_conv3x3_bnstats_nhwc_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout,
                                const float* bias, float* partial, size_t partial_bytes, int* slices,
                                ud_stream_t stream);
int ud_conv1x1_bnstats_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout,
                                const float* bias, float* partial, size_t partial_bytes, int* slices,
                                ud_stream_t stream);
/* Weight gradient of the same convolution: dw [Cout][9][Cin] fp32 = sum over pixels of
 * dy [B][H][W][Cout] (bf16) x shifted x [B][H][W][Cin] (bf16); fp32 accumulation, fixed-order
 * reduction of pixel slices (deterministic).  Cin % 64 == 0, Cout % 8 == 0. */
size_t ud_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ud_conv3x3_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin,
                               int Cout, void* workspace, size_t workspace_bytes, ud_stream_t stream);
/* Weight gradient of a 1x1 / stride-1 convolution (reference: the ResNet bottleneck and neck 1x1 convs,
 * nn.Conv2d backward): dw [Cout][Cin] fp32 = sum over the P = B*H*W pixels of dy [P][Cout] (bf16) x
 * x [P][Cin] (bf16), channels-last rows.  Pixel-sliced MFMA GEMM with a fixed-order reduction of the
 * slices (deterministic).  Cin % 64 == 0, Cout % 8 == 0, else UD_ERR_UNSUPPORTED. */
size_t ud_conv1x1_wgrad_workspace_bytes(int64_t P, int Cin, int Cout);
int ud_conv1x1_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, int64_t P, int Cin, int Cout,
                               void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* The convolutions whose im2col is a pure permutation, on the same 1x1 kernels through a pixel-address map
 * {mode, s, Ho, Wo, H, W, C, a, b} (9 ints, HOST; NULL = plain [P][K] rows; a = b = 0 unless stated):
 *   mode 1: row p = (b, oy, ox) of the virtual matrix is the s x s block of the channels-last tensor [B,H,W,C] at
 *           (s*oy, s*ox), its K = s*s*C elements ordered (dy, dx, c)  -- conv k = s / stride s (neck levels, reference
 *           lss_fpn.py:143-149 / SECONDFPN) on the input side, transposed conv k = s / stride s (BaseBEVBackbone
 *           deblocks, base_bev_backbone.py:67-92; neck) on the output side;  (s*C) % 64 == 0 on an input side;
 *   mode 2: row p = pixel (s*oy + a, s*ox + b), K = C -- the stride-s 1x1 shortcut convs of the ResNet stages;
 *   mode 3 (input / x side only): im2col of a 3x3 / pad 1 / stride s convolution, K = 9*C ordered (tap, c), zeros
 *           outside the tensor -- the stride-2 3x3 convs of the ResNet stages and of BaseBEVBackbone's second level
 *           (forward and weight gradient; C % 64 == 0);
 *   mode 4 (input side only, s = 2): the data gradient of that convolution for the input pixels of parity class
 *           (a, b): row p = (batch, i, j) is pixel (2i + a, 2j + b) (written through a mode-2 output map with the same
 *           (a, b)), K = (1 + a)(1 + b) * C gathered from dy -- four launches cover the tensor with exactly the
 *           multiply-adds of the convolution (no zero-stuffed taps).  With a mode-4 input map `w` is the whole
 *           transposed tap-major weight [Cin][3][3][Cout] of the convolution (the kernel picks the class's taps).
 * Forward and data gradient are ud_conv1x1_mapped_nhwc_bf16 with the map on the input or the output (a data gradient
 * through a mode-2 output map writes only the sampled pixels: clear dx first); weight gradients are
 * ud_conv1x1_wgrad_mapped_nhwc_bf16 with the map on x or dy.  w / dw as in the unmapped calls: [Cout][K] / [Cout][Cin]. */
int ud_conv1x1_mapped_nhwc_bf16(const void* x, const void* w, void* y, int64_t P, int Cin, int Cout,
                                const int* in_map, const int* out_map, ud_stream_t stream);
/* fp32 twin (the reference's arithmetic): forward and data gradient of the strided / transposed convolutions of the fp32
 * mode on v_mfma_f32_16x16x4_f32; map channels % 4 == 0, Cin % 32 == 0, Cout % 4 == 0, a 32-channel slice inside one tap. */
int ud_conv1x1_mapped_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout,
                               const int* in_map, const int* out_map, ud_stream_t stream);
/* Persistent form of the fp32 1x1 family (plain and mapped, csrc/conv2d_f32_1x1p.hip; the same layers as ud_conv1x1_nhwc_f32 /
 * ud_conv1x1_mapped_nhwc_f32): 512 resident workgroups walk (128-pixel tile, 64-channel block) units data parallel + a stream-K
 * tail of 32-channel slices (pieces of a cut unit are summed in slice order by a fix-up launch: deterministic), slices flowing
 * through a three-stage LDS ring two ahead of the MFMAs, epilogue (bias, folded BN, residual, ReLU if flags & 1) from registers.
 * partial != NULL: BatchNorm partial sums [*slices = ceil(P / 128)][Cout][2] (ud_conv1x1_bnstats_bytes).  in_map / out_map as in
 * ud_conv1x1_mapped_nhwc_f32 (NULL = plain; mapped launches take no epilogue inputs); x_elems = elements of the mapped input
 * tensor, y_elems of the mapped output tensor (ignored for plain sides).  workspace (ud_conv1x1p_f32_workspace_bytes) enables the stream-K tail.  Tensors < 4 GB.
 * ud_conv1x1_f32_persistent(mode): -1 default (UD_F32_1X1P or 1), 0 = callers use the grid-per-tile kernels, 1 = this one. */
size_t ud_conv1x1p_f32_workspace_bytes(void);
void ud_conv1x1_f32_persistent(int mode);
void ud_conv1x1p_stream_k(int mode);   /* tests / tuning: -1 default (the launcher's cost model), 0 whole units only, 2 cut every tail that can be cut */
int ud_conv1x1_f32_persistent_enabled(void);
int ud_conv1x1p_nhwc_f32(const float* x, const float* w, float* y, int64_t P, int Cin, int Cout, const float* bias,
                         const float* scale, const float* shift, const float* residual, int flags, float* partial,
                         size_t partial_bytes, int* slices, const int* in_map, const int* out_map, size_t x_elems,
                         size_t y_elems, void* workspace, size_t workspace_bytes, ud_stream_t stream);
int ud_conv1x1_wgrad_mapped_nhwc_bf16(const void* x, const void* dy, float* dw, int64_t P, int Cin, int Cout,
                                      const int* x_map, const int* dy_map, void* workspace,
                                      size_t workspace_bytes, ud_stream_t stream);
/* fp32 weight gradients (the reference's arithmetic; replaces nn.Conv2d's backward-weights pass of the BEV trunk,
 * base_bev_backbone.py:38-115, the CenterHead convolutions, center_head.py:58-99,311-355, and the image branch) on
 * v_mfma_f32_16x16x4_f32, exact fp32 products; pixel slices reduced in a fixed order (deterministic, no atomics).
 *   3x3 / stride 1 / pad 1: x [B][H][W][Cin], dy [B][H][W][Cout] -> dw [Cout][3][3][Cin];
 *   1x1 over pixel maps (NULL = plain rows; maps as above, on x or dy): dw [Cout][Cin'].
 * Cin % 4 == 0, Cout % 4 == 0 (and map channels % 4 == 0), else UD_ERR_UNSUPPORTED. */
size_t ud_conv3x3_wgrad_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ud_conv3x3_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                              void* workspace, size_t workspace_bytes, ud_stream_t stream);
size_t ud_conv1x1_wgrad_f32_workspace_bytes(int64_t P, int Cin, int Cout);
int ud_conv1x1_wgrad_mapped_nhwc_f32(const float* x, const float* dy, float* dw, int64_t P, int Cin, int Cout,
                                     const int* x_map, const int* dy_map, void* workspace, size_t workspace_bytes,
                                     ud_stream_t stream);

/* fp32 3x3 / stride 1 / pad 1 as Winograd F(2x2, 3x3) on the fp32 MFMA pipe (2.25x fewer matrix flops than the direct form;
 * same layers as ud_conv3x3_nhwc_f32: base_bev_backbone.py:30-110, center_head.py:311-420, lss_fpn.py:143-149; the framework the
 * reference runs on picks the same algorithm family for fp32 3x3 layers).  Weights are transformed once per weight version:
 *   ud_conv3x3_wino_f32_weights(w, strides of (n, c, ky, kx) in elements, N, C, flip, U): U = G g G^T of g[n][ky][kx][c]; the forward
 *   pass uses (n, c) = (Cout, Cin), the data gradient (n, c) = (Cin, Cout) with flip = 1 (taps reversed);
 *   U holds ud_conv3x3_wino_f32_weight_bytes(C, N) bytes.
 * ud_conv3x3_wino_nhwc_f32: y = conv(x) (+ bias) (* scale + shift: folded eval-mode BatchNorm, both or NULL) (+ residual) (ReLU if flags & 1); partial != NULL also returns the per-workgroup
 * BatchNorm partial sums [*slices][Cout][2] (ud_conv3x3_wino_bnstats_bytes).  Cin % 8 == 0, Cout % 4 == 0. */
size_t ud_conv3x3_wino_f32_weight_bytes(int Cin, int Cout);
size_t ud_conv3x3_wino_bnstats_bytes(int B, int H, int W, int Cout);
int ud_conv3x3_wino_f32_blocks(int H, int W);   /* 64-tile blocks per image of the launch plan (fill = ceil(H/2) ceil(W/2) / (64 blocks)) */
int ud_conv3x3_wino_f32_weights(const float* w, int64_t s_n, int64_t s_c, int64_t s_y, int64_t s_x, int N, int C, int flip,
                                float* U, ud_stream_t stream);
int ud_conv3x3_wino_nhwc_f32(const float* x, const float* U, float* y, int B, int H, int W, int Cin, int Cout,
                             const float* bias, const float* scale, const float* shift, const float* residual, int flags,
                             float* partial, size_t partial_bytes, int* slices, ud_stream_t stream);

/* The same layers as Winograd F(4x4, 3x3) (csrc/conv2d_f32_wino4.hip): 36 multiplications per 4 x 4 outputs, 4x fewer matrix flops than
 * the direct form and 1.78x fewer than F(2x2, 3x3); rounding 3-5e-6 of the output's max against fp64 (tested per layer shape).  Same
 * contract as the ud_conv3x3_wino_* entry points; U from ud_conv3x3_wino4_f32_weights (ud_conv3x3_wino4_f32_weight_bytes(C, N) bytes);
 * ud_conv3x3_wino4_f32_blocks: 32-tile blocks per image of the launch plan (fill = ceil(H/4) ceil(W/4) / (32 blocks)).
 * Cin % 8 == 0, Cout % 4 == 0. */
size_t ud_conv3x3_wino4_f32_weight_bytes(int Cin, int Cout);
size_t ud_conv3x3_wino4_bnstats_bytes(int B, int H, int W, int Cout);
int ud_conv3x3_wino4_f32_blocks(int H, int W);
int ud_conv3x3_wino4_f32_weights(const float* w, int64_t s_n, int64_t s_c, int64_t s_y, int64_t s_x, int N, int C, int flip,
                                 float* U, ud_stream_t stream);
size_t ud_conv3x3_wino4_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout);   /* stream-K partial tiles (optional: NULL = whole units only) */
void ud_conv3x3_wino4_stream_k(int mode);   /* schedule rule: -1 default (= 2), 0 whole units only, 1 stream-K tail when Cin >= 256, 2 every tail that beats one more round */
int ud_conv3x3_wino4_nhwc_f32(const float* x, const float* U, float* y, int B, int H, int W, int Cin, int Cout,
                              const float* bias, const float* scale, const float* shift, const float* residual, int flags,
                              float* partial, size_t partial_bytes, int* slices, void* workspace, size_t workspace_bytes,
                              ud_stream_t stream);

/* Weight gradient of the same layers through the Winograd form (gradient of ud_conv3x3_wino_nhwc_f32: 16 instead of 36 multiplications
 * per 2 x 2 tile and (n, c)); same contract as ud_conv3x3_wgrad_nhwc_f32: dw [Cout][3][3][Cin], tile slices reduced in a fixed order. */
size_t ud_conv3x3_wino_wgrad_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ud_conv3x3_wino_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                   void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* ... and through the F(4x4, 3x3) form (csrc/conv2d_f32_wino4_wgrad.hip: 36 multiplications per 16 output pixels and (n, c) instead of
 * 64): same contract; Cin % 32 == 0, Cout % 64 == 0, both tensors below 2 GB, else UD_ERR_UNSUPPORTED. */
size_t ud_conv3x3_wino4_wgrad_f32_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ud_conv3x3_wino4_wgrad_nhwc_f32(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                    void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* ---- LiDAR input side ------------------------------------------------------------------
 * Replaces the numpy point transforms of the reference's data pipeline:
 * CollectLidarSweeps.forward (unidistill/data/multisensorfusion/transforms3d.py:379-414) and the point part
 * of BevAffineTransformation.forward (:417-443).  in/out f32 [rows][D] (D >= 3; in == out allowed); seg:
 * S+1 ascending row offsets (device, int64); mats: S row-major 4x4 float64 matrices (device); segment s
 * gets xyz <- (mats[s] @ [x y z 1]^T)[:3] evaluated in float64 in numpy's order and rounded to float32
 * (bit-identical to the reference), columns 3.. copied, and, when `last` (device, f32[S]) is given and
 * last[s] is not NaN, column D-1 <- last[s] (the sweep's time lag).  max_rows = longest segment. */
int ud_points_transform(const float* in, float* out, const int64_t* seg, const double* mats, const float* last,
                        int S, int D, int64_t max_rows, ud_stream_t stream);

/* ---- camera input side + collate ------------------------------------------------------
 * ImageNormalize.forward (data/multisensorfusion/transforms3d.py:350-368 -> mmcv.imnormalize, third party:
 * float32(img) [channel order reversed when to_rgb] - float32(mean), * float32(1 / float64(std))) fused with the
 * dataset's HWC -> CHW permute + stack (nuscenes_multimodal.py:262-293).  img u8 [NI][H][W][3] (device), out f32
 * [NI][3][H][W], or [NI][H][W][3] memory when out_channels_last (a channels-last view for the NHWC model);
 * mean/std: HOST float[3]. */
int ud_image_normalize(const unsigned char* img, float* out, const float* mean, const float* std, int to_rgb,
                       int NI, int H, int W, int out_channels_last, ud_stream_t stream);
/* collate_fn.fill_batch_tensor for ragged samples (nuscenes_multimodal.py:441-463): out f32 [B][L][W] (device),
 * out[b, :rows[b]] = samples[b] (device pointers, HOST array of B), zero rows up to L.  rows: HOST int64[B]. */
int ud_collate_pad(const float* const* samples, const int64_t* rows, int B, int64_t L, int W, float* out,
                   ud_stream_t stream);

/* ---- frozen ResNet stem (image branch) ----------------------------------------------------------------
 * conv1 (7x7 / stride 2 / pad 3, 3 -> 64, no bias) + bn1 (eval mode, folded to scale / shift) + ReLU, then max-pool 3x3 /
 * stride 2 / pad 1, of the mmdet ResNet-50 the reference builds in unidistill/layers/blocks_3d/mmdet3d/lss_fpn.py:143-149 with
 * frozen_stages = 0 (exps/.../BEVFusion_nuscenes_centerhead_fusion_exp.py:24-31).  Forward only (the stem takes no gradient).
 *   ud_stem_pack_weights     : HOST helper: w = host pointer to the [64][3][7][7] filters (strides in floats) ->
 *                              packed f32[7*6*4*16*4] in the kernel's MFMA operand order (once per frozen filter)
 *   ud_stem_conv7x7_bn_relu  : x f32 [B][3][H][W] through its strides in floats (NCHW planes or channels-last memory),
 *                              packed_w / scale[64] / shift[64] device f32 -> y [B][OH][OW][64] channels-last, f32 or bf16
 *                              (OH = (H - 1) / 2 + 1, same for OW); fp32 MFMA, exact products
 *   ud_maxpool3x3s2_nhwc     : x [B][H][W][C] -> y [B][(H - 1) / 2 + 1][(W - 1) / 2 + 1][C], f32 (C % 4 == 0) or bf16 (C % 8 == 0) */
int ud_stem_pack_weights(const float* w, int64_t sn, int64_t sc, int64_t sky, int64_t skx, float* packed);
int ud_stem_conv7x7_bn_relu(const float* x, int64_t sb, int64_t sc, int64_t sy, int64_t sx, int B, int H, int W,
                            const float* packed_w, const float* scale, const float* shift, void* y, int out_bf16,
                            ud_stream_t stream);
int ud_maxpool3x3s2_nhwc(const void* x, void* y, int B, int H, int W, int C, int is_bf16, ud_stream_t stream);

/* ---- BatchNorm2d (+ residual) (+ ReLU), channels-last bf16 ------------------------------------------
 * The Conv2d -> BatchNorm2d -> ReLU links of the reference's dense layers (base_bev_backbone.py:48-66,
 * center_head.py:408-420, the mmdet ResNet bottlenecks) as HBM-bound streaming kernels over
 * x [P rows][C] bf16 -- P = B*H*W pixels of a channels-last map, or the active voxels of a sparse tensor
 * (the BatchNorm1d of the reference's sparse blocks, spconv_backbone.py:10-113).
 *   ud_bn_stats   : training-mode batch statistics -> mean, biased var, invstd, folded scale/shift;
 *                   running_mean / running_var (optional, both or neither) are updated in place like
 *                   nn.BatchNorm does (unbiased variance); batches_tracked (optional device int64) += 1.
 *                   C % 16 == 0.
 *   ud_bn_act_fwd : y = act(x * scale + shift (+ residual)), scale = gamma*invstd, shift = beta - mean*scale
 *                   (batch statistics in training, running statistics in eval).  C % 8 == 0.
 *   ud_bn_act_bwd : training-mode backward: dx (bf16), dgamma, dbeta [C] and, when `dresidual` is given,
 *                   the masked gradient for the residual branch.  Pass `y` (the saved output) when a
 *                   residual took part in the forward, NULL otherwise (the ReLU mask is then recomputed
 *                   from x).  C % 16 == 0.  Deterministic two-stage reductions. */
size_t ud_bn_act_workspace_bytes(int C);
int ud_bn_stats(const void* x, long long P, int C, const float* gamma, const float* beta, float eps,
                float* mean, float* var, float* invstd, float* scale, float* shift, float* running_mean,
                float* running_var, float momentum, long long* batches_tracked, void* workspace,
                size_t workspace_bytes, ud_stream_t stream);
int ud_bn_act_fwd(const void* x, const void* residual, const float* scale, const float* shift, void* y,
                  long long P, int C, int relu, ud_stream_t stream);
int ud_bn_act_bwd(const void* x, const void* y, const void* dy, const float* scale, const float* shift,
                  const float* mean, const float* invstd, void* dx, void* dresidual, float* dgamma,
                  float* dbeta, long long P, int C, int relu, void* workspace, size_t workspace_bytes,
                  ud_stream_t stream);
/* ud_bn_stats for a tensor whose first statistics pass came out of a convolution epilogue
 * (ud_conv*_bnstats_nhwc_*): partial [slices][C][2] per-tile (sum, sum of squares) over the P rows, reduced in slice
 * order (deterministic); outputs and running-statistics update exactly as ud_bn_stats. */
int ud_bn_stats_from_partials(const float* partial, int slices, long long P, int C, const float* gamma,
                              const float* beta, float eps, float* mean, float* var, float* invstd, float* scale,
                              float* shift, float* running_mean, float* running_var, float momentum,
                              long long* batches_tracked, ud_stream_t stream);
/* fp32 twins for the fp32 (reference-arithmetic) mode: x / residual / y / dy / dx are FP32 rows, everything
 * else as above (C % 16 == 0). */
int ud_bn_stats_f32(const float* x, long long P, int C, const float* gamma, const float* beta, float eps,
                    float* mean, float* var, float* invstd, float* scale, float* shift, float* running_mean,
                    float* running_var, float momentum, long long* batches_tracked, void* workspace,
                    size_t workspace_bytes, ud_stream_t stream);
int ud_bn_act_fwd_f32(const float* x, const float* residual, const float* scale, const float* shift, float* y,
                      long long P, int C, int relu, ud_stream_t stream);
int ud_bn_act_bwd_f32(const float* x, const float* y, const float* dy, const float* scale, const float* shift,
                      const float* mean, const float* invstd, float* dx, float* dresidual, float* dgamma,
                      float* dbeta, long long P, int C, int relu, void* workspace, size_t workspace_bytes,
                      ud_stream_t stream);
/* y (forward) / dy (backward) as a CHANNEL SLICE of a wider channels-last map: rows of C elements every `ld` elements
 * (ld >= C, ld % 8 == 0, the slice's first element 16-byte aligned); everything else as the functions above.  The upsampling
 * heads of the BEV trunk (reference base_bev_backbone.py:117-141: `torch.cat(ups, dim=1)` over the deblocks' BatchNorm + ReLU
 * outputs) write straight into the concatenated map and read its gradient in place: no cat kernel forward, no strided-slice
 * copies backward. */
int ud_bn_act_fwd_ld(const void* x, const void* residual, const float* scale, const float* shift, void* y,
                     long long P, int C, long long y_ld, int relu, ud_stream_t stream);
int ud_bn_act_fwd_ld_f32(const float* x, const float* residual, const float* scale, const float* shift, float* y,
                         long long P, int C, long long y_ld, int relu, ud_stream_t stream);
int ud_bn_act_bwd_ld(const void* x, const void* y, const void* dy, long long dy_ld, const float* scale, const float* shift,
                     const float* mean, const float* invstd, void* dx, void* dresidual, float* dgamma,
                     float* dbeta, long long P, int C, int relu, void* workspace, size_t workspace_bytes,
                     ud_stream_t stream);
int ud_bn_act_bwd_ld_f32(const float* x, const float* y, const float* dy, long long dy_ld, const float* scale,
                         const float* shift, const float* mean, const float* invstd, float* dx, float* dresidual,
                         float* dgamma, float* dbeta, long long P, int C, int relu, void* workspace,
                         size_t workspace_bytes, ud_stream_t stream);

/* ---- Proposal layer: rotated-BEV IoU + greedy NMS -----------------------------------------------------
 * Replaces `iou3d_nms_cuda.nms_gpu(boxes, keep, thresh)` (reference layers/head/det3d/generate_proposals/
 * centerpoint_gen_proposals.py:85-105; OpenPCDet iou3d_nms, binary absent from the tree).
 * boxes f32[N,7] = (x, y, z, dx, dy, dz, heading) sorted by descending score.  keep i64[N] receives the
 * kept indices in order (rest = -1), *num_keep (device int) their count; nothing synchronises with the
 * host.  N <= 16384.  ud_boxes_iou_bev: iou f32[Na,Nb] of the BEV footprints. */
size_t ud_nms_bev_workspace_bytes(int N);
int ud_nms_rotated_bev(const float* boxes, int N, float thresh, long long* keep, int* num_keep,
                       void* workspace, size_t workspace_bytes, ud_stream_t stream);
int ud_boxes_iou_bev(const float* a, int Na, const float* b, int Nb, float* iou, ud_stream_t stream);

/* ---- Proposal layer on the device: top-K decode + IoU-aware score + filters + rotated NMS + roi packing ----
 * Replaces IouAwareGenProposals / CenterPointGenProposals.generate_predicted_boxes (reference layers/head/
 * det3d/generate_proposals/iou_aware_gen_proposals.py:43-139, centerpoint_gen_proposals.py:66-105,232-340)
 * for all T <= 8 tasks and B samples in three launches; nothing synchronises with the host.
 *   heads   HOST array [T*7] of device pointers: hm, reg, height, dim, rot, vel, iou of task t (raw head
 *           outputs, fp32, logical shape [B, c, H, W]); vel may be NULL when box_dim == 7, iou when iou_alpha
 *           is NULL (plain CenterPoint: NMS ranks by the heat-map score).
 *   strides HOST array [T*7*3]: (batch, channel, pixel) strides in elements of each tensor (NCHW planes:
 *           (c*H*W, H*W, 1); channel slices of a channels-last packed map: (H*W*Ctot, 1, Ctot)).
 *   num_classes / class_offsets / iou_alpha  HOST arrays [T]: classes of the task's heat map, number of
 *           classes of the tasks before it (labels are 1-based global ids), IoU-aware exponent a
 *           (NMS score = score^(1-a) * clamp(iou/2+0.5, 0, 1)^a).
 *   K = nms_pre_max_size (<= 2048), post_max = nms_post_max_size (<= 512); box = (x, y, z, dx, dy, dz, rot
 *   [, vx, vy]); dx..dz = clamp(exp(dim), 0.001, 30) unless no_log; x = (col + reg.x) * out_size_factor *
 *   voxel_x + pc_x (same fp32 operation order as the reference); kept iff center_range[0:3] <= (x,y,z) <=
 *   center_range[3:6] (HOST array [6]) and score > score_threshold.
 * Outputs: rois f32[B, T*post_max, box_dim], roi_scores f32[B, T*post_max], roi_labels i64[B, T*post_max]
 * (tasks back to back, each in NMS order, zero padded), num_boxes i32[B] = rows in use per sample. */
size_t ud_proposal_workspace_bytes(int B, int T, int K);
int ud_proposal_layer(const float* const* heads, const long long* strides, const int* num_classes,
                      const int* class_offsets, const float* iou_alpha, int B, int T, int H, int W, int K,
                      int post_max, int box_dim, int no_log, float out_size_factor, float voxel_x,
                      float voxel_y, float pc_x, float pc_y, const float* center_range, float score_threshold,
                      float nms_threshold, float* rois, float* roi_scores, long long* roi_labels,
                      int* num_boxes, void* workspace, size_t workspace_bytes, ud_stream_t stream);

/* ---- Detection loss of the CenterPoint heads (focal + gathered regression / IoU terms) ------------------
 * CenterHeadIouAware.get_loss (reference layers/head/det3d/center_head_iou_aware.py:55-298, losses/det3d.py:
 * 287-421) for all T <= 8 tasks at once.  Head tensors are given as device-pointer tables (any
 * [B, c, H, W] fp32 tensors with contiguous H*W planes; *_bstride = elements between batch entries).
 *   ud_det_focal_fwd : prob [T,B,ncm,HW] = clamp(sigmoid(hm), 1e-4, 1-1e-4) (0 in padded channels),
 *                      pos_neg [T,2] = (sum log(p)(1-p)^g a [gt==1], sum log(1-p+1e-4) p^g (1-a) [gt==0]).
 *   ud_det_focal_bwd : dlogit [T,B,ncm,HW] from g_prob (optional), g_pos [T], g_neg [T].
 *   ud_det_reg_fwd   : head[t*11+j] = plane of gathered value j (reg.x reg.y height dim0..2 rot.sin rot.cos
 *                      vel.x vel.y iou); ind i64 / mask u8 / tgt f32 [T,B,K(,tgt_dim)], num_obj f32 [T];
 *                      losses [T,12] = box L1 per code dim (10), IoU loss, IoU-aware L1, all normalised;
 *                      loc [T,B,K,17] = local derivatives kept for the backward.  nb = 10 (nuScenes codes).
 *   ud_det_reg_bwd   : dhead [T,B,11,HW] (caller zero-fills) += scaled local derivatives at the slots.
 * Ordered two-stage reductions; no host synchronisation. */
size_t ud_det_loss_workspace_bytes(int T);
int ud_det_focal_fwd(const float* const* hm, const long long* hm_bstride, const int* ncls, int T, int B,
                     int ncm, int HW, const float* gt, float alpha, float gamma, float* prob,
                     float* pos_neg, void* workspace, size_t workspace_bytes, ud_stream_t stream);
int ud_det_focal_bwd(const int* ncls, int T, int B, int ncm, int HW, const float* gt, const float* prob,
                     const float* g_prob, const float* g_pos, const float* g_neg, float alpha,
                     float gamma, float* dlogit, ud_stream_t stream);
int ud_det_reg_fwd(const float* const* head, const long long* head_bstride, int T, int B, int K, int HW,
                   int nb, const long long* ind, const unsigned char* mask, const float* tgt, int tgt_dim,
                   const float* num_obj, float sx, float sy, float* losses, float* loc, void* workspace,
                   size_t workspace_bytes, ud_stream_t stream);
int ud_det_reg_bwd(int T, int B, int K, int HW, int nb, const long long* ind, const unsigned char* mask,
                   const float* loc, const float* g_box, const float* g_iou, const float* g_aw,
                   float* dhead, ud_stream_t stream);

/* ---- FCOS-style target assignment of the CenterPoint heads -----------------------------------
 * FCOSAssigner.assign_targets (reference layers/head/det3d/target_assigner/fcos_assigner.py:73-285) for all
 * T tasks and B samples in one launch.  gt f32[B,M,cols] (x y z dx dy dz yaw [extra...] class(1-based));
 * task_of / off_of: HOST int8 tables indexed by class id (task of the class or -1, offset inside the task).
 * Outputs: hm f32[T,B,ncm,h*w] one-hot class map of the positive anchors, ind i64 / mask u8 / cat i64
 * [T,B,K] (positives in ascending anchor order), enc f32[T,B,K,enc_dim] box encoding relative to the
 * anchor (log sizes, sin/cos yaw; +-inf for zero-size boxes like the reference).  topk <= 9, M <= 512,
 * grid <= 180 x 180, else UD_ERR_UNSUPPORTED. */
int ud_assign_targets(const float* gt, int B, int M, int cols, const signed char* task_of,
                      const signed char* off_of, int n_classes, int T, int ncm, int w, int h, int K,
                      int topk, int enc_dim, float osf, float pc0, float pc1, float vs0, float vs1,
                      float* hm, long long* ind, unsigned char* mask, long long* cat, float* enc,
                      ud_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIDISTILL_HIP_H */
