"""CPU oracle for the UniDistill hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package (see ``ud_oracle.c`` header).  It wraps ``oracle/_build/libud_oracle.so`` (plain C,
built by ``oracle/Makefile``) behind numpy arrays and adds a few numpy-only restatements.
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "libud_oracle.so")
_lib = None


def build(force=False):
    """Compile ud_oracle.c with gcc (idempotent)."""
    src = os.path.join(_DIR, "ud_oracle.c")
    so_omp = _SO.replace("libud_oracle.so", "libud_oracle_omp.so")
    stale = any((not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src) for so in (_SO, so_omp))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _DIR])
    return _SO


_SO_OMP = _SO.replace("libud_oracle.so", "libud_oracle_omp.so")
_lib_omp = None
_use_omp = False


def use_openmp(on):
    """bench.py's cpu_baseline only: route bev_pool_fwd / bev_pool_bwd / lss_lift to the OpenMP build of the same source
    (same bits: the per-cell summation order is kept).  The checker (tests, smoke) always uses the scalar build."""
    global _use_omp
    prev, _use_omp = _use_omp, bool(on)
    return prev


def lib():
    global _lib, _lib_omp
    if _use_omp:
        if _lib_omp is None:
            build()
            _lib_omp = ctypes.CDLL(_SO_OMP)
        return _lib_omp
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


F = ctypes.c_float
I = ctypes.c_int32


def bev_pool_fwd(geom, feat, nx, ny, nz):
    """geom i32[B,N,3], feat f32[B,N,C] -> (out f32[B,ny,nx,C], pos i32[B,N,3])."""
    geom, feat = _i32(geom), _f32(feat)
    B, N, C = feat.shape
    out = np.zeros((B, ny, nx, C), np.float32)
    pos = np.full((B, N, 3), -1, np.int32)
    lib().oracle_bev_pool_fwd(_p(geom, I), _p(feat, F), _p(out, F), _p(pos, I), B, N, C, nx, ny, nz)
    return out, pos


def bev_pool_bwd(gout_nchw, pos):
    """gout f32[B,C,ny,nx], pos i32[B,N,3] -> gfeat f32[B,N,C]."""
    gout_nchw, pos = _f32(gout_nchw), _i32(pos)
    B, C, ny, nx = gout_nchw.shape
    N = pos.shape[1]
    gfeat = np.empty((B, N, C), np.float32)
    lib().oracle_bev_pool_bwd(_p(gout_nchw, F), _p(pos, I), _p(gfeat, F), B, N, C, nx, ny)
    return gfeat


def voxelize(points, voxel_size, pc_range, max_points, max_voxels, with_voxels=True):
    """points f32[B,N,F] -> dict(voxels[M,P,F]|None, coords i32[M,4] (b,z,y,x), num i32[M],
    mean f32[M,F], m i32[B+1])."""
    points = _f32(points)
    if points.ndim == 2:
        points = points[None]
    B, N, Fd = points.shape
    cap = min(B * max_voxels, B * N)
    vs, rg = _f32(voxel_size), _f32(pc_range)
    voxels = np.zeros((cap, max_points, Fd), np.float32) if with_voxels else None
    coords = np.zeros((cap, 4), np.int32)
    num = np.zeros((cap,), np.int32)
    mean = np.zeros((cap, Fd), np.float32)
    m = np.zeros((B + 1,), np.int32)
    fn = lib().oracle_voxelize
    fn.restype = ctypes.c_int
    M = fn(_p(points, F), B, N, Fd, _p(vs, F), _p(rg, F), max_points, max_voxels,
           _p(voxels, F) if with_voxels else None, _p(coords, I), _p(num, I), _p(mean, F), _p(m, I))
    return dict(voxels=voxels[:M] if with_voxels else None, coords=coords[:M], num=num[:M],
                mean=mean[:M], m=m)


def mean_vfe(voxels, num):
    voxels, num = _f32(voxels), _i32(num)
    M, P, Fd = voxels.shape
    out = np.empty((M, Fd), np.float32)
    lib().oracle_mean_vfe(_p(voxels, F), _p(num, I), _p(out, F), M, P, Fd)
    return out


# --------------------------------------------------------------------------------------------
# numpy restatements (small cases only)
# --------------------------------------------------------------------------------------------
def lss_frustum(final_dim, downsample, d_bound):
    """create_frustum (lss_fpn.py:173-198): u[fW], v[fH], d[D] float32 vectors."""
    H, W = final_dim
    fH, fW = H // downsample, W // downsample
    d = np.arange(d_bound[0], d_bound[1], d_bound[2], dtype=np.float32)
    u = np.linspace(0, W - 1, fW, dtype=np.float32)
    v = np.linspace(0, H - 1, fH, dtype=np.float32)
    return u, v, d


def _matvec4_seq(A, x):
    """A [...,4,4] @ x [...,4] the way torch's CPU small-bmm kernel evaluates it
    (aten/native/LinearAlgebra.cpp baddbmm_cpu_kernel: acc = 0; acc += a[k]*b[k] in fp32, product and
    sum rounded separately -- no FMA)."""
    f32 = np.float32
    acc = np.zeros(np.broadcast_shapes(A.shape[:-2], x.shape[:-1]) + (4,), f32)
    for k in range(4):
        acc = (acc + (A[..., :, k] * x[..., k:k + 1]).astype(f32)).astype(f32)
    return acc


def lss_geometry(sensor2ego, intrin, ida, bda, u, v, d, voxel_coord, voxel_size, ida_inv=None,
                 intrin_inv=None):
    """get_geometry + binning (lss_fpn.py:200-240, 311-313) in float32 numpy, operation for operation.
    sensor2ego/intrin/ida [B,ncam,4,4], bda [B,4,4] -> geom f32[B,ncam,D,fH,fW,3], bins i32[...,3].
    ida_inv / intrin_inv: the reference's fp32 torch.inverse results (then the output is bit-identical
    to the reference's CPU output, pinned by tests/golden/lss_geometry.npz); None -> the correctly
    rounded inverse (fp64, one rounding)."""
    f32 = np.float32
    D, fH, fW = len(d), len(v), len(u)
    fr = np.stack([np.broadcast_to(u.reshape(1, 1, fW), (D, fH, fW)),
                   np.broadcast_to(v.reshape(1, fH, 1), (D, fH, fW)),
                   np.broadcast_to(d.reshape(D, 1, 1), (D, fH, fW)),
                   np.ones((D, fH, fW), f32)], -1).astype(f32)              # [D,fH,fW,4]
    B, ncam = sensor2ego.shape[:2]
    if ida_inv is None:
        ida_inv = np.linalg.inv(ida.astype(np.float64)).astype(f32)
    if intrin_inv is None:
        intrin_inv = np.linalg.inv(intrin.astype(np.float64)).astype(f32)
    ida_inv, k_inv, s2e = ida_inv.astype(f32), intrin_inv.astype(f32), sensor2ego.astype(f32)
    combine = np.zeros_like(s2e)
    for k in range(4):
        combine = (combine + (s2e[..., :, k:k + 1] * k_inv[..., k:k + 1, :]).astype(f32)).astype(f32)
    p = _matvec4_seq(ida_inv.reshape(B, ncam, 1, 1, 1, 4, 4), fr[None, None])
    p = np.concatenate([(p[..., :2] * p[..., 2:3]).astype(f32), p[..., 2:]], -1)
    p = _matvec4_seq(combine.reshape(B, ncam, 1, 1, 1, 4, 4), p)
    if bda is not None:
        p = _matvec4_seq(bda.astype(f32).reshape(B, 1, 1, 1, 1, 4, 4), p)
    geom = p[..., :3]
    vc, vs = np.asarray(voxel_coord, f32), np.asarray(voxel_size, f32)
    lo = (vc - vs / f32(2.0)).astype(f32)
    bins = np.trunc(((geom - lo).astype(f32) / vs).astype(f32)).astype(np.int32)
    return geom, bins


def lss_lift_c(depth_feature, D, C):
    """The C twin of lss_lift (ud_oracle.c:oracle_lss_lift; OpenMP in the baseline build)."""
    x = _f32(depth_feature)
    BN, _, fH, fW = x.shape
    lifted = np.empty((BN, D, fH, fW, C), np.float32)
    prob = np.empty((BN, D, fH, fW), np.float32)
    lib().oracle_lss_lift(_p(x, F), BN, D, C, fH, fW, _p(lifted, F), _p(prob, F))
    return lifted, prob


def lss_lift(depth_feature, D, C):
    """softmax(depth) (x) context, permuted to [BN, D, fH, fW, C] (lss_fpn.py:289-310)."""
    if _use_omp:
        return lss_lift_c(depth_feature, D, C)
    x = depth_feature.astype(np.float32)
    z = x[:, :D]
    e = np.exp(z - z.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
    feat = prob[:, None] * x[:, D:D + C][:, :, None]           # [BN, C, D, fH, fW]
    return np.ascontiguousarray(feat.transpose(0, 2, 3, 4, 1)), prob


I64 = ctypes.c_int64


def spconv_subm_rulebook(coords, shape, ksize):
    """coords i32[M,4], shape (B,Dz,Hy,Wx), ksize (kz,ky,kx) -> nbr i32[M,K]."""
    coords = _i32(coords)
    M = coords.shape[0]
    K = ksize[0] * ksize[1] * ksize[2]
    nbr = np.empty((M, K), np.int32)
    lib().oracle_spconv_subm_rulebook(_p(coords, I), M, *[int(v) for v in shape],
                                      *[int(v) for v in ksize], _p(nbr, I))
    return nbr


def spconv_down(coords, shape, ksize, stride, pad):
    """-> (out_coords i32[Mout,4] sorted, out_nbr i32[Mout,K], in_nbr i32[Min,K], out_shape)."""
    coords = _i32(coords)
    Min = coords.shape[0]
    K = ksize[0] * ksize[1] * ksize[2]
    ks, st, pd = (np.asarray(v, np.int32) for v in (ksize, stride, pad))
    fn = lib().oracle_spconv_down
    fn.restype = ctypes.c_int
    cap = max(Min * K, 1)
    oc = np.zeros((cap, 4), np.int32)
    Mout = fn(_p(coords, I), Min, *[int(v) for v in shape], _p(ks, I), _p(st, I), _p(pd, I),
              _p(oc, I), None, None)
    out_nbr = np.empty((Mout, K), np.int32)
    in_nbr = np.empty((Min, K), np.int32)
    fn(_p(coords, I), Min, *[int(v) for v in shape], _p(ks, I), _p(st, I), _p(pd, I), _p(oc, I),
       _p(out_nbr, I), _p(in_nbr, I))
    B, Dz, Hy, Wx = shape
    od = [(d + 2 * p - k) // s + 1 for d, k, s, p in zip((Dz, Hy, Wx), ksize, stride, pad)]
    return oc[:Mout].copy(), out_nbr, in_nbr, (B, *od)


def spconv_conv(feat, nbr, W, bias=None, mirror=False, transpose=False):
    """feat f32[Min,Cin], nbr i32[Mout,K], W f32[Cout,K,Cin] (KRSC).  transpose=True computes the
    input gradient: feat is gout[.,Cout], result has Cin columns."""
    feat, nbr, W = _f32(feat), _i32(nbr), _f32(W)
    Cout, K, Cin = W.shape
    Mout = nbr.shape[0]
    fn = lib().oracle_spconv_conv
    fn.argtypes = [ctypes.c_void_p] * 3 + [I64] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] \
        + [ctypes.c_int] * 4
    if transpose:
        out = np.empty((Mout, Cin), np.float32)
        fn(feat.ctypes.data, nbr.ctypes.data, W.ctypes.data, 1, Cin, K * Cin, int(mirror), None,
           out.ctypes.data, Mout, K, Cout, Cin)
    else:
        b = None if bias is None else _f32(bias)
        out = np.empty((Mout, Cout), np.float32)
        fn(feat.ctypes.data, nbr.ctypes.data, W.ctypes.data, K * Cin, Cin, 1, int(mirror),
           None if b is None else b.ctypes.data, out.ctypes.data, Mout, K, Cin, Cout)
    return out


def spconv_wgrad(feat, nbr, gout, Cout):
    feat, nbr, gout = _f32(feat), _i32(nbr), _f32(gout)
    Mout, K = nbr.shape
    Cin = feat.shape[1]
    gW = np.empty((Cout, K, Cin), np.float32)
    lib().oracle_spconv_wgrad(_p(feat, F), _p(nbr, I), _p(gout, F), _p(gW, F), Mout, K, Cin, Cout)
    return gW


def sparse_to_dense(feat, coords, shape):
    feat, coords = _f32(feat), _i32(coords)
    M, C = feat.shape
    B, Dz, Hy, Wx = shape
    dense = np.empty((B, C, Dz, Hy, Wx), np.float32)
    lib().oracle_sparse_to_dense(_p(feat, F), _p(coords, I), M, C, B, Dz, Hy, Wx, _p(dense, F))
    return dense


def iou_bev(a, b):
    """f32[Na,Nb] BEV IoU of boxes (x,y,z,dx,dy,dz,heading); same float algorithm as the HIP kernel."""
    a, b = _f32(a).reshape(-1, 7), _f32(b).reshape(-1, 7)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().oracle_iou_bev_matrix(_p(a, F), a.shape[0], _p(b, F), b.shape[0], _p(out, F))
    return out


def iou_bev_f64(a, b):
    """Independent double-precision overlap (chord integration) for checking the restatement."""
    import ctypes
    L = lib()
    L.oracle_iou_bev_f64.restype = ctypes.c_double
    a, b = _f32(a).reshape(-1, 7), _f32(b).reshape(-1, 7)
    out = np.empty((a.shape[0], b.shape[0]), np.float64)
    for i in range(a.shape[0]):
        for j in range(b.shape[0]):
            out[i, j] = L.oracle_iou_bev_f64(_p(np.ascontiguousarray(a[i]), F), _p(np.ascontiguousarray(b[j]), F))
    return out


def nms_bev(boxes, thresh):
    """Greedy NMS over score-sorted boxes f32[N,7] -> kept indices (int64), in order."""
    import ctypes
    boxes = _f32(boxes).reshape(-1, 7)
    keep = np.empty((max(boxes.shape[0], 1),), np.int64)
    L = lib()
    L.oracle_nms_bev.restype = ctypes.c_int
    n = L.oracle_nms_bev(_p(boxes, F), boxes.shape[0], ctypes.c_float(thresh),
                         keep.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)))
    return keep[:n].copy()


def proposal_layer(heads, class_names, post_center_limit_range, score_threshold, pc_range, out_size_factor,
                   voxel_size, no_log, nms_iou_threshold, nms_pre_max_size, nms_post_max_size,
                   iou_aware_list=None, with_vel=True):
    """numpy restatement of IouAwareGenProposals / CenterPointGenProposals.generate_predicted_boxes
    (reference layers/head/det3d/generate_proposals/iou_aware_gen_proposals.py:43-139 proposal_layer,
    centerpoint_gen_proposals.py:66-83 _topk, :85-105 _nms_gpu_3d, :232-340 generate_predicted_boxes), with
    the missing `iou3d_nms_cuda.nms_gpu` = nms_bev above.  heads: list over tasks of dicts of f32[B,c,H,W]
    (hm reg height dim rot [vel] [iou], raw head outputs).  fp32 operations in the reference's order.
    -> rois f32[B,T*post,nb], roi_scores f32[B,T*post], roi_labels i64[B,T*post], counts[B]."""
    f = np.float32
    T = len(heads)
    B, _, H, W = heads[0]["hm"].shape
    K, post = nms_pre_max_size, nms_post_max_size
    nb = 9 if with_vel else 7
    rois = np.zeros((B, T * post, nb), f)
    roi_scores = np.zeros((B, T * post), f)
    roi_labels = np.zeros((B, T * post), np.int64)
    counts = np.zeros((B,), np.int64)
    lo, hi = np.asarray(post_center_limit_range[:3], f), np.asarray(post_center_limit_range[3:], f)
    offset = 1                                                  # centerpoint_gen_proposals.py: labels start at 1
    for t, pred in enumerate(heads):
        hm = _f32(pred["hm"])
        heat = (f(1) / (f(1) + np.exp(-hm))).astype(f)         # pred_dict["hm"].sigmoid()
        nc = heat.shape[1]
        flat = lambda a: _f32(a).reshape(B, -1, H * W)
        dim = flat(pred["dim"])
        if not no_log:
            dim = np.clip(np.exp(dim), f(0.001), f(30)).astype(f)
        for b in range(B):
            sc = heat[b].reshape(-1)
            # _topk: per-class top-K then top-K of the union == top-K of all (class, pixel) scores;
            # equal scores: lower (class, pixel) first (stable)
            order = np.argsort(-sc, kind="stable")[:min(K, sc.size)]
            scores = sc[order]
            cls = (order // (H * W)).astype(f)
            pix = order % (H * W)
            ys, xs = (pix // W).astype(f), (pix % W).astype(f)
            reg = flat(pred["reg"])[b][:, pix]
            xs = xs + reg[0]
            ys = ys + reg[1]
            rot = flat(pred["rot"])[b][:, pix]
            rot = np.arctan2(rot[0], rot[1]).astype(f)
            hei = flat(pred["height"])[b][0, pix]
            xs = (xs * f(out_size_factor) * f(voxel_size[0]) + f(pc_range[0])).astype(f)
            ys = (ys * f(out_size_factor) * f(voxel_size[1]) + f(pc_range[1])).astype(f)
            cols = [xs, ys, hei, dim[b][0, pix], dim[b][1, pix], dim[b][2, pix], rot]
            if with_vel:
                vel = flat(pred["vel"])[b][:, pix]
                cols += [vel[0], vel[1]]
            boxes = np.stack(cols, 1).astype(f)
            nms_scores = scores
            if iou_aware_list is not None:
                iou = np.clip(flat(pred["iou"])[b][0, pix] / f(2) + f(0.5), f(0), f(1)).astype(f)
                a = iou_aware_list[t]
                nms_scores = (np.power(scores, f(1 - a)) * np.power(iou, f(a))).astype(f)
            mask = (boxes[:, :3] >= lo).all(1) & (boxes[:, :3] <= hi).all(1) & (scores > f(score_threshold))
            boxes, scores, cls, nms_scores = boxes[mask], scores[mask], cls[mask], nms_scores[mask]
            sel = np.zeros((0,), np.int64)
            if len(nms_scores):
                o = np.argsort(-nms_scores, kind="stable")[:K]
                kept = nms_bev(boxes[o][:, :7], float(nms_iou_threshold))
                sel = o[kept][:post]
            n0, n = counts[b], len(sel)
            rois[b, n0:n0 + n] = boxes[sel]
            roi_scores[b, n0:n0 + n] = scores[sel]
            roi_labels[b, n0:n0 + n] = cls[sel].astype(np.int64) + offset
            counts[b] += n
        offset += len(class_names[t])
    return rois, roi_scores, roi_labels, counts


def conv3x3_nhwc(x, w, bias=None):
    """y[b,h,w,n] = sum_{ty,tx,c} x[b,h+ty-1,w+tx-1,c] * w[n,ty,tx,c] (zero padding), float64 accumulate.
    Plain numpy restatement of nn.Conv2d(k=3, s=1, p=1) on channels-last data (reference
    base_bev_backbone.py:48-66) for checking the MFMA convolution kernel."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C))
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((B, H, W, w.shape[0]))
    for ty in range(3):
        for tx in range(3):
            y += xp[:, ty:ty + H, tx:tx + W] @ w[:, ty, tx].T
    if bias is not None:
        y += np.asarray(bias, np.float64)
    return y


# ---- LiDAR input side -- numpy restatements, pinned by tests/golden/input_prep.npz --------
def sweep_to_key_matrix(key_lidar_to_ego, key_ego_to_global, sweep_pose):
    """The 4x4 (float64) that CollectLidarSweeps applies to a sweep's points
    (data/multisensorfusion/transforms3d.py:394-400; numpy's @ is left-associative, so the four matrices
    are multiplied first and the product meets the points last)."""
    L, G, S = (np.asarray(m, np.float64) for m in (key_lidar_to_ego, key_ego_to_global, sweep_pose))
    return np.linalg.inv(L) @ np.linalg.inv(G) @ S @ L


def points_transform(points, mat, last=None):
    """points f32[N,D]: xyz <- (mat @ [x y z 1]^T)[:3] in float64, stored back as float32; the other columns
    are kept, the last one is overwritten with ``last`` when given (transforms3d.py:392-408, 434-438)."""
    pts = np.array(points, dtype=np.float32, copy=True)
    h = np.ones((pts.shape[0], 4))
    h[:, :3] = pts[:, :3]
    pts[:, :3] = (np.asarray(mat, np.float64) @ h.T).T[:, :3]
    if last is not None:
        pts[:, -1] = last
    return pts


def collect_lidar_sweeps(points, sweeps, key_lidar_to_ego, key_ego_to_global, timestamp, sweep_poses,
                         sweep_timestamps):
    """CollectLidarSweeps.forward (transforms3d.py:379-414) on plain arrays."""
    allp = np.array(points, dtype=np.float32, copy=True)
    if allp.shape[-1] == 5:
        allp[:, -1] = 0.0
    for frame, pose, ts in zip(sweeps, sweep_poses, sweep_timestamps):
        lag = (int(timestamp) - int(ts)) / 1e6 if allp.shape[-1] == 5 else None
        allp = np.concatenate([allp, points_transform(frame, sweep_to_key_matrix(key_lidar_to_ego, key_ego_to_global,
                                                                                pose), lag)])
    return allp


def bev_transform_matrix(rotate_deg, scale, trans, flip_dx, flip_dy):
    """functional.bev_transform's matrix (data/multisensorfusion/functional.py:595-632)."""
    a = rotate_deg / 180 * np.pi
    s, c = np.sin(a), np.cos(a)
    rot = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    sc = np.diag([scale, scale, scale, 1.0])
    tr = np.eye(4)
    tr[:3, 3] = trans
    flip = np.eye(4)
    if flip_dx:
        flip = flip @ np.diag([-1.0, 1.0, 1.0, 1.0])
    if flip_dy:
        flip = flip @ np.diag([1.0, -1.0, 1.0, 1.0])
    return flip @ tr @ sc @ rot


def bev_transform_boxes(gt_boxes, rotate_deg, scale, trans, flip_dx, flip_dy):
    """functional.bev_transform's box update (functional.py:633-646): centres through the matrix, sizes
    scaled, yaw rotated / mirrored, velocities through the matrix's 2x2 block."""
    boxes = np.array(gt_boxes, dtype=np.float32, copy=True)
    rotate_deg, scale = float(rotate_deg), float(scale)    # python scalars: the float32 array ops stay float32
    mat = bev_transform_matrix(rotate_deg, scale, trans, flip_dx, flip_dy)
    if boxes.shape[0] > 0:
        h = np.ones((boxes.shape[0], 4))
        h[:, :3] = boxes[:, :3]
        boxes[:, :3] = (mat @ h.T).T[:, :3]
        boxes[:, 3:6] *= scale
        boxes[:, 6] += rotate_deg / 180 * np.pi
        if flip_dx:
            boxes[:, 6] = np.pi - boxes[:, 6]
        if flip_dy:
            boxes[:, 6] = -boxes[:, 6]
        if boxes.shape[1] > 7:
            boxes[:, 7:] = (mat[:2, :2] @ boxes[:, 7:].T).T
    return boxes, mat


def image_normalize(img_u8, mean, std, to_rgb=True):
    """mmcv.imnormalize as ImageNormalize.forward calls it (transforms3d.py:361-366).  mmcv (pinned mmcv-full
    1.4.2, README.md:17) is absent from the reference tree -- PARITY UNPINNED; published algorithm restated:
    img = float32(img); cv2.cvtColor(BGR2RGB) when to_rgb (channel reversal); cv2.subtract(img, float64(mean));
    cv2.multiply(img, 1 / float64(std)) -- OpenCV evaluates both on a float32 image in float32.  HWC in, HWC out."""
    x = np.asarray(img_u8).astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]
    m = np.asarray(mean, np.float64).astype(np.float32)
    sinv = (1.0 / np.asarray(std, np.float64)).astype(np.float32)
    return ((x - m).astype(np.float32) * sinv).astype(np.float32)


def collate_fill(batch_data):
    """fill_batch_tensor of the reference's collate_fn (nuscenes_multimodal.py:441-463): stack equal-length
    samples, zero-pad ragged ones to the longest; float32."""
    arrs = [np.asarray(d) for d in batch_data]
    lens = [len(a) for a in arrs]
    if max(lens) == min(lens):
        return np.stack(arrs).astype(np.float32)
    tail = next(a.shape[1:] for a in arrs if a.size != 0)
    out = np.zeros((len(arrs), max(lens)) + tuple(tail), np.float32)
    for i, a in enumerate(arrs):
        if a.size != 0:
            out[i, :len(a)] = a
    return out
