"""CPU: the host-side filter packing of the frozen ResNet stem (ud_stem_pack_weights, csrc/stem.hip) against a numpy
restatement of the kernel's operand order [ky][s][g][li][t] <- w[16 t + li][c][ky][kx] with (kx, c) = divmod(6 g + s, 3);
entries 21..23 of a kernel row are the zero pads the MFMA steps multiply with finite patch values."""
import numpy as np


def test_stem_weight_packing(hip_lib):
    rng = np.random.default_rng(5)
    w = rng.standard_normal((64, 3, 7, 7)).astype(np.float32)
    for arr in (w, np.ascontiguousarray(w.transpose(0, 2, 3, 1)).transpose(0, 3, 1, 2)):   # NCHW and channels-last strides
        out = np.full(7 * 6 * 4 * 16 * 4, np.nan, np.float32)
        sn, sc, sky, skx = (s // 4 for s in arr.strides)
        assert hip_lib.ud_stem_pack_weights(arr.ctypes.data, sn, sc, sky, skx, out.ctypes.data) == 0
        ref = np.zeros((7, 6, 4, 16, 4), np.float32)
        for s in range(6):
            for g in range(4):
                j = 6 * g + s
                if j < 21:
                    kx, c = divmod(j, 3)
                    ref[:, s, g] = w[:, c, :, kx].reshape(4, 16, 7).transpose(2, 1, 0)      # [ky][li][t]
        assert np.array_equal(out.reshape(ref.shape), ref)
    assert hip_lib.ud_stem_pack_weights(None, 1, 1, 1, 1, out.ctypes.data) != 0
