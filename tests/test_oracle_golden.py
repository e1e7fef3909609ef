"""CPU: pin the oracle (oracle/) against golden vectors captured from the reference python."""
import numpy as np

import oracle


def test_bev_pool_fwd_matches_reference_contract(golden):
    g = golden("bev_pool")
    out, pos = oracle.bev_pool_fwd(g["geom"], g["feat"], int(g["nx"]), int(g["ny"]), int(g["nz"]))
    np.testing.assert_array_equal(pos, g["pos"])
    # reference wrapper returns out.permute(0,3,1,2); index_add order == ascending point order
    np.testing.assert_allclose(out.transpose(0, 3, 1, 2), g["out_nchw"], rtol=1e-6, atol=1e-6)


def test_bev_pool_bwd_matches_reference(golden):
    g = golden("bev_pool")
    gfeat = oracle.bev_pool_bwd(g["gout_nchw"], g["pos"])
    np.testing.assert_array_equal(gfeat, g["gfeat"])
