"""CPU: FusionEncoder (a11) against the golden produced by the reference class
(BEVFusion_nuscenes_base_exp.py:107-135): state_dict keys, eval / train forward, gradients."""
import numpy as np
import torch

from unidistill_amd.layers.bev import FusionEncoder


def test_fusion_encoder_matches_reference(golden):
    g = golden("fusion_encoder")
    m = FusionEncoder(use_elementwise=False, input_channel=128, output_channel=64)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    x1 = torch.from_numpy(g["x1"]).requires_grad_(True)
    x2 = torch.from_numpy(g["x2"]).requires_grad_(True)
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(m(x1, x2).numpy(), g["y_eval"], rtol=1e-5, atol=1e-6)
    m.train()
    y = m(x1, x2)
    np.testing.assert_allclose(y.detach().numpy(), g["y_train"], rtol=1e-4, atol=1e-5)
    y.backward(torch.from_numpy(g["gy"]))
    np.testing.assert_allclose(x1.grad.numpy(), g["g1"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(x2.grad.numpy(), g["g2"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(m.reduce_conv[0].weight.grad.numpy(), g["gw"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(m.att[1].weight.grad.numpy(), g["gatt"], rtol=1e-3, atol=1e-6)
    np.testing.assert_array_equal(FusionEncoder(use_elementwise=True)(x1, x2).detach().numpy(), g["y_sum"])
