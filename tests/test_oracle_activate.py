"""The oracle's statement of the activations (gso_activate, oracle/gs_oracle.c; scene/gaussian_model.py:33-41,95-118):
accuracy against float64 exp / sigmoid / normalize, the chain rule helper against autograd, and the edge cases."""
import numpy as np
import torch

from oracle import c_oracle


def test_activate_matches_the_reference_activations_to_fp32_rounding():
    rng = np.random.default_rng(0)
    ls = rng.uniform(-12.0, 4.0, (50_000, 3)).astype(np.float32)
    q = (rng.standard_normal((50_000, 4)) * rng.uniform(0.01, 10.0, (50_000, 1))).astype(np.float32)
    lo = rng.uniform(-15.0, 15.0, (50_000, 1)).astype(np.float32)
    s, r, o = c_oracle.activate(ls, q, lo)
    # torch (the reference's own functions) in float64
    s_ref = torch.exp(torch.from_numpy(ls).double()).numpy()
    r_ref = torch.nn.functional.normalize(torch.from_numpy(q).double()).numpy()
    o_ref = torch.sigmoid(torch.from_numpy(lo).double()).numpy()
    assert np.abs(s / s_ref - 1.0).max() < 1.5e-6          # exp_det: < 2e-6 on |x| <= 20
    assert np.abs(r - r_ref).max() < 2e-7
    assert np.abs(o - o_ref).max() < 2e-7
    # and against torch's float32 activations (what train.py feeds the rasteriser): a few ulp
    s32 = torch.exp(torch.from_numpy(ls)).numpy()
    assert np.abs(s / s32 - 1.0).max() < 2e-6


def test_activate_edge_cases():
    s, r, o = c_oracle.activate(np.array([[-200.0, 0.0, 88.0]], np.float32), np.zeros((1, 4), np.float32),
                                np.array([[-200.0]], np.float32))
    assert s[0, 0] >= 0.0 and s[0, 0] < 1e-37 and s[0, 1] == 1.0 and np.isfinite(s[0, 2])
    assert np.array_equal(r, np.zeros((1, 4), np.float32))           # F.normalize: x / max(|x|, 1e-12)
    assert o[0, 0] == 0.0
    s, _, o = c_oracle.activate(np.array([[0.0, 1.0, -1.0]], np.float32), None, np.array([[0.0]], np.float32))
    assert o[0, 0] == 0.5 and abs(s[0, 1] - np.e) < 1e-6
    # NaN in -> NaN out, as torch.exp / torch.sigmoid (the polynomial's underflow clamp must not swallow it: a diverged
    # parameter has to surface in the loss, not render as opacity 1.0)
    s, _, o = c_oracle.activate(np.array([[np.nan, 0.0, 0.0]], np.float32), None, np.array([[np.nan]], np.float32))
    assert np.isnan(s[0, 0]) and s[0, 1] == 1.0 and np.isnan(o[0, 0])


def test_activate_backward_is_the_chain_rule():
    rng = np.random.default_rng(1)
    ls, q, lo = (rng.standard_normal((64, k)).astype(np.float32) for k in (3, 4, 1))
    g_s, g_q, g_o = (rng.standard_normal((64, k)) for k in (3, 4, 1))
    s, r, o = c_oracle.activate(ls, q, lo)
    gs, gq, go = c_oracle.activate_backward(q, s, r, o, g_s, g_q, g_o)
    t = [torch.from_numpy(a).double().requires_grad_(True) for a in (ls, q, lo)]
    outs = (torch.exp(t[0]), torch.nn.functional.normalize(t[1]), torch.sigmoid(t[2]))
    torch.autograd.backward(outs, [torch.from_numpy(g) for g in (g_s, g_q, g_o)])
    for mine, ref in zip((gs, gq, go), t):
        assert np.abs(mine - ref.grad.numpy()).max() <= 1e-5 * max(1.0, np.abs(ref.grad.numpy()).max())
