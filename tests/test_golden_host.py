"""Host-side logic against fixtures generated from the reference's own Python
(oracle/gen_golden.py -> tests/golden/*.npz).  Runs on CPU: none of these touch a HIP kernel."""
import os
import types

import numpy as np
import pytest
import torch

from raw_ngp_amd.nerf import renderer as R
from raw_ngp_amd.nerf import utils as U


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_grid_offsets_tables(golden_dir, orc):
    g = load(golden_dir, "grid_offsets.npz")
    from raw_ngp_amd.gridencoder.grid import level_table
    cases = {"bound1": dict(desired=2048), "bound2": dict(desired=4096), "plumbing_L8": dict(L=8, desired=2048),
             "prop0": dict(L=5, log2T=17, desired=128), "prop1": dict(L=5, log2T=17, desired=256)}
    for tag, kw in cases.items():
        L, log2T, desired = kw.get("L", 16), kw.get("log2T", 19), kw["desired"]
        scale = np.exp2(np.log2(desired / 16) / (L - 1))
        assert scale == float(g[f"{tag}_scale"])
        ours = level_table(3, L, scale, 16, log2T)
        np.testing.assert_array_equal(ours, g[f"{tag}_offsets"])                       # product
        off, s = orc.grid_offsets(num_levels=L, log2_hashmap_size=log2T, desired_resolution=desired)
        np.testing.assert_array_equal(off, g[f"{tag}_offsets"])                        # oracle
        assert int(g[f"{tag}_n_params"]) == int(ours[-1]) * 2
        assert list(g[f"{tag}_emb_shape"]) == [int(ours[-1]), 2]


def test_near_far_torch(golden_dir):
    g = load(golden_dir, "near_far_torch.npz")
    n, f = R.near_far_from_aabb(torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"]),
                                torch.from_numpy(g["aabb"]), float(g["min_near"]))
    np.testing.assert_array_equal(n.numpy(), g["nears"])
    np.testing.assert_array_equal(f.numpy(), g["fars"])
    assert (g["nears"] == 1e9).sum() > 0        # the fixture contains misses


@pytest.mark.parametrize("kind", ["gaussian", "planck", "hanning"])
def test_hdr_loss_weights_match_the_reference(golden_dir, kind):
    """nerf.utils.hdr_loss_weight (what the per-op Trainer and the fused step multiply the HDR residuals with) against the
    reference's raw_utils functions run on the same targets (raw/raw_utils.py:30-53, called as train_utils.py:520-527 does)."""
    g = load(golden_dir, "loss_weights.npz")
    w = U.hdr_loss_weight(kind, torch.from_numpy(g["gt_rgb"]))
    assert w.shape == g[kind].shape and not w.requires_grad
    np.testing.assert_allclose(w.numpy(), g[kind], rtol=2e-6, atol=1e-7)
    assert U.hdr_loss_weight("none", torch.from_numpy(g["gt_rgb"])) is None


def test_contract_roundtrip(golden_dir):
    g = load(golden_dir, "contract.npz")
    z = R.contract(torch.from_numpy(g["x"]))
    np.testing.assert_array_equal(z.numpy(), g["z"])
    np.testing.assert_array_equal(R.uncontract(z.clone()).numpy(), g["x_roundtrip"])
    np.testing.assert_allclose(g["x_roundtrip"], g["x"], rtol=1e-4, atol=1e-5)


def test_sample_pdf(golden_dir):
    g = load(golden_dir, "sample_pdf.npz")
    out = R.sample_pdf(torch.from_numpy(g["bins"]), torch.from_numpy(g["weights"]), int(g["T"]), perturb=False)
    np.testing.assert_array_equal(out.numpy(), g["out"])


def test_get_rays(golden_dir):
    g = load(golden_dir, "get_rays.npz")
    for i in range(2):
        r = U.get_rays(torch.from_numpy(g["poses"][i:i + 1]), g["intrinsics"], int(g["H"]), int(g["W"]), -1)
        np.testing.assert_allclose(r["rays_o"].numpy(), g[f"rays_o_{i}"], rtol=0, atol=0)
        np.testing.assert_allclose(r["rays_d"].numpy(), g[f"rays_d_{i}"], rtol=1e-6, atol=1e-7)
    # random sampling picks pixel centres of the same grid
    gen = torch.Generator().manual_seed(0)
    r = U.get_rays(torch.from_numpy(g["poses"][:1]), g["intrinsics"], int(g["H"]), int(g["W"]), 7, generator=gen)
    full = g["rays_d_0"].reshape(int(g["H"]), int(g["W"]), 3)
    np.testing.assert_allclose(r["rays_d"].numpy(), full[r["j"].numpy(), r["i"].numpy()], rtol=1e-6, atol=1e-7)


def test_trunc_exp(golden_dir):
    g = load(golden_dir, "trunc_exp.npz")
    from raw_ngp_amd.activation import trunc_exp
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = trunc_exp(x)
    (gx,) = torch.autograd.grad(y.sum(), x)
    np.testing.assert_array_equal(y.detach().numpy(), g["y"])
    np.testing.assert_array_equal(gx.numpy(), g["grad"])


@pytest.mark.parametrize("act", ["relu", "softplus"])
def test_mlp(golden_dir, act):
    g = load(golden_dir, f"mlp_{act}.npz")
    from raw_ngp_amd.nerf.network import MLP
    mlp = MLP(32, 16, 64, 3, types.SimpleNamespace(internal_activation=act, beta=2.0), bias=False)
    assert [k for k, _ in mlp.state_dict().items()] == ["net.0.weight", "net.1.weight", "net.2.weight"]
    with torch.no_grad():
        for i, p in enumerate(mlp.parameters()):
            p.copy_(torch.from_numpy(g[f"w{i}"]))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = mlp(x)
    grads = torch.autograd.grad((y * torch.from_numpy(g["gy"])).sum(), [x] + list(mlp.parameters()))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(grads[0].numpy(), g["gx"], rtol=1e-5, atol=1e-6)
    for i, gw in enumerate(grads[1:]):
        np.testing.assert_allclose(gw.numpy(), g[f"gw{i}"], rtol=1e-5, atol=1e-6)


class _FixedFeat(torch.nn.Module):
    def __init__(self, feat):
        super().__init__()
        self.feat = feat

    def forward(self, x, bound=1):
        return self.feat.clone()


@pytest.mark.parametrize("mode", ["barf", "baangp"])
def test_level_windows(golden_dir, mode):
    g = load(golden_dir, "level_windows.npz")
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    opt = Options(bound=1.0, device="cpu")
    net = NeRFNetwork(opt)
    # state-dict contract of the reference checkpoints
    keys = set(net.state_dict().keys())
    assert {"grid_encoder.embeddings", "grid_encoder.offsets", "density_grid", "density_bitfield", "aabb_train",
            "aabb_infer", "grid_mlp.net.0.weight", "grid_mlp.net.2.weight", "view_mlp.net.0.weight"} <= keys
    assert tuple(net.view_mlp.net[0].weight.shape) == (64, 31)
    opt.pose_opt = mode
    net.grid_encoder = _FixedFeat(torch.from_numpy(g["feat"]))
    seen = []
    net.grid_mlp.register_forward_pre_hook(lambda m, inp: seen.append(inp[0].detach().clone()))
    for ann in (0.0, 0.1, 0.33, 1.0):
        net.update_annealing(np.float16(ann))
        seen.clear()
        net.common_forward(torch.zeros(4, 3))
        np.testing.assert_allclose(seen[0].numpy(), g[f"{mode}_{ann}"], rtol=1e-6, atol=1e-6)


def test_oracle_barf_window_matches_the_reference_windows(golden_dir):
    """oracle.barf_window -- what the device step_window kernel is tested against -- reproduces the level weights the
    reference's common_forward applied (network.py:99-109, float16 annealing of train_utils.py:488) at annealing
    {0, 0.1, 0.33, 1}: the fixture holds features x weights for features 0.1 .. 3.2."""
    from oracle import oracle as orc
    g = load(golden_dir, "level_windows.npz")
    feat = g["feat"]
    for ann, step in ((0.0, 0), (0.1, 100), (0.33, 330), (1.0, 1000)):
        w, a16 = orc.barf_window(step, 1000, 0.0, 0.33, 16)
        assert a16 == np.float16(ann)
        np.testing.assert_allclose(feat * np.repeat(w, 2)[None, :], g[f"barf_{ann}"], rtol=1e-6, atol=1e-7)
    # (mid-window values exist in the fixture: not all weights are 0 or 1)
    w, _ = orc.barf_window(100, 1000, 0.0, 0.33, 16)
    assert np.any((w > 0.01) & (w < 0.99))


def test_run_sampler_with_analytic_field(golden_dir):
    g = load(golden_dir, "run_analytic.npz")
    from raw_ngp_amd.nerf.options import Options

    class Analytic(R.NeRFRenderer):
        def density(self, x, proposal=-1, **kw):
            return {"sigma": 30.0 * torch.exp(-3.0 * (x ** 2).sum(-1)) * (1.0 + 0.5 * (proposal + 1))}

        def forward(self, x, d, **kw):
            return {"sigma": 30.0 * torch.exp(-3.0 * (x ** 2).sum(-1)),
                    "color": torch.sigmoid(3.0 * x) * (0.75 + 0.25 * d[..., :1])}

    opt = Options(bound=1.0, cuda_ray=False, num_steps=[int(v) for v in g["num_steps"]], lambda_proposal=0.0)
    ren = Analytic(opt).eval()
    with torch.no_grad():
        out = ren.render(torch.from_numpy(g["rays_o"]), torch.from_numpy(g["rays_d"]), bg_color=None, perturb=False)
    np.testing.assert_allclose(out["image"].numpy(), g["image"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["depth"].numpy(), g["depth"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out["weights_sum"].numpy(), g["weights_sum"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag,kw", [("cuda_ray", dict(cuda_ray=True)), ("sampler", dict(cuda_ray=False)),
                                    ("cuda_ray_rfield", dict(cuda_ray=True, rfield=True))])
def test_state_dict_layout_matches_reference_checkpoints(golden_dir, tag, kw):
    """A checkpoint written by the reference (nerf/train_utils.py:1141-1180 saves model.state_dict()) loads into
    our NeRFNetwork: same keys, shapes and dtypes as the reference module built with the same options."""
    import json
    from raw_ngp_amd.nerf.network import NeRFNetwork
    from raw_ngp_amd.nerf.options import Options
    with open(os.path.join(golden_dir, "state_dict_layout.json")) as f:
        want = {k: (tuple(shape), dtype) for k, shape, dtype in json.load(f)[tag]}
    net = NeRFNetwork(Options(bound=1.0, **kw))
    got = {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in net.state_dict().items()}
    assert got == want
