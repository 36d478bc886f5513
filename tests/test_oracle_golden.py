"""Pins the CPU oracle (oracle/bridge_ref.py) to tensors produced by the reference's own
LBMModel.forward / LBMModel.sample / Tiler (tests/golden/bridge_golden.safetensors, made by
tests/golden/make_golden.py).  fp32 tolerance: 1e-6 relative (same arithmetic, same order)."""
import os

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

from oracle import bridge_ref as br
from oracle.diffusers_ref import FlowMatchEulerOracle

GOLD = load_file(os.path.join(os.path.dirname(__file__), "golden", "bridge_golden.safetensors"))


def toy_denoise(x, t):
    tt = t.float().view(-1, 1, 1, 1) / 1000.0
    return 0.7 * x + 0.2 * torch.tanh(x) * tt - 0.1


def close(a, b, tol=1e-6):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)) < tol


@pytest.mark.parametrize("name,loss,masked", [("l2", "l2", False), ("l1", "l1", False), ("masked", "l2", True)])
def test_training_forward_matches_reference(name, loss, masked):
    z, zs = GOLD["fwd.z"], GOLD["fwd.z_src"]
    mask = None
    if masked:
        mask = GOLD["fwd.mask"].bool().repeat(1, z.shape[1], 1, 1)
    res = br.training_forward(z, zs, GOLD[f"fwd_{name}.timestep"], GOLD[f"fwd_{name}.eps"], toy_denoise, 0.005,
                              mask=mask, loss_type=loss)
    assert close(res["noisy_sample"], GOLD[f"fwd_{name}.noisy_sample"])
    assert close(res["predicted_hr"], GOLD[f"fwd_{name}.predicted_hr"])
    assert close(res["loss"].reshape(1), GOLD[f"fwd_{name}.loss"])
    # the t == 1000 rows are exact copies of the source latent (lbm_model.py:146-148)
    for i, t in enumerate(GOLD[f"fwd_{name}.timestep"]):
        if int(t) == 1000:
            assert torch.equal(res["noisy_sample"][i], zs[i])


@pytest.mark.parametrize("n", [1, 4])
def test_sample_matches_reference(n):
    zs = GOLD["fwd.z_src"]
    eps = [GOLD[f"sample_{n}.eps{i}"] for i in range(n - 1)]
    out = br.sample(zs, toy_denoise, n, eps, 0.005)
    assert close(out, GOLD[f"sample_{n}.out"])


def test_one_step_closed_form():
    # 1 NFE: sigma grid [1, 0] => x - pred (lbm_model.py:320,346)
    zs = GOLD["fwd.z_src"]
    out = br.sample(zs, toy_denoise, 1, [], 0.005)
    assert close(out, zs - toy_denoise(zs, torch.full((zs.shape[0],), 1000.0)), 1e-6)


@pytest.mark.parametrize("n", [1, 2, 4, 20])
def test_scheduler_tables(n):
    s = FlowMatchEulerOracle()
    s.set_timesteps(np.linspace(1, 1 / n, n))
    assert torch.equal(s.sigmas, GOLD[f"sched_{n}.sigmas"])
    assert torch.equal(s.timesteps, GOLD[f"sched_{n}.timesteps"])
    assert s.sigmas[-1] == 0 and len(s.sigmas) == n + 1


def test_training_schedule_endpoints():
    s = FlowMatchEulerOracle()
    assert len(s.timesteps) == 1000 and float(s.timesteps[0]) == 1000.0 and float(s.timesteps[-1]) == 1.0
    assert float(s.sigmas[0]) == 1.0
    sig = br.get_sigmas(s, torch.tensor([250, 500, 750, 1000]))
    assert sig.shape == (4, 1, 1, 1)
    assert torch.allclose(sig.flatten(), torch.tensor([0.25, 0.5, 0.75, 1.0]))


def test_tiler_matches_reference():
    zt, proj = GOLD["tiler.z"], GOLD["tiler.proj"]

    def toy_decode(t):
        up = torch.nn.functional.interpolate(t, scale_factor=2.0, mode="nearest")
        return torch.einsum("oc,bchw->bohw", proj, up)

    merged = br.tiled_decode(zt, toy_decode, tile=(16, 16), overlap=(4, 4), scale=2)
    assert merged.shape == GOLD["tiler.merged"].shape == (1, 3, 80, 72)
    assert close(merged, GOLD["tiler.merged"], 1e-6)
    assert close(br.gaussian_weights(24, 32).float(), GOLD["tiler.weights_24x32"], 1e-6)
    # ragged grid: starts 0,12,24,36 over 40 rows (last tile 4 rows) — src/lbm/models/utils.py:69-71
    assert br.tile_starts(40, 16, 4) == [0, 12, 24, 36]
    assert br.tile_starts(16, 16, 4) == [0]
