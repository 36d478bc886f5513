"""Pins the host-side pieces of lbm_amd that mirror reference-owned Python — the conditioners, the conditioning
weights of LBMModel._get_conditioning, LBMModel._timestep_sampling, StateDictAdapter / StateDictRenamer — to outputs
of the reference's own classes (tests/golden/glue_golden.safetensors, timestep_sampling.json, glue_meta.json, written
by tests/golden/make_golden_glue.py).  These paths are plain PyTorch on whatever device the tensors live on, so they
are checked on the CPU here; the -m gpu twin (tests/test_glue_gpu.py) runs forward / sample with the conditioners on
the device.  Same arithmetic in the same order: exact equality unless noted."""
import json
import os

import numpy as np
import pytest
import torch
from safetensors.torch import load_file

GDIR = os.path.join(os.path.dirname(__file__), "golden")
GOLD = load_file(os.path.join(GDIR, "glue_golden.safetensors"))
with open(os.path.join(GDIR, "timestep_sampling.json")) as f:
    TS = json.load(f)
with open(os.path.join(GDIR, "glue_meta.json")) as f:
    META = json.load(f)


def sub(prefix):
    return {k[len(prefix):]: v for k, v in GOLD.items() if k.startswith(prefix)}


def build_wrapper():
    from lbm_amd.models.embedders import (ConditionerWrapper, LatentsConcatEmbedderConfig, LightingParamsEmbedder,
                                          LightingParamsEmbedderConfig, RawConcatEmbedder)
    light = LightingParamsEmbedder(LightingParamsEmbedderConfig(
        mapping_size=16, scale=2.0, hidden_dim=24, out_dim=12, seq_len=4, param_groups=[[0, 1], [2], [3, 4, 5], [6]],
        unconditional_conditioning_rate=0.5))
    light.load_state_dict(sub("cond.light_sd."), strict=True)       # key names are the reference's
    raw1 = RawConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=["img_a"], mask_keys=["msk"], input_key="img_a"))
    raw2 = RawConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=["img_b"], mask_keys=None, input_key="img_b",
                                                         unconditional_conditioning_rate=0.25))
    return ConditionerWrapper([raw1, light, raw2])


def test_lighting_embedder_construction_draws_match_reference():
    """seeded construction reproduces the reference's buffers and weights (same order of random draws)"""
    from lbm_amd.models.embedders import LightingParamsEmbedder, LightingParamsEmbedderConfig
    torch.manual_seed(5)
    light = LightingParamsEmbedder(LightingParamsEmbedderConfig(
        mapping_size=16, scale=2.0, hidden_dim=24, out_dim=12, seq_len=4, param_groups=[[0, 1], [2], [3, 4, 5], [6]]))
    ref = sub("cond.light_sd.")
    sd = light.state_dict()
    assert sorted(sd) == sorted(ref)
    for k in ref:
        assert torch.equal(sd[k], ref[k]), k


def test_conditioner_wrapper_matches_reference():
    batch = sub("cond.batch.")
    wrapper = build_wrapper()
    torch.manual_seed(77)
    res = wrapper(batch)
    assert set(res) == {"cond"} and set(res["cond"]) == {"concat", "crossattn"}
    assert torch.equal(res["cond"]["concat"], GOLD["cond.wrapper.concat"])          # mask | img_a | img_b on dim 1
    torch.testing.assert_close(res["cond"]["crossattn"], GOLD["cond.wrapper.crossattn"], rtol=1e-6, atol=1e-6)
    assert torch.equal(torch.rand(1), GOLD["cond.wrapper.next_rand"])               # three CPU draws were consumed
    torch.manual_seed(77)
    wrapper(batch, set_ucg_rate_zero=True)
    assert torch.equal(torch.rand(1), GOLD["cond.wrapper_zero.next_rand"])          # one draw (lighting embedder)
    torch.manual_seed(77)
    wrapper(batch, ucg_keys=["img_b"])
    assert torch.equal(torch.rand(1), GOLD["cond.wrapper_keys.next_rand"])          # forced key: no draw for it


def test_latents_concat_image_path_matches_reference():
    from lbm_amd.models.embedders import LatentsConcatEmbedder, LatentsConcatEmbedderConfig

    class ToyVAE:
        downsampling_factor = 8

        def encode(self, x, batch_size=8):
            return torch.einsum("oc,bchw->bohw", GOLD["cond.vae_mix"], torch.nn.functional.avg_pool2d(x, 8))

    lat = LatentsConcatEmbedder(LatentsConcatEmbedderConfig(image_keys=["big"], mask_keys=None, input_key="big"))
    res = lat(sub("cond.batch."), vae=ToyVAE())
    assert list(res) == ["concat"]
    torch.testing.assert_close(res["concat"], GOLD["cond.latents_concat.concat"], rtol=1e-6, atol=1e-6)


def test_get_conditioning_weights_match_reference():
    from lbm_amd.models.lbm import LBMConfig, LBMModel
    cfg = LBMConfig(source_key="src", target_key="tgt", lighting_condition_weight=0.7, concat_condition_weight=1.3)
    model = LBMModel(cfg, denoiser=None, conditioner=build_wrapper())
    torch.manual_seed(31)
    c = model._get_conditioning(sub("cond.batch."), set_ucg_rate_zero=True)
    torch.testing.assert_close(c["cond"]["concat"], GOLD["getcond.concat"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(c["cond"]["crossattn"], GOLD["getcond.crossattn"], rtol=1e-6, atol=1e-6)
    assert LBMModel(cfg, denoiser=None, conditioner=None)._get_conditioning({}) is None


@pytest.mark.parametrize("key", sorted(TS))
def test_timestep_sampling_matches_reference(key):
    """uniform / log_normal / custom_timesteps under fixed torch + numpy seeds (lbm_model.py:277-291)"""
    from lbm_amd.models.lbm import LBMConfig, LBMModel
    from lbm_amd.models.schedulers import FlowMatchEulerDiscreteScheduler
    mode, seed = key.rsplit(".", 1)
    rec = TS[key]
    model = LBMModel(LBMConfig(timestep_sampling=mode, **rec["config"]), denoiser=None,
                     training_noise_scheduler=FlowMatchEulerDiscreteScheduler(),
                     sampling_noise_scheduler=FlowMatchEulerDiscreteScheduler())
    torch.manual_seed(int(seed))
    np.random.seed(int(seed))
    t = model._timestep_sampling(n_samples=16)
    assert str(t.dtype) == rec["dtype"]
    assert [float(v) for v in t] == rec["values"]


@pytest.mark.parametrize("strategy", ["zeros", "normal"])
def test_state_dict_adapter_matches_reference(strategy):
    from lbm_amd.trainer.utils import StateDictAdapter
    torch.manual_seed(61)
    res = StateDictAdapter()(model_state_dict=sub("adapter.model."), checkpoint_state_dict=sub("adapter.ckpt."),
                             regex_keys=META["adapter_regex"], strategy=strategy)
    ref = sub(f"adapter.{strategy}.")
    assert sorted(res) == sorted(ref)
    for k in ref:
        assert res[k].shape == ref[k].shape, k
        assert torch.equal(res[k], ref[k]), k       # "normal": same draws in the same order, same mean / std source


def test_state_dict_renamer_matches_reference():
    from lbm_amd.trainer.utils import StateDictRenamer
    res = StateDictRenamer()(sub("adapter.ckpt."), {"vec.weight": "renamed.weight", "absent": "x"})
    assert sorted(res) == META["renamer_keys"]
    with pytest.raises(AssertionError):
        StateDictRenamer()(sub("adapter.ckpt."), {"vec.weight": "sq.weight"})
