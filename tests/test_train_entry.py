"""The config-driven training entry (lbm_amd.trainer.main) against the contract of the reference's training scripts
(examples/training/train_*.py main(**config), examples/training/train_base.py): every key of the shipped YAML files is
accepted, unknown keys are refused, the task's builder receives the model keys and the injected loader factory the data
keys, checkpoints land in <save_ckpt_path>/<task>/epoch=E-step=S.ckpt and a second run resumes from the newest one.
The YAML key lists are restated here (examples/training/config/{relight,fill_lighting,fill_lighting_simple,albedo}.yaml)."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn as nn

BASE_KEYS = ["wandb_project", "backbone_signature", "unet_input_channels", "vae_num_channels", "timestep_sampling",
             "selected_timesteps", "prob", "pixel_loss_type", "pixel_loss_weight", "latent_loss_type", "latent_loss_weight",
             "bridge_noise_sigma", "conditioning_images_keys", "conditioning_masks_keys", "source_key", "target_key",
             "train_data_root", "validation_data_root", "image_size", "num_workers", "batch_size", "learning_rate",
             "optimizer", "num_steps", "log_interval", "resume_from_checkpoint", "max_epochs", "save_interval",
             "resume_ckpt_path", "save_ckpt_path"]
YAML_KEYS = {
    "relight": BASE_KEYS + ["source_folder", "target_folder", "shading_folder", "normal_folder", "train_random_flip",
                            "train_random_scale_min", "train_random_scale_max"],
    "albedo": BASE_KEYS + ["train_random_flip", "train_random_scale_min", "train_random_scale_max"],
    "fill_lighting": BASE_KEYS + ["lighting_embedder_config"],
    "fill_lighting_simple": BASE_KEYS + ["lighting_embedder_config", "lighting_condition_weight", "concat_condition_weight"],
}


class ToyModel(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.built_with = kw
        self.denoiser = nn.Linear(4, 4)
        self.denoiser.config = {"toy": True}
        sched = SimpleNamespace(config=SimpleNamespace(num_train_timesteps=1000, shift=1.0))
        self.training_noise_scheduler = self.sampling_noise_scheduler = sched
        self.vae = None
        self.conditioner = None

    def forward(self, batch):
        return {"loss": (self.denoiser(batch["source"]) - batch["target"]).pow(2).mean()}


def test_every_shipped_yaml_key_is_a_key_of_its_task():
    from lbm_amd.trainer import TASKS, task_keys
    assert sorted(TASKS) == ["albedo", "albedo_refine", "fill_lighting", "fill_lighting_refine", "fill_lighting_simple", "relight"]
    for task, keys in YAML_KEYS.items():
        missing = [k for k in keys if k not in task_keys(task)]
        assert not missing, (task, missing)
    # defaults are the reference signatures' (train_relight.py:206-254, train_fill_lighting_refine.py)
    assert task_keys("relight")["unet_input_channels"] == 12 and task_keys("relight")["learning_rate"] == 5e-5
    assert task_keys("fill_lighting")["unet_input_channels"] == 10 and task_keys("fill_lighting_simple")["unet_input_channels"] == 6
    assert task_keys("fill_lighting_refine")["image_size"] == 1024 and "vae_num_channels" not in task_keys("fill_lighting_refine")
    assert "lighting_embedder_config" not in task_keys("relight")


def test_main_runs_a_task_from_its_config(tmp_path):
    from lbm_amd.trainer import main
    torch.manual_seed(0)
    seen = {}

    def loaders(**kw):
        seen["data"] = kw
        g = torch.Generator().manual_seed(1)
        data = [{"source": torch.randn(2, 4, generator=g), "target": torch.randn(2, 4, generator=g)} for _ in range(3)]
        return data, data[:1]

    def build(**kw):
        seen["model"] = kw
        return ToyModel(**kw)

    config = dict(
        wandb_project="lbm-relight", backbone_signature="stable-diffusion-v1-5/stable-diffusion-v1-5", unet_input_channels=12,
        vae_num_channels=4, timestep_sampling="custom_timesteps", selected_timesteps=[250, 500, 750, 1000],
        prob=[0.25, 0.25, 0.25, 0.25], pixel_loss_type="lpips", pixel_loss_weight=10.0, latent_loss_type="l2",
        latent_loss_weight=1.0, bridge_noise_sigma=0.005, conditioning_images_keys=["shading", "normal"],
        conditioning_masks_keys=[], source_key="source", target_key="target", train_data_root="/data/train",
        validation_data_root="/data/val", source_folder="source", target_folder="target", shading_folder="shading",
        normal_folder="normal", image_size=512, num_workers=4, train_random_flip=True, train_random_scale_min=0.9,
        train_random_scale_max=1.1, batch_size=1, learning_rate=1e-2, optimizer="SGD", num_steps=[1, 4], log_interval=500,
        resume_from_checkpoint=True, max_epochs=2, save_interval=2, resume_ckpt_path=None, save_ckpt_path=str(tmp_path))
    assert sorted(config) == sorted(YAML_KEYS["relight"])
    steps = []
    pipe = main("relight", get_dataloaders=loaders, build_model=build, config_yaml=dict(config),
                on_step=lambda s, out: steps.append((s, float(out["loss"]))), **config)
    # the builder got the model keys and nothing about data / optimisation; the loader factory the data keys
    assert seen["model"]["unet_input_channels"] == 12 and seen["model"]["conditioning_images_keys"] == ["shading", "normal"]
    assert seen["model"]["pixel_loss_weight"] == 10.0 and seen["model"]["selected_timesteps"] == [250, 500, 750, 1000]
    assert not {"batch_size", "learning_rate", "train_data_root", "save_ckpt_path", "max_epochs"} & set(seen["model"])
    assert seen["data"] == dict(train_data_root="/data/train", validation_data_root="/data/val", batch_size=1, image_size=512,
                                num_workers=4, source_folder="source", target_folder="target", shading_folder="shading",
                                normal_folder="normal", train_random_flip=True, train_random_scale_min=0.9,
                                train_random_scale_max=1.1)
    assert pipe.global_step == 6 and [s for s, _ in steps] == [1, 2, 3, 4, 5, 6]
    assert steps[-1][1] < steps[0][1]                                    # SGD on a fixed toy problem: the loss falls
    assert pipe.pipeline_config.trainable_params == ["denoiser.*"] and pipe.pipeline_config.optimizer_name == "SGD"
    assert pipe.pipeline_config.log_keys == ["source", "target", "shading", "normal"]
    assert pipe.concat_keys == ["source", "target", "shading", "normal", "samples_1_steps", "samples_4_steps"]
    assert pipe.run_name.endswith("-LBM-Relight")
    task_dir = tmp_path / "relight"
    names = sorted(p.name for p in task_dir.glob("*.ckpt"))
    assert names == ["epoch=0-step=2.ckpt", "epoch=1-step=4.ckpt", "epoch=1-step=6.ckpt"]
    ck = torch.load(str(task_dir / "epoch=1-step=6.ckpt"), map_location="cpu", weights_only=True)
    assert ck["global_step"] == 6 and ck["epoch"] == 1
    assert ck["hyper_parameters"]["config_yaml"]["unet_input_channels"] == 12
    assert ck["hyper_parameters"]["training"]["learning_rate"] == 1e-2
    assert ck["hyper_parameters"]["denoiser"] == {"toy": True}
    assert all(k.startswith("model.") for k in ck["state_dict"])
    # a second run of the same config resumes from the newest checkpoint of the task directory and goes on from step 6,
    # in the epoch the checkpoint was written in (epoch 1: the loader starts over, so epochs 1 and 2 run in full)
    steps2 = []
    pipe2 = main("relight", get_dataloaders=loaders, build_model=build, on_step=lambda s, out: steps2.append(s),
                 **dict(config, max_epochs=3))
    assert steps2 == [7, 8, 9, 10, 11, 12] and pipe2.global_step == 12
    for a, b in zip(pipe.model.state_dict().values(), ck["state_dict"].values()):
        assert torch.equal(a, b)
    # resume_from_checkpoint: false starts over
    steps3 = []
    main("relight", get_dataloaders=loaders, build_model=build, on_step=lambda s, out: steps3.append(s), max_steps=2,
         **dict(config, resume_from_checkpoint=False, save_ckpt_path=str(tmp_path / "fresh")))
    assert steps3 == [1, 2]


def test_main_refuses_what_the_reference_signature_refuses(tmp_path):
    from lbm_amd.trainer import main
    with pytest.raises(TypeError, match="unexpected keyword argument 'lighting_embedder_config'"):
        main("relight", build_model=ToyModel, train_loader=[], lighting_embedder_config={}, save_ckpt_path=str(tmp_path))
    with pytest.raises(TypeError, match="unexpected keyword argument 'shading_folder'"):
        main("albedo", build_model=ToyModel, train_loader=[], shading_folder="s", save_ckpt_path=str(tmp_path))
    with pytest.raises(ValueError, match="unknown task"):
        main("depth", save_ckpt_path=str(tmp_path))
    with pytest.raises(ValueError, match="no data"):
        main("albedo", build_model=ToyModel, save_ckpt_path=str(tmp_path))
    # defaults for conditioning keys the scripts fill in when the config leaves them out
    seen = {}
    main("fill_lighting", build_model=lambda **kw: seen.update(kw) or ToyModel(**kw), train_loader=[], optimizer="SGD",
         save_ckpt_path=str(tmp_path), max_epochs=1)
    assert seen["conditioning_images_keys"] == ["shading"] and seen["conditioning_masks_keys"] == ["depth"]
    assert seen["lighting_conditioning"] is True and seen["unet_input_channels"] == 10


def test_main_from_config_reads_the_yaml(tmp_path):
    import yaml
    from lbm_amd.trainer import main_from_config
    cfg = dict(unet_input_channels=4, optimizer="SGD", learning_rate=1e-2, max_epochs=1, save_interval=1,
               save_ckpt_path=str(tmp_path / "ck"), resume_from_checkpoint=False, conditioning_images_keys=[],
               conditioning_masks_keys=[])
    path = tmp_path / "albedo.yaml"
    path.write_text(yaml.safe_dump(cfg))
    g = torch.Generator().manual_seed(2)
    data = [{"source": torch.randn(2, 4, generator=g), "target": torch.randn(2, 4, generator=g)}]
    pipe = main_from_config(str(path), build_model=ToyModel, train_loader=data)
    assert pipe.global_step == 1 and pipe.hparams["config_yaml"] == cfg
    assert (tmp_path / "ck" / "albedo" / "epoch=0-step=1.ckpt").exists()
