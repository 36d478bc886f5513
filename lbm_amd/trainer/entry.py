"""Config-driven training entry: what ``main(**config)`` of the reference's training scripts does
(examples/training/train_relight.py:206-337, train_albedo.py, train_albedo_refine.py, train_fill_lighting.py,
train_fill_lighting_simple.py, train_fill_lighting_refine.py; shared parts in examples/training/train_base.py:21-172),
for the six tasks, with the same key set as their YAML files: build the model with the task's builder, wrap it in a
``TrainingPipeline`` with the task's trainable-parameter patterns, resolve the task directory and the checkpoint to
resume from, and run the fit loop (periodic ``epoch=E-step=S.ckpt`` files).

Differences, all forced by what is out of scope or absent here:
  * the folder datasets (PIL / torchvision disk I/O) are not part of this package: the loaders are INJECTED, either as
    ``train_loader=`` / ``validation_loader=`` or as a ``get_dataloaders=`` callable that receives the task's data keys
    (``train_data_root``, ``batch_size``, ``image_size``, ``num_workers``, the folder names, the augmentation switches);
  * PyTorch Lightning and W&B are not installed: the loop is ``lbm_amd.trainer.fit`` (one process per GPU; under
    ``torch.distributed.run`` the gradient exchange runs over RCCL), ``wandb_project`` / ``log_interval`` are accepted
    and only recorded, ``devices`` / ``num_nodes`` are checked against the launcher's world size;
  * a hub ``backbone_signature`` cannot be downloaded: pass a local diffusers-layout directory, or
    ``allow_random_init=True`` to train from an untrained initialisation.
An unknown key raises ``TypeError`` exactly like calling the reference's ``main`` with it would.
"""
import datetime
import inspect
import logging
from typing import Any, Callable, Dict, Iterable, List, Optional

from .fit import fit, resolve_resume_checkpoint, resolve_task_dir
from .training_config import TrainingConfig

# keys every task's main() takes (train_relight.py:206-254 and its siblings)
_COMMON_KEYS = dict(
    train_data_root="path/to/train", validation_data_root="path/to/validation",
    backbone_signature="runwayml/stable-diffusion-v1-5", source_key="source", target_key="target", mask_key=None,
    wandb_project=None, batch_size=8, num_steps=[1, 2, 4], learning_rate=5e-5, learning_rate_scheduler=None,
    learning_rate_scheduler_kwargs={}, optimizer="AdamW", optimizer_kwargs={}, timestep_sampling="uniform",
    logit_mean=0.0, logit_std=1.0, pixel_loss_type="lpips", latent_loss_type="l2", latent_loss_weight=1.0,
    pixel_loss_weight=0.0, selected_timesteps=None, prob=None, conditioning_images_keys=None,
    conditioning_masks_keys=None, image_size=512, num_workers=4, config_yaml=None, save_ckpt_path="./checkpoints",
    log_interval=100, resume_from_checkpoint=True, max_epochs=100, bridge_noise_sigma=0.005, save_interval=1000,
    resume_ckpt_path=None, devices=None, num_nodes=1)
_LATENT_KEYS = dict(vae_num_channels=4)
_AUGMENT_KEYS = dict(train_random_flip=True, train_random_scale_min=1.0, train_random_scale_max=1.0)
_LIGHTING_KEYS = dict(lighting_embedder_config=None, lighting_condition_weight=1.0, concat_condition_weight=1.0)
# keys that only describe where the data lives / how it is loaded: handed to get_dataloaders, never to the model
_DATA_KEYS = ("train_data_root", "validation_data_root", "batch_size", "image_size", "num_workers", "source_folder",
              "target_folder", "shading_folder", "normal_folder", "train_random_flip", "train_random_scale_min",
              "train_random_scale_max", "robedo_suffix", "robedo_extension")

# per task: (checkpoint sub-directory, builder name, run label, trainable patterns, log keys, extra keys with defaults,
# defaults that differ from _COMMON_KEYS, default conditioning keys when the config gives None)
TASKS: Dict[str, Dict[str, Any]] = {
    "relight": dict(
        dir="relight", builder="build_relight_model", label="Relight", trainable=["denoiser.*"],
        log_keys=["source", "target", "shading", "normal"],
        extra={**_LATENT_KEYS, **_AUGMENT_KEYS, "unet_input_channels": 12, "source_folder": "source",
               "target_folder": "target", "shading_folder": "shading", "normal_folder": "normal", "path_config": None},
        defaults=dict(wandb_project="lbm-relight"), cond_defaults={}),
    "albedo": dict(
        dir="albedo", builder="build_relight_model", label="Albedo", trainable=["denoiser.*"], log_keys=["source", "target"],
        extra={**_LATENT_KEYS, **_AUGMENT_KEYS, "unet_input_channels": 4, "path_config": None},
        defaults=dict(wandb_project="lbm-albedo", conditioning_images_keys=[], conditioning_masks_keys=[]), cond_defaults={}),
    "albedo_refine": dict(
        dir="albedo_refine", builder="build_relight_model", label="Albedo-Refine", trainable=["denoiser.*"],
        log_keys=["source", "target", "rgb"],
        extra={**_LATENT_KEYS, **_AUGMENT_KEYS, "unet_input_channels": 8, "robedo_suffix": "_rlb", "robedo_extension": ".png"},
        defaults=dict(wandb_project="lbm-albedo-refine", conditioning_masks_keys=[]),
        cond_defaults=dict(conditioning_images_keys=["rgb"])),
    "fill_lighting": dict(
        dir="fill_lighting", builder="build_filllight_model", label="Fill-Lighting",
        trainable=["denoiser.*", "conditioner.*"], log_keys=["source", "target", "shading", "depth"],
        extra={**_LATENT_KEYS, **_LIGHTING_KEYS, "unet_input_channels": 10},
        defaults=dict(wandb_project="lbm-fill-lighting"),
        cond_defaults=dict(conditioning_images_keys=["shading"], conditioning_masks_keys=["depth"])),
    "fill_lighting_simple": dict(
        dir="fill_lighting_simple", builder="build_filllight_model", label="Fill-Lighting-Simple",
        trainable=["denoiser.*", "conditioner.*"], log_keys=["source", "target", "depth"],
        extra={**_LATENT_KEYS, **_LIGHTING_KEYS, "unet_input_channels": 6},
        defaults=dict(wandb_project="lbm-fill-lighting-simple"),
        cond_defaults=dict(conditioning_images_keys=[], conditioning_masks_keys=["depth"])),
    "fill_lighting_refine": dict(
        dir="fill_lighting_refine", builder="build_filllight_refine_model", label="Fill-Lighting-Refine",
        trainable=["denoiser.*"], log_keys=["source", "target", "rgb"],
        extra={"unet_input_channels": 6, "unet_output_channels": 3, "block_out_channels": None, "layers_per_block": 2,
               "train_random_flip": True},
        defaults=dict(wandb_project="lbm-fill-lighting-refine", backbone_signature="stable-diffusion-v1-5/stable-diffusion-v1-5",
                      batch_size=1, num_steps=(1, 2, 4), pixel_loss_weight=1.0, image_size=1024),
        cond_defaults=dict(conditioning_images_keys=["rgb"], conditioning_masks_keys=[])),
}


def task_keys(task: str) -> Dict[str, Any]:
    """every key ``main(task, **config)`` accepts for ``task``, with its default (the signature of the reference's main)"""
    spec = TASKS[task]
    return {**_COMMON_KEYS, **spec["extra"], **spec["defaults"]}


def create_run_name(task_label: str) -> str:
    return f"{datetime.datetime.now().strftime('%Y-%m-%d_%H-%M-%S')}-LBM-{task_label}"


def build_concat_keys(training_config: TrainingConfig) -> List[str]:
    """columns of the sample grids (train_base.py:85-100): the logged batch keys, then one column per sampling depth"""
    keys = list(training_config.log_keys or []) if not isinstance(training_config.log_keys, str) else [training_config.log_keys]
    num_steps = training_config.log_samples_model_kwargs.get("num_steps", [])
    for step in ([num_steps] if isinstance(num_steps, int) else num_steps):
        keys.append(f"samples_{step}_steps")
    return list(dict.fromkeys(keys))


def setup_pipeline(model, training_config: TrainingConfig, config_yaml: Optional[dict], dist=None, **pipeline_kwargs):
    """TrainingPipeline + the hyper-parameter record the reference saves with every checkpoint (train_base.py:56-82)"""
    from .trainer import TrainingPipeline
    pipeline = TrainingPipeline(model=model, pipeline_config=training_config, dist=dist, **pipeline_kwargs)
    hp: Dict[str, Any] = {}
    if getattr(model, "conditioner", None) is not None and getattr(model.conditioner, "conditioners", None):
        for i, embedder in enumerate(model.conditioner.conditioners):
            hp[f"embedder_{i}"] = embedder.config.to_dict()
    hp.update(denoiser=getattr(model.denoiser, "config", None), config_yaml=config_yaml, training=training_config.to_dict(),
              training_noise_scheduler=dict(vars(model.training_noise_scheduler.config)),
              sampling_noise_scheduler=dict(vars(model.sampling_noise_scheduler.config)))
    if getattr(model, "vae", None) is not None:
        hp["vae"] = model.vae.config.to_dict()
    pipeline.hparams = hp
    return pipeline


def main(task: str = "relight", *, train_loader: Optional[Iterable] = None, validation_loader: Optional[Iterable] = None,
         get_dataloaders: Optional[Callable] = None, dist=None, allow_random_init: bool = False, device=None,
         max_steps: Optional[int] = None, on_step: Optional[Callable] = None, to_device: Optional[Callable] = None,
         build_model: Optional[Callable] = None, **config):
    """Runs one training task from a config mapping (the parsed YAML).  Returns the pipeline (its ``global_step``,
    ``hparams`` and ``model`` are what a caller may want next).  Keyword-only arguments before ``**config`` are this
    package's additions; everything in ``config`` must be a key of the reference's ``main`` for the task."""
    if task not in TASKS:
        raise ValueError(f"unknown task '{task}' (one of {sorted(TASKS)})")
    spec = TASKS[task]
    known = task_keys(task)
    unknown = [k for k in config if k not in known]
    if unknown:
        raise TypeError(f"main() got an unexpected keyword argument '{unknown[0]}' (task '{task}')")
    cfg = {**known, **config}
    for k, v in spec["cond_defaults"].items():
        if cfg.get(k) is None:
            cfg[k] = list(v)

    if dist is not None:
        world = dist.get_world_size()
        asked = cfg["devices"] * (cfg["num_nodes"] or 1) if cfg["devices"] is not None else None
        if asked is not None and asked != world:
            raise ValueError(f"config asks for devices x num_nodes = {asked} processes, the launcher started {world}")

    task_dir = resolve_task_dir(cfg["save_ckpt_path"], spec["dir"])

    # ---- model: the task's builder with exactly the keys the reference passes it
    from ..inference import relight as builders
    builder = build_model if build_model is not None else getattr(builders, spec["builder"])
    model_keys = ["backbone_signature", "unet_input_channels", "source_key", "target_key", "mask_key", "timestep_sampling",
                  "logit_mean", "logit_std", "pixel_loss_type", "latent_loss_type", "latent_loss_weight", "pixel_loss_weight",
                  "selected_timesteps", "prob", "conditioning_images_keys", "conditioning_masks_keys", "bridge_noise_sigma"]
    if "vae_num_channels" in known:
        model_keys.append("vae_num_channels")
    if spec["builder"] == "build_filllight_model":
        model_keys += ["lighting_embedder_config", "lighting_condition_weight", "concat_condition_weight"]
    if spec["builder"] == "build_filllight_refine_model":
        model_keys += ["unet_output_channels", "block_out_channels", "layers_per_block"]
        if cfg["block_out_channels"] is None:
            cfg["block_out_channels"] = [64, 128, 256, 256]
    model_kwargs = {k: cfg[k] for k in model_keys}
    if spec["builder"] == "build_filllight_model":
        model_kwargs["lighting_conditioning"] = True
    if build_model is None or "allow_random_init" in inspect.signature(builder).parameters:
        model_kwargs["allow_random_init"] = allow_random_init
    model = builder(**model_kwargs)
    if device is not None:
        model = model.to(device)

    # ---- data: injected
    if train_loader is None:
        if get_dataloaders is None:
            raise ValueError("no data: pass train_loader= (and validation_loader=), or get_dataloaders=, a callable taking the "
                             "task's data keys (the folder datasets of the reference's scripts are not part of this package)")
        train_loader, validation_loader = get_dataloaders(**{k: cfg[k] for k in _DATA_KEYS if k in known})

    # ---- pipeline (train_relight.py:292-308 and siblings)
    input_shape = (3, cfg["image_size"], cfg["image_size"]) if task == "fill_lighting_refine" else None
    training_config = TrainingConfig(
        learning_rate=cfg["learning_rate"], lr_scheduler_name=cfg["learning_rate_scheduler"],
        lr_scheduler_kwargs=cfg["learning_rate_scheduler_kwargs"], log_keys=list(spec["log_keys"]),
        trainable_params=list(spec["trainable"]), optimizer_name=cfg["optimizer"], optimizer_kwargs=cfg["optimizer_kwargs"],
        log_samples_model_kwargs={"input_shape": input_shape, "num_steps": list(cfg["num_steps"])})
    pipeline = setup_pipeline(model, training_config, cfg["config_yaml"], dist=dist)
    pipeline.run_name = create_run_name(spec["label"])
    pipeline.concat_keys = build_concat_keys(training_config)
    pipeline.validation_loader = validation_loader
    ckpt_path = resolve_resume_checkpoint(task_dir, cfg["resume_from_checkpoint"], cfg["resume_ckpt_path"])
    logging.info("task %s: run %s, checkpoints under %s, resuming from %s", task, pipeline.run_name, task_dir, ckpt_path)
    fit(pipeline, train_loader, max_epochs=cfg["max_epochs"], save_dir=task_dir, save_interval=cfg["save_interval"],
        ckpt_path=ckpt_path, to_device=to_device, on_step=on_step, max_steps=max_steps)
    return pipeline


def main_from_config(path_config: str, task: Optional[str] = None, **kwargs):
    """``main_from_config`` of the reference's scripts: reads the YAML, passes it both as the keys and as ``config_yaml``.
    ``task`` defaults to the file's stem (``relight.yaml`` -> "relight")."""
    import os

    import yaml
    with open(path_config, "r") as f:
        config = yaml.safe_load(f)
    if task is None:
        task = os.path.splitext(os.path.basename(path_config))[0]
    extra = {"path_config": path_config} if "path_config" in task_keys(task) else {}
    return main(task, **kwargs, **config, config_yaml=config, **extra)
