from .fit import checkpoint_name, fit, resolve_resume_checkpoint, resolve_task_dir
from .training_config import TrainingConfig
from .utils import StateDictAdapter, StateDictRenamer

__all__ = ["TrainingConfig", "StateDictAdapter", "StateDictRenamer", "TrainingPipeline", "fit", "resolve_task_dir",
           "resolve_resume_checkpoint", "checkpoint_name", "main", "main_from_config", "task_keys", "TASKS"]


def __getattr__(name):
    # the pipeline pulls in the HIP library; keep `import lbm_amd.trainer` usable for the config classes alone
    if name == "TrainingPipeline":
        from .trainer import TrainingPipeline
        return TrainingPipeline
    if name in ("main", "main_from_config", "task_keys", "TASKS"):
        import importlib
        return getattr(importlib.import_module(__name__ + ".entry"), name)
    raise AttributeError(name)
