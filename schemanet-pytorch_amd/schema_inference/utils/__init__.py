"""`schema_inference.utils` -- only the hot-path member (IngredientModelWrapper) and the
trivial `move_data_to_device` are implemented here.  The reference's orchestration helpers
(LogArgs, DistLaunchArgs, load_pretrain_model, customs_param_group; reference
schema_inference/utils/__init__.py:5-8) depend on the un-vendored `cv_lib`; when the reference
checkout is on sys.path behind this package they are resolved lazily from there.
"""
import importlib
from pkgutil import extend_path
from typing import Dict, Tuple

import torch

from .ingredient_model_wrapper import IngredientModelWrapper

__path__ = extend_path(__path__, __name__)

_LAZY = {
    "LogArgs": "dist_utils", "DistLaunchArgs": "dist_utils",
    "load_pretrain_model": "model", "customs_param_group": "customs_param_group",
}


def __getattr__(name):
    if name in _LAZY:
        mod = importlib.import_module(f"{__name__}.{_LAZY[name]}")   # found via the extended __path__
        obj = getattr(mod, name)
        globals()[name] = obj       # like `from .x import x`: rebinds a name the submodule import shadowed
        return obj
    raise AttributeError(name)


def move_data_to_device(x: torch.Tensor, targets: Dict[str, torch.Tensor], device: torch.device
                        ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    x = x.to(device)
    if targets is not None:
        for k, v in targets.items():
            if isinstance(v, torch.Tensor):
                targets[k] = v.to(device)
    return x, targets
