"""Checkpoint compatibility with the reference.

The reference saves plain ``model.state_dict()`` files - ``latest.pth`` every epoch (train.py:247 via
utils.py:7-11) and ``best.pth`` from the early stopper (train.py:97) - resumes with ``strict=False``
(train.py:323-326), and loads pretrained sub-models through small renaming loaders: ``module.`` prefixes stripped
(avformer.py:28-35), ``base_model.`` -> ``s_former.`` (vformer.py:344-352, tformer.py:349-357), optionally under a
``'state_dict'`` key (vformer.py:333-342).  Because the HIP modules keep the reference's parameter names and shapes,
those files load as they are; this module provides the loaders so a maintainer does not need the reference's copies
(one of which drops the ``module.`` strip by overwriting ``new_name``, vformer.py:338-339 - both renames apply here).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Iterable, Optional, Tuple

import torch

DEFAULT_RENAMES = (("module.", ""), ("base_model.", "s_former."))


def remap_state_dict(sd: Dict[str, torch.Tensor], renames: Iterable[Tuple[str, str]] = DEFAULT_RENAMES) -> "OrderedDict":
    out = OrderedDict()
    for k, v in sd.items():
        for old, new in renames:
            k = k.replace(old, new)
        out[k] = v
    return out


def load_pretrain(model: torch.nn.Module, weight_path: Optional[str], renames=DEFAULT_RENAMES, key: Optional[str] = None,
                  freeze: bool = False):
    """Load a reference checkpoint into ``model`` with ``strict=False``.  Returns the ``load_state_dict`` result, or
    ``None`` when the file is absent (the reference hard-codes ``K:\\...`` paths and crashes; here it is skipped)."""
    if not weight_path or not os.path.exists(weight_path):
        return None
    sd = torch.load(weight_path, map_location="cpu")
    if key is None and isinstance(sd, dict) and "state_dict" in sd and not torch.is_tensor(sd["state_dict"]):
        key = "state_dict"  # vformer.py:335
    if key is not None:
        sd = sd[key]
    res = model.load_state_dict(remap_state_dict(sd, renames), strict=False)
    if freeze:  # avformer.py:80-85
        for p in model.parameters():
            p.requires_grad = False
    return res


def save_checkpoint(state, filepath: str = "./weights", filename: str = "latest.pth") -> str:
    """reference utils.py:7-11"""
    os.makedirs(filepath, exist_ok=True)
    path = os.path.join(filepath, filename)
    torch.save(state, path)
    return path


def resume(model: torch.nn.Module, checkpoint_dir: str, filename: str = "latest.pth"):
    """reference train.py:323-326: ``latest.pth`` under ``<exp_dir>/pretrain`` with strict=False, if present"""
    return load_pretrain(model, os.path.join(checkpoint_dir, filename), renames=())
