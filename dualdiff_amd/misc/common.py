"""Dotted-path class loading — the drop-in hook of the reference
(magicdrive/misc/common.py:11-15 `load_module`; used at multiview_runner.py:155,160,
test_utils.py:103,143,151)."""
import importlib


def load_module(name):
    p, m = name.rsplit(".", 1)
    return getattr(importlib.import_module(p), m)


def move_to(obj, device, filter=lambda x: True):
    """magicdrive/misc/common.py:18-40."""
    import torch
    if torch.is_tensor(obj):
        return obj.to(device) if filter(obj) else obj
    if isinstance(obj, dict):
        return {k: move_to(v, device, filter) for k, v in obj.items()}
    if isinstance(obj, list):
        return [move_to(v, device, filter) for v in obj]
    if obj is None or isinstance(obj, bool):
        return obj
    raise TypeError(f"Invalid type {obj.__class__} for move_to.")
