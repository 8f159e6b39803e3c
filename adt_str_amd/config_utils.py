"""YAML configuration: default (+) experiment deep merge, as the reference does with OmegaConf
(``utils/config_utils.py:4-14``): mappings merge recursively, everything else (scalars, lists) is replaced."""
from __future__ import annotations

import os
from typing import Any, Dict

import yaml


def load_config_from_yaml(config_path: str) -> Dict[str, Any]:
    with open(config_path, "r") as fh:
        return yaml.safe_load(fh) or {}


def deep_merge_dicts(base: Dict[str, Any], override: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for key, val in (override or {}).items():
        if isinstance(val, dict) and isinstance(out.get(key), dict):
            out[key] = deep_merge_dicts(out[key], val)
        else:
            out[key] = val
    return out


DEFAULT_CONFIG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "config_default.yaml")


def load_merged(config_path: str, default_path: str = DEFAULT_CONFIG) -> Dict[str, Any]:
    return deep_merge_dicts(load_config_from_yaml(default_path), load_config_from_yaml(config_path))
