"""YAML config loading for the drop-in path.

The reference reads its configs with OmegaConf (inference_resshift.py:77-163, sampler.py:83-106);
OmegaConf is not a dependency here.  `load_config` parses the same YAML files with PyYAML, resolves
the `${a.b.c}` interpolations they use (e.g. `${autoencoder.params.embed_dim}`) and returns a small
attribute-access mapping so that `configs.model.params.lq_size`-style code keeps working.

`resshift_amd/configs/` carries inference-only digests (model / diffusion / autoencoder sections) of
the reference task configs; a full reference YAML (with its degradation / data / train sections)
loads just the same.
"""
from __future__ import annotations

import os
import re
from typing import Any

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")
_INTERP = re.compile(r"^\$\{([^}]+)\}$")


class ConfigNode(dict):
    """dict with attribute access (OmegaConf-like, read/write)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x: Any) -> Any:
    if isinstance(x, dict):
        return ConfigNode({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _lookup(root: dict, path: str) -> Any:
    cur: Any = root
    for part in path.split("."):
        cur = cur[part]
    return cur


def _resolve(node: Any, root: dict, depth: int = 0) -> Any:
    if depth > 8:
        raise ValueError("interpolation too deep")
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _INTERP.match(node.strip())
        if m:
            return _resolve(_lookup(root, m.group(1)), root, depth + 1)
    return node


def load_config(path_or_name: str) -> ConfigNode:
    """Load `configs/<name>.yaml` shipped with the package, or any YAML path (e.g. a reference config)."""
    path = path_or_name
    if not os.path.exists(path):
        cand = os.path.join(CONFIG_DIR, path_or_name if path_or_name.endswith(".yaml") else path_or_name + ".yaml")
        if not os.path.exists(cand):
            raise FileNotFoundError(path_or_name)
        path = cand
    with open(path) as fh:
        raw = yaml.safe_load(fh)
    return _wrap(_resolve(raw, raw))


def to_plain(node: Any) -> Any:
    if isinstance(node, dict):
        return {k: to_plain(v) for k, v in node.items()}
    if isinstance(node, list):
        return [to_plain(v) for v in node]
    return node
