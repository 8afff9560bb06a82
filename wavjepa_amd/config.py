"""YAML config tree + `group=name` / `a.b=value` overrides (the subset of Hydra the reference's train.py uses:
configs/base.yaml defaults list, group selection, dotted value overrides; reference train.py:225, train.sh:18)."""
from __future__ import annotations

import ast
import os
from typing import Any, Dict, List

import yaml


class Cfg(dict):
    """dict with attribute access and .get, like an OmegaConf node."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    return x


def _parse_value(s: str) -> Any:
    low = s.lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none"):
        return None
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        return s


def load_config(config_dir: str, overrides: List[str] = (), config_name: str = "base") -> Cfg:
    with open(os.path.join(config_dir, f"{config_name}.yaml")) as fh:
        base = yaml.safe_load(fh)
    groups: Dict[str, str] = {}
    for item in base.pop("defaults", []):
        (g, name), = item.items()
        groups[g.strip()] = str(name).strip()
    values = []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' must look like key=value")
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        if "." not in k and k in groups and os.path.exists(os.path.join(config_dir, k, f"{v}.yaml")):
            groups[k] = v
        else:
            values.append((k, v))
    cfg: Dict[str, Any] = dict(base)
    for g, name in groups.items():
        with open(os.path.join(config_dir, g, f"{name}.yaml")) as fh:
            cfg[g] = yaml.safe_load(fh)
    for k, v in values:
        node = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = _parse_value(v)
    return _wrap(cfg)


def parse_conv_spec(spec) -> list:
    """The reference eval()s the spec string (train.py:62); here only a literal list expression is evaluated."""
    if not isinstance(spec, str):
        return [tuple(x) for x in spec]
    tree = ast.parse(spec, mode="eval")
    for node in ast.walk(tree):
        if not isinstance(node, (ast.Expression, ast.BinOp, ast.Add, ast.Mult, ast.List, ast.Tuple, ast.Constant, ast.Load)):
            raise ValueError(f"unsupported expression in conv_layers_spec: {spec}")
    return [tuple(int(v) for v in t) for t in eval(compile(tree, "<conv_layers_spec>", "eval"), {"__builtins__": {}})]
