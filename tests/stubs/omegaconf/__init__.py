"""Test stand-in for ``omegaconf`` (absent from this image): attribute-access dictionaries, ``.get`` and ``to_yaml`` --
exactly what reference src/scripts/train.py / evaluate.py use of it."""
import re

import yaml

# PyYAML (YAML 1.1) reads ``5e-4`` as a string; omegaconf's own grammar reads it as a float
_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)$")


class DictConfig(dict):
    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = _wrap(v)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = _wrap(value)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, DictConfig):
        return DictConfig(v)
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    if isinstance(v, str) and _FLOAT.match(v):
        return float(v)
    return v


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_plain(x) for x in v]
    return v


class OmegaConf:
    @staticmethod
    def create(data):
        return DictConfig(data)

    @staticmethod
    def to_yaml(cfg):
        return yaml.safe_dump(_plain(cfg), sort_keys=False)
