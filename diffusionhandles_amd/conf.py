"""Minimal attribute-style config (OmegaConf is not required; a DictConfig works too)."""
import os

import yaml


class Conf(dict):
    """dict with attribute access, nested."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        if isinstance(d, dict) and not isinstance(d, Conf):
            return Conf({k: Conf.wrap(v) for k, v in d.items()})
        return d


def load_default():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "default.yaml")
    return load(path)


def load(path):
    with open(path) as f:
        return Conf.wrap(yaml.safe_load(f))
