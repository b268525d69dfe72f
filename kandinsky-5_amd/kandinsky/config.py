"""Minimal attribute-dict config loader for the reference's YAML schema.

omegaconf (used by the reference, kandinsky/utils.py:8,92) is not a dependency here: PyYAML +
`Conf` give the same `conf.model.dit_params.patch_size` / `conf["model"]` access the code needs.
"""
import yaml


class Conf(dict):
    """dict with attribute access, recursively applied (subset of OmegaConf DictConfig behaviour)."""

    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Conf):
            return Conf(v)
        if isinstance(v, (list, tuple)):
            return [Conf._wrap(i) for i in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Conf._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        def un(v):
            if isinstance(v, Conf):
                return {k: un(x) for k, x in v.items()}
            if isinstance(v, list):
                return [un(x) for x in v]
            return v
        return un(self)


def load_config(path):
    with open(path) as f:
        return Conf(yaml.safe_load(f))
