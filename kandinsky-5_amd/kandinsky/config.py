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


DEFAULT_CONFIG_NAMES = ("config_5s_sft.yaml", "config_5s_nocfg.yaml", "config_5s_pretrain.yaml", "config_5s_distil.yaml",
                        "config_10s_sft.yaml", "config_10s_nocfg.yaml", "config_10s_pretrain.yaml", "config_10s_distil.yaml")


def default_configs():
    """The eight model / sampling configurations of the reference's schema, kept as ONE data file
    (kandinsky/default_configs.json: checkpoint-relative paths, dit_params, attention incl. NABLA windows, num_steps,
    guidance_weight, MagCache ratio tables)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "default_configs.json")) as f:
        return json.load(f)


def write_default_configs(directory):
    """Materialise the default configurations as `<directory>/config_*.yaml` (the files the CLI's --config default and the
    README launch lines name).  Existing files are left alone."""
    import os
    os.makedirs(directory, exist_ok=True)
    written = []
    for name, data in default_configs().items():
        path = os.path.join(directory, name)
        if not os.path.exists(path):
            with open(path, "w") as f:
                yaml.safe_dump(data, f, sort_keys=False)
            written.append(path)
    return written


def load_config(path):
    """YAML config by path; a missing `.../config_<name>.yaml` whose name is one of the defaults is created first."""
    import os
    if not os.path.exists(path) and os.path.basename(path) in DEFAULT_CONFIG_NAMES:
        write_default_configs(os.path.dirname(os.path.abspath(path)))
    with open(path) as f:
        return Conf(yaml.safe_load(f))
