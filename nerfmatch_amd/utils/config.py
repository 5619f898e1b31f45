"""Config helpers with the semantics of the reference's yaml->Namespace plumbing.

Mirrors (behaviour, not code) nerfmatch/utils/config.py:26-90 of the reference:
  * dict2namespace / namespace2dict : recursive conversion
  * merge_configs  : shallow "new wins" merge                      (config.py:60-62)
  * update_configs : only keys already present in the defaults     (config.py:65-71)
  * load_yaml_config: yaml file with an optional one-level `inherit` (config.py:74-90)
"""
from argparse import Namespace
from pathlib import Path


def _as_dict(conf):
    if isinstance(conf, Namespace):
        return vars(conf)
    if isinstance(conf, dict):
        return conf
    raise TypeError(f"config must be dict or Namespace, got {type(conf)}")


def dict2namespace(d):
    return Namespace(**{k: dict2namespace(v) if isinstance(v, dict) else v for k, v in d.items()})


def namespace2dict(ns):
    return {k: namespace2dict(v) if isinstance(v, Namespace) else v for k, v in vars(ns).items()}


def merge_configs(old_conf, new_conf):
    out = dict(_as_dict(old_conf))
    out.update(_as_dict(new_conf))
    return Namespace(**out)


def update_configs(defaults, new_conf):
    new = _as_dict(new_conf)
    return Namespace(**{k: new.get(k, v) for k, v in _as_dict(defaults).items()})


def load_yaml_config(cfg_path):
    import yaml

    cfg_path = Path(cfg_path)
    with open(cfg_path) as f:
        cfg = yaml.safe_load(f)
    inherit = cfg.pop("inherit", None)
    if inherit:
        with open(cfg_path.parent / inherit["path"]) as f:
            parent = yaml.safe_load(f)
        if "key" in inherit:
            parent = parent[inherit["key"]]
        clash = set(parent) & set(cfg)
        if clash:  # the reference's dict(**parent, **config) raises on duplicates too
            raise TypeError(f"duplicate keys between parent and child config: {sorted(clash)}")
        cfg = {**parent, **cfg}
    return dict2namespace(cfg), cfg
