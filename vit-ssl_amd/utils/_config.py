"""Tiny helpers to read Hydra/OmegaConf-style configs, plain dicts or attribute bags."""


def cfg_get(cfg, *path, default=None):
    cur = cfg
    for key in path:
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(key, None)
        elif hasattr(cur, "get") and not hasattr(cur, key):
            cur = cur.get(key, None)
        else:
            cur = getattr(cur, key, None)
    return default if cur is None else cur


def to_plain(cfg):
    try:
        from omegaconf import OmegaConf
        if OmegaConf.is_config(cfg):
            return OmegaConf.to_container(cfg, resolve=True)
    except Exception:
        pass
    return cfg
