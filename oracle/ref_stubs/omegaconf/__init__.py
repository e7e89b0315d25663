"""Stub of omegaconf — ONLY so that /root/reference (fairseq+espresso) can be imported in the
build container to generate golden fixtures (oracle/gen_golden.py).  Test infrastructure; never
imported by the product package."""
MISSING = "???"


def II(s):
    return "${" + s + "}"


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


class ListConfig(list):
    pass


class _OpenDict:
    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        return self.cfg

    def __exit__(self, *a):
        return False


def open_dict(cfg):
    return _OpenDict(cfg)


class OmegaConf:
    @staticmethod
    def create(x=None):
        return DictConfig(x or {})

    @staticmethod
    def is_config(x):
        return isinstance(x, (DictConfig, ListConfig))

    @staticmethod
    def is_dict(x):
        return isinstance(x, DictConfig)

    @staticmethod
    def is_list(x):
        return isinstance(x, ListConfig)

    @staticmethod
    def set_struct(cfg, flag):
        return None

    @staticmethod
    def to_container(cfg, resolve=False, enum_to_str=False):
        return dict(cfg) if isinstance(cfg, dict) else cfg

    @staticmethod
    def merge(*cfgs):
        out = DictConfig()
        for c in cfgs:
            out.update(c)
        return out

    @staticmethod
    def structured(x):
        return x
