"""Stub of omegaconf — ONLY so that /root/reference (fairseq+espresso) can be imported in the
build container to generate golden fixtures (oracle/gen_golden.py).  Test infrastructure; never
imported by the product package."""
MISSING = "???"


def II(s):
    return "${" + s + "}"


class DictConfig(dict):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.__dict__["_parent"] = None  # fairseq/dataclass/utils.py:501 copies this slot

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
        """Shallow-recursive merge; a dataclass instance contributes its fields (fairseq merges a user's config INTO the
        registered dataclass's defaults: fairseq/dataclass/utils.py:487-503)."""
        import dataclasses

        def plain(c):
            if dataclasses.is_dataclass(c) and not isinstance(c, type):
                return DictConfig({f.name: plain(getattr(c, f.name)) for f in dataclasses.fields(c)})
            if isinstance(c, dict) and not isinstance(c, DictConfig):
                return DictConfig({k: plain(v) for k, v in c.items()})
            return c

        out = DictConfig()
        for c in cfgs:
            for k, v in plain(c).items():
                if isinstance(v, dict) and isinstance(out.get(k), dict):
                    out[k] = OmegaConf.merge(out[k], v)
                else:
                    out[k] = plain(v)
        return out

    @staticmethod
    def structured(x):
        return x
