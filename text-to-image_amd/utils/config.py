"""YAML -> attribute dict (reference utils/config.py:5-7: yaml.load -> EasyDict).  Keys are kept verbatim
(reference models/wgancls/cfg/flowers.yml:9-37)."""
import yaml


class AttrDict(dict):
    """Minimal EasyDict: nested dicts become attribute-accessible."""

    def __init__(self, d=None):
        super(AttrDict, self).__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super(AttrDict, self).__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def config_from_yaml(file_path):
    with open(file_path, 'r') as f:
        return AttrDict(yaml.safe_load(f))
