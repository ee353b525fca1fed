"""Loader for mmcv-style python config files (``mmcv.Config.fromfile`` as used at
``tools/train.py:89``): executes the file, resolves ``_base_`` (str or list, relative to the
file), merges child over base recursively, honours ``_delete_=True`` and ``--cfg-options``
style dotted overrides.  Enough of it to load ``configs/swin/*.py`` unchanged.
"""
import copy
import os

BASE_KEY = '_base_'
DELETE_KEY = '_delete_'


class ConfigDict(dict):
    """dict with attribute access (the subset of addict.Dict the reference relies on)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


def _merge_a_into_b(a, b):
    b = dict(b)
    for k, v in a.items():
        if isinstance(v, dict) and k in b and not v.get(DELETE_KEY, False):
            if not isinstance(b[k], dict):
                raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} is a dict in the '
                                f'child config but is of type {type(b[k])} in base config; set {DELETE_KEY}=True')
            b[k] = _merge_a_into_b(v, b[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            b[k] = v
    return b


def _file2dict(filename):
    filename = os.path.abspath(os.path.expanduser(filename))
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    if not filename.endswith('.py'):
        raise IOError('Only py type is supported')
    ns = {'__file__': filename}
    with open(filename) as f:
        exec(compile(f.read(), filename, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    if BASE_KEY in cfg:
        base = cfg.pop(BASE_KEY)
        base = base if isinstance(base, list) else [base]
        merged = {}
        for b in base:
            bd = _file2dict(os.path.join(os.path.dirname(filename), b))
            dup = merged.keys() & bd.keys()
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {dup}')
            merged.update(bd)
        cfg = _merge_a_into_b(cfg, merged)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_file2dict(filename), filename)

    def merge_from_dict(self, options):
        """``--cfg-options a.b=c`` (tools/train.py:55-64, 91)."""
        nested = {}
        for full_key, v in options.items():
            d = nested
            keys = full_key.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge_a_into_b(nested, self._cfg_dict)))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, k, default=None):
        return self._cfg_dict.get(k, default)

    def to_dict(self):
        return copy.deepcopy(self._cfg_dict)
