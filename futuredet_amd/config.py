"""Config loader: python-module configs -> attribute-access dict.

Follows det3d/torchie/utils/config.py:12-29 (ConfigDict: missing key raises
KeyError on item access and AttributeError on attribute access) and :77-100
(Config.fromfile imports the file as a module and keeps every name that does
not start with "__").  The reference builds ConfigDict on the third-party
``addict.Dict``; that package is not part of this image, so the nested
attribute dict is implemented here.
"""
import os.path as osp
import sys
from importlib import import_module, invalidate_caches


def _wrap(value):
    if isinstance(value, ConfigDict):
        return value
    if isinstance(value, dict):
        return ConfigDict(value)
    if isinstance(value, (list, tuple)):
        return type(value)(_wrap(v) for v in value)
    return value


class ConfigDict(dict):
    """dict with attribute access; nested dicts (also inside lists) are wrapped."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        super().__setitem__(key, _wrap(value))

    def __setattr__(self, key, value):
        self[key] = value

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(
                "'{}' object has no attribute '{}'".format(type(self).__name__, name)
            )

    def __delattr__(self, name):
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name)

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def copy(self):
        return ConfigDict(self)

    def to_dict(self):
        def plain(v):
            if isinstance(v, dict):
                return {k: plain(x) for k, x in v.items()}
            if isinstance(v, (list, tuple)):
                return type(v)(plain(x) for x in v)
            return v

        return plain(self)


class Config(object):
    """Same surface as the reference's Config: ``Config.fromfile(path)``,
    attribute / item access into the parsed dict, ``.filename``, ``.text``."""

    @staticmethod
    def fromfile(filename):
        from . import compat

        compat.install_det3d_alias()  # config files import det3d.utils.config_tool
        filename = osp.abspath(osp.expanduser(filename))
        if not osp.isfile(filename):
            raise FileNotFoundError('file "{}" does not exist'.format(filename))
        if not filename.endswith(".py"):
            raise IOError("Only py type configs are supported on this path")
        module_name = osp.basename(filename)[:-3]
        if "." in module_name:
            raise ValueError("Dots are not allowed in config file path.")
        sys.path.insert(0, osp.dirname(filename))
        try:
            invalidate_caches()
            sys.modules.pop(module_name, None)
            prev = sys.dont_write_bytecode
            sys.dont_write_bytecode = True  # config dirs may be read-only
            try:
                mod = import_module(module_name)
            finally:
                sys.dont_write_bytecode = prev
        finally:
            sys.path.pop(0)
        cfg_dict = {k: v for k, v in mod.__dict__.items() if not k.startswith("__")}
        return Config(cfg_dict, filename=filename)

    def __init__(self, cfg_dict=None, filename=None):
        if cfg_dict is None:
            cfg_dict = {}
        elif not isinstance(cfg_dict, dict):
            raise TypeError("cfg_dict must be a dict, but got {}".format(type(cfg_dict)))
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict))
        object.__setattr__(self, "_filename", filename)
        text = ""
        if filename:
            with open(filename, "r") as f:
                text = f.read()
        object.__setattr__(self, "_text", text)

    @property
    def filename(self):
        return self._filename

    @property
    def text(self):
        return self._text

    def __repr__(self):
        return "Config (path: {}): {}".format(self.filename, dict.__repr__(self._cfg_dict))

    def __len__(self):
        return len(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __iter__(self):
        return iter(self._cfg_dict)

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)
