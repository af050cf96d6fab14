"""mm*-style backbone registry (the registry half of the drop-in boundary).

The reference registers its backbone with `@MODELS.register_module()` (mmseg/mmdet/mmpretrain/opencd/mmrotate1.x,
e.g. RS_Tasks_Finetune/Semantic_Segmentation/mmseg/models/backbones/vit_rvsa_mtp.py:577) or
`@ROTATED_BACKBONES.register_module()` (mmrotate 0.3.4) and builds it from a config dict
`dict(type='RVSA_MTP', img_size=..., ...)`.  When mmengine is importable its `Registry` is used, so the classes
appear in the host framework's own `MODELS`; otherwise a minimal local Registry with the same
`register_module()` / `build(cfg)` / `get(name)` semantics stands in.
"""
import inspect


class _LocalRegistry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, module, name=None, force=False):
        if not inspect.isclass(module) and not inspect.isfunction(module):
            raise TypeError("module must be a class or a function, got %s" % type(module))
        names = [module.__name__] if name is None else ([name] if isinstance(name, str) else list(name))
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError("%s is already registered in %s" % (n, self._name))
            self._module_dict[n] = module

    def register_module(self, name=None, force=False, module=None):
        """Usable as `@R.register_module()`, `@R.register_module(name='x')` or `R.register_module(module=cls)`."""
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, *args, **kwargs):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError("cfg must be a dict containing the key 'type'")
        cfg = dict(cfg)
        t = cfg.pop("type")
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError("%s is not in the %s registry" % (t, self._name))
        return cls(*args, **cfg, **kwargs)


def _make(name):
    try:
        from mmengine.registry import Registry  # noqa: WPS433  (optional dependency)
        return Registry(name)
    except Exception:
        return _LocalRegistry(name)


MODELS = _make("mtp_amd_models")
BACKBONES = _make("mtp_amd_backbones")
ROTATED_BACKBONES = BACKBONES


def build_backbone(cfg):
    return MODELS.build(cfg)
