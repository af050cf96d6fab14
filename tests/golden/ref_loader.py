"""Import the reference backbone (THIS CONTAINER ONLY) with in-memory stubs.

The reference file /root/reference/Multi-Task_Pretrain/backbone/vit_win_rvsa_v3_wsz7.py
needs four symbols that are not installed here:
  timm.models.layers.{drop_path, to_2tuple, trunc_normal_}  (VIT:22)
  mmengine.dist.get_dist_info                                 (VIT:24)
They touch init/RNG helpers only, none of the arithmetic under test.

This module is used only by make_golden.py; it is never imported by tests,
bench.py or the product (the reference does not exist on the GPU box).
"""
import importlib.util
import sys
import types

import torch

REF_VIT = "/root/reference/Multi-Task_Pretrain/backbone/vit_win_rvsa_v3_wsz7.py"


def _drop_path(x, drop_prob: float = 0.0, training: bool = False):
    if drop_prob == 0.0 or not training:
        return x
    keep = 1 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    mask = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
    mask.floor_()
    return x.div(keep) * mask


def _to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def load_reference():
    if "ref_vit" in sys.modules:
        return sys.modules["ref_vit"]
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_layers.drop_path = _drop_path
    timm_layers.to_2tuple = _to_2tuple
    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models = timm_models
    timm_models.layers = timm_layers
    mmengine = types.ModuleType("mmengine")
    mmengine_dist = types.ModuleType("mmengine.dist")
    mmengine_dist.get_dist_info = lambda: (0, 1)
    mmengine.dist = mmengine_dist
    for name, mod in [("timm", timm), ("timm.models", timm_models),
                      ("timm.models.layers", timm_layers),
                      ("mmengine", mmengine), ("mmengine.dist", mmengine_dist)]:
        sys.modules.setdefault(name, mod)
    spec = importlib.util.spec_from_file_location("ref_vit", REF_VIT)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_vit"] = mod
    spec.loader.exec_module(mod)
    return mod


REF_DET = "/root/reference/RS_Tasks_Finetune/Horizontal_Detection/mmdet/models/backbones/vit_rvsa_mtp.py"


def load_reference_det():
    """The mmdet fine-tune copy (`RVSA_MTP`, ViTDet style: final norm, one map through fpn1-4, full attention without
    rel-pos).  One more stub: `mmdet.registry.MODELS.register_module()` (a decorator that returns the class)."""
    if "ref_vit_det" in sys.modules:
        return sys.modules["ref_vit_det"]
    load_reference()     # installs the timm / mmengine stubs

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls
    mmdet = types.ModuleType("mmdet")
    mmdet_registry = types.ModuleType("mmdet.registry")
    mmdet_registry.MODELS = _Reg()
    mmdet.registry = mmdet_registry
    sys.modules.setdefault("mmdet", mmdet)
    sys.modules.setdefault("mmdet.registry", mmdet_registry)
    spec = importlib.util.spec_from_file_location("ref_vit_det", REF_DET)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_vit_det"] = mod
    spec.loader.exec_module(mod)
    return mod


REF_CLS = "/root/reference/RS_Tasks_Finetune/Scene_Classification/mmpretrain/models/backbones/vit_rvsa_mtp.py"


def load_reference_cls():
    """The mmpretrain fine-tune copy (`RVSA_MTP` returning the raw taps as NCHW, no fpn ops).  Stub: `mmpretrain.registry.MODELS`."""
    if "ref_vit_cls" in sys.modules:
        return sys.modules["ref_vit_cls"]
    load_reference()

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls
    pkg = types.ModuleType("mmpretrain")
    reg = types.ModuleType("mmpretrain.registry")
    reg.MODELS = _Reg()
    pkg.registry = reg
    sys.modules.setdefault("mmpretrain", pkg)
    sys.modules.setdefault("mmpretrain.registry", reg)
    spec = importlib.util.spec_from_file_location("ref_vit_cls", REF_CLS)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_vit_cls"] = mod
    spec.loader.exec_module(mod)
    return mod


REF_DCN_FUNC = "/root/reference/Multi-Task_Pretrain/backbone/ops_dcnv3/functions/dcnv3_func.py"


def load_reference_dcnv3():
    """`dcnv3_core_pytorch` -- the pure-torch DCNv3 core the reference's own test (ops_dcnv3/test.py) checks its CUDA extension
    against.  Stubs: the compiled `DCNv3` extension module (never called here) and `pkg_resources.get_distribution('DCNv3')`
    (dcnv3_func.py:16-19 reads only `.version`)."""
    if "ref_dcnv3_func" in sys.modules:
        return sys.modules["ref_dcnv3_func"]
    sys.modules.setdefault("DCNv3", types.ModuleType("DCNv3"))
    pkg = types.ModuleType("pkg_resources")
    pkg.get_distribution = lambda name: types.SimpleNamespace(version="1.1")
    sys.modules.setdefault("pkg_resources", pkg)
    spec = importlib.util.spec_from_file_location("ref_dcnv3_func", REF_DCN_FUNC)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_dcnv3_func"] = mod
    spec.loader.exec_module(mod)
    return mod


REF_BACKBONE_DIR = "/root/reference/Multi-Task_Pretrain/backbone"


def load_reference_internimage():
    """`backbone/intern_image.py` with `core_op='DCNv3_pytorch'` (last row).  The file does a relative import of
    its `ops_dcnv3` package, so it is loaded as the submodule `ref_backbone.intern_image` of a synthetic package whose path is
    the reference's backbone directory.  Extra stubs: `timm.models.layers.DropPath` (identity when drop_prob == 0 or eval),
    the compiled `DCNv3` extension and `pkg_resources` (see load_reference_dcnv3)."""
    if "ref_backbone.intern_image" in sys.modules:
        return sys.modules["ref_backbone.intern_image"]
    load_reference()              # timm / mmengine stubs
    load_reference_dcnv3()        # DCNv3 / pkg_resources stubs

    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return _drop_path(x, self.drop_prob, self.training)
    sys.modules["timm.models.layers"].DropPath = DropPath
    pkg = types.ModuleType("ref_backbone")
    pkg.__path__ = [REF_BACKBONE_DIR]
    sys.modules["ref_backbone"] = pkg
    import importlib
    return importlib.import_module("ref_backbone.intern_image")
