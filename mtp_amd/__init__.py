"""mtp_amd -- MI355X-native (gfx950) ViT+RVSA backbone hot path of ViTAE-Transformer/MTP.

Python host mirrors the reference's backbone interface (mtp_amd.backbone); compute is libmtp_hip.so (C ABI in
include/mtp_hip.h, kernels in mtp_amd/csrc).  No CPU / eager-PyTorch fallback.
"""
import os as _os

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run one after the
# other.  A training process here has the compute stream, the weight-gradient side stream, the gradient-exchange stream and RCCL's own: with 4
# queues the side stream lands on the compute stream's queue and overlaps nothing (measured).  Read when HIP initialises, i.e. at
# the first device call -- a value the user has set wins.
_HWQ_BEFORE_IMPORT = _os.environ.get("GPU_MAX_HW_QUEUES")
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _hip_already_initialised():
    try:
        import torch as _torch
        return bool(_torch.cuda.is_initialized())
    except Exception:
        return False


# HIP reads the variable once, at its first device call: an import that comes after that cannot change the queue count any more.
_HIP_INIT_BEFORE_IMPORT = _hip_already_initialised()
_warned_hwq = False


def hw_queue_note():
    """None, or why the weight-gradient side stream (on by default) may not get a hardware queue of its own in this process -- asked once per process by the
    engines, which pass it on as a warning: the setting is a side effect on the whole process (it changes the queue allocation of every HIP library in it) and
    has no effect when HIP was initialised before `import mtp_amd`."""
    global _warned_hwq
    if _warned_hwq:
        return None
    val = _os.environ.get("GPU_MAX_HW_QUEUES")
    note = None
    if _HIP_INIT_BEFORE_IMPORT and _HWQ_BEFORE_IMPORT is None:
        note = ("HIP was initialised before `import mtp_amd`: GPU_MAX_HW_QUEUES=8 could not take effect, the weight-gradient side stream may share the "
                "compute stream's hardware queue (no overlap, nothing wrong); import mtp_amd or set GPU_MAX_HW_QUEUES=8 before the first CUDA call")
    elif val is not None and val.isdigit() and int(val) < 8:
        note = ("GPU_MAX_HW_QUEUES=%s (set by the environment): with fewer than 8 hardware queues the weight-gradient side stream may share the compute "
                "stream's queue next to the gradient exchange (no overlap, nothing wrong)" % val)
    _warned_hwq = True
    return note

from .backbone import (InternImage, internimage_xl, RVSA_MTP, RVSA_MTP_branches, RVSA_MTP_det, RVSA_MTP_taps, ViT_Win_RVSA_V3_WSZ7, vit_b_rvsa, vit_l_rvsa,  # noqa: F401
                       window_partition, window_reverse)
from .registry import BACKBONES, MODELS, build_backbone  # noqa: F401

__version__ = "0.6.0"      # = mtp_version() of libmtp_hip.so ("mtp_hip 0.6 (gfx950)"): the round of the build
