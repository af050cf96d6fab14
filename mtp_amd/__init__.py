"""mtp_amd -- MI355X-native (gfx950) ViT+RVSA backbone hot path of ViTAE-Transformer/MTP.

Python host mirrors the reference's backbone interface (mtp_amd.backbone); compute is libmtp_hip.so (C ABI in
include/mtp_hip.h, kernels in mtp_amd/csrc).  No CPU / eager-PyTorch fallback.
"""
import os as _os

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run one after the
# other.  A training process here has the compute stream, the weight-gradient side stream, the gradient-exchange stream and RCCL's own: with 4
# queues the side stream lands on the compute stream's queue and overlaps nothing (measured).  Read when HIP initialises, i.e. at
# the first device call -- a value the user has set wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .backbone import (InternImage, internimage_xl, RVSA_MTP, RVSA_MTP_branches, RVSA_MTP_det, RVSA_MTP_taps, ViT_Win_RVSA_V3_WSZ7, vit_b_rvsa, vit_l_rvsa,  # noqa: F401
                       window_partition, window_reverse)
from .registry import BACKBONES, MODELS, build_backbone  # noqa: F401

__version__ = "0.1.0"
