"""mtp_amd -- MI355X-native (gfx950) ViT+RVSA backbone hot path of ViTAE-Transformer/MTP.

Python host mirrors the reference's backbone interface (mtp_amd.backbone); compute is libmtp_hip.so (C ABI in
include/mtp_hip.h, kernels in mtp_amd/csrc).  No CPU / eager-PyTorch fallback.
"""
from .backbone import (InternImage, internimage_xl, RVSA_MTP, RVSA_MTP_branches, RVSA_MTP_det, RVSA_MTP_taps, ViT_Win_RVSA_V3_WSZ7, vit_b_rvsa, vit_l_rvsa,  # noqa: F401
                       window_partition, window_reverse)
from .registry import BACKBONES, MODELS, build_backbone  # noqa: F401

__version__ = "0.1.0"
