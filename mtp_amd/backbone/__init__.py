from .vit_win_rvsa_v3_wsz7 import (RVSA_MTP, RVSA_MTP_branches, RVSA_MTP_det, RVSA_MTP_taps, ViT_Win_RVSA_V3_WSZ7, vit_b_rvsa,  # noqa: F401
                                   vit_l_rvsa, window_partition, window_reverse)

from .intern_image import InternImage, internimage_xl  # noqa: F401

__all__ = ["InternImage", "internimage_xl", "ViT_Win_RVSA_V3_WSZ7", "RVSA_MTP", "RVSA_MTP_branches", "RVSA_MTP_det", "RVSA_MTP_taps", "vit_b_rvsa", "vit_l_rvsa",
           "window_partition", "window_reverse"]
