"""MI355X-native DCNv3 core operator (InternImage), mirroring Multi-Task_Pretrain/backbone/ops_dcnv3."""
from .functions import DCNv3Function, dcnv3_backward, dcnv3_forward  # noqa: F401

__all__ = ["DCNv3Function", "dcnv3_forward", "dcnv3_backward"]
