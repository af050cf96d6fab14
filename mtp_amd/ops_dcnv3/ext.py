"""Stand-in for the reference's compiled `DCNv3` extension module (ops_dcnv3/src/vision.cpp:14-17): the two functions it
exports, backed by libmtp_hip.so.  `import mtp_amd.ops_dcnv3.ext as DCNv3` is the one-line change in the reference's
functions/dcnv3_func.py."""
from .functions import dcnv3_backward, dcnv3_forward  # noqa: F401

__version__ = "1.1"   # dcnv3_func.py:19 reads the distribution version to decide whether `remove_center` is passed (> 1.0: always)
