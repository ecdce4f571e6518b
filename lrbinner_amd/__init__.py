"""lrbinner_amd -- MI355X-native implementation of LRBinner's profile hot path.

The arithmetic lives in liblrb_hip.so (HIP, gfx950) behind the C ABI of
include/lrb_hip.h; this package is the host-side mirror of the reference's
``mbcclr_utils`` interface for that path.  There is no CPU fallback.
"""
__version__ = "0.1.0"
