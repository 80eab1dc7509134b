"""bundle_adjustment_amd — MI355X-native LM / implicit-Schur PCG bundle adjustment.

The product is the C-ABI library libmiba.so (include/miba.h, csrc/); `miba` is its ctypes
binding, `synth` generates BAL-format problems.  The C++ mirror of the reference's
BundleAdjustment / BAConstraint / BalDataloader classes lives in host/ and include/bundleadjust/.
"""
from . import synth  # noqa: F401
from . import miba   # noqa: F401
