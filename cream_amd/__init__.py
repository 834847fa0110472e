"""cream_amd — MI355X (gfx950) native hot path of microsoft/Cream: AutoFormer's
weight-entangled supernet step and iRPE's rpe_index / RPE attention, as hand-written
HIP kernels behind a C ABI (include/cream_amd.h)."""
__version__ = "0.1.0"

import os as _os

# Kernel arguments in device memory (a HIP runtime switch): shortens the dispatch of back-to-back kernels — a
# supernet step is ~330 launches; +3.3 % images/s in a same-box A/B.  Only effective if the HIP runtime has not
# initialised yet (import cream_amd before the first device call); an explicit user setting wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
