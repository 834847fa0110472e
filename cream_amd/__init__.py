"""cream_amd — MI355X (gfx950) native hot path of microsoft/Cream: AutoFormer's
weight-entangled supernet step and iRPE's rpe_index / RPE attention, as hand-written
HIP kernels behind a C ABI (include/cream_amd.h)."""
__version__ = "0.1.0"
