"""Drop-in for the reference's compiled extension module `rpe_index_cpp`
(iRPE/DeiT-with-iRPE/rpe_ops/setup.py:17-24, rpe_index.cpp:130-141).

Put `cream_amd/dropin` on sys.path (cream_amd.dropin.install()) and the reference's
`rpe_ops/rpe_index.py` / `irpe.py` import this module unchanged and find the same five
symbols; on ROCm `input.device.type == 'cuda'`, so `forward_gpu` / `backward_gpu` are
the names the reference dispatches to (rpe_ops/rpe_index.py:36-37,52-53).
"""
from cream_amd.rpe_index import (version, forward_cpu, backward_cpu,  # noqa: F401
                                 forward_gpu, backward_gpu)
