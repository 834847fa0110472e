"""Import-path drop-ins: modules named exactly as the reference's callers import them
(`rpe_index_cpp`, `rpe_ops.rpe_index`, `model.module.*`, `model.utils`).

    import cream_amd.dropin as d; d.install()

prepends this directory to sys.path so that the reference's unchanged callers
(`irpe.py`, `rpe_vision_transformer.py`, `model/supernet_transformer.py`) resolve their
imports to the MI355X implementations.  See INTEGRATION.md.
"""
import os
import sys

PATH = os.path.dirname(os.path.abspath(__file__))


def install():
    if PATH not in sys.path:
        sys.path.insert(0, PATH)
    return PATH
