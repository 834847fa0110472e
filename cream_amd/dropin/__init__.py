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


def install_autoformer(reference_model_dir=None):
    """Make `import model.supernet_transformer` resolve the REFERENCE's unchanged
    supernet_transformer.py (from `reference_model_dir`, e.g. .../AutoFormer/model) while
    `model.module.*` and `model.utils` resolve to the MI355X implementations here."""
    import importlib
    install()
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
        del sys.modules[k]
    pkg = importlib.import_module("model")
    if reference_model_dir and reference_model_dir not in pkg.__path__:
        pkg.__path__.append(reference_model_dir)      # searched AFTER our directory
    return pkg
