"""Compile the REFERENCE's own rpe_index.cpp (CPU-only) into oracle/_ref/ — TEST
INFRASTRUCTURE.

The source is compiled where it lies under /root/reference (never copied); only the
build products land in oracle/_ref/ (git-ignored, but shipped to the GPU box with the
snapshot).  The result is a Python extension module named `rpe_index_cpp`, loaded by
`load_ref()` under the private name `rpe_index_cpp_ref` so that it can never shadow the
product's drop-in module.

    python oracle/build_ref.py        # ~40 s with g++ 11 + torch headers
"""
import glob
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF_SRC = "/root/reference/iRPE/DeiT-with-iRPE/rpe_ops/rpe_index.cpp"


def built_path():
    hits = glob.glob(os.path.join(OUT, "rpe_index_cpp*.so"))
    return hits[0] if hits else None


def build(verbose=False):
    """Build if the reference checkout is present; return the .so path or None."""
    if built_path():
        return built_path()
    if not os.path.exists(REF_SRC):
        return None
    from torch.utils import cpp_extension
    os.makedirs(OUT, exist_ok=True)
    cpp_extension.load(name="rpe_index_cpp", sources=[REF_SRC],
                       extra_cflags=["-fopenmp", "-O3"], extra_ldflags=["-fopenmp"],
                       build_directory=OUT, verbose=verbose, is_python_module=False)
    return built_path()


def load_ref():
    """Import the compiled reference extension (None when it was never built)."""
    path = built_path()
    if path is None:
        return None
    import torch  # noqa: F401  (libtorch must be loaded before the extension)
    spec = importlib.util.spec_from_file_location("rpe_index_cpp", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    p = build(verbose="-v" in sys.argv)
    print(p or "reference checkout not present; nothing built")
