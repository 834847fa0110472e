"""Build libcream_amd.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m cream_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels
with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libcream_amd.so")
STAMP = os.path.join(HERE, ".libcream_amd.stamp")
ARCH = "gfx950"
HIPCC_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off",
               "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")
    return exe


def sources():
    out = []
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".cpp")):
            out.append(os.path.join(CSRC, name))
    return out


def _digest():
    h = hashlib.sha256()
    files = sources() + [os.path.join(INCLUDE, "cream_amd.h")]
    files += sorted(os.path.join(CSRC, n) for n in os.listdir(CSRC) if n.endswith((".h", ".hpp")))
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode()) if 'EXTRA_FLAGS' in globals() else None
    for f in files:
        h.update(f.encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def is_current():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == _digest()


# per-file code generation choices (measured, profiles/r02_irpe_attention.md): kernels that run VALU work on
# MFMA results every tile (bias gathers, softmax) keep the accumulators in architectural VGPRs — with
# the default AGPR form the compiler copies every score tile AGPR -> VGPR (240 v_accvgpr moves per
# iteration in the fused iRPE forward) and the AGPR half of the register file caps occupancy
EXTRA_FLAGS = {"irpe_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def build(force=False, verbose=False):
    """Compile every source under csrc/ and link libcream_amd.so.  Returns the path."""
    if not force and is_current():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    tag = f"{ARCH};hipcc;{_digest()[:12]}"
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + HIPCC_FLAGS + [f"-I{INCLUDE}", f"-I{CSRC}", f'-DCREAM_BUILD_TAG="{tag}"',
                                       "-x", "hip" if src.endswith(".hip") else "c++", "-c", src, "-o", obj]
        cmd[1:1] = EXTRA_FLAGS.get(os.path.basename(src), [])
        if not src.endswith(".hip"):
            # plain host C++: no offload needed (HIP host APIs only)
            cmd = [c for c in cmd if not c.startswith("--offload-arch")]
            cmd[cmd.index("-x") + 1] = "c++"
            cmd += ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[cream_amd.build] FAILED {src}\n{out.decode(errors='replace')}\n")
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("cream_amd: HIP compilation failed")
    # no vendor GEMM library: every kernel of the path is in csrc/
    link = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-lpthread"]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(path)
