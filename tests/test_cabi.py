"""The drop-in boundary: the C-ABI library loads, exports exactly what include/*.h
declares, and the product never reaches into oracle/."""
import ctypes
import os
import re

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "cream_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cream_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_symbols():
    syms = _header_symbols()
    assert "cream_rpe_index_fwd" in syms and "cream_rpe_index_bwd" in syms
    assert "cream_version" in syms


def test_library_exports_every_declared_symbol():
    from cream_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/cream_amd.h but not exported"


def test_ctypes_table_matches_header():
    from cream_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    _lib.load()


def test_version_handshake():
    # rpe_ops/rpe_index.py:5-8 asserts rpe_index_cpp.version() == "1.2.0"
    from cream_amd import _lib
    assert _lib.version() == "1.2.0"
    assert "gfx950" in _lib.build_info()


def test_bad_arguments_return_codes_not_crashes():
    from cream_amd import _lib
    lib = _lib.load()
    # unknown dtype, host entry (no GPU needed)
    import numpy as np
    x = np.zeros((1, 1, 2, 3), np.float32)
    idx = np.zeros((2, 2), np.int32)
    y = np.zeros((1, 1, 2, 2), np.float32)
    rc = lib.cream_rpe_index_fwd_host(y.ctypes.data, x.ctypes.data, idx.ctypes.data, 1, 1, 2, 2, 3, 99)
    assert rc == -2
    rc = lib.cream_rpe_index_fwd_host(None, x.ctypes.data, idx.ctypes.data, 1, 1, 2, 2, 3, 0)
    assert rc == -1
    rc = lib.cream_rpe_index_fwd_host(y.ctypes.data, x.ctypes.data, idx.ctypes.data, -1, 1, 2, 2, 3, 0)
    assert rc == -1
    # empty problem is a no-op success
    rc = lib.cream_rpe_index_fwd_host(None, None, None, 0, 1, 2, 2, 3, 0)
    assert rc == 0


def test_product_does_not_import_oracle():
    """Nothing under cream_amd/ may import, link or execute oracle/ (test infrastructure)."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[/.]", re.M)
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cream_amd")):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                p = os.path.join(dirpath, f)
                if pat.search(open(p, errors="replace").read()):
                    bad.append(p)
    assert not bad, f"product files reference oracle/: {bad}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cream_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.CreamLibraryError):
        _lib.load()


def test_block_workspace_layout_is_pure_host_arithmetic():
    """cream_block_fwd_workspace / cream_block_bwd_workspace: sizes and offsets of the one flat
    workspace per block and direction (no GPU needed: nothing is launched or allocated)."""
    from cream_amd import _lib
    lib = _lib.load()
    d = _lib.BlockDesc()
    d.B, d.N, d.E, d.H, d.F = 128, 197, 384, 6, 1344
    d.gh, d.gw, d.mr = 14, 14, 14
    for name in ("wqkv", "wqkv_t", "bqkv", "wproj", "wproj_t", "bproj", "w1", "w1_t", "b1", "w2", "w2_t", "b2", "ln1_g",
                 "ln1_b", "ln2_g", "ln2_b", "tkv", "tkh", "tvv", "tvh"):
        setattr(d, name, 0x1000)                                   # any non-null pointer: only checked, never read
    o = [ctypes.c_int64() for _ in range(6)]
    ft = lib.cream_block_fwd_workspace(ctypes.byref(d), ctypes.byref(o[0]), ctypes.byref(o[1]), ctypes.byref(o[2]))
    bt = lib.cream_block_bwd_workspace(ctypes.byref(d), ctypes.byref(o[3]), ctypes.byref(o[4]), ctypes.byref(o[5]))
    M, E, Q, F = 128 * 197, 384, 384, 1344
    # everything the backward reads from the forward: x, a, qkv, o, x1, c, h, g, f (+ small stats, sp)
    assert ft >= M * E * 4 * 2 + M * E * 2 * 4 + M * 3 * Q * 2 + M * Q * 2 + 2 * M * F * 2
    assert ft < 1.2 * (M * E * 4 * 2 + M * E * 2 * 4 + M * 3 * Q * 2 + M * Q * 2 + 2 * M * F * 2 + 128 * 6 * 64 * 224 * 2) + (1 << 22)
    assert bt > 0 and all(v.value % 256 == 0 for v in o)
    assert 0 <= o[0].value < o[1].value < o[2].value < ft            # x | ... | x1 | ... | f
    assert o[3].value < o[4].value < o[5].value < bt                 # dx | df_prev | pl1
    d.E = 380                                                        # E % 8 != 0
    assert lib.cream_block_fwd_workspace(ctypes.byref(d), None, None, None) == -1
    assert lib.cream_block_bwd_workspace(ctypes.byref(d), None, None, None) == -1


def test_new_entry_points_validate_arguments_without_launching():
    from cream_amd import _lib
    lib = _lib.load()
    assert lib.cream_grad_finalize(None, 0, None) == 0               # nothing to do
    assert lib.cream_grad_finalize(None, 1, None) == -1
    assert lib.cream_grad_finalize(None, _lib.MAX_GRAD_JOBS + 1, None) == -1
    j = (_lib.GradJob * 1)()
    j[0].dst, j[0].src, j[0].ld, j[0].pstride, j[0].nparts, j[0].rows, j[0].cols = 0x1000, 0x2000, 6, 8, 2, 1, 8
    assert lib.cream_grad_finalize(ctypes.cast(j, ctypes.c_void_p), 1, None) == -1      # ld % 4 != 0
    assert lib.cream_linear_fwd(None, None, None, None, 0, 8, 8, 8, None) == 0          # empty problem
    assert lib.cream_linear_fwd(None, None, None, None, 4, 8, 8, 8, None) == -1
    assert lib.cream_linear_fwd(0x1000, 0x1000, 0x1000, None, 4, 8, 16, 8, None) == -1   # ldw < K
    assert lib.cream_linear_fwd(0x1000, 0x1000, 0x1000, None, 4, 12, 16, 16, None) == -1  # N % 8 != 0
    assert lib.cream_linear_fwd(0x1000, 0x1008, 0x1000, None, 4, 8, 16, 16, None) == -1  # x not 16-byte aligned
    assert lib.cream_linear_dgrad(0x1000, 0x1000, 0x1000, 4, 16, 8, 8, None) == -1       # ld of W^T < N
    assert lib.cream_linear_fwd_seg(0x1000, 0x1000, 0x1000, None, 4, 384, 64, 64, 64, 4096, None) == -1   # > 3 segments
    assert lib.cream_linear_dgrad_seg(0x1000, 0x1000, 0x1000, 4, 192, 64, 64, 32, 4096, None) == -1       # kseg % 64 != 0
    assert lib.cream_linear_wgrad_parts(0x1000, None, 0x1000, 0x1000, 10, 8, 8, 0, None) == -1            # S < 1
    assert lib.cream_linear_wgrad_parts(None, None, 0x1000, 0x1000, 10, 8, 8, 1, None) == -1
    assert lib.cream_linear_wgrad_splits(25216, 384, 384) == 16 and lib.cream_linear_wgrad_splits(64, 384, 384) == 1
    assert lib.cream_linear_wgrad_splits(25216, 1344, 384) == 15          # 33 tiles x 15 = 495 <= 512 resident workgroups
    assert lib.cream_gemm_rows_per_colsum_slab() == 128
    assert lib.cream_param_job_tiles(96, 64) == 1 and lib.cream_param_job_tiles(97, 65) == 4 and lib.cream_param_job_tiles(0, 5) == 0
    assert lib.cream_adamw_step(None, None, 0, 0, 1, 1e-3, 0.9, 0.999, 1e-8, 1, None) == 0      # nothing to do
    assert lib.cream_adamw_step(None, None, 2, 5, 1, 1e-3, 0.9, 0.999, 1e-8, 1, None) == -1
    assert lib.cream_adamw_step(0x1000, 0x1000, 2, 5, 1, 1e-3, 0.9, 0.999, 1e-8, 0, None) == -1  # update needs step >= 1
    assert lib.cream_colsum128_slabs(25216) == 197 and lib.cream_colsum128_slabs(0) == 0
    assert lib.cream_ln_partials() > 0
    # table-gradient partials of the attention backward: one per persistent workgroup, never more than the (b, h) items
    assert lib.cream_attn_rpe2d_dtab_parts(0, 6) == 0 and lib.cream_attn_rpe2d_dtab_parts(2, 3) == 6
    assert 1 <= lib.cream_attn_rpe2d_dtab_parts(128, 6) <= 768


def test_switches_without_a_device():
    """Entry points that take no device pointers: the process-wide kernel-choice switches (each returns the previous value)."""
    from cream_amd import _lib
    lib = _lib.load()
    for fn in (lib.cream_gemm_nt8, lib.cream_gemm_tn8, lib.cream_gemm_nt256, lib.cream_block_wgrad_bf16):
        prev = fn(-1)
        assert fn(1) == prev and fn(prev) == 1 and fn(-1) == prev
    # one token slice per workgroup: the macro-tile kernel of the weight gradients takes fewer, longer slices than the 128 x 128 one
    assert 1 <= lib.cream_linear_wgrad_splits_bf16(25216, 1344, 384) <= lib.cream_linear_wgrad_splits(25216, 1344, 384)
    assert lib.cream_linear_wgrad_splits_bf16(0, 8, 8) == 0


def test_cu_reserve_sizes_every_persistent_grid_and_moves_the_layout_epoch():
    """cream_cu_reserve (the data-parallel driver reserves CUs for RCCL's channels, cream_amd/comm.py): the grids follow — the
    attention kernels' workgroup count (= number of table-gradient partial blocks), the weight gradients' token slices — and the
    workspace-layout epoch moves, so cached workspace sizes are re-queried."""
    from cream_amd import _lib
    lib = _lib.load()
    prev = lib.cream_cu_reserve(-1)
    try:
        lib.cream_cu_reserve(0)
        full, e0 = lib.cream_cu_count(), lib.cream_block_layout_epoch()
        s0 = lib.cream_linear_wgrad_splits_bf16(25216, 1344, 384)
        assert full % 8 == 0 and lib.cream_attn_rpe2d_dtab_parts(128, 6) == min(768, full)
        assert lib.cream_cu_reserve(16) == 0
        assert lib.cream_cu_count() == full - 16 and lib.cream_block_layout_epoch() == e0 + 1
        assert lib.cream_attn_rpe2d_dtab_parts(128, 6) == min(768, full - 16)
        assert lib.cream_linear_wgrad_splits_bf16(25216, 1344, 384) <= s0
        assert lib.cream_cu_reserve(16) == 16 and lib.cream_block_layout_epoch() == e0 + 1      # unchanged value: no move
        lib.cream_cu_reserve(10 ** 6)
        assert lib.cream_cu_count() == 8                                                         # never below one CU per XCD
    finally:
        lib.cream_cu_reserve(prev)


def test_library_has_no_undefined_kernel_stubs():
    """Every kernel instantiation the host code launches must have its host stub in the library: clang silently drops the stub
    of a __global__ template whose body fails a DEFERRED host-side check (round 4: a target builtin inside a lambda called under
    `if constexpr`), the shared object still links — and the first launch dies with an unresolved symbol on the GPU box."""
    import subprocess
    from cream_amd import build
    out = subprocess.run(["nm", "-D", "--undefined-only", build.build()], capture_output=True, text=True, check=True).stdout
    bad = [l for l in out.splitlines() if "__device_stub__" in l or " cream_" in l]
    assert not bad, bad
