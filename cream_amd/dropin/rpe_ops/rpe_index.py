"""Drop-in for the reference package module `rpe_ops.rpe_index`
(iRPE/DeiT-with-iRPE/rpe_ops/rpe_index.py) — `irpe.py:8-15` does
`from rpe_ops.rpe_index import RPEIndexFunction`."""
import rpe_index_cpp

EXPECTED_VERSION = "1.2.0"
assert rpe_index_cpp.version() == EXPECTED_VERSION, (
    f"Unmatched `rpe_index_cpp` version: {rpe_index_cpp.version()}, "
    f"expected version: {EXPECTED_VERSION}")

from cream_amd.rpe_index import RPEIndexFunction  # noqa: E402,F401
