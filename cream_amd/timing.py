"""Optional per-kernel timing with HIP events on the stream the kernel is launched on
(torch's current stream — every C-ABI call of this package launches there).  Disabled by
default (one attribute test per call); bench.py enables it for the timed region to obtain
the dominant kernel's average launch duration for the roofline fraction."""
import contextlib

import torch

_enabled = False
_only = None           # optional set of region names to time (None = all)
_records = {}          # name -> list of (start_event, end_event, algorithmic_bytes, flops)


def enable(flag=True, only=None):
    """`only`: iterable of region names to time; the other regions cost one dict test."""
    global _enabled, _only
    _enabled = bool(flag)
    _only = set(only) if only else None


def reset():
    _records.clear()


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _Null()


def region(name, nbytes=0, flops=0):
    """Context manager around one or more launches.  Disabled / filtered regions return a
    shared no-op object (this sits on the launch path of every kernel)."""
    if not _enabled or (_only is not None and name not in _only):
        return _NULL
    return _timed(name, nbytes, flops)


@contextlib.contextmanager
def _timed(name, nbytes, flops):
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _records.setdefault(name, []).append((a, b, nbytes, flops))


def summary():
    """-> {name: dict(launches, total_ms, avg_ms, bytes, flops)}; synchronises."""
    torch.cuda.synchronize()
    out = {}
    for name, recs in _records.items():
        total = sum(a.elapsed_time(b) for a, b, _, _ in recs)
        out[name] = dict(launches=len(recs), total_ms=total, avg_ms=total / max(1, len(recs)),
                         bytes=sum(r[2] for r in recs), flops=sum(r[3] for r in recs))
    return out
