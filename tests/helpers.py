import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b):
    """max |a-b| / max|b| — the '<= 1e-3 rel' measure used for logits/grads."""
    a = torch.as_tensor(np.asarray(a)).double().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def config_of(fix):
    return json.loads(bytes(fix["config"]).decode())


def check_against_fixture(fix, logits, loss, grads, tol, stride=997, skip_small=1e-12):
    """Compare a step's outputs with a golden step fixture.  Returns the worst relative error."""
    worst = {}
    worst["logits"] = max_rel(logits.detach().cpu().float(), fix["logits"])
    worst["loss"] = abs(float(loss) - float(fix["loss"][0])) / abs(float(fix["loss"][0]))
    for k, v in fix.items():
        if k.startswith("full|"):
            name = k[5:]
            worst[k] = max_rel(grads[name].detach().cpu().float(), v)
        elif k.endswith("|sample"):
            name = k[:-7]
            ref_norm = float(fix[name + "|norm"][0])
            g = grads[name].detach().cpu().double().flatten()
            if ref_norm < skip_small:
                assert float(g.norm()) < 1e-6, f"{name}: reference grad is zero, ours is not"
                continue
            # strided sample, scaled by the tensor's own magnitude (norm / sqrt(n))
            scale = ref_norm / max(1.0, g.numel()) ** 0.5
            worst[k] = float((g[::stride] - torch.from_numpy(v)).abs().max() / max(scale, 1e-30)) / 10.0
            worst[name + "|norm"] = abs(float(g.norm()) - ref_norm) / ref_norm
    bad = {k: e for k, e in worst.items() if not (e <= tol)}
    assert not bad, f"exceeds tol {tol}: " + ", ".join(f"{k}={e:.2e}" for k, e in sorted(bad.items(), key=lambda t: -t[1])[:8])
    return max(worst.values())


def forbid_framework_matmul(what="this path"):
    """Context manager: the framework's matrix products (= the vendor library) raise when they receive a device tensor —
    torch.mm / matmul / bmm / baddbmm / einsum, F.linear and the `@` operator.  Products must then run on the own kernels."""
    import contextlib
    import unittest.mock as mock

    def guard(real):
        def f(*a, **k):
            flat = [t for x in a for t in (x if isinstance(x, (list, tuple)) else [x])]
            if any(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
                raise AssertionError(f"framework matmul on a device tensor inside {what}")
            return real(*a, **k)
        return f
    stack = contextlib.ExitStack()
    for owner, name in [(torch, "mm"), (torch, "matmul"), (torch, "bmm"), (torch, "baddbmm"), (torch, "einsum"),
                        (torch.nn.functional, "linear"), (torch.Tensor, "__matmul__"), (torch.Tensor, "__rmatmul__"),
                        (torch.Tensor, "matmul"), (torch.Tensor, "mm"), (torch.Tensor, "bmm")]:
        stack.enter_context(mock.patch.object(owner, name, guard(getattr(owner, name))))
    return stack
