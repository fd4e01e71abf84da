import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "pytorch-retinanet_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_cases(npz):
    """Group 'case.key' entries of an npz into {case: {key: array}}."""
    out = {}
    for k in npz.files:
        case, key = k.split(".", 1)
        out.setdefault(case, {})[key] = npz[k]
    return out


@pytest.fixture(scope="session")
def focal_golden():
    return golden_cases(load_golden("focal_loss.npz"))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))) if a.size else 0.0


@pytest.fixture(scope="session")
def focal_logits_golden():
    return golden_cases(load_golden("focal_loss_logits.npz"))


def logits_grad_close(got, want, logits, rtol=1e-5, atol_rel=1e-7):
    """Gradient check for the logits-in loss.

    The loss is a function of p = float32 sigmoid(x).  Two float32 sigmoids (torch's, numpy's,
    the GPU's) may differ in the last bit or two of p, and the loss uses 1 - p: for p close to 1
    that last bit is a relative change of ulp(p) / (1 - p) in (1 - p) and in the gradient, however
    the rest is computed.  So the bar per element is  rtol + 4 ulp / (1 - p)  relative (the second
    term is below 1e-5 until p > 0.976), plus the usual absolute floor; and where p lies within a
    few ulp of a clamp bound (1e-4, 1 - 1e-4: losses.py:57) either side of the clamp, zero gradient
    or not, is accepted.
    """
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    x = np.asarray(logits, np.float64)
    with np.errstate(over="ignore"):
        p = 1.0 / (1.0 + np.exp(-x))
    eps = float(np.finfo(np.float32).eps)
    rel = rtol + 4.0 * eps / np.maximum(1.0 - p, eps)
    atol = atol_rel * float(np.max(np.abs(want))) if want.size else 0.0
    edge = (np.abs(p - 1e-4) <= 4 * eps * 1e-4) | (np.abs(p - (1.0 - 1e-4)) <= 4 * eps)
    bad = (np.abs(got - want) > rel * np.abs(want) + atol) & ~edge
    assert not bad.any(), f"{bad.sum()} of {bad.size} elements off; worst abs {np.max(np.abs(got - want)[~edge])}"
    assert np.array_equal((got == 0) | edge, (want == 0) | edge)
