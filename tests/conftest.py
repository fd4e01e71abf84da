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


RTOL = 1e-5
# Regression gradients: inside the quadratic zone of smooth-L1 the gradient is 9*(t - r), so a
# one-ulp difference of log() in the target t (|t| up to a few units) moves it by 9*ulp(t) however
# small t - r is: a floor any float32 evaluation has.  Measured over every regression-gradient
# comparison of the GPU suite (the full-size configurations and the two-rank runs included; the
# numbers are written out by test_zz_tolerance_floors_report): a few dozen of the non-zero
# elements pass only thanks to the floor, the largest excess being about 1e-6 of the largest
# gradient (the linear zone's +-1); the floor is set to twice that.  Classification gradients
# need no floor at all: theirs is 0.
REG_ATOL_REL = 2e-6

# What the absolute floors of grad_close actually did, per kind of gradient: filled as the tests
# run, checked and written out by test_zz_tolerance_floors_report (tests/test_gpu_parity.py).
FLOOR_STATS = {}


def grad_close(got, want, rtol=RTOL, atol_rel=0.0, kind="d_cls"):
    """THE gradient comparison of the GPU suite: |got - want| <= rtol |want| + atol_rel max|want|,
    zeros in the same places; every call adds to FLOOR_STATS[kind] what the floor was needed for."""
    want = np.asarray(want, np.float64)
    got = np.asarray(got, np.float64)
    top = float(np.max(np.abs(want))) if want.size else 0.0
    err = np.abs(got - want)
    bad = err > rtol * np.abs(want) + atol_rel * top
    assert not bad.any(), f"{bad.sum()} of {bad.size} elements off; worst abs {np.max(err)}"
    assert np.array_equal(got == 0, want == 0)
    if want.size and top > 0:
        over = err > rtol * np.abs(want)              # these pass only thanks to the absolute floor
        big = np.abs(want) > 1e-3 * top
        st = FLOOR_STATS.setdefault(kind, dict(elements=0, nonzero=0, needed_floor=0, floor_used_rel=0.0,
                                               worst_rel_above_milli_max=0.0))
        st["elements"] += int(want.size)
        st["nonzero"] += int(np.count_nonzero(want))
        st["needed_floor"] += int(over.sum())
        if over.any():
            st["floor_used_rel"] = max(st["floor_used_rel"],
                                       float(np.max(err[over] - rtol * np.abs(want[over])) / top))
        if big.any():
            st["worst_rel_above_milli_max"] = max(st["worst_rel_above_milli_max"],
                                                    float(np.max(err[big] / np.abs(want[big]))))


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
