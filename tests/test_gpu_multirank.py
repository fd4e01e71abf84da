"""The real loss module under several ranks on the one GPU of the test box: two fresh processes
(gloo; both on cuda:0), each running DataParallelFocalLoss(FocalLoss(batch_divisor=B)) on its
shard of the 'normal' golden fixture.  Global losses must equal the fixture's (the reference run
on the whole batch, losses.py:139-141) and the ranks' gradients must concatenate to its gradients.
The same for normalizer='global' (the staged path: assign, all-reduce of the positive count,
stream) against the oracle."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REG_ATOL_REL, RTOL, grad_close, load_golden
from oracle import dense_head_oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_ranks(world, normalizer, tmp_path):
    port = str(_free_port())
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_gpu_worker.py"), str(r), str(world),
                               port, normalizer, outs[r]], env=env) for r in range(world)]
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert codes == [0] * world, codes
    return [np.load(o) for o in outs]


@pytest.mark.parametrize("normalizer", ["per_image", "global"])
def test_two_ranks_equal_single_process(tmp_path, normalizer):
    world = 2
    res = _run_ranks(world, normalizer, tmp_path)
    g = load_golden("focal_loss.npz")
    cls, reg, anc, ann = (g["normal." + k] for k in ("cls", "reg", "anchors", "ann"))
    if normalizer == "per_image":       # the reference's own outputs
        want_cls, want_reg = g["normal.cls_loss"], g["normal.reg_loss"]
        scale = g["normal.g"]
        want_dc, want_dr = g["normal.d_cls"] / scale[0], g["normal.d_reg"] / scale[1]
    else:                               # no counterpart in the reference: the oracle's restatement
        ref = O.focal_loss_fwd_bwd(cls, reg, anc, ann, normalizer="global")
        want_cls, want_reg = ref["classification_loss"], ref["regression_loss"]
        want_dc, want_dr = ref["d_classifications"], ref["d_regressions"]
    per_image_pos = [O.assign(anc[0], a)["num_pos"] for a in ann]
    assert [tuple(r["bounds"]) for r in res] == [(0, 2), (2, 3)]
    for r in res:
        for step in (0, 1):
            assert np.allclose(r[f"cls_loss{step}"], want_cls, rtol=RTOL, atol=0)
            assert np.allclose(r[f"reg_loss{step}"], want_reg, rtol=RTOL, atol=0)
        assert float(r["global_num_pos"][0]) == float(sum(per_image_pos))
        assert float(r["global_images"][0]) == float(len(ann))
        assert float(r["ranks_seen"][0]) == float(world)
        lo, hi = r["bounds"]
        assert list(r["local_num_pos"]) == per_image_pos[lo:hi]
    for step in (0, 1):
        dc = np.concatenate([r[f"d_cls{step}"] for r in res])
        dr = np.concatenate([r[f"d_reg{step}"] for r in res])
        grad_close(dc, want_dc)
        grad_close(dr, want_dr, atol_rel=REG_ATOL_REL, kind="d_reg")
