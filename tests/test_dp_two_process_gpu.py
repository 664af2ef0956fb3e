"""GPU: two real processes (one rank each, sharing the test box's single GPU; gloo collective) run the batch-sharded
learner on the HIP path and must reproduce the single-process run: same index / crop / REDQ streams, device noise
indexed by the global sample, gradients all-reduced between *_grads and apply."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, world, iters=3, parallel="dp"):
    out = str(tmp_path / f"w{world}{parallel}.npz")
    worker = os.path.join(ROOT, "tests", "dp_worker.py")
    if world == 1:
        cmd = [sys.executable, worker, out, str(iters)]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", "29541", worker, out, str(iters)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, SERL_TEST_PARALLEL=parallel))
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_two_ranks_reproduce_one(gpu, tmp_path):
    one, two = _run(tmp_path, 1), _run(tmp_path, 2)
    assert int(one["step"]) == int(two["step"]) == 9
    # replicated inserts: both ranks of the 2-rank job hold the identical buffer and drew the identical index stream,
    # and it is the stream the single process drew (same transitions applied at the same batch boundaries)
    two_r1 = np.load(str(tmp_path / "w2dp.rank1.npz"))
    for other in (two_r1, one):
        assert np.array_equal(two["valid"], other["valid"]) and int(two["insert_index"]) == int(other["insert_index"])
        assert int(two["size"]) == int(other["size"]) and np.array_equal(two["idx"], other["idx"])
    assert int(two["size"]) == 150 + 3 * 7 + 9          # + one first-frame slot per episode begun (t = 0, 20, ..., 160)
    for k in two.files:                                  # the two ranks applied bit-identical updates
        if k.startswith(("critic", "actor", "enc", "temp")):
            assert np.array_equal(two[k], two_r1[k]), k
    assert np.allclose(one["info"], two["info"], rtol=2e-4, atol=1e-6), (one["info"], two["info"])
    lr = 3e-4
    for k in one.files:
        if not k.startswith(("critic", "actor", "enc", "temp")):
            continue
        a, b = one[k].astype(np.float64), two[k].astype(np.float64)
        err = np.abs(a - b)
        scale = max(np.abs(a).max(), 1e-30)
        # Adam's sign-like first steps make isolated elements with |g| ~ 1e-8 differ by up to 2*lr per step (fp32 sum
        # order of the two half-batch gradients): the bulk must agree closely, every element within the Adam bound
        assert np.quantile(err, 0.999) / scale < 1e-4, (k, np.quantile(err, 0.999) / scale)
        assert err.max() <= 2.1 * lr * 9 + 1e-4 * scale, (k, err.max())


def test_trunk_farm_is_bit_identical_to_one_gpu(gpu, tmp_path):
    """serl_amd/parallel.py TrunkFarmLearner with two real processes (rank 0 updates and never runs the trunk, rank 1 runs the
    frozen trunk of every batch and ships the features; gloo send / recv staged through the host, RCCL needs a GPU per rank):
    the trunk is frozen and its output stop-gradiented (vision/resnet_v1.py:286), so the updater's parameters must equal the
    single-process learner's TO THE BIT -- same index / crop / REDQ streams, same full-batch kernels, no gradient reduction."""
    one, farm = _run(tmp_path, 1), _run(tmp_path, 2, parallel="farm")
    assert int(one["step"]) == int(farm["step"]) == 9
    worker = np.load(str(tmp_path / "w2farm.rank1.npz"))
    assert int(worker["step"]) == 0                       # the worker never applied an update
    assert np.array_equal(farm["idx"], one["idx"]) and np.array_equal(worker["idx"], one["idx"])
    assert np.array_equal(farm["valid"], one["valid"]) and int(farm["size"]) == int(one["size"])
    assert np.array_equal(one["info"], farm["info"]), (one["info"], farm["info"])
    for k in one.files:
        if k.startswith(("critic", "actor", "enc", "temp")):
            assert np.array_equal(one[k].view(np.uint32), farm[k].view(np.uint32)), k
