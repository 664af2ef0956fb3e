"""GPU: `bench.py --gpus N` starts its own ranks (torch.distributed.run, one process per GPU, RCCL group) and never
silently measures a smaller world (reference collective: common/common.py:213-214)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--fill", "600", "--capacity", "2000"]


def _bench(*args, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("overlap,n_ar", [("off", 2), ("on", 3)])
def test_one_rank_through_the_launcher(gpu, overlap, n_ar):
    r = _bench("--gpus", "1", "--force-launcher", "--overlap-reduce", overlap, *SMALL)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 3
    # --parallel auto (the default) is north_star's partition for every N: batch-sharded DP + RCCL all-reduce;
    # the trunk farm is only ever an explicit --parallel farm
    assert out["config"]["parallelism"] == "dp1"
    assert set(out["projection"]["by_n_gpus"]) == {"2", "4", "8"} and "not this run" in out["projection"]["source"]
    c = out["collective"]
    assert c["world_size"] == 1 and c["launcher"] == "torch.distributed.run" and c["backend"].startswith("nccl")
    # per step (CAR = 1): the critic gradients (default: one all-reduce on the update stream; opt-in: two overlapped buckets
    # [ensemble | head | proprio | scalars], then the encoder heads) and one [scalars | actor grads] all-reduce
    assert c["all_reduces_per_step"] == n_ar
    assert c["bytes_per_step"] > 17.5e6
    assert len(c["avg_us_by_bytes"]) == n_ar


def test_refuses_more_ranks_than_gpus(gpu):
    import torch
    n = torch.cuda.device_count() + 1
    r = _bench("--gpus", str(n), *SMALL, timeout=120)
    assert r.returncode != 0
    assert "needs" in r.stderr and "GPUs" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines()), "no bench line may be printed for a refused job"


def test_world_size_mismatch_is_an_error(gpu):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *SMALL], capture_output=True,
                       text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
