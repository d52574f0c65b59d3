"""bench.py host logic that needs no GPU."""
import os
import subprocess
import sys

from tests.helpers import ROOT


def test_bench_gpus_n_started_as_plain_python_spawns_its_ranks():
    """`python bench.py --gpus 2` (the form the driver uses for N = 1) must start one rank per GPU itself, as a child
    torch.distributed.run, and pass the child's exit code through.  Without a GPU each rank stops at the loud
    "needs a HIP device" check -- which proves two ranks were started with RANK / WORLD_SIZE set."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env["MASTER_PORT"] = "29641"
    env["HIP_VISIBLE_DEVICES"] = ""  # also on a GPU box: make the ranks stop early
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "must be launched with" not in out
    assert out.count("bench.py needs a HIP device") >= 1, out[-2000:]
    assert "nproc-per-node" in out or "local_rank" in out or "torch.distributed" in out, out[-2000:]
