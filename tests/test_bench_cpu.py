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


def test_roofline_traffic_comes_from_the_newest_profile_of_the_same_graph():
    """bench.py's `traffic` = bytes leaving L2 per aggregation, summed over the aggregation's two kernels (hub + streaming) of the newest
    committed rocprofv3 PMC summary of the workload, with the file's provenance (commit of the profiled build, nnz, mtime) beside it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod_cpu", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pats = [(r"^spmm_hub_kernel<\d+, 0, ", True), (r"^spmm_hubpc_kernel<0, ", False), (r"^spmm_(stream_)?kernel<\d+, \d+, \d+, 0, ", True)]
    import hashlib
    spmm_sha = hashlib.sha256(open(os.path.join(ROOT, "gnn.cpp_amd", "csrc", "gnnx_spmm.hip"), "rb").read()).hexdigest()
    for wl, lo, hi in (("rmat10m_100m_f256", 100e9, 112.2e9), ("rmat1m_10m_f128", 3e9, 5.7e9), ("products_2p4m_62m_f100", 26e9, 36e9)):
        tr = bench.profiled_traffic(wl, pats)
        assert tr is not None, wl
        total, src = tr
        assert lo < total < hi, (wl, total)
        assert src["file"].startswith("profiles/r05_") and len(src["kernels"]) in (2, 3) and src["profiled_nnz"] and src["git_head_of_profiled_build"]
        assert any("spmm_hub_kernel" in k for k in src["kernels"]) and any("spmm_stream_kernel" in k for k in src["kernels"])
        # bench.py refuses a profile made with another aggregation source: the committed one must match the tree
        assert src["spmm_source_sha256_of_profiled_build"] == spmm_sha, (wl, "re-run scripts/profile_all.sh: gnnx_spmm.hip changed since the profile")
    assert bench.profiled_traffic("no_such_workload", pats) is None
    # the algorithmic byte count of SURVEY.md 8(d) for the headline graph
    assert bench.spmm_bytes(10_000_000, 10_000_000, 99_100_605, 256, bias=True) == 112_195_422_968


def test_hbm_side_model_bounds_and_limits():
    """bench.hbm_side_model (the HBM-pin estimate of the roofline line): between the cache-perfect bound (every row once) and the
    algorithmic count (one row per edge); a graph whose gathered matrix fits the cache is charged one touch per row; rows used once
    are never resident; a cache of zero bytes gives the algorithmic count."""
    import importlib.util

    import torch
    spec = importlib.util.spec_from_file_location("bench_mod_cpu2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    F, n = 256, 200_000
    g = torch.Generator().manual_seed(3)
    deg = torch.distributions.Pareto(torch.tensor(1.0), torch.tensor(1.1)).sample((n,)).to(torch.int64).clamp(max=50_000)
    nnz = int(deg.sum())
    streams = 4 * (n + 1) + 4 * nnz + 4 * n + 4 * F * n
    m = bench.hbm_side_model(deg, nnz, n, F)
    assert streams + 4 * F * int((deg > 0).sum()) <= m["hbm_bytes"] + 4 * F * n and m["hbm_bytes"] <= streams + 4 * F * nnz
    # everything fits (n x 1 KiB = 200 MB < 256 MiB, and the one-touch traffic between two uses of a hot row is small): most edges hit
    assert m["resident_edge_share"] > 0.5 and m["resident_rows"] > 0
    zero = bench.hbm_side_model(deg, nnz, n, F, cache_bytes=0)
    assert zero["resident_rows"] == 0 and zero["hbm_bytes"] == streams + 4 * F * nnz
    once = bench.hbm_side_model(torch.ones(n, dtype=torch.int64), n, n, F)
    assert once["resident_rows"] == 0 and once["hbm_bytes"] == 4 * (n + 1) + 4 * n + 4 * n + 4 * F * n + 4 * F * n
    bigger = bench.hbm_side_model(deg, nnz, n, F, cache_bytes=2 * bench.INFINITY_CACHE_BYTES)
    assert bigger["hbm_bytes"] <= m["hbm_bytes"] and bigger["resident_rows"] >= m["resident_rows"]


def test_row_chunks_fall_on_whole_rounds_of_tiles():
    """shard.chunk_bounds: the row chunks of the pipelined exchange -- plain n k / K cuts for small shards, the nearest multiple of a
    round of tiles (256 CUs x 256 rows) once a chunk is two rounds long; always n_chunks + 1 ascending bounds from 0 to n."""
    import importlib

    from __graft_entry__ import load_package
    load_package()
    shard = importlib.import_module("gnncpp_amd.shard")
    assert shard.chunk_bounds(1000, 4) == [0, 250, 500, 750, 1000]
    assert shard.chunk_bounds(0, 4) == [0, 0, 0, 0, 0] and shard.chunk_bounds(7, 1) == [0, 7]
    R = shard.ROUND_ROWS
    assert shard.chunk_bounds(1_250_000, 4) == [0, 5 * R, 10 * R, 14 * R, 1_250_000]
    for n in (2 * R * 4, 2 * R * 4 + 1, 1_250_000, 10_000_000, 123_456_789):
        for K in (2, 3, 4, 8):
            b = shard.chunk_bounds(n, K)
            assert len(b) == K + 1 and b[0] == 0 and b[-1] == n and all(x < y for x, y in zip(b, b[1:]))
            if n // K >= 2 * R:
                assert all(x % R == 0 for x in b[1:-1])
