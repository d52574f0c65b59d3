"""Multi-GPU path on CPU: world_size-2/3 gloo jobs run the product partition + halo plan + exchange
(gnn.cpp_amd/shard.py) and check bit-identity with the unsharded oracle (tests/shard_worker.py)."""
import importlib
import os
import subprocess
import sys

import pytest
import torch

from tests.helpers import ROOT

shard = importlib.import_module("gnncpp_amd.shard")


def run_world(world, port, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    e["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "SHARD_OK" in r.stdout, r.stdout[-2000:]
    return r.stdout


@pytest.mark.parametrize("world,port", [(2, 29621), (3, 29622)])
def test_sharded_aggregation_bit_identical(world, port):
    run_world(world, port)


@pytest.mark.parametrize("world,port,chunks", [(2, 29625, 4), (3, 29626, 3)])
def test_sharded_aggregation_pipelined_exchange_chunks(world, port, chunks):
    """The training schedule's pipelined exchange: local rows cut into row chunks, halo tail and send buffer chunk-major, one
    all-to-all-v per chunk issued as soon as that chunk's rows exist -- norm, forward and backward aggregation stay bit-identical."""
    run_world(world, port, CHUNKS=chunks)


def test_sharded_aggregation_bit_identical_contiguous_partition():
    """The round-1 partition (contiguous original-id ranges) stays available and exact."""
    run_world(2, 29624, PARTITION="contiguous")


def test_sharded_tiny_graph_with_empty_halos():
    """More ranks than structure: 40 nodes, 60 edges over 2 ranks (some peers exchange nothing)."""
    run_world(2, 29623, N=40, E=60, F=5)


def test_balanced_cuts():
    w = torch.tensor([1] * 10, dtype=torch.int64)
    assert shard.balanced_cuts(w, 2) == [0, 5, 10]
    w = torch.tensor([100, 1, 1, 1, 1, 1, 1, 1], dtype=torch.int64)
    c = shard.balanced_cuts(w, 4)
    assert c[0] == 0 and c[-1] == 8 and c == sorted(c)
    assert shard.balanced_cuts(torch.zeros(0, dtype=torch.int64), 3) == [0, 0, 0, 0]


def test_deal_partition_balances_rows_and_weight():
    g = torch.Generator().manual_seed(3)
    w = (torch.rand(10_001, generator=g) ** 8 * 5000).to(torch.int64) + 1  # heavy tail
    for world in (2, 3, 8):
        owner, nid, cuts = shard.deal_partition(w, world)
        counts = torch.bincount(owner.long(), minlength=world)
        assert int(counts.max() - counts.min()) <= 1 and cuts[-1] == w.numel()
        loads = torch.zeros(world, dtype=torch.int64).index_add_(0, owner.long(), w)
        assert float(loads.max()) / float(loads.float().mean()) < 1.02
        assert sorted(nid.tolist()) == list(range(w.numel()))  # a permutation
        _, nid_asc, _ = shard.deal_partition(w, world, scramble=False)
        for p in range(world):  # rank-contiguous new ids; un-scrambled: ascending original order inside a rank; scrambled: spread
            mine = torch.nonzero(owner == p).reshape(-1)
            assert nid_asc[mine].tolist() == list(range(cuts[p], cuts[p + 1]))
            assert sorted(nid[mine].tolist()) == list(range(cuts[p], cuts[p + 1]))
            k = nid_asc[mine].long() - cuts[p]
            assert torch.equal(nid[mine].long(), cuts[p] + (k * shard.SCRAMBLE_MUL) % (cuts[p + 1] - cuts[p]))
    owner, nid, cuts = shard.deal_partition(w, 1, scramble=False)
    assert nid.tolist() == list(range(w.numel())) and cuts == [0, w.numel()]
    owner, nid, cuts = shard.deal_partition(w, 1)      # one rank: the single-GPU relabelling
    n1 = w.numel()
    assert nid.tolist() == [(v * shard.SCRAMBLE_MUL) % n1 for v in range(n1)] and sorted(nid.tolist()) == list(range(n1))
    # ties are dealt by ascending id, heaviest first: weights 5 5 1 1 over 2 ranks -> snake 0 1 1 0
    owner, _, _ = shard.deal_partition(torch.tensor([1, 5, 1, 5]), 2)
    assert owner.tolist() == [1, 0, 0, 1]
