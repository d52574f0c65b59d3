#!/usr/bin/env python3
"""Per-rank compute time of the sharded step on ONE GPU: the P ranks of the bench graph are built as threads over the
in-process loopback world of tests/test_gpu_sharded_loopback.py (real plans, real send lists, one real exchange), then
every rank's step is timed ALONE with the exchange stubbed out (halo rows keep the values of the real exchange).
Gives the compute side of DESIGN.md section 6's scaling model: what a rank does per step besides waiting for xGMI.

    python scripts/exp_shard_compute.py [world ...] [--workload tiny] [--replicate-input-halo] [--train-layers L] [--schedule overlap|training]
                                        [--ranks 0,7]   (time only these ranks; all are built)
"""
import importlib
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402
from bench import WORKLOADS  # noqa: E402
from tests.test_gpu_sharded_loopback import LoopbackWorld  # noqa: E402

pkg = load_package()
ops = importlib.import_module("gnncpp_amd.ops")
capi = importlib.import_module("gnncpp_amd.capi")
shard = importlib.import_module("gnncpp_amd.shard")
dev = torch.device("cuda:0")


class NullDist:
    class _W:
        def wait(self):
            return True

    def all_to_all_single(self, out, inp, osz=None, isz=None, async_op=False):
        return self._W() if async_op else None

    def all_reduce(self, t, op=None):
        return None

    def barrier(self):
        return None


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    workload = "rmat10m_100m_f256"
    if "--workload" in sys.argv:
        workload = sys.argv[sys.argv.index("--workload") + 1]
        args = [a for a in args if a != workload]
    layers = 0
    if "--train-layers" in sys.argv:
        layers = int(sys.argv[sys.argv.index("--train-layers") + 1])
        args = [a for a in args if a != str(layers)] if args.count(str(layers)) == 1 else args
    schedule = "overlap"
    if "--schedule" in sys.argv:
        schedule = sys.argv[sys.argv.index("--schedule") + 1]
        args = [a for a in args if a != schedule]
    only = None
    if "--ranks" in sys.argv:
        tok = sys.argv[sys.argv.index("--ranks") + 1]
        only = [int(x) for x in tok.split(",")]
        args = [a for a in args if a != tok]
    worlds = [int(a) for a in args] or [8]
    n, e, F, abc, seed = WORKLOADS[workload]
    for world in worlds:
        lw = LoopbackWorld(world)
        runners, errors = [None] * world, []

        def rank_main(rank):
            try:
                torch.cuda.set_device(0)
                if layers:
                    r = shard.ShardedTrain(ops, capi, pkg, lw.rank_view(rank), dev, rank, world, n, e, F, abc, seed, 1024, layers)
                else:
                    r = shard.ShardedBench(ops, capi, pkg, lw.rank_view(rank), dev, rank, world, n, e, F, abc, seed, 1024,
                                           replicate_input_halo="--replicate-input-halo" in sys.argv, schedule=schedule)
                r.step()
                torch.cuda.synchronize()
                runners[rank] = r
            except Exception:  # noqa: BLE001
                import traceback
                errors.append(traceback.format_exc())
                lw.bar.abort()

        th = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errors:
            print(errors[0])
            sys.exit(1)
        for r in runners:
            if only is not None and r.rank not in only:
                continue
            r.dist = NullDist()
            if layers:
                r.net.dist = NullDist()
            else:
                r.set_schedule(schedule)
            for _ in range(2):
                r.step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                r.step(timed=True)
            b.record()
            torch.cuda.synchronize()
            kt = {k: round(v, 3) for k, v in r.kernel_times().items()}
            p = r.plan
            print(json.dumps({"world": world, "rank": r.rank, "schedule": schedule if not layers else f"train-layers {layers}", "exchange_chunks": p.n_chunks, "rows": p.n_local, "nnz": p.nnz_local, "halo_fwd": p.fwd.n_halo,
                              "halo_bwd": p.bwd.n_halo, "send_rows_fwd": int(p.fwd.send_idx.numel()),
                              "send_rows_bwd": int(p.bwd.send_idx.numel()),
                              "send_per_peer_fwd": p.fwd.send_counts, "compute_ms_per_step": round(a.elapsed_time(b) / 5, 3),
                              "kernels_ms": kt}), flush=True)
        del runners
        ops._ws_cache.clear()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
