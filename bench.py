#!/usr/bin/env python3
"""bench.py -- GCN-layer edges/sec + achieved HBM GB/s vs roofline on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the synthetic graph: one GCN layer forward
(H = X.W^T ; Out = norm (.) (A.H) + bias) and its backward (dbias, dH = A^T.(norm (.) G), dX = dH.W,
dW = dH^T.X) -- SURVEY.md section 8 rows a1-a11.  Graph build (COO->CSR, transpose, norm, plans, halo plan)
is outside the timed region and reported separately.  value = nnz (edges after dedupe / self-loop strip,
what the reference semantics sum over) / time per step, whole job.

Workload at N = 1: BASELINE configs[3], the one the metric's target is quoted on (RMAT 10M nodes / 100M
edges, 256 features); it fits one GPU (~60 GB).  At N > 1 the same graph is sharded by 1-D vertex partition
(degree-sorted snake deal: equal rows, non-zeros and per-link halo volume) with a halo exchange per aggregation
(RCCL all-to-all-v), both exchanges asynchronous and each chain's compute under the other chain's exchange
=> "scaling": "strong".  Started as plain `python bench.py --gpus N` it spawns torch.distributed.run itself.

Besides the contract fields the JSON line carries
  roofline     for the dominant kernel (forward SpMM): algorithmic bytes B_gather / HIP-event time
  cpu_baseline the CPU oracle (the port of the reference arithmetic) timed on this box's host cores on a
               bounded sample (rank 0, N = 1 only)
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

# multi-process GPU work on this driver stack shares device memory through dmabuf IPC only (RCCL's peer mappings need it); the
# HSA runtime reads the switch when the first HIP call initialises it, so it is set before torch is imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s

WORKLOADS = {
    # name: (n_nodes, n_edges, F, rmat (a,b,c), seed)
    "rmat10m_100m_f256": (10_000_000, 100_000_000, 256, (0.57, 0.19, 0.19), 2),   # BASELINE configs[3]
    "rmat1m_10m_f128": (1_000_000, 10_000_000, 128, (0.57, 0.19, 0.19), 1),       # BASELINE configs[2]
    "products_2p4m_62m_f100": (2_400_000, 62_000_000, 100, (0.45, 0.22, 0.22), 3),  # BASELINE configs[4]
    "cora_2708_10556": (2708, 10556, 16, None, 42),                                # BASELINE configs[1] (layer 2 width)
    "tiny": (20_000, 200_000, 64, (0.57, 0.19, 0.19), 5),
}


def profiled_traffic(workload, kernel_patterns):
    """Per-launch bytes leaving L2 (PMC) of ONE aggregation = the sum over its kernels (hub kernels + streaming kernel), from the
    newest committed rocprofv3 PMC summary of this workload (profiles/*_hbm_traffic.json, made by scripts/summarize_profile.py from
    separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same bench command, gfx950-corrected).  kernel_patterns: (regular
    expression, required) pairs; a required one must match exactly one kernel of the file, an optional one at most one (the
    producer / consumer hub kernel only runs when the graph has rows long enough).  Returns (bytes, provenance dict) or None."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):   # names sort by round tag: the last match wins
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        total, names, l2_hit, l2_miss, l2_per = 0.0, [], 0.0, 0.0, {}
        for pat, required in kernel_patterns:
            hits = [k for k in d.get("kernels", {}) if re.search(pat, k)]
            if len(hits) > 1 or (required and len(hits) != 1):
                break
            for h in hits:
                total += d["kernels"][h]["hbm_bytes_corrected"]
                names.append(h)
                if d["kernels"][h].get("tcc_hit") is not None:   # L2 hit rate (TCC_HIT_sum / TCC_MISS_sum pass, MI355X_MICROARCH.md section L2)
                    l2_hit += d["kernels"][h]["tcc_hit"]
                    l2_miss += d["kernels"][h]["tcc_miss"]
                    l2_per[h] = d["kernels"][h].get("l2_hit_rate")
        else:
            best = (total, {"file": os.path.relpath(f, ROOT), "kernels": names, "git_head_of_profiled_build": d.get("git_head"),
                            "l2_hit_rate": (l2_hit / (l2_hit + l2_miss)) if (l2_hit + l2_miss) > 0 else None, "l2_hit_rate_per_kernel": l2_per or None,
                            "spmm_source_sha256_of_profiled_build": d.get("spmm_source_sha256"),
                            "profiled_nnz": d.get("nnz"), "file_mtime": time.strftime("%Y-%m-%d %H:%M:%S", time.gmtime(os.path.getmtime(f)))})
    return best


_ceil_lib = None


def _ceilings_lib():
    """bench_kernels/libbench_ceilings.so: the two measurement kernels (float4 copy, whole-row gather).  Measurement only."""
    global _ceil_lib
    if _ceil_lib is None:
        path = os.path.join(ROOT, "bench_kernels", "libbench_ceilings.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        _ceil_lib = C.CDLL(path)
        _ceil_lib.ceil_copy_f4.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _ceil_lib.ceil_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    return _ceil_lib


def _timed_ms(capi, fn, reps=5, warmup=2):
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(warmup):
        rc = fn(stream)
        if rc != 0:
            raise RuntimeError(f"ceiling kernel launch failed ({rc})")
    a, b = capi.Event(), capi.Event()
    a.record(stream)
    for _ in range(reps):
        fn(stream)
    b.record(stream)
    b.sync()
    return a.elapsed_ms(b) / reps


def copy_ceiling(capi, dev, gib=10, variant=0):
    """float4 copy of `gib` GiB (read + written bytes / time), GB/s.  (10 GiB -- the size of one of the step's matrices: 6.0 TB/s; 4 GiB:
    5.7, the launch's ramp and tail weigh more.)"""
    L = _ceilings_lib()
    n = gib * 2 ** 30 // 4
    src = torch.full((n,), 1.0, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    ms = _timed_ms(capi, lambda st: L.ceil_copy_f4(src.data_ptr(), dst.data_ptr(), n * 4, variant, st))
    return 2 * n * 4 / (ms * 1e-3) / 1e9


def gather_ceiling(capi, dev, table_mb=160, gathered_gb=16, variant=0):
    """Sum of 8 uniformly random whole 1-KiB rows of a `table_mb` MB table per streamed output row: (gathered + index + written)
    bytes / time, GB/s.  With the table inside the 256 MiB Infinity Cache this is what the fabric behind L2 delivers to a row gather."""
    L = _ceilings_lib()
    deg = 8
    rows = table_mb * 2 ** 20 // 1024
    n_out = int(gathered_gb * 1e9 / 1024 / deg) // 256 * 256
    table = torch.full((rows, 256), 1.0, dtype=torch.float32, device=dev)
    idx = torch.randint(0, rows, (n_out * deg,), dtype=torch.int32, device=dev)
    out = torch.empty((n_out, 256), dtype=torch.float32, device=dev)
    ms = _timed_ms(capi, lambda st: L.ceil_gather_rows(table.data_ptr(), idx.data_ptr(), n_out, variant, out.data_ptr(), st))
    return (n_out * deg * 1024 + n_out * deg * 4 + n_out * 1024) / (ms * 1e-3) / 1e9


def measure_ceilings(capi, dev):
    """The in-run ceilings of the roofline line (N = 1, outside the timed region, a few ms of kernels each).  A ceiling is the best
    the box does: the maximum of three trials of five launches."""
    def best(fn):
        return max(fn() for _ in range(3))
    c = {"hbm_copy_GBps": best(lambda: copy_ceiling(capi, dev)),
         "fabric_gather_GBps": best(lambda: gather_ceiling(capi, dev, table_mb=160)),
         # for context: the same gather from a table small enough to sit whole in the Infinity Cache beside everything else (the
         # guide's 8.6 TB/s figure); the aggregation's hot set is 168 MB, so the 160 MB table is the ceiling it is held against
         "fabric_gather_38MB_GBps": best(lambda: gather_ceiling(capi, dev, table_mb=38)),
         "fabric_gather": "sum of 8 uniformly random 1-KiB rows of a 160 MB table (Infinity-Cache resident) per streamed output row; "
                          "gathered + index + written bytes / time (bench_kernels/ceilings.hip)",
         "hbm_copy": "float4 copy of 10 GiB, read + written bytes / time"}
    torch.cuda.empty_cache()
    return c


def spmm_bytes(n_rows, n_cols_rows_written, nnz, F, bias=True):
    """Algorithmic bytes of one SpMM launch, Mode REF: rowptr + colidx + one neighbour
    row per edge + rowscale + Y write (+ bias)."""
    return 4 * (n_rows + 1) + 4 * nnz + 4 * F * nnz + 4 * n_rows + 4 * F * n_cols_rows_written + (4 * F if bias else 0)


INFINITY_CACHE_BYTES = 256 * 2 ** 20   # MI355X_MICROARCH.md: 256 MiB die-level last-level cache


def hbm_side_model(in_degree, nnz, n_rows, F, bytes_per_feature=4, cache_bytes=INFINITY_CACHE_BYTES):
    """HBM-side bytes of ONE aggregation launch, computed from the graph itself (no counter sees the HBM pins: FETCH_SIZE counts
    Infinity-Cache hits, the algorithmic count charges every cache hit as an HBM read).

    Residency rule of MI355X_MICROARCH.md (Infinity Cache): a line stays resident only while the table it belongs to plus every byte
    loaded or stored between two uses of that line fits in ~256 MiB.  Gathered row c of the feature matrix is used in_degree[c] times
    per launch, spread over the launch, so between two of its uses flow 1 / in_degree[c] of the launch's ONE-TOUCH bytes (column
    indices, row pointers, scales, the written rows, and the gathers of rows that are NOT resident); the rows used at least as often
    as c are the table that must stay beside them.  Sort rows by in-degree, descending; the resident set is the longest prefix K with

        K * row_bytes  +  one_touch_bytes(K) / in_degree[K-th row]  <=  cache_bytes,     in_degree >= 2,

    one_touch_bytes(K) = streams + row_bytes * (non-zeros that point outside the prefix).  A resident row is charged ONCE (its first
    touch), every other gather per edge.  L2 hits are inside the resident share (a row that stays in a 4 MiB L2 stays in the 256 MiB
    cache behind it).  A model, not a measurement: it says what the HBM pins must deliver at least under that rule."""
    row_bytes = bytes_per_feature * F
    streams = 4 * (n_rows + 1) + 4 * nnz + 4 * n_rows + 4 * F * n_rows   # rowptr, colidx, rowscale, Y (f32 out)
    deg = torch.sort(in_degree.to(torch.int64), descending=True).values
    deg = deg[deg >= 2]
    if deg.numel() == 0:
        return {"hbm_bytes": streams + row_bytes * nnz, "resident_rows": 0, "resident_MB": 0.0, "resident_edge_share": 0.0}
    covered = torch.cumsum(deg, 0)                                        # non-zeros that point into the prefix
    k = torch.arange(1, deg.numel() + 1, device=deg.device, dtype=torch.float64)
    one_touch = streams + row_bytes * (nnz - covered).to(torch.float64)
    need = k * row_bytes + one_touch / deg.to(torch.float64)
    ok = need <= cache_bytes
    K = int(ok.to(torch.int64).sum().item()) if bool(ok[0]) else 0   # `need` grows with k on a sorted degree list: the prefix is the count
    if K and not bool(ok[:K].all()):
        K = int(torch.nonzero(~ok)[0].item())
    hot_edges = int(covered[K - 1].item()) if K else 0
    hbm = streams + row_bytes * K + row_bytes * (nnz - hot_edges)
    return {"hbm_bytes": float(hbm), "resident_rows": K, "resident_MB": K * row_bytes / 1e6,
            "resident_edge_share": hot_edges / max(1, nnz), "cache_bytes": cache_bytes,
            "rule": "resident prefix by in-degree: K*row_bytes + one_touch_bytes/in_degree[K] <= 256 MiB (MI355X_MICROARCH.md, Infinity Cache)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="rmat10m_100m_f256", choices=sorted(WORKLOADS))
    ap.add_argument("--chunk", type=int, default=1024,
                    help="plan: rows longer than this go to the sequential hub kernel (one accumulator per feature, the reference's order; 0 = no plan)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-layers", type=int, default=0,
                    help="N=1 only: time a whole training step of an L-layer GCN instead (forward, softmax-CE, backward, SGD); "
                         "value counts L*nnz edges per step")
    ap.add_argument("--vertex-order", default="auto", choices=["auto", "scrambled", "as-generated"],
                    help="scrambled: the graph builder stores vertex v at row (v * 2654435761) mod n -- R-MAT puts its hubs on the ids "
                         "with few one-bits, and with a power-of-two row stride their feature rows pile onto a few memory channels; every "
                         "vertex's result has the same bits, at another row.  as-generated: vertex v at row v (round 1).  auto (default): "
                         "scrambled when a feature row is a multiple of 512 bytes (F = 256: aggregation 19 -> 13.7 ms; F = 128: -3 %%), "
                         "as-generated otherwise (F = 100: 400-byte rows spread by themselves, the scramble costs 2 %%)")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the in-run copy / gather ceilings (roofline.peak falls back to the HBM spec)")
    ap.add_argument("--no-cpp-api", action="store_true",
                    help="skip the C++ call-site leg (tests/cpp/bench_host_api as a child process after the timed region: cpp_api)")
    ap.add_argument("--no-order-control", action="store_true",
                    help="skip the informational re-run of a few steps on the as-generated vertex order (vertex_order_control)")
    ap.add_argument("--no-pad-features", action="store_true",
                    help="store feature rows at their own width even when it is not a multiple of 128 floats (default: pad the stride)")
    ap.add_argument("--sym", action="store_true",
                    help="N=1: Mode SYM, the textbook D^-1/2 A D^-1/2 aggregation of the north_star (per-edge scale) instead of the "
                         "reference's factorised norm (Mode REF, the parity-graded default); same kernel, +4 B/edge of traffic")
    ap.add_argument("--bf16-features", action="store_true",
                    help="N=1: opt-in bf16 FEATURE STORAGE for the two aggregations (f32 accumulate); halves the gather bytes but "
                         "rounds features to 8 bits -- outside the 1e-5 parity bar, never the default")
    ap.add_argument("--split-gemm", action="store_true",
                    help="N=1: OPT-IN split-precision GEMM for X.W^T and dH.W (bf16 matrix cores, exact 3-way operand split, f32 "
                         "accumulation: f32-level accuracy but not the reference's arithmetic) -- never the default")
    ap.add_argument("--hip-graph", action="store_true",
                    help="N=1: capture one step (6 kernel launches + their small helpers) into a hipGraph and replay it in the "
                         "timed loop -- for launch-bound sizes such as the Cora-sized config")
    ap.add_argument("--native-comm", action="store_true",
                    help="N>1: halo all-to-all-v and all-reduce through the C-ABI (gnnx_halo_exchange_f32, RCCL send/recv group on a "
                         "communication stream of its own) instead of torch.distributed.  UNVERIFIED for more than one rank: no "
                         "multi-GPU node was available to the builder (one-rank communicator and the in-process transport are tested)")
    ap.add_argument("--force-sharded", action="store_true", help="run the N>1 code path even with one rank (rehearsal)")
    ap.add_argument("--schedule", default="overlap", choices=["overlap", "training", "sequential"],
                    help="N>1: overlap (default) = the step as N = 1 defines it: X and the upstream gradient G are both INPUTS of the layer "
                         "step, so the two chains are independent and each chain's compute runs under the other chain's halo exchange "
                         "(same bits); training = the dependence of a real training step honoured (forward -> loss -> backward: nothing of "
                         "the backward chain starts before the forward aggregation has finished): the transform runs in --exchange-chunks "
                         "row chunks and every finished chunk's rows are sent while the next is multiplied, G's rows leave chunk by chunk, "
                         "dbias runs under the exchange; sequential = GEMM -> exchange -> SpMM ... on one stream")
    ap.add_argument("--exchange-chunks", type=int, default=0,
                    help="N>1: row chunks of the pipelined halo exchange (chunk-major halo tail; 0 = 4 for --schedule training and for "
                         "--train-layers, 1 otherwise)")
    ap.add_argument("--partition", default="deal", choices=["deal", "deal-ascending", "contiguous"],
                    help="N>1: deal = degree-sorted snake deal (equal rows / non-zeros / per-link volume); contiguous = ranges of "
                         "original ids balanced on degree (round 1)")
    ap.add_argument("--replicate-input-halo", action="store_true",
                    help="N>1, OPT-IN, valid for a FIRST layer only (its input is data, the same every step): fetch the halo rows of X once "
                         "and compute their X.W^T locally every step (same bits) -- one halo exchange per step instead of two")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1: nccl (= RCCL over xGMI, the measured path) or gloo = REHEARSAL: device tensors staged through host "
                         "memory over a CPU process group, so that the whole multi-process job can run on a box with one GPU")
    ap.add_argument("--all-ranks-on-device0", action="store_true",
                    help="rehearsal: every rank uses GPU 0 (needs --dist-backend gloo; RCCL refuses two ranks on one device)")
    ap.add_argument("--cpu-sample-nodes", type=int, default=0,
                    help="0 (default): the CPU baseline runs the bench's OWN workload -- graph, features and gradient copied from the "
                         "device -- on all host threads (about 35 s at 10 M / 100 M / 256 on 128 threads); > 0: a sample of that many "
                         "nodes from the same generator (same average degree).  The single-thread leg always runs on 1/40 of the nodes")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # started as plain `python bench.py --gpus N`: start the ranks ourselves -- a FRESH child process running
            # torch.distributed.run, before anything in this process has touched the GPU -- and pass its JSON line and
            # exit code through
            import subprocess
            port = os.environ.get("MASTER_PORT", "29533")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                   "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            sys.exit(subprocess.call(cmd, env=env))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device: the product path has no CPU fallback")
    if args.all_ranks_on_device0:
        if args.dist_backend != "gloo":
            sys.exit("--all-ranks-on-device0 is a rehearsal mode and needs --dist-backend gloo")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        try:
            if args.dist_backend == "gloo":
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
                dist.barrier()   # the first collective creates the RCCL communicator: fail here, loudly, not inside the timed loop
        except Exception as ex:  # no retry, no re-exec: report the transport's own message and exit non-zero
            print(f"bench.py: rank {rank}/{world}: {args.dist_backend} process group could not be initialised: {ex}", file=sys.stderr, flush=True)
            sys.exit(2)

    pkg = load_package()
    if dist is not None and args.dist_backend == "gloo":   # rehearsal transport (shard.HostStagedDist): never the measured path
        dist = importlib.import_module("gnncpp_amd.shard").HostStagedDist(dist)
    ops = importlib.import_module("gnncpp_amd.ops")
    capi = importlib.import_module("gnncpp_amd.capi")

    n, e, F, abc, seed = WORKLOADS[args.workload]
    relabel = "scramble" if args.vertex_order == "scrambled" or (args.vertex_order == "auto" and (4 * F) % 512 == 0) else None
    t_build0 = time.time()
    if world == 1 and not args.force_sharded and args.train_layers > 0:
        runner = TrainStep(ops, capi, pkg, dev, n, e, F, abc, seed, args.chunk, args.train_layers, relabel=relabel)
        runner.workload = args.workload
    elif world == 1 and not args.force_sharded:
        runner = SingleGpu(ops, capi, pkg, dev, n, e, F, abc, seed, args.chunk, pad=not args.no_pad_features, relabel=relabel)
        runner.workload = args.workload
        runner.sym = args.sym
        runner.bf16_features = args.bf16_features
        runner.split_gemm = args.split_gemm
    elif args.train_layers > 0:
        shard = importlib.import_module("gnncpp_amd.shard")
        runner = shard.ShardedTrain(ops, capi, pkg, dist, dev, rank, world, n, e, F, abc, seed, args.chunk, args.train_layers,
                                    partition=args.partition, n_chunks=args.exchange_chunks or 4)
    else:
        shard = importlib.import_module("gnncpp_amd.shard")
        runner = shard.ShardedBench(ops, capi, pkg, dist, dev, rank, world, n, e, F, abc, seed, args.chunk, native_comm=args.native_comm,
                                    schedule=args.schedule, partition=args.partition, replicate_input_halo=args.replicate_input_halo,
                                    n_chunks=args.exchange_chunks or None)
    torch.cuda.synchronize()
    t_build = time.time() - t_build0

    for _ in range(args.warmup):
        runner.step()
    torch.cuda.synchronize()
    graph = None
    if args.hip_graph and world == 1 and not args.force_sharded and not args.train_layers:
        # the C-ABI compute calls neither allocate nor synchronise, so a step is capturable as is (workspaces were
        # grown by the warm-up); capture runs on a side stream, replay on the current one
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            runner.step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            runner.step()
        torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            graph.replay()
        else:
            runner.step(timed=True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    nnz_total = runner.nnz_total * max(1, args.train_layers)
    if graph is not None:  # per-kernel HIP events are not recorded inside a replayed graph: time the kernels once eagerly
        runner.step(timed=True)
        torch.cuda.synchronize()
    ceilings = None
    if world == 1 and not args.force_sharded and not args.no_ceilings and rank == 0:
        ceilings = measure_ceilings(capi, dev)   # outside the timed region: two small kernels, a few ms each
    if isinstance(runner, SingleGpu):
        roof = runner.roofline(ceilings)
    else:
        roof = sharded_roofline(runner.aggregation_stats())
        if roof["achieved"] is not None:  # slowest rank's forward SpMM defines the job's roofline line
            t = torch.tensor([roof["achieved"]], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            roof["achieved"] = float(t.item())
            roof["frac"] = roof["achieved"] / roof["peak"]

    roof_mfma = runner.mfma_roofline() if hasattr(runner, "mfma_roofline") else None
    kernels_ms = runner.kernel_times()
    nnz_one, Fp_run, order_run = runner.nnz_total, getattr(runner, "Fp", F), getattr(runner, "vertex_order", None)
    runner_chunks = getattr(getattr(runner, "plan", None), "n_chunks", 1)
    # control for the vertex order (informational, outside the timed region above): the same runner on the as-generated labels
    order_control = None
    if (world == 1 and not args.force_sharded and not args.train_layers and relabel and graph is None and not args.no_order_control
            and isinstance(runner, SingleGpu)):
        order_control = runner.as_generated_control(pkg, n, e, abc, seed, args.chunk)

    if order_control and roof.get("avg_launch_ms") and order_control.get("spmm_fwd_ms"):
        # the as-generated order: hub rows pile onto a few channels, HBM-side bound -> algorithmic bytes against the HBM spec
        order_control["hbm_spec_frac_as_generated"] = (roof["algorithmic_bytes_per_launch"] / (order_control["spmm_fwd_ms"] * 1e-3) / 1e9
                                                       / HBM_PEAK_GBS)
    cpu = cpu_ref = cpp = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(pkg, args, F, abc, seed, runner=runner)
        cpu_ref = cpu_reference(pkg)
    if (rank == 0 and world == 1 and not args.force_sharded and not args.train_layers and not args.no_cpp_api and abc is not None
            and abc == (0.57, 0.19, 0.19)):
        # free this process's device memory first: the child builds the same graph and tensors through the C++ API
        del runner
        ops._ws_cache.clear()
        torch.cuda.empty_cache()
        cpp = cpp_api_bench(n, e, F, seed, both_orders=(4 * F) % 512 == 0)

    if rank == 0:
        line = {
            "metric": "GCN-layer edges/sec (fwd+bwd: X.W^T + normalized-adjacency SpMM + their gradients)",
            "value": nnz_total / (ms_per_step * 1e-3),
            "unit": "edges/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "n_nodes": n, "n_edges_generated": e, "nnz": nnz_one,
                       "features": F, "layer": f"{F}->{F}", "feature_row_stride": Fp_run,
                       "vertex_order": order_run or "dealt by degree, spread inside every rank's range",
                       "step": "layer fwd+bwd" if not args.train_layers else
                       f"{args.train_layers}-layer GCN training step (fwd, softmax-CE, bwd of every parameter, SGD; no input gradient); value counts {args.train_layers}*nnz",
                       "parallelism": "single" if world == 1 and not args.force_sharded else
                       f"1-D vertex shard x{world} ({args.partition}), halo all-to-all-v, schedule {'training' if args.train_layers else args.schedule}" +
                       (" (X and G both inputs of the step, as at N = 1: each chain's compute under the other chain's exchange)"
                        if args.schedule == "overlap" and not args.train_layers else "") +
                       (f" (forward -> backward dependence honoured; exchanges pipelined in {runner_chunks} row chunks)"
                        if (args.schedule == "training" or args.train_layers) else "") +
                       (", input halo replicated (first-layer form: one exchange per step)" if args.replicate_input_halo else "") +
                       (" -- REHEARSAL: gloo through host memory, all ranks on GPU 0: not a measurement" if args.dist_backend == "gloo" else ""),
                       "plan_chunk": args.chunk, "hip_graph": graph is not None, "mode": "SYM" if args.sym else "REF",
                       "feature_storage": "bf16 for the aggregations (opt-in, NOT the parity path)" if args.bf16_features else "f32",
                       "gemm": "split-bf16 x6 for X.W^T and dH.W (opt-in, NOT the parity path)" if args.split_gemm else "f32 MFMA"},
            "roofline": roof,
            "roofline_mfma": roof_mfma,
            "vertex_order_control": order_control,
            "cpp_api": cpp,
            "cpu_baseline": cpu,
            "cpu_reference": cpu_ref,
            "kernels_ms": kernels_ms,
            "build_s": t_build,
            "device": torch.cuda.get_device_name(local_rank),
        }
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


def sharded_roofline(st):
    """N > 1: the slowest rank's forward aggregation, ALGORITHMIC bytes (one feature row per local non-zero) / time against the HBM
    spec -- no PMC traffic and no in-run ceiling exist for a shard (the N = 1 line carries those)."""
    r = {"bound": "hbm", "bound_is": "fabric behind L2",
         "kernel": "forward aggregation of one rank (spmm_hub_kernel + spmm_stream_kernel, spmm_hubpc_kernel beside), slowest rank",
         "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
         "frac_is": "ALGORITHMIC bytes (one feature row per non-zero: every cache hit charged as an HBM read, so above 1 is possible) / "
                    "launch time / HBM spec -- a shard has no PMC traffic figure; the N = 1 line carries the three readings side by side",
         "achieved_is": "algorithmic_bytes_per_launch / avg_launch_ms", "peak_is": "HBM spec"}
    if st.get("spmm_fwd_ms"):
        B = spmm_bytes(st["local_rows"], st["local_rows"], st["local_nnz"], st["features"], bias=True)
        r.update(achieved=B / (st["spmm_fwd_ms"] * 1e-3) / 1e9, algorithmic_bytes_per_launch=B, avg_launch_ms=st["spmm_fwd_ms"])
        r["frac"] = r["achieved"] / r["peak"]
    else:
        r["kernel"] = "n/a for --train-layers (see the default run)"
    r.update({k: v for k, v in st.items() if k.startswith(("halo_", "send_")) or k in ("local_rows", "schedule")})
    return r


class SingleGpu:
    """Whole graph on one MI355X.  All buffers are allocated here, outside the timed region."""

    def __init__(self, ops, capi, pkg, dev, n, e, F, abc, seed, chunk, pad=True, relabel="scramble"):
        self.ops, self.capi, self.F, self.n = ops, capi, F, n
        self.vertex_order = "scrambled" if relabel else "as-generated"
        # When the width is not a multiple of 128 floats (the products-shaped F = 100) the rows that are only STREAMED -- X, dH, dX,
        # and W / dW -- are stored with a 128-float stride (pad columns zero, and they stay zero), so the two backward products run
        # as 128-wide ones on the LDS-DMA kernels (zero columns add exact zeros at the end of every fmaf chain: the same bits in
        # the first F columns).  The rows that are GATHERED -- H and G -- keep their own width: measured, a 512-byte stride there
        # costs the aggregation 11 % (5.03 vs 4.53 ms; 28 % more footprint in L2 / MALL), more than all three products gain;
        # X.W^T (K = 128 padded, N = F) writes the F-wide H through the LDS-DMA kernel's guarded last column tile.
        Fp = -(-F // 128) * 128 if (pad and F % 128 and F >= 64 and n >= 100_000) else F
        self.Fp = Fp
        if abc is None:
            s, d = pkg.synth.uniform_edges(seed, n, e)
            src, dst = torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)
        else:
            src, dst = ops.rmat_edges(seed, n, e, *abc, device=dev)
        # X and G below are synthetic rows drawn directly in the graph's row order (to compare with an as-generated run, move
        # them with g.to_new_order / g.to_vertex_order: tests/test_gpu_parity.py::test_vertex_relabelling_same_bits)
        self.g = g = ops.CsrGraph.from_coo(src, dst, n, relabel=relabel)
        del src, dst
        ops._ws_cache.clear()
        torch.cuda.empty_cache()
        if chunk > 0:
            g.make_plans(chunk, F)
        self.nnz_total = g.nnz
        def padded(t, rows):   # [rows, Fp] zero-padded copy of t (or t itself); the GEMMs take the padded tensor, everything
            if Fp == F:        # else its [:, :F] view
                return t
            p = torch.zeros((rows, Fp), dtype=torch.float32, device=dev)
            p[:, :F] = t
            return p

        self.Xp = padded(ops.uniform_pm1(seed + 10, (n, F), device=dev), n)
        Wf = ops.uniform_pm1(seed + 11, (F, F), scale=F ** -0.5, device=dev)
        self.Wp = padded(Wf, F) if Fp == F else torch.zeros((Fp, Fp), dtype=torch.float32, device=dev)
        if Fp != F:
            self.Wp[:F, :F] = Wf
        self.bias = torch.zeros(F, dtype=torch.float32, device=dev)  # graph.cpp:167
        self.Gp = ops.uniform_pm1(seed + 12, (n, F), device=dev)
        alloc = torch.zeros if Fp != F else torch.empty   # pad columns must start (and stay) zero
        self.Hp = torch.empty((n, F), dtype=torch.float32, device=dev)
        self.outp = torch.empty((n, F), dtype=torch.float32, device=dev)
        self.dHp = alloc((n, Fp), dtype=torch.float32, device=dev)
        self.dXp = alloc((n, Fp), dtype=torch.float32, device=dev)
        self.dWp = alloc((Fp, Fp), dtype=torch.float32, device=dev)
        self.X, self.W, self.G = self.Xp[:, :F], self.Wp[:F, :F], self.Gp[:, :F]
        self.H, self.out, self.dH, self.dX, self.dW = (self.Hp[:, :F], self.outp[:, :F], self.dHp[:, :F], self.dXp[:, :F],
                                                        self.dWp[:F, :F])
        self.dbias = torch.empty(F, dtype=torch.float32, device=dev)
        self.names = ["colsum", "spmm_bwd", "gemm_dX", "gemm_dW", "gemm_xwT", "spmm_fwd"]   # launch order of step()
        self.ev = []  # per timed step: list of (start, stop) HIP events on the launch stream

    def step(self, timed=False):
        ops, g = self.ops, self.g
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        evs = []

        def run(fn):
            if timed:
                a, b = self.capi.Event(), self.capi.Event()
                a.record(stream)
                fn()
                b.record(stream)
                evs.append((a, b))
            else:
                fn()

        sym = getattr(self, "sym", False)
        if getattr(self, "bf16_features", False):
            # opt-in bf16 feature storage: the aggregation gathers 2-byte features (not the parity path)
            if not hasattr(self, "Hb"):
                self.Hb = torch.empty(self.H.shape, dtype=torch.bfloat16, device=self.H.device)
                self.Gb = torch.empty(self.G.shape, dtype=torch.bfloat16, device=self.G.device)
                self.names = ["colsum", "to_bf16_G", "spmm_bwd", "gemm_dX", "gemm_dW", "gemm_xwT_bf16out", "spmm_fwd"]
            run(lambda: ops.colsum(self.G, out=self.dbias))
            run(lambda: ops.to_bf16(self.G, out=self.Gb))
            run(lambda: ops.aggregate_bwd(g, self.Gb, out=self.dH))
            run(lambda: ops.gemm(self.dH, self.W, out=self.dX))
            run(lambda: ops.gemm(self.dH, self.X, transA=True, out=self.dW))
            run(lambda: ops.linear_fwd_bf16(self.X, self.W, out=self.Hb))   # the product's epilogue stores bf16 H
            run(lambda: ops.aggregate_fwd(g, self.Hb, self.bias, out=self.out))
            if timed:
                self.ev.append(evs)
            return
        split = getattr(self, "split_gemm", False)
        # X and the upstream gradient G are both inputs of the layer step, so its two chains (G -> aggregate^T -> dH.W, dH^T.X and
        # X.W^T -> aggregate) may run in either order.  The backward chain goes first so that the three dense products are
        # consecutive: measured (scripts/exp_gemm_in_step.py), the f32 MFMA product takes 9.4 ms behind another product but 10.6 ms
        # behind an aggregation and 11.6 ms after 50 ms of idle -- the core clock has to come back up after a memory-bound
        # kernel -- so one such transition per step instead of two.
        if getattr(self, "gather_pitch", False):
            # vertices as the data set numbers them (no relabelling): the two matrices the aggregations GATHER from sit on the padded
            # row pitch (gnnx_gather_row_stride) -- H is the layer's own intermediate (the product writes it there), the caller's
            # upstream gradient G is copied there by the pass that sums its columns anyway (gnnx_colsum_copy_f32)
            run(lambda: ops.colsum_copy(self.G, self.Gpad, out=self.dbias))
            run(lambda: ops.aggregate_bwd(g, self.Gpad, out=self.dH))
        else:
            run(lambda: ops.colsum(self.G, out=self.dbias))
            run(lambda: ops.aggregate_bwd_sym(g, self.G, out=self.dH) if sym else ops.aggregate_bwd(g, self.G, out=self.dH))
        run(lambda: ops.gemm_split(self.dH, self.W, transB=False, out=self.dX) if split else ops.gemm(self.dHp, self.Wp, out=self.dXp))
        run(lambda: ops.gemm(self.dHp, self.Xp, transA=True, out=self.dWp))
        run(lambda: ops.gemm_split(self.X, self.W, transB=True, out=self.H) if split else ops.linear_fwd(self.Xp, self.Wp[:self.F], out=self.H))
        run(lambda: ops.aggregate_fwd_sym(g, self.H, self.bias, out=self.out) if sym else ops.aggregate_fwd(g, self.H, self.bias, out=self.out))
        if timed:
            self.ev.append(evs)

    def as_generated_control(self, pkg, n, e, abc, seed, chunk, steps=5, warmup=2):
        """A few steps of the SAME runner (same buffers, same kernels) on the graph in its as-generated vertex order: the number the
        default order is to be compared with.  Restores the runner afterwards."""
        ops = self.ops
        if abc is None:
            s_, d_ = pkg.synth.uniform_edges(seed, n, e)
            src, dst = torch.from_numpy(s_).to(self.X.device), torch.from_numpy(d_).to(self.X.device)
        else:
            src, dst = ops.rmat_edges(seed, n, e, *abc, device=self.X.device)
        g2 = ops.CsrGraph.from_coo(src, dst, n)
        del src, dst
        if chunk > 0:
            g2.make_plans(chunk, self.F)
        keep_g, keep_ev, keep_H = self.g, self.ev, self.H
        self.g = g2

        def run_steps():
            self.ev = []
            for _ in range(warmup):
                self.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(timed=True)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3, self.kernel_times()

        ms, kt = run_steps()   # (1) nothing but the labels changed: gathered rows on their own width
        out = {"order": "as-generated", "steps": steps, "ms_per_step": ms, "spmm_fwd_ms": kt.get("spmm_fwd"), "spmm_bwd_ms": kt.get("spmm_bwd"),
               "note": "same per-vertex bits as the default order (tests/test_gpu_parity.py::test_headline_config_whole_graph_vs_oracle)"}
        if ops.gather_row_stride(n, self.F) != self.F and self.Fp == self.F:
            # (2) what the library does for a caller who did not relabel the data set: the gathered matrices on the padded row pitch
            self.H = ops.empty_gathered(n, self.F, device=self.X.device)
            self.Gpad = ops.empty_gathered(n, self.F, device=self.X.device)
            self.gather_pitch = True
            ms2, kt2 = run_steps()
            self.gather_pitch = False
            del self.Gpad
            out["padded_gather_pitch"] = {"ms_per_step": ms2, "spmm_fwd_ms": kt2.get("spmm_fwd"), "spmm_bwd_ms": kt2.get("spmm_bwd"),
                                          "colsum_and_copy_ms": kt2.get("colsum"), "gathered_row_pitch_floats": ops.gather_row_stride(n, self.F),
                                          "what": "vertices as generated; H (the layer's intermediate) and a copy of the upstream gradient made "
                                                  "by the column-sum pass sit on the padded row pitch (gnnx_gather_row_stride)"}
        self.g, self.ev, self.H = keep_g, keep_ev, keep_H
        del g2
        ops._ws_cache.clear()
        torch.cuda.empty_cache()
        return out

    def kernel_times(self):
        out = {}
        for i, nm in enumerate(self.names):
            out[nm] = float(np.mean([s[i][0].elapsed_ms(s[i][1]) for s in self.ev])) if self.ev else None
        return out

    def roofline(self, ceilings=None):
        """The contract's roofline object for the dominant kernel pair: the forward aggregation = spmm_hub_kernel (hub rows) +
        spmm_stream_kernel (all other rows), timed together by one pair of HIP events on the launch stream.
          peak     = 8000 GB/s, the HBM3E spec (MI355X_MICROARCH.md) -- a fixed hardware figure, never a self-measured one;
          achieved = bytes that leave L2 per aggregation (PMC: 2 FETCH_SIZE + WRITE_SIZE of both kernels = `traffic`, taken FROM THE
                     COMMITTED rocprofv3 passes of this bench command -- `traffic_from_committed_profile` names the file, and the file
                     is refused when gnnx_spmm.hip has changed since it was made) / the HIP-event time of THIS run; without such a
                     profile: the algorithmic bytes (one feature row per non-zero) / that time;
          frac     = achieved / peak; `frac_algorithmic_vs_hbm_spec` = algorithmic bytes / time / 8000 beside it.
        Both readings are in the record because they differ for a power-law graph: 62 % of the edges of the bench graph point at
        168 MB of hub rows that live in the 256 MiB Infinity Cache, FETCH_SIZE counts Infinity-Cache hits, and the algorithmic count
        charges L2 hits as HBM reads -- so the algorithmic reading exceeds 1 and the traffic reading is a fraction of what the
        FABRIC behind L2 moved, not of the HBM pins alone.  `frac_traffic_vs_fabric_gather_ceiling` holds the same traffic against the
        whole-row gather ceiling measured in this run (bench_kernels/ceilings.hip), for context only."""
        ms = self.kernel_times()["spmm_fwd"]
        B = spmm_bytes(self.n, self.n, self.g.nnz, self.F, bias=True)
        if getattr(self, "bf16_features", False):
            B -= 2 * self.F * self.g.nnz  # 2-byte features in the gather
        eff = B / (ms * 1e-3) / 1e9
        tr = None
        if not getattr(self, "bf16_features", False) and not getattr(self, "sym", False):
            tr = profiled_traffic(self.workload, [(r"^spmm_hub_kernel<\d+, 0, ", True), (r"^spmm_hubpc_kernel<0, ", False),
                                                  (r"^spmm_(stream_)?kernel<\d+, \d+, \d+, 0, ", True)])
        stale = None
        if tr and tr[1].get("profiled_nnz") not in (None, self.g.nnz):
            tr, stale = None, "the committed profile is of another graph"
        if tr:
            import hashlib
            cur = hashlib.sha256(open(os.path.join(ROOT, "gnn.cpp_amd", "csrc", "gnnx_spmm.hip"), "rb").read()).hexdigest()
            was = tr[1].get("spmm_source_sha256_of_profiled_build")
            if was != cur:   # (a profile that does not say which source it was made from is as good as stale)
                tr, stale = None, "gnnx_spmm.hip has changed since the committed profile was made"
        achieved = (tr[0] if tr else B) / (ms * 1e-3) / 1e9
        evs = [st[self.names.index("spmm_fwd")] for st in self.ev]
        # HBM-side byte model from the graph itself: in-degree of a column of A = number of times its feature row is gathered
        model = hbm_side_model(self.g.rowptr_t[1:] - self.g.rowptr_t[:-1], self.g.nnz, self.n, self.F,
                               bytes_per_feature=2 if getattr(self, "bf16_features", False) else 4)
        hbm_model_GBps = model["hbm_bytes"] / (ms * 1e-3) / 1e9
        l2 = (tr[1].get("l2_hit_rate") if tr else None)
        r = {"bound": "hbm",   # the contract's enum (memory-bound, not MFMA-bound); WHICH part of the memory system binds: next two keys
             "bound_is": "fabric behind L2",
             "bound_detail": "the memory system behind the XCD L2s (Infinity Cache + HBM) on a whole-row gather -- NOT the HBM pins alone: "
                             "`frac` holds the bytes that left L2 (Infinity-Cache hits included) against the 8 TB/s HBM spec; "
                             "`fractions.hbm_model_vs_hbm_spec` is the HBM-pin estimate",
             "kernel": "forward aggregation = spmm_hub_kernel<VEC, 0, LAS, float> (+ spmm_hubpc_kernel<0, false, 16> for the longest rows) + "
                       "spmm_stream_kernel<G, VEC, 8, 0, 64, float>",
             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
             "frac_is": ("traffic behind L2 (2 FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits counted) / launch time / HBM spec" if tr else
                         "algorithmic bytes / launch time / HBM spec"),
             "achieved_is": ("traffic / avg_launch_ms" if tr else "algorithmic_bytes_per_launch / avg_launch_ms (no usable PMC profile of this "
                             "workload, graph and kernel source is committed" + (": " + stale if stale else "") + ")"),
             "peak_is": "HBM3E spec, 8000 GB/s (MI355X_MICROARCH.md)",
             "traffic": tr[0] if tr else None, "traffic_from_committed_profile": tr[1] if tr else None,
             "avg_launch_ms": ms, "median_launch_ms": float(np.median([a.elapsed_ms(b) for a, b in evs])) if evs else None,
             "algorithmic_bytes_per_launch": B, "bytes_per_edge": B / max(1, self.g.nnz), "effective_GBps": eff,
             "frac_algorithmic_vs_hbm_spec": eff / HBM_PEAK_GBS,
             # three readings of the same launch, each against the 8 TB/s HBM spec
             "fractions": {"traffic_behind_L2_vs_hbm_spec": (tr[0] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None,
                           "algorithmic_vs_hbm_spec": eff / HBM_PEAK_GBS,
                           "hbm_model_vs_hbm_spec": hbm_model_GBps / HBM_PEAK_GBS},
             "frac_hbm_model": hbm_model_GBps / HBM_PEAK_GBS, "hbm_model": dict(model, GBps=hbm_model_GBps),
             "l2_hit_rate": l2,
             # the cache-perfect lower bound (every feature row read once) beside the gather figure
             "cache_perfect_bytes": 4 * (self.n + 1) + 4 * self.g.nnz + 4 * self.n + 8 * self.F * self.n,
             "ceilings": ceilings, "vertex_order": self.vertex_order}
        if ceilings:
            r["frac_traffic_vs_fabric_gather_ceiling"] = achieved / ceilings["fabric_gather_GBps"] if tr else None
            r["traffic_over_hbm_copy_ceiling"] = achieved / ceilings["hbm_copy_GBps"] if tr else None
        return r

    def mfma_roofline(self):
        """The second bound of the step: the three dense products against the fp32 matrix peak (MI355X_MICROARCH.md: 157.3 TFLOP/s,
        exact-f32 MFMA).  Informational; the contract's `roofline` object is the dominant kernel, the forward SpMM."""
        kt = self.kernel_times()
        if not all(kt.get(k) for k in ("gemm_xwT", "gemm_dX", "gemm_dW")):
            return None
        fl = 2.0 * self.n * self.F * self.F
        out = {"bound": "mfma", "peak": 157.3, "unit": "TFLOP/s", "flops_per_launch": fl}
        for k in ("gemm_xwT", "gemm_dX", "gemm_dW"):
            tf = fl / (kt[k] * 1e-3) / 1e12
            out[k] = {"achieved": tf, "frac": tf / 157.3, "avg_launch_ms": kt[k]}
        return out


class TrainStep(SingleGpu):
    """Whole training step of an L-layer GCN (ops.GcnStack): forward with ReLU between layers, softmax cross-entropy
    against synthetic labels, backward, SGD.  Reuses SingleGpu's graph / plans / roofline bookkeeping."""

    def __init__(self, ops, capi, pkg, dev, n, e, F, abc, seed, chunk, layers, relabel="scramble"):
        super().__init__(ops, capi, pkg, dev, n, e, F, abc, seed, chunk, pad=False, relabel=relabel)
        for nm in ("H", "out", "dH", "dX", "G", "Hp", "outp", "dHp", "dXp", "Gp"):
            setattr(self, nm, None)  # free the single-layer buffers
        torch.cuda.empty_cache()
        self.net = ops.GcnStack(self.g, [F] * (layers + 1), seed=seed + 100, device=dev)
        self.X = self.net.pad_input(self.X)   # static features: stored once in the stack's streamed layout (a no-op for F % 128 == 0)
        self.target = (torch.arange(n, device=dev, dtype=torch.int64) * 7 + 3).remainder(F).to(torch.int32)
        self.names = ["train_step"]

    def step(self, timed=False):
        ops = self.ops
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if timed:
            a, b = self.capi.Event(), self.capi.Event()
            a.record(stream)
        logits = self.net.forward(self.X)
        _, dlog = ops.softmax_ce(logits, self.target, colsum_out=self.net.db[-1], grad_out=self.net.grad_buffer())   # last layer's bias gradient from the loss kernel
        self.net.backward(dlog, input_grad=False, have_last_bias_grad=True)           # the features are data: no dX of layer 1
        self.net.step(lr=1e-3)
        if timed:
            b.record(stream)
            self.ev.append([(a, b)])

    def roofline(self, ceilings=None):
        return {"bound": "hbm", "kernel": "n/a for --train-layers (see the default run)", "achieved": None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": None, "traffic": None, "ceilings": ceilings}


def cpp_api_bench(n, e, F, seed, both_orders=True, steps=5):
    """The same layer through the drop-in boundary itself: tests/cpp/bench_host_api drives graph::GCNConv on graph::Data (the
    reference's call sites graph.cpp:170-191, nn.cpp:205-211: `layer(data)`, `out->backward(G)`) over libgnncpp_host.so +
    libgnnx_hip.so -- started as a FRESH CHILD PROCESS after the timed region (never an exec of this process), its JSON lines parsed.
    hot path (BatchNorm / ReLU switched off: the bench's step) and the reference's full layer, in the data set's as-generated vertex
    order and with its labels scrambled once at load time (what bench.py's graph builder does for this row width)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "bench_host_api")
    if not os.path.exists(exe):
        return {"error": "tests/cpp/bench_host_api not built"}
    try:
        r = subprocess.run([exe, str(n), str(e), str(F), str(steps), "2", "2" if both_orders else "0", str(seed), "1"], capture_output=True,
                           text=True, timeout=900)
        rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    except Exception as ex:  # informational leg: never fail the bench because of it
        return {"error": str(ex)[:200]}
    if r.returncode != 0 or not rows:
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-200:]}"}
    out = {"program": "tests/cpp/bench_host_api (child process; graph::GCNConv::forward + tensor::backward through libgnncpp_host.so)",
           "steps": steps}
    for row in rows:
        key = ("scrambled_labels" if row["scrambled_labels"] else "as_generated")
        d = out.setdefault(key, {})
        if not row["hot_path_only"] and not row.get("fuse_bn_stats"):   # the layer with GCNConv::fuse_bn_stats = false (two-pass statistics): beside the default
            d["full_layer_ms_two_pass_stats"] = row["ms_per_step"]
            continue
        d["hot_path_ms" if row["hot_path_only"] else "full_layer_ms"] = row["ms_per_step"]
        d["first_call_s_hot" if row["hot_path_only"] else "first_call_s_full"] = row["first_call_s"]
    best = out.get("scrambled_labels") or out.get("as_generated")
    out["hot_path_ms"], out["full_layer_ms"] = best.get("hot_path_ms"), best.get("full_layer_ms")
    # the process's very first layer(data) + backward: uploads of X / G / the edge list from host valarrays, both CSRs, norm, plans
    # (later configurations find the features resident)
    out["first_call_s"] = rows[0]["first_call_s"]
    return out


def cpu_baseline(pkg, args, F, abc, seed, runner=None):
    """The CPU oracle (port of the reference arithmetic) timed on this box's host cores.  Two legs:
      * OpenMP over rows on every host thread (the reported `value`) on the bench's OWN workload -- the very graph (CSR of A and
        A^T, norm), features and upstream gradient of the GPU run, copied from the device (--cpu-sample-nodes 0, the default; about
        35 s at 10 M / 100 M / 256 on 128 threads), or on a smaller sample of the same generator when --cpu-sample-nodes says so;
      * one thread -- the reference itself is single-threaded -- on a sample 40x smaller."""
    import oracle  # checker / reported baseline only
    wl_n, wl_e = WORKLOADS[args.workload][:2]

    def timed_layer(rp, ci, rT, cT, norm, X, W, bias, G, threads):
        oracle.set_threads(threads)
        t0 = time.perf_counter()
        H = oracle.linear_fwd(X, W)
        oracle.aggregate_fwd(rp, ci, H, norm, bias)
        oracle.colsum(G)
        dH = oracle.aggregate_bwd(rT, cT, G, norm)
        oracle.linear_bwd(dH, X, W)
        return time.perf_counter() - t0

    def sample_leg(n, threads):
        e = int(round(wl_e * (n / wl_n)))
        if abc is None:
            src, dst = pkg.synth.uniform_edges(seed, n, e)
        else:
            src, dst = pkg.synth.rmat_edges(seed, n, e, *abc)
        rng = np.random.default_rng(seed)  # same distribution as the device inputs (U[-1,1)), cheaper to draw
        X = rng.random((n, F), dtype=np.float32) * 2 - 1
        W = pkg.synth.uniform_pm1(seed + 11, (F, F), scale=F ** -0.5)
        bias = np.zeros(F, dtype=np.float32)
        G = rng.random((n, F), dtype=np.float32) * 2 - 1
        rp, ci = oracle.coo_to_csr(src, dst, n)
        rT, cT = oracle.csr_transpose(rp, ci, n)
        s, norm = oracle.degree_norm(rp, ci, n)
        return n, e, len(ci), timed_layer(rp, ci, rT, cT, norm, X, W, bias, G, threads)

    cores = oracle.max_threads()
    full = args.cpu_sample_nodes <= 0 or args.cpu_sample_nodes >= wl_n
    if full and isinstance(runner, SingleGpu):
        # the GPU run's own inputs: nothing is regenerated, the host sees the graph in the bench's row order
        g = runner.g
        h = lambda t: t.detach().cpu().numpy()  # noqa: E731
        rp, ci = h(g.rowptr).astype(np.int64), h(g.colidx)
        rT, cT = h(g.rowptr_t).astype(np.int64), h(g.colidx_t)
        X, G = np.ascontiguousarray(h(runner.X)), np.ascontiguousarray(h(runner.G))
        dt = timed_layer(rp, ci, rT, cT, h(g.norm), X, np.ascontiguousarray(h(runner.W)), h(runner.bias), G, cores)
        n, e, nnz = runner.n, wl_e, int(g.nnz)
        what = f"{args.workload}: the bench's own graph and inputs (copied from the device), {n} nodes / nnz {nnz}"
        del rp, ci, rT, cT, X, G
    else:
        n, e, nnz, dt = sample_leg(min(max(args.cpu_sample_nodes, 1), wl_n) if not full else wl_n, cores)
        what = f"{args.workload} scaled to {n} nodes / {e} generated edges (nnz {nnz})" if n < wl_n else f"{args.workload}, regenerated on the host (nnz {nnz})"
    n1, e1, nnz1, dt1 = sample_leg(max(1, wl_n // 40), 1)
    oracle.set_threads(cores)
    return {"value": nnz / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{what}, {F} features, one layer fwd+bwd in {dt:.2f} s, OpenMP over rows on {cores} threads",
            "seconds": dt, "full_workload": bool(n >= wl_n),
            "single_thread": {"value": nnz1 / dt1, "unit": "edges/s", "cores": 1,
                              "sample": f"{n1} nodes / {e1} generated edges (nnz {nnz1}), {F} features, {dt1:.2f} s"}}


def cpu_reference(pkg):
    """The REAL reference (oracle/_ref/ref_driver, compiled from /root/reference in the build container) timed on this
    box on the largest BASELINE config it can run: Cora-sized N=2708 / E=10556, 1433->16 (its dense N x N path is
    O(N^2 F) and overflows `int` beyond N = 46340, so the bench workload itself is out of its reach).  Single-threaded,
    like the reference.  None when the binary did not travel."""
    import subprocess
    import tempfile
    import oracle  # locates oracle/_ref
    drv = oracle.ref_driver_path()
    if drv is None:
        return None
    n, e, fin, fout = 2708, 10556, 1433, 16
    src, dst = pkg.synth.uniform_edges(42, n, e)
    X = pkg.synth.uniform_pm1(171, (n, fin))
    W = pkg.synth.uniform_pm1(172, (fout, fin), scale=fin ** -0.5)
    b = np.zeros(fout, dtype=np.float32)
    G = pkg.synth.uniform_pm1(174, (n, fout))
    try:
        with tempfile.TemporaryDirectory() as td:
            cpath = os.path.join(td, "case.bin")
            with open(cpath, "wb") as f:
                np.array([n, e, fin, fout], dtype=np.int32).tofile(f)
                for a in (src, dst):
                    np.ascontiguousarray(a, dtype=np.int32).tofile(f)
                for a in (X, W, b, G):
                    np.ascontiguousarray(a, dtype=np.float32).tofile(f)
            r = subprocess.run([drv, cpath, td], capture_output=True, text=True, timeout=300)
            t = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as ex:  # the baseline is informational: never fail the bench because of it
        return {"error": str(ex)[:200]}
    secs = t["t_linear"] + t["t_aggregate"] + t["t_backward"]
    return {"value": t["nnz"] / secs, "unit": "edges/s", "cores": 1, "kind": "reference",
            "sample": f"Cora-sized N={n} E={e} (nnz {t['nnz']}), {fin}->{fout}: Linear {t['t_linear']:.3f} s + aggregate "
                      f"{t['t_aggregate']:.3f} s + backward {t['t_backward']:.3f} s (+ dense norm {t['t_norm']:.3f} s, not counted)",
            "seconds": secs}


if __name__ == "__main__":
    main()
