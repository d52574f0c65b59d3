"""Drop-in boundary on the GPU: oracle/ref_driver.cpp -- the driver that is compiled against the REAL reference
to produce the golden vectors -- is compiled UNCHANGED against gnn.cpp_amd/host/include (cyg::tensor, nn::Module,
graph::GCNConv ... over the C-ABI) into tests/cpp/dropin_driver, run on the golden inputs, and its dumps
are compared with what the reference produced for the same calls."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle
from tests.golden_util import CASES, load_case, same
from tests.helpers import ROOT
from tests.test_gpu_parity import assert_close

pytestmark = pytest.mark.gpu
DRIVER = os.path.join(ROOT, "tests", "cpp", "dropin_driver")


def run_driver(d, unfused=False, no_prologue_fusion=False, reference_quirks=False):
    assert os.path.exists(DRIVER), "build it with __graft_entry__.build()"
    with tempfile.TemporaryDirectory() as td:
        cpath = os.path.join(td, "case.bin")
        with open(cpath, "wb") as f:
            np.array([d["n"], len(d["src"]), d["fin"], d["fout"]], dtype=np.int32).tofile(f)
            for k in ("src", "dst"):
                np.ascontiguousarray(d[k], dtype=np.int32).tofile(f)
            for k in ("X", "W", "bias", "G"):
                np.ascontiguousarray(d[k], dtype=np.float32).tofile(f)
            if "w" in d:
                np.ascontiguousarray(d["w"], dtype=np.float32).tofile(f)
        env = dict(os.environ)
        if unfused:
            env["GNNCPP_UNFUSED"] = "1"  # op-by-op MatMul/Mul/Add instead of the fused aggregation op
        if no_prologue_fusion:
            env["GNNCPP_NO_PROLOGUE_FUSION"] = "1"  # BatchNorm and ReLU as their own kernels in front of the aggregation
        env.pop("GNNCPP_REFERENCE_QUIRKS", None)
        if reference_quirks:
            env["GNNCPP_REFERENCE_QUIRKS"] = "1"  # BatchNorm backward as the reference's traversal delivers it
        r = subprocess.run([DRIVER, cpath, td, "full"] + (["weighted"] if "w" in d else []), capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        n, fin, fout = d["n"], d["fin"], d["fout"]
        rd = lambda nm, dt: np.fromfile(os.path.join(td, nm), dtype=dt)  # noqa: E731
        extra = {}
        if "w" in d:
            extra = dict(w_deg=rd("w_deg.f32", np.float32), w_mm=rd("w_mm.f32", np.float32).reshape(n, fout),
                         w_fill_ei=rd("w_fill_ei.i32", np.int32).reshape(2, -1), w_fill_ea=rd("w_fill_ea.f32", np.float32),
                         w_strip_ei=rd("w_strip_ei.i32", np.int32).reshape(2, -1), w_strip_ea=rd("w_strip_ea.f32", np.float32))
        return dict(**extra, ei2=rd("ei2.i32", np.int32).reshape(2, -1), s=rd("s.f32", np.float32), norm=rd("norm.f32", np.float32),
                    H=rd("H.f32", np.float32).reshape(n, fout), agg=rd("agg.f32", np.float32).reshape(n, fout),
                    out=rd("out.f32", np.float32).reshape(n, fout), dX=rd("dX.f32", np.float32).reshape(n, fin),
                    dW=rd("dW.f32", np.float32).reshape(fout, fin), dbias=rd("dbias.f32", np.float32),
                    out_full=rd("out_full.f32", np.float32).reshape(n, fout), Hbn=rd("Hbn.f32", np.float32).reshape(n, fout),
                    Hrelu=rd("Hrelu.f32", np.float32).reshape(n, fout), loss=rd("loss.f32", np.float32),
                    full_dX=rd("full_dX.f32", np.float32).reshape(n, fin), full_dW=rd("full_dW.f32", np.float32).reshape(fout, fin),
                    full_dbias=rd("full_dbias.f32", np.float32), full_dgamma=rd("full_dgamma.f32", np.float32),
                    full_dbeta=rd("full_dbeta.f32", np.float32))


@pytest.mark.parametrize("name", ["karate_l1", "rmat1024", "cora_l2"])
def test_fused_and_op_by_op_paths_give_the_same_bits(name):
    d = load_case(name)
    a, b = run_driver(d, unfused=False), run_driver(d, unfused=True)
    c = run_driver(d, no_prologue_fusion=True)
    for k in ("ei2", "s", "norm", "H", "agg", "out", "dX", "dW", "dbias", "out_full", "Hbn", "Hrelu", "loss"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
        assert np.array_equal(a[k], c[k], equal_nan=True), k + " (BatchNorm+ReLU folded into the gather vs separate kernels)"


@pytest.mark.parametrize("name", CASES)
def test_reference_call_sites_run_on_the_hip_backend(name):
    d = load_case(name)
    got = run_driver(d)
    # graph layer: add_self_loops -> same [2, nnz] edge list as the reference
    assert np.array_equal(got["ei2"], d["ref_ei2"])
    # degree / norm block through the tensor ops (sum, +1, pow, mm, *=): bit-exact
    assert same(got["s"], d["ref_s"]) and same(got["norm"], d["ref_norm"])
    # Linear through tensor::mm on a transposed view -> MFMA GEMM
    assert_close(got["H"], d["ref_H"], "H")
    # aggregate_and_update + bias through MatMul(CSR) / Mul / Add: bit-exact given the H this run produced
    rp, ci = oracle.coo_to_csr(d["src"], d["dst"], d["n"])
    assert same(got["agg"], oracle.aggregate_fwd(rp, ci, got["H"], d["ref_norm"], None))
    assert same(got["out"], oracle.aggregate_fwd(rp, ci, got["H"], d["ref_norm"], d["bias"]))
    assert_close(got["out"], d["ref_out"], "out")
    # the whole layer as the reference runs it: layer(data) = transform -> BatchNorm -> ReLU -> aggregate -> + bias
    ref_bn, _, _ = oracle.bn_relu_fwd(got["H"], do_relu=False)   # oracle == reference on these (pinned), fed this run's H
    assert_close(got["Hbn"], ref_bn, "nn::BatchNorm forward")
    assert np.array_equal(got["Hrelu"], np.maximum(got["Hbn"], 0)), "nn::ReLU forward"
    assert_close(got["Hrelu"], d["ref_Hrelu"], "BatchNorm+ReLU vs reference")
    assert_close(got["out_full"], d["ref_out_full"], "GCNConv::forward (full layer)")
    # nn::cross_entropy_loss on the layer output (targets (7i+3) mod F_out); NaN where the reference's own value is NaN
    if np.isnan(d["ref_loss"][0]):
        assert np.isnan(got["loss"][0])
    else:
        assert abs(float(got["loss"][0]) - float(d["ref_loss"][0])) <= 1e-5 * max(1.0, abs(float(d["ref_loss"][0])))
    # autograd: out->backward(G) through Add -> Mul -> MatMul(CSR) -> MatMul -> Transpose
    G64, X64 = d["G"].astype(np.float64), d["X"].astype(np.float64)
    rT, cT = oracle.csr_transpose(rp, ci, d["n"])
    dH64 = oracle.aggregate_bwd(rT, cT, d["G"], d["ref_norm"]).astype(np.float64)
    assert_close(got["dbias"], d["ref_dbias"], "dbias", absum=np.abs(G64).sum(0), exact=G64.sum(0))
    assert_close(got["dW"], d["ref_dW"], "dW", absum=np.abs(dH64).T @ np.abs(X64), exact=dH64.T @ X64)
    if "ref_dX" in d:
        assert_close(got["dX"], d["ref_dX"], "dX")
    else:
        assert_close(got["dX"][d["ref_dX_rows"]], d["ref_dX_sample"], "dX rows")


def test_operator_classes_like_the_reference_operation_tests():
    """tests/cpp/test_host_ops_gpu.cpp: Add / Mul / Div forward + backward closed forms of the reference's
    tests/operation.test.cpp:32-89 (exact), the broadcast shapes of the path, in-place operators, dense row sums."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_ops_gpu")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("name", ["weighted6", "weighted_rmat64"])
def test_edge_attr_forms_of_the_graph_functions(name):
    """edge_to_adj_mat(ei, edge_attr, N) -> sum(-1) / mm(x), add_self_loops(ei, edge_attr, fill, N): the same driver calls,
    on the HIP backend, against what the reference returned (last duplicate wins, diagonal fill, int() filter)."""
    d = load_case(name)
    got = run_driver(d)
    assert same(got["w_deg"], d["ref_w_deg"])
    rp, ci, va = oracle.coo_to_csr_weighted(d["src"], d["dst"], d["w"], d["n"])
    assert same(got["w_mm"], oracle.spmm_vals(rp, ci, va, got["H"]))   # bit-exact given the H this run's GEMM produced
    assert_close(got["w_mm"], d["ref_w_mm"], "weighted adj->mm")
    for key in ("w_fill", "w_strip"):
        assert np.array_equal(got[key + "_ei"], d["ref_" + key + "_ei"]), key
        assert same(got[key + "_ea"], d["ref_" + key + "_ea"]), key


def test_gather_pitch_through_the_cpp_call_sites_same_bits():
    """graph::GCNConv on an R-MAT graph in its as-generated vertex order, large enough for the padded gather pitch (70 000 x 256): the
    layer notices the hub ids and keeps the transform's output and the copy of the upstream gradient on the pitch -- hot path and the
    full BatchNorm + ReLU layer give the same output, input gradient and weight gradient, bit for bit, as with the pitch switched
    off (GNNCPP_NO_GATHER_PITCH=1)."""
    import json
    exe = os.path.join(ROOT, "tests", "cpp", "bench_host_api")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"

    def run(env_extra):
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([exe, "70000", "1400000", "256", "1", "2", "0", "5"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
        rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(rows) == 2
        return ({row["hot_path_only"]: (row["probe_out"], row["probe_dx"], row["probe_dw"]) for row in rows},
                {row["gathered_row_pitch"] for row in rows})

    (on, pitch_on), (off, pitch_off) = run({}), run({"GNNCPP_NO_GATHER_PITCH": "1"})
    assert pitch_on == {320} and pitch_off == {256}, (pitch_on, pitch_off)
    assert on == off, (on, off)
    assert len({v for v in on.values()}) == 2   # (hot path and full layer are different computations: the probes are not degenerate)


def test_sharded_layer_through_the_cpp_api():
    """tests/cpp/test_host_sharded_gpu.cpp: graph::Partition + GCNConv::shard() with P = 2, 4, 8 ranks as threads over the
    in-process communicator (the C-ABI's gnnx_partition_deal / gnnx_halo_plan_* / gnnx_halo_exchange_rows_f32 underneath)
    against the unsharded layer -- hot path and the full BatchNorm + ReLU layer with cross-shard statistics -- and the
    content-keyed graph cache."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_host_sharded_gpu")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    # GNNCPP_TEST_DIAGNOSE: the stage-by-stage comparison (s local, s halo, norm, H local, H halo, out, G halo, dY, dX against host
    # references in the reference's summation order) also runs on a PASSING step, so the diagnosis a failure would print is itself
    # under test: every rank of every run must report that all traced stages match.  On a failure the binary names, per rank, the
    # first stage that differs -- in the same run, from device snapshots taken on the rank's stream.
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, GNNCPP_TEST_DIAGNOSE="1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("SHARDED_HOST_OK"), r.stdout[-6000:] + r.stderr[-3000:]
    diag = [ln for ln in r.stdout.splitlines() if ln.startswith("DIAG ")]
    # hot path and full layer at world 2 and 4, the 128-wide layer (send rows from the product kernel's epilogue) at 2 and 4, the tiny
    # graph at world 8
    assert len(diag) == 2 * (2 + 4) + (2 + 4) + 8, diag
    assert all("every traced stage matches" in ln for ln in diag), diag


@pytest.mark.parametrize("name", ["karate_l1", "rmat64", "rmat1024", "cora_l2"])
def test_through_layer_gradients_match_the_reference_in_quirk_mode(name):
    """layer(data)->backward(G) through transform <- BatchNorm <- ReLU <- aggregation with GNNCPP_REFERENCE_QUIRKS=1: the unchanged
    driver on the HIP backend reproduces the gradients the REFERENCE itself computes (ref_full_*, where every arrival at an op
    after the first is dropped, operation.h:80-88); without the switch the backend returns the mathematical gradient, which
    differs.  ReLU decisions on pre-activations within rounding of zero may flip, so rows are compared through a norm-wise
    bound on the gradients that sum over all nodes and element-wise on dX away from such units."""
    d = load_case(name)
    got = run_driver(d, reference_quirks=True)
    plain = run_driver(d)
    G64 = d["G"].astype(np.float64)
    assert_close(got["full_dbias"], d["ref_full_dbias"], "dbias", absum=np.abs(G64).sum(0), exact=G64.sum(0))
    for k in ("full_dW", "full_dgamma", "full_dbeta"):
        ref = d["ref_" + k].reshape(got[k].shape)
        assert np.abs(got[k] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), k
    if "ref_full_dX" in d:
        ref = d["ref_full_dX"]
        bad = np.abs(got["full_dX"] - ref) > 1e-4 * np.maximum(1.0, np.abs(ref))
        assert bad.mean() <= 1e-3, f"{bad.sum()} of {bad.size} elements of dX differ"
        assert np.abs(plain["full_dX"] - ref).max() > 1e-2, "without the switch the gradient is the mathematical one"


def _run_bench(args, timeout=600):
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line"
    return json.loads(lines[0])


def test_bench_contract_single_gpu_and_sharded_rehearsal():
    """bench.py's JSON line (the driver's contract): metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling /
    dtype / data / config.workload, a roofline object for the dominant kernel and the cpu_baseline object at N = 1; and the
    N > 1 code path (partition, halo plan, overlap schedule, all-reduce) rehearsed on one rank."""
    d = _run_bench(["--workload", "tiny", "--steps", "3", "--warmup", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "edges/s" and d["dtype"] == "f32"
    assert d["config"]["workload"] == "tiny" and d["vs_baseline"] is None and d["data"] == "synthetic"
    ro = d["roofline"]
    # the peak is a fixed hardware figure (the HBM spec); the ceilings measured in the same run (bench_kernels/ceilings.hip) ride along
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-9
    # a reader of the roofline object alone can tell fabric-bound from HBM-bound: what binds is named, `frac` says what it is a
    # fraction of, and the three readings of the launch (traffic behind L2, algorithmic bytes, HBM-side model) sit side by side
    assert ro["bound_is"] == "fabric behind L2" and "frac_is" in ro and set(ro["fractions"]) == {
        "traffic_behind_L2_vs_hbm_spec", "algorithmic_vs_hbm_spec", "hbm_model_vs_hbm_spec"}
    assert 0 < ro["frac_hbm_model"] <= ro["frac_algorithmic_vs_hbm_spec"] + 1e-12 and ro["hbm_model"]["hbm_bytes"] <= ro["algorithmic_bytes_per_launch"]
    assert abs(ro["fractions"]["hbm_model_vs_hbm_spec"] - ro["frac_hbm_model"]) < 1e-12 and "l2_hit_rate" in ro
    assert ro["peak"] == 8000.0 and 3000.0 < ro["ceilings"]["fabric_gather_GBps"] < 12000.0
    assert abs(ro["frac_algorithmic_vs_hbm_spec"] - ro["effective_GBps"] / 8000.0) < 1e-9 and "traffic_from_committed_profile" in ro
    assert 2500.0 < ro["ceilings"]["hbm_copy_GBps"] < 8000.0 and ro["effective_GBps"] > 0 and ro["avg_launch_ms"] > 0
    nc = _run_bench(["--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-ceilings", "--no-cpp-api"])
    assert nc["roofline"]["peak"] == 8000.0 and nc["roofline"]["ceilings"] is None and nc["cpp_api"] is None
    cp = d["cpp_api"]   # the C++ call-site leg ran as a child process on the same workload
    assert cp and "error" not in cp and cp["hot_path_ms"] > 0 and cp["full_layer_ms"] > cp["hot_path_ms"] * 0.9, cp
    assert abs(d["value"] - d["config"]["nnz"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["full_workload"] and "the bench's own graph" in cb["sample"], "the CPU baseline runs the bench's own workload by default"
    for sched in ("overlap", "training", "sequential"):
        s = _run_bench(["--workload", "tiny", "--steps", "2", "--warmup", "1", "--force-sharded", "--no-cpu-baseline", "--schedule", sched])
        assert s["n_gpus"] == 1 and "halo all-to-all-v" in s["config"]["parallelism"] and s["roofline"]["schedule"] == sched
        assert s["config"]["nnz"] == d["config"]["nnz"], "the sharded path must see the same graph"
    t = _run_bench(["--workload", "tiny", "--steps", "2", "--warmup", "1", "--force-sharded", "--no-cpu-baseline", "--train-layers", "2"])
    assert t["n_gpus"] == 1 and "training step" in t["config"]["step"] and t["roofline"]["frac"] is None
    assert abs(t["value"] - 2 * t["config"]["nnz"] / (t["ms_per_step"] * 1e-3)) <= 1e-6 * t["value"]


def test_multi_process_job_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver starts it for N > 1 -- plain python, which spawns torch.distributed.run,
    one PROCESS per rank, rendezvous on 127.0.0.1, partition + halo plan + overlap schedule + parameter all-reduce, max-over-ranks
    timing, ONE JSON line from rank 0 -- with the one thing a one-GPU box cannot provide swapped out: the ranks share GPU 0 and
    the collectives are staged through host memory over gloo (--dist-backend gloo --all-ranks-on-device0).  Checks the line
    and that the sharded job saw the same graph as the single-GPU one; 3 ranks too (odd world, uneven n/P)."""
    single = _run_bench(["--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    for world, port in ((2, "29651"), (3, "29652")):
        env_port = dict(os.environ, MASTER_PORT=port)
        import json
        import sys
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--workload", "tiny", "--steps", "2",
                            "--warmup", "1", "--dist-backend", "gloo", "--all-ranks-on-device0"], capture_output=True, text=True,
                           timeout=600, cwd=ROOT, env=env_port)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["nnz"] == single["config"]["nnz"]
        assert "REHEARSAL" in d["config"]["parallelism"] and d["value"] > 0 and d["roofline"]["frac"] > 0


def test_bench_padded_feature_rows_same_bits():
    """bench.py stores the streamed F = 100 rows (X, dH, dX) with a 128-float stride (products-shaped config) so the backward
    products run on the 128-wide LDS-DMA kernels: every output of the step equals, bit for bit (dW: to split-K rounding), the one computed on rows stored at their own width."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pkg = bench.load_package()
    from importlib import import_module
    ops, capi = import_module("gnncpp_amd.ops"), import_module("gnncpp_amd.capi")
    dev = torch.device("cuda:0")
    outs = []
    for pad in (True, False):
        r = bench.SingleGpu(ops, capi, pkg, dev, 120_000, 1_000_000, 100, (0.57, 0.19, 0.19), 5, 4096, pad=pad)
        r.step()
        torch.cuda.synchronize()
        outs.append({k: getattr(r, k).clone() for k in ("H", "out", "dH", "dX", "dW", "dbias")})
        assert r.Fp == (128 if pad else 100)
        if pad:
            for nm in ("dHp", "dXp"):
                assert not getattr(r, nm)[:, 100:].any(), nm + ": pad columns must stay zero"
    for k in outs[0]:
        if k == "dW":   # split-K over the 120 000 rows: the two kernels cut K differently (tolerance as in test_gpu_parity's dW cases)
            torch.testing.assert_close(outs[0][k], outs[1][k], rtol=1e-5, atol=1e-5 * float(outs[1][k].abs().max()))
        else:
            assert torch.equal(outs[0][k], outs[1][k]), k
