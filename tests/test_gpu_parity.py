"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI
(include/gnnx.h via gnn.cpp_amd/capi.py), against

  1. the golden vectors produced by the REAL reference (tests/golden/*.npz), and
  2. the CPU oracle (oracle/gcn_oracle.c, itself pinned bit-exact to the reference) on seeded inputs.

Bars:
  * integer / index work (CSR build): bit-exact;
  * aggregation (SpMM fwd/bwd), s, norm: bit-exact (numeric ==), with or without a load-balancing plan -- every
    kernel adds neighbour rows into ONE accumulator per output element in the reference's own order with separately
    rounded fp32 ops;
  * feature-dimension GEMMs (X.W^T, dH.W; K = F <= a few thousand): |gpu - ref| <= 1e-5 * max(1, |ref|)
    (BASELINE.json north_star tolerance; the MFMA is an fma chain, the reference rounds product and sum
    separately);
  * NODE-dimension reductions (dW = dH^T.X and dbias sum over N nodes): the
    reference's own sequential fp32 sum of n cancelling terms is off from exact arithmetic by
    ~eps*sqrt(n)*|partial sums|, i.e. by MORE than 1e-5*|result| once n reaches a few thousand, so two
    correct summation orders cannot agree to 1e-5 of a cancelled result.  There the bound is the
    condition-aware form of the same tolerance, |gpu - ref| <= 1e-5 * max(1, |ref|, sum_k |term_k|), and in
    addition the GPU must be at least as close to float64 arithmetic as the reference is (x2 slack).
"""
import importlib

import numpy as np
import pytest

import oracle
from tests.golden_util import CASES, load_case, same
from tests.helpers import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-5  # north_star: "within 1e-5 relative fp32"


@pytest.fixture(scope="module")
def env():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no HIP device visible: the gpu-marked tests must run on an MI355X")
    ops = importlib.import_module("gnncpp_amd.ops")
    capi = importlib.import_module("gnncpp_amd.capi")
    assert capi.device_count() >= 1
    return dict(torch=torch, ops=ops, capi=capi, dev=torch.device("cuda:0"))


def dev(env, a):
    return env["torch"].from_numpy(np.ascontiguousarray(a)).to(env["dev"])


def host(t):
    return t.detach().cpu().numpy()


def assert_close(got, ref, what="", absum=None, exact=None):
    """|got - ref| <= RTOL * max(1, |ref|[, absum]).  absum = sum_k |term_k| per output element, passed only for
    node-dimension / hub reductions (module docstring).  exact = float64 result: then also require the GPU to
    be no further from it than twice the reference's own error (plus 1e-6 of the result scale)."""
    got64, ref64 = got.astype(np.float64), ref.astype(np.float64)
    err = np.abs(got64 - ref64)
    scale = np.maximum(1.0, np.abs(ref64))
    if absum is not None:
        scale = np.maximum(scale, absum)
    worst = float((err / (RTOL * scale)).max()) if err.size else 0.0
    assert worst <= 1.0, f"{what}: max err/bound = {worst:.3f}"
    if exact is not None and err.size:
        e_gpu, e_ref = np.abs(got64 - exact).max(), np.abs(ref64 - exact).max()
        assert e_gpu <= 2.0 * e_ref + 1e-6 * max(1.0, np.abs(exact).max()), \
            f"{what}: GPU error vs float64 {e_gpu:.3e} exceeds reference's own {e_ref:.3e}"


# ------------------------------------------------------------------ golden vectors (real reference)
@pytest.fixture(scope="module", params=CASES)
def gcase(request, env):
    d = load_case(request.param)
    ops = env["ops"]
    g = ops.CsrGraph.from_coo(dev(env, d["src"]), dev(env, d["dst"]), d["n"])
    d["g"] = g
    return d


def test_golden_csr_build_bit_exact(gcase, env):
    g, ei2 = gcase["g"], gcase["ref_ei2"]
    rowptr = host(g.rowptr)
    assert rowptr[0] == 0 and rowptr[-1] == ei2.shape[1] == g.nnz
    assert np.array_equal(np.repeat(np.arange(gcase["n"], dtype=np.int32), np.diff(rowptr)), ei2[0])
    assert np.array_equal(host(g.colidx), ei2[1])
    # transposed CSR == CSR of the swapped edge list (oracle)
    rp, ci = oracle.coo_to_csr(gcase["dst"], gcase["src"], gcase["n"])
    assert np.array_equal(host(g.rowptr_t), rp.astype(np.int32)) and np.array_equal(host(g.colidx_t), ci)


def test_golden_degree_norm_bit_exact(gcase):
    assert same(host(gcase["g"].s), gcase["ref_s"])
    assert same(host(gcase["g"].norm), gcase["ref_norm"])


def test_golden_linear_forward(gcase, env):
    H = env["ops"].linear_fwd(dev(env, gcase["X"]), dev(env, gcase["W"]))
    assert_close(host(H), gcase["ref_H"], "H = X.W^T")


def test_golden_aggregate_forward_bit_exact(gcase, env):
    """Fed the reference's own H, the SpMM + norm + bias epilogue reproduces the reference bit for bit."""
    ops, g = env["ops"], gcase["g"]
    H = dev(env, gcase["ref_H"])
    agg = ops.aggregate_fwd(g, H, None)
    out = ops.aggregate_fwd(g, H, dev(env, gcase["bias"]))
    assert same(host(agg), gcase["ref_agg"])
    assert same(host(out), gcase["ref_out"])
    # unfused path (op by op, as the tensor API does it) gives the same bits
    plain = ops.spmm(g.rowptr, g.colidx, H)
    un = ops.bias_add(ops.rowscale(plain, g.norm), dev(env, gcase["bias"]))
    assert same(host(un), gcase["ref_out"])


def test_golden_layer_forward_chained(gcase, env):
    ops = env["ops"]
    _, out = ops.gcn_layer_fwd(gcase["g"], dev(env, gcase["X"]), dev(env, gcase["W"]), dev(env, gcase["bias"]))
    assert_close(host(out), gcase["ref_out"], "layer forward")


def test_golden_backward(gcase, env):
    ops = env["ops"]
    b = ops.gcn_layer_bwd(gcase["g"], dev(env, gcase["X"]), dev(env, gcase["W"]), dev(env, gcase["G"]))
    G64, X64 = gcase["G"].astype(np.float64), gcase["X"].astype(np.float64)
    dH64 = host(b["dH"]).astype(np.float64)  # bit-exact vs the oracle, asserted below
    assert_close(host(b["dbias"]), gcase["ref_dbias"], "dbias", absum=np.abs(G64).sum(0), exact=G64.sum(0))
    assert_close(host(b["dW"]), gcase["ref_dW"], "dW", absum=np.abs(dH64).T @ np.abs(X64), exact=dH64.T @ X64)
    if "ref_dX" in gcase:
        assert_close(host(b["dX"]), gcase["ref_dX"], "dX")
    else:
        assert_close(host(b["dX"])[gcase["ref_dX_rows"]], gcase["ref_dX_sample"], "dX rows")
    # the aggregation's backward is bit-exact against the oracle (pinned to the reference on dX/dW above)
    rT, cT = oracle.csr_transpose(*oracle.coo_to_csr(gcase["src"], gcase["dst"], gcase["n"]), gcase["n"])
    dH_ref = oracle.aggregate_bwd(rT, cT, gcase["G"], gcase["ref_norm"])
    assert same(host(b["dH"]), dH_ref)


# ------------------------------------------------------------------ seeded cases vs the oracle
def make_graph(env, n, e, seed, kind="rmat"):
    if kind == "rmat":
        src, dst = synth.rmat_edges(seed, n, e)
    else:
        src, dst = synth.uniform_edges(seed, n, e)
    rp, ci = oracle.coo_to_csr(src, dst, n)
    g = env["ops"].CsrGraph.from_coo(dev(env, src), dev(env, dst), n)
    return src, dst, rp, ci, g


@pytest.mark.parametrize("n,e,F", [(20000, 200000, 256), (20000, 200000, 128), (20000, 200000, 100),
                                   (5000, 60000, 64), (5000, 60000, 16), (3000, 20000, 7), (3000, 20000, 1),
                                   (2000, 30000, 512), (2000, 30000, 300), (1000, 5000, 33)])
def test_spmm_forward_backward_bit_exact_vs_oracle(env, n, e, F):
    ops = env["ops"]
    src, dst, rp, ci, g = make_graph(env, n, e, seed=7 + F)
    assert np.array_equal(host(g.rowptr), rp.astype(np.int32)) and np.array_equal(host(g.colidx), ci)
    s, norm = oracle.degree_norm(rp, ci, n)
    assert same(host(g.s), s) and same(host(g.norm), norm)
    H = synth.uniform_pm1(11, (n, F))
    bias = synth.uniform_pm1(12, (F,))
    out = ops.aggregate_fwd(g, dev(env, H), dev(env, bias))
    assert same(host(out), oracle.aggregate_fwd(rp, ci, H, norm, bias))
    G = synth.uniform_pm1(13, (n, F))
    rT, cT = oracle.csr_transpose(rp, ci, n)
    dH = ops.aggregate_bwd(g, dev(env, G))
    assert same(host(dH), oracle.aggregate_bwd(rT, cT, G, norm))


def test_spmm_accumulate_and_strided(env):
    """beta = 1 (`_grad +=`, tensor.h:268-271) and ld > F (feature blocks inside wider buffers)."""
    ops, torch = env["ops"], env["torch"]
    n, F, LD = 4000, 96, 160
    src, dst, rp, ci, g = make_graph(env, n, 40000, seed=3)
    H = synth.uniform_pm1(21, (n, F))
    Y0 = synth.uniform_pm1(22, (n, F))
    ref = Y0 + oracle.aggregate_fwd(rp, ci, H, g_norm := oracle.degree_norm(rp, ci, n)[1], None)
    Xw = torch.zeros((n, LD), dtype=torch.float32, device=env["dev"])
    Xw[:, 32:32 + F] = dev(env, H)
    Yw = torch.full((n, LD), 7.0, dtype=torch.float32, device=env["dev"])
    Yw[:, 16:16 + F] = dev(env, Y0)
    ops.spmm(g.rowptr, g.colidx, Xw[:, 32:32 + F], out=Yw[:, 16:16 + F], rowscale=g.norm, beta=1.0)
    got = host(Yw)
    assert same(got[:, 16:16 + F], ref.astype(np.float32))
    assert np.all(got[:, :16] == 7.0) and np.all(got[:, 16 + F:] == 7.0)  # nothing outside the block is touched
    del g_norm


def test_spmm_split_rows_plan(env):
    """Power-law rows handed to the hub kernel: the plan reports them, and the planned result is the oracle's, bit for bit,
    forward and backward, run after run."""
    ops = env["ops"]
    n, F = 30000, 128
    src, dst, rp, ci, g = make_graph(env, n, 600000, seed=5)
    deg = np.diff(rp)
    assert deg.max() > 2000, "test graph should have hubs"
    s, norm = oracle.degree_norm(rp, ci, n)
    H = synth.uniform_pm1(31, (n, F))
    ref = oracle.aggregate_fwd(rp, ci, H, norm, None)
    plan, plan_t = g.make_plans(chunk=256, max_feat=F)
    assert plan.n_split_rows == int((deg > 256).sum()) and plan.n_hub_nnz == int(deg[deg > 256].sum())
    out1 = host(ops.aggregate_fwd(g, dev(env, H)))
    out2 = host(ops.aggregate_fwd(g, dev(env, H)))
    assert np.array_equal(out1, out2), "run-to-run deterministic"
    assert same(out1, ref)
    G = synth.uniform_pm1(32, (n, F))
    rT, cT = oracle.csr_transpose(rp, ci, n)
    dref = oracle.aggregate_bwd(rT, cT, G, norm)
    assert same(host(ops.aggregate_bwd(g, dev(env, G))), dref)
    assert same(host(ops.aggregate_fwd(g, dev(env, H), use_plan=False)), ref)


@pytest.mark.parametrize("n,e,F,chunk", [(30000, 600000, 256, 256), (30000, 600000, 128, 64), (20000, 400000, 100, 64),
                                         (20000, 400000, 36, 64), (8000, 200000, 33, 64), (8000, 200000, 7, 64),
                                         (8000, 200000, 320, 128), (6000, 150000, 1, 64)])
def test_plan_hub_rows_keep_the_reference_order(env, n, e, F, chunk):
    """The plan's split rows are summed by spmm_hub_kernel: one accumulator per feature, top column first, whatever the degree
    (functional.h:433-439) -- so a planned aggregation is BIT-EXACT on every row, forward (rowscale + bias), backward (per-entry
    norm), accumulate (beta = 1) and ReLU epilogue, vector and scalar lanes, full and ragged 64-feature slabs, rows of every length
    from chunk + 1 up (sub-chunk tails, rows shorter than the ring's look-ahead)."""
    ops, torch = env["ops"], env["torch"]
    src, dst, rp, ci, g = make_graph(env, n, e, seed=70 + F)
    deg = np.diff(rp)
    assert (deg > chunk).sum() >= 8 and deg.max() > 4 * chunk, "test graph should have hubs"
    s, norm = oracle.degree_norm(rp, ci, n)
    H = synth.uniform_pm1(41, (n, F))
    bias = synth.uniform_pm1(42, (F,))
    plan, plan_t = g.make_plans(chunk=chunk, max_feat=F)
    assert plan.n_split_rows == int((deg > chunk).sum())
    ref = oracle.aggregate_fwd(rp, ci, H, norm, bias)
    out = ops.aggregate_fwd(g, dev(env, H), dev(env, bias))
    assert same(host(out), ref), "forward"
    assert same(host(ops.aggregate_fwd(g, dev(env, H), dev(env, bias), use_plan=False)), ref)
    G = synth.uniform_pm1(43, (n, F))
    rT, cT = oracle.csr_transpose(rp, ci, n)
    dref = oracle.aggregate_bwd(rT, cT, G, norm)
    assert same(host(ops.aggregate_bwd(g, dev(env, G))), dref), "backward"
    # beta = 1 (the reference's `_grad +=`) and the ReLU epilogue: planned == unplanned, bit for bit
    acc0 = synth.uniform_pm1(44, (n, F))
    a1 = ops.aggregate_bwd(g, dev(env, G), out=dev(env, acc0), beta=1.0)
    a0 = ops.aggregate_bwd(g, dev(env, G), out=dev(env, acc0), beta=1.0, use_plan=False)
    assert torch.equal(a1, a0), "beta = 1"
    r1 = ops.aggregate_fwd(g, dev(env, H), dev(env, bias), relu_out=True)
    r0 = ops.aggregate_fwd(g, dev(env, H), dev(env, bias), relu_out=True, use_plan=False)
    assert torch.equal(r1, r0) and same(host(r1), np.where(ref > 0, ref, np.float32(0))), "ReLU epilogue"
    # Mode SYM (per-column scale) and per-entry values + column scale: planned == unplanned
    y1 = ops.aggregate_fwd_sym(g, dev(env, H), dev(env, bias))
    y0 = ops.aggregate_fwd_sym(g, dev(env, H), dev(env, bias), use_plan=False)
    assert torch.equal(y1, y0), "colscale"
    vals = dev(env, synth.uniform_pm1(45, (len(ci),)))
    z1 = ops.spmm(g.rowptr, g.colidx, dev(env, H), vals=vals, colscale=g.s, rowscale=g.norm, plan=g.plan)
    z0 = ops.spmm(g.rowptr, g.colidx, dev(env, H), vals=vals, colscale=g.s, rowscale=g.norm)
    assert torch.equal(z1, z0), "vals + colscale"
    # strided X / Y (ld > F) and a run-to-run check
    if F % 4 == 0:
        Hp = torch.zeros((n, F + 12), dtype=torch.float32, device=env["dev"])
        Hp[:, :F] = dev(env, H)
        Yp = torch.zeros((n, F + 4), dtype=torch.float32, device=env["dev"])
        ops.aggregate_fwd(g, Hp[:, :F], dev(env, bias), out=Yp[:, :F])
        assert torch.equal(Yp[:, :F], out) and float(Yp[:, F:].abs().max()) == 0.0, "strided rows"
    assert torch.equal(ops.aggregate_fwd(g, dev(env, H), dev(env, bias)), out)


@pytest.mark.parametrize("n,F", [(70001, 256), (5000, 128), (3001, 100), (2000, 33), (4096, 1024)])
def test_colsum_copy_and_gather_pitch(env, n, F):
    """gnnx_colsum_copy_f32: the column sums are gnnx_colsum_f32's bits, the copy holds G's rows on the other pitch (pad columns
    untouched); gnnx_gather_row_stride pads only large matrices of 512-byte-multiple rows; an aggregation gathering from the
    padded copy gives the bits of the one gathering from G."""
    ops, torch = env["ops"], env["torch"]
    G = ops.uniform_pm1(77, (n, F), device=env["dev"])
    ld = F + 64
    buf = torch.full((n, ld), -7.0, dtype=torch.float32, device=env["dev"])
    s1 = ops.colsum_copy(G, buf[:, :F])
    assert torch.equal(s1, ops.colsum(G)) and torch.equal(buf[:, :F], G) and bool((buf[:, F:] == -7.0).all())
    assert ops.gather_row_stride(10_000_000, 256) == 320 and ops.gather_row_stride(10_000_000, 128) == 192
    assert ops.gather_row_stride(10_000_000, 100) == 100 and ops.gather_row_stride(1000, 256) == 256
    src, dst = synth.rmat_edges(5, n, 8 * n)
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), n)
    assert torch.equal(ops.aggregate_bwd(g, buf[:, :F]), ops.aggregate_bwd(g, G))
    assert torch.equal(ops.aggregate_fwd(g, buf[:, :F]), ops.aggregate_fwd(g, G))


def test_gcn_stack_on_the_gather_pitch(env):
    """ops.GcnStack on an as-generated R-MAT graph large enough for the padded gather pitch: the stack notices the hub ids
    (gnnx_spmm_plan_hub_ids_structured) and keeps H_l and the gradients G_l -- all written by its own kernels -- on the pitch; one
    training step (forward, softmax-CE into net.grad_buffer(), backward) gives the bits of the same stack on contiguous rows."""
    ops, torch = env["ops"], env["torch"]
    n, e, dims = 70_000, 1_400_000, [256, 256, 128]
    src, dst = ops.rmat_edges(31, n, e, device=env["dev"])
    g = ops.CsrGraph.from_coo(src, dst, n)
    g.make_plans(64, 256)
    X = ops.uniform_pm1(32, (n, dims[0]), device=env["dev"]) * 1e-4   # (hub rows sum thousands of terms: keep the max-free softmax finite)
    t = ((7 * torch.arange(n, device=env["dev"]) + 3) % dims[-1]).to(torch.int32)
    nets = [ops.GcnStack(g, dims, seed=40), ops.GcnStack(g, dims, seed=40)]
    assert nets[0].gather_pitch and ops.gather_row_stride(n, 256) == 320
    nets[1].gather_pitch = False
    res = []
    for net in nets:
        logits = net.forward(X)
        loss, dlog = ops.softmax_ce(logits, t, colsum_out=net.db[-1], grad_out=net.grad_buffer())
        dX = net.backward(dlog, have_last_bias_grad=True)
        res.append((logits.clone(), loss.clone(), dX.clone(), [w.clone() for w in net.dW], [b.clone() for b in net.db]))
    assert nets[0]._buf[("H", 0)].stride(0) == 320 and nets[1]._buf[("H", 0)].stride(0) == 256
    assert nets[0].grad_buffer().stride(0) == dims[-1]          # 128 floats = 512-byte rows, but 70 000 x 128 is below the size that pads
    a, b = res
    assert bool(torch.isfinite(a[1]).all()) and float(a[2].abs().max()) > 0
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for x, y in zip(a[3] + a[4], b[3] + b[4]):
        assert torch.equal(x, y)


def test_hub_id_structure_detector(env):
    """gnnx_spmm_plan_hub_ids_structured: an R-MAT graph as generated (hubs on the ids with few one-bits) is flagged, the same graph
    with its labels spread by the multiplicative hash or with its hubs sorted to the front as one dense block is not."""
    ops, torch = env["ops"], env["torch"]
    n, e = 400_000, 6_000_000
    src, dst = ops.rmat_edges(9, n, e, device=env["dev"])
    g = ops.CsrGraph.from_coo(src, dst, n, norm=False)
    g.make_plans(64, 256)
    assert g.plan.n_split_rows >= 512 and g.plan.hub_ids_structured() and g.plan_t.hub_ids_structured()
    gs = ops.CsrGraph.from_coo(src, dst, n, norm=False, relabel="scramble")
    gs.make_plans(64, 256)
    assert not gs.plan.hub_ids_structured() and not gs.plan_t.hub_ids_structured()
    deg = (g.rowptr[1:] - g.rowptr[:-1]) + (g.rowptr_t[1:] - g.rowptr_t[:-1])
    order = torch.sort(deg, descending=True, stable=True).indices          # vertex of rank k
    nid = torch.empty(n, dtype=torch.int32, device=env["dev"])
    nid[order] = torch.arange(n, dtype=torch.int32, device=env["dev"])
    gd = ops.CsrGraph.from_coo(src, dst, n, norm=False, relabel=nid)
    gd.make_plans(64, 256)
    assert not gd.plan.hub_ids_structured() and not gd.plan_t.hub_ids_structured()


_PC_TIMEOUT_SCRIPT = r"""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.environ["GNNX_TEST_ROOT"])
import numpy as np, torch
from __graft_entry__ import load_package
load_package()
ops = importlib.import_module("gnncpp_amd.ops"); capi = importlib.import_module("gnncpp_amd.capi")
assert capi.LIB_PATH.endswith("libgnnx_hip_exp.so")
dev = torch.device("cuda:0")
n, F = 4096, 64
rng = np.random.default_rng(5)
src = np.concatenate([np.zeros(3000, np.int32), rng.integers(0, n, 20000).astype(np.int32)])
dst = np.concatenate([rng.permutation(n)[:3000].astype(np.int32), rng.integers(0, n, 20000).astype(np.int32)])
g = ops.CsrGraph.from_coo(torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev), n)
g.make_plans(chunk=1024, max_feat=F, big_rows=0)          # every hub row (row 0) on the producer / consumer kernel
assert g.plan.n_split_rows >= 1
H = torch.rand((n, F), device=dev); bias = torch.zeros(F, device=dev)
out = torch.full((n, F), 12345.0, device=dev)
ops.aggregate_fwd(g, H, bias, out=out)                    # GNNX_PC_EXP=4: the producers leave at once, the consumer's wait times out
torch.cuda.synchronize()
assert capi.lib().gnnx_spmm_plan_status(g.plan.h) != 0, "the plan's error word is not set"
assert bool((out[0] == 12345.0).all()), "the row was written from slots that never landed"
ref = ops.aggregate_fwd(g, H, bias, use_plan=False)
assert torch.equal(out[1:], ref[1:]), "rows of the other kernels are unaffected"
try:
    ops.aggregate_fwd(g, H, bias)
except capi.GnnxError as e:
    assert "timed out" in str(e), str(e)
    print("PC_TIMEOUT_IS_LOUD")
else:
    raise SystemExit("the next call on the plan did not report the timeout")
"""


def test_pc_kernel_wait_that_gives_up_is_an_error_not_a_wrong_sum():
    """ADVICE round 4 / VERDICT weak #6: spmm_hubpc_kernel's waits are bounded (kSpinCap polls); a wait that gives up used to `break`
    and the consumer then added LDS slots that never landed -- silently wrong sums.  Now the consumer raises the plan's error word
    (pinned host memory), leaves WITHOUT storing the row, and the plan's next call -- and gnnx_spmm_plan_status -- return
    GNNX_ERR_HIP.  Forced here with the EXPERIMENTS build's GNNX_PC_EXP=4 (the producers return at once) in a child process; the
    product library has no such switch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = os.path.join(root, "gnn.cpp_amd", "libgnnx_hip_exp.so")
    assert os.path.exists(exp), "build it with __graft_entry__.build() (make -C gnn.cpp_amd/csrc EXPERIMENTS=1)"
    env_ = dict(os.environ, GNNX_HIP_LIB="exp", GNNX_PC_EXP="4", GNNX_TEST_ROOT=root)
    r = subprocess.run([sys.executable, "-c", _PC_TIMEOUT_SCRIPT], capture_output=True, text=True, timeout=600, env=env_)
    assert r.returncode == 0 and "PC_TIMEOUT_IS_LOUD" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]



@pytest.mark.parametrize("n,e,F,chunk,big", [(30000, 600000, 256, 256, 0), (30000, 600000, 256, 64, 700), (30000, 600000, 128, 64, 0),
                                             (20000, 400000, 100, 64, 0), (20000, 400000, 36, 64, 0), (8000, 200000, 320, 128, 300),
                                             (8000, 200000, 64, 16, 0), (3000, 200000, 256, 16, 0)])
def test_hub_rows_on_the_producer_consumer_kernel(env, n, e, F, chunk, big):
    """spmm_hubpc_kernel -- the longest hub rows: one consumer wavefront adds (the reference's order, one accumulator per feature),
    three producer wavefronts keep the row's slices coming through a 128 KiB LDS ring -- forced onto EVERY hub row (big = 0) or the
    rows above `big` (the rest on spmm_hub_kernel): every mode against the oracle / the unplanned kernels, BIT FOR BIT: forward
    (rowscale + bias), backward (per-entry norm), beta = 1, ReLU epilogue, column scale, values + column scale, the BatchNorm / ReLU
    prologue, the BatchNorm-backward sums; rows of every length (a few neighbours to tens of thousands: shorter than the ring, sub-chunk
    tails, rows longer than the ring many times over), full and ragged slabs, strided rows, run to run."""
    ops, torch = env["ops"], env["torch"]
    src, dst, rp, ci, g = make_graph(env, n, e, seed=270 + F + chunk)
    deg = np.diff(rp)
    assert (deg > max(chunk, big)).sum() >= 4, "test graph should have hub rows above the threshold"
    s, norm = oracle.degree_norm(rp, ci, n)
    H = synth.uniform_pm1(41, (n, F))
    bias = synth.uniform_pm1(42, (F,))
    g.make_plans(chunk=chunk, max_feat=F, big_rows=big)
    ref = oracle.aggregate_fwd(rp, ci, H, norm, bias)
    Hd, bd = dev(env, H), dev(env, bias)
    out = ops.aggregate_fwd(g, Hd, bd)
    assert same(host(out), ref), "forward"
    G = synth.uniform_pm1(43, (n, F))
    Gd = dev(env, G)
    rT, cT = oracle.csr_transpose(rp, ci, n)
    assert same(host(ops.aggregate_bwd(g, Gd)), oracle.aggregate_bwd(rT, cT, G, norm)), "backward"
    acc0 = synth.uniform_pm1(44, (n, F))
    assert torch.equal(ops.aggregate_bwd(g, Gd, out=dev(env, acc0), beta=1.0), ops.aggregate_bwd(g, Gd, out=dev(env, acc0), beta=1.0, use_plan=False))
    assert torch.equal(ops.aggregate_fwd(g, Hd, bd, relu_out=True), ops.aggregate_fwd(g, Hd, bd, relu_out=True, use_plan=False)), "ReLU epilogue"
    assert torch.equal(ops.aggregate_fwd_sym(g, Hd, bd), ops.aggregate_fwd_sym(g, Hd, bd, use_plan=False)), "colscale"
    vals = dev(env, synth.uniform_pm1(45, (len(ci),)))
    assert torch.equal(ops.spmm(g.rowptr, g.colidx, Hd, vals=vals, colscale=g.s, rowscale=g.norm, plan=g.plan),
                       ops.spmm(g.rowptr, g.colidx, Hd, vals=vals, colscale=g.s, rowscale=g.norm)), "vals + colscale"
    # the BatchNorm / ReLU prologue on every gathered row (modes 3, 4, 5)
    mean, var = ops.bn_stats(Hd)
    gamma, beta = dev(env, synth.uniform_pm1(46, (F,)) + 1.5), dev(env, synth.uniform_pm1(47, (F,), scale=0.3))
    for bn, relu_in in (((mean, var, gamma, beta, 1e-5), True), ((mean, var, gamma, beta, 1e-5), False), (None, True)):
        assert torch.equal(ops.aggregate_fwd(g, Hd, bd, bn=bn, relu_in=relu_in), ops.aggregate_fwd(g, Hd, bd, bn=bn, relu_in=relu_in, use_plan=False)), (bn is None, relu_in)
    if F % 4 == 0 and F > 64:   # BatchNorm-backward sums riding in the backward aggregation: same dY, sums to rounding, run to run
        dY1, dg1, db1 = ops.aggregate_bwd_bn_sums(g, Gd, Hd, mean, var, gamma, beta)
        dY0, dg0, db0 = ops.aggregate_bwd_bn_sums(g, Gd, Hd, mean, var, gamma, beta, use_plan=False)
        assert torch.equal(dY1, dY0) and torch.equal(dY1, ops.aggregate_bwd(g, Gd))
        tol = 1e-5 * float(dY1.abs().double().sum(0).max()) * 4
        assert float((dg1 - dg0).abs().max()) <= tol and float((db1 - db0).abs().max()) <= tol
        dY2, dg2, db2 = ops.aggregate_bwd_bn_sums(g, Gd, Hd, mean, var, gamma, beta)
        assert torch.equal(dg2, dg1) and torch.equal(db2, db1)
        Hp = torch.zeros((n, F + 12), dtype=torch.float32, device=env["dev"])
        Hp[:, :F] = Hd
        Yp = torch.zeros((n, F + 4), dtype=torch.float32, device=env["dev"])
        ops.aggregate_fwd(g, Hp[:, :F], bd, out=Yp[:, :F])
        assert torch.equal(Yp[:, :F], out) and float(Yp[:, F:].abs().max()) == 0.0, "strided rows"
    for _ in range(3):
        assert torch.equal(ops.aggregate_fwd(g, Hd, bd), out), "run to run"


@pytest.mark.parametrize("n,e,F,chunk,relu,affine", [(30000, 600000, 256, 256, True, True), (30000, 600000, 128, 64, True, True),
                                                     (20000, 400000, 100, 64, True, False), (20000, 300000, 256, 0, False, True)])
def test_backward_aggregation_with_batchnorm_sums(env, n, e, F, chunk, relu, affine):
    """gnnx_spmm_csr_bn_sums_f32: the backward aggregation of a fused BatchNorm + ReLU layer with BatchNorm's column sums accumulated in
    the store epilogue (nn.cpp:301-330's backward, operation.h:144-167): dY keeps the bits of the plain backward aggregation; dgamma
    and dbeta agree with float64 and with the separate sums pass (gnnx_bn_relu_bwd_f32) to rounding; run-to-run identical; narrow
    widths are refused loudly."""
    ops, torch, capi = env["ops"], env["torch"], env["capi"]
    src, dst, rp, ci, g = make_graph(env, n, e, seed=170 + F)
    if chunk:
        g.make_plans(chunk, F)
    H = dev(env, synth.uniform_pm1(141, (n, F)))
    G = dev(env, synth.uniform_pm1(142, (n, F)))
    gamma = dev(env, 1.0 + 0.3 * synth.uniform_pm1(143, (F,))) if affine else None
    beta = dev(env, 0.2 * synth.uniform_pm1(144, (F,))) if affine else None
    mean, var = ops.bn_stats(H)
    dY0 = ops.aggregate_bwd(g, G)
    dY, dgamma, dbeta = ops.aggregate_bwd_bn_sums(g, G, H, mean, var, gamma, beta, 1e-5, relu)
    assert torch.equal(dY, dY0), "dY must keep the plain backward aggregation's bits"
    # float64 sums with the device's own mask decision (sign of the forward value, recomputed in float32 by the separate kernel)
    Y = ops.bn_relu_fwd(H, mean, var, gamma, beta, 1e-5, relu=True)
    mask = (Y > 0) if relu else torch.ones_like(Y, dtype=torch.bool)
    g64 = torch.where(mask, dY0.double(), torch.zeros_like(dY0, dtype=torch.float64))
    xhat = (H.double() - mean.double().reshape(1, -1)) / torch.sqrt(var.double().reshape(1, -1) + 1e-5)
    ref_b, ref_g = g64.sum(0), (g64 * xhat).sum(0)
    ab_b, ab_g = g64.abs().sum(0), (g64 * xhat).abs().sum(0)
    assert float(((dbeta.double() - ref_b).abs() / torch.clamp(ab_b, min=1.0)).max()) <= 1e-5
    assert float(((dgamma.double() - ref_g).abs() / torch.clamp(ab_g, min=1.0)).max()) <= 1e-5
    _, dg_sep, db_sep = ops.bn_relu_bwd(H, None, dY0, mean, var, gamma, 1e-5, relu, beta=beta)
    assert float(((dbeta - db_sep).abs() / torch.clamp(ab_b.float(), min=1.0)).max()) <= 1e-5
    assert float(((dgamma - dg_sep).abs() / torch.clamp(ab_g.float(), min=1.0)).max()) <= 1e-5
    dY2, dgamma2, dbeta2 = ops.aggregate_bwd_bn_sums(g, G, H, mean, var, gamma, beta, 1e-5, relu)
    assert torch.equal(dgamma2, dgamma) and torch.equal(dbeta2, dbeta) and torch.equal(dY2, dY), "deterministic"
    with pytest.raises(capi.GnnxError):
        ops.aggregate_bwd_bn_sums(g, G[:, :32].contiguous(), H[:, :32].contiguous(), mean[..., :32].contiguous(),
                                  var[..., :32].contiguous(), None, None, 1e-5, relu)


@pytest.mark.parametrize("F", [256, 100, 16, 7])
def test_division_by_column_constant_is_ieee_exact(env, F):
    """The BatchNorm prologue of the fused aggregation divides by a per-column constant with ONE f64 multiply by the precomputed
    reciprocal (gnnx_spmm.hip: div_by_const; proof of correct rounding there).  Swept against numpy's IEEE float32 division on a ring
    graph (every row has exactly one neighbour, so the output IS the normalised element): operands over the whole exponent range,
    quotients that are subnormal, zero, huge and infinite, divisors from 1e-19 to 1e18 -- every element equal, NaN for NaN."""
    ops, torch = env["ops"], env["torch"]
    n = 262144
    rng = np.random.default_rng(1234 + F)
    mant = rng.uniform(1.0, 2.0, size=(n, F)).astype(np.float32)
    expo = rng.integers(-140, 120, size=(n, F))
    X = (np.where(rng.random((n, F)) < 0.5, -1.0, 1.0) * np.ldexp(mant.astype(np.float64), expo)).astype(np.float32)
    X[::97, 0] = 0.0
    X[5::101, F - 1] = np.inf
    var = np.ldexp(rng.uniform(1.0, 2.0, size=F), rng.integers(-126, 120, size=F)).astype(np.float32)
    mean = (rng.uniform(-1, 1, size=F) * np.ldexp(1.0, rng.integers(-30, 30, size=F))).astype(np.float32)
    mean[::3] = 0.0   # then x - mean is x itself: the sweep of quotients is the sweep of x
    eps = np.float32(1e-5)
    rowptr = torch.arange(n + 1, dtype=torch.int32, device=env["dev"])
    colidx = ((torch.arange(n, dtype=torch.int64, device=env["dev"]) + 1) % n).to(torch.int32)
    got = host(ops.spmm(rowptr, colidx, dev(env, X), bn=(dev(env, mean), dev(env, var), None, None, float(eps))))
    with np.errstate(all="ignore"):
        sd = np.sqrt((var + eps).astype(np.float32)).astype(np.float32)
        d = (X - mean[None, :]).astype(np.float32)
        want = (d / sd[None, :]).astype(np.float32)
        want = (want * np.float32(1.0)).astype(np.float32) + np.float32(0.0)
    want = np.roll(want, -1, axis=0)   # row i holds the element of row i + 1
    same_val = (got == want) | (np.isnan(got) & np.isnan(want))
    assert same_val.all(), f"{(~same_val).sum()} of {same_val.size} quotients differ from IEEE division"
    assert (np.abs(want[np.isfinite(want)]) < 1.2e-38).sum() > 100 and np.isinf(want).sum() > 10   # the sweep reached the edges


@pytest.mark.parametrize("F", [16, 7, 33])
def test_spmm_split_rows_plan_narrow_features(env, F):
    """The plan (hub kernel + the one-row-per-group kernel used at F <= 64), vector and scalar lanes: the oracle's bits."""
    ops = env["ops"]
    n = 5000
    src, dst, rp, ci, g = make_graph(env, n, 120000, seed=15)
    deg = np.diff(rp)
    assert deg.max() > 500
    s, norm = oracle.degree_norm(rp, ci, n)
    H = synth.uniform_pm1(33, (n, F))
    bias = synth.uniform_pm1(34, (F,))
    ref = oracle.aggregate_fwd(rp, ci, H, norm, bias)
    plan = ops.SpmmPlan(g.rowptr, 64, F)
    assert plan.n_split_rows == int((deg > 64).sum())
    out = host(ops.spmm(g.rowptr, g.colidx, dev(env, H), rowscale=g.norm, bias=dev(env, bias), plan=plan))
    assert same(out, ref)
    assert same(host(ops.spmm(g.rowptr, g.colidx, dev(env, H), rowscale=g.norm, bias=dev(env, bias))), ref)


def test_spmm_vals_and_sym_mode(env):
    """Mode SYM (textbook D^-1/2 A D^-1/2) and per-edge values, vs float64 numpy."""
    ops = env["ops"]
    n, F = 3000, 64
    src, dst, rp, ci, g = make_graph(env, n, 30000, seed=9)
    H = synth.uniform_pm1(41, (n, F))
    s = host(g.s)
    vals = synth.uniform_pm1(42, (len(ci),))
    got = host(ops.spmm(g.rowptr, g.colidx, dev(env, H), vals=dev(env, vals), colscale=g.s, rowscale=g.s))
    ref = np.zeros((n, F))
    rows = np.repeat(np.arange(n), np.diff(rp))
    np.add.at(ref, rows, (vals.astype(np.float64) * s[ci])[:, None] * H[ci].astype(np.float64))
    ref *= s[:, None]
    assert_close(got, ref.astype(np.float32), "SYM + vals")


@pytest.mark.parametrize("M,N,K", [(2708, 16, 1433), (2708, 7, 16), (1000, 128, 128), (777, 256, 256),
                                   (4096, 100, 100), (130, 130, 130), (1, 1, 1), (513, 257, 33)])
def test_gemm_variants_vs_oracle(env, M, N, K):
    """All three products of the path on one shape triple: X.W^T, dH.W, dH^T.X."""
    ops = env["ops"]
    X = synth.uniform_pm1(51, (M, K))
    W = synth.uniform_pm1(52, (N, K), scale=1.0 / np.sqrt(K))
    H = host(ops.linear_fwd(dev(env, X), dev(env, W)))
    assert_close(H, oracle.linear_fwd(X, W), "X.W^T")
    dH = synth.uniform_pm1(53, (M, N))
    dX, dW = ops.linear_bwd(dev(env, dH), dev(env, X), dev(env, W))
    rdX, rdW = oracle.linear_bwd(dH, X, W)
    assert_close(host(dX), rdX, "dH.W")
    d64, x64 = dH.astype(np.float64), X.astype(np.float64)
    assert_close(host(dW), rdW, "dH^T.X", absum=np.abs(d64).T @ np.abs(x64), exact=d64.T @ x64)


def test_gemm_splitk_long_reduction(env):
    """dW over many node rows runs split-K + in-order slab reduction: check against float64 and determinism."""
    ops = env["ops"]
    n, fo, fi = 200000, 64, 96
    dH = synth.uniform_pm1(61, (n, fo))
    X = synth.uniform_pm1(62, (n, fi))
    a = host(ops.gemm(dev(env, dH), dev(env, X), transA=True))
    b = host(ops.gemm(dev(env, dH), dev(env, X), transA=True))
    assert np.array_equal(a, b)
    ref = dH.astype(np.float64).T @ X.astype(np.float64)
    # error of an fp32 sum of n terms of magnitude <= 1: bound by 1e-5 * sum|terms| ~ generous vs sqrt(n)*eps
    assert np.abs(a - ref).max() <= 1e-5 * np.abs(dH).astype(np.float64).T.dot(np.abs(X).astype(np.float64)).max()


@pytest.mark.parametrize("n,fo,fi", [(1_000_000, 256, 256), (262_144, 512, 256), (200_064, 256, 256), (500_032, 128, 128), (131_072, 128, 384)])
def test_gemm_splitk_lds_dma_kernel(env, n, fo, fi):
    """dW = dH^T . X with widths that are multiples of 128 (256 x 256 tiles where both are multiples of 256) and K % 64 == 0 takes gemm_dma_tn_kernel (LDS-DMA, split-K slabs, in-order slab
    reduction): float64 check with the condition-aware bound, run-to-run identical, and the same bits with beta = 1 as adding
    to the previous result by hand."""
    ops, torch = env["ops"], env["torch"]
    dH = ops.uniform_pm1(63, (n, fo), device=env["dev"])
    X = ops.uniform_pm1(64, (n, fi), device=env["dev"])
    a = ops.gemm(dH, X, transA=True)
    b = ops.gemm(dH, X, transA=True)
    assert torch.equal(a, b)
    ref = dH[:, :64].double().t() @ X.double()                     # 64 output rows in float64 (covers the first wavefront row)
    bound = 1e-5 * float((dH[:, :64].abs().double().t() @ X.abs().double()).max())
    assert float((a[:64].double() - ref).abs().max()) <= bound
    c = a.clone()
    ops.gemm(dH, X, transA=True, out=c, beta=1.0)
    assert torch.equal(c, a + a)


@pytest.mark.parametrize("M,N,K", [(300_077, 256, 256), (70_000, 512, 128), (5_000, 256, 256), (40_000, 128, 128), (3_000, 100, 100)])
def test_gemm_with_fused_relu_mask_and_colsum_epilogue(env, M, N, K):
    """gnnx_gemm_relu_colsum_f32 -- G = (dH . W) (.) (Y > 0) and db = colsum(G) in the GEMM epilogue (stacked layers' backward) --
    against the three separate passes: G bit-identical (same GEMM chain, the mask is exact), db within rounding of the
    float64 column sums; shapes with whole 256-row tiles + ragged tail (fused kernel + fallback rows), widths the fused kernel
    does not take (pure fallback), and run-to-run identical."""
    ops, torch = env["ops"], env["torch"]
    dH = ops.uniform_pm1(81, (M, K), device=env["dev"])
    W = ops.uniform_pm1(82, (K, N), scale=K ** -0.5, device=env["dev"])
    Y = torch.relu(ops.uniform_pm1(83, (M, N), device=env["dev"]))      # a ReLU output: about half zeros
    G, db = ops.gemm_relu_colsum(dH, W, Y)
    G2, db2 = ops.gemm_relu_colsum(dH, W, Y)
    assert torch.equal(G, G2) and torch.equal(db, db2)
    ref = ops.gemm(dH, W)
    ref = torch.where(Y > 0, ref, torch.zeros_like(ref))
    assert torch.equal(G, ref), "masked GEMM output differs from gemm + mask"
    exact = ref.double().sum(0)
    bound = 1e-5 * max(1.0, float(ref.abs().double().sum(0).max()))
    assert float((db.double() - exact).abs().max()) <= bound


@pytest.mark.parametrize("M,N,K,offset", [(300_077, 256, 256, 0.0), (100_000, 128, 128, 3.0), (50_000, 256, 64, -20.0), (4_000, 100, 100, 0.5)])
def test_gemm_with_fused_batchnorm_statistics_opt_in(env, M, N, K, offset):
    """gnnx_gemm_bn_stats_f32 (opt-in): H = X . W^T with the batch mean / biased variance of H's columns from the GEMM epilogue
    (single-pass, shifted by row 0 of H).  H must be the bits of the plain product; the statistics are checked against float64
    -- also with a large common offset in the data (mean^2 >> var), where an unshifted E[x^2] - E[x]^2 in float would lose
    everything -- and against the exact two-pass kernel; a shape the fused kernel does not take falls back to the exact pair."""
    ops, torch = env["ops"], env["torch"]
    X = ops.uniform_pm1(85, (M, K), device=env["dev"]) + offset
    W = ops.uniform_pm1(86, (N, K), scale=K ** -0.5, device=env["dev"])
    H, mean, var = ops.linear_fwd_bn_stats(X, W)
    H0 = ops.linear_fwd(X, W)
    assert torch.equal(H, H0)
    m64, v64 = H0.double().mean(0), H0.double().var(0, unbiased=False)
    assert float((mean.double() - m64).abs().max()) <= 1e-5 * max(1.0, float(m64.abs().max()))
    assert float(((var.double() - v64).abs() / v64.clamp_min(1e-30)).max()) <= 1e-4
    m2, v2 = ops.bn_stats(H0)
    assert float(((var - v2).abs() / v2.clamp_min(1e-30)).max()) <= 1e-4


def test_gcn_stack_backward_fused_equals_unfused(env):
    ops, torch = env["ops"], env["torch"]
    n, e, F = 30_000, 300_000, 256
    src, dst, rp, ci, g = make_graph(env, n, e, seed=41)
    net = ops.GcnStack(g, [F, F, F], seed=7, device=env["dev"])
    X = ops.uniform_pm1(91, (n, F), device=env["dev"])
    dOut = ops.uniform_pm1(92, (n, F), device=env["dev"])
    net.forward(X)
    ga = net.backward(dOut, fused=True).clone()
    dWa, dba = [w.clone() for w in net.dW], [b.clone() for b in net.db]
    gb = net.backward(dOut, fused=False)
    assert torch.equal(ga, gb)
    for a, b in zip(dWa, net.dW):
        assert torch.equal(a, b)
    for a, b in zip(dba, net.db):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("dims", [[100, 100, 100, 100], [100, 72, 47], [128, 120, 100, 47]])
def test_gcn_stack_padded_streamed_layout_same_bits(env, dims):
    """Widths off the 128 grid: the stack that stores its streamed matrices (Y_l, dH_l, W_l, dW_l) on 128-float strides and keeps
    the gathered ones (H_l, G_l) at their own width gives the same logits and input gradient bit for bit as the
    stack on packed storage (dW, db: to the rounding of split-K / of the column-sum order), pads stay zero through forward, backward and an SGD step."""
    ops, torch = env["ops"], env["torch"]
    n, e = 120_000, 1_000_000
    src, dst, rp, ci, g = make_graph(env, n, e, seed=43)
    X = ops.uniform_pm1(93, (n, dims[0]), device=env["dev"])
    dOut = ops.uniform_pm1(94, (n, dims[-1]), device=env["dev"])
    nets = [ops.GcnStack(g, dims, seed=9, device=env["dev"], pad_streamed=ps) for ps in (True, False)]
    assert nets[0].padded and not nets[1].padded and nets[0].P[0] == 128
    res = []
    for net in nets:
        for l in range(len(dims) - 1):
            net.b[l].copy_(ops.uniform_pm1(95 + l, (dims[l + 1],), scale=0.2, device=env["dev"]))
        out = net.forward(net.pad_input(X)).clone()
        gin = net.backward(dOut).clone()
        res.append((out, gin, [w.clone() for w in net.dW], [b.clone() for b in net.db]))
    (oa, ga, dWa, dba), (ob, gb, dWb, dbb) = res
    assert torch.equal(oa, ob) and torch.equal(ga, gb)
    for a, b in zip(dba[:-1], dbb[:-1]):   # column sums from the fused epilogue vs the separate pass: different (fixed) orders
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
    assert torch.equal(dba[-1], dbb[-1])
    for a, b in zip(dWa, dWb):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()))
    net = nets[0]
    # a plain (unpadded) input is accepted too, same result; a wide buffer with non-zero pad columns is refused
    assert torch.equal(net.forward(X), oa)
    if net.P[0] != dims[0]:
        wide = torch.ones((n, net.P[0]), dtype=torch.float32, device=env["dev"])
        wide[:, :dims[0]] = X
        with pytest.raises(ValueError):
            net.forward(wide[:, :dims[0]])
    with pytest.raises(ValueError):
        ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), n, relabel=torch.zeros(n, dtype=torch.int32, device=env["dev"]))
    # two more training steps on both stacks: pads stay zero (two layers whose widths differ inside one 128-float bucket -- 120 and
    # 100 -- must not share a zero-padded gradient buffer), and the padded stack keeps tracking the packed one
    for step in range(2):
        for nt in nets:
            nt.forward(nt.pad_input(X))
            nt.backward(dOut)
            nt.step(lr=0.01, weight_decay=1e-4)
        for l in range(len(dims) - 1):
            do, di = dims[l + 1], dims[l]
            for t in (net.Wp[l], net.dWp[l]):
                assert not t[do:].any() and not t[:, di:].any(), f"pads of W / dW must stay zero (step {step}, layer {l})"
            torch.testing.assert_close(net.W[l], nets[1].W[l], rtol=1e-4, atol=1e-5 * float(nets[1].W[l].abs().max()))
        for (hp, Yp), l in zip(net._saved, range(len(dims) - 1)):
            assert not Yp[:, dims[l + 1]:].any() and not hp[:, dims[l]:].any()


def test_gemm_beta_accumulate(env):
    ops = env["ops"]
    A = synth.uniform_pm1(71, (300, 40))
    B = synth.uniform_pm1(72, (50, 40))
    C0 = synth.uniform_pm1(73, (300, 50))
    out = dev(env, C0.copy())
    ops.gemm(dev(env, A), dev(env, B), transB=True, out=out, alpha=1.0, beta=1.0)
    assert_close(host(out), (C0.astype(np.float64) + A.astype(np.float64) @ B.astype(np.float64).T).astype(np.float32))


@pytest.mark.parametrize("n,F", [(100000, 256), (50000, 100), (30000, 16), (999, 7), (5, 300)])
def test_colsum_vs_oracle(env, n, F):
    G = synth.uniform_pm1(81, (n, F))
    got = host(env["ops"].colsum(dev(env, G)))
    ref64 = G.astype(np.float64).sum(0)
    assert np.abs(got - ref64).max() <= 1e-5 * max(1.0, np.abs(G).astype(np.float64).sum(0).max())
    if n <= 50000:
        assert_close(got, oracle.colsum(G), "colsum vs sequential oracle", absum=np.abs(G).astype(np.float64).sum(0),
                     exact=ref64)


def test_halo_pack_unpack(env):
    ops, torch = env["ops"], env["torch"]
    n, F = 5000, 128
    X = synth.uniform_pm1(91, (n, F))
    idx = np.random.default_rng(0).permutation(n)[:1234].astype(np.int32)
    got = host(ops.gather_rows(dev(env, X), dev(env, idx)))
    assert np.array_equal(got, X[idx])
    Y = synth.uniform_pm1(92, (n, F))
    Yd = dev(env, Y.copy())
    ops.scatter_add_rows(dev(env, got), dev(env, idx), Yd)
    ref = Y.copy()
    ref[idx] += got
    assert np.array_equal(host(Yd), ref)
    del torch


# ------------------------------------------------------------------ edge cases the domain has
def test_edge_cases_empty_isolated_and_errors(env):
    ops, capi, torch = env["ops"], env["capi"], env["torch"]
    # no edges at all: every row is bias (reference: A = 0 => agg = 0, norm = 0)
    n, F = 17, 8
    e0 = torch.empty(0, dtype=torch.int32, device=env["dev"])
    g = ops.CsrGraph.from_coo(e0, e0, n)
    assert g.nnz == 0 and host(g.rowptr).tolist() == [0] * (n + 1)
    assert np.all(host(g.s) == 1.0) and np.all(host(g.norm) == 0.0)
    bias = synth.uniform_pm1(1, (F,))
    out = host(ops.aggregate_fwd(g, dev(env, synth.uniform_pm1(2, (n, F))), dev(env, bias)))
    assert np.array_equal(out, np.tile(bias, (n, 1)))
    # only self loops + duplicates
    src = np.array([0, 1, 2, 2, 2, 3, 3], dtype=np.int32)
    dst = np.array([0, 1, 2, 3, 3, 3, 2], dtype=np.int32)
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), 4)
    assert host(g.rowptr).tolist() == [0, 0, 0, 1, 2] and host(g.colidx).tolist() == [3, 2]
    # out-of-range endpoint -> the reference's Data ctor message (graph.cpp:89-90)
    bad = np.array([0, 9], dtype=np.int32)
    with pytest.raises(capi.GnnxError) as ei:
        ops.CsrGraph.from_coo(dev(env, bad), dev(env, bad[::-1].copy()), 4)
    assert ei.value.status == -3 and "max value in edge_index" in str(ei.value)
    # shape errors surface as statuses, not crashes
    with pytest.raises(capi.GnnxError):
        ops.gemm(dev(env, np.zeros((4, 5), np.float32)), dev(env, np.zeros((6, 7), np.float32)))
    # single node, single feature
    g1 = ops.CsrGraph.from_coo(e0, e0, 1)
    assert host(ops.aggregate_fwd(g1, dev(env, np.ones((1, 1), np.float32)))).tolist() == [[0.0]]


def test_device_generators_match_numpy(env):
    ops = env["ops"]
    s, d = ops.rmat_edges(1, 1_000_000, 300_000)
    s2, d2 = synth.rmat_edges(1, 1_000_000, 300_000)
    assert np.array_equal(host(s), s2) and np.array_equal(host(d), d2)
    s, d = ops.rmat_edges(3, 777, 10_000, a=0.45, b=0.22, c=0.22, first_edge=123)
    s2, d2 = synth.rmat_edges(3, 777, 10_000, a=0.45, b=0.22, c=0.22, first_edge=123)
    assert np.array_equal(host(s), s2) and np.array_equal(host(d), d2)
    u = ops.uniform_pm1(5, (1000, 33), scale=0.125)
    assert np.array_equal(host(u), synth.uniform_pm1(5, (1000, 33), scale=0.125))


# ------------------------------------------------------------------ full-size, size-independent properties
@pytest.mark.parametrize("n,e,F,abc,seed", [(1_000_000, 10_000_000, 128, (0.57, 0.19, 0.19), 1),      # BASELINE configs[2]
                                            (2_400_000, 62_000_000, 100, (0.45, 0.22, 0.22), 3),     # configs[4]
                                            (10_000_000, 100_000_000, 256, (0.57, 0.19, 0.19), 2)])  # configs[3]
def test_full_size_properties(env, n, e, F, abc, seed):
    """BASELINE.json's full-size graphs: too big for the CPU oracle to be quick on every feature, so check
    (a) the CSR against its own invariants, (b) the aggregation by a checksum of checksums:
    1^T (A.H) == indeg^T . H in float64, (c) linearity, (d) sampled rows exactly against the oracle's arithmetic,
    (e) <A.H, G> == <H, A^T.G> (forward/backward adjointness)."""
    ops, torch = env["ops"], env["torch"]
    src, dst = ops.rmat_edges(seed, n, e, *abc)
    g = ops.CsrGraph.from_coo(src, dst, n)
    del src, dst
    g.make_plans(chunk=1024, max_feat=F)
    rp, ci = host(g.rowptr).astype(np.int64), host(g.colidx)
    assert rp[0] == 0 and rp[-1] == g.nnz and np.all(np.diff(rp) >= 0)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    key = rows * n + ci
    assert np.all(np.diff(key) > 0), "CSR must be strictly sorted by (row, col): dedupe + order"
    assert not np.any(rows == ci), "self loops must be stripped"
    H = ops.uniform_pm1(2, (n, F))
    Y = ops.spmm(g.rowptr, g.colidx, H, plan=g.plan)
    indeg = torch.from_numpy(np.bincount(ci, minlength=n).astype(np.float64)).to(env["dev"])
    lhs = Y.double().sum(0)
    rhs = indeg @ H.double()
    scale = (indeg @ H.double().abs()).clamp_min(1.0)
    assert float(((lhs - rhs).abs() / scale).max()) < 1e-6
    # sampled rows, exact (rows below the plan's chunk keep the reference order)
    Hh = host(H)
    Yh = host(Y)
    deg = np.diff(rp)
    rng = np.random.default_rng(1)
    for r in list(rng.integers(0, n, 200)) + [int(np.argmax(deg))]:
        acc = np.zeros(F, dtype=np.float32)
        for p in range(rp[r + 1] - 1, rp[r] - 1, -1):
            acc = acc + Hh[ci[p]]
        if deg[r] <= 1024:
            assert np.array_equal(Yh[r], acc)
        else:
            assert np.abs(Yh[r] - acc).max() <= 1e-5 * max(1.0, np.abs(acc).max())
    # linearity and adjointness
    H2 = ops.uniform_pm1(3, (n, F))
    Y2 = ops.spmm(g.rowptr, g.colidx, H2, plan=g.plan)
    Y12 = ops.spmm(g.rowptr, g.colidx, H + H2, plan=g.plan)
    assert float((Y12 - (Y + Y2)).abs().max()) <= 1e-5 * float(Y12.abs().max())
    G = ops.uniform_pm1(4, (n, F))
    dH = ops.spmm(g.rowptr_t, g.colidx_t, G, plan=g.plan_t)
    a = float((Y.double() * G.double()).sum())
    b = float((H.double() * dH.double()).sum())
    assert abs(a - b) <= 1e-5 * max(abs(a), abs(b), 1.0)


# ------------------------------------------------------------------ sharded layout on one GPU
def test_shard_layout_on_gpu_bit_identical(env):
    """Rank r of a world-3 partition, built by the product planner (gnn.cpp_amd/shard.py) with the device CSR
    builder; halo rows are filled from the full matrix (what the all-to-all-v delivers -- the exchange itself
    is covered by the gloo tests).  The [local | halo] SpMM must equal the unsharded rows bit for bit."""
    ops, torch = env["ops"], env["torch"]
    shard = importlib.import_module("gnncpp_amd.shard")
    n, e, F, world = 30000, 400000, 128, 3
    src, dst = synth.rmat_edges(17, n, e)
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), n)
    H = ops.uniform_pm1(18, (n, F))
    G = ops.uniform_pm1(19, (n, F))
    bias = ops.uniform_pm1(20, (F,))
    out_ref = host(ops.aggregate_fwd(g, H, bias))
    dH_ref = host(ops.aggregate_bwd(g, G))

    def builder(s_, d_, n_rows, n_cols):
        rp, ci = ops.CsrGraph.csr_from_coo(s_, d_, max(n_rows, n_cols), flags=1)
        return rp[: n_rows + 1].contiguous(), ci

    for partition in ("deal", "contiguous"):
        part = None
        for rank in range(world):
            plan = shard.ShardPlan(dev(env, src), dev(env, dst), n, rank, world, None, builder, partition=part or partition)
            part = (plan.owner, plan.nid, plan.cuts)
            cuts = plan.cuts
            v, nl = plan.verts, plan.n_local   # original ids of this rank's rows, in local order
            f, b = plan.fwd, plan.bwd
            hf, hb = plan.orig_ids(f.halo), plan.orig_ids(b.halo)
            Hext = torch.cat([H[v], H[hf]])
            out = ops.spmm(f.rowptr, f.colidx, Hext, rowscale=g.norm[v].contiguous(), bias=bias, n_rows=nl)
            assert np.array_equal(host(out), out_ref[host(v)])
            Gext = torch.cat([G[v], G[hb]])
            norm_ext = torch.cat([g.norm[v], g.norm[hb]])
            dH = ops.spmm(b.rowptr, b.colidx, Gext, colscale=norm_ext, n_rows=nl)
            assert np.array_equal(host(dH), dH_ref[host(v)])
            # with a plan (hub rows on the sequential hub kernel) the shard's result keeps its bits
            pl = ops.SpmmPlan(f.rowptr, 256, F)
            out_p = ops.spmm(f.rowptr, f.colidx, Hext, rowscale=g.norm[v].contiguous(), bias=bias, n_rows=nl, plan=pl)
            assert torch.equal(out_p, out)
    assert cuts[0] == 0 and cuts[-1] == n


# ------------------------------------------------------------------ next row: BatchNorm + ReLU
def test_golden_full_layer_with_batchnorm_relu(gcase, env):
    """transform -> BatchNorm -> ReLU -> aggregate -> + bias == the reference's GCNConv::forward output."""
    ops, g = env["ops"], gcase["g"]
    H = dev(env, gcase["ref_H"])
    mean, var = ops.bn_stats(H)
    Hn = ops.bn_relu_fwd(H, mean, var, relu=True)
    ref_Hn, ref_mean, ref_var = oracle.bn_relu_fwd(gcase["ref_H"])
    assert_close(host(mean), ref_mean, "batch mean")
    assert_close(host(var), ref_var, "batch var")
    assert_close(host(Hn), ref_Hn, "BatchNorm+ReLU")
    out = ops.aggregate_fwd(g, Hn, dev(env, gcase["bias"]))
    assert_close(host(out), gcase["ref_out_full"], "full layer", absum=None)
    # the normalise/affine/ReLU pass itself is bit-exact when fed the reference-order statistics
    Hn2 = ops.bn_relu_fwd(H, dev(env, ref_mean), dev(env, ref_var), relu=True)
    sd_ok = np.power(ref_var + np.float32(1e-5), np.float32(0.5), dtype=np.float32) == np.sqrt(ref_var + np.float32(1e-5), dtype=np.float32)
    got, ref = host(Hn2), ref_Hn
    assert np.array_equal(got[:, sd_ok], ref[:, sd_ok])


def test_batchnorm_relu_backward_reference_quirk_vs_oracle(env, gcase):
    """gnnx_bn_relu_bwd_quirk_f32 (opt-in): the gradient through BatchNorm + ReLU that the REFERENCE's traversal delivers (batch
    statistics as constants, operation.h:80-88).  Checker: the oracle's restatement, which is pinned bit-exact to the reference's
    own through-layer gradients (tests/test_oracle_vs_reference.py)."""
    ops = env["ops"]
    H = gcase["ref_H"]
    dY = synth.uniform_pm1(77, H.shape)
    gamma = synth.uniform_pm1(78, (H.shape[1],)) + 1.5
    beta = synth.uniform_pm1(79, (H.shape[1],), scale=0.3)
    Hd, dYd, gd, bd = dev(env, H), dev(env, dY), dev(env, gamma), dev(env, beta)
    mean, var = ops.bn_stats(Hd)
    dX, dgamma, dbeta = ops.bn_relu_bwd(Hd, None, dYd, mean, var, gd, relu=True, beta=bd, reference_quirk=True)
    rX, rgamma, rbeta = oracle.bn_relu_bwd_quirk(H, dY, gamma, beta)
    # a unit whose pre-activation is within rounding of 0 may fall on the other side of the ReLU: compare where both agree
    Y64 = ((H.astype(np.float64) - H.astype(np.float64).mean(0)) / np.sqrt(H.astype(np.float64).var(0) + 1e-5)) * gamma + beta
    safe = np.abs(Y64) > 1e-4
    err = np.abs(host(dX) - rX) / np.maximum(1.0, np.abs(rX))
    assert err[safe].max() <= 1e-5
    if safe.all():
        g = np.abs(dY.astype(np.float64))
        assert np.abs(host(dbeta) - rbeta).max() <= 1e-5 * max(1.0, g.sum(0).max())
        assert np.abs(host(dgamma) - rgamma).max() <= 1e-5 * max(1.0, (g * np.abs(Y64 - beta) / np.abs(gamma)).sum(0).max())
    # and against the mathematically correct backward the two differ (that is the point of the switch)
    if H.shape[0] >= 30:
        dXc, _, _ = ops.bn_relu_bwd(Hd, None, dYd, mean, var, gd, relu=True, beta=bd)
        assert float((dXc - dX).abs().max()) > 1e-3


@pytest.mark.parametrize("n,F,relu,bn", [(5000, 64, True, True), (3000, 100, True, True), (2000, 7, False, True), (4000, 256, True, False)])
def test_batchnorm_relu_backward_vs_float64(env, n, F, relu, bn):
    """Gradients of y = relu(bn(x)) against a float64 evaluation of the textbook formulas."""
    ops = env["ops"]
    X = synth.uniform_pm1(201, (n, F)) * 2.0 + 0.3
    gamma = synth.uniform_pm1(202, (F,)) + 1.5
    beta = synth.uniform_pm1(203, (F,), scale=0.3)
    dY = synth.uniform_pm1(204, (n, F))
    Xd, gd, bd, dYd = dev(env, X), dev(env, gamma), dev(env, beta), dev(env, dY)
    if bn:
        mean, var = ops.bn_stats(Xd)
        Y = ops.bn_relu_fwd(Xd, mean, var, gd, bd, relu=relu)
        dX, dgamma, dbeta = ops.bn_relu_bwd(Xd, Y, dYd, mean, var, gd, relu=relu)
    else:
        Y = ops.bn_relu_fwd(Xd, relu=True)
        dX, dgamma, dbeta = ops.bn_relu_bwd(Xd, Y, dYd, relu=True)
    x = X.astype(np.float64)
    if bn:
        mu, v = x.mean(0), x.var(0)
        rstd = 1.0 / np.sqrt(v + 1e-5)
        xhat = (x - mu) * rstd
        y = xhat * gamma + beta
    else:
        y = x
    g = dY.astype(np.float64) * ((y > 0) if relu else 1.0)
    assert_close(host(Y), (np.maximum(y, 0) if relu else y).astype(np.float32), "forward")
    if bn:
        dbeta_ref, dgamma_ref = g.sum(0), (g * xhat).sum(0)
        dx_ref = gamma * rstd * (g - dbeta_ref / n - xhat * dgamma_ref / n)
        scale = np.abs(g).sum(0)
        assert np.abs(host(dbeta) - dbeta_ref).max() <= 1e-5 * max(1.0, scale.max())
        assert np.abs(host(dgamma) - dgamma_ref).max() <= 1e-5 * max(1.0, (np.abs(g * xhat)).sum(0).max())
        assert np.abs(host(dX) - dx_ref).max() <= 1e-4 * max(1.0, np.abs(dx_ref).max())
    else:
        assert np.array_equal(host(dX), g.astype(np.float32))


@pytest.mark.parametrize("n,F", [(50001, 256), (7000, 100), (4099, 1024), (3000, 8), (2003, 1020), (5, 12)])
def test_batchnorm_backward_vector_kernels_equal_the_scalar_ones(env, n, F):
    """The 16-bytes-per-lane forms of the two backward passes (taken for 16-byte aligned operands, F % 4 == 0) against the scalar
    kernels (forced by an odd leading dimension): dX bit for bit -- the element arithmetic is the same, given the same sums --
    dgamma / dbeta to the rounding of a different (fixed) summation order; with and without the stored forward output, and the
    mask-only and the reference-quirk forms."""
    ops, torch = env["ops"], env["torch"]
    dev_ = env["dev"]
    X = ops.uniform_pm1(1400, (n, F), device=dev_) * 2.0 + 0.3
    dY = ops.uniform_pm1(1401, (n, F), device=dev_)
    gamma = ops.uniform_pm1(1402, (F,), device=dev_) + 1.5
    beta = ops.uniform_pm1(1403, (F,), scale=0.3, device=dev_)
    odd = lambda t: torch.cat([t, torch.zeros((t.shape[0], 1), dtype=t.dtype, device=dev_)], 1)[:, :F]   # ld = F + 1: scalar kernels
    mean, var = ops.bn_stats(X)
    Y = ops.bn_relu_fwd(X, mean, var, gamma, beta, relu=True)
    for Yv in (Y, None):
        for quirk in (False, True):
            a = ops.bn_relu_bwd(X, Yv, dY, mean, var, gamma, relu=True, beta=beta, reference_quirk=quirk)
            b = ops.bn_relu_bwd(odd(X), None if Yv is None else odd(Yv), odd(dY), mean, var, gamma, relu=True, beta=beta, reference_quirk=quirk)
            g64 = (dY.double() * (Y > 0)).abs().sum(0)
            for k in (1, 2):   # dgamma, dbeta
                assert float((a[k].double() - b[k].double()).abs().max()) <= 1e-5 * max(1.0, float(g64.max()) * 4.0)
            if quirk:
                assert torch.equal(a[0], b[0])   # no sums in the element formula
            else:           # same element formula on sums that differ by rounding
                torch.testing.assert_close(a[0], b[0], rtol=1e-4, atol=1e-5 * max(1.0, float(b[0].abs().max())))
    # given the SAME sums the apply pass is bit-identical: the sharded halves take the sums as inputs
    shard_sums = ops.bn_relu_bwd(odd(X), None, odd(dY), mean, var, gamma, relu=True, beta=beta)
    import ctypes as C
    capi = env["capi"]
    outs = []
    for Xv, dYv in ((X, dY), (odd(X), odd(dY))):
        dX = torch.empty((n, F), dtype=torch.float32, device=dev_)
        wsb = C.c_size_t(0)
        capi.call("gnnx_bn_workspace", n, F, C.byref(wsb))
        ws = torch.empty(wsb.value, dtype=torch.uint8, device=dev_)
        capi.call("gnnx_bn_relu_bwd_apply_f32", ops._ptr(Xv), Xv.stride(0), None, 0, ops._ptr(dYv), dYv.stride(0), n, F, ops._ptr(mean),
                  ops._ptr(var), 1e-5, ops._ptr(gamma), ops._ptr(beta), 1, ops._ptr(shard_sums[1]), ops._ptr(shard_sums[2]), n, ops._ptr(dX), F,
                  ops._ptr(ws), wsb.value, ops._stream())
        outs.append(dX)
    assert torch.equal(outs[0], outs[1])
    # mask only (the ReLU between stacked layers)
    assert torch.equal(ops.bn_relu_bwd(Y, Y, dY, relu=True)[0], ops.bn_relu_bwd(odd(Y), odd(Y), odd(dY), relu=True)[0])


@pytest.mark.parametrize("n,F", [(40003, 256), (9000, 100), (513, 1024), (3000, 8), (2003, 1020), (5, 12)])
def test_streaming_elementwise_vector_kernels_equal_the_scalar_ones(env, n, F):
    """The 16-bytes-per-thread forms of the broadcast binary ops (rowscale / bias-add are two of them), the row sum and the
    BatchNorm forward apply, against the scalar kernels (forced by an odd leading dimension or a misaligned base): bit for bit."""
    ops, torch = env["ops"], env["torch"]
    dev_ = env["dev"]
    X = ops.uniform_pm1(1500, (n, F), device=dev_) * 3.0
    Y = ops.uniform_pm1(1501, (n, F), device=dev_) + 2.0
    v = ops.uniform_pm1(1502, (n,), device=dev_) + 1.5
    b = ops.uniform_pm1(1503, (F,), device=dev_) + 1.5
    wide = torch.zeros((n, F + 1), dtype=torch.float32, device=dev_)
    wide[:, :F] = X
    Xo = wide[:, :F]                                    # ld = F + 1: scalar kernels
    assert torch.equal(ops.rowscale(X, v), ops.rowscale(Xo, v))
    assert torch.equal(ops.bias_add(X, b), ops.bias_add(Xo, b))
    assert torch.equal(ops.rowsum(X), ops.rowsum(Xo))
    # the broadcast op on [N,F] (op) {[N,F], [N,1], [F], scalar}: vector form vs the scalar form on a base shifted by 4 bytes
    flat = torch.zeros(n * F + 1, dtype=torch.float32, device=dev_)
    flat[1:] = X.reshape(-1)
    Xm = flat[1:].view(n, F)                            # contiguous but not 16-byte aligned: scalar kernel
    one = torch.full((1,), 1.25, dtype=torch.float32, device=dev_)
    for op in ("add", "sub", "mul", "div"):
        for other in (Y, v.reshape(n, 1).contiguous(), b, one):
            assert torch.equal(ops.binary(op, X, other), ops.binary(op, Xm, other)), (op, tuple(other.shape))
            assert torch.equal(ops.binary(op, other, X), ops.binary(op, other, Xm)), (op, tuple(other.shape), "swapped")
    ref = X.double() * v.double()[:, None]
    assert float((ops.rowscale(X, v).double() - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    mean, var = ops.bn_stats(X)
    for kw in (dict(), dict(gamma=b), dict(gamma=b, beta=v[:F].contiguous() if n >= F else b)):
        a1 = ops.bn_relu_fwd(X, mean, var, relu=True, **kw)
        a2 = ops.bn_relu_fwd(Xo, mean, var, relu=True, **kw)
        assert torch.equal(a1, a2)
    assert torch.equal(ops.bn_relu_fwd(X, relu=True), ops.bn_relu_fwd(Xo, relu=True))


@pytest.mark.parametrize("n,e,F,abc", [(200_000, 2_000_000, 256, (0.57, 0.19, 0.19)), (50_001, 400_000, 100, (0.57, 0.19, 0.19)),
                                       (3000, 20_000, 64, None)])
def test_vertex_relabelling_same_bits(env, n, e, F, abc):
    """CsrGraph.from_coo(relabel="scramble") -- vertex v stored at row (v * 2654435761) mod n, what bench.py runs by default -- gives
    every vertex the SAME BITS as the as-generated order, in norm, forward and backward aggregation (exact mode and with the
    load-balancing plan, whose chunks cut a hub row at the same entries) and through a whole layer step; and both equal the oracle
    on the original labels.  The C-ABI relabelling (gnnx_partition_scramble with one rank) is the same permutation."""
    ops, capi, torch = env["ops"], env["capi"], env["torch"]
    dev_ = env["dev"]
    if abc is None:
        s_np, d_np = synth.uniform_edges(51, n, e)
        src, dst = dev(env, s_np), dev(env, d_np)
    else:
        src, dst = ops.rmat_edges(51, n, e, *abc, device=dev_)
    g0 = ops.CsrGraph.from_coo(src, dst, n)
    g1 = ops.CsrGraph.from_coo(src, dst, n, relabel="scramble")
    nid = g1.nid.long()
    assert sorted(g1.nid.tolist()) == list(range(n)) if n <= 5000 else int(torch.unique(nid).numel()) == n
    import ctypes as C
    nid_c = torch.arange(n, dtype=torch.int32, device=dev_)
    cuts = (C.c_int64 * 2)(0, n)
    capi.call("gnnx_partition_scramble", None, n, 1, cuts, ops._ptr(nid_c), ops._stream())
    assert torch.equal(nid_c, g1.nid)
    assert g0.nnz == g1.nnz
    assert torch.equal(g1.to_vertex_order(g1.norm), g0.norm) and torch.equal(g1.to_vertex_order(g1.s), g0.s)
    H = ops.uniform_pm1(52, (n, F), device=dev_)
    G = ops.uniform_pm1(53, (n, F), device=dev_)
    bias = ops.uniform_pm1(54, (F,), device=dev_)
    H1, G1 = g1.to_new_order(H), g1.to_new_order(G)
    assert torch.equal(g1.to_vertex_order(H1), H)
    for use_plan in (False, True):
        if use_plan:
            g0.make_plans(64, F)
            g1.make_plans(64, F)
        a0 = ops.aggregate_fwd(g0, H, bias, use_plan=use_plan)
        a1 = ops.aggregate_fwd(g1, H1, bias, use_plan=use_plan)
        assert torch.equal(g1.to_vertex_order(a1), a0), f"forward, plan={use_plan}"
        b0 = ops.aggregate_bwd(g0, G, use_plan=use_plan)
        b1 = ops.aggregate_bwd(g1, G1, use_plan=use_plan)
        assert torch.equal(g1.to_vertex_order(b1), b0), f"backward, plan={use_plan}"
    if n <= 60_000:   # and the oracle on the original labels
        rp, ci = oracle.coo_to_csr(host(src), host(dst), n)
        ref = oracle.aggregate_fwd(rp, ci, host(H), host(g0.norm), host(bias))
        assert same(host(g1.to_vertex_order(ops.aggregate_fwd(g1, H1, bias, use_plan=False))), ref)
    # a whole layer step (transform, aggregate, both gradients): rows are independent in the dense products
    W = ops.uniform_pm1(55, (F, F), scale=F ** -0.5, device=dev_)
    o0 = ops.aggregate_fwd(g0, ops.linear_fwd(H, W), bias)
    o1 = ops.aggregate_fwd(g1, ops.linear_fwd(H1, W), bias)
    assert torch.equal(g1.to_vertex_order(o1), o0)
    dX0 = ops.gemm(ops.aggregate_bwd(g0, G), W)
    dX1 = ops.gemm(ops.aggregate_bwd(g1, G1), W)
    assert torch.equal(g1.to_vertex_order(dX1), dX0)
    dW0 = ops.gemm(ops.aggregate_bwd(g0, G), H, transA=True)
    dW1 = ops.gemm(ops.aggregate_bwd(g1, G1), H1, transA=True)      # sums over the rows in another order: rounding only
    assert float((dW0 - dW1).abs().max()) <= 1e-5 * max(1.0, float(dW0.abs().max()))


def test_native_rccl_comm_single_rank(env):
    """gnnx_comm_* / gnnx_halo_exchange_f32 / gnnx_allreduce_sum_f32 on a one-rank communicator (all this box has):
    the self-exchange must copy the packed rows into the halo tail and the all-reduce must be the identity."""
    ops, capi, torch = env["ops"], env["capi"], env["torch"]
    shard = importlib.import_module("gnncpp_amd.shard")
    comm = shard.NativeComm(capi, ops, None, 0, 1, env["dev"])
    send = ops.uniform_pm1(301, (1000, 64))
    recv = torch.zeros((1000, 64), dtype=torch.float32, device=env["dev"])
    comm.halo_exchange(send, [1000], recv, [1000], 64)
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    w = ops.uniform_pm1(302, (256, 256))
    w0 = w.clone()
    comm.allreduce(w)
    torch.cuda.synchronize()
    assert torch.equal(w, w0)
    comm.halo_exchange(send[:0], [0], recv[:0], [0], 64)  # empty exchange is legal


@pytest.mark.parametrize("n,e,F,L,abc,seed", [(1_000_000, 10_000_000, 128, 2, (0.57, 0.19, 0.19), 1),     # BASELINE configs[2]
                                              (2_400_000, 62_000_000, 100, 3, (0.45, 0.22, 0.22), 3)])    # BASELINE configs[4]
def test_multi_layer_gcn_at_baseline_sizes_vs_oracle(env, n, e, F, L, abc, seed):
    """BASELINE configs[2] (2-layer, 1M/10M, F=128) and configs[4] (3-layer, products-shaped 2.4M/62M, F=100): the hot
    path of every layer, forward and backward, against the CPU oracle on the WHOLE graph (OpenMP; seconds on the GPU
    box's host cores).  Every aggregation -- the PLANNED kernels bench.py runs (hub rows > 1024 non-zeros on the sequential hub
    kernel) -- is compared BIT-EXACTLY when fed identical inputs; the GEMMs within the (condition-aware) tolerance; the unplanned
    kernels give the same bits."""
    ops, torch = env["ops"], env["torch"]
    srcd, dstd = ops.rmat_edges(seed, n, e, *abc)
    src, dst = host(srcd), host(dstd)
    g = ops.CsrGraph.from_coo(srcd, dstd, n)
    del srcd, dstd
    rp, ci = oracle.coo_to_csr(src, dst, n)
    assert np.array_equal(host(g.rowptr), rp.astype(np.int32)) and np.array_equal(host(g.colidx), ci)
    rT, cT = oracle.csr_transpose(rp, ci, n)
    del src, dst
    s, norm = oracle.degree_norm(rp, ci, n)
    # the degree block is the reference's by default (s from the host libm's powf table, functional.h:253): s and norm bit-exact
    assert same(host(g.s), s), "s = deg^-1/2 is not the host libm's powf"
    assert same(host(g.norm), norm), "norm"
    norm_g = host(g.norm)

    def a64(x):
        return np.abs(x).astype(np.float64)

    act = ops.uniform_pm1(401, (n, F))
    Ws = [synth.uniform_pm1(410 + l, (F, F), scale=F ** -0.5) for l in range(L)]
    bs = [synth.uniform_pm1(420 + l, (F,), scale=0.1) for l in range(L)]
    inputs, Hs = [], []
    g.make_plans(chunk=1024, max_feat=F)
    assert g.plan.n_split_rows == int((np.diff(rp) > 1024).sum()) and g.plan_t.n_split_rows == int((np.diff(rT) > 1024).sum())
    for l in range(L):  # ---- forward, the bench's planned kernels
        inputs.append(act)
        H = ops.linear_fwd(act, dev(env, Ws[l]))
        O = ops.aggregate_fwd(g, H, dev(env, bs[l]))
        xin = host(act)
        # K = F <= 256: the plain 1e-5 * max(1, |ref|) bar for the transform of O(1) inputs (layer 0).  From layer 1 on the inputs are
        # aggregated rows -- a hub row's activations reach 1e3-1e4 -- so an element of H is a cancelled sum of terms thousands of times
        # its size, where no two correct summation orders agree to 1e-5 of the RESULT: there the condition-aware form of the bound.
        assert_close(host(H), oracle.linear_fwd(xin, Ws[l]), f"layer {l} transform", absum=None if l == 0 else a64(xin) @ a64(Ws[l]).T)
        assert same(host(O), oracle.aggregate_fwd(rp, ci, host(H), norm_g, bs[l])), f"layer {l} aggregation not bit-exact"
        Hs.append(H)
        act = O
    Ou = ops.aggregate_fwd(g, Hs[-1], dev(env, bs[-1]), use_plan=False)   # ---- without the plan: the same bits
    assert torch.equal(Ou, act)
    del Ou, Hs
    G = ops.uniform_pm1(430, (n, F))
    for l in reversed(range(L)):  # ---- backward, planned kernels: dX of layer l is G of layer l-1
        Gh = host(G)
        dH = ops.aggregate_bwd(g, G)
        assert same(host(dH), oracle.aggregate_bwd(rT, cT, Gh, norm_g)), f"layer {l} aggregation backward not bit-exact"
        dX, dW = ops.linear_bwd(dH, inputs[l], dev(env, Ws[l]))
        dHh = host(dH)
        rdX, _ = oracle.linear_bwd(dHh, host(inputs[l]), Ws[l], need_dw=False)
        assert_close(host(dX), rdX, f"layer {l} dX", absum=a64(dHh) @ a64(Ws[l]))
        G = dX


def test_headline_config_whole_graph_vs_oracle(env):
    """BASELINE configs[3] -- RMAT 10 M nodes / 100 M edges, 256 features -- exactly as bench.py runs it: the bench's seed (2), the
    bench's row order (vertex v stored at row (v * 2654435761) mod n), the bench's plan (hub rows > 1024 non-zeros on the sequential
    hub kernel), forward and backward aggregation of the WHOLE graph against the CPU oracle (OpenMP on the box's host cores; the
    oracle runs once per direction, in vertex order; the GPU result is moved back to vertex order to be compared):
      * CSR of A and of A^T (as-generated order) equal the oracle's; the relabelled graph has the same norm per vertex;
      * fed the same norm, every one of the 10 M rows of both PLANNED aggregations is BIT-EXACT -- and the unplanned kernels and the
        as-generated order give those bits too;
      * the degree block is the reference's BY DEFAULT -- s looked up in a table of the host libm's powf (functional.h:253; glibc's
        powf is 1 ulp from the correctly rounded value for a few degrees >= 1058) -- so s, norm and, end to end with the ORACLE'S OWN
        norm, both aggregations are BIT-EXACT: nothing tolerance-level is left on the aggregation path;
      * a caller-supplied table (CsrGraph.norm_from_pow_table) gives the same bits."""
    ops, torch = env["ops"], env["torch"]
    n, e, F, abc, seed = 10_000_000, 100_000_000, 256, (0.57, 0.19, 0.19), 2
    srcd, dstd = ops.rmat_edges(seed, n, e, *abc)
    src, dst = host(srcd), host(dstd)
    g0 = ops.CsrGraph.from_coo(srcd, dstd, n)
    g = ops.CsrGraph.from_coo(srcd, dstd, n, relabel="scramble")       # bench.py's default order for F = 256
    del srcd, dstd
    ops._ws_cache.clear()
    torch.cuda.empty_cache()
    rp, ci = oracle.coo_to_csr(src, dst, n)
    assert np.array_equal(host(g0.rowptr), rp.astype(np.int32)) and np.array_equal(host(g0.colidx), ci)
    rT, cT = oracle.csr_transpose(rp, ci, n)
    assert np.array_equal(host(g0.rowptr_t), rT.astype(np.int32)) and np.array_equal(host(g0.colidx_t), cT)
    del src, dst
    assert g.nnz == g0.nnz and torch.equal(g.to_vertex_order(g.norm), g0.norm) and torch.equal(g.to_vertex_order(g.s), g0.s)
    s, norm = oracle.degree_norm(rp, ci, n)
    assert same(host(g0.s), s), "default s is not the host libm's powf(deg, -0.5)"
    assert same(host(g0.norm), norm), "default norm is not the oracle's"
    norm_g = host(g0.norm)
    bias = synth.uniform_pm1(421, (F,), scale=0.1)
    g.make_plans(chunk=1024, max_feat=F)                                 # bench.py's default plan
    g0.make_plans(chunk=1024, max_feat=F)
    assert g.plan.n_split_rows == int((np.diff(rp) > 1024).sum()) and g.plan_t.n_split_rows == int((np.diff(rT) > 1024).sum())

    # ---- forward
    H = ops.uniform_pm1(401, (n, F))                                    # rows in vertex order
    Hh = host(H)
    H1 = g.to_new_order(H)
    O1 = ops.aggregate_fwd(g, H1, dev(env, bias))                       # bench order, bench plan
    O_v = g.to_vertex_order(O1)
    ref = oracle.aggregate_fwd(rp, ci, Hh, norm_g, bias)
    assert same(host(O_v), ref), "planned forward aggregation of the headline graph (bench order) is not bit-exact"
    del ref
    assert torch.equal(ops.aggregate_fwd(g, H1, dev(env, bias), use_plan=False), O1), "unplanned == planned"
    del O1, H1
    assert torch.equal(ops.aggregate_fwd(g0, H, dev(env, bias)), O_v), "as-generated order == bench order, per vertex"
    ref2 = oracle.aggregate_fwd(rp, ci, Hh, norm, bias)                 # the oracle's own norm: end to end
    got = host(O_v)
    assert same(got, ref2), "forward vs oracle (own norm), end to end"
    # a caller-supplied table of the host libm's powf (the former opt-in): the same bits as the default
    pow_table = dev(env, oracle.powf_table(int(np.diff(rp).max()) + 3))
    s_l, norm_l = g.norm_from_pow_table(pow_table)
    assert same(host(g.to_vertex_order(s_l)), s) and same(host(g.to_vertex_order(norm_l)), norm), "libm-exact s / norm"
    O_l = ops.spmm(g.rowptr, g.colidx, g.to_new_order(H), rowscale=norm_l, bias=dev(env, bias), plan=g.plan)
    assert same(host(g.to_vertex_order(O_l)), ref2), "libm-exact mode: forward aggregation bit-exact end to end"
    del got, ref2, O_v, O_l, H, Hh
    torch.cuda.empty_cache()
    # ---- backward
    G = ops.uniform_pm1(430, (n, F))
    Gh = host(G)
    G1 = g.to_new_order(G)
    D1 = ops.aggregate_bwd(g, G1)
    D_v = g.to_vertex_order(D1)
    ref = oracle.aggregate_bwd(rT, cT, Gh, norm_g)
    assert same(host(D_v), ref), "planned backward aggregation of the headline graph (bench order) is not bit-exact"
    del ref
    assert torch.equal(ops.aggregate_bwd(g, G1, use_plan=False), D1), "unplanned == planned"
    del D1, G1
    assert torch.equal(ops.aggregate_bwd(g0, G), D_v), "as-generated order == bench order, per vertex"
    ref2 = oracle.aggregate_bwd(rT, cT, Gh, norm)
    got = host(D_v)
    assert same(got, ref2), "backward vs oracle (own norm), end to end"
    vals_l = ops.gather_rows(norm_l.reshape(-1, 1), g.colidx_t).reshape(-1)   # libm-exact norm as the backward's per-entry scale
    D_l = ops.spmm(g.rowptr_t, g.colidx_t, g.to_new_order(G), vals=vals_l, plan=g.plan_t)
    assert same(host(g.to_vertex_order(D_l)), ref2), "libm-exact mode: backward aggregation bit-exact end to end"
    del got, ref2, D_v, D_l, G, Gh, g, g0
    ops._ws_cache.clear()
    torch.cuda.empty_cache()


# ------------------------------------------------------------------ next row: loss + optimiser + multi-layer step
def test_softmax_cross_entropy_vs_oracle_and_float64(env):
    ops = env["ops"]
    n, c = 20000, 47
    X = synth.uniform_pm1(501, (n, c)) * 4.0
    t = ((7 * np.arange(n) + 3) % c).astype(np.int32)
    loss, d = ops.softmax_ce(dev(env, X), dev(env, t))
    ref = oracle.cross_entropy(X, t)
    assert abs(float(host(loss)[0]) - ref) <= 1e-5 * max(1.0, abs(ref))
    x = X.astype(np.float64)
    p = np.exp(x) / np.exp(x).sum(1, keepdims=True)
    p[np.arange(n), t] -= 1.0
    assert np.abs(host(d) - p / n).max() <= 1e-5 / n * 10
    with pytest.raises(env["capi"].GnnxError):
        ops.softmax_ce(dev(env, X), dev(env, np.full(n, c, dtype=np.int32)))


@pytest.mark.parametrize("n,c", [(20001, 256), (777, 48), (5000, 1000), (3001, 1028), (100, 7), (9, 4)])
def test_softmax_cross_entropy_shapes_and_fused_bias_gradient(env, n, c):
    """Both kernels of gnnx_softmax_ce_colsum_f32 (the one-16-byte-load-per-lane form for class counts that are multiples of 4 up
    to 1024 -- the bench's 256 is one 1-KiB wave-instruction per row -- and the generic walk) against float64, an odd number of
    rows (the two-rows-in-flight tail), a padded row stride, and the column sums of dlogits written by the same kernel against a
    float64 column sum and against gnnx_colsum_f32 over the stored gradient."""
    ops, torch = env["ops"], env["torch"]
    X = synth.uniform_pm1(900 + c, (n, c)) * 3.0
    t = ((7 * np.arange(n) + 3) % c).astype(np.int32)
    x = X.astype(np.float64)
    p = np.exp(x) / np.exp(x).sum(1, keepdims=True)
    loss_ref = float(-np.log(p[np.arange(n), t]).mean())
    p[np.arange(n), t] -= 1.0
    p /= n
    for pad in (0, 4):
        Xd = torch.zeros((n, c + pad), dtype=torch.float32, device="cuda")
        Xd[:, :c] = dev(env, X)
        db = torch.full((c,), 7.0, dtype=torch.float32, device="cuda")
        loss, d = ops.softmax_ce(Xd[:, :c], dev(env, t), colsum_out=db)
        assert abs(float(host(loss)[0]) - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
        assert abs(float(host(loss)[0]) - oracle.cross_entropy(X, t)) <= 1e-5 * max(1.0, abs(loss_ref))
        assert np.abs(host(d) - p).max() <= 1e-5 / n * 10
        ref_db = p.sum(0)
        assert np.abs(host(db) - ref_db).max() <= 1e-5 * max(np.abs(p).sum(0).max(), 1e-12)
        sep = ops.colsum(d)
        assert np.abs(host(db) - host(sep)).max() <= 1e-5 * max(np.abs(p).sum(0).max(), 1e-12)
        loss2, d2 = ops.softmax_ce(Xd[:, :c], dev(env, t))   # without the column sums: the same loss and gradient bits
        assert torch.equal(d, d2) and torch.equal(loss, loss2)
    with pytest.raises(env["capi"].GnnxError):
        ops.softmax_ce(dev(env, X), dev(env, np.full(n, c, dtype=np.int32)))


def test_two_layer_training_step_vs_float64(env):
    """Whole step of a 2-layer GCN (transform, aggregate, ReLU, transform, aggregate, softmax-CE, backward, SGD) against a
    float64 numpy evaluation of the same network; then the loss goes down under SGD."""
    ops, torch = env["ops"], env["torch"]
    n, e, dims = 3000, 24000, [32, 16, 7]
    src, dst = synth.uniform_edges(77, n, e)  # no hubs: logits stay O(1), the reference's max-free softmax does not overflow
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), n)
    rp, ci = oracle.coo_to_csr(src, dst, n)
    _, norm = oracle.degree_norm(rp, ci, n)
    X = synth.uniform_pm1(601, (n, dims[0]))
    t = ((7 * np.arange(n) + 3) % dims[-1]).astype(np.int32)
    net = ops.GcnStack(g, dims, seed=610)
    for l in range(2):
        net.b[l].copy_(dev(env, synth.uniform_pm1(620 + l, (dims[l + 1],), scale=0.2)))
    W = [host(w).astype(np.float64) for w in net.W]
    b = [host(v).astype(np.float64) for v in net.b]
    logits = net.forward(dev(env, X))
    loss, dlog = ops.softmax_ce(logits, dev(env, t))
    net.backward(dlog)
    # ---- float64 reference
    import scipy.sparse as sp
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    S = sp.diags(norm.astype(np.float64)) @ A
    x0 = X.astype(np.float64)
    z1 = S @ (x0 @ W[0].T) + b[0]
    y1 = np.maximum(z1, 0)
    z2 = S @ (y1 @ W[1].T) + b[1]
    ex = np.exp(z2)
    p = ex / ex.sum(1, keepdims=True)
    loss_ref = float(-np.log(p[np.arange(n), t]).mean())
    dz2 = p.copy()
    dz2[np.arange(n), t] -= 1
    dz2 /= n
    dh2 = S.T @ dz2
    dW1, db1 = dh2.T @ y1, dz2.sum(0)
    dz1 = (dh2 @ W[1]) * (z1 > 0)
    dh1 = S.T @ dz1
    dW0, db0 = dh1.T @ x0, dz1.sum(0)
    assert abs(float(host(loss)[0]) - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref))
    for got, ref, nm in ((net.dW[1], dW1, "dW1"), (net.db[1], db1, "db1"), (net.dW[0], dW0, "dW0"), (net.db[0], db0, "db0")):
        err = np.abs(host(got) - ref).max()
        assert err <= 2e-5 * max(np.abs(ref).max(), 1e-3), f"{nm}: {err:.3e} vs scale {np.abs(ref).max():.3e}"
    # ---- SGD: the loss decreases
    losses = []
    # ---- the training-step form bench.py --train-layers runs: bias gradient of the last layer from the loss kernel, no input
    # gradient: the same parameter gradients
    keep = [host(v).copy() for v in net.dW + net.db]
    logits = net.forward(dev(env, X))
    loss, dlog = ops.softmax_ce(logits, dev(env, t), colsum_out=net.db[-1])
    assert net.backward(dlog, input_grad=False, have_last_bias_grad=True) is None
    for a, bref, nm in zip(net.dW + net.db, keep, ("dW0", "dW1", "db0", "db1")):
        tol = 0.0 if nm != "db1" else 1e-6 * float(np.abs(bref).max())
        assert np.abs(host(a) - bref).max() <= tol, nm
    for _ in range(20):
        logits = net.forward(dev(env, X))
        loss, dlog = ops.softmax_ce(logits, dev(env, t), colsum_out=net.db[-1])
        losses.append(float(host(loss)[0]))
        net.backward(dlog, input_grad=False, have_last_bias_grad=True)
        net.step(lr=0.05)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_csr_validate_rejects_foreign_csr_with_bad_columns(env):
    """The SpMM trusts its CSR; gnnx_csr_validate is the one-off check for a CSR that did not come from the builder."""
    ops, capi, torch = env["ops"], env["capi"], env["torch"]
    src, dst, rp, ci, g = make_graph(env, 500, 4000, seed=31)
    capi.call("gnnx_csr_validate", ops._ptr(g.rowptr), ops._ptr(g.colidx), g.n, g.n, ops._stream())
    bad = g.colidx.clone()
    bad[17] = g.n  # one past the last column
    with pytest.raises(capi.GnnxError) as ei:
        capi.call("gnnx_csr_validate", ops._ptr(g.rowptr), ops._ptr(bad), g.n, g.n, ops._stream())
    assert ei.value.status == -3
    rp2 = g.rowptr.clone()
    rp2[5] = rp2[6] + 1  # not monotone
    with pytest.raises(capi.GnnxError):
        capi.call("gnnx_csr_validate", ops._ptr(rp2), ops._ptr(g.colidx), g.n, g.n, ops._stream())


@pytest.mark.parametrize("self_term", [False, True])
def test_mode_sym_textbook_normalisation(env, self_term):
    """The textbook D^-1/2 (A [+ I]) D^-1/2 . H of the north_star (Mode SYM), forward and backward, vs float64."""
    import scipy.sparse as sp
    ops = env["ops"]
    n, F = 6000, 96
    src, dst, rp, ci, g = make_graph(env, n, 50000, seed=23)
    s = host(g.s).astype(np.float64)
    A = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    M = sp.diags(s) @ (A + (sp.identity(n) if self_term else 0 * sp.identity(n))) @ sp.diags(s)
    H = synth.uniform_pm1(701, (n, F))
    G = synth.uniform_pm1(702, (n, F))
    bias = synth.uniform_pm1(703, (F,))
    out = host(ops.aggregate_fwd_sym(g, dev(env, H), dev(env, bias), self_term=self_term))
    ref = M @ H.astype(np.float64) + bias
    absum = abs(M) @ np.abs(H).astype(np.float64) + np.abs(bias)
    assert_close(out, ref.astype(np.float32), "SYM forward", absum=absum)
    dH = host(ops.aggregate_bwd_sym(g, dev(env, G), self_term=self_term))
    refb = M.T @ G.astype(np.float64)
    assert_close(dH, refb.astype(np.float32), "SYM backward", absum=abs(M).T @ np.abs(G).astype(np.float64))
    # adjointness of the pair
    a = float((out.astype(np.float64) - bias) .ravel() @ G.astype(np.float64).ravel())
    b = float(H.astype(np.float64).ravel() @ dH.astype(np.float64).ravel())
    assert abs(a - b) <= 1e-5 * max(abs(a), abs(b), 1.0)


# ---- the API's elementwise operators with broadcast (gnnx_binary_bcast_f32) and dense row sums (gnnx_rowsum_f32)
@pytest.mark.parametrize("op,fn", [("add", np.add), ("sub", np.subtract), ("mul", np.multiply), ("div", np.divide)])
def test_binary_broadcast_is_ieee_exact(env, op, fn):
    ops = env["ops"]
    n, f = 1000, 37
    A = synth.uniform_pm1(900, (n, f))
    col = synth.uniform_pm1(901, (n, 1)) + np.float32(2)
    row = synth.uniform_pm1(902, (f,)) + np.float32(2)
    one = np.array([1.5], dtype=np.float32)
    for lhs, rhs in ((A, A[::-1].copy()), (A, col), (A, row), (col, A), (row, A), (A, one), (col, row.reshape(1, f))):
        got = host(ops.binary(op, dev(env, lhs), dev(env, rhs)))
        l2 = lhs.reshape(1, -1) if lhs.ndim == 1 else lhs
        r2 = rhs.reshape(1, -1) if rhs.ndim == 1 else rhs
        assert same(got, fn(l2, r2).astype(np.float32)), (op, lhs.shape, rhs.shape)


def test_rowsum_walks_up_like_functional_sum(env):
    ops = env["ops"]
    X = synth.uniform_pm1(903, (513, 64))
    exp = np.zeros(513, dtype=np.float32)
    for j in range(64):
        exp = exp + X[:, j]          # ascending, one rounding per add
    assert same(host(ops.rowsum(dev(env, X))), exp)
    Xw = synth.uniform_pm1(904, (77, 300))  # wider than a wavefront: lane slices, then left to right
    got = host(ops.rowsum(dev(env, Xw)))
    assert np.abs(got.astype(np.float64) - Xw.astype(np.float64).sum(1)).max() <= 1e-5 * np.abs(Xw).sum(1).max()


# ---- fusion: BatchNorm / ReLU folded into the aggregation (gnnx_spmm_csr_fused_f32) ----------------
@pytest.mark.parametrize("n,e,F,chunk", [(3000, 40000, 256, 0), (3000, 40000, 256, 64), (2000, 30000, 100, 0), (1500, 20000, 16, 0),
                                         (1500, 20000, 7, 0), (4000, 60000, 128, 32), (900, 9000, 50, 16)])
def test_fused_prologue_epilogue_same_bits_as_separate_kernels(env, n, e, F, chunk):
    """relu(bn(H)) gathered on the fly == bn_relu_fwd into a buffer, then the plain aggregation: same ops in the same
    order, so the same bits -- every kernel variant (row / streaming, vector / scalar lanes, with and without hub chunks)."""
    ops = env["ops"]
    src, dst = synth.rmat_edges(700 + F, n, e)
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), n)
    if chunk:
        g.make_plans(chunk, F)
    H = dev(env, synth.uniform_pm1(701, (n, F)) * 1.7 + 0.2)
    gamma = dev(env, synth.uniform_pm1(702, (F,)) + 1.5)
    beta = dev(env, synth.uniform_pm1(703, (F,), scale=0.3))
    bias = dev(env, synth.uniform_pm1(704, (F,), scale=0.5))
    mean, var = ops.bn_stats(H)
    for (gm, bt, relu) in ((gamma, beta, True), (gamma, beta, False), (None, None, True), (gamma, None, True)):
        sep = ops.aggregate_fwd(g, ops.bn_relu_fwd(H, mean, var, gm, bt, relu=relu), bias)
        fus = ops.aggregate_fwd(g, H, bias, bn=(mean, var, gm, bt, 1e-5), relu_in=relu)
        assert same(host(fus), host(sep)), (F, chunk, gm is not None, bt is not None, relu)
    # ReLU only, as prologue and as epilogue
    sep = ops.aggregate_fwd(g, ops.bn_relu_fwd(H, relu=True), bias)
    assert same(host(ops.aggregate_fwd(g, H, bias, relu_in=True)), host(sep))
    plain = ops.aggregate_fwd(g, H, bias)
    assert same(host(ops.aggregate_fwd(g, H, bias, relu_out=True)), np.maximum(host(plain), 0))
    # backward of the fused forward: the ReLU mask is recomputed from H instead of read from a stored output
    dY = dev(env, synth.uniform_pm1(705, (n, F)))
    Y = ops.bn_relu_fwd(H, mean, var, gamma, beta, relu=True)
    a = ops.bn_relu_bwd(H, Y, dY, mean, var, gamma, relu=True, beta=beta)
    b = ops.bn_relu_bwd(H, None, dY, mean, var, gamma, relu=True, beta=beta)
    for x, y in zip(a, b):
        assert same(host(x), host(y))
    Yr = ops.bn_relu_fwd(H, relu=True)
    assert same(host(ops.bn_relu_bwd(H, Yr, dY, relu=True)[0]), host(ops.bn_relu_bwd(H, None, dY, relu=True)[0]))


def test_fused_prologue_rejects_backward_modes(env):
    ops, capi = env["ops"], env["capi"]
    src, dst = synth.rmat_edges(710, 100, 600)
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), 100)
    H = dev(env, synth.uniform_pm1(711, (100, 8)))
    with pytest.raises(capi.GnnxError):
        ops.spmm(g.rowptr, g.colidx, H, colscale=g.norm, relu_in=True)


# ---- weighted adjacency (edge_attr): gnnx_csr_from_coo_weighted + per-entry values in the SpMM ----
@pytest.mark.parametrize("name", ["weighted6", "weighted_rmat64"])
def test_weighted_adjacency_vs_reference_golden(env, name):
    ops = env["ops"]
    d = load_case(name)
    n = d["n"]
    src, dst, w = dev(env, d["src"]), dev(env, d["dst"]), dev(env, d["w"])
    rp, ci, va = ops.csr_from_coo_weighted(src, dst, w, n)
    orp, oci, ova = oracle.coo_to_csr_weighted(d["src"], d["dst"], d["w"], n)
    assert np.array_equal(host(rp), orp) and np.array_equal(host(ci), oci) and same(host(va), ova)
    assert same(host(ops.csr_rowsum(rp, va)), d["ref_w_deg"])                       # adj->sum(-1)
    assert same(host(ops.spmm(rp, ci, dev(env, d["ref_H"]), vals=va)), d["ref_w_mm"])  # adj->mm(x), the reference's own bits
    for mode, fill, key in ((ops.DIAG_FILL, 2.5, "w_fill"), (ops.DIAG_STRIP, 0.0, "w_strip")):
        rp, ci, va = ops.csr_from_coo_weighted(src, dst, w, n, diag_mode=mode, diag_value=fill, flags=ops.CSR_DROP_TRUNCATED_ZERO)
        rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(host(rp)))
        assert np.array_equal(rows, d["ref_" + key + "_ei"][0]) and np.array_equal(host(ci), d["ref_" + key + "_ei"][1]), key
        assert same(host(va), d["ref_" + key + "_ea"]), key


@pytest.mark.parametrize("n,e,F", [(5000, 80000, 64), (20000, 400000, 256), (3000, 20000, 7)])
def test_weighted_adjacency_vs_oracle_seeded(env, n, e, F):
    """Bigger seeded graphs with many duplicate edges: the winner of every duplicate run must be the LAST edge of the list."""
    ops = env["ops"]
    src, dst = synth.rmat_edges(800 + F, n, e)
    w = synth.uniform_pm1(801, (e,), scale=3.0)
    X = synth.uniform_pm1(802, (n, F))
    for mode, fill, flags in ((ops.DIAG_KEEP, 0.0, 0), (ops.DIAG_FILL, -1.5, ops.CSR_DROP_TRUNCATED_ZERO), (ops.DIAG_STRIP, 0.0, 0)):
        rp, ci, va = ops.csr_from_coo_weighted(dev(env, src), dev(env, dst), dev(env, w), n, diag_mode=mode, diag_value=fill, flags=flags)
        orp, oci, ova = oracle.coo_to_csr_weighted(src, dst, w, n, diag_mode=mode, diag_value=fill,
                                                   drop_truncated_zero=bool(flags & ops.CSR_DROP_TRUNCATED_ZERO))
        assert np.array_equal(host(rp), orp) and np.array_equal(host(ci), oci) and same(host(va), ova), mode
        assert same(host(ops.spmm(rp, ci, dev(env, X), vals=va)), oracle.spmm_vals(orp, oci, ova, X)), mode
    with pytest.raises(env["capi"].GnnxError):
        bad = src.copy()
        bad[3] = n
        ops.csr_from_coo_weighted(dev(env, bad), dev(env, dst), dev(env, w), n)


# ---- opt-in bf16 feature storage (gnnx_f32_to_bf16 + gnnx_spmm_csr_bf16_f32) -- not the parity path
@pytest.mark.parametrize("n,e,F,chunk", [(4000, 60000, 256, 0), (4000, 60000, 256, 64), (3000, 30000, 100, 0), (2000, 20000, 7, 0),
                                         (8000, 200000, 7, 64), (8000, 200000, 70, 64), (8000, 200000, 100, 64)])
def test_bf16_feature_storage(env, n, e, F, chunk):
    """The last three cases: a PLAN WITH HUB ROWS on bf16 rows -- widths that are not a multiple of 4 (7, 70: the scalar-lane
    kernels; their hub rows stay with the row / streaming kernel, LDS-DMA has no 2-byte-per-lane layout) and 100 (8-byte pieces: the
    hub kernel's bf16 ring) -- planned == unplanned, bit for bit."""
    ops, torch = env["ops"], env["torch"]
    src, dst = synth.rmat_edges(850 + F, n, e)
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), n)
    if chunk:
        g.make_plans(chunk, F)
        if n >= 8000:
            assert g.plan.n_split_rows > 0 and g.plan_t.n_split_rows > 0, "the case is meant to have hub rows"
    X = synth.uniform_pm1(851, (n, F))
    X[0, :3] = [np.nan, np.inf, -np.inf]
    Xd = dev(env, X)
    Xb = ops.to_bf16(Xd)
    # the conversion is round-to-nearest-even and keeps NaN / Inf: same bits as torch's own cast
    assert torch.equal(Xb.view(torch.int16), Xd.to(torch.bfloat16).view(torch.int16))
    X[0, :3] = 0
    Xd = dev(env, X)
    Xb = ops.to_bf16(Xd)
    bias = dev(env, synth.uniform_pm1(852, (F,), scale=0.5))
    # (1) widening is exact: on features that ARE bf16 numbers the bf16 path is the f32 path, bit for bit (fwd and bwd modes)
    Xr = Xb.float().contiguous()
    assert same(host(ops.aggregate_fwd(g, Xb, bias)), host(ops.aggregate_fwd(g, Xr, bias)))
    assert same(host(ops.aggregate_bwd(g, Xb)), host(ops.aggregate_bwd(g, Xr)))
    if chunk:   # planned == unplanned on the bf16 rows themselves
        assert same(host(ops.aggregate_fwd(g, Xb, bias)), host(ops.aggregate_fwd(g, Xb, bias, use_plan=False)))
        assert same(host(ops.aggregate_bwd(g, Xb)), host(ops.aggregate_bwd(g, Xb, use_plan=False)))
    # (2) against the f32 features: one bf16 rounding (at most 2^-8 relative) per gathered element, nothing else
    full = host(ops.aggregate_fwd(g, Xd, bias)).astype(np.float64)
    got = host(ops.aggregate_fwd(g, Xb, bias)).astype(np.float64)
    rp, ci = oracle.coo_to_csr(src, dst, n)
    absum = oracle.aggregate_fwd(rp, ci, np.abs(X), host(g.norm), None).astype(np.float64)  # norm_i * sum_j |x_j|
    assert np.all(np.abs(got - full) <= 2.0 ** -8 * absum * 1.001 + 1e-6)


# ---- the resident streaming GEMM kernel (tall products with whole tiles; the ragged remainder goes to the generic kernel) --
@pytest.mark.parametrize("M,N,K", [(300077, 256, 256), (2 * 131072 + 1, 128, 128), (262144 + 255, 256, 64), (600000, 512, 32),
                                   (300077, 100, 128), (262144 + 255, 200, 64), (100000, 68, 192), (100003, 380, 128)])
def test_gemm_streaming_kernel_tall_products(env, M, N, K):
    """X.W^T and dH.W on hundreds of thousands of rows take gemm_stream_kernel for the whole 256-row (128-row) tiles and
    gemm_kernel for the rest: every row -- first tile, tile seams, the ragged tail -- against float64, and bit-identical to
    the same rows computed as a SHORT product (which takes the generic kernel only): both are the same fmaf chain over k."""
    ops, torch = env["ops"], env["torch"]
    X = ops.uniform_pm1(950, (M, K), device=env["dev"])
    W = ops.uniform_pm1(951, (N, K), scale=K ** -0.5, device=env["dev"])     # [N, K]  (X.W^T: NT)
    Wn = W.t().contiguous()                                                  # [K, N]  (dH.W form: NN)
    H = ops.gemm(X, W, transB=True)
    H2 = ops.gemm(X, Wn)
    rows = np.unique(np.concatenate([np.arange(0, 600), np.arange(M - 700, M), np.random.default_rng(1).integers(0, M, 3000),
                                     np.arange(255 * 1000 - 5, 255 * 1000 + 300)]))
    rows = rows[(rows >= 0) & (rows < M)]
    ridx = torch.from_numpy(rows).to(env["dev"])
    ref = X[ridx].double() @ W.double().t()
    for got in (H, H2):
        err = (got[ridx].double() - ref).abs().max().item()
        assert err <= 1e-5 * max(1.0, ref.abs().max().item()), err
    # same bits as the generic kernel on a short slice (rows [s, s+1000) straddle tile seams)
    for s in (0, 255 * 1000, M - 1000):
        short = ops.gemm(X[s:s + 1000].contiguous(), W, transB=True)
        assert torch.equal(short, H[s:s + 1000])
        short2 = ops.gemm(X[s:s + 1000].contiguous(), Wn)
        assert torch.equal(short2, H2[s:s + 1000])


@pytest.mark.parametrize("N", [100, 68, 132, 252])
def test_gemm_guarded_last_column_tile(env, N):
    """Widths off the 128 grid (the products-shaped F = 100) on the LDS-DMA kernel: the last column tile is guarded (B lanes past
    column N do not load, C lanes past it do not store).  C is a column slice of a wider buffer filled with a sentinel: nothing
    outside the N columns may change; every variant (plain NN / NT, fused mask + column sums, fused BN statistics) against the
    same call on a short slice / the separate passes."""
    ops, torch = env["ops"], env["torch"]
    dev_ = env["dev"]
    M, K = 70000 + 13, 128
    A = ops.uniform_pm1(1200 + N, (M, K), device=dev_)
    W = ops.uniform_pm1(1201 + N, (N, K), scale=K ** -0.5, device=dev_)
    Wn = W.t().contiguous()
    for transB, B in ((True, W), (False, Wn)):
        Cw = torch.full((M, N + 60), 123.0, dtype=torch.float32, device=dev_)
        Cv = Cw[:, 28:28 + N]
        ops.gemm(A, B, transB=transB, out=Cv)
        assert float((Cw[:, :28] - 123.0).abs().max()) == 0.0 and float((Cw[:, 28 + N:] - 123.0).abs().max()) == 0.0
        ref = A[:3000].double() @ W.double().t()
        assert (Cv[:3000].double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
        for s0 in (0, 255 * 100, M - 1000):
            short = ops.gemm(A[s0:s0 + 1000].contiguous(), B, transB=transB)
            assert torch.equal(short, Cv[s0:s0 + 1000])
    # fused: mask + column sums (dH . W of a stacked layer) vs the separate passes
    Y = torch.relu(ops.uniform_pm1(1202 + N, (M, N), device=dev_))
    db = torch.empty(N, dtype=torch.float32, device=dev_)
    Gf, _ = ops.gemm_relu_colsum(A, Wn, Y, colsum_out=db)
    Gs = ops.gemm(A, Wn)
    Gs, _, _ = ops.bn_relu_bwd(Y, Y, Gs, relu=True)
    assert torch.equal(Gf, Gs)
    ref_db = Gs.double().sum(0)
    assert (db.double() - ref_db).abs().max().item() <= 1e-5 * max(1.0, Gs.double().abs().sum(0).max().item())
    # fused BN statistics (opt-in) vs float64 statistics of the same H
    H, mean, var = ops.linear_fwd_bn_stats(A, W)
    assert torch.equal(H, ops.gemm(A, W, transB=True))
    assert (mean.double() - H.double().mean(0)).abs().max().item() <= 1e-5
    assert (var.double() - H.double().var(0, unbiased=False)).abs().max().item() <= 1e-5 * max(1.0, float(H.double().var(0).max()))


@pytest.mark.parametrize("M,N,K", [(300077, 256, 256), (70013, 100, 128), (5000, 128, 64), (2048, 64, 64), (1000, 50, 30)])
def test_transform_with_bf16_output_epilogue(env, M, N, K):
    """Opt-in bf16 feature storage: X . W^T stored as bf16 by the product's epilogue (gnnx_gemm_nt_bf16out_f32) equals the f32
    product followed by gnnx_f32_to_bf16 bit for bit (same fmaf chain, same round-to-nearest-even) -- whole tiles, the ragged
    tail, a guarded last column tile and (last case) the two-step fallback of a shape the kernel does not take; the output is a
    column slice of a wider bf16 buffer: nothing outside it changes."""
    ops, torch = env["ops"], env["torch"]
    X = ops.uniform_pm1(1300, (M, K), device=env["dev"])
    W = ops.uniform_pm1(1301, (N, K), scale=K ** -0.5, device=env["dev"])
    ref = ops.to_bf16(ops.gemm(X, W, transB=True))
    wide = torch.full((M, N + 24), 3.0, dtype=torch.bfloat16, device=env["dev"])
    out = ops.linear_fwd_bf16(X, W, out=wide[:, 8:8 + N])
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    assert bool((wide[:, :8] == 3.0).all()) and bool((wide[:, 8 + N:] == 3.0).all())
    if M >= 2048 and K % 64 == 0 and N % 4 == 0 and N >= 64:   # the C-ABI entry refuses other shapes loudly
        with pytest.raises(env["capi"].GnnxError):
            import ctypes as C
            capi = env["capi"]
            capi.call("gnnx_gemm_nt_bf16out_f32", 100, N, K, ops._ptr(X), K, ops._ptr(W), K, ops._ptr(out), N + 24, None, 0, ops._stream())


def test_leading_dimensions_wider_than_the_matrices(env):
    """Every hot entry point takes leading dimensions: operands that are column slices of wider buffers (ld > width) must give
    the same bits as packed copies -- streaming GEMM + its ragged tail, fused / bf16 / plain SpMM."""
    ops, torch = env["ops"], env["torch"]
    dev_ = env["dev"]
    M, K, N = 300077, 128, 256
    Xw = ops.uniform_pm1(970, (M, K + 64), device=dev_)
    X = Xw[:, 32:32 + K]                                  # lda = K + 64, 16-byte aligned start
    W = ops.uniform_pm1(971, (N, K), scale=K ** -0.5, device=dev_)
    Cw = torch.zeros((M, N + 128), dtype=torch.float32, device=dev_)
    Cv = Cw[:, 64:64 + N]
    ops.gemm(X, W, transB=True, out=Cv)
    ref = ops.gemm(X.contiguous(), W, transB=True)
    assert torch.equal(Cv, ref)
    assert float(Cw[:, :64].abs().max()) == 0.0 and float(Cw[:, 64 + N:].abs().max()) == 0.0   # nothing written outside the view
    ops.gemm(X, W.t().contiguous(), out=Cv)               # the k-major-B form
    assert torch.equal(Cv, ref)
    # aggregation reading a column slice and writing a column slice
    n, e, F = 6000, 90000, 128
    src, dst = synth.rmat_edges(972, n, e)
    g = ops.CsrGraph.from_coo(dev(env, src), dev(env, dst), n)
    g.make_plans(64, F)
    Hw = ops.uniform_pm1(973, (n, F + 32), device=dev_)
    H = Hw[:, 16:16 + F]
    Hc = H.contiguous()
    bias = ops.uniform_pm1(974, (F,), scale=0.5, device=dev_)
    Ow = torch.full((n, F + 64), 7.0, dtype=torch.float32, device=dev_)
    Ov = Ow[:, 32:32 + F]
    mean, var = ops.bn_stats(Hc)
    for kw in ({}, {"relu_out": True}, {"bn": (mean, var, None, None, 1e-5), "relu_in": True}):
        ops.aggregate_fwd(g, H, bias, out=Ov, **kw)
        assert torch.equal(Ov, ops.aggregate_fwd(g, Hc, bias, **kw)), kw
        assert float((Ow[:, :32] - 7.0).abs().max()) == 0.0 and float((Ow[:, 32 + F:] - 7.0).abs().max()) == 0.0
    Hb = torch.zeros((n, F + 32), dtype=torch.bfloat16, device=dev_)
    ops.to_bf16(H, out=Hb[:, 8:8 + F])
    assert torch.equal(ops.aggregate_fwd(g, Hb[:, 8:8 + F], bias), ops.aggregate_fwd(g, ops.to_bf16(Hc), bias))
    assert torch.equal(ops.aggregate_bwd(g, H), ops.aggregate_bwd(g, Hc))


# ---- OPT-IN split-precision GEMM (gnnx_gemm_split_bf16_f32): bf16 matrix cores, exact 3-way operand split, f32 accumulate ----
@pytest.mark.parametrize("M,N,K", [(300, 128, 64), (1000, 256, 256), (4097, 128, 128), (5000, 256, 1024), (70001, 384, 32)])
def test_split_gemm_accuracy_is_f32_level(env, M, N, K):
    """Not the parity path (different arithmetic from the reference's f32 products) -- but its error against float64 must be
    no worse than 1.5x the f32 MFMA GEMM's (measured: smaller, the bf16 MFMA adds 16 products per instruction), and it must
    pass the very same 1e-5 bar as the default path; X.W^T and the k-major-B form give the same bits."""
    ops, torch = env["ops"], env["torch"]
    X = ops.uniform_pm1(980, (M, K), device=env["dev"])
    W = ops.uniform_pm1(981, (N, K), scale=K ** -0.5, device=env["dev"])
    ref = X.double() @ W.double().t()
    f32 = ops.gemm(X, W, transB=True)
    nt = ops.gemm_split(X, W, transB=True)
    nn = ops.gemm_split(X, W.t().contiguous(), transB=False)
    assert torch.equal(nt, nn)
    e_split = (nt.double() - ref).abs()
    e_f32 = (f32.double() - ref).abs()
    assert e_split.max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    assert e_split.pow(2).mean().sqrt().item() <= 1.5 * e_f32.pow(2).mean().sqrt().item() + 1e-9
    # operands that ARE bf16 numbers need one piece only: the result then equals the exact product sum rounded per MFMA
    Xb, Wb = X.bfloat16().float(), W.bfloat16().float()
    got = ops.gemm_split(Xb, Wb, transB=True)
    refb = Xb.double() @ Wb.double().t()
    assert (got.double() - refb).abs().max().item() <= 4e-7 * max(1.0, refb.abs().max().item()) * max(1, K // 256)


def test_split_gemm_rejects_shapes_it_does_not_cover(env):
    ops, capi = env["ops"], env["capi"]
    X = ops.uniform_pm1(982, (100, 40), device=env["dev"])
    W = ops.uniform_pm1(983, (128, 40), device=env["dev"])
    with pytest.raises(capi.GnnxError):
        ops.gemm_split(X, W, transB=True)          # K % 16 != 0
    with pytest.raises(capi.GnnxError):
        ops.gemm_split(ops.uniform_pm1(984, (100, 64), device=env["dev"]), ops.uniform_pm1(985, (100, 64), device=env["dev"]), transB=True)  # N % 128


def test_layer_step_captured_in_a_hip_graph_same_bits(env):
    """The C-ABI compute calls neither allocate nor synchronise (workspaces grown by a first eager step), so one layer step -- transform,
    planned aggregation with BOTH hub kernels forked onto the side streams and joined back with events, column sums, backward
    aggregation, the two gradient products -- is capturable in a hipGraph as it stands (bench.py --hip-graph): three replays give the
    bits of the eager step."""
    ops, torch = env["ops"], env["torch"]
    n, e, F = 30000, 600000, 128
    src, dst, rp, ci, g = make_graph(env, n, e, seed=911)
    g.make_plans(chunk=64, max_feat=F, big_rows=700)   # rows above 700 on the producer / consumer kernel, the other hub rows on spmm_hub_kernel
    deg = np.diff(rp)
    assert (deg > 700).sum() >= 1 and ((deg > 64) & (deg <= 700)).sum() >= 4
    X, W = dev(env, synth.uniform_pm1(1, (n, F))), dev(env, synth.uniform_pm1(2, (F, F), scale=F ** -0.5))
    G, bias = dev(env, synth.uniform_pm1(3, (n, F))), dev(env, synth.uniform_pm1(4, (F,)))
    bufs = {k: torch.empty((n, F), dtype=torch.float32, device=env["dev"]) for k in ("H", "out", "dH", "dX")}
    dW, db = torch.empty((F, F), dtype=torch.float32, device=env["dev"]), torch.empty(F, dtype=torch.float32, device=env["dev"])

    def step():
        ops.linear_fwd(X, W, out=bufs["H"])
        ops.aggregate_fwd(g, bufs["H"], bias, out=bufs["out"])
        ops.colsum(G, out=db)
        ops.aggregate_bwd(g, G, out=bufs["dH"])
        ops.gemm(bufs["dH"], W, out=bufs["dX"])
        ops.gemm(bufs["dH"], X, transA=True, out=dW)

    step()
    torch.cuda.synchronize()
    ref = {k: v.clone() for k, v in bufs.items()}
    ref_dW, ref_db = dW.clone(), db.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()   # (the capture stream's own side streams and workspaces exist before the capture starts)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
    for _ in range(3):
        for v in bufs.values():
            v.zero_()
        dW.zero_()
        db.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for k in bufs:
            assert torch.equal(bufs[k], ref[k]), k
        assert torch.equal(dW, ref_dW) and torch.equal(db, ref_db)


@pytest.mark.parametrize("n,offset", [(10_000_000, 0.0), (10_000_000, 3.0), (300_000, 3.0)])
def test_bn_stats_from_the_transform_vs_float64(env, n, offset):
    """The full layer's default BatchNorm statistics (GCNConv::fuse_bn_stats: gnnx_gemm_bn_stats_f32, a shifted single-pass variance
    finished in double, out of the transform's epilogue) held to float64 AT THE BENCH SIZE (10 M x 256), beside the two-pass kernels
    (gnnx_bn_stats_f32): neither is the reference's own sequential sum (nn.cpp:303,312), so the bar is the exact value -- the mean
    within 1e-5 of the column's standard deviation, the variance within 1e-6 relative -- and the one-pass form must be no worse than
    twice the two-pass form's error (+ a rounding floor).  An input offset of 3 sigma is the case a naive single pass would lose."""
    ops, torch = env["ops"], env["torch"]
    F = 256
    X = ops.uniform_pm1(1, (n, F), device=env["dev"]) + offset
    W = ops.uniform_pm1(2, (F, F), scale=F ** -0.5, device=env["dev"])
    H, m1, v1 = ops.linear_fwd_bn_stats(X, W)
    m2, v2 = ops.bn_stats(H)
    assert torch.equal(H, ops.linear_fwd(X, W)), "the statistics ride along: H itself is the plain product's bits"
    del X
    m64 = torch.zeros(F, dtype=torch.float64, device=env["dev"])
    q64 = torch.zeros(F, dtype=torch.float64, device=env["dev"])
    step = 1_000_000
    for r in range(0, n, step):
        m64 += H[r:r + step].double().sum(0)
    m64 /= n
    for r in range(0, n, step):
        q64 += ((H[r:r + step].double() - m64) ** 2).sum(0)
    v64 = q64 / n
    sd = v64.sqrt()
    em1, em2 = (((m.double() - m64).abs() / sd).max().item() for m in (m1, m2))
    ev1, ev2 = (((v.double() - v64).abs() / v64).max().item() for v in (v1, v2))
    assert em1 <= 1e-5 and ev1 <= 1e-6, (em1, ev1)
    assert em2 <= 1e-5 and ev2 <= 1e-6, (em2, ev2)
    assert em1 <= 2 * em2 + 2e-7 and ev1 <= 2 * ev2 + 2e-7, (em1, em2, ev1, ev2)


def test_pow_is_the_libm_table_for_the_degree_exponent_and_the_device_otherwise(env):
    """gnnx_pow_f32 (deg->pow(-0.5) of the op-by-op degree block, functional.h:253): exponent -0.5 on a vector of non-negative
    integers is looked up in the process-wide table of the HOST libm's powf -- the same bits as the fused degree block's s -- and
    repeated calls reuse the table; any other exponent or argument vector is evaluated on the device (tolerance-level), without the
    table path's scan and synchronisation."""
    import ctypes as C
    ops, torch, capi = env["ops"], env["torch"], env["capi"]
    src, dst, rp, ci, g = make_graph(env, 20000, 300000, seed=9)
    deg1 = (g.rowptr[1:] - g.rowptr[:-1] + 1).to(torch.float32)
    out = torch.empty_like(deg1)
    for _ in range(2):
        capi.call("gnnx_pow_f32", ops._ptr(deg1), deg1.numel(), C.c_float(-0.5), ops._ptr(out), ops._stream())
        assert torch.equal(out, g.s), "pow(deg + 1, -0.5) is not the degree block's s"
    x = torch.tensor([0.0, -0.0, 1.0, 4.0, 2.5, 1e6], dtype=torch.float32, device=env["dev"])
    y = torch.empty_like(x)
    capi.call("gnnx_pow_f32", ops._ptr(x), x.numel(), C.c_float(-0.5), ops._ptr(y), ops._stream())   # 2.5 is no integer: device path
    ref = np.power(host(x).astype(np.float64), -0.5)
    got = host(y).astype(np.float64)
    assert np.isinf(got[0]) and np.isinf(got[1]) and got[0] > 0 and got[1] > 0   # pow(+-0, -0.5) = +inf (C99: only ODD integer exponents keep the sign)
    assert np.allclose(got[2:], ref[2:], rtol=2e-7, atol=0)
    z = torch.arange(0, 1000, dtype=torch.float32, device=env["dev"])
    w = torch.empty_like(z)
    capi.call("gnnx_pow_f32", ops._ptr(z), z.numel(), C.c_float(2.0), ops._ptr(w), ops._stream())   # another exponent: device pow
    assert np.allclose(host(w), host(z).astype(np.float64) ** 2, rtol=2e-7)


@pytest.mark.parametrize("n,F,world", [(50_000, 256, 8), (30_000, 128, 4), (20_000, 64, 8), (5_000, 16, 3), (1_000, 256, 2)])
def test_pack_from_the_producers_side_equals_the_gather_pack(env, n, F, world):
    """gnnx_rows_to_slots_f32 (the sharded step's halo pack, driven by the rows instead of the send list): the send buffer equals
    the gather pack's (gnnx_gather_rows_f32 by the send list) byte for byte -- rows that go to no peer, to one, to all world - 1;
    with the column sums riding in the pass they are the bits of gnnx_colsum_f32, beta = 1 accumulates; a world of more than 8 ranks
    (a row with more than 7 slots) gets no table and keeps the gather pack."""
    ops, torch = env["ops"], env["torch"]
    gen = torch.Generator(device="cpu").manual_seed(n + F)
    # peer-major send list: every peer wants a random subset of the rows, ascending inside a peer (what HaloSide delivers)
    parts = [torch.sort(torch.randperm(n, generator=gen)[: int(n * frac)]).values for frac in torch.rand(world - 1, generator=gen).tolist()]
    parts[0] = torch.arange(n)   # one peer wants every row: rows with many slots exist
    send_idx = torch.cat(parts).to(torch.int32).to(env["dev"])
    X = ops.uniform_pm1(7, (n, F), device=env["dev"])
    want = ops.gather_rows(X, send_idx)
    table = ops.slot_table(send_idx, n)
    assert table is not None and int((table >= 0).sum()) == send_idx.numel()
    got = torch.full_like(want, float("nan"))
    ops.rows_to_slots(X, table, got)
    assert torch.equal(got, want)
    got2 = torch.full_like(want, float("nan"))
    sums = torch.empty(F, dtype=torch.float32, device=env["dev"])
    ops.rows_to_slots(X, table, got2, colsum_out=sums)
    assert torch.equal(got2, want) and torch.equal(sums, ops.colsum(X))
    acc = ops.colsum(X)
    ops.rows_to_slots(X, table, got2, colsum_out=acc, beta=1.0)
    assert torch.equal(acc, ops.colsum(X, out=ops.colsum(X), beta=1.0))
    # a strided source (rows at the head of a [local | halo] buffer are contiguous; a column slice is not): ld is honoured
    Xp = torch.zeros((n, F + 8), dtype=torch.float32, device=env["dev"])
    Xp[:, :F] = X
    got3 = torch.empty_like(want)
    ops.rows_to_slots(Xp[:, :F], table, got3)
    assert torch.equal(got3, want)
    too_many = torch.cat([torch.arange(n)] * 8).to(torch.int32).to(env["dev"])
    assert ops.slot_table(too_many, n) is None


@pytest.mark.parametrize("M,F,world", [(256 * (256 + 18) + 100, 256, 8),    # 256 whole tiles per CU round + 18 left: 128 x 128 tail + ragged rows
                                       (256 * (256 + 100) + 5, 256, 4),    # 100 left: 256 x 128 tail
                                       (256 * 40, 256, 8),                  # less than a round: no tail launch
                                       (256 * (256 + 66) + 3, 128, 8),      # 128 wide: 256 x 128 tiles, 128 x 128 tail
                                       (256 * 9 + 17, 64, 3),               # narrow output: guarded column tile -> product, then the pack pass
                                       (1000, 256, 2)])                     # below the LDS-DMA kernel's sizes: the two calls it replaces
def test_transform_with_the_pack_in_its_epilogue(env, M, F, world):
    """gnnx_gemm_nt_rows_to_slots_f32 (the sharded step's transform: the halo pack rides in the product's epilogue): H has the bits of
    gnnx_gemm_f32's X . W^T and the send buffer is the gather pack of H, byte for byte -- rows that go to no peer, to one, to all
    world - 1, across the main launch, the smaller-tile launch of the last partial round and the ragged rows.  The smaller-tile launch
    itself: its rows equal the same rows multiplied at the head of a product that has no tail (every geometry runs the same MFMA chain
    per output element)."""
    ops, torch = env["ops"], env["torch"]
    gen = torch.Generator(device="cpu").manual_seed(M + F)
    parts = [torch.sort(torch.randperm(M, generator=gen)[: int(M * frac)]).values for frac in (0.6 * torch.rand(world - 1, generator=gen)).tolist()]
    parts[-1] = torch.arange(0, M, 3)   # (rows with many slots exist wherever the random subsets overlap)
    send_idx = torch.cat(parts).to(torch.int32).to(env["dev"])
    table = ops.slot_table(send_idx, M)
    X = ops.uniform_pm1(11, (M, F), device=env["dev"])
    W = ops.uniform_pm1(12, (F, F), scale=F ** -0.5, device=env["dev"])
    H_ref = ops.linear_fwd(X, W)
    want = ops.gather_rows(H_ref, send_idx)
    H = torch.full_like(H_ref, float("nan"))
    send = torch.full_like(want, float("nan"))
    ops.linear_fwd_rows_to_slots(X, W, H, table, send)
    assert torch.equal(H, H_ref)
    assert torch.equal(send, want)
    # strided destinations: H at the head of a wider buffer (the [local | halo] buffer on the gather pitch), send buffer likewise
    Hp = torch.zeros((M, F + 64), dtype=torch.float32, device=env["dev"])
    sp = torch.zeros((want.shape[0], F + 8), dtype=torch.float32, device=env["dev"])
    ops.linear_fwd_rows_to_slots(X, W, Hp[:, :F], table, sp[:, :F])
    assert torch.equal(Hp[:, :F], H_ref) and torch.equal(sp[:, :F], want)
    assert float(Hp[:, F:].abs().max()) == 0.0 and float(sp[:, F:].abs().max()) == 0.0
    # nothing is written outside the send rows: the buffer sits between two sentinel rows
    guard = torch.full((want.shape[0] + 2, F), 7.0, dtype=torch.float32, device=env["dev"])
    H2 = torch.empty_like(H_ref)
    ops.linear_fwd_rows_to_slots(X, W, H2, table, guard[1:-1])
    assert torch.equal(guard[1:-1], want) and bool((guard[0] == 7.0).all()) and bool((guard[-1] == 7.0).all()) and torch.equal(H2, H_ref)
    # the last partial round against a product without one: 2048 of its rows multiplied at the head of an 8-tile product
    if M >= 256 * 256 + 2048:
        r0 = (M // 256) // 256 * 256 * 256
        assert torch.equal(ops.linear_fwd(X[r0:r0 + 2048].contiguous(), W), H_ref[r0:r0 + 2048])


@pytest.mark.parametrize("K,F", [(64 * 1024 + 64 * 5 + 17, 256), (64 * 1500 + 63, 128), (64 * 1100, 256)])
def test_weight_gradient_over_a_row_count_off_the_k_tile_grid(env, K, F):
    """dW = dH^T . X over a shard's rows (1.25 M is no multiple of 64): the LDS-DMA kernel takes the whole K-tile pairs -- dealt to the
    splits evenly, 76 or 77 pairs each, none empty -- and the last K % 64 rows arrive through one more slab of the in-order reduction.
    Against float64 at the GEMM bar, 1e-5 * max(1, |ref|) (the reference's own sum order over 10^5 rows is no closer to float64), run
    to run identical, and beta = 1 accumulates on top."""
    ops, torch = env["ops"], env["torch"]
    dH = ops.uniform_pm1(41, (K, F), device=env["dev"])
    X = ops.uniform_pm1(42, (K, F), device=env["dev"])
    ref = dH.double().t() @ X.double()
    a = ops.gemm(dH, X, transA=True)
    b = ops.gemm(dH, X, transA=True)
    assert torch.equal(a, b)
    assert (a.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    c = a.clone()
    ops.gemm(dH, X, transA=True, out=c, beta=1.0)
    assert (c.double() - 2 * ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_transform_with_the_pack_edge_cases(env):
    """gnnx_gemm_nt_rows_to_slots_f32 at its edges: no row listed anywhere (the send buffer is not touched), every row listed seven
    times (a world of 8: each row to every peer), an empty product, and the error for rows that are no 16-byte pieces."""
    ops, capi, torch = env["ops"], env["capi"], env["torch"]
    dev_ = env["dev"]
    M, F = 256 * 20 + 9, 256
    X = ops.uniform_pm1(21, (M, F), device=dev_)
    W = ops.uniform_pm1(22, (F, F), scale=F ** -0.5, device=dev_)
    H_ref = ops.linear_fwd(X, W)
    none = torch.full((M, ops.SLOTS_PER_ROW), -1, dtype=torch.int32, device=dev_)
    send = torch.full((4, F), 3.0, dtype=torch.float32, device=dev_)
    H = torch.empty_like(H_ref)
    ops.linear_fwd_rows_to_slots(X, W, H, none, send)
    assert torch.equal(H, H_ref) and bool((send == 3.0).all())
    send_idx = torch.cat([torch.arange(M)] * 7).to(torch.int32).to(dev_)    # peer-major: every peer wants every row
    table = ops.slot_table(send_idx, M)
    assert table is not None and int((table[:, :7] >= 0).sum()) == 7 * M and bool((table[:, 7] == -1).all())
    send7 = torch.empty((7 * M, F), dtype=torch.float32, device=dev_)
    ops.linear_fwd_rows_to_slots(X, W, H, table, send7)
    assert torch.equal(send7, H_ref.repeat(7, 1)) and torch.equal(H, H_ref)
    ops.linear_fwd_rows_to_slots(X[:0], W, H[:0], table[:0], send7)          # M == 0: nothing to do, no error
    Wn = ops.uniform_pm1(23, (6, F), scale=F ** -0.5, device=dev_)           # 6 output columns: rows of 24 bytes
    with pytest.raises(capi.GnnxError):
        ops.linear_fwd_rows_to_slots(X, Wn, torch.empty((M, 6), dtype=torch.float32, device=dev_), table, torch.empty((7 * M, 6), dtype=torch.float32, device=dev_))
