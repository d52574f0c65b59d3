"""Build-time lint of the hand-scheduled LDS reads (CPU only; scripts/check_lds_asm_discipline.py).

The hub kernels of the aggregation and the LDS-DMA products issue LDS reads from inline asm and cover them with hand-counted
`s_waitcnt lgkmcnt(N)`; the compiler may copy or overwrite an asm's output registers before the wait that covers the load (it did,
once: round 5, spmm_hubpc_kernel with an early return in its loop -- run-to-run different sums on every hub row).  The lint walks
every path of the built kernels with the queue of outstanding LDS operations and fails on any instruction that touches a register
whose load no wait has covered yet."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINT = os.path.join(ROOT, "scripts", "check_lds_asm_discipline.py")


@pytest.mark.parametrize("obj,kernels,at_least", [("gnnx_spmm.o", "spmm_hubpc_kernel|spmm_hub_kernel", 20), ("gnnx_gemm.o", "gemm_dma", 8)])
def test_no_instruction_touches_an_lds_read_before_its_wait(obj, kernels, at_least):
    path = os.path.join(ROOT, "gnn.cpp_amd", "csrc", obj)
    assert os.path.exists(path), "build it with __graft_entry__.build()"
    r = subprocess.run([sys.executable, LINT, path, "--kernels", kernels], capture_output=True, text=True, timeout=900)
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("ok ")]
    assert r.returncode == 0 and len(ok) >= at_least, r.stdout[-4000:] + r.stderr[-2000:]


def test_the_lint_sees_a_planted_violation(tmp_path):
    """The checker itself: a register copy of an in-flight read set at a loop's back edge (the shape of the round-5 bug) is reported,
    the same code with the copy behind the wait is not."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_lds_asm_discipline as lint

    def prog(copy_before_wait):
        body = [(0x00, "ds_read2_b32", "v[16:17], v30 offset1:16"), (0x08, "ds_read2_b32", "v[18:19], v30 offset0:32 offset1:48")]
        if copy_before_wait:
            body += [(0x10, "v_mov_b64_e32", "v[0:1], v[16:17]"), (0x14, "s_waitcnt", "lgkmcnt(0)")]
        else:
            body += [(0x10, "s_waitcnt", "lgkmcnt(0)"), (0x14, "v_mov_b64_e32", "v[0:1], v[16:17]")]
        body += [(0x18, "v_add_f32_e32", "v2, v0, v1"), (0x1c, "s_cbranch_scc1", "65528"), (0x20, "s_endpgm", "")]   # back to 0x00
        return body
    bad, _ = lint.check("planted", prog(True), 10000)
    good, _ = lint.check("planted", prog(False), 10000)
    assert len(bad) == 1 and "v16" in bad[0] and not good
    # a counted wait retires the OLDEST reads only: with lgkmcnt(1) the second read is still in flight
    p = [(0x00, "ds_read_b32", "v4, v9"), (0x08, "ds_read_b32", "v5, v9 offset:4"), (0x10, "s_waitcnt", "lgkmcnt(1)"),
         (0x14, "v_add_f32_e32", "v6, v4, v4"), (0x18, "v_add_f32_e32", "v7, v5, v5"), (0x1c, "s_endpgm", "")]
    v, _ = lint.check("counted", p, 10000)
    assert len(v) == 1 and "v5" in v[0]
