#!/usr/bin/env python3
"""BatchNorm-backward apply (gnnx_bn_relu_bwd_apply_f32) at the bench size, 10 M x 256: 30 GB streamed (H and dY read, dH written).
EXPERIMENTS build: GNNX_BN_APPLY_BLOCKS caps the grid.   usage (GPU box): GNNX_HIP_LIB=exp GNNX_BN_APPLY_BLOCKS=512 python scripts/exp_bn_apply.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from __graft_entry__ import load_package  # noqa: E402

load_package()
ops = importlib.import_module("gnncpp_amd.ops")
capi = importlib.import_module("gnncpp_amd.capi")
dev = torch.device("cuda:0")
M, F = int(os.environ.get("M", 10_000_000)), 256
ld = ops.gather_row_stride(M, F) if os.environ.get("PITCH", "1") == "1" else F
Hb = torch.empty((M, ld), dtype=torch.float32, device=dev)
H = Hb[:, :F]
H.copy_(ops.uniform_pm1(1, (M, F), device=dev))
dY = ops.uniform_pm1(2, (M, F), device=dev)
dH = torch.empty((M, F), dtype=torch.float32, device=dev)
mean = H[:100000].mean(0).contiguous()
var = H[:100000].var(0, unbiased=False).contiguous()
gamma = torch.ones(F, device=dev)
beta = torch.zeros(F, device=dev)
dgamma = ops.uniform_pm1(3, (F,), device=dev)
dbeta = ops.uniform_pm1(4, (F,), device=dev)
ws = torch.empty(8 << 20, dtype=torch.uint8, device=dev)


def run():
    capi.call("gnnx_bn_relu_bwd_apply_f32", ops._ptr(H), ld, None, 0, ops._ptr(dY), F, M, F, ops._ptr(mean), ops._ptr(var), 1e-5, ops._ptr(gamma),
              ops._ptr(beta), 1, ops._ptr(dgamma), ops._ptr(dbeta), M, ops._ptr(dH), F, ops._ptr(ws), ws.numel(), ops._stream())


for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    run()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print(json.dumps({"blocks_cap": os.environ.get("GNNX_BN_APPLY_BLOCKS", "default"), "ld_H": ld, "ms": round(ms, 3), "TBps": round(3.0 * M * F * 4 / ms / 1e9, 2)}), flush=True)
