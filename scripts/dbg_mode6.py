import importlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package()
ops = importlib.import_module("gnncpp_amd.ops")
dev = torch.device("cuda:0")
n, F = 5000, 16
for L in (40, 200, 3000):
    cols = (torch.randperm(n - 1, device=dev)[:L].to(torch.int32) + 1)
    src = torch.zeros(L, dtype=torch.int32, device=dev)
    g = ops.CsrGraph.from_coo(src, cols, n, transpose=False, norm=False)
    g.make_plans(16, F, big_rows=0)
    ci = g.colidx.long()
    X = torch.ones((n, F), dtype=torch.float32, device=dev)
    one_c = torch.ones(n, dtype=torch.float32, device=dev)
    one_v = torch.ones(L, dtype=torch.float32, device=dev)
    pos = torch.arange(L, dtype=torch.float32, device=dev) + 1          # value of entry p (ascending column order) = p + 1
    colv = torch.zeros(n, dtype=torch.float32, device=dev); colv[ci] = pos   # the same number, by column
    for name, vals, sc in (("vals=p, sc=1", pos, one_c), ("vals=1, sc=p", one_v, colv), ("vals=p, sc=p", pos, colv), ("vals=1, sc=1", one_v, one_c)):
        z1 = ops.spmm(g.rowptr, g.colidx, X, vals=vals, colscale=sc, plan=g.plan)[0, 0].item()
        z0 = ops.spmm(g.rowptr, g.colidx, X, vals=vals, colscale=sc)[0, 0].item()
        print(f"L={L} {name}: hubpc {z1} ref {z0}")
    Xp = torch.zeros((n, F), dtype=torch.float32, device=dev); Xp[ci] = pos[:, None]
    z1 = ops.spmm(g.rowptr, g.colidx, Xp, vals=one_v, colscale=one_c, plan=g.plan)[0, 0].item()
    z0 = ops.spmm(g.rowptr, g.colidx, Xp, vals=one_v, colscale=one_c)[0, 0].item()
    print(f"L={L} X=p: hubpc {z1} ref {z0}")
