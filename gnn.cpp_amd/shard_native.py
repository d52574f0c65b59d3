"""ctypes view of the sharding entry points of the C-ABI (gnnx_vertex_weights, gnnx_partition_deal,
gnnx_shard_select_edges, gnnx_halo_plan_*, gnnx_comm_init_local, gnnx_halo_exchange_rows_f32): the plan the C++ host
(gnn.cpp_amd/host, graph::Partition) builds, exposed to the tests so that it can be compared array for array with the
torch-index-op plan of shard.py and driven through a whole sharded step.  torch only owns the device buffers."""
import ctypes as C

import torch

from . import capi, ops


def local_comms(world):
    """Handles of an in-process communicator group, one per rank (gnnx_comm_init_local)."""
    arr = (C.c_void_p * world)()
    capi.call("gnnx_comm_init_local", arr, world)
    return [C.c_void_p(arr[r]) for r in range(world)]


def comm_destroy(h):
    capi.lib().gnnx_comm_destroy(h)


class NativeHaloSide:
    """One direction of the exchange behind the C-ABI: shard CSR ([local | halo] columns) + gnnx_halo_plan."""

    def __init__(self, src, dst, owner, nid, cuts, rank, world, n_nodes, transpose):
        dev = src.device
        E = int(src.numel())
        lo, hi = cuts[rank], cuts[rank + 1]
        self.n_local = hi - lo
        rows = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        cols = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
        nsel = C.c_int64(0)
        capi.call("gnnx_shard_select_edges", ops._ptr(src), ops._ptr(dst), E, ops._ptr(owner), ops._ptr(nid), rank, lo, int(transpose),
                  ops._ptr(rows), ops._ptr(cols), C.byref(nsel), ops._stream())
        m = nsel.value
        rp, ci = ops.CsrGraph.csr_from_coo(rows[:m], cols[:m], max(self.n_local, n_nodes), flags=1)
        self.rowptr = rp[: self.n_local + 1].contiguous()
        self.colidx = torch.empty_like(ci)
        self.h = C.c_void_p()
        ccuts = (C.c_int64 * (world + 1))(*cuts)
        capi.call("gnnx_halo_plan_create", ops._ptr(ci), int(ci.numel()), ops._ptr(nid), n_nodes, world, rank, ccuts, ops._ptr(self.colidx),
                  C.byref(self.h), ops._stream())
        self.world = world
        self.refresh()

    def refresh(self):
        w = self.world
        nl, nh, ns = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = (C.c_int64 * w)()
        halo_p, send_p = C.c_void_p(), C.c_void_p()
        capi.call("gnnx_halo_plan_info", self.h, C.byref(nl), C.byref(nh), C.byref(ns), rc, None, C.byref(halo_p), C.byref(send_p))
        self.n_halo, self.n_send = nh.value, ns.value
        self.recv_counts = list(rc)
        self.send_counts = None
        if ns.value >= 0:
            sc = (C.c_int64 * w)()
            capi.call("gnnx_halo_plan_info", self.h, None, None, None, None, sc, None, None)
            self.send_counts = list(sc)
        self._halo_ptr, self._send_ptr = halo_p.value, send_p.value

    def _copy_i32(self, ptr, n, device):
        out = torch.empty(max(n, 0), dtype=torch.int32, device=device)
        if n > 0:
            capi.call("gnnx_memcpy_d2d", ops._ptr(out), C.c_void_p(ptr), n * 4, ops._stream())
        return out

    def halo(self, device):
        return self._copy_i32(self._halo_ptr, self.n_halo, device)

    def send_idx(self, device):
        return self._copy_i32(self._send_ptr, self.n_send, device)

    def exchange_requests(self, comm):
        capi.call("gnnx_halo_plan_exchange_requests", self.h, comm, ops._stream())
        self.refresh()

    def exchange_rows(self, comm, buf, send_buf):
        capi.call("gnnx_halo_exchange_rows_f32", self.h, comm, ops._ptr(buf), buf.stride(0), buf.shape[1], ops._ptr(send_buf), ops._stream())

    def slot_table(self, device):
        """gnnx_halo_plan_slot_table: the plan's own [n_local, 8] table (a copy; None when the plan has none)."""
        p = C.c_void_p()
        capi.call("gnnx_halo_plan_slot_table", self.h, C.byref(p))
        if not p.value:
            return None
        return self._copy_i32(p.value, self.n_local * 8, device).view(-1, 8)

    def exchange_packed(self, comm, buf, send_buf):
        """gnnx_halo_exchange_packed_f32: the exchange of a send buffer its producer has filled (ops.linear_fwd_rows_to_slots)."""
        capi.call("gnnx_halo_exchange_packed_f32", self.h, comm, ops._ptr(buf), buf.stride(0), buf.shape[1], ops._ptr(send_buf), ops._stream())

    def __del__(self):
        try:
            capi.lib().gnnx_halo_plan_destroy(self.h)
        except Exception:
            pass


class NativeShardPlan:
    """Partition + both halo plans of rank `rank`, every step a C-ABI call."""

    def __init__(self, src, dst, n_nodes, rank, world, comm, row_weight=1):
        dev = src.device
        E = int(src.numel())
        w = torch.empty(n_nodes, dtype=torch.int32, device=dev)
        capi.call("gnnx_vertex_weights", ops._ptr(src), ops._ptr(dst), E, n_nodes, int(row_weight), ops._ptr(w), ops._stream())
        self.owner = torch.empty(n_nodes, dtype=torch.int32, device=dev)
        self.nid = torch.empty(n_nodes, dtype=torch.int32, device=dev)
        ccuts = (C.c_int64 * (world + 1))()
        capi.call("gnnx_partition_deal", ops._ptr(w), n_nodes, world, ops._ptr(self.owner), ops._ptr(self.nid), ccuts, ops._stream())
        capi.call("gnnx_partition_scramble", ops._ptr(self.owner), n_nodes, world, ccuts, ops._ptr(self.nid), ops._stream())
        self.cuts = list(ccuts)
        self.rank, self.world = rank, world
        self.lo, self.hi = self.cuts[rank], self.cuts[rank + 1]
        self.n_local = self.hi - self.lo
        self.fwd = NativeHaloSide(src, dst, self.owner, self.nid, self.cuts, rank, world, n_nodes, transpose=False)
        self.bwd = NativeHaloSide(src, dst, self.owner, self.nid, self.cuts, rank, world, n_nodes, transpose=True)
        if comm is not None:
            self.fwd.exchange_requests(comm)
            self.bwd.exchange_requests(comm)
