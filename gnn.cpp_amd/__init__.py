"""gnn.cpp_amd -- MI355X-native GCN message-passing hot path behind the walexi/gnn.cpp API.

Layout:
  csrc/      hand-written HIP kernels for gfx950 + the C-ABI (include/gnnx.h) -> libgnnx_hip.so
  host/      C++ mirror of the reference's cyg::tensor / Operation / nn::Module / graph::GCNConv API,
             dispatching to the C-ABI (the drop-in boundary of SURVEY.md section 8(b))
  capi.py    ctypes binding of the C-ABI for tests / bench (plain pointers, no torch types)
  synth.py   deterministic synthetic graphs / features
  shard.py   1-D vertex partition + halo plan (host index logic) for the multi-GPU path

The directory name contains a dot; load it with __graft_entry__.load_package() (alias `gnncpp_amd`).
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
