"""TEST INFRASTRUCTURE ONLY: CPU oracle of the gnn.cpp GCN hot path (see gcn_oracle.c).

May be imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from .oracle import *  # noqa: F401,F403
