"""Shared test helpers: load the in-tree package (its directory name `gnn.cpp_amd` contains a dot, so
it is loaded by path and registered under the importable alias `gnncpp_amd`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
synth = pkg.synth
