"""The C-ABI library loads and exports every symbol include/boundmpc.h declares (no GPU calls)."""
import os
import re

from boundplanner_amd import solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header():
    import __graft_entry__ as ge
    ge.build()
    hdr = open(os.path.join(ROOT, "include", "boundmpc.h")).read()
    declared = set(re.findall(r"\b(bmpc_[a-z0-9_]+)\s*\(", hdr))
    lib = solver.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(solver.EXPORTS)


def test_no_cpu_fallback_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        solver.HipBoundMPC(10)
