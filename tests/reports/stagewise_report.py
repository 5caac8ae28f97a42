#!/usr/bin/env python3
"""Writes the stage-wise parity table of tests/test_stagewise_parity.py for the library named by CSA_LIB_PATH (default: the
product build).  Run on the GPU box once with the product build and once with the CSA_FAST_GATES=0 diagnostic build
(python -m climsim_amd.build --exact-gates); both tables go to profiles/r2_stagewise_parity.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from test_stagewise_parity import format_rows, stage_errors      # noqa: E402

print(f"library: {os.environ.get('CSA_LIB_PATH', 'climsim_amd/libclimsim_amd.so (product build, v_exp_f32 / v_rcp_f32 gates)')}")
for tag in ("v4_stateless", "v4_memory"):
    print(format_rows(tag, stage_errors(tag)))
