"""CPU-side guard for the asm-pipelined recurrent kernels: hipcc must not touch a prefetch register between the
asm-issued load and the s_waitcnt that lands it (tools/check_asm_prefetch.py; the hazard is described in DESIGN.md
section 4.1)."""
import importlib.util
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_asm_prefetch.py")


def _tool():
    spec = importlib.util.spec_from_file_location("check_asm_prefetch", TOOL)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


LOOP = """
_Z4demo:
	global_load_dwordx2 v[10:11], v[2:3], off
.LBB0_1:
	;;#ASMSTART
	global_load_dwordx2 v[12:13], v[4:5], off
	;;#ASMEND
	v_pk_fma_f32 v[20:21], v[30:31], v[32:33], v[20:21]
	{first}
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_f32_e32 v40, v10, v20
	;;#ASMSTART
	global_load_dwordx2 v[10:11], v[4:5], off
	;;#ASMEND
	{second}
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_f32_e32 v41, v12, v21
	s_cbranch_scc0 .LBB0_1
	s_endpgm
"""


@pytest.mark.parametrize("first,second,bad", [
    ("s_nop 0", "s_nop 0", False),
    ("v_mov_b64_e32 v[50:51], v[12:13]", "s_nop 0", True),          # the copy the broken lstm_rec4_kernel had
    ("s_nop 0", "v_accvgpr_write_b32 a3, v13", True),              # AGPR copy of an in-flight value
    ("scratch_store_dwordx2 off, v[12:13], s0", "s_nop 0", True),  # spill
    ("s_nop 0", "v_pk_mov_b32 v[60:61], v[10:11], v[10:11]", True),  # v[10:11] was re-issued just above
    ("v_mov_b32_e32 v12, 0", "s_nop 0", True),                     # register re-used as a temporary
    ("v_mov_b32_e32 v52, v10", "s_nop 0", True),                   # wrap-around: v[10:11] of the previous iteration has seen ONE wait
])
def test_lint_state_machine_on_synthetic_isa(first, second, bad):
    res = _tool().lint_kernel("demo", LOOP.format(first=first, second=second).splitlines())
    assert res is not None
    dests, findings = res
    assert dests == [(10, 11), (12, 13)]
    assert bool(findings) == bad, findings


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_no_compiler_instruction_touches_an_in_flight_prefetch():
    r = subprocess.run([sys.executable, TOOL], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # every asm-prefetch kernel family, every instantiated width, must have been seen by the lint
    for fam in ("lstm_rec1_kernel", "lstm_rec2_kernel", "lstm_rec4_kernel", "lstm_rec4m_kernel", "gru_rec1_kernel", "gru_rec2_kernel",
                "lstm_bwd_rec_kernel"):
        seen = [l for l in r.stdout.splitlines() if fam in l]
        assert len(seen) >= 3 and all(l.rstrip().endswith("OK") for l in seen), (fam, seen)
    assert sum("lstm_rec2_kernel" in l and "Lb1E" in l for l in r.stdout.splitlines()) == 4      # the TRAIN variants too
    assert sum("gru_bwd_rec_kernel" in l and l.rstrip().endswith("OK") for l in r.stdout.splitlines()) == 2   # BPTT of the GRU (round 3)
    # the split-bf16 projection GEMM: its two trailing prefetches stay tied to the final wait (an untied diagnostic build let the
    # epilogue write v81 of an in-flight load and faulted on the GPU; this lint reports exactly that instruction for that build)
    assert sum("proj_gemm_b3_kernel" in l and l.rstrip().endswith("OK") for l in r.stdout.splitlines()) == 1


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_split_gemm_main_loop_keeps_its_software_pipeline(tmp_path):
    """proj_gemm_b3_kernel: two 16-deep chunks per loop trip = 48 bf16 MFMAs, and the ONLY vector-memory waits in
    the loop are the two counted `s_waitcnt vmcnt(4)` that leave the prefetch of chunk c+2 in flight -- a compiler-placed
    `vmcnt(0)` there is what made the first pipelined version slower than the unpipelined one."""
    import re
    out = tmp_path / "gemm.s"
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast",
                           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                           os.path.join(ROOT, "climsim_amd", "csrc", "gemm.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    a = next(i for i, l in enumerate(lines) if l.startswith("_Z19proj_gemm_b3_kernel"))
    b = next(i for i, l in enumerate(lines) if i > a and l.strip().startswith("s_endpgm"))
    body = lines[a:b]
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append(body[labels[m.group(1)]:i])
    main = [seg for seg in loops if any("v_mfma_f32_32x32x16_bf16" in x for x in seg)]
    assert len(main) == 1
    seg = main[0]
    assert sum("v_mfma_f32_32x32x16_bf16" in x for x in seg) == 48
    assert [x.strip() for x in seg if "vmcnt" in x] == ["s_waitcnt vmcnt(4)", "s_waitcnt vmcnt(4)"]
    assert sum(x.strip().startswith("global_load_dwordx4") for x in seg) == 8
    assert not any("scratch_" in x for x in body)      # no spills at three workgroups per CU
