#!/usr/bin/env python3
"""ISA lint for the asm-pipelined kernels (rec.hip, train_rec.hip; gemm.hip::proj_gemm_b3_kernel).

The projections of step t+1 are fetched with `asm volatile("global_load_dword{,x2,x4} ...")` at the top of step t and
become valid behind an `asm("s_waitcnt vmcnt(N)" : "+v"(reg))` in step t+1 (double buffering: the registers of a prefetch
are consumed behind the SECOND asm wait that follows its load).  hipcc does not know that a load is in flight, so ANY
instruction of its own that touches a prefetch destination inside that window -- a `v_mov` / `v_pk_mov` / `v_accvgpr_write`
copy, a `scratch_store` / `buffer_store` spill, a re-use of the register as a temporary -- reads or destroys stale data.

This script compiles the source to gfx950 assembly and, per kernel, walks prologue -> loop body -> loop body (the back
edge) -> epilogue with a small state machine:
    any VMEM instruction       -> appended to an in-order queue (asm loads with their destination registers D)
    `s_waitcnt vmcnt(N)`       -> all but the newest N entries of the queue have landed (gfx9: loads and stores retire in order)
    compiler instruction whose operands overlap the D of a queued asm load -> FINDING
Instructions between `;;#ASMSTART` and `;;#ASMEND` are the hand-written ones and exempt.  Expected output: every kernel that
uses the asm prefetch is listed with OK.  (A first version of lstm_rec4_kernel had a finding)

    python tools/check_asm_prefetch.py [source.hip ...]      exit status 1 on any finding
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = [os.path.join(ROOT, "climsim_amd", "csrc", f) for f in ("rec.hip", "train_rec.hip", "gemm.hip")]      # gemm.hip: proj_gemm_b3_kernel
LOAD = re.compile(r"^\s*global_load_dword(?:x([234]))?\s+(v\[(\d+):(\d+)\]|v(\d+)),")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = []
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.append((int(m.group(3)), int(m.group(3))))
        else:
            out.append((int(m.group(1)), int(m.group(2))))
    return out


def overlap(a, b):
    return not (a[1] < b[0] or a[0] > b[1])


def lint_kernel(name, body):
    # mark asm blocks
    in_asm, tagged = False, []
    for l in body:
        if "#ASMSTART" in l:
            in_asm = True
            continue
        if "#ASMEND" in l:
            in_asm = False
            continue
        tagged.append((in_asm, l))
    asm_loads = [i for i, (a, l) in enumerate(tagged) if a and LOAD.match(l)]
    if not asm_loads:
        return None
    # main loop = innermost label..backward-branch span that contains the first asm load
    labels = {}
    for i, (_, l) in enumerate(tagged):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    loop = None
    for i, (_, l) in enumerate(tagged):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= asm_loads[0] < i:
            if loop is None or labels[m.group(1)] > loop[0]:
                loop = (labels[m.group(1)], i + 1)
    order = list(range(len(tagged)))
    if loop:
        order = list(range(0, loop[1])) + list(range(loop[0], loop[1])) + list(range(loop[1], len(tagged)))
    # in-order VMEM queue (gfx9 family: loads and stores retire in issue order, `s_waitcnt vmcnt(N)` leaves the newest N in flight)
    queue, findings, dests = [], [], set()
    VMEM = re.compile(r"^\s*(global|buffer|scratch|flat)_(load|store|atomic)")
    for i in order:
        a, l = tagged[i]
        ins = l.split("//")[0].split(";")[0].strip()
        if not ins or ins.endswith(":") or ins.startswith("."):
            continue
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
        if m:
            n = int(m.group(1))
            queue = queue[len(queue) - n:] if n else []
            continue
        if a and LOAD.match(l):
            m = LOAD.match(l)
            d = (int(m.group(3)), int(m.group(4))) if m.group(3) is not None else (int(m.group(5)), int(m.group(5)))
            queue.append(d)
            dests.add(d)
            continue
        if a:
            continue
        inflight = [d for d in queue if d is not None]
        for r in regs_of(ins.split(None, 1)[1] if " " in ins else ""):
            for d in inflight:
                if overlap(r, d):
                    findings.append(f"line {i}: `{ins}` touches v[{d[0]}:{d[1]}] while its asm load is in flight")
        if VMEM.match(l):
            queue.append(None)          # a compiler-issued load / store occupies a slot of the counter too
    return sorted(dests), sorted(set(findings))


def main(srcs):
    total = 0
    for src in srcs:
        out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".lint.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                               "-ffp-contract=fast", "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out],
                              stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
        starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^(_Z\w+):", l)]
        ends = [i for i, l in enumerate(lines) if l.strip().startswith("s_endpgm")]
        for a, name in starts:
            b = min([e for e in ends if e > a] + [len(lines)]) + 1
            res = lint_kernel(name, lines[a:b])
            if res is None:
                continue
            dests, findings = res
            print(f"{name[:80]:80s} prefetch registers {dests}: {'OK' if not findings else findings}")
            total += len(findings)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or DEFAULT))
