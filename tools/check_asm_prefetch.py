#!/usr/bin/env python3
"""ISA lint for the asm-pipelined recurrent kernels (rec.hip): the projections of step t+1 are fetched with
`asm volatile("global_load_dwordx{2,4} ...")` and become valid behind `asm("s_waitcnt vmcnt(N)" : "+v"(reg))` a step
later.  hipcc does not know that a load is in flight, so any register copy (`v_mov`) of a prefetch destination that it
places between the load and the wait reads stale data.  This script compiles rec.hip to gfx950 assembly and reports every
`v_mov` whose source overlaps a prefetch destination and for which, scanning backwards, the asm load comes before any
vmcnt wait.  Expected output: no finding.  (A first version of lstm_rec4_kernel had one)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "climsim_amd", "csrc", "rec.hip")
out = os.path.join(tempfile.gettempdir(), "rec_lint.s")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast",
                       "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().splitlines()
starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^(_Z\w+):", l)] + [(len(lines), "END")]
findings = 0
for (a, name), (b, _) in zip(starts, starts[1:]):
    body = lines[a:b]
    dests = set()
    for i, l in enumerate(body):
        m = re.search(r"global_load_dwordx[24] v\[(\d+):(\d+)\]", l)
        if m and i > 0 and "ASMSTART" in body[i - 1]:
            dests.add((int(m.group(1)), int(m.group(2))))
    if not dests:
        continue
    bad = []
    for i, l in enumerate(body):
        m = re.search(r"v_mov_b64_e32 v\[\d+:\d+\], v\[(\d+):(\d+)\]", l)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
        else:
            m = re.search(r"v_mov_b32_e32 v\d+, v(\d+)$", l)
            if not m:
                continue
            lo = hi = int(m.group(1))
        if not any(not (hi < d0 or lo > d1) for d0, d1 in dests):
            continue
        for j in range(i - 1, -1, -1):
            if "vmcnt" in body[j] or re.match(r"^\.LBB\d+_\d+:", body[j]):
                break
            mm = re.search(r"global_load_dwordx[24] v\[(\d+):(\d+)\]", body[j])
            if mm and not (hi < int(mm.group(1)) or lo > int(mm.group(2))):
                bad.append((i, l.strip()))
                break
            # a copy directly in front of a wait statement is the pattern the broken kernel had
        nxt = "\n".join(body[i + 1:i + 5])
        if "s_waitcnt vmcnt" in nxt and "ASMSTART" in nxt and (i, l.strip()) not in bad:
            bad.append((i, l.strip() + "   <- copy placed directly before an asm wait"))
    print(f"{name[:70]:70s} prefetch registers {sorted(dests)}: {'OK' if not bad else bad}")
    findings += len(bad)
sys.exit(1 if findings else 0)
