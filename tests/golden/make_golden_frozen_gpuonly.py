#!/usr/bin/env python3
"""Weights-only fixtures for the eight `_gpu_wrapped.pt` exports that hold weights of their OWN (no `_cpu` twin, or another checkpoint
than their twin).  They carry CUDA device literals and do not execute in the build container, so they have no reference outputs;
what CAN be taken from them is data: the graph constants, named by climsim_amd/frozen_extract.py, and the switches of their
serialised code -- every one of which lies inside a code variant whose `_cpu` files pin the restatement (tests/golden/frozen_index.json).
tests/test_physrnn_frozen.py then holds the HIP path to the float64 restatement with THESE weights: parity of these eight files is
pinned through the restatement only (no output of the file itself) -- said so where the test is documented.
Output: tests/golden/frozen_gpuonly_<num>.npz -- named weights ("w."), switches ("flag.").  Data only."""
import json
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from frozen_extract import Extractor  # noqa: E402
from make_golden_frozen import DIR, KEEP, DROP, code_hash  # noqa: E402


def main():
    idx = json.load(open(os.path.join(OUT, "frozen_index.json")))
    for name, e in sorted(idx.items()):
        if not (e["status"].startswith("another checkpoint") or e["status"].startswith("no _cpu twin")):
            continue
        m = torch.jit.load(DIR + name, map_location="cpu")
        P, Fl = Extractor(m).run()
        assert not Fl["unnamed"], (name, Fl["unnamed"])
        d = {"w." + k: v.cpu().numpy() for k, v in P.items() if k.startswith(KEEP) and k not in DROP}
        for k, v in Fl.items():
            if isinstance(v, (bool, int)):
                d["flag." + k] = np.array(int(v), np.int64)
        d["cfg.band_idx"] = np.array(Fl["band_idx"] if Fl.get("band_idx") else [0] * Fl["nreg"], np.int64)
        d["artefact"], d["code"] = np.array(name), np.array(code_hash(m.code))
        tag = name.split("_num")[1].split("_script")[0]
        np.savez_compressed(f"{OUT}/frozen_gpuonly_{tag}.npz", **d)
        print(tag, e["code"], {k: Fl[k] for k in ("nreg", "rnn3", "pred_subgrid_liq_frac", "sw_mlp")}, len(d))


if __name__ == "__main__":
    if not os.path.isdir(DIR):
        sys.exit("reference not present: fixtures can only be regenerated in the build container")
    main()
