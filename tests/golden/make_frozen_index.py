#!/usr/bin/env python3
"""Index of the 82 frozen physRNN exports (rnn/saved_models/*_wrapped.pt): serialised-code variant of every file, whether the variant
is built (a fixture tests/golden/frozen_<hash>.npz exists and the HIP path reproduces it), and for the `_gpu` exports -- frozen with
horizontally fused Linear heads and CUDA device literals, so they do not execute in the CPU-only build container -- whether the
constants named by frozen_extract.py equal those of the `_cpu` twin (then it is the same model and the same HIP path).
Output: tests/golden/frozen_index.json (file names, hashes, statuses; data only)."""
import glob
import json
import os
import sys

import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, OUT)
from frozen_extract import Extractor  # noqa: E402
from make_golden_frozen import DIR, code_hash  # noqa: E402

FLAGS = ("nreg", "rnn3", "pred_subgrid_liq_frac", "rad_updated_qv", "albedo_mix_learned", "sw_gas_reduce", "cld_band_matrix", "ice_optics_on_ice_radius", "sw_mlp")


def main():
    torch.set_num_threads(4)
    built = {os.path.basename(f)[7:15] for f in glob.glob(os.path.join(OUT, "frozen_*.npz"))}
    idx, cache = {}, {}

    def extract(path):
        if path not in cache:
            m = torch.jit.load(path, map_location="cpu")
            h = code_hash(m.code)
            try:
                P, F = Extractor(m).run()
                if F["unnamed"] or (F.get("band_idx") is None and not F["cld_band_matrix"] and not F["sw_mlp"]):
                    P, why = None, "outside the built family: " + ", ".join(f"{k}={F[k]}" for k in ("unnamed",) if F[k]) + (
                        "" if F.get("band_idx") else " other cloud-optics band code")
                else:
                    why = None
            except Exception as e:
                P, F, why = None, None, "constants not named (another sub-generation of the decoder / SW scheme): " + repr(e)[:70]
            cache[path] = (h, P, F, why)
        return cache[path]
    for f in sorted(glob.glob(DIR + "*_wrapped.pt")):
        name = os.path.basename(f)
        h, P, F, why = extract(f)
        e = {"code": h}
        if F:
            e["flags"] = {k: F[k] for k in FLAGS}
        if "_cpu_wrapped" in name:
            e["status"] = "built" if h in built else ("not built: " + (why or "no fixture"))
        else:
            twin = f.replace("_gpu_wrapped", "_cpu_wrapped")
            if P is None:
                e["status"] = "not built: " + why
            elif os.path.exists(twin):
                ht, Pt, Ft, _ = extract(twin)
                same = Pt is not None and set(P) == set(Pt) and all(torch.equal(P[k].cpu(), Pt[k]) for k in Pt if k != "solar_weights") and \
                    float((P["solar_weights"].cpu() - Pt["solar_weights"]).abs().max()) <= 6e-8
                e["cpu_twin_code"] = ht
                e["status"] = ("twin of a built variant: constants identical (folded solar weights to 1 ulp)" if same and ht in built
                               else "twin of a variant that is not built" if ht not in built else "another checkpoint than its _cpu twin (constants differ); switches inside the built family; not executable without CUDA")
            else:
                e["status"] = "no _cpu twin; constants named, all switches inside the built family; not executable without CUDA"
        idx[name] = e
    json.dump(idx, open(os.path.join(OUT, "frozen_index.json"), "w"), indent=1, sort_keys=True)
    import collections
    print(collections.Counter(v["status"].split(":")[0] for v in idx.values()))


if __name__ == "__main__":
    if not os.path.isdir(DIR):
        sys.exit("reference not present")
    main()
