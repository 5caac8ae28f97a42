"""CPU: the C-ABI library loads and exports every symbol include/climsim_amd.h declares; the host
mirror fails loudly without a GPU (no silent CPU fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_npz_model


def test_library_exports_every_declared_symbol():
    from climsim_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "climsim_amd.h")).read()
    declared = set(re.findall(r"\b(csa_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"csa_status"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.csa_version()


def test_struct_layouts_match_header():
    from climsim_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "climsim_amd.h")).read()
    cfg_body = re.search(r"typedef struct \{(.*?)\} csa_config;", hdr, re.S).group(1)
    cfg_body = re.sub(r"/\*.*?\*/", "", cfg_body, flags=re.S)
    names = [n.strip() for part in re.findall(r"int32_t([^;]*);", cfg_body) for n in part.split(",")]
    assert names == _lib.CONFIG_FIELDS
    par_body = re.search(r"typedef struct \{(.*?)\} csa_params;", hdr, re.S).group(1)
    par_body = re.sub(r"/\*.*?\*/", "", par_body, flags=re.S)
    names = [n.strip().lstrip("*") for part in re.findall(r"const float([^;]*);", par_body) for n in part.split(",")]
    assert names == _lib.PARAM_FIELDS


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_no_cpu_fallback():
    import climsim_amd
    consts, weights, _ = load_npz_model("v4_memory")
    with pytest.raises(RuntimeError, match="no CPU"):
        climsim_amd.NewModel_constraint(consts, weights)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "climsim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f


@pytest.mark.gpu
def test_small_accessors_of_the_abi():
    """The getters no Python mirror method needs: packed width, max batch, version string, raw parameter pointers."""
    import ctypes
    import numpy as np
    import climsim_amd
    from climsim_amd import _lib
    from climsim_amd.train import Trainer
    from climsim_amd.baselines import CNNTrainer
    L = _lib.lib()
    consts, weights, _ = load_npz_model("v4_memory")
    m = climsim_amd.NewModel_constraint(consts, weights, max_batch=33)
    assert L.csa_packed_width(m.emulator._h) == 368 + 960 == m.emulator.packed_width
    assert L.csa_max_batch(m.emulator._h) == 33
    L.csa_version.restype = ctypes.c_char_p
    assert b"climsim" in L.csa_version().lower() or len(L.csa_version()) > 0
    consts, weights, flags = load_npz_model("cur_lstm128")
    grid = np.load(os.path.join(os.path.dirname(__file__), "golden", "grid_consts.npz"))
    tr = Trainer(consts, weights, grid["hyai"], grid["hybi"], max_batch=4, max_window=1)
    L.csa_train_params.restype = ctypes.c_void_p
    assert L.csa_train_params(tr._h)
    assert L.csa_train_sync_params(tr._h, None) == 0
    ws = [np.zeros((8, 6, 3), np.float32), np.zeros((8, 8, 3), np.float32), np.zeros((8, 6, 1), np.float32),
          np.zeros((10, 8, 1), np.float32), np.zeros((10, 10, 1), np.float32)]
    ct = CNNTrainer(ws, [np.zeros(w.shape[0], np.float32) for w in ws], depth=1, width=8, dropout=0.0, max_batch=2)
    assert L.csa_cnn_train_params(ct._h)
