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
