"""CPU: the C-ABI library loads and exports every symbol include/ld_mi355x.h declares (no compute without a GPU)."""
import os
import re

from conftest import ROOT


def test_header_and_library_agree():
    from lightdiffusion_amd._lib import LIB_PATH, SIGNATURES, lib
    assert os.path.exists(LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    hdr = open(os.path.join(ROOT, "include", "ld_mi355x.h")).read()
    declared = set(re.findall(r"\b(ld_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(SIGNATURES), declared ^ set(SIGNATURES)
    l = lib()
    for name in declared:
        assert hasattr(l, name), name
    assert l.ld_version().decode().startswith("ld_mi355x")
    assert l.ld_status_string(2).decode() == "unsupported shape or alignment"


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "lightdiffusion_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f
