"""CPU: the C-ABI library loads and exports every symbol include/ld_mi355x.h declares (no compute without a GPU)."""
import os
import re
import pytest

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


def test_isa_check_of_untracked_loads(tmp_path):
    """the counted waits behind hand-issued loads (conv8's ld16_sc1, gemm's epi_stage_load) are only as good as the code
    hipcc generates around them.  tools/isa_check.py re-checks the assembly: here its hazard model on two synthetic kernels, then the real
    conv8.hip (gemm.hip takes minutes: `make -C lightdiffusion_amd/csrc isa-check`)."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import isa_check
    good = tmp_path / "good.s"
    good.write_text("k_good:\n global_load_dwordx4 v[4:7], v[0:1], off\n global_load_lds_dwordx4 v[2:3], off\n v_add_u32 v9, v8, v8\n"
                    " s_waitcnt vmcnt(1)\n ds_write_b128 v10, v[4:7]\n s_endpgm\n")
    bad = tmp_path / "bad.s"
    bad.write_text("k_bad:\n global_load_dwordx4 v[4:7], v[0:1], off\n global_load_lds_dwordx4 v[2:3], off\n s_waitcnt vmcnt(2)\n"
                   " v_mov_b32 v11, v5\n s_waitcnt vmcnt(0)\n s_endpgm\n")
    assert isa_check.check_asm(str(good)) == []
    f = isa_check.check_asm(str(bad))
    assert len(f) == 1 and "v_mov_b32" in f[0] and "k_bad" in f[0]
    if shutil.which(isa_check.HIPCC) is None:
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_check.py"), "conv8.hip"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
