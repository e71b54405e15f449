"""CPU: the C-ABI library loads, exports every symbol include/eirgrid_hip.h declares, and refuses to run without a GPU."""
import ctypes as C
import os
import re

import pytest

from eirgrid_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "eirgrid_hip.h")).read()
    declared = set(re.findall(r"\b(eg_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    L = N.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, f"libeirgrid_hip.so lacks {missing}"
    assert declared == set(N.EXPORTS), declared ^ set(N.EXPORTS)


def test_constants_agree_with_header(built):
    header = open(os.path.join(ROOT, "include", "eirgrid_hip.h")).read()
    for name, val in (("EG_YEARS", N.YEARS), ("EG_N_ACTIONS", N.N_ACTIONS), ("EG_N_DEFICIT", N.N_DEFICIT),
                      ("EG_N_COUNTS", N.N_COUNTS), ("EG_MAX_GENS", N.MAX_GENS), ("EG_MAX_OFFSETS", N.MAX_OFFSETS),
                      ("EG_RUN_CAP", N.RUN_CAP), ("EG_DEF_CAP", N.DEF_CAP), ("EG_ACT_CAP", N.ACT_CAP),
                      ("EG_YEARLY_FIELDS", N.YEARLY_FIELDS), ("EG_ONCHIP_GENS", N.ONCHIP_GENS)):
        assert int(re.search(rf"#define {name} (\d+)", header).group(1)) == val


def test_no_cpu_fallback(built, world):
    """Without a HIP device the product path must fail loudly (never route through the oracle or any CPU path)."""
    L = N.lib()
    if L.eg_device_count() > 0:
        pytest.skip("a GPU is present")
    from eirgrid_amd.engine import Engine
    with pytest.raises(N.EirgridError):
        Engine(world)
    # and the package never imports the oracle
    import subprocess, sys
    code = "import sys; import eirgrid_amd, eirgrid_amd.engine, eirgrid_amd._native; print(any(m.startswith('oracle') for m in sys.modules))"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.stdout.strip() == "False", out.stdout + out.stderr


def test_product_sources_do_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "eirgrid_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "eg_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
