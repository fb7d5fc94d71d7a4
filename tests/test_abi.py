"""The C-ABI library loads on a machine without a GPU and exports every symbol include/pta_replicator_amd.h
declares, with the argument counts the ctypes binding assumes.  No compute calls here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pta_replicator_amd.h")


def declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int64_t|int|const char \*)\s*\*?\s*(pta_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_header_declares_the_expected_surface():
    d = declared()
    for name in ("pta_rn_basis", "pta_rn_synth", "pta_wn", "pta_ecorr", "pta_quantize_epochs", "pta_orf_hd", "pta_orf_basis",
                 "pta_potrf_batched", "pta_gwb_twiddle", "pta_gwb_idft", "pta_gwb_idft_rng", "pta_gwb_mix", "pta_gwb_interp",
                 "pta_cgw", "pta_engine_synth", "pta_td_cov_assemble", "pta_td_trmm", "pta_td_trmm_rng", "pta_potrf_batched_ex", "pta_dgemm",
                 "pta_rng_fill_normal"):
        assert name in d, name
    assert not [n for n in d if n.startswith("pta_set_")], "the ABI has no process-wide switches"


def test_library_exports_every_declared_symbol():
    from pta_replicator_amd import _lib
    syms = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (pta_\w+)", syms))
    d = declared()
    missing = set(d) - exported
    assert not missing, f"declared in the header but not exported: {missing}"
    # and the ctypes binding covers the header with matching arity
    assert set(_lib.EXPORTS) == set(d), set(_lib.EXPORTS) ^ set(d)
    for name, nargs in d.items():
        assert len(getattr(_lib.lib, name).argtypes) == nargs, name


def test_error_channel_without_gpu():
    from pta_replicator_amd import _lib
    assert _lib.lib.pta_abi_version() == 8
    rc = _lib.lib.pta_quantize_epochs(None, 0, 1.0, None, None, None, None)
    assert rc == -1 and "NULL" in _lib.last_error()
    with pytest.raises(_lib.PtaError):
        _lib.call("pta_quantize_epochs", None, 0, 1.0, None, None, None, None)


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under pta_replicator_amd/ may reference it."""
    pkg = os.path.join(ROOT, "pta_replicator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle/" not in text.replace("oracle/pta_oracle.py", "").replace("(oracle/", "("), f


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: the header must compile as C99 (and as C++) without any HIP / torch include"""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "pta_replicator_amd.h"\nint main(void){ pta_engine_plan p; pta_engine_tables t; pta_td_plan q; (void)p; (void)t; (void)q; return PTA_OK; }\n')
    for cc, flags in (("gcc", ["-std=c99", "-pedantic"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(cc + " not available")
        r = subprocess.run([cc, *flags, "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(root, "include"), str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
