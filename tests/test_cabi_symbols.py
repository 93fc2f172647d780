"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/rp_engine.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rp_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from resource_packing_self_play_amd import _lib
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), "librp_engine.so does not export %s" % n
    assert set(names) == set(_lib._SIGS), "ctypes table and header disagree: %s" % (set(names) ^ set(_lib._SIGS))
    assert L.rp_version() == _lib.ABI_VERSION


def test_config_struct_layout_matches_header(tmp_path):
    """The ctypes mirror of rp_config against the C compiler's view of include/rp_engine.h: size and the offset of every field."""
    import subprocess
    from resource_packing_self_play_amd import _lib
    fields = [f[0] for f in _lib.RpConfig._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rp_engine.h"\nint main(void) {\n  printf("%zu\\n", sizeof(rp_config));\n'
                   + "".join('  printf("%s %%zu\\n", offsetof(rp_config, %s));\n' % (f, f) for f in fields) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    assert int(out[0]) == ctypes.sizeof(_lib.RpConfig)
    for line in out[1:]:
        if line.strip():
            name, off = line.split()
            assert getattr(_lib.RpConfig, name).offset == int(off), name
    # int32 x6, double x2, int32 x6, uint64 x2, int32 x2, void*, int64 x2
    assert ctypes.sizeof(_lib.RpConfig) == 6 * 4 + 2 * 8 + 6 * 4 + 2 * 8 + 2 * 4 + 8 + 2 * 8
    text = open(os.path.join(ROOT, "include", "rp_engine.h")).read()
    assert int(re.search(r"#define RP_ABI_VERSION (\d+)", text).group(1)) == _lib.ABI_VERSION


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from resource_packing_self_play_amd import _lib
    with pytest.raises(_lib.EngineError) as ei:
        _lib.Engine(10, 10, 8, 1, 25)
    assert ei.value.code == _lib.ERR_DEVICE
    assert "no CPU fallback" in str(ei.value)
