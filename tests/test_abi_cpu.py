"""CPU-side checks of the C-ABI: the library loads and exports every symbol include/cchess_hip.h
declares; host tables (no GPU needed) equal the oracle's and the reference pins."""
import hashlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cchess_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cz_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    import __graft_entry__  # noqa: F401  (build() makes sure the .so exists)
    __graft_entry__.build_hip_only()
    import ctypes
    from cchess_zero_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(L, n), "symbol %s declared in include/cchess_hip.h is not exported" % n
    # and the Python binding covers exactly the declared surface
    assert sorted(_lib.EXPORTS) == names


def test_host_tables_match_reference_pins(tables_golden):
    from cchess_zero_amd import _lib
    from oracle import oracle as O
    t = _lib.tables()
    assert len(t["labels"]) == 2086
    assert hashlib.sha256("\n".join(t["labels"]).encode()).hexdigest() == tables_golden["labels_sha256"]
    assert hashlib.sha256(t["unflip"].astype(np.int16).tobytes()).hexdigest() == tables_golden["unflip_sha256"]
    for k, v in tables_golden["label2i"].items():
        assert t["label2i"][k] == v
    assert np.array_equal(t["lut"], O.lut())
    assert np.array_equal(t["srcdst"], O.label_srcdst())
    zt, zs = O.zobrist_table()
    assert np.array_equal(t["zobrist"][1:], zt[1:]) and t["zobrist_side"] == zs


def test_missing_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cchess_zero_amd import _lib, engine
    with pytest.raises(_lib.CchessHipError):
        engine.Context(4, 16)
