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


def test_hot_kernels_use_no_scratch(tmp_path):
    """Regression guard: the kernels of the search step must not spill to scratch (private segment 0 bytes).  hipcc has
    twice turned hoisted address arrays and prefetch buffers into scratch traffic that cost 5-10 % without any warning;
    the device assembly is generated here (no GPU needed) and the code-object metadata is checked."""
    import re
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from cchess_zero_amd import build
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cchess_zero_amd", "csrc")
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    hot = {"cz_conv.hip": ["k_tower8_c128", "k_trunk_split_c128", "k_trunk_mx_c128"], "cz_search.hip": ["k_select", "k_expand_backup", "k_advance", "k_root_stats"],
           "cz_heads.hip": ["k_policy_fc", "k_value_fc"], "cz_rules.hip": ["k_movegen", "k_movegen_mask", "k_encode_planes"],
           "cz_selfplay.hip": ["k_sp_choose", "k_sp_adjudicate", "k_sp_flush"]}

    def asm(src):
        out = str(tmp_path / (src + ".s"))
        subprocess.run([build.hipcc()] + flags + ["--offload-device-only", "-S", "-o", out, os.path.join(csrc, src)], check=True,
                       stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        return src, open(out).read()
    with ThreadPoolExecutor(4) as ex:
        texts = dict(ex.map(asm, hot))
    checked = 0
    for src, names in hot.items():
        # metadata entries look like:  .name: <mangled>  ...  .private_segment_fixed_size: N
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", texts[src]):
            if any(n in m.group(1) for n in names):   # k_select also matches k_select_k, k_expand_backup the _k variant
                assert int(m.group(2)) == 0, "%s uses %s bytes of scratch" % (m.group(1), m.group(2))
                checked += 1
    # register budgets that decide occupancy (MI355X_MICROARCH.md: <= 168 VGPRs for 3 waves per SIMD, <= 64 for 8): round 5 lost 26 %
    # of k_movegen_mask to nine registers (163 -> 172) without any test noticing
    budgets = {"k_movegen_mask": 168, "k_movegen_listILb0": 168, "k_movegen_listILb1": 256, "k_trunk_mx_c128": 256, "k_trunk_split_c128": 256, "k_tower8_c128": 256}
    seen = set()
    for src in ("cz_rules.hip", "cz_conv.hip"):
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", texts[src]):
            for name, lim in budgets.items():
                if name in m.group(1):
                    assert int(m.group(2)) <= lim, "%s uses %s VGPRs (budget %d)" % (m.group(1), m.group(2), lim)
                    seen.add(name)
    assert seen == set(budgets), seen
    # 2 + 2 + 1 trunk instantiations; 4 select + 2 select_k + 3 expand + 2 expand_k + advance + root_stats; 2 heads; 3 rules; 3 self-play
    assert checked >= 5 + 13 + 2 + 3 + 3, checked   # (k_movegen also matches k_movegen_mask)


def test_generated_slab_asm_is_in_sync(tmp_path):
    """cchess_zero_amd/csrc/cz_tower_slab_asm.inc, cz_trunk_split_asm.inc and cz_trunk_mx_asm.inc are generated
    (tools/gen_tower_asm.py: the hand-scheduled slab bodies of the three trunk kernels): the committed files must be what the
    generator writes (with none of its experiment knobs set)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("MX_ABLATE", "MX_DMA_PLACE")}
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tower_asm.py"), str(tmp_path)], check=True, stdout=subprocess.DEVNULL, env=env)
    for f in ("cz_tower_slab_asm.inc", "cz_trunk_split_asm.inc", "cz_trunk_mx_asm.inc"):
        assert open(str(tmp_path / f)).read() == open(os.path.join(ROOT, "cchess_zero_amd", "csrc", f)).read(), f


def test_trunk_kernel_layout_constants_match_the_emulation():
    """k_tower8_c128's class-tiled cell order (cz_conv_kernel.h): the skip tables in the kernel source are the ones the emulation
    derives from the row map (every off-board (tile, tap) pair, nothing else), the row map is a bijection with the kernel's
    decode as its inverse, and with the kernel's lane relabelling every ds_read_b128 lane group is conflict-free except the one
    that shares tile 0 with the padding rows (tools/experiments/trunk_layout_emulation.py)."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("trunk_layout_emulation", os.path.join(ROOT, "tools", "experiments", "trunk_layout_emulation.py"))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    conflicts, skippable = emu.analyse()
    assert len(skippable) == 15
    assert {t for (t, _, _) in conflicts} <= {0}
    tabs = emu.skiptab_of(skippable)
    src = open(os.path.join(ROOT, "cchess_zero_amd", "csrc", "cz_conv_kernel.h")).read()
    m = re.search(r"skiptab = __builtin_amdgcn_readfirstlane\(wr < 2 \? (0x[0-9A-Fa-f]+) : \(wr == 2 \? (0x[0-9A-Fa-f]+) : (0x[0-9A-Fa-f]+)\)\)", src)
    assert m, "skiptab expression not found in cz_conv_kernel.h"
    assert [tabs[0], tabs[1], tabs[2], tabs[3]] == [int(m.group(1), 16), int(m.group(1), 16), int(m.group(2), 16), int(m.group(3), 16)]
    # the kernel's row map and key function, literally
    assert "return x < 8 ? 8 + 8 * p + x : 2 * p + (x - 8);" in src and "return 8 * ((y + p) & 1) + ((x + y) & 7);" in src
    assert "l31 < 4 ? l31 : l31 < 12 ? l31 + 12 : l31 < 16 ? l31 - 8 : l31 < 20 ? l31 + 8 : l31 < 28 ? l31 - 12 : l31" in src


def test_mx_pack_layer_matches_the_emulation_cpu():
    """cchess_zero_amd.net.mx_pack_layer (the weight slabs of k_trunk_mx_c128, include/cchess_hip.h: cz_net_trunk_mx) decoded
    block by block — 32 six-bit E2M3 slots, the E8M0 byte with the 2^-11 of the lo halves folded in — equals the weights
    tests/mxemu.py's CPU emulation of the kernel multiplies with; the fp16 hi part is the strict engine's hi layout; and the
    emulation of a whole 3-block net holds north_star's 1e-3 against the float64 graph (the design's precision, pinned on the
    CPU: 7.3e-5 / 1.2e-4 measured)."""
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mxemu
    import nethelpers as H
    from cchess_zero_amd.net import PolicyValueModule, mx_pack_layer

    def dq(c):
        s, e, m = c >> 5, (c >> 3) & 3, c & 7
        v = (1 + m / 8.0) * 2.0 ** (e - 1) if e else m / 8.0
        return -v if s else v
    torch.manual_seed(0)
    with torch.no_grad():
        m = PolicyValueModule(3, seed=1)

        class _N:
            module, refresh = m, staticmethod(lambda: None)
        H.trained_like_(_N)
        w, _ = m.blocks[1][1].folded()
        pk = mx_pack_layer(w).numpy().reshape(9, 4, 16384)
        w_hi, w_lo = mxemu.split16(w)
        wl6, wh6 = mxemu.mxq_pair(w_lo.to(torch.float16).float() * 2048.0, w_hi, 1)
        rng = np.random.default_rng(3)
        for _ in range(200):
            tap, Q, h, co = int(rng.integers(9)), int(rng.integers(4)), int(rng.integers(2)), int(rng.integers(128))
            sl = pk[tap, Q]
            o = h * 128 + co
            blk = int.from_bytes(sl[8192 + o * 16:8192 + o * 16 + 16].tobytes() + sl[12288 + o * 8:12288 + o * 8 + 8].tobytes(), "little")
            sc = 2.0 ** (int(sl[14336 + o * 4]) + 11 - 127)
            assert not sl[14336 + o * 4 + 1:14336 + o * 4 + 4].any() and not sl[15360:].any()
            for j in range(16):
                ci = 32 * Q + 8 * (j // 4) + 4 * h + j % 4
                assert dq((blk >> (12 * j)) & 63) * sc == float(wl6[co, ci, tap // 3, tap % 3]), (tap, Q, h, co, j)
                assert dq((blk >> (12 * j + 6)) & 63) * sc == float(wh6[co, ci, tap // 3, tap % 3]), (tap, Q, h, co, j)
        hi = np.frombuffer(pk[5, 1][:8192].tobytes(), np.float16).reshape(4, 128, 8).astype(np.float32)
        assert np.array_equal(hi, w_hi[:, 32:64, 1, 2].reshape(128, 4, 8).permute(1, 0, 2).numpy())
        x = torch.from_numpy(H.positions(24, 2)).permute(0, 3, 1, 2).contiguous()
        l, v, _ = mxemu.forward_mx(m, x)
        m64 = PolicyValueModule(3, seed=1).double()
        m64.load_state_dict({k: t.double() for k, t in m.state_dict().items()})
        l64, v64 = m64(x.double())
        dl, dv = float((l.double() - l64).abs().max()), float((v.double() - v64).abs().max())
        print("mx emulation, 3 blocks trained-like: dlogit %.3g dvalue %.3g (max |logit| %.3g)" % (dl, dv, float(l64.abs().max())))
        assert dl <= 1e-3 and dv <= 1e-3 and float(l64.abs().max()) > 5
