"""ctypes binding of libcchess_hip.so (C-ABI: include/cchess_hip.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CCHESS_HIP_LIB: another build of the SAME library (same-box A/B of kernel changes, tools/ab_lib.sh); default: in-tree
LIB_PATH = os.environ.get("CCHESS_HIP_LIB") or os.path.join(_HERE, "libcchess_hip.so")

NLABELS = 2086
MAXMOVES = 128
NSQ = 90
MASK_WORDS = 66
F32, BF16, F16 = 0, 1, 2
# one packed self-play record (include/cchess_hip.h: CZ_REC_*)
REC_BYTES, REC_SIDE, REC_COUNT, REC_Z, REC_FLAGS, REC_PLY, REC_LABELS, REC_VISITS = 608, 90, 91, 92, 93, 94, 96, 352
SP_STATS = ("games", "red_wins", "black_wins", "draws", "plies", "stalled", "dropped", "sims")

_u8p, _u16p, _i32p, _f32p, _vp = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p

_SIGS = {
    "cz_last_error": (C.c_char_p, []),
    "cz_version": (C.c_int, []),
    "cz_crc32c": (C.c_uint, [C.c_char_p, C.c_size_t]),
    "cz_tables": (C.c_int, [C.POINTER(C.c_void_p)] * 4),
    "cz_zobrist": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "cz_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cz_destroy": (None, [C.c_void_p]),
    "cz_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cz_synchronize": (C.c_int, [C.c_void_p]),
    "cz_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "cz_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cz_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cz_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "cz_movegen": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, _u16p, _u16p, _vp]),
    "cz_movegen_ex": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, _u16p, _u16p, _vp, C.c_int]),
    "cz_apply_move": (C.c_int, [C.c_void_p, _u8p, _u8p, _u16p, C.c_int, _vp, _u8p, _vp]),
    "cz_hash": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, _vp]),
    "cz_encode_planes": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_int, _vp, C.c_int, C.c_int, C.c_int]),
    "cz_search_reset": (C.c_int, [C.c_void_p, _u8p, _u8p, _i32p, C.c_int]),
    "cz_search_set_eval_cache": (C.c_int, [C.c_void_p, C.c_int]),
    "cz_search_eval_cache_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "cz_search_eval_cache_collisions": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "cz_search_debug_eval_cache_key_bits": (C.c_int, [C.c_void_p, C.c_int]),
    "cz_search_debug_advance_in_global_memory": (C.c_int, [C.c_void_p, C.c_int]),
    "cz_search_set_xcache": (C.c_int, [C.c_void_p, C.c_int]),
    "cz_search_xcache_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "cz_search_xcache_stats5": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong)]),
    "cz_search_reload": (C.c_int, [C.c_void_p, _u8p, _u8p, _u8p, _i32p]),
    "cz_search_select": (C.c_int, [C.c_void_p, C.c_int, _u8p, _vp, C.c_int, C.c_int, _u8p]),
    "cz_search_expand_backup": (C.c_int, [C.c_void_p, _vp, _vp, C.c_int]),
    "cz_search_expand_backup_fc": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, C.c_int]),
    "cz_search_select_compact": (C.c_int, [C.c_void_p, C.c_int, _vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "cz_set_batch_count": (C.c_int, [C.c_void_p, _vp]),
    "cz_probe_mfma_peak": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "cz_set_clock_probe": (C.c_int, [C.c_void_p, _vp, C.c_int]),
    "cz_clock_probe_last_grid": (C.c_int, [C.c_void_p]),
    "cz_search_eval_totals": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "cz_search_set_width": (C.c_int, [C.c_void_p, C.c_int]),
    "cz_search_set_sim_target": (C.c_int, [C.c_void_p, C.c_int]),
    "cz_search_set_terminal_extra": (C.c_int, [C.c_void_p, C.c_int]),
    "cz_search_select_k": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _u8p, _vp, C.c_int, C.c_int, _u8p]),
    "cz_search_expand_backup_k": (C.c_int, [C.c_void_p, C.c_int, _vp, _vp, C.c_int]),
    "cz_search_root_stats": (C.c_int, [C.c_void_p, _u16p, _i32p, _f32p, _f32p, _f32p, _u16p]),
    "cz_search_advance": (C.c_int, [C.c_void_p, _u16p]),
    "cz_search_pick_ready": (C.c_int, [C.c_void_p, _i32p, C.c_int, _u16p, _u8p, _vp]),
    "cz_search_reload_finished": (C.c_int, [C.c_void_p, _u8p, _u16p, _u8p, _u8p, _i32p, _vp]),
    "cz_search_status": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p]),
    "cz_search_root_state": (C.c_int, [C.c_void_p, _u8p, _u8p, _i32p]),
    "cz_search_tree_dump": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "cz_selfplay_begin": (C.c_int, [C.c_void_p, C.c_int, _u8p, _u8p, _i32p]),
    "cz_selfplay_active": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cz_selfplay_choose": (C.c_int, [C.c_void_p, _f32p, _f32p, _u16p, C.c_double, C.c_float, C.c_int, _u16p]),
    "cz_selfplay_adjudicate": (C.c_int, [C.c_void_p, C.c_int, _u16p, _i32p]),
    "cz_selfplay_flush": (C.c_int, [C.c_void_p, _i32p, _vp, _u8p, C.c_longlong, _vp]),
    "cz_selfplay_stats": (C.c_int, [C.c_void_p, _vp]),
    "cz_conv3x3_c128_bf16": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]),
    "cz_tower_c128_bf16": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, C.c_int, C.c_int]),
    "cz_net_trunk_bf16": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]),
    "cz_net_trunk_f16": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]),
    "cz_net_trunk_split": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "cz_net_trunk_mx": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]),
    "cz_fc_heads_f32": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "cz_tower_heads_c128_bf16": (C.c_int, [C.c_void_p, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int]),
}

EXPORTS = tuple(sorted(_SIGS))
_lib = None


class CchessHipError(RuntimeError):
    pass


def lib():
    """Loads libcchess_hip.so (built by `python -m cchess_zero_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CchessHipError(
                "libcchess_hip.so is missing at %s — build it with __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path." % LIB_PATH)
        # libcchess_hip.so must share ONE HIP runtime with PyTorch (device memory and streams are
        # torch's): import torch first so that its bundled libamdhip64 is the one already mapped.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().cz_last_error()
        raise CchessHipError("%s failed (%d): %s" % (what or "cchess_hip call", rc, msg.decode() if msg else "?"))


_tables = None


def tables():
    """Host copies of the static tables: dict(lut[90,90] i16, unflip[2086] i16, labels[2086] str, srcdst[2086] u16)."""
    global _tables
    if _tables is None:
        L = lib()
        p = [C.c_void_p() for _ in range(4)]
        check(L.cz_tables(*[C.byref(x) for x in p]), "cz_tables")
        lut = np.ctypeslib.as_array(C.cast(p[0], C.POINTER(C.c_int16)), shape=(NSQ * NSQ,)).reshape(NSQ, NSQ).copy()
        unflip = np.ctypeslib.as_array(C.cast(p[1], C.POINTER(C.c_int16)), shape=(NLABELS,)).copy()
        raw = C.string_at(p[2], NLABELS * 5)
        labels = [raw[i * 5:i * 5 + 4].decode() for i in range(NLABELS)]
        srcdst = np.ctypeslib.as_array(C.cast(p[3], C.POINTER(C.c_uint16)), shape=(NLABELS,)).copy()
        zk = C.c_void_p()
        sk = C.c_uint64()
        check(L.cz_zobrist(C.byref(zk), C.byref(sk)), "cz_zobrist")
        zob = np.ctypeslib.as_array(C.cast(zk, C.POINTER(C.c_uint64)), shape=(15 * NSQ,)).reshape(15, NSQ).copy()
        _tables = dict(lut=lut, unflip=unflip, labels=labels, srcdst=srcdst, zobrist=zob, zobrist_side=int(sk.value),
                       label2i={s: i for i, s in enumerate(labels)})
    return _tables
