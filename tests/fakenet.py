"""Deterministic stand-in for policy_value_network.forward used by the search parity tests.

Exact integer arithmetic only (uint64 wrap-around hashing, 16-bit mantissas), so the
same (planes -> logits, value) map is bit-reproducible on any host: it is fed to the
unmodified reference (when generating tests/golden), to the C oracle and to the HIP
engine alike.  positions: [B,9,10,14] float32 one-hot planes (what generate_inputs
builds, main.py:531) -> (logits [B,2086] f32, value [B,1] f32) like
policy_value_network.forward (policy_value_network.py:202-214).
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(t):
    t = t ^ (t >> np.uint64(30))
    t = t * _M1
    t = t ^ (t >> np.uint64(27))
    t = t * _M2
    return t ^ (t >> np.uint64(31))


with np.errstate(over="ignore"):
    _KEYS = _mix((np.arange(1, 1261, dtype=np.uint64)) * _G)
    _J = np.arange(2086, dtype=np.uint64) * _G


def position_key(positions):
    p = np.asarray(positions)
    bits = (p.reshape(p.shape[0], -1) > 0.5)
    with np.errstate(over="ignore"):
        return (bits.astype(np.uint64) * _KEYS[None, :]).sum(axis=1, dtype=np.uint64)


_DEEP_M = ((8 - np.arange(4)) / 8.0).astype(np.float32)   # 1.0, 0.875, 0.75, 0.625: exact binary fractions


def make_forward(mode="pos", salt=0, log=None):
    """mode 'pos': logits in [0,1) ; 'signed': logits in [-0.5,0.5) (exercises quirk Q3:
    raw-logit priors that can be negative); 'deep': every fourth label halves the logit — the labels are ranked by
    their hash and logit = (1 - (rank % 4) / 8) * 2^-(rank / 4) (exponent clamped at -120: no denormals), value
    scaled by 1/64 — so one legal move holds nearly all of a node's prior mass and a search digs one line far below
    depth 32 (the long-path backups of the device search)."""
    salt = np.uint64(salt)
    if mode == "deep":
        def forward_deep(positions):
            positions = np.asarray(positions, dtype=np.float32)
            if positions.ndim == 3:
                positions = positions[None]
            h = position_key(positions) ^ salt
            if log is not None:
                log.extend(int(x) for x in h)
            with np.errstate(over="ignore"):
                t = _mix(h[:, None] + _J[None, :])
                order = np.argsort(t, axis=1, kind="stable")
                rank = np.empty_like(order)
                np.put_along_axis(rank, order, np.arange(2086, dtype=order.dtype)[None, :].repeat(t.shape[0], 0), axis=1)
                logits = np.ldexp(_DEEP_M[rank % 4], -np.minimum(rank // 4, 120).astype(np.int32)).astype(np.float32)
                v = (((_mix(h + np.uint64(12345)) >> np.uint64(48)).astype(np.float32) / np.float32(32768.0)) - np.float32(1.0)) * np.float32(1.0 / 64.0)
            return logits, v.astype(np.float32).reshape(-1, 1)
        return forward_deep

    def forward(positions):
        positions = np.asarray(positions, dtype=np.float32)
        if positions.ndim == 3:
            positions = positions[None]
        h = position_key(positions) ^ salt
        if log is not None:
            log.extend(int(x) for x in h)
        with np.errstate(over="ignore"):
            t = _mix(h[:, None] + _J[None, :])
            logits = (t >> np.uint64(48)).astype(np.float32) / np.float32(65536.0)
            v = ((_mix(h + np.uint64(12345)) >> np.uint64(48)).astype(np.float32) / np.float32(32768.0)) - np.float32(1.0)
        if mode == "signed":
            logits = logits - np.float32(0.5)
        return logits.astype(np.float32), v.astype(np.float32).reshape(-1, 1)

    return forward
