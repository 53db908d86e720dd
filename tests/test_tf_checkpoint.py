"""SURVEY §8 row f2 — the reference's own checkpoints (tf.train.Saver, policy_value_network.py:148,164-184) read without
TensorFlow: cchess_zero_amd/tf_checkpoint.py against bundles written byte by byte from the published format by
tests/tf_bundle_writer.py (an independent implementation; TensorFlow itself is not installable here), with the reference
graph's variable names written out literally."""
import os
import sys

import numpy as np
import pytest
import torch

import tf_bundle_writer as W
from cchess_zero_amd import tf_checkpoint as T
from oracle import net_numpy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _manual_layout(v, blocks):
    """TF names -> the keys oracle/net_numpy.py consumes, spelled out here (not through the package's name map):
    conv2d[_i] / BatchNorm[_i] in creation order = input conv, two per block, policy head, value head;
    fully_connected, _1, _2 = policy FC, value FC1, value FC2 (policy_value_network.py:45-74)."""
    d = {}
    for i in range(1 + 2 * blocks + 2):
        sfx = "" if i == 0 else "_%d" % i
        d["conv%d/kernel" % i] = v["conv2d%s/kernel" % sfx]
        d["conv%d/bias" % i] = v["conv2d%s/bias" % sfx]
        d["bn%d/moving_mean" % i] = v["BatchNorm%s/moving_mean" % sfx]
        d["bn%d/moving_variance" % i] = v["BatchNorm%s/moving_variance" % sfx]
    for ours, tfn in (("policy_fc", "fully_connected"), ("value_fc1", "fully_connected_1"), ("value_fc2", "fully_connected_2")):
        d[ours + "/weights"] = v[tfn + "/weights"]
        d[ours + "/biases"] = v[tfn + "/biases"]
    return d


@pytest.mark.parametrize("snappy,block_size,with_crc", [(False, 512, True), (True, 300, True), (False, 1 << 20, False)])
def test_bundle_reader_roundtrip(tmp_path, snappy, block_size, with_crc):
    v = W.reference_graph_variables(2, np.random.default_rng(1))
    v["a_double"] = np.linspace(0, 1, 7)                   # float64
    v["an_int64_matrix"] = np.arange(12, dtype=np.int64).reshape(3, 4) - 5
    p = str(tmp_path / "best_model.ckpt-77")
    W.write_bundle(p, v, block_size=block_size, snappy_blocks=snappy, with_crc=with_crc)
    got = T.read_checkpoint(p, verify_crc=True)
    assert sorted(got) == sorted(v)
    for k in v:
        assert got[k].dtype == v[k].dtype and got[k].shape == np.shape(v[k]) and np.array_equal(got[k], v[k]), k
    lv = dict((n, (s, d)) for n, s, d in T.list_variables(p + ".index"))
    assert lv["conv2d/kernel"] == ((3, 3, 14, 128), np.float32) and lv["global_step"] == ((), np.int32)
    assert T.is_tf_checkpoint(p) and T.is_tf_checkpoint(p + ".data-00000-of-00001") and not T.is_tf_checkpoint(str(tmp_path / "nope"))
    # the `checkpoint` state file the Saver keeps (get_checkpoint_state, policy_value_network.py:165)
    assert T.latest_checkpoint(str(tmp_path)) is None
    open(tmp_path / "checkpoint", "w").write('model_checkpoint_path: "best_model.ckpt-77"\nall_model_checkpoint_paths: "best_model.ckpt-77"\n')
    assert T.latest_checkpoint(str(tmp_path)) == p


def test_bundle_reader_rejects_damage(tmp_path):
    v = {"w": np.arange(100, dtype=np.float32), "global_step": np.asarray(3, np.int32)}
    p = str(tmp_path / "m")
    W.write_bundle(p, v)
    idx = bytearray(open(p + ".index", "rb").read())
    bad = bytearray(idx)
    bad[-1] ^= 0xFF                                          # table magic
    open(p + ".index", "wb").write(bytes(bad))
    with pytest.raises(T.CheckpointError, match="magic"):
        T.read_checkpoint(p)
    bad = bytearray(idx)
    bad[3] ^= 0x01                                           # a byte of the first data block: block checksum
    open(p + ".index", "wb").write(bytes(bad))
    with pytest.raises(T.CheckpointError, match="checksum"):
        T.read_checkpoint(p)
    open(p + ".index", "wb").write(bytes(idx))
    dat = bytearray(open(p + ".data-00000-of-00001", "rb").read())
    dat[17] ^= 0x40                                          # a tensor byte: tensor checksum
    open(p + ".data-00000-of-00001", "wb").write(bytes(dat))
    with pytest.raises(T.CheckpointError, match="tensor checksum"):
        T.read_checkpoint(p)
    open(p + ".data-00000-of-00001", "wb").write(bytes(dat[:-8]))   # truncated shard
    with pytest.raises(T.CheckpointError):
        T.read_checkpoint(p, verify_crc=False)
    os.remove(p + ".data-00000-of-00001")
    with pytest.raises(T.CheckpointError, match="missing"):
        T.read_checkpoint(p)


@pytest.mark.parametrize("blocks", [2, 7])
def test_reference_checkpoint_loads_by_tf1_variable_names(tmp_path, blocks):
    """A checkpoint carrying the reference graph's TF1 variable names -> PolicyValueModule through from_tf_variables /
    load_tf_layout: its forward equals the NumPy restatement of the TF graph fed the same arrays through a name table
    spelled out in THIS file; block count and global_step are recovered; the Momentum slots land in the optimiser so
    that the next training step continues the checkpoint's trajectory (float64 restatement of MomentumOptimizer)."""
    import nethelpers as H
    from cchess_zero_amd.net import PolicyValueModule, from_tf_variables, momentum_slots_from_tf_variables
    from cchess_zero_amd.train import Trainer
    v = W.reference_graph_variables(blocks, np.random.default_rng(blocks), with_slots=(blocks == 2))
    p = str(tmp_path / "best_model.ckpt-4321")
    W.write_bundle(p, v, block_size=4096)
    ck = T.read_checkpoint(p)
    d, nb, gs = from_tf_variables(ck)
    assert nb == blocks and gs == 4321
    m = PolicyValueModule(blocks, seed=99)
    m.load_tf_layout(d)
    x = H.positions(6, 3)
    with torch.no_grad():
        lt, vt = m(torch.from_numpy(x).permute(0, 3, 1, 2))
    ln, vn = net_numpy.forward(_manual_layout(v, blocks), x, blocks)
    assert np.abs(lt.numpy() - ln).max() < 2e-4 * max(1.0, np.abs(ln).max()) and np.abs(vt.numpy() - vn).max() < 1e-5
    if blocks != 2:
        return
    # momentum slots: one training step from the checkpoint == the float64 restatement started from the same accumulators
    from test_train import _batch, _run_restatement
    slots = momentum_slots_from_tf_variables(ck, nb)
    assert len(slots) == 2 * (1 + 2 * blocks + 2) + 6
    tr = Trainer(m)
    tr.load_tf_momentum(slots)
    w0 = m.export_tf_layout()
    batch = _batch(12, 5)
    manual_slots = _manual_layout({k[:-len("/Momentum")]: a for k, a in v.items() if k.endswith("/Momentum")} |
                                  {k: a for k, a in v.items() if "BatchNorm" in k}, blocks)
    ref = _run_restatement(w0, [batch], 0.1, blocks, accum0={k: a for k, a in manual_slots.items() if not k.startswith("bn")})
    tr.train_step(batch[0], batch[1], batch[2], 0.1)
    now = m.export_tf_layout()
    for k, rw in ref[0][2].items():
        d_got = now[k].astype(np.float64) - w0[k].astype(np.float64)
        d_ref = rw - np.asarray(w0[k], np.float64)
        assert np.abs(d_got - d_ref).max() <= 2e-3 * np.abs(d_ref).max() + 5e-7, k
    # without the slots the step is a different one (the accumulators matter)
    m2 = PolicyValueModule(blocks, seed=99)
    m2.load_tf_layout(d)
    tr2 = Trainer(m2)
    tr2.train_step(batch[0], batch[1], batch[2], 0.1)
    assert np.abs(m2.export_tf_layout()["conv1/kernel"] - now["conv1/kernel"]).max() > 1e-5


@pytest.mark.gpu
def test_facade_restores_a_tf_model_directory(tmp_path):
    """policy_value_network(...) pointed at a directory as the reference's Saver leaves it (checkpoint state file +
    best_model.ckpt-N.index/.data): train_restore picks the TF checkpoint up, forward() answers with its weights, the
    step counter continues from its global_step; restore(prefix) does the same explicitly."""
    import nethelpers as H
    sys.path.insert(0, ROOT)
    from policy_value_network import policy_value_network
    v = W.reference_graph_variables(2, np.random.default_rng(11), global_step=250)
    mdir = tmp_path / "models"
    mdir.mkdir()
    W.write_bundle(str(mdir / "best_model.ckpt-250"), v, block_size=4096)
    open(mdir / "checkpoint", "w").write('model_checkpoint_path: "best_model.ckpt-250"\n')
    pv = policy_value_network(2, save_dir=str(mdir), seed=5)
    assert pv.global_step == 250
    x = H.positions(16, 4)
    logits, value = pv.forward(x)
    ln, vn = net_numpy.forward(_manual_layout(v, 2), x, 2)
    e = H.errors(logits, value, ln, vn)
    assert e["dlogit_rel"] <= 2e-3 and e["dvalue"] <= 2e-3, e       # the default fp16 engine on the checkpoint's weights
    other = tmp_path / "other"
    pv2 = policy_value_network(2, save_dir=str(other), seed=6)
    assert pv2.global_step == 0
    pv2.restore(str(mdir / "best_model.ckpt-250"))
    l2, v2 = pv2.forward(x)
    assert pv2.global_step == 250 and np.array_equal(l2, logits) and np.array_equal(v2, value)
    # the momentum slots are in the optimiser: a step from here changes the weights by lr * (grad + 0.9 * accum) with accum != 0
    buf = pv2.trainer.opt.state[pv2.module.policy_fc.weight]["momentum_buffer"]
    assert torch.equal(buf.cpu(), torch.from_numpy(v["fully_connected/weights/Momentum"]).t())
    # and the way back: export_tf_checkpoint writes what the reference's Saver would — every variable, slot and the step
    # come back bit for bit through the reader, under the same names
    out_dir = tmp_path / "exported"
    prefix = pv2.export_tf_checkpoint(save_dir=str(out_dir))
    assert prefix.endswith("best_model.ckpt-250") and T.latest_checkpoint(str(out_dir)) == prefix
    back = T.read_checkpoint(prefix)
    assert sorted(back) == sorted(v)
    for k in v:
        assert back[k].dtype == v[k].dtype and np.array_equal(back[k], v[k]), k
    # both kinds in one directory: the newest (larger global step) is the model — first a .pt at step 300 beside the TF
    # checkpoint of step 250, then a TF export at step 400 beside that .pt
    pv2.global_step = 300
    pv2.save_dir = str(mdir)
    pv2.save(300)
    assert policy_value_network(2, save_dir=str(mdir), seed=7).global_step == 300
    pv2.global_step = 400
    pv2.export_tf_checkpoint(save_dir=str(mdir))
    assert policy_value_network(2, save_dir=str(mdir), seed=7).global_step == 400


def test_product_writer_equals_the_independent_writer_byte_for_byte(tmp_path):
    """cchess_zero_amd.tf_checkpoint.write_checkpoint (the way back into the reference's TF graph) against the test-side
    writer: two implementations of the same published format must produce the SAME bytes for the same tensors and block size,
    and the native crc32c helper of the library (cz_crc32c, slicing-by-8) must agree with both pure-Python tables."""
    v = W.reference_graph_variables(2, np.random.default_rng(8))
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    T.write_checkpoint(a, v, block_size=512)
    W.write_bundle(b, v, block_size=512)
    for suffix in (".index", ".data-00000-of-00001"):
        assert open(a + suffix, "rb").read() == open(b + suffix, "rb").read(), suffix
    got = T.read_checkpoint(a, verify_crc=True)
    assert all(np.array_equal(got[k], v[k]) for k in v)
    blob = np.random.default_rng(1).integers(0, 256, 100003, dtype=np.uint8).tobytes()
    assert T.crc32c(blob) == W._crc32c(blob)                 # native (len >= 4096) vs the test-side table
    assert T.crc32c(blob[:1000]) == W._crc32c(blob[:1000])   # pure Python in the package vs the test-side table
    assert T.crc32c(b"123456789") == 0xE3069283              # the CRC-32C check value


def test_trainer_momentum_slots_round_trip():
    """Trainer.tf_momentum_slots <-> load_tf_momentum: what export_tf_checkpoint writes as `<variable>/Momentum` is what a
    restore reads back, layouts included; untouched parameters export zero slots like a fresh MomentumOptimizer."""
    import nethelpers as H
    from cchess_zero_amd.net import PolicyValueModule
    from cchess_zero_amd.train import Trainer
    from test_train import _batch
    m = PolicyValueModule(1, seed=2)
    tr = Trainer(m)
    z = tr.tf_momentum_slots()
    assert z["conv1/kernel"].shape == (3, 3, 128, 128) and z["policy_fc/weights"].shape == (180, 2086) and not any(a.any() for a in z.values())
    x, pi, zz = _batch(8, 2)
    tr.train_step(x, pi, zz, 0.05)
    s1 = tr.tf_momentum_slots()
    assert any(a.any() for a in s1.values())
    m2 = PolicyValueModule(1, seed=2)
    tr2 = Trainer(m2)
    tr2.load_tf_momentum(s1)
    s2 = tr2.tf_momentum_slots()
    assert all(np.array_equal(s1[k], s2[k]) for k in s1)
    for p, q in zip(m.parameters(), m2.parameters()):
        assert torch.equal(tr.opt.state[p]["momentum_buffer"], tr2.opt.state[q]["momentum_buffer"])


# ---- known answers from the PUBLISHED specifications (not from this repo's own writer) ---------------------------------------------
# The reader above is otherwise checked against tests/tf_bundle_writer.py, written by the same author from the same reading of the
# format.  These vectors come from elsewhere: RFC 3720 appendix B.4 (CRC32C of iSCSI, the same polynomial; LevelDB's
# util/crc32c_test.cc "StandardResults" lists the same five), LevelDB's crc32c.h (the mask: rotate right by 15, add 0xa282ead8) and
# table_format.md / coding.cc (varint32, BlockHandle, block trailer, footer magic 0xdb4775248b80fb57).
RFC3720_CRC32C = [
    (bytes(32), 0x8A9136AA),                              # 32 bytes of zeros
    (bytes([0xFF]) * 32, 0x62A8AB43),                     # 32 bytes of ones
    (bytes(range(32)), 0x46DD794E),                       # 32 incrementing bytes
    (bytes(range(31, -1, -1)), 0x113FDB5C),               # 32 decrementing bytes
    (bytes([0x01, 0xC0, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x14, 0x00, 0x00, 0x00, 0x00, 0x00,
            0x04, 0x00, 0x00, 0x00, 0x00, 0x14, 0x00, 0x00, 0x00, 0x18, 0x28, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x02, 0x00, 0x00, 0x00,
            0x00, 0x00, 0x00, 0x00]), 0xD9963A56),        # an iSCSI SCSI Read (10) command PDU
    (b"123456789", 0xE3069283),                           # the CRC catalogue's check value for CRC-32C
]


def test_crc32c_known_answers_rfc3720_and_leveldb_mask():
    for data, want in RFC3720_CRC32C:
        assert T.crc32c(data) == want, (data[:4], hex(T.crc32c(data)), hex(want))
        half = len(data) // 2                               # incremental form: crc(a + b) = crc(b, crc(a))
        assert T.crc32c(data[half:], T.crc32c(data[:half])) == want
    big = bytes(range(256)) * 64                            # >= 4096 bytes: the library's C helper when it is built — same answer
    t = T._crc_table()
    c = 0xFFFFFFFF
    for b in big:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    assert T.crc32c(big) == c ^ 0xFFFFFFFF
    # leveldb::crc32c::Mask: ((crc >> 15) | (crc << 17)) + 0xa282ead8; crc32c_test.cc: Mask(Value("foo")) != Value("foo"), Unmask(Mask(x)) == x
    c = T.crc32c(b"foo")
    m = T.masked_crc32c(b"foo")
    assert m == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF and m != c
    rot = (m - 0xA282EAD8) & 0xFFFFFFFF
    assert ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF == c


def _bitwise_crc32c(data):
    """CRC-32C straight from its definition (reflected polynomial 0x82F63B78, init and final xor 0xFFFFFFFF), one bit at a time."""
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def test_reads_a_bundle_assembled_byte_by_byte_from_the_published_format(tmp_path):
    """A one-tensor V2 checkpoint put together here, byte by byte, from the published format descriptions alone (LevelDB
    table_format.md: blocks of prefix-compressed entries + restart array + 5-byte trailer, footer of two BlockHandles padded to
    40 bytes + the magic; TensorFlow tensor_bundle.proto: BundleHeaderProto under the empty key, BundleEntryProto per tensor,
    checksums masked CRC-32C) with a bit-at-a-time CRC — none of the repo's writers is involved."""
    import struct
    assert all(_bitwise_crc32c(d) == w for d, w in RFC3720_CRC32C)

    def masked(b):
        c = _bitwise_crc32c(b)
        return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    values = np.array([[1.5, -2.0, 0.25], [3.0, 4.5, -8.0]], np.float32)
    data = values.tobytes()                                                     # .data shard: the tensor's bytes, little-endian
    # BundleEntryProto: dtype = 1 (DT_FLOAT) | shape { dim { size: 2 } dim { size: 3 } } | offset 0 omitted | size = 24 | crc32c fixed32
    shape = bytes([0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03])             # TensorShapeProto: field 2 (dim), each dim: field 1 (size)
    entry = bytes([0x08, 0x01, 0x12, len(shape)]) + shape + bytes([0x28, 24, 0x35]) + struct.pack("<I", masked(data))
    header = bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])                        # num_shards = 1, version { producer: 1 }; endianness LITTLE = 0 omitted
    key = b"conv2d/kernel"

    def block(entries):        # every entry a restart point: shared = 0
        body, restarts = b"", []
        for k, v in entries:
            restarts.append(len(body))
            body += bytes([0, len(k), len(v)]) + k + v                          # varint32 shared, non_shared, value_length (< 128: one byte each)
        body += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
        return body + b"\x00" + struct.pack("<I", masked(body + b"\x00"))       # trailer: compression type 0 (none) + masked crc of block + type
    d = block([(b"", header), (key, entry)])
    meta = (struct.pack("<I", 0) + struct.pack("<I", 1))                        # the (empty) metaindex block: one restart at offset 0
    meta = meta + b"\x00" + struct.pack("<I", masked(meta + b"\x00"))

    def handle(offset, size):  # BlockHandle: varint64 offset, varint64 size (size excludes the 5-byte trailer)
        out = b""
        for n in (offset, size):
            while n >= 128:
                out += bytes([(n & 0x7F) | 0x80])
                n >>= 7
            out += bytes([n])
        return out
    index = block([(b"d", handle(0, len(d) - 5))])                              # a separator >= the data block's last key
    footer = handle(len(d), len(meta) - 5) + handle(len(d) + len(meta), len(index) - 5)
    footer += bytes(40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    assert footer[-8:] == bytes([0x57, 0xFB, 0x80, 0x8B, 0x24, 0x75, 0x47, 0xDB]) and len(footer) == 48
    prefix = str(tmp_path / "best_model.ckpt-7")
    open(prefix + ".index", "wb").write(d + meta + index + footer)
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    assert T.is_tf_checkpoint(prefix)
    got = T.read_checkpoint(prefix)
    assert list(got) == ["conv2d/kernel"] and got["conv2d/kernel"].dtype == np.float32
    assert np.array_equal(got["conv2d/kernel"], values)
    bad = bytearray(data)
    bad[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(bad))
    with pytest.raises(T.CheckpointError):
        T.read_checkpoint(prefix)
