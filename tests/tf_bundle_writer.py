"""TEST FIXTURE BUILDER — writes a TensorFlow V2 checkpoint ("tensor bundle": prefix.index + prefix.data-00000-of-00001)
byte by byte from the published format, independently of cchess_zero_amd/tf_checkpoint.py (the reader under test):

  tensorflow/core/util/tensor_bundle/tensor_bundle.cc   BundleWriter: entries sorted by key, header under the key ""
  tensorflow/core/lib/io/{table_builder,block_builder,format}.cc   LevelDB-style table: data blocks with prefix-compressed
      keys and restart points every 16 entries, an index block of (separator key -> BlockHandle), an empty metaindex
      block, 5-byte block trailers (compression type + masked crc32c), 48-byte footer ending in the table magic
  tensorflow/core/protobuf/tensor_bundle.proto, framework/tensor_shape.proto, framework/types.proto

TensorFlow is not installable in this image, so this is the closest thing to a real checkpoint the reader can be pinned
to; the variable NAMES and shapes used by the tests are the reference graph's (policy_value_network.py:45-74,151-162 built
without variable scopes: conv2d, conv2d_1, ..., BatchNorm, ..., fully_connected, ...; plus Momentum slots and global_step).
"""
import struct

import numpy as np

_DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def _crc_byte(b):
    for _ in range(8):
        b = (b >> 1) ^ (0x82F63B78 if b & 1 else 0)
    return b


_T8 = [_crc_byte(i) for i in range(256)]


def _crc32c(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), byte-wise table."""
    crc = 0xFFFFFFFF
    for b in data:
        crc = _T8[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked(data):
    c = _crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_varint(field, v):
    return _vi(field << 3) + _vi(v if v >= 0 else v + (1 << 64))


def _pb_bytes(field, b):
    return _vi((field << 3) | 2) + _vi(len(b)) + b


def _entry_proto(arr, offset, with_crc=True):
    shape = b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in arr.shape)
    raw = arr.tobytes()
    e = _pb_varint(1, _DT[arr.dtype]) + _pb_bytes(2, shape)        # dtype, shape (shard_id 0 is the proto default: omitted)
    if offset:
        e += _pb_varint(4, offset)
    e += _pb_varint(5, len(raw))
    if with_crc:
        e += _vi((6 << 3) | 5) + struct.pack("<I", _masked(raw))
    return e


def _header_proto():
    return _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))        # num_shards = 1, version { producer: 1 }


class _Block:
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.n, self.last, self.ri = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.n % self.ri == 0:
            if self.n:
                self.restarts.append(len(self.buf))
        else:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _vi(shared) + _vi(len(key) - shared) + _vi(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _snappy_literals(data):
    """A valid raw-snappy stream made of literal elements only (what the reader's decompressor must accept)."""
    out = bytearray(_vi(len(data)))
    pos = 0
    while pos < len(data):
        chunk = data[pos:pos + 60]
        out.append((len(chunk) - 1) << 2)
        out += chunk
        pos += len(chunk)
    return bytes(out)


def write_bundle(prefix, tensors, block_size=512, snappy_blocks=False, with_crc=True):
    """tensors: {name: ndarray}.  Small block_size on purpose: the table then has many data blocks and restart points."""
    names = sorted(tensors)
    data = bytearray()
    items = [(b"", _header_proto())]
    for n in names:
        a = np.asarray(tensors[n], order="C")     # (ascontiguousarray would turn a scalar into shape (1,))
        items.append((n.encode(), _entry_proto(a, len(data), with_crc)))
        data += a.tobytes()
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))

    f = bytearray()

    def emit(block_bytes):
        ctype = 1 if snappy_blocks else 0
        body = _snappy_literals(block_bytes) if snappy_blocks else block_bytes
        off = len(f)
        f.extend(body)
        f.append(ctype)
        f.extend(struct.pack("<I", _masked(body + bytes([ctype]))))
        return off, len(body)

    index = _Block(restart_interval=1)
    blk = _Block()
    for key, value in items:
        blk.add(key, value)
        if len(blk.buf) >= block_size:
            off, size = emit(blk.finish())
            index.add(blk.last, _vi(off) + _vi(size))
            blk = _Block()
    if blk.n:
        off, size = emit(blk.finish())
        index.add(blk.last, _vi(off) + _vi(size))
    moff, msize = emit(_Block().finish())
    ioff, isize = emit(index.finish())
    footer = _vi(moff) + _vi(msize) + _vi(ioff) + _vi(isize)
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    f.extend(footer)
    open(prefix + ".index", "wb").write(bytes(f))


def reference_graph_variables(res_block_nums, rng, with_slots=True, global_step=4321):
    """Arrays under the names a tf.train.Saver of the reference graph writes (TF1 creation-order suffixes, HWIO kernels,
    [in,out] FC weights, BatchNorm without beta / gamma), written out literally — not through the package's name map."""
    v = {}
    n_conv = 1 + 2 * res_block_nums + 2
    for i in range(n_conv):
        sfx = "" if i == 0 else "_%d" % i
        if i == 0:
            shp = (3, 3, 14, 128)
        elif i == n_conv - 2:
            shp = (1, 1, 128, 2)           # policy head conv, policy_value_network.py:57
        elif i == n_conv - 1:
            shp = (1, 1, 128, 1)           # value head conv, :66
        else:
            shp = (3, 3, 128, 128)
        v["conv2d%s/kernel" % sfx] = (rng.standard_normal(shp) * 0.05).astype(np.float32)
        v["conv2d%s/bias" % sfx] = (rng.standard_normal(shp[3]) * 0.05).astype(np.float32)
        v["BatchNorm%s/moving_mean" % sfx] = (rng.standard_normal(shp[3]) * 0.05).astype(np.float32)
        v["BatchNorm%s/moving_variance" % sfx] = (rng.random(shp[3]) * 0.5 + 0.75).astype(np.float32)
    for j, (i_, o_) in enumerate(((180, 2086), (90, 256), (256, 1))):
        sfx = "" if j == 0 else "_%d" % j
        v["fully_connected%s/weights" % sfx] = (rng.standard_normal((i_, o_)) * 0.05).astype(np.float32)
        v["fully_connected%s/biases" % sfx] = (rng.standard_normal(o_) * 0.05).astype(np.float32)
    if with_slots:
        for k in [k for k in v if k.split("/")[-1] in ("kernel", "bias", "weights", "biases")]:
            v[k + "/Momentum"] = (rng.standard_normal(v[k].shape) * 0.01).astype(np.float32)
    v["global_step"] = np.asarray(global_step, np.int32)
    return v
