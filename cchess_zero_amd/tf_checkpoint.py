"""Reader and writer for TensorFlow checkpoints in the V2 ("tensor bundle") format, pure Python + NumPy — no TensorFlow needed.

The reference saves and restores its weights with tf.train.Saver (policy_value_network.py:148,164-184): files
`<save_dir>/best_model.ckpt-<step>.index` + `.data-00000-of-00001` and a `checkpoint` state file.  TF >= 0.12 writes
the V2 format by default, which is what a cchess-zero model trained with the reference ("TensorFlow 1.0", README) is in.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table — a LevelDB-style sorted string table):
  .index   sequence of blocks + 48-byte footer.  footer = metaindex BlockHandle, index BlockHandle (varint64 offset,
           varint64 size each), zero padding, 8-byte magic 0xdb4775248b80fb57 (little endian).  A block =
           entries + uint32 restart offsets[] + uint32 num_restarts, followed in the file by a 5-byte trailer
           (1 byte compression: 0 none / 1 snappy, 4 bytes masked crc32c of block + type).  An entry = varint32
           shared, varint32 non_shared, varint32 value_len, key suffix, value (keys are prefix-compressed against the
           previous key).  The index block maps separator keys to the BlockHandles of the data blocks.  Data-block keys
           are tensor names; the key "" holds the BundleHeaderProto; values are BundleEntryProto messages:
             1 dtype (enum DataType), 2 shape (TensorShapeProto: 2 dim { 1 size }), 3 shard_id, 4 offset, 5 size,
             6 crc32c (fixed32, masked), 7 slices (partitioned variables: not used by the reference graph)
  .data-SSSSS-of-NNNNN   raw little-endian tensor bytes at [offset, offset + size) of shard shard_id.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


class CheckpointError(ValueError):
    pass


# ---- crc32c (Castagnoli), as LevelDB / TensorFlow mask it -------------------------------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data, crc=0):
    if crc == 0 and len(data) >= 4096:   # long buffers: the library's host-side helper (plain C, ~1 GB/s) when it is there
        try:
            import ctypes
            from . import _lib
            fn = _lib.lib().cz_crc32c
            b = bytes(data)
            return int(fn(ctypes.c_char_p(b), ctypes.c_size_t(len(b)))) & 0xFFFFFFFF
        except Exception:
            pass
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xFFFFFFFF


# ---- varints / protobuf wire format --------------------------------------------------------------------------------------
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise CheckpointError("varint too long")


def _fields(buf):
    """Yields (field number, wire type, value) of one protobuf message; length-delimited values as bytes."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise CheckpointError("truncated protobuf field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    dims = []
    for num, wt, v in _fields(buf):
        if num == 2 and wt == 2:                      # Dim
            size = 0
            for n2, w2, v2 in _fields(v):
                if n2 == 1 and w2 == 0:
                    size = _signed64(v2)
            dims.append(size)
        elif num == 3 and wt == 0 and v:
            raise CheckpointError("tensor of unknown rank in a checkpoint")
    return tuple(dims)


def _parse_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 0:
            e["dtype"] = v
        elif num == 2 and wt == 2:
            e["shape"] = _parse_shape(v)
        elif num == 3 and wt == 0:
            e["shard_id"] = v
        elif num == 4 and wt == 0:
            e["offset"] = v
        elif num == 5 and wt == 0:
            e["size"] = v
        elif num == 6 and wt == 5:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] += 1
    return e


def _parse_header(buf):
    h = dict(num_shards=1, endianness=0, version=None)
    for num, wt, v in _fields(buf):
        if num == 1 and wt == 0:
            h["num_shards"] = v
        elif num == 2 and wt == 0:
            h["endianness"] = v
    return h


# ---- snappy (raw format) ------------------------------------------------------------------------------------------------------
def _snappy_decompress(buf):
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                  # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy block")
        for _ in range(ln):                            # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy block: %d bytes, header says %d" % (len(out), n))
    return bytes(out)


# ---- the sorted string table -------------------------------------------------------------------------------------------------
def _read_block(f, offset, size, verify=True):
    raw = f[offset:offset + size + 5]
    if len(raw) != size + 5:
        raise CheckpointError("block [%d, +%d) runs past the end of the index file" % (offset, size))
    body, ctype = raw[:size], raw[size]
    if verify:
        want = struct.unpack_from("<I", raw, size + 1)[0]
        if masked_crc32c(raw[:size + 1]) != want:
            raise CheckpointError("block checksum mismatch at offset %d" % offset)
    if ctype == 1:
        body = _snappy_decompress(body)
    elif ctype != 0:
        raise CheckpointError("unknown block compression type %d" % ctype)
    return body


def _block_entries(block):
    """(key, value) pairs of one block, in order (prefix compression undone)."""
    if len(block) < 4:
        raise CheckpointError("block too small")
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestarts
    if end < 0:
        raise CheckpointError("bad restart array")
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key):
            raise CheckpointError("bad key prefix length")
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _block_handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def read_index(index_path, verify=True):
    """-> (header dict, {tensor name: entry dict}) of a `.index` file."""
    f = open(index_path, "rb").read()
    if len(f) < 48:
        raise CheckpointError("%s: too short for a table footer" % index_path)
    footer = f[-48:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise CheckpointError("%s: not a TensorFlow V2 checkpoint index (bad table magic)" % index_path)
    _, _, pos = _block_handle(footer, 0)               # metaindex (empty in a bundle)
    ioff, isize, _ = _block_handle(footer, pos)
    header, entries = None, {}
    for _, handle in _block_entries(_read_block(f, ioff, isize, verify)):
        boff, bsize, _ = _block_handle(handle, 0)
        for key, value in _block_entries(_read_block(f, boff, bsize, verify)):
            if key == b"":
                header = _parse_header(value)
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    if header is None:
        raise CheckpointError("%s: no bundle header entry" % index_path)
    if header["endianness"] != 0:
        raise CheckpointError("big-endian checkpoints are not supported")
    return header, entries


def _prefix_of(path):
    """Accepts the checkpoint prefix or the name of its .index / .data-* / .meta file."""
    p = str(path)
    for suffix in (".index", ".meta"):
        if p.endswith(suffix):
            return p[:-len(suffix)]
    m = re.match(r"^(.*)\.data-\d{5}-of-\d{5}$", p)
    return m.group(1) if m else p


def is_tf_checkpoint(path):
    return os.path.isfile(_prefix_of(path) + ".index")


def list_variables(path):
    """[(name, shape, numpy dtype)] like tf.train.list_variables."""
    _, entries = read_index(_prefix_of(path) + ".index")
    return [(k, e["shape"], _DTYPES.get(e["dtype"])) for k, e in sorted(entries.items())]


def read_checkpoint(path, verify_crc=True):
    """-> {variable name: ndarray} of a V2 checkpoint (what tf.train.load_checkpoint(path).get_tensor(name) returns for
    every name).  verify_crc: True / False / "small" (tensor checksums only up to 1 MB); block checksums of the index are always
    verified.  crc32c runs in the HIP library's host helper (cz_crc32c, ~1 GB/s) when the library is built, else in pure
    Python (~5 MB/s)."""
    prefix = _prefix_of(path)
    header, entries = read_index(prefix + ".index")
    shards = {}
    out = {}
    for name, e in sorted(entries.items()):
        if e["slices"]:
            raise CheckpointError("%s: partitioned (sliced) variables are not supported" % name)
        dt = _DTYPES.get(e["dtype"])
        if dt is None:
            raise CheckpointError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        sid = e["shard_id"]
        if sid not in shards:
            sp = "%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"])
            if not os.path.isfile(sp):
                raise CheckpointError("data shard %s is missing" % sp)
            shards[sid] = np.memmap(sp, dtype=np.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if len(raw) != e["size"] or e["size"] != count * np.dtype(dt).itemsize:
            raise CheckpointError("%s: %d bytes in the shard, shape %s of %s needs %d" %
                                  (name, len(raw), e["shape"], np.dtype(dt).name, count * np.dtype(dt).itemsize))
        if e["crc32c"] is not None and (verify_crc is True or (verify_crc == "small" and e["size"] <= (1 << 20))):
            if masked_crc32c(raw.tobytes()) != e["crc32c"]:
                raise CheckpointError("%s: tensor checksum mismatch" % name)
        out[name] = np.frombuffer(raw.tobytes(), dtype=np.dtype(dt).newbyteorder("<")).astype(dt).reshape(e["shape"])
    return out


# ---- writer: the weights back in the reference's own format -----------------------------------------------------------------
_DTYPE_ENUM = {np.dtype(v): k for k, v in _DTYPES.items()}


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | 0x80 if n else b)
        if not n:
            return bytes(out)


def _pb_varint(field, v):
    return _vi(field << 3) + _vi(v if v >= 0 else v + (1 << 64))


def _pb_bytes(field, b):
    return _vi((field << 3) | 2) + _vi(len(b)) + b


class _BlockBuilder:
    """tensorflow/core/lib/io/block_builder.cc: prefix-compressed entries, a restart point every `interval` entries."""

    def __init__(self, interval=16):
        self.buf, self.restarts, self.n, self.last, self.interval = bytearray(), [0], 0, b"", interval

    def add(self, key, value):
        shared = 0
        if self.n % self.interval == 0:
            if self.n:
                self.restarts.append(len(self.buf))
        else:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _vi(shared) + _vi(len(key) - shared) + _vi(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_checkpoint(prefix, tensors, block_size=4096):
    """{variable name: ndarray} -> `prefix.index` + `prefix.data-00000-of-00001`, a V2 checkpoint tf.train.Saver.restore /
    tf.train.load_checkpoint read (one shard, uncompressed blocks, masked crc32c on every block and tensor) — the way back
    for weights trained here into the reference's own graph (policy_value_network.py:176-184).  Names are written as given:
    pass net.to_tf_variables(module) for the reference graph's names."""
    names = sorted(tensors)
    data = bytearray()
    header = _pb_varint(1, 1) + _pb_bytes(3, _pb_varint(1, 1))          # num_shards = 1, version { producer: 1 }
    items = [(b"", header)]
    for n in names:
        a = np.asarray(tensors[n], order="C")
        if a.dtype not in _DTYPE_ENUM:
            raise CheckpointError("%s: dtype %s cannot be written" % (n, a.dtype))
        raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
        e = _pb_varint(1, _DTYPE_ENUM[a.dtype]) + _pb_bytes(2, b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in a.shape))
        if len(data):
            e += _pb_varint(4, len(data))
        e += _pb_varint(5, len(raw)) + _vi((6 << 3) | 5) + struct.pack("<I", masked_crc32c(raw))
        items.append((n.encode("utf-8"), e))
        data += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block)
        out.append(0)                                                      # kNoCompression
        out.extend(struct.pack("<I", masked_crc32c(block + b"\0")))
        return off, len(block)

    index, blk = _BlockBuilder(interval=1), _BlockBuilder()
    for key, value in items:
        blk.add(key, value)
        if len(blk.buf) >= block_size:
            off, size = emit(blk.finish())
            index.add(blk.last, _vi(off) + _vi(size))
            blk = _BlockBuilder()
    if blk.n:
        off, size = emit(blk.finish())
        index.add(blk.last, _vi(off) + _vi(size))
    moff, msize = emit(_BlockBuilder().finish())
    ioff, isize = emit(index.finish())
    footer = _vi(moff) + _vi(msize) + _vi(ioff) + _vi(isize)
    out.extend(footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


def write_checkpoint_state(save_dir, prefix_name):
    """The `checkpoint` state file tf.train.Saver keeps beside its checkpoints (get_checkpoint_state, :165)."""
    with open(os.path.join(save_dir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (prefix_name, prefix_name))


def latest_checkpoint(save_dir):
    """tf.train.latest_checkpoint / get_checkpoint_state(...).model_checkpoint_path (policy_value_network.py:165-168):
    the prefix named by the `checkpoint` state file of save_dir, or None."""
    state = os.path.join(save_dir, "checkpoint")
    if not os.path.isfile(state):
        return None
    for line in open(state, "r", errors="replace"):
        m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"\s*$', line)
        if m:
            p = m.group(1)
            p = p if os.path.isabs(p) else os.path.join(save_dir, p)
            return p if os.path.isfile(p + ".index") else None
    return None
