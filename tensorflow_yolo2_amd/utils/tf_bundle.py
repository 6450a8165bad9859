"""Reader / writer of TensorFlow "V2" checkpoints (the tensor bundle tf.train.Saver writes: `<prefix>.index` +
`<prefix>.data-SSSSS-of-NNNNN`), in plain Python + numpy -- no TensorFlow, no protobuf package.

Why: the reference restores and saves its variables with tf.train.Saver (src/yolo2_nets/net_utils.py:64-110,
src/pascal/pascal_train_darknet.py:88,111-114) and publishes trained weights as `.ckpt` files (README.md:22-24);
this module lets those files be read into (and written from) the flat parameter buffers of this repo.

Format, as published in the TensorFlow sources (tensorflow/core/util/tensor_bundle/, core/lib/io/table*,
core/protobuf/tensor_bundle.proto) and the LevelDB table format it reuses:
  .index  = an SSTable: data blocks of prefix-compressed (key, value) entries with a restart array, each block
            followed by a 5-byte trailer (compression type, masked CRC-32C), a meta-index block, an index block
            (last key of each data block -> BlockHandle) and a 48-byte footer ending in the magic
            0xdb4775248b80fb57.  key "" -> BundleHeaderProto, key <tensor name> -> BundleEntryProto
            (dtype, shape, shard_id, offset, size, masked crc32c of the bytes).
  .data-* = the raw little-endian tensor bytes at [offset, offset + size).
No TF-written file is available in this environment: the reader is checked against this writer, against
hand-assembled blocks with shared-prefix keys and snappy-compressed blocks, and against the published CRC-32C /
snappy test vectors (tests/test_tf_bundle.py) -- "unpinned against TensorFlow itself" in the sense of DESIGN.md.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64 = 1, 2, 3, 9
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_DOUBLE: np.dtype("<f8"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("float64"): DT_DOUBLE, np.dtype("int32"): DT_INT32,
          np.dtype("int64"): DT_INT64}


# ---------------------------------------------------------------------------------------------- CRC-32C
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def _crc32c_py(data, crc=0):
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of a bytes-like object; large buffers go through the library's host routine
    (y2_crc32c: hardware crc32 instruction), small ones through the table above"""
    mv = memoryview(data).cast("B")
    if len(mv) < 4096:
        return _crc32c_py(mv, crc)
    try:
        import ctypes as C
        from .. import _lib
        lib = _lib.load()
        buf = np.frombuffer(mv, dtype=np.uint8)
        return int(lib.y2_crc32c(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.size), C.c_uint32(crc)))
    except Exception:       # library not built: the slow path is still correct
        return _crc32c_py(mv, crc)


def mask_crc(crc):
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    rot = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---------------------------------------------------------------------------------------------- varints / protobuf
def _get_varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("varint too long")


def _put_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb_fields(buf):
    """[(field number, wire type, value)] of one protobuf message; value: int (varint / fixed) or bytes"""
    buf = bytes(buf)
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = buf[pos:pos + n]
            if len(v) != n:
                raise ValueError("truncated protobuf field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        out.append((num, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _pb_tag(num, wt):
    return _put_varint((num << 3) | wt)


def _pb_bytes(num, payload):
    return _pb_tag(num, 2) + _put_varint(len(payload)) + payload


def _pb_varint(num, v):
    return _pb_tag(num, 0) + _put_varint(v)


def parse_entry(buf):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c, slices)"""
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=0, slices=0)
    for num, wt, v in _pb_fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:                                  # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
            dims = []
            for n2, _w2, v2 in _pb_fields(v):
                if n2 == 2:
                    size = 0
                    for n3, _w3, v3 in _pb_fields(v2):
                        if n3 == 1:
                            size = _signed64(v3)
                    dims.append(size)
                elif n2 == 3 and v2:
                    raise ValueError("tensor of unknown rank in checkpoint")
            e["shape"] = tuple(dims)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = _signed64(v)
        elif num == 5:
            e["size"] = _signed64(v)
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["slices"] += 1
    return e


def build_entry(dtype, shape, shard_id, offset, size, crc):
    shp = b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)
    out = _pb_varint(1, dtype) + _pb_bytes(2, shp)
    if shard_id:
        out += _pb_varint(3, shard_id)
    if offset:
        out += _pb_varint(4, offset)
    out += _pb_varint(5, size)
    out += _pb_tag(6, 5) + struct.pack("<I", crc)
    return out


def parse_header(buf):
    h = dict(num_shards=0, endianness=0, version=(0, 0))
    for num, _wt, v in _pb_fields(buf):
        if num == 1:
            h["num_shards"] = v
        elif num == 2:
            h["endianness"] = v
        elif num == 3:
            d = {n: x for n, _w, x in _pb_fields(v)}
            h["version"] = (d.get(1, 0), d.get(2, 0))
    return h


def build_header(num_shards):
    return _pb_varint(1, num_shards) + _pb_bytes(3, _pb_varint(1, 1))     # little endian (0: omitted), producer 1


# ---------------------------------------------------------------------------------------------- snappy (decode only)
def snappy_decompress(buf):
    """raw snappy block format (format_description.txt of google/snappy)"""
    buf = bytes(buf)
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
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
            raise ValueError("corrupt snappy stream")
        for _ in range(ln):                             # byte-wise: copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError("snappy length mismatch")
    return bytes(out)


# ---------------------------------------------------------------------------------------------- SSTable
def _read_block(f, offset, size, verify=True):
    f.seek(offset)
    raw = f.read(size + 5)
    if len(raw) != size + 5:
        raise ValueError("truncated table block")
    contents, ctype = raw[:size], raw[size]
    if verify:
        stored = struct.unpack_from("<I", raw, size + 1)[0]
        if unmask_crc(stored) != crc32c(raw[:size + 1]):
            raise ValueError("table block checksum mismatch at offset %d" % offset)
    if ctype == 0:
        return contents
    if ctype == 1:
        return snappy_decompress(contents)
    raise ValueError("unknown block compression %d" % ctype)


def _block_entries(block):
    """(key, value) pairs of one block (prefix-compressed entries, restart array ignored: a full scan needs none)"""
    if len(block) < 4:
        raise ValueError("bad table block")
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * num_restarts
    if end < 0:
        raise ValueError("bad restart array")
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key):
            raise ValueError("bad shared-prefix length")
        key = key[:shared] + block[pos:pos + unshared]
        pos += unshared
        out.append((key, block[pos:pos + vlen]))
        pos += vlen
    return out


def read_table(path, verify=True):
    """every (key, value) of an SSTable file, in key order"""
    with open(path, "rb") as f:
        f.seek(0, os.SEEK_END)
        fsize = f.tell()
        if fsize < 48:
            raise ValueError("%s: too short for a table" % path)
        f.seek(fsize - 48)
        footer = f.read(48)
        if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
            raise ValueError("%s: not a TensorFlow V2 checkpoint index (bad magic)" % path)
        pos = 0
        _mo, pos = _get_varint(footer, pos)
        _ms, pos = _get_varint(footer, pos)
        io, pos = _get_varint(footer, pos)
        isz, pos = _get_varint(footer, pos)
        out = []
        for _k, handle in _block_entries(_read_block(f, io, isz, verify)):
            bo, p2 = _get_varint(handle, 0)
            bs, _ = _get_varint(handle, p2)
            out += _block_entries(_read_block(f, bo, bs, verify))
    return out


class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b""

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def write_table(path, items, block_size=4096):
    """items: (key bytes, value bytes) sorted by key.  Uncompressed blocks, restart interval 16 (data) / 1 (index)."""
    items = list(items)
    assert all(items[i][0] < items[i + 1][0] for i in range(len(items) - 1)), "keys must be sorted and unique"
    with open(path, "wb") as f:
        def emit(contents):
            off = f.tell()
            trailer = bytes([0])
            f.write(contents + trailer + struct.pack("<I", mask_crc(crc32c(contents + trailer))))
            return off, len(contents)
        index = _BlockBuilder(1)
        blk = _BlockBuilder(16)
        for key, value in items:
            blk.add(key, value)
            if blk.size() >= block_size:
                off, sz = emit(blk.finish())
                index.add(blk.last, _put_varint(off) + _put_varint(sz))
                blk = _BlockBuilder(16)
        if blk.count or not items:
            off, sz = emit(blk.finish())
            index.add(blk.last, _put_varint(off) + _put_varint(sz))
        moff, msz = emit(_BlockBuilder(1).finish())
        ioff, isz = emit(index.finish())
        footer = _put_varint(moff) + _put_varint(msz) + _put_varint(ioff) + _put_varint(isz)
        f.write(footer + b"\0" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


# ---------------------------------------------------------------------------------------------- bundle
def _shard_name(prefix, shard, num):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num)


def is_bundle(prefix):
    return os.path.isfile(prefix + ".index")


class BundleReader:
    """tf.train.load_checkpoint(prefix) counterpart: names(), shape / dtype map, get_tensor(name)"""

    def __init__(self, prefix, verify_index=True):
        self.prefix = prefix
        table = read_table(prefix + ".index", verify_index)
        if not table or table[0][0] != b"":
            raise ValueError("%s.index: no bundle header" % prefix)
        self.header = parse_header(table[0][1])
        if self.header["endianness"] != 0:
            raise ValueError("big-endian checkpoints are not supported")
        if self.header["version"][1] > 1:
            raise ValueError("checkpoint needs a newer reader (min_consumer %d)" % self.header["version"][1])
        self.entries = {k.decode("utf-8"): parse_entry(v) for k, v in table[1:]}

    def names(self):
        return sorted(self.entries)

    def variable_to_shape_map(self):
        return {k: e["shape"] for k, e in self.entries.items()}

    def has_tensor(self, name):
        return name in self.entries

    def get_tensor(self, name, verify=True):
        e = self.entries[name]
        if e["slices"]:
            raise ValueError("%s: partitioned variables are not supported" % name)
        dt = _DTYPES.get(e["dtype"])
        if dt is None:
            raise ValueError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if count * dt.itemsize != e["size"]:
            raise ValueError("%s: %d bytes stored, shape %s needs %d" % (name, e["size"], e["shape"], count * dt.itemsize))
        with open(_shard_name(self.prefix, e["shard_id"], self.header["num_shards"]), "rb") as f:
            f.seek(e["offset"])
            raw = f.read(e["size"])
        if len(raw) != e["size"]:
            raise ValueError("%s: data file truncated" % name)
        if verify and unmask_crc(e["crc32c"]) != crc32c(raw):
            raise ValueError("%s: checksum mismatch" % name)
        return np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()


def write_bundle(prefix, tensors):
    """tensors: name -> numpy array (float32 / float64 / int32 / int64).  One shard."""
    items = [(b"", build_header(1))]
    with open(_shard_name(prefix, 0, 1), "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode("utf-8")):
            a = np.asarray(tensors[name])
            if a.dtype not in _DT_OF:
                raise ValueError("%s: dtype %s not supported" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="C")
            off = f.tell()
            f.write(raw)
            items.append((name.encode("utf-8"), build_entry(_DT_OF[a.dtype], a.shape, 0, off, len(raw),
                                                             mask_crc(crc32c(raw)))))
    write_table(prefix + ".index", items)
    return sorted(tensors)


# ---------------------------------------------------------------------------------------------- V1 checkpoints
# The "V1" format (tensorflow/core/util/tensor_slice_writer.cc, saved_tensor_slice.proto; what tf.train.Saver wrote
# before TF 0.12 and what the slim model zoo's resnet_v1_50.ckpt is -- the file restore_resnet_tf_variables loads,
# src/yolo2_nets/net_utils.py:137-196): ONE file, the same SSTable container as the V2 index, whose values are
# SavedTensorSlices messages:
#   key ""                         -> { meta  { tensor { name, shape, type, slice* }*, versions } }
#   key OrderedCode(0, name, slice) -> { data { name, slice { extent { start, length }* }, data = TensorProto } }
# with the values of a slice in the TensorProto's typed repeated field (float_val = packed little-endian floats) or
# its tensor_content bytes.  A Saver writes each variable as one full slice; partial slices are placed by their extents.
_TP_FIELD = {DT_FLOAT: 5, DT_DOUBLE: 6, DT_INT32: 7, DT_INT64: 10}


def _parse_shape(buf):
    dims = []
    for num, _wt, v in _pb_fields(buf):
        if num == 2:
            size = 0
            for n2, _w2, v2 in _pb_fields(v):
                if n2 == 1:
                    size = _signed64(v2)
            dims.append(size)
    return tuple(dims)


def _parse_slice(buf):
    """TensorSliceProto -> [(start, length or None)]"""
    ext = []
    for num, _wt, v in _pb_fields(buf):
        if num == 1:
            start, length = 0, None
            for n2, _w2, v2 in _pb_fields(v):
                if n2 == 1:
                    start = _signed64(v2)
                elif n2 == 2:
                    length = _signed64(v2)
            ext.append((start, length))
    return ext


def _parse_tensor_proto(buf):
    """TensorProto -> (dtype, flat numpy array)"""
    dtype, content, chunks = 0, None, []
    fields = _pb_fields(buf)
    for num, _wt, v in fields:
        if num == 1:
            dtype = v
    if dtype not in _DTYPES:
        raise ValueError("unsupported tensor dtype %d in a V1 checkpoint" % dtype)
    np_dt, want = _DTYPES[dtype], _TP_FIELD[dtype]
    for num, wt, v in fields:
        if num == 4:
            content = np.frombuffer(v, dtype=np_dt)
        elif num == want:
            if wt == 2 and dtype in (DT_FLOAT, DT_DOUBLE):          # packed fixed-width values
                chunks.append(np.frombuffer(v, dtype=np_dt))
            elif wt == 2:                                            # packed varints
                vals, pos = [], 0
                while pos < len(v):
                    x, pos = _get_varint(v, pos)
                    vals.append(_signed64(x))
                chunks.append(np.asarray(vals, dtype=np_dt))
            elif wt == 5:
                chunks.append(np.frombuffer(struct.pack("<I", v), dtype=np_dt))
            elif wt == 1:
                chunks.append(np.frombuffer(struct.pack("<Q", v), dtype=np_dt))
            else:
                chunks.append(np.asarray([_signed64(v)], dtype=np_dt))
    if content is not None and content.size:
        return dtype, content
    return dtype, (np.concatenate(chunks) if chunks else np.zeros(0, np_dt))


def is_v1_checkpoint(path):
    """a single-file table whose first key is "" holding a SavedTensorSlices meta record"""
    if not os.path.isfile(path) or os.path.isfile(path + ".index"):
        return False
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            if f.tell() < 48:
                return False
            f.seek(-8, os.SEEK_END)
            return struct.unpack("<Q", f.read(8))[0] == TABLE_MAGIC
    except OSError:
        return False


def read_checkpoint_v1(path, verify=True):
    """{tensor name: numpy array} of a V1 checkpoint file"""
    entries = read_table(path, verify)
    if not entries or entries[0][0] != b"":
        raise ValueError("%s: no SavedTensorSlices header record" % path)
    meta = {}
    for num, _wt, v in _pb_fields(entries[0][1]):
        if num != 1:
            continue
        for n2, _w2, v2 in _pb_fields(v):
            if n2 != 1:
                continue
            name, shape, dtype = "", (), 0
            for n3, _w3, v3 in _pb_fields(v2):
                if n3 == 1:
                    name = v3.decode()
                elif n3 == 2:
                    shape = _parse_shape(v3)
                elif n3 == 3:
                    dtype = v3
            meta[name] = (shape, dtype)
    out = {}
    for key, value in entries[1:]:
        for num, _wt, v in _pb_fields(value):
            if num != 2:
                continue
            name, ext, tp = "", [], None
            for n2, _w2, v2 in _pb_fields(v):
                if n2 == 1:
                    name = v2.decode()
                elif n2 == 2:
                    ext = _parse_slice(v2)
                elif n2 == 3:
                    tp = v2
            if name not in meta or tp is None:
                raise ValueError("%s: slice of %r without a header entry" % (path, name))
            shape, dtype = meta[name]
            dt, flat = _parse_tensor_proto(tp)
            if dt != dtype:
                raise ValueError("%s: %s is stored as dtype %d, the header says %d" % (path, name, dt, dtype))
            full = all(length is None and start == 0 for start, length in ext) or not ext
            if full:
                if flat.size != int(np.prod(shape, dtype=np.int64)):
                    raise ValueError("%s: %s holds %d values, shape %s" % (path, name, flat.size, shape))
                out[name] = flat.reshape(shape).copy()
            else:
                arr = out.setdefault(name, np.zeros(shape, _DTYPES[dtype]))
                idx = tuple(slice(s, None if ln is None else s + ln) for s, ln in ext)
                arr[idx] = flat.reshape(arr[idx].shape)
    missing = sorted(set(meta) - set(out))
    if missing:
        raise ValueError("%s: no data record for %s" % (path, ", ".join(missing[:5])))
    return out


def _ordered_num(v):
    """OrderedCode::WriteNumIncreasing: length byte + big-endian bytes"""
    body = b""
    while v:
        body = bytes([v & 0xFF]) + body
        v >>= 8
    return bytes([len(body)]) + body


def _ordered_string(b):
    """OrderedCode::WriteString: 0x00 -> 00 ff, 0xff -> ff 00, terminator 00 01"""
    return b"".join(b"\x00\xff" if c == 0 else (b"\xff\x00" if c == 255 else bytes([c])) for c in b) + b"\x00\x01"


def _v1_key(name, rank):
    """EncodeTensorNameSlice of a full slice: 0, name, rank, then (start 0, length -1) per dimension -- small signed
    numbers take one byte, 0x80 ^ value"""
    return _ordered_num(0) + _ordered_string(name.encode()) + _ordered_num(rank) + bytes([0x80, 0x7f]) * rank


def write_checkpoint_v1(path, tensors):
    """the file tf.train.Saver(write_version=V1) writes for {name: array}: each tensor one full slice with its values
    in the typed repeated field (used by the tests and to hand a backbone to the reference)"""
    def shape_proto(shape):
        return b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)

    def slice_proto(rank):
        return b"".join(_pb_bytes(1, b"") for _ in range(rank))
    meta, items = b"", []
    for name in sorted(tensors):
        a = np.asarray(tensors[name])            # (ascontiguousarray would turn a scalar into shape (1,))
        if a.dtype not in _DT_OF:
            raise ValueError("%s: dtype %s cannot be written" % (name, a.dtype))
        dt = _DT_OF[a.dtype]
        meta += _pb_bytes(1, _pb_bytes(1, name.encode()) + _pb_bytes(2, shape_proto(a.shape)) + _pb_varint(3, dt) +
                          _pb_bytes(4, slice_proto(a.ndim)))
        if dt in (DT_FLOAT, DT_DOUBLE):
            payload = a.astype(_DTYPES[dt]).tobytes()
        else:
            payload = b"".join(_put_varint(int(x)) for x in a.reshape(-1))
        tp = _pb_varint(1, dt) + _pb_bytes(_TP_FIELD[dt], payload)
        data = _pb_bytes(1, name.encode()) + _pb_bytes(2, slice_proto(a.ndim)) + _pb_bytes(3, tp)
        items.append((_v1_key(name, a.ndim), _pb_bytes(2, data)))
    header = _pb_bytes(1, meta + _pb_bytes(2, _pb_varint(1, 1)))          # versions { producer: 1 }
    write_table(path, [(b"", header)] + sorted(items))
