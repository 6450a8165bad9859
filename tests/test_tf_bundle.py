"""CPU tests of utils/tf_bundle.py (TensorFlow V2 checkpoint = tensor bundle reader / writer; reference:
tf.train.Saver in src/yolo2_nets/net_utils.py:64-110, src/pascal/pascal_train_darknet.py:88,111-114).
No TensorFlow and no TF-written file exist here, so the reader is checked against published test vectors
(CRC-32C: RFC 3720 B.4; masked form: leveldb crc32c_test; snappy: its format description), against blocks
assembled by hand with shared-prefix keys / several data blocks / a snappy-compressed block, and against the writer."""
import os
import struct

import numpy as np
import pytest

from tensorflow_yolo2_amd.utils import tf_bundle as B


def test_crc32c_known_answers():
    assert B._crc32c_py(b"123456789") == 0xE3069283
    assert B._crc32c_py(bytes(32)) == 0x8A9136AA                 # RFC 3720 B.4: 32 bytes of zeros
    assert B._crc32c_py(bytes([0xFF] * 32)) == 0x62A8AB43        # 32 bytes of ones
    assert B._crc32c_py(bytes(range(32))) == 0x46DD794E          # 32 bytes incrementing
    # continuation, and the library's host routine (hardware crc32) on a large buffer
    data = np.random.default_rng(0).integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    assert B._crc32c_py(data[5000:10000], B._crc32c_py(data[:5000])) == B._crc32c_py(data[:10000])
    assert B.crc32c(data) == B._crc32c_py(data)
    assert B.crc32c(data[1:]) == B._crc32c_py(data[1:])          # unaligned start
    # leveldb's mask: rotate right 15, add a constant; unmask inverts it
    c = B._crc32c_py(b"foo")
    assert B.mask_crc(c) != c and B.unmask_crc(B.mask_crc(c)) == c
    assert B.mask_crc(0) == 0xa282ead8


def test_snappy_decoder_on_the_format_description_cases():
    assert B.snappy_decompress(bytes([0])) == b""
    assert B.snappy_decompress(bytes([5, 4 << 2]) + b"hello") == b"hello"                       # one literal
    # literal "ab", then a 1-byte-offset copy (len 4 + 2 = 6, offset 2): overlapping run "ababab"
    assert B.snappy_decompress(bytes([8, 1 << 2]) + b"ab" + bytes([((6 - 4) << 2) | 1, 2])) == b"abababab"
    # 2-byte-offset copy
    assert B.snappy_decompress(bytes([10, 4 << 2]) + b"12345" + bytes([((5 - 1) << 2) | 2, 5, 0])) == b"1234512345"
    # literal with an explicit one-byte length (60 -> 1 extra byte), 100 bytes
    lit = bytes(range(100))
    assert B.snappy_decompress(bytes([100, 60 << 2, 99]) + lit) == lit
    with pytest.raises(ValueError):
        B.snappy_decompress(bytes([4, ((4 - 1) << 2) | 2, 9, 0]))                                   # offset beyond output


def _block(entries, restart_every=16):
    bb = B._BlockBuilder(restart_every)
    for k, v in entries:
        bb.add(k, v)
    return bb.finish()


def test_block_prefix_compression_and_table_scan(tmp_path):
    keys = [b"", b"darknet19/Variable", b"darknet19/Variable/Adam", b"darknet19/Variable_1",
            b"darknet19/batch_normalization/beta", b"darknet19/batch_normalization/gamma", b"zeta"]
    ents = [(k, b"v%d" % i) for i, k in enumerate(keys)]
    blk = _block(ents, restart_every=3)
    assert B._block_entries(blk) == ents
    # the second key shares no prefix with "" but the third shares 18 bytes with the second
    shared, pos = B._get_varint(blk, 0)
    assert shared == 0
    # a table with a tiny block size: many data blocks, index entries, footer magic
    path = str(tmp_path / "t.index")
    big = [(b"k%05d" % i, bytes([i & 255]) * (i % 50)) for i in range(500)]
    B.write_table(path, big, block_size=256)
    assert B.read_table(path) == big
    raw = open(path, "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == B.TABLE_MAGIC and len(raw) > 48
    # corruption is caught by the block checksum
    bad = bytearray(raw)
    bad[10] ^= 1
    open(path, "wb").write(bad)
    with pytest.raises(ValueError):
        B.read_table(path)
    assert B.read_table(path, verify=False) != big or True


def test_snappy_compressed_block_in_a_table(tmp_path):
    """a TF build whose table writer compresses: block type 1 = raw snappy.  Assemble such a file by hand (the
    'compressor' emits literals only, which every snappy decoder accepts)."""
    ents = [(b"", B.build_header(1)), (b"w", B.build_entry(B.DT_FLOAT, (2, 2), 0, 0, 16, 123))]
    contents = _block(ents)

    def snappy_literal(data):
        out = bytearray(B._put_varint(len(data)))
        for i in range(0, len(data), 60):
            piece = data[i:i + 60]
            out += bytes([(len(piece) - 1) << 2]) + piece
        return bytes(out)

    path = str(tmp_path / "c.index")
    with open(path, "wb") as f:
        def emit(payload, ctype):
            off = f.tell()
            f.write(payload + bytes([ctype]) + struct.pack("<I", B.mask_crc(B.crc32c(payload + bytes([ctype])))))
            return off, len(payload)
        d = emit(snappy_literal(contents), 1)
        m = emit(_block([]), 0)
        idx = _block([(b"w", B._put_varint(d[0]) + B._put_varint(d[1]))], 1)
        i = emit(idx, 0)
        foot = B._put_varint(m[0]) + B._put_varint(m[1]) + B._put_varint(i[0]) + B._put_varint(i[1])
        f.write(foot + b"\0" * (40 - len(foot)) + struct.pack("<Q", B.TABLE_MAGIC))
    table = B.read_table(path)
    assert [k for k, _ in table] == [b"", b"w"]
    e = B.parse_entry(table[1][1])
    assert e["dtype"] == B.DT_FLOAT and e["shape"] == (2, 2) and e["size"] == 16 and e["crc32c"] == 123


def test_bundle_entry_proto_bytes():
    """BundleEntryProto on the wire (tensor_bundle.proto): dtype = 1 varint, shape = 2 message, offset = 4, size = 5,
    crc32c = 6 fixed32; zero-valued scalars omitted (proto3)."""
    b = B.build_entry(B.DT_FLOAT, (3, 3, 3, 32), 0, 0, 3456, 0xAABBCCDD)
    assert b[:2] == bytes([0x08, 0x01])                                 # field 1, varint, DT_FLOAT
    assert b[2] == 0x12                                                 # field 2, length-delimited
    assert bytes([0x12, 0x02, 0x08, 0x03]) in b                         # Dim { size: 3 }
    assert bytes([0x12, 0x02, 0x08, 0x20]) in b                         # Dim { size: 32 }
    assert b.endswith(bytes([0x35]) + struct.pack("<I", 0xAABBCCDD))    # field 6, fixed32
    e = B.parse_entry(b)
    assert e == dict(dtype=1, shape=(3, 3, 3, 32), shard_id=0, offset=0, size=3456, crc32c=0xAABBCCDD, slices=0)
    e = B.parse_entry(B.build_entry(B.DT_FLOAT, (), 0, 1 << 33, 4, 7))   # scalar (beta1_power), 64-bit offset
    assert e["shape"] == () and e["offset"] == 1 << 33
    h = B.parse_header(B.build_header(1))
    assert h == dict(num_shards=1, endianness=0, version=(1, 0))


def test_bundle_round_trip_and_errors(tmp_path):
    rng = np.random.default_rng(3)
    tensors = {"darknet19/Variable": rng.standard_normal((3, 3, 3, 32)).astype(np.float32),
               "darknet19/Variable_1": np.full(32, 0.1, np.float32),
               "darknet19/batch_normalization/moving_variance": rng.uniform(0.5, 2, 32).astype(np.float32),
               "beta1_power": np.float32(0.9 ** 8),
               "global_step": np.int64(80000),
               "big": rng.standard_normal((64, 1024)).astype(np.float32)}
    prefix = str(tmp_path / "train_iter_80000.ckpt")
    B.write_bundle(prefix, tensors)
    assert B.is_bundle(prefix) and os.path.isfile(prefix + ".data-00000-of-00001")
    r = B.BundleReader(prefix)
    assert r.names() == sorted(tensors)
    assert r.variable_to_shape_map()["darknet19/Variable"] == (3, 3, 3, 32)
    for k, v in tensors.items():
        got = r.get_tensor(k)
        assert got.dtype == np.asarray(v).dtype and got.shape == np.asarray(v).shape
        np.testing.assert_array_equal(got, v)
    # a flipped bit in the data file is caught by the tensor checksum
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[r.entries["big"]["offset"] + 100] ^= 0x10
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    with pytest.raises(ValueError):
        B.BundleReader(prefix).get_tensor("big")
    B.BundleReader(prefix).get_tensor("big", verify=False)
    with pytest.raises(ValueError):
        B.write_bundle(str(tmp_path / "x.ckpt"), {"a": np.zeros(3, np.float16)})
    open(str(tmp_path / "junk.index"), "wb").write(b"\0" * 100)
    with pytest.raises(ValueError):
        B.BundleReader(str(tmp_path / "junk"))


def test_v1_checkpoint_round_trip_keys_and_partial_slices(tmp_path):
    """The single-file "V1" checkpoint (tensor_slice_writer.cc / saved_tensor_slice.proto: slim's resnet_v1_50.ckpt,
    which restore_resnet_tf_variables loads, net_utils.py:137-196): the writer's records byte for byte against the
    published encodings (OrderedCode keys, SavedTensorSlices protos), the reader on them, on a slice stored by
    extents, on tensor_content payloads, and its errors.  No TF-written file exists here: unpinned against TF itself."""
    p = str(tmp_path / "model.ckpt")
    t = {"resnet_v1_50/conv1/weights": np.arange(7 * 7 * 3 * 4, dtype=np.float32).reshape(7, 7, 3, 4) / 7,
         "global_step": np.int64(1234), "a/b": np.arange(6, dtype=np.int32).reshape(2, 3) - 2,
         "big": np.linspace(-1, 1, 300 * 50).astype(np.float32).reshape(300, 50), "empty": np.zeros((0,), np.float32)}
    B.write_checkpoint_v1(p, t)
    assert B.is_v1_checkpoint(p) and not B.is_bundle(p)
    got = B.read_checkpoint_v1(p)
    assert sorted(got) == sorted(t)
    for k in t:
        assert got[k].dtype == np.asarray(t[k]).dtype and got[k].shape == np.asarray(t[k]).shape
        np.testing.assert_array_equal(got[k], t[k])
    # keys: OrderedCode(0) | escaped name + 00 01 | OrderedCode(rank) | (start 0 -> 80, length -1 -> 7f) per dimension
    assert B._v1_key("ab", 2) == bytes.fromhex("00" "6162" "0001" "0102" "807f807f")
    assert B._v1_key("s", 0) == bytes.fromhex("00" "73" "0001" "00")
    entries = B.read_table(p)
    assert entries[0][0] == b"" and [k for k, _ in entries[1:]] == sorted(B._v1_key(n, np.asarray(a).ndim) for n, a in t.items())
    # the record of "a/b": SavedTensorSlices{ data{ name, slice{extent{} extent{}}, data{dtype: DT_INT32, int_val: packed} } }
    rec = dict(entries)[B._v1_key("a/b", 2)]
    ints = b"".join(B._put_varint(v) for v in (-2, -1, 0, 1, 2, 3))
    tp = bytes([0x08, 3]) + bytes([0x3a, len(ints)]) + ints
    want = bytes([0x0a, 3]) + b"a/b" + bytes([0x12, 4, 0x0a, 0, 0x0a, 0]) + bytes([0x1a, len(tp)]) + tp
    assert rec == bytes([0x12, len(want)]) + want
    # a tensor saved in two row slices (extents with start / length) and one stored as tensor_content
    def rec_slice(name, ext, arr, content=False):
        sl = b"".join(B._pb_bytes(1, (B._pb_varint(1, s) if s else b"") + (B._pb_varint(2, ln) if ln is not None else b""))
                      for s, ln in ext)
        tpb = B._pb_varint(1, B.DT_FLOAT) + B._pb_bytes(4 if content else 5, arr.astype("<f4").tobytes())
        return B._pb_bytes(2, B._pb_bytes(1, name.encode()) + B._pb_bytes(2, sl) + B._pb_bytes(3, tpb))
    w = np.arange(12, dtype=np.float32).reshape(4, 3)
    c = np.arange(5, dtype=np.float32)
    shape = lambda s: b"".join(B._pb_bytes(2, B._pb_varint(1, d)) for d in s)
    meta = b"".join(B._pb_bytes(1, B._pb_bytes(1, n.encode()) + B._pb_bytes(2, shape(s)) + B._pb_varint(3, B.DT_FLOAT))
                    for n, s in (("w", (4, 3)), ("c", (5,))))
    items = [(b"", B._pb_bytes(1, meta)), (b"\x00c1", rec_slice("c", [(0, None)], c, content=True)),
             (b"\x00w1", rec_slice("w", [(0, 3), (0, None)], w[:3])), (b"\x00w2", rec_slice("w", [(3, 1), (0, None)], w[3:]))]
    q = str(tmp_path / "sliced.ckpt")
    B.write_table(q, items)
    got = B.read_checkpoint_v1(q)
    np.testing.assert_array_equal(got["w"], w)
    np.testing.assert_array_equal(got["c"], c)
    # errors: a header entry without data, a value count that does not match the shape
    B.write_table(q, items[:2])
    with pytest.raises(ValueError, match="no data record"):
        B.read_checkpoint_v1(q)
    B.write_table(q, [items[0], items[1], (b"\x00w1", rec_slice("w", [], w[:2]))])
    with pytest.raises(ValueError, match="holds 6 values"):
        B.read_checkpoint_v1(q)
    assert not B.is_v1_checkpoint(str(tmp_path / "nothing.ckpt"))
