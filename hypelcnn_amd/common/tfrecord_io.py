"""TFRecord files and tf.train.Example messages without TensorFlow (reference importer/TFRecordImporter.py,
utilities/tfrecord_writer.py).

Record framing (tensorflow/core/lib/io/record_writer.cc): uint64 length | masked CRC-32C of the 8 length bytes |
payload | masked CRC-32C of the payload, little endian; optionally the whole file gzip-compressed
(TFRecordCompressionType.GZIP).  Payloads here are tf.train.Example protos:
    Example{features=1: Features{feature=1: map<string, Feature{bytes_list=1 | float_list=2 | int64_list=3}>}}
with packed repeated values.  Unpinned against TensorFlow itself (none here); known-answer bytes and round trips
in tests/test_tf_checkpoint.py."""
import gzip
import struct

import numpy

from hypelcnn_amd.common.tf_checkpoint import _pb_bytes, _pb_fields, crc32c, get_varint, mask_crc, put_varint, \
    unmask_crc


# ------------------------------------------------------------------------------------------------ Example protos
def encode_example(features):
    """{name: 1-D array-like of float32 / int64 / bytes objects} -> serialized tf.train.Example (keys sorted, as the
    C++ map serializer of deterministic builds does)."""
    entries = b""
    for key in sorted(features):
        value = features[key]
        if isinstance(value, (bytes, bytearray)) or (isinstance(value, (list, tuple)) and value
                                                      and isinstance(value[0], (bytes, bytearray))):
            items = [value] if isinstance(value, (bytes, bytearray)) else list(value)
            feature = _pb_bytes(1, b"".join(_pb_bytes(1, bytes(v)) for v in items))
        else:
            arr = numpy.asarray(value)
            if arr.dtype.kind == "f":
                payload = arr.astype("<f4").reshape(-1).tobytes()
                feature = _pb_bytes(2, _pb_bytes(1, payload))
            else:
                payload = b"".join(put_varint(int(v)) for v in arr.reshape(-1))
                feature = _pb_bytes(3, _pb_bytes(1, payload))
        entries += _pb_bytes(1, _pb_bytes(1, key.encode()) + _pb_bytes(2, feature))
    return _pb_bytes(1, entries)


def _decode_feature(buf):
    for field, wt, v in _pb_fields(buf):
        if field == 1:  # BytesList
            return [x for f, _, x in _pb_fields(v) if f == 1]
        if field == 2:  # FloatList: packed (wire type 2) or repeated fixed32
            out = []
            for f, w2, x in _pb_fields(v):
                if f == 1:
                    out.append(numpy.frombuffer(x, "<f4") if w2 == 2 else
                               numpy.frombuffer(struct.pack("<I", x), "<f4"))
            return numpy.concatenate(out) if out else numpy.zeros(0, numpy.float32)
        if field == 3:  # Int64List
            vals = []
            for f, w2, x in _pb_fields(v):
                if f != 1:
                    continue
                if w2 == 2:
                    pos = 0
                    while pos < len(x):
                        val, pos = get_varint(x, pos)
                        vals.append(val - (1 << 64) if val >= (1 << 63) else val)
                else:
                    vals.append(x - (1 << 64) if x >= (1 << 63) else x)
            return numpy.asarray(vals, numpy.int64)
    return None


def decode_example(buf):
    out = {}
    for field, _, features in _pb_fields(buf):
        if field != 1:
            continue
        for f2, _, entry in _pb_fields(features):
            if f2 != 1:
                continue
            key, val = None, None
            for f3, _, x in _pb_fields(entry):
                if f3 == 1:
                    key = x.decode()
                elif f3 == 2:
                    val = _decode_feature(x)
            out[key] = val
    return out


# ------------------------------------------------------------------------------------------------ record files
def _open(path, mode, compressed):
    return gzip.open(path, mode) if compressed else open(path, mode)


def write_records(path, payloads, compressed=False):
    n = 0
    with _open(path, "wb", compressed) as f:
        for p in payloads:
            head = struct.pack("<Q", len(p))
            f.write(head + struct.pack("<I", mask_crc(crc32c(head))) + p + struct.pack("<I", mask_crc(crc32c(p))))
            n += 1
    return n


def read_records(path, verify=True):
    with open(path, "rb") as f:
        magic = f.read(2)
    with _open(path, "rb", magic == b"\x1f\x8b") as f:
        while True:
            head = f.read(8)
            if not head:
                return
            if len(head) != 8:
                raise ValueError(f"{path}: truncated record header")
            (length,) = struct.unpack("<Q", head)
            (hcrc,) = struct.unpack("<I", f.read(4))
            if verify and unmask_crc(hcrc) != crc32c(head):
                raise ValueError(f"{path}: corrupted record length")
            data = f.read(length)
            (dcrc,) = struct.unpack("<I", f.read(4))
            if len(data) != length or (verify and unmask_crc(dcrc) != crc32c(data)):
                raise ValueError(f"{path}: corrupted record")
            yield data
