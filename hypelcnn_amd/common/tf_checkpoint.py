"""TensorFlow checkpoint bundles without TensorFlow.

The reference saves and restores `model.ckpt-N` bundles (classify/monitored_session_runner.py:164-171 via
MonitoredTrainingSession; gan/wrappers/cycle_gan_wrapper.py:140-147 restores the published shadow generators that
loader/GRSS2013DataLoader.py:27-33 points at).  A bundle is two files:

  <prefix>.index                 an SSTable (the LevelDB table format, tensorflow/core/lib/io/table*.cc): key "" ->
                                 BundleHeaderProto, key <variable name> -> BundleEntryProto (dtype, shape, shard,
                                 offset, size, masked CRC-32C)
  <prefix>.data-00000-of-00001   the tensors' raw little-endian bytes at those offsets

`read_checkpoint` / `write_checkpoint` implement that format for single-shard bundles with uncompressed index blocks
(what BundleWriter emits) and the dtypes this project stores (float32/float64/int32/int64/uint8/bool).

Parity note: there is no TensorFlow in the build environment, so no TF-written file was available to read and no TF
to load a written file back.  The implementation follows the published format (table format doc of LevelDB,
tensor_bundle.proto) and is tested by round trip, by hand-assembled known-answer bytes for every primitive (varints,
protobuf fields, block trailer, masked CRC-32C with the RFC 3720 vectors) and by corruption detection.
"""
import ctypes
import os
import struct

import numpy

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DTYPES = {1: numpy.float32, 2: numpy.float64, 3: numpy.int32, 4: numpy.uint8, 9: numpy.int64, 10: numpy.bool_}
DTYPE_CODES = {numpy.dtype(v): k for k, v in DTYPES.items()}

_lib = None


def crc32c(data, crc=0):
    """CRC-32C through the host helper of libhypel_hip.so (Python has no fast Castagnoli CRC)."""
    global _lib
    if _lib is None:
        from hypelcnn_amd import backend
        _lib = backend.load_library()
        _lib.hypel_crc32c.restype = ctypes.c_uint32
        _lib.hypel_crc32c.argtypes = [ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint64]
    data = bytes(data)
    return int(_lib.hypel_crc32c(crc, data, len(data)))


def mask_crc(crc):
    """tensorflow/core/lib/hash/crc32c.h: rotate right by 15 and add a constant (CRCs of CRCs stay well mixed)."""
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------------------------ varints / protobuf
def put_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7f) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _pb_fields(buf):
    """Yields (field number, wire type, value) of one protobuf message; value is int or bytes."""
    pos = 0
    while pos < len(buf):
        key, pos = get_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield field, wt, v


def _pb_varint(field, v):
    return put_varint(field << 3) + put_varint(v)


def _pb_bytes(field, payload):
    return put_varint((field << 3) | 2) + put_varint(len(payload)) + payload


def _encode_shape(shape):
    return b"".join(_pb_bytes(2, _pb_varint(1, int(d))) for d in shape)  # TensorShapeProto.dim[].size


def _decode_shape(buf):
    dims = []
    for field, _, v in _pb_fields(buf):
        if field == 2:
            size = 0
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    size = v2 - (1 << 64) if v2 >= (1 << 63) else v2
            dims.append(size)
    return tuple(dims)


def encode_entry(dtype_code, shape, shard_id, offset, size, crc_masked):
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto); zero-valued scalar fields are omitted as
    proto3 serialisers do."""
    out = _pb_varint(1, dtype_code) + _pb_bytes(2, _encode_shape(shape))
    if shard_id:
        out += _pb_varint(3, shard_id)
    if offset:
        out += _pb_varint(4, offset)
    if size:
        out += _pb_varint(5, size)
    out += put_varint((6 << 3) | 5) + struct.pack("<I", crc_masked)
    return out


def decode_entry(buf):
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": 0, "slices": 0}
    for field, _, v in _pb_fields(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            e["shape"] = _decode_shape(v)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = v
        elif field == 7:
            e["slices"] += 1
    return e


def encode_header(num_shards=1):
    """BundleHeaderProto: num_shards = 1, endianness LITTLE (0, omitted), version { producer: 1 }."""
    return _pb_varint(1, num_shards) + _pb_bytes(3, _pb_varint(1, 1))


# ------------------------------------------------------------------------------------------------ SSTable
def _read_block(raw, offset, size, verify=True):
    contents = raw[offset:offset + size]
    ctype = raw[offset + size]
    stored = struct.unpack_from("<I", raw, offset + size + 1)[0]
    if verify and unmask_crc(stored) != crc32c(raw[offset:offset + size + 1]):
        raise ValueError("checkpoint index: block checksum mismatch")
    if ctype != 0:
        raise ValueError("checkpoint index: compressed (snappy) blocks are not supported")
    return contents


def _block_entries(block):
    """(key, value) pairs of one table block (prefix-compressed keys, restart array at the end)."""
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = get_varint(block, pos)
        non_shared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _build_block(pairs, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += put_varint(shared) + put_varint(len(k) - shared) + put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _emit_block(f, contents):
    """Writes block + trailer (type 0 = uncompressed, masked CRC of contents + type); returns the BlockHandle bytes."""
    offset = f.tell()
    f.write(contents)
    f.write(b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00"))))
    return put_varint(offset) + put_varint(len(contents))


def read_index(path, verify=True):
    """{key bytes: value bytes} of an SSTable file."""
    raw = open(path, "rb").read()
    if len(raw) < FOOTER_LEN or struct.unpack_from("<Q", raw, len(raw) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow checkpoint index (bad table magic)")
    footer = raw[-FOOTER_LEN:]
    _, pos = get_varint(footer, 0)          # metaindex handle (unused)
    _, pos = get_varint(footer, pos)
    idx_off, pos = get_varint(footer, pos)
    idx_size, pos = get_varint(footer, pos)
    out = {}
    for _, handle in _block_entries(_read_block(raw, idx_off, idx_size, verify)):
        off, p2 = get_varint(handle, 0)
        size, _ = get_varint(handle, p2)
        for k, v in _block_entries(_read_block(raw, off, size, verify)):
            out[k] = v
    return out


def write_index(path, items, block_size=4096):
    """items: iterable of (key bytes, value bytes) in ascending key order."""
    with open(path, "wb") as f:
        index_pairs, cur, cur_bytes = [], [], 0
        for k, v in items:
            cur.append((k, v))
            cur_bytes += len(k) + len(v) + 6
            if cur_bytes >= block_size:
                index_pairs.append((cur[-1][0], _emit_block(f, _build_block(cur))))
                cur, cur_bytes = [], 0
        if cur:
            index_pairs.append((cur[-1][0], _emit_block(f, _build_block(cur))))
        meta_handle = _emit_block(f, _build_block([]))
        index_handle = _emit_block(f, _build_block(index_pairs, restart_interval=1))
        footer = meta_handle + index_handle
        f.write(footer + b"\x00" * (FOOTER_LEN - 8 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))


# ------------------------------------------------------------------------------------------------ bundles
def _data_path(prefix, shard, n):
    return f"{prefix}.data-{shard:05d}-of-{n:05d}"


def is_checkpoint(prefix):
    return os.path.exists(prefix + ".index")


def read_checkpoint(prefix, names=None, verify_data=True):
    """{variable name: ndarray} of a bundle; `names` restricts the tensors that are materialised."""
    table = read_index(prefix + ".index")
    if b"" not in table:
        raise ValueError(f"{prefix}.index: bundle header missing")
    num_shards, endian = 1, 0
    for field, _, v in _pb_fields(table[b""]):
        if field == 1:
            num_shards = v
        elif field == 2:
            endian = v
    if endian != 0:
        raise ValueError("big-endian bundles are not supported")
    shards = {}
    out = {}
    for key, value in table.items():
        if key == b"":
            continue
        name = key.decode()
        if names is not None and name not in names:
            continue
        e = decode_entry(value)
        if e["slices"]:
            raise ValueError(f"{name}: partitioned (sliced) variables are not supported")
        if e["dtype"] not in DTYPES:
            raise ValueError(f"{name}: unsupported dtype enum {e['dtype']}")
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = numpy.memmap(_data_path(prefix, sid, num_shards), dtype=numpy.uint8, mode="r")
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        if verify_data and unmask_crc(e["crc32c"]) != crc32c(raw.tobytes()):
            raise ValueError(f"{name}: tensor checksum mismatch")
        dt = numpy.dtype(DTYPES[e["dtype"]])
        arr = numpy.frombuffer(raw.tobytes(), dtype=dt)
        out[name] = arr.reshape(e["shape"]).copy()
    return out


def write_checkpoint(prefix, variables):
    """Single-shard bundle of {name: array}; tensors are laid out in key order as BundleWriter does."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = [(b"", encode_header(1))]
    offset = 0
    with open(_data_path(prefix, 0, 1), "wb") as data:
        for name in sorted(variables, key=lambda s: s.encode()):
            arr = numpy.asarray(variables[name])
            if arr.dtype not in DTYPE_CODES:
                raise ValueError(f"{name}: dtype {arr.dtype} has no checkpoint encoding here")
            raw = numpy.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<"), copy=False).tobytes()
            data.write(raw)
            items.append((name.encode(), encode_entry(DTYPE_CODES[arr.dtype], arr.shape, 0, offset, len(raw),
                                                      mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_index(prefix + ".index", items)
    return prefix


# ------------------------------------------------------------------------------------------------ session glue
# Slot / accumulator names of the reference's optimisers.  The classifier builds
# `AdamOptimizer(lr, name="nn_core/Adam")` (or MomentumOptimizer(name="nn_core/Momentum")) inside
# name_scope("training_optimizer") (reference common/common_nn_ops.py:215-230): TF1 names the slots
# `<variable>/nn_core/Adam`, `<variable>/nn_core/Adam_1` (`<variable>/nn_core/Momentum`) and the power accumulators
# `training_optimizer/beta1_power`, `training_optimizer/beta2_power` -- which is why the reference's Saver restores
# include=["training_optimizer"] (classify/monitored_session_runner.py:164-168).  The GAN optimisers keep the default
# name "Adam" (gan/wrappers/gan_common.py:264-265): `<variable>/Adam`, `<variable>/Adam_1`.
CLASSIFIER_OPTIMIZER = "nn_core/Adam"
CLASSIFIER_MOMENTUM = "nn_core/Momentum"
CLASSIFIER_ACCUMULATOR_SCOPE = "training_optimizer"
# spellings accepted on read, most specific first: (m suffix, v suffix or None for single-slot optimisers)
SLOT_SPELLINGS = (("/nn_core/Adam", "/nn_core/Adam_1"), ("/Adam", "/Adam_1"), ("/nn_core/Momentum", None),
                  ("/Momentum", None))


def session_to_variables(sess, optimizer_name=CLASSIFIER_OPTIMIZER, accumulator_scope=CLASSIFIER_ACCUMULATOR_SCOPE,
                         beta1=0.9, beta2=0.999, momentum=False):
    """The session's state under the names a TF1 Saver would use: variables, global_step, the optimiser slots per
    variable (`<var>/<optimizer_name>`, `<var>/<optimizer_name>_1`) and, for Adam, the beta power accumulators under
    `accumulator_scope` with the session's beta1 / beta2 (GAN sessions: optimizer_name="Adam", beta1=0.5)."""
    d = {v.name: sess.get_variable(v.name) for v in sess.store.order}
    d["global_step"] = numpy.asarray(sess.global_step, numpy.int64)
    m = sess.slot_m.detach().cpu().numpy()
    vv = sess.slot_v.detach().cpu().numpy()
    if momentum and optimizer_name == CLASSIFIER_OPTIMIZER:
        optimizer_name = CLASSIFIER_MOMENTUM
    for v in sess.trainable:
        d[f"{v.name}/{optimizer_name}"] = m[v.offset:v.offset + v.size].reshape(v.shape).copy()
        if not momentum:
            d[f"{v.name}/{optimizer_name}_1"] = vv[v.offset:v.offset + v.size].reshape(v.shape).copy()
    if not momentum:
        t = sess.global_step
        pre = accumulator_scope + "/" if accumulator_scope else ""
        # TF1 Adam keeps beta^(t+1) after t updates (initialised to beta, multiplied once per apply_gradients)
        d[pre + "beta1_power"] = numpy.asarray(beta1 ** (t + 1), numpy.float32)
        d[pre + "beta2_power"] = numpy.asarray(beta2 ** (t + 1), numpy.float32)
    return d


def variables_to_session(sess, variables):
    """Loads what the session holds (other names -- e.g. the training-only reconstruction head when restoring an
    inference graph -- are ignored, as a Saver built from get_variables_to_restore does).  Optimiser slots are
    accepted under every spelling in SLOT_SPELLINGS; a checkpoint that says global_step > 0 but carries no slots
    gets a warning (Adam's bias correction would then assume step t with zero moments)."""
    import torch
    import warnings
    for v in sess.store.order:
        if v.name in variables:
            sess.set_variable(v.name, variables[v.name])
    if "global_step" in variables:
        sess.global_step = int(variables["global_step"])
    if sess.slot_m is not None:
        m = sess.slot_m.detach().cpu().numpy().copy()
        vv = sess.slot_v.detach().cpu().numpy().copy()
        touched = 0
        for v in sess.trainable:
            for sm, sv in SLOT_SPELLINGS:
                a = variables.get(v.name + sm)
                b = variables.get(v.name + sv) if sv is not None else None
                if a is None or (sv is not None and b is None):
                    continue
                if sv is None:  # a Momentum slot: the session is a MomentumOptimizer run (export keeps the spelling)
                    sess.optimizer_kind = "momentum"
                m[v.offset:v.offset + v.size] = numpy.asarray(a, numpy.float32).reshape(-1)
                if b is not None:
                    vv[v.offset:v.offset + v.size] = numpy.asarray(b, numpy.float32).reshape(-1)
                touched += 1
                break
        if touched:
            sess.slot_m.copy_(torch.from_numpy(m))
            sess.slot_v.copy_(torch.from_numpy(vv))
        if touched < len(sess.trainable) and sess.global_step > 0 and \
                any(v.name in variables for v in sess.trainable):
            warnings.warn(f"checkpoint at global_step {sess.global_step} carries optimiser slots for {touched} of "
                          f"{len(sess.trainable)} trainable variables: the missing moments start from zero while "
                          f"the bias correction assumes step {sess.global_step}")
