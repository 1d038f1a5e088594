"""TensorFlow checkpoint bundles without TensorFlow (hypelcnn_amd/common/tf_checkpoint.py): known-answer bytes for
every primitive of the format, round trips, corruption detection, and the session glue (a classifier trained on the
kernel emulation exported as a bundle and restored into a fresh session / an inference graph)."""
import json
import os
import struct

import numpy
import numpy as np
import pytest

from hypelcnn_amd.common import tf_checkpoint as T


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC-32C
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == 0xE3069283          # incremental
    # masking (crc32c.h): rotate right 15, add 0xa282ead8; unmask inverts it
    assert T.mask_crc(0) == 0xA282EAD8
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF, 0x12345678):
        assert T.unmask_crc(T.mask_crc(c)) == c
    assert T.mask_crc(0xE3069283) != 0xE3069283


def test_varints_and_protobuf_known_bytes():
    assert T.put_varint(0) == b"\x00" and T.put_varint(127) == b"\x7f" and T.put_varint(128) == b"\x80\x01"
    assert T.put_varint(300) == b"\xac\x02"                               # the protobuf documentation's example
    for v in (0, 1, 127, 128, 300, 2 ** 31, 2 ** 63 - 1):
        assert T.get_varint(T.put_varint(v) + b"\xff", 0) == (v, len(T.put_varint(v)))
    # BundleHeaderProto{num_shards: 1, version{producer: 1}} = 08 01 1a 02 08 01
    assert T.encode_header(1) == bytes.fromhex("08011a020801")
    # BundleEntryProto{dtype: DT_FLOAT, shape{dim{size:3} dim{size:5}}, offset: 60, size: 60, crc32c: 0x01020304}
    e = T.encode_entry(1, (3, 5), 0, 60, 60, 0x01020304)
    assert e == bytes.fromhex("0801" "1208" "12020803" "12020805" "203c" "283c" "35" "04030201")
    d = T.decode_entry(e)
    assert (d["dtype"], d["shape"], d["shard_id"], d["offset"], d["size"], d["crc32c"]) == (1, (3, 5), 0, 60, 60,
                                                                                          0x01020304)
    assert T.decode_entry(T.encode_entry(9, (), 0, 0, 8, 7))["shape"] == ()   # scalar (global_step)


def test_table_block_layout_known_bytes(tmp_path):
    # one data block with prefix compression: keys "", "ab", "abc" -> shared lengths 0, 0, 2
    block = T._build_block([(b"", b"H"), (b"ab", b"1"), (b"abc", b"22")])
    assert block == (bytes([0, 0, 1]) + b"H" + bytes([0, 2, 1]) + b"ab1" + bytes([2, 1, 2]) + b"c22"
                     + struct.pack("<II", 0, 1))
    assert list(T._block_entries(block)) == [(b"", b"H"), (b"ab", b"1"), (b"abc", b"22")]
    p = str(tmp_path / "t.index")
    T.write_index(p, [(b"", b"H"), (b"ab", b"1"), (b"abc", b"22")])
    raw = open(p, "rb").read()
    assert raw[:len(block)] == block and raw[len(block)] == 0                     # uncompressed block type
    assert struct.unpack_from("<I", raw, len(block) + 1)[0] == T.mask_crc(T.crc32c(block + b"\x00"))
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw[-48:]) == 48
    assert T.read_index(p) == {b"": b"H", b"ab": b"1", b"abc": b"22"}


def test_bundle_round_trip_many_blocks_and_corruption(tmp_path):
    rng = np.random.default_rng(0)
    variables = {f"nn_core/layer_{i}/weights": rng.standard_normal((3, 3, 7, i + 1)).astype(np.float32)
                 for i in range(150)}                                           # > 1 index data block
    variables["global_step"] = np.asarray(12345, np.int64)
    variables["flags"] = np.asarray([True, False, True])
    variables["nn_core/empty"] = np.zeros((0, 4), np.float32)
    prefix = str(tmp_path / "model.ckpt-12345")
    T.write_checkpoint(prefix, variables)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    back = T.read_checkpoint(prefix)
    assert set(back) == set(variables)
    for k, v in variables.items():
        assert back[k].dtype == np.asarray(v).dtype and back[k].shape == np.asarray(v).shape
        np.testing.assert_array_equal(back[k], v)
    only = T.read_checkpoint(prefix, names={"global_step"})
    assert list(only) == ["global_step"] and int(only["global_step"]) == 12345
    # flipped data byte -> tensor checksum error; flipped index byte -> block checksum error
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[100] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(data)
    with pytest.raises(ValueError, match="tensor checksum"):
        T.read_checkpoint(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0x01
    open(prefix + ".index", "wb").write(idx)
    with pytest.raises(ValueError, match="block checksum"):
        T.read_checkpoint(prefix)


def test_session_export_and_restore_through_tf_bundle(tmp_path):
    from hypelcnn_amd.classify import monitored_session_runner as M
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    alg = {"drop_out_ratio": 0.7, "filter_count": 16, "learning_rate": 3e-3, "learning_rate_decay_factor": 0.96,
           "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.9,
           "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 1, "spatial_hierarchy_level": 1,
           "degradation_coeff": 3, "use_residual": True}
    rng = np.random.default_rng(3)
    x = rng.random((8, 3, 3, 5)).astype(np.float32)
    onehot = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 8)]

    def fresh(seed):
        built = U.build("HYPELCNNModel", 3, 5, 3, alg, EmuBackend(), with_eval=False)
        built.ctx.seed = seed
        return built, built.ctx.session()

    b1, s1 = fresh(1)
    for _ in range(3):
        U.run_train_step(b1, x, onehot, {})
        s1.adam_step(3e-3)
    prefix = M.export_tf_checkpoint(s1, str(tmp_path / "model.ckpt-3"))
    names = set(T.read_index(prefix + ".index"))
    # names of the reference's Saver: AdamOptimizer(name="nn_core/Adam") under name_scope("training_optimizer")
    # (common/common_nn_ops.py:215-230; Saver include=["training_optimizer"], monitored_session_runner.py:164-168)
    assert {b"", b"global_step", b"training_optimizer/beta1_power", b"training_optimizer/beta2_power",
            b"nn_core/conv_enc_0/weights", b"nn_core/conv_enc_0/weights/nn_core/Adam",
            b"nn_core/conv_enc_0/weights/nn_core/Adam_1", b"nn_core/fc_final/BatchNorm/moving_variance"} <= names
    assert b"beta1_power" not in names and b"nn_core/conv_enc_0/weights/Adam" not in names
    acc = T.read_checkpoint(prefix, names={"training_optimizer/beta1_power", "training_optimizer/beta2_power"})
    assert abs(float(acc["training_optimizer/beta1_power"]) - 0.9 ** 4) < 1e-7
    assert abs(float(acc["training_optimizer/beta2_power"]) - 0.999 ** 4) < 1e-7
    assert M.latest_checkpoint(str(tmp_path)) == prefix
    b2, s2 = fresh(2)
    assert not np.array_equal(s2.params.numpy(), s1.params.numpy())
    M.restore_checkpoint(s2, prefix)
    np.testing.assert_array_equal(s2.params.numpy(), s1.params.numpy())
    np.testing.assert_array_equal(s2.state.numpy(), s1.state.numpy())
    np.testing.assert_array_equal(s2.slot_m.numpy(), s1.slot_m.numpy())
    np.testing.assert_array_equal(s2.slot_v.numpy(), s1.slot_v.numpy())
    assert s2.global_step == s1.global_step == 3
    # the next step from the restored session is the step the original session takes
    for b, s in ((b1, s1), (b2, s2)):
        U.run_train_step(b, x, onehot, {})
        s.adam_step(3e-3)
    np.testing.assert_array_equal(s2.params.numpy(), s1.params.numpy())
    # the default-named spelling (`<var>/Adam`, what a GAN session or an older export of this build wrote) restores
    # too; a bundle without any slots warns
    allv = T.read_checkpoint(prefix)
    legacy = {k.replace("/nn_core/Adam", "/Adam"): v for k, v in allv.items()}
    T.write_checkpoint(str(tmp_path / "legacy" / "model.ckpt-3"), legacy)
    b3, s3 = fresh(3)
    M.restore_checkpoint(s3, str(tmp_path / "legacy" / "model.ckpt-3"))
    restored = T.read_checkpoint(prefix)
    for v in s3.trainable:
        np.testing.assert_array_equal(s3.slot_v.numpy()[v.offset:v.offset + v.size].reshape(v.shape),
                                      restored[v.name + "/nn_core/Adam_1"])
    T.write_checkpoint(str(tmp_path / "bare" / "model.ckpt-3"),
                       {k: v for k, v in allv.items() if "Adam" not in k and "_power" not in k})
    b4, s4 = fresh(4)
    with pytest.warns(UserWarning, match="optimiser slots"):
        M.restore_checkpoint(s4, str(tmp_path / "bare" / "model.ckpt-3"))
    assert s4.global_step == 3 and float(s4.slot_m.abs().max()) == 0.0


def test_momentum_session_round_trips_under_the_momentum_slot_names(tmp_path):
    """A MomentumOptimizer session (alg_param_concnn.json; common/common_nn_ops.py:225) is exported as
    `<var>/nn_core/Momentum` -- one slot, no beta powers -- and restores to the same optimiser state."""
    from hypelcnn_amd.classify import monitored_session_runner as M
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    alg = {"drop_out_ratio": 0.7, "filter_count": 16, "learning_rate": 3e-3, "learning_rate_decay_factor": 0.96,
           "learning_rate_decay_step": 350, "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.9,
           "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 1, "spatial_hierarchy_level": 1,
           "degradation_coeff": 3, "use_residual": True}
    rng = np.random.default_rng(5)
    x = rng.random((8, 3, 3, 5)).astype(np.float32)
    onehot = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 8)]

    def fresh(seed):
        built = U.build("HYPELCNNModel", 3, 5, 3, alg, EmuBackend(), with_eval=False)
        built.ctx.seed = seed
        return built, built.ctx.session()

    b1, s1 = fresh(1)
    for _ in range(3):
        U.run_train_step(b1, x, onehot, {})
        s1.momentum_step(1e-2, 0.9)
    prefix = M.export_tf_checkpoint(s1, str(tmp_path / "model.ckpt-3"))
    names = set(T.read_index(prefix + ".index"))
    assert b"nn_core/conv_enc_0/weights/nn_core/Momentum" in names
    assert not any(b"Adam" in n or b"_power" in n for n in names), sorted(names)[:8]
    b2, s2 = fresh(2)
    M.restore_checkpoint(s2, prefix)
    np.testing.assert_array_equal(s2.params.numpy(), s1.params.numpy())
    np.testing.assert_array_equal(s2.slot_m.numpy(), s1.slot_m.numpy())
    assert s2.global_step == 3
    for b, s in ((b1, s1), (b2, s2)):
        U.run_train_step(b, x, onehot, {})
        s.momentum_step(1e-2, 0.9)
    np.testing.assert_array_equal(s2.params.numpy(), s1.params.numpy())


def test_gan_session_exports_default_adam_names_with_its_beta1(tmp_path):
    """GAN optimisers keep TF's default name and beta1 = 0.5 (gan/wrappers/gan_common.py:264-265)."""
    from tests import parity_util as U
    from tests.emu_backend import EmuBackend
    built = U.build("HYPELCNNModel", 3, 5, 3, {"drop_out_ratio": 0.3, "filter_count": 32, "learning_rate": 3e-3,
                                               "learning_rate_decay_factor": 0.96, "learning_rate_decay_step": 350,
                                               "lrelu_alpha": 0.18, "optimizer": "AdamOptimizer", "bn_decay": 0.9,
                                               "l2regularizer_scale": 1e-5, "spectral_hierarchy_level": 1,
                                               "spatial_hierarchy_level": 1, "degradation_coeff": 3,
                                               "use_residual": True}, EmuBackend(), with_eval=False)
    sess = built.ctx.session()
    sess.global_step = 2
    d = T.session_to_variables(sess, optimizer_name="Adam", accumulator_scope="", beta1=0.5)
    assert "nn_core/conv_enc_0/weights/Adam_1" in d and abs(float(d["beta1_power"]) - 0.5 ** 3) < 1e-7


# ------------------------------------------------------------------------------------------------ TFRecord files
def test_tfrecord_example_known_bytes():
    from hypelcnn_amd.common.tfrecord_io import decode_example, encode_example
    # hand-assembled tf.train.Example{features{feature{key:"a" value{int64_list{value:[3]}}}}}
    want = bytes([0x0a, 0x0c, 0x0a, 0x0a, 0x0a, 0x01, ord("a"), 0x12, 0x05, 0x1a, 0x03, 0x0a, 0x01, 0x03])
    assert encode_example({"a": numpy.asarray([3])}) == want
    got = decode_example(want)
    assert got["a"].tolist() == [3]
    # float_list packed: field 2 -> 0x12, inner field 1 wire type 2
    ex = encode_example({"f": numpy.asarray([1.0, -2.5], numpy.float32)})
    assert ex[-8:] == numpy.asarray([1.0, -2.5], "<f4").tobytes()
    assert decode_example(ex)["f"].tolist() == [1.0, -2.5]
    # unpacked repeated encodings (older writers) decode too: int64_list{value:1 value:-1}
    neg = bytes([0x08, 0x01, 0x08] + [0xff] * 9 + [0x01])
    unpacked = b"\x0a" + bytes([len(neg) + 9]) + b"\x0a" + bytes([len(neg) + 7]) + b"\x0a\x01k\x12" + \
        bytes([len(neg) + 2]) + b"\x1a" + bytes([len(neg)]) + neg
    assert decode_example(unpacked)["k"].tolist() == [1, -1]
    assert decode_example(encode_example({"b": [b"xy", b""], "n": numpy.asarray([-7, 1 << 40])}))["n"].tolist() == \
        [-7, 1 << 40]


@pytest.mark.parametrize("compressed", [False, True])
def test_tfrecord_framing_round_trip(tmp_path, compressed):
    import struct
    from hypelcnn_amd.common.tf_checkpoint import crc32c, mask_crc
    from hypelcnn_amd.common.tfrecord_io import read_records, write_records
    path = str(tmp_path / "x.tfrecord")
    payloads = [b"", b"abc", bytes(range(256)) * 40]
    assert write_records(path, payloads, compressed) == 3
    assert list(read_records(path)) == payloads
    if not compressed:
        raw = open(path, "rb").read()
        # first record: zero length, crc of eight zero bytes, no payload, crc of the empty string (masked 0 -> delta)
        assert raw[:8] == b"\0" * 8 and struct.unpack("<I", raw[8:12])[0] == mask_crc(crc32c(b"\0" * 8))
        assert struct.unpack("<I", raw[12:16])[0] == 0xa282ead8
        bad = bytearray(raw)
        bad[16 + 12 + 1] ^= 1  # a payload byte of the second record
        open(path, "wb").write(bytes(bad))
        with pytest.raises(ValueError):
            list(read_records(path))


def test_tfrecord_importer_matches_in_memory(tmp_path):
    """utilities/tfrecord_writer export -> TFRecordImporter gives the in-memory importer's arrays back."""
    from hypelcnn_amd.common.common_nn_ops import get_importer_from_name
    from hypelcnn_amd.utilities import tfrecord_writer
    from hypelcnn_amd.importer.TFRecordImporter import TFRecordImporter
    out = str(tmp_path / "rec")
    spec = "grss2018:h=24:w=28:bands=12:classes=4"
    tfrecord_writer.main(["--loader_name", "SyntheticDataLoader", "--path", spec, "--neighborhood", "1",
                          "--train_ratio", "0.2", "--target_path", out, "--compressed", "True"])
    mem = get_importer_from_name("InMemoryImporter").read_data_set("SyntheticDataLoader", spec, 0.2, 0.05, 1, True)
    imp = get_importer_from_name("TFRecordImporter")
    assert isinstance(imp, TFRecordImporter) and imp.requires_separate_validation_branch() is False
    train, test, val, shadow, class_range, scene_shape, colors = imp.read_data_set(
        "SyntheticDataLoader", spec + ":base_dir=" + out, 0.2, 0.05, 1, True)
    assert shadow is None and scene_shape is None and class_range == range(0, 4)
    for info, ref in ((train, mem[0]), (test, mem[1]), (val, mem[2])):
        assert tuple(info.data.shape) == ref.data.shape
        data, labels = imp.load_file(info.path, tuple(info.data.shape[1:]))
        numpy.testing.assert_array_equal(data, ref.data)
        numpy.testing.assert_array_equal(labels, ref.labels)
    testing_t, training_t, validation_t = imp.convert_data_to_tensor(test, train, val, class_range)
    assert validation_t is testing_t and training_t.dataset.element_shape == (3, 3, 13)
    assert training_t.dataset.class_count == 4
