"""The TF-free TFRecord reader against records it did NOT write: the test encodes `tf.train.Example` messages with the
official protobuf runtime (schema of tensorflow/core/example/{example,feature}.proto built from descriptors: Example{
features=1} / Features{map<string,Feature> feature=1} / Feature{oneof bytes_list=1, float_list=2, int64_list=3}), frames
them as TFRecords with a pure-Python CRC-32C (uint64 length, masked crc of the length, payload, masked crc of the payload),
and writes the sidecar `tfr_config.txt` in the form the reference's writer produces (tfrecords/tfr_util.py:47-77,
tfrecords/tfrecord_reader.py:11-45).  What comes out of TfrecordReader must be what went in, in the reference's feature-dict
contract (tfrecord_reader.py:61-108: image -> float32 in [-1, 1], image5d reshaped by imshape)."""
import json
import struct

import numpy as np
import pytest
import torch

SNIPPET, H, W = 5, 6, 10


def crc32c(data):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), bit by bit."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def example_classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="xpt_test_example.proto", package="xpttest", syntax="proto3")

    def add_list(name, ftype):
        m = fd.message_type.add(name=name)
        m.field.add(name="value", number=1, type=ftype, label=descriptor_pb2.FieldDescriptorProto.LABEL_REPEATED)
    T = descriptor_pb2.FieldDescriptorProto
    add_list("BytesList", T.TYPE_BYTES)
    add_list("FloatList", T.TYPE_FLOAT)
    add_list("Int64List", T.TYPE_INT64)
    feat = fd.message_type.add(name="Feature")
    feat.oneof_decl.add(name="kind")
    for i, (n, t) in enumerate((("bytes_list", "BytesList"), ("float_list", "FloatList"), ("int64_list", "Int64List")), 1):
        feat.field.add(name=n, number=i, type=T.TYPE_MESSAGE, type_name=f".xpttest.{t}", label=T.LABEL_OPTIONAL, oneof_index=0)
    feats = fd.message_type.add(name="Features")
    entry = feats.nested_type.add(name="FeatureEntry")
    entry.options.map_entry = True
    entry.field.add(name="key", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    entry.field.add(name="value", number=2, type=T.TYPE_MESSAGE, type_name=".xpttest.Feature", label=T.LABEL_OPTIONAL)
    feats.field.add(name="feature", number=1, type=T.TYPE_MESSAGE, type_name=".xpttest.Features.FeatureEntry",
                    label=T.LABEL_REPEATED)
    ex = fd.message_type.add(name="Example")
    ex.field.add(name="features", number=1, type=T.TYPE_MESSAGE, type_name=".xpttest.Features", label=T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("xpttest.Example"))


def make_examples(n):
    rng = np.random.default_rng(7)
    out = []
    for i in range(n):
        out.append({"image": rng.integers(0, 256, (SNIPPET * H, W, 3), dtype=np.uint8),
                    "intrinsic": np.array([[5.0, 0, 5.0], [0, 6.0, 3.0], [0, 0, 1]], dtype=np.float32),
                    "depth_gt": rng.random((H, W, 1), dtype=np.float32) * 50,
                    "pose_gt": rng.standard_normal((SNIPPET - 1, 4, 4)).astype(np.float32)})
    return out


def test_crc32c_reference_values():
    assert crc32c(b"123456789") == 0xE3069283 and crc32c(b"\x00" * 32) == 0x8A9136AA          # RFC 3720 B.4


def test_reader_consumes_protobuf_encoded_records(tmp_path):
    from xpt_mde_2021_amd.tfrecords.tfrecord_reader import TfrecordReader
    Example = example_classes()
    examples = make_examples(4)
    with open(tmp_path / "shard_000.tfrecord", "wb") as f:
        for ex in examples:
            msg = Example()
            for key, arr in ex.items():
                msg.features.feature[key].bytes_list.value.append(arr.tobytes())
            payload = msg.SerializeToString()
            header = struct.pack("<Q", len(payload))
            f.write(header + struct.pack("<I", masked(crc32c(header))) + payload + struct.pack("<I", masked(crc32c(payload))))
    config = {k: {"parse_type": "tf.string", "decode_type": "tf.uint8" if v.dtype == np.uint8 else "tf.float32",
                  "shape": list(v.shape)} for k, v in examples[0].items()}
    config["length"] = len(examples)
    config["imshape"] = [SNIPPET, H, W, 3]
    with open(tmp_path / "tfr_config.txt", "w") as f:
        json.dump(config, f)
    reader = TfrecordReader(str(tmp_path), shuffle=False, batch_size=2)
    batches = list(reader.get_dataset())
    assert len(batches) == 2 and reader.get_total_steps() == 2
    for b, feats in enumerate(batches):
        for j in range(2):
            ex = examples[2 * b + j]
            image = ex["image"].astype(np.float32) / 255.0 * 2.0 - 1.0                       # util_funcs.py:79-80 to_float_image
            assert torch.allclose(feats["image"][j], torch.from_numpy(image), atol=1e-6)
            assert torch.allclose(feats["image5d"][j], torch.from_numpy(image.reshape(SNIPPET, H, W, 3)), atol=1e-6)
            for key in ("intrinsic", "depth_gt", "pose_gt"):
                assert torch.equal(feats[key][j], torch.from_numpy(ex[key])), key
    # a corrupted payload byte is caught by the record checksum
    raw = bytearray(open(tmp_path / "shard_000.tfrecord", "rb").read())
    raw[40] ^= 0xFF
    open(tmp_path / "shard_000.tfrecord", "wb").write(bytes(raw))
    with pytest.raises(Exception):
        list(TfrecordReader(str(tmp_path), shuffle=False, batch_size=2).get_dataset())
