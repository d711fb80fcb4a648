"""TF-free TFRecord reader (and a matching writer) honouring the reference's input contract
(tfrecords/tfrecord_reader.py:11-114; writer side tfrecords/tfr_util.py:8-77).

Format read here, exactly what `tf.data.TFRecordDataset` + `tf.io.parse_single_example` + `tf.io.decode_raw` consume:
  * directory with `*.tfrecord` shards (read in sorted order) and the JSON side-car `tfr_config.txt`
    ({key: {"parse_type": "tf.string"|"tf.int64", "decode_type": "tf.uint8"|"tf.float32", "shape": [...]}, ...,
    "length": int, "imshape": [snippet, H, W, 3]});
  * record framing: uint64 length | uint32 masked_crc32c(length) | payload | uint32 masked_crc32c(payload);
  * payload: a serialized `tf.train.Example` whose features are one `bytes_list` (ndarray.tobytes()) or one
    `int64_list` value per key.  The protobuf wire format is parsed by hand (varints, length-delimited fields).

get_dataset() yields feature dicts of torch tensors with the reference's keys / shapes / dtypes: `image` float32
[B, 5H, W, 3] in [-1, 1] (uint8/255*2-1, utils/util_funcs.py:79-80), `image5d` [B, 5, H, W, 3], `intrinsic` [B,3,3],
`depth_gt` [B,H,W,1], `pose_gt` [B,4,4,4] and, when present, the `_R` variants and `stereo_T_LR`; shuffle buffer 200,
`batch(drop_remainder=True)`.  In data-parallel runs every rank reads the examples `rank, rank+world, ...`.
"""
import ctypes
import glob
import json
import os.path as op
import random
import struct

import numpy as np
import torch

from ..hip import lib as _lib

_MASK_DELTA = 0xA282EAD8


def masked_crc32c(data):
    crc = _lib.load().xpt_crc32c(ctypes.c_char_p(bytes(data)), len(data))
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yields (field_number, wire_type, value) of one message; value = int for varints, memoryview for bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wire == 5:
            val = buf[pos:pos + 4]
            pos += 4
        elif wire == 1:
            val = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        yield field, wire, val


def parse_example(payload):
    """Serialized tf.train.Example -> {key: bytes | int}  (bytes_list[0] / int64_list[0] / float_list[0])."""
    out = {}
    view = memoryview(payload)
    for f, _, features in _fields(view):
        if f != 1:
            continue
        for f2, _, entry in _fields(features):               # map<string, Feature> entries
            if f2 != 1:
                continue
            key, feature = None, None
            for f3, _, val in _fields(entry):
                if f3 == 1:
                    key = bytes(val).decode("utf-8")
                elif f3 == 2:
                    feature = val
            if key is None or feature is None:
                continue
            for kind, _, lst in _fields(feature):             # oneof: 1 bytes_list, 2 float_list, 3 int64_list
                for f5, wire, val in _fields(lst):
                    if f5 != 1:
                        continue
                    if kind == 1:
                        out[key] = bytes(val)
                    elif kind == 3:
                        out[key] = _varint(val, 0)[0] if wire == 2 else val
                    elif kind == 2:
                        out[key] = struct.unpack("<f", bytes(val[:4]))[0]
                    break
    return out


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(example_dict):
    """{key: ndarray | int} -> serialized tf.train.Example (the reference's Serializer, tfr_util.py:8-44)."""
    entries = b""
    for key, value in example_dict.items():
        if value is None:
            continue
        if isinstance(value, np.ndarray):
            feature = _ld(1, _ld(1, value.tobytes()))
        elif isinstance(value, (int, np.integer)):
            feature = _ld(3, _ld(1, _enc_varint(int(value))))
        else:
            raise TypeError(f"[serialize_example] wrong data type: {type(value)}")
        entries += _ld(1, _ld(1, key.encode("utf-8")) + _ld(2, feature))
    return _ld(1, entries)


def read_data_config(key, value):
    """tfr_util.py:56-77: the per-key entry of tfr_config.txt."""
    if isinstance(value, np.ndarray):
        decode = {np.dtype(np.uint8): "tf.uint8", np.dtype(np.float32): "tf.float32"}.get(value.dtype)
        assert decode is not None, f"[read_data_config] Wrong numpy type: {value.dtype}, key={key}"
        return {"parse_type": "tf.string", "decode_type": decode, "shape": list(value.shape)}
    assert isinstance(value, (int, np.integer)), f"[read_data_config] Wrong type: {type(value)}, key={key}"
    return {"parse_type": "tf.int64", "decode_type": "", "shape": None}


class TfrecordWriter:
    """Writes shards + tfr_config.txt in the reference's format (used by tests and to export synthetic data)."""

    def __init__(self, tfrpath, shard_size=2000):
        import os
        os.makedirs(tfrpath, exist_ok=True)
        self.tfrpath, self.shard_size = tfrpath, shard_size
        self.count, self.shard, self.fp, self.config = 0, 0, None, None

    def write(self, example_dict):
        if self.fp is None:
            self._open()
        elif self.count % self.shard_size == 0:
            self._rollover()
            self._open()
        payload = serialize_example(example_dict)
        header = struct.pack("<Q", len(payload))
        self.fp.write(header + struct.pack("<I", masked_crc32c(header)) + payload + struct.pack("<I", masked_crc32c(payload)))
        if self.config is None:
            self.config = {k: read_data_config(k, v) for k, v in example_dict.items() if v is not None}
        self.count += 1

    def _rollover(self):
        self.fp.close()
        self.shard += 1
        return True

    def _open(self):
        self.fp = open(op.join(self.tfrpath, f"shard_{self.shard:03d}.tfrecord"), "wb")

    def close(self, imshape):
        if self.fp is not None:
            self.fp.close()
        cfg = dict(self.config or {})
        cfg["length"] = self.count
        cfg["imshape"] = list(imshape)
        with open(op.join(self.tfrpath, "tfr_config.txt"), "w") as fw:
            json.dump(cfg, fw)


def iterate_records(filename, verify_crc=True):
    with open(filename, "rb") as f:
        while True:
            header = f.read(8)
            if len(header) < 8:
                return
            (length,) = struct.unpack("<Q", header)
            (hcrc,) = struct.unpack("<I", f.read(4))
            payload = f.read(length)
            (pcrc,) = struct.unpack("<I", f.read(4))
            if verify_crc and (hcrc != masked_crc32c(header) or pcrc != masked_crc32c(payload)):
                raise IOError(f"corrupted TFRecord in {filename} (CRC32C mismatch)")
            yield payload


class TfrecordReader:
    def __init__(self, tfrpath, shuffle=False, epochs=1, batch_size=None, rank=0, world_size=1, device="cpu",
                 verify_crc=True, shuffle_buffer=200, seed=0):
        from ..config import opts
        self.tfrpath = tfrpath
        self.shuffle = shuffle
        self.epochs = epochs
        self.batch_size = opts.BATCH_SIZE if batch_size is None else batch_size
        self.rank, self.world_size = rank, world_size
        self.device = device
        self.verify_crc = verify_crc
        self.shuffle_buffer = shuffle_buffer
        self.rng = random.Random(seed + rank)
        self.config = self.read_tfrecord_config(tfrpath)

    def read_tfrecord_config(self, tfrpath):
        """tfrecord_reader.py:20-45."""
        with open(op.join(tfrpath, "tfr_config.txt"), "r") as fr:
            config = json.load(fr)
        for key, feat_conf in config.items():
            if not isinstance(feat_conf, dict):
                continue
            if feat_conf["parse_type"] not in ("tf.string", "tf.int64"):
                raise TypeError("[read_tfrecord_config] invalid parse_type")
            if feat_conf["parse_type"] == "tf.string" and feat_conf["decode_type"] not in ("tf.uint8", "tf.float32"):
                raise TypeError("[read_tfrecord_config] invalid decode_type")
        return config

    def decode_example(self, payload):
        """tfrecord_reader.py:77-98: parse, decode_raw, reshape, uint8 image -> float [-1, 1], add image5d."""
        parsed = parse_example(payload)
        decoded = {}
        for key, feat_conf in self.config.items():
            if not isinstance(feat_conf, dict) or key not in parsed:
                continue
            if feat_conf["parse_type"] == "tf.int64":
                decoded[key] = torch.tensor(parsed[key], dtype=torch.int64)
                continue
            dtype = np.uint8 if feat_conf["decode_type"] == "tf.uint8" else np.float32
            arr = np.frombuffer(parsed[key], dtype=dtype)
            if feat_conf["shape"] is not None:
                arr = arr.reshape(feat_conf["shape"])
            decoded[key] = torch.from_numpy(arr.copy())
        for sfx in ("", "_R"):
            if "image" + sfx in decoded:
                img = decoded["image" + sfx].to(torch.float32) * (2.0 / 255.0) - 1.0
                decoded["image" + sfx] = img
                decoded["image5d" + sfx] = img.reshape(self.config["imshape"])
        return decoded

    def _examples(self):
        filenames = sorted(glob.glob(op.join(self.tfrpath, "*.tfrecord")))
        index = 0
        for _ in range(self.epochs):
            for filename in filenames:
                for payload in iterate_records(filename, self.verify_crc):
                    if index % self.world_size == self.rank:
                        yield self.decode_example(payload)
                    index += 1

    def get_dataset(self):
        """Generator of batched feature dicts on `device` (tfrecord_reader.py:61-108)."""
        return _BatchedDataset(self)

    def get_total_steps(self):
        return self.config["length"] // (self.batch_size * self.world_size)

    def get_tfr_config(self):
        return self.config


class _BatchedDataset:
    def __init__(self, reader):
        self.reader = reader

    def __iter__(self):
        rd = self.reader
        buffer, batch = [], []
        source = rd._examples()

        def emit(example):
            batch.append(example)
            if len(batch) == rd.batch_size:
                out = {k: torch.stack([b[k] for b in batch]).to(rd.device, non_blocking=True) for k in batch[0]}
                batch.clear()
                return out
            return None

        for example in source:
            if rd.shuffle:
                buffer.append(example)
                if len(buffer) < rd.shuffle_buffer:
                    continue
                example = buffer.pop(rd.rng.randrange(len(buffer)))
            out = emit(example)
            if out is not None:
                yield out
        while buffer:
            out = emit(buffer.pop(rd.rng.randrange(len(buffer))))
            if out is not None:
                yield out                                           # a trailing partial batch is dropped
