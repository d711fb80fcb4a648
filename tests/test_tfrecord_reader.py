"""TFRecord contract (reference tfrecords/tfrecord_reader.py + tfr_util.py): framing, CRC32C, Example parsing,
batching, rank sharding.  Uses the host helper xpt_crc32c from the C-ABI library (no GPU compute)."""
import json
import os
import struct

import numpy as np
import pytest
import torch

from xpt_mde_2021_amd.tfrecords import tfrecord_reader as tr

SNIPPET, H, W = 5, 8, 12


def _example(i, stereo=False):
    rng = np.random.default_rng(i)
    ex = {"image": rng.integers(0, 256, (SNIPPET * H, W, 3), dtype=np.uint8),
          "intrinsic": np.array([[W / 2, 0, W / 2], [0, H / 2, H / 2], [0, 0, 1]], dtype=np.float32),
          "depth_gt": rng.random((H, W, 1), dtype=np.float32),
          "pose_gt": np.tile(np.eye(4, dtype=np.float32), (4, 1, 1)),
          "index": i}
    if stereo:
        ex["image_R"] = rng.integers(0, 256, (SNIPPET * H, W, 3), dtype=np.uint8)
        ex["intrinsic_R"] = ex["intrinsic"].copy()
        ex["stereo_T_LR"] = np.eye(4, dtype=np.float32)
    return ex


def _write(path, n, shard_size=2000, stereo=False):
    wr = tr.TfrecordWriter(str(path), shard_size=shard_size)
    for i in range(n):
        wr.write(_example(i, stereo))
    wr.close((SNIPPET, H, W, 3))


def test_crc32c_known_answers():
    lib = tr._lib.load()
    assert lib.xpt_crc32c(b"123456789", 9) == 0xE3069283        # the CRC-32C check value (RFC 3720 B.4)
    assert lib.xpt_crc32c(b"\x00" * 32, 32) == 0x8A9136AA        # RFC 3720 B.4: 32 bytes of zeros
    assert lib.xpt_crc32c(b"\xff" * 32, 32) == 0x62A8AB43        # RFC 3720 B.4: 32 bytes of ones
    assert lib.xpt_crc32c(bytes(range(32)), 32) == 0x46DD794E    # RFC 3720 B.4: incrementing bytes
    assert lib.xpt_crc32c(b"", 0) == 0


def test_example_wire_roundtrip():
    ex = _example(3, stereo=True)
    parsed = tr.parse_example(tr.serialize_example(ex))
    assert set(parsed) == set(ex)
    assert parsed["index"] == 3
    assert parsed["image"] == ex["image"].tobytes()
    assert np.array_equal(np.frombuffer(parsed["pose_gt"], np.float32).reshape(4, 4, 4), ex["pose_gt"])
    big = tr.parse_example(tr.serialize_example({"index": 2 ** 40 + 5}))
    assert big["index"] == 2 ** 40 + 5


def test_config_sidecar(tmp_path):
    _write(tmp_path, 3)
    cfg = json.load(open(tmp_path / "tfr_config.txt"))
    assert cfg["length"] == 3 and cfg["imshape"] == [SNIPPET, H, W, 3]
    assert cfg["image"] == {"parse_type": "tf.string", "decode_type": "tf.uint8", "shape": [SNIPPET * H, W, 3]}
    assert cfg["intrinsic"]["decode_type"] == "tf.float32"
    assert cfg["index"]["parse_type"] == "tf.int64"


def test_reader_batches_and_contract(tmp_path):
    _write(tmp_path, 11, shard_size=4, stereo=True)
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".tfrecord")]) == 3
    rd = tr.TfrecordReader(str(tmp_path), shuffle=False, batch_size=4)
    assert rd.get_total_steps() == 2
    batches = list(rd.get_dataset())
    assert len(batches) == 2                                      # drop_remainder
    b0 = batches[0]
    assert b0["image"].shape == (4, SNIPPET * H, W, 3) and b0["image"].dtype == torch.float32
    assert b0["image5d"].shape == (4, SNIPPET, H, W, 3) and b0["image5d_R"].shape == (4, SNIPPET, H, W, 3)
    assert b0["intrinsic"].shape == (4, 3, 3) and b0["depth_gt"].shape == (4, H, W, 1)
    assert b0["pose_gt"].shape == (4, 4, 4, 4) and b0["stereo_T_LR"].shape == (4, 4, 4)
    assert b0["index"].tolist() == [0, 1, 2, 3] and batches[1]["index"].tolist() == [4, 5, 6, 7]
    expect = _example(2, True)["image"].astype(np.float32) / 255. * 2 - 1
    assert np.allclose(b0["image"][2].numpy(), expect, atol=1e-6)
    assert b0["image"].min() >= -1 and b0["image"].max() <= 1
    assert torch.equal(b0["image5d"][1, 3], b0["image"][1, 3 * H:4 * H])


def test_reader_epochs_shuffle_and_ranks(tmp_path):
    _write(tmp_path, 12, shard_size=5)
    rd = tr.TfrecordReader(str(tmp_path), shuffle=True, epochs=2, batch_size=3, shuffle_buffer=4, seed=1)
    seen = [i for b in rd.get_dataset() for i in b["index"].tolist()]
    assert sorted(seen) == sorted(list(range(12)) * 2) and seen != sorted(seen)
    parts = []
    for rank in range(2):
        rd = tr.TfrecordReader(str(tmp_path), batch_size=3, rank=rank, world_size=2)
        assert rd.get_total_steps() == 2
        parts.append([i for b in rd.get_dataset() for i in b["index"].tolist()])
    assert parts[0] == [0, 2, 4, 6, 8, 10] and parts[1] == [1, 3, 5, 7, 9, 11]


def test_corruption_detected(tmp_path):
    _write(tmp_path, 2)
    fn = tmp_path / "shard_000.tfrecord"
    raw = bytearray(fn.read_bytes())
    raw[40] ^= 0x01
    fn.write_bytes(bytes(raw))
    with pytest.raises(IOError):
        list(tr.TfrecordReader(str(tmp_path), batch_size=1).get_dataset())
    assert len(list(tr.TfrecordReader(str(tmp_path), batch_size=1, verify_crc=False).get_dataset())) == 2


def test_record_framing_bytes(tmp_path):
    _write(tmp_path, 1)
    raw = (tmp_path / "shard_000.tfrecord").read_bytes()
    (length,) = struct.unpack("<Q", raw[:8])
    assert len(raw) == 8 + 4 + length + 4
    assert struct.unpack("<I", raw[8:12])[0] == tr.masked_crc32c(raw[:8])


def test_bad_config_rejected(tmp_path):
    _write(tmp_path, 1)
    cfg = json.load(open(tmp_path / "tfr_config.txt"))
    cfg["image"]["decode_type"] = "tf.float64"
    json.dump(cfg, open(tmp_path / "tfr_config.txt", "w"))
    with pytest.raises(TypeError):
        tr.TfrecordReader(str(tmp_path), batch_size=1)


@pytest.mark.parametrize("shuffle", [False, True])
def test_prefetching_reader_yields_the_same_batches(tmp_path, shuffle):
    """prefetch > 0: a producer thread (with decode workers) reads ahead -- tf.data's role in tfrecord_reader.py:61-108 --
    and must hand out exactly the batches of the synchronous generator: same order, same keys, same values."""
    _write(tmp_path, 23, shard_size=6, stereo=True)
    kw = dict(shuffle=shuffle, epochs=2, batch_size=4, shuffle_buffer=5, seed=3)
    plain = list(tr.TfrecordReader(str(tmp_path), **kw).get_dataset())
    ahead = list(tr.TfrecordReader(str(tmp_path), prefetch=2, workers=3, **kw).get_dataset())
    assert len(plain) == len(ahead) == (23 * 2) // 4
    for a, b in zip(plain, ahead):
        assert sorted(a) == sorted(b)
        for k in a:
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k


def test_prefetching_reader_propagates_errors_and_stops_early(tmp_path):
    _write(tmp_path, 6)
    fn = tmp_path / "shard_000.tfrecord"
    raw = bytearray(fn.read_bytes())
    raw[len(raw) // 2] ^= 0xFF
    fn.write_bytes(bytes(raw))
    with pytest.raises(IOError):
        list(tr.TfrecordReader(str(tmp_path), batch_size=1, prefetch=2).get_dataset())
    _write(tmp_path, 40)
    it = iter(tr.TfrecordReader(str(tmp_path), batch_size=2, prefetch=1, epochs=50).get_dataset())
    first = next(it)
    assert first["index"].tolist() == [0, 1]
    it.close()                                   # the producer thread is told to stop (no leak of a blocked thread)
    import threading
    import time
    time.sleep(0.5)
    assert not [t for t in threading.enumerate() if t.name == "xpt-tfrecord-prefetch" and t.is_alive()]


def test_prefetching_reader_keeps_its_descriptor_count_over_epochs(tmp_path):
    """train() iterates ONE dataset object every epoch: the shard mappings must be reused, not re-created (a mapping per
    shard and epoch ran a long training into EMFILE)."""
    _write(tmp_path, 16, shard_size=4)
    ds = tr.TfrecordReader(str(tmp_path), batch_size=2, prefetch=2, workers=2).get_dataset()
    counts = []
    for _ in range(5):
        assert len(list(ds)) == 8
        counts.append(len(os.listdir("/proc/self/fd")))
    # (no growth: a leak adds a descriptor per shard and epoch; an earlier test's worker thread letting go of its own files
    #  while this one runs may LOWER the count)
    assert max(counts[1:]) <= counts[0], counts
    assert len(ds._maps) == 4
    ds.close()
    assert not ds._maps
