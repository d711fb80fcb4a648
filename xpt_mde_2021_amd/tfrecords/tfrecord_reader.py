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

`prefetch=n > 0` is the counterpart of tf.data's background threads (tfrecord_reader.py:61-108 hands the step a
`tf.data` pipeline): a producer thread reads, CRC-checks, parses and decodes ahead of the training step (`workers`
decode threads: the CRC, the buffer copies and the tensor ops release the GIL), keeps the images uint8 until the batch is
assembled in PINNED host memory, sends it to the device on a side stream (a quarter of the float bytes) and converts it
there; the consumer finds up to n ready batches and only waits on an event.  Same batches, same order, same values as
the synchronous generator (tests/test_tfrecord_reader.py).
"""
import concurrent.futures
import ctypes
import glob
import json
import os.path as op
import queue
import random
import struct
import threading

import numpy as np
import torch

from ..hip import lib as _lib

_MASK_DELTA = 0xA282EAD8


def default_workers():
    """Decode threads of the prefetching reader: the cores this process may run on (os.sched_getaffinity), shared among the
    ranks of the node, minus one for the training thread; at least 2, at most 16."""
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 4
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    return max(2, min(16, cores // ranks - 1))


def masked_crc32c(data):
    if isinstance(data, memoryview):             # zero-copy: the address of the (memory-mapped) bytes
        arr = np.frombuffer(data, dtype=np.uint8)
        crc = _lib.load().xpt_crc32c(ctypes.c_void_p(arr.ctypes.data), arr.size) if arr.size else 0
    else:
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


def iterate_records(filename, verify_crc=True, deferred=False):
    """Payloads of one shard.  deferred=True: yields (payload, check) and leaves the payload's CRC to check() -- the
    prefetching reader runs it in its decode workers, off the sequential read path."""
    with open(filename, "rb", buffering=1 << 22) as f:
        while True:
            header = f.read(8)
            if len(header) < 8:
                return
            (length,) = struct.unpack("<Q", header)
            (hcrc,) = struct.unpack("<I", f.read(4))
            payload = f.read(length)
            (pcrc,) = struct.unpack("<I", f.read(4))
            if verify_crc and hcrc != masked_crc32c(header):
                raise IOError(f"corrupted TFRecord in {filename} (CRC32C mismatch)")

            def check(payload=payload, pcrc=pcrc):
                if verify_crc and pcrc != masked_crc32c(payload):
                    raise IOError(f"corrupted TFRecord in {filename} (CRC32C mismatch)")
            if deferred:
                yield payload, check
            else:
                check()
                yield payload


class TfrecordReader:
    def __init__(self, tfrpath, shuffle=False, epochs=1, batch_size=None, rank=0, world_size=1, device="cpu",
                 verify_crc=True, shuffle_buffer=200, seed=0, prefetch=0, workers=4):
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
        self.prefetch, self.workers = int(prefetch), max(int(workers), 1)
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

    def decode_example(self, payload, raw_images=False):
        """tfrecord_reader.py:77-98: parse, decode_raw, reshape, uint8 image -> float [-1, 1], add image5d.
        raw_images: leave the images uint8 (finish_images converts the assembled batch, on the device when there is one)."""
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
            decoded[key] = torch.from_numpy(arr.copy())          # (the payload buffer is read-only and short-lived)
        return decoded if raw_images else self.finish_images(decoded)

    def finish_images(self, feats):
        """uint8 image -> float32 in [-1, 1] (util_funcs.py:79-80) + the image5d view; works on one example or a batch."""
        for sfx in ("", "_R"):
            img = feats.get("image" + sfx)
            if img is not None:
                if img.dtype == torch.uint8:
                    img = img.to(torch.float32) * (2.0 / 255.0) - 1.0
                    feats["image" + sfx] = img
                lead = tuple(img.shape[:img.dim() - 3])
                feats["image5d" + sfx] = img.reshape(lead + tuple(self.config["imshape"]))
        return feats

    def _payloads(self, deferred=False):
        filenames = sorted(glob.glob(op.join(self.tfrpath, "*.tfrecord")))
        index = 0
        for _ in range(self.epochs):
            for filename in filenames:
                for payload in iterate_records(filename, self.verify_crc, deferred):
                    if index % self.world_size == self.rank:
                        yield payload
                    index += 1

    def _examples(self):
        for payload in self._payloads():
            yield self.decode_example(payload)

    def get_dataset(self):
        """Generator of batched feature dicts on `device` (tfrecord_reader.py:61-108)."""
        return _PrefetchedDataset(self) if self.prefetch > 0 else _BatchedDataset(self)

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


class _PrefetchedDataset:
    """The same batches as _BatchedDataset, produced ahead of the consumer by a background thread (module docstring).

    Bytes move ONCE on the host: the shards are memory-mapped, the shuffle buffer holds references (payload views), and a
    decode worker checks the CRC of its record and copies the decoded arrays straight into the record's row of the batch's
    pinned staging buffers; the upload and the uint8 -> float conversion run on a side stream."""

    def __init__(self, reader):
        self.reader = reader
        self.reader_seconds = 0.0             # time the consumer spent waiting for a batch (diagnostics)
        self._maps = {}                       # filename -> (mmap, size, ..., frame index): every shard is mapped ONCE per dataset object
        self._row_tables = {}                 # id(staging buffers) -> per-row (pointer, size) tables
        self._key_table = None

    def _map(self, filename):
        """(mapping, size, base address, payload offsets, lengths, stored CRCs) of a shard: mapped and framed ONCE per dataset
        object (xpt_tfrecord_index: header CRCs checked there) and reused by every later epoch / iteration (a mapping per
        epoch leaked one mmap and one descriptor per shard and epoch: EMFILE after a few dozen epochs)."""
        import mmap
        hit = self._maps.get(filename)
        if hit is None:
            rd = self.reader
            with open(filename, "rb") as f:
                size = f.seek(0, 2)
                mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) if size else None
            if mm is None:
                hit = (None, 0, 0, None, None, None)
            else:
                lib = _lib.load()
                base = np.frombuffer(mm, dtype=np.uint8)            # (keeps the mapping exported: released in close())
                cap = max(16, size // 64)
                while True:
                    off = np.empty(cap, dtype=np.uint64)
                    ln = np.empty(cap, dtype=np.uint64)
                    crc = np.empty(cap, dtype=np.uint32)
                    n = lib.xpt_tfrecord_index(ctypes.c_void_p(base.ctypes.data), size, int(rd.verify_crc),
                                               ctypes.c_void_p(off.ctypes.data), ctypes.c_void_p(ln.ctypes.data),
                                               ctypes.c_void_p(crc.ctypes.data), cap)
                    if n < 0:
                        raise IOError(f"corrupted or truncated TFRecord in {filename} (record {-n - 1})")
                    if n < cap:
                        break
                    cap *= 2
                hit = (mm, size, base, off[:n].tolist(), ln[:n].tolist(), crc[:n].tolist())
            self._maps[filename] = hit
        return hit

    def close(self):
        """Releases the shard mappings (call between iterations only: decode jobs address the mapped bytes)."""
        maps, self._maps = self._maps, {}
        for key in list(maps):
            mm = maps.pop(key)[0]             # (the popped tuple held the numpy view that exported the mapping)
            if mm is not None:
                try:
                    mm.close()
                except BufferError:           # a view is still exported: the mapping dies with its last view
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:                     # noqa: BLE001  (interpreter shutdown)
            pass

    def _refs(self):
        """(payload address, length, stored CRC, file name) of this rank's records in file order; nothing is copied."""
        rd = self.reader
        filenames = sorted(glob.glob(op.join(rd.tfrpath, "*.tfrecord")))
        index = 0
        for _ in range(rd.epochs):
            for filename in filenames:
                mm, size, base, offs, lens, crcs = self._map(filename)
                if size == 0:
                    continue
                addr = base.ctypes.data
                for k in range(len(offs)):
                    if index % rd.world_size == rd.rank:
                        yield addr + offs[k], lens[k], crcs[k], filename
                    index += 1

    def _batched_refs(self):
        rd = self.reader
        buffer, batch = [], []
        for ref in self._refs():
            if rd.shuffle:
                buffer.append(ref)
                if len(buffer) < rd.shuffle_buffer:
                    continue
                ref = buffer.pop(rd.rng.randrange(len(buffer)))
            batch.append(ref)
            if len(batch) == rd.batch_size:
                yield batch
                batch = []
        while buffer:
            batch.append(buffer.pop(rd.rng.randrange(len(buffer))))
            if len(batch) == rd.batch_size:
                yield batch
                batch = []                                          # a trailing partial batch is dropped

    def _keys(self):
        rd = self.reader
        if getattr(self, "_key_table", None) is None:
            names = [k for k, conf in rd.config.items() if isinstance(conf, dict)]
            self._key_names = names
            self._key_table = (ctypes.c_char_p * len(names))(*[k.encode("utf-8") for k in names])
        return self._key_names, self._key_table

    def _staging(self, pinned):
        """Host buffers of one batch: [batch, *shape] per feature of the side-car config (uint8 images stay uint8), and per
        batch row the destination-pointer / size tables xpt_tfrecord_decode fills them through."""
        rd = self.reader
        names, _ = self._keys()
        bufs = {}
        for key in names:
            conf = rd.config[key]
            if conf["parse_type"] == "tf.int64":
                t = torch.empty((rd.batch_size,), dtype=torch.int64)
            else:
                dtype = torch.uint8 if conf["decode_type"] == "tf.uint8" else torch.float32
                t = torch.empty((rd.batch_size,) + tuple(conf["shape"] or ()), dtype=dtype)
            bufs[key] = t.pin_memory() if pinned else t
        rows = []
        for row in range(rd.batch_size):
            ptrs = (ctypes.c_void_p * len(names))(*[bufs[k][row].data_ptr() for k in names])
            sizes = (ctypes.c_size_t * len(names))(*[bufs[k][row].numel() * bufs[k].element_size() for k in names])
            rows.append((ptrs, sizes))
        self._row_tables[id(bufs)] = rows
        return bufs

    def _decode_into(self, ref, bufs, row):
        """CRC check, tf.train.Example walk and the one host copy of the record's bytes, in C (ctypes drops the GIL)."""
        rd = self.reader
        addr, length, pcrc, filename = ref
        names, table = self._keys()
        ptrs, sizes = self._row_tables[id(bufs)][row]
        rc = _lib.load().xpt_tfrecord_decode(ctypes.c_void_p(addr), length, pcrc, int(rd.verify_crc), len(names), table, ptrs,
                                             sizes)
        if rc == 0:
            return
        if rc == -10:
            raise IOError(f"corrupted TFRecord in {filename} (CRC32C mismatch)")
        if -164 <= rc <= -100:
            raise IOError(f"record without feature '{names[-rc - 100]}' in {filename}")
        if -1064 <= rc <= -1000:
            raise IOError(f"feature '{names[-rc - 1000]}' of a record in {filename} does not have the size of its side-car shape")
        raise IOError(f"malformed tf.train.Example in {filename} (code {rc})")

    def _produce(self, out, stop):
        rd = self.reader
        on_gpu = torch.device(rd.device).type == "cuda"
        stream = torch.cuda.Stream(device=rd.device) if on_gpu else None
        nslots = rd.prefetch + 3                                    # a slot is reused only after its batch was consumed
        staging = [None] * nslots
        uploaded = [None] * nslots                                  # event behind the last asynchronous upload out of a slot
        inflight = []                                               # (futures, slot) of batches being decoded

        def finish(futures, slot):
            for fut in futures:
                fut.result()                                        # re-raises a worker's exception
            host = staging[slot]
            if on_gpu:
                with torch.cuda.stream(stream):
                    feats = rd.finish_images({k: v.to(rd.device, non_blocking=True) for k, v in host.items()})
                    ready = torch.cuda.Event()
                    ready.record(stream)
                uploaded[slot] = ready
                item = (feats, ready)
            else:
                item = (rd.finish_images({k: v.clone() for k, v in host.items()}), None)
            while not stop.is_set():
                try:
                    out.put(item, timeout=0.1)
                    return
                except queue.Full:
                    continue

        try:
            with concurrent.futures.ThreadPoolExecutor(max_workers=rd.workers) as pool:
                slot = 0
                for refs in self._batched_refs():
                    if stop.is_set():
                        return
                    if staging[slot] is None:
                        staging[slot] = self._staging(on_gpu)
                    if uploaded[slot] is not None:                  # the DMA out of these pinned rows must be over before
                        uploaded[slot].synchronize()                # a decode worker writes them again
                        uploaded[slot] = None
                    inflight.append(([pool.submit(self._decode_into, ref, staging[slot], i) for i, ref in enumerate(refs)], slot))
                    slot = (slot + 1) % nslots
                    if len(inflight) > 1:                           # the next batch decodes while this one is uploaded
                        finish(*inflight.pop(0))
                for job in inflight:
                    finish(*job)
            out.put(None)
        except BaseException as e:             # noqa: BLE001  (handed to the consumer, which re-raises it)
            out.put(e)

    def __iter__(self):
        import time
        rd = self.reader
        out = queue.Queue(maxsize=rd.prefetch)
        stop = threading.Event()
        thread = threading.Thread(target=self._produce, args=(out, stop), daemon=True, name="xpt-tfrecord-prefetch")
        thread.start()
        try:
            while True:
                t0 = time.perf_counter()
                item = out.get()
                self.reader_seconds += time.perf_counter() - t0
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                feats, ready = item
                if ready is not None:
                    torch.cuda.current_stream().wait_event(ready)
                    for v in feats.values():
                        v.record_stream(torch.cuda.current_stream())
                yield feats
        finally:
            stop.set()
            thread.join(timeout=5.0)
