"""Entry points with the reference's names (model/model_main.py:20-160): train_by_plan, train, predict_by_plan,
predict, create_training_parts, get_dataset, try_load_weights, save_model_weights.

`python -m xpt_mde_2021_amd.model.model_main` trains by opts.TRAINING_PLAN exactly as the reference's
`python model/model_main.py` does; a dataset named "synthetic[_stereo]" streams seeded KITTI-shaped snippets
(no tfrecords needed), anything else is read from opts.DATAPATH_TFR/{dataset}_{split} by the TF-free
TfrecordReader.  Under torchrun (RANK/WORLD_SIZE set) and TRAIN_MODE="distributed" every rank runs this file.
"""
import os
import os.path as op

import numpy as np
import pandas as pd
import torch

from ..config import opts
from ..utils import util_funcs as uf
from ..utils import synthetic_data as sd
from . import train_val as tv
from .build_model.model_factory import ModelFactory
from .loss_and_metric.loss_factory import loss_factory
from .model_util.augmentation import augmentation_factory
from .model_util.distributer import DistributionStrategy
from .model_util.optimizers import optimizer_factory


def train_by_plan():
    """model_main.py:20-27."""
    set_configs()
    print(f"\n===== [CONFIG] device={opts.DEVICE}, ckpt={opts.CKPT_NAME}")
    target_epoch = 0
    for net_names, dataset_name, epoch, learning_rate, loss_weights, scale_weights, save_ckpt in opts.TRAINING_PLAN:
        target_epoch += epoch
        train(net_names, dataset_name, target_epoch, learning_rate, loss_weights, scale_weights, save_ckpt)


def train(net_names, dataset_name, target_epoch, learning_rate, loss_weights, scale_weights, save_ckpt):
    """model_main.py:30-56: resume from history.csv, build parts, epoch loop, save weights."""
    initial_epoch = uf.read_previous_epoch(opts.CKPT_NAME)
    if target_epoch <= initial_epoch:
        print(f"!! target_epoch {target_epoch} <= initial_epoch {initial_epoch}, no need to train")
        return
    dataset_train, tfr_config, train_steps = get_dataset(dataset_name, "train", True)
    dataset_val, _, val_steps = get_dataset(dataset_name, "val", False)
    model, augmenter, loss_object, optimizer = \
        create_training_parts(initial_epoch, tfr_config, learning_rate, loss_weights, scale_weights, net_names)
    trainer, validater = tv.train_val_factory(opts.TRAIN_MODE, model, loss_object, train_steps, opts.STEREO,
                                              augmenter, optimizer)
    print(f"\n\n========== START TRAINING ON {opts.CKPT_NAME} ==========")
    for epoch in range(initial_epoch, target_epoch):
        print(f"========== Start epoch: {epoch}/{target_epoch} ==========")
        result_train = trainer.run_an_epoch(dataset_train)
        validater.steps_per_epoch = val_steps
        result_val = validater.run_an_epoch(dataset_val)
        if is_chief():
            save_log(epoch, dataset_name, result_train, result_val)
            save_model_weights(model, "latest")
    if save_ckpt and is_chief():
        save_model_weights(model, f"ep{target_epoch:02}")


def is_chief():
    strategy = DistributionStrategy.strategy
    return strategy is None or strategy.rank == 0


def set_configs():
    """model_main.py:59-78 (checkpoint directory; device selection instead of TF memory growth)."""
    np.set_printoptions(precision=3, suppress=True)
    os.makedirs(op.join(opts.DATAPATH_CKP, opts.CKPT_NAME), exist_ok=True)
    if opts.TRAIN_MODE == "distributed":
        DistributionStrategy.get_strategy()
    if torch.cuda.is_available():
        torch.backends.cudnn.benchmark = bool(getattr(opts, "MIOPEN_FIND", True))    # fast find (train_val.configure_backend)
        print("Visible GPUs:", torch.cuda.device_count(), torch.cuda.get_device_name(torch.cuda.current_device()))


def device():
    if str(opts.DEVICE).startswith("cuda") and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def create_training_parts(initial_epoch, tfr_config, learning_rate, loss_weights, scale_weights, net_names=None,
                          ckpt_name=None, weight_suffix="latest"):
    """model_main.py:81-96."""
    ckpt_name = opts.CKPT_NAME if ckpt_name is None else ckpt_name
    pretrained_weight = (initial_epoch == 0) and opts.PRETRAINED_WEIGHT
    model = ModelFactory(tfr_config, net_names=net_names, global_batch=opts.BATCH_SIZE,
                         pretrained_weight=pretrained_weight).get_model()
    model = try_load_weights(model, ckpt_name, weight_suffix)
    model.to(device())
    augmenter = augmentation_factory(opts.AUGMENT_PROBS)
    loss_object = loss_factory(tfr_config, loss_weights, scale_weights, opts.STEREO,
                               weights_to_regularize=model.weights_to_regularize(), batch_size=opts.BATCH_SIZE)
    optimizer = optimizer_factory(opts.OPTIMIZER, learning_rate, initial_epoch)
    if "flow_reg" in loss_object.loss_weights:       # gradient of the L2 regulariser: one launch in the optimizer
        optimizer.add_l2(model.weights_to_regularize(), loss_object.loss_weights["flow_reg"])
    return model, augmenter, loss_object, optimizer


def try_load_weights(model, ckpt_name, weight_suffix="latest"):
    """model_main.py:99-106."""
    if opts.CKPT_NAME:
        model_dir_path = op.join(opts.DATAPATH_CKP, ckpt_name, "ckpt")
        if op.isdir(model_dir_path):
            model.load_weights(model_dir_path, weight_suffix)
        else:
            print("===== train from scratch:", model_dir_path)
    return model


class SyntheticDataset:
    """Iterable of feature dicts already resident in HBM (seeded per rank)."""

    def __init__(self, batch_size, steps, height, width, stereo, dev, seed=20211119, pool=4):
        strategy = DistributionStrategy.strategy
        rank = 0 if strategy is None else strategy.rank
        self.steps = steps
        self.batches = []
        for i in range(pool):
            feats = sd.make_features(batch_size, height, width, opts.SNIPPET_LEN, seed + 1000 * rank + i, stereo)
            self.batches.append({k: v.to(dev) for k, v in feats.items()})
        self.config = sd.tfr_config_for({k: v[:1] for k, v in self.batches[0].items()})
        self.config["length"] = steps * batch_size

    def __iter__(self):
        for i in range(self.steps):
            yield self.batches[i % len(self.batches)]


def get_dataset(dataset_name, split, shuffle, batch_size=None):
    """model_main.py:109-118 -> (dataset, tfr_config, steps_per_epoch).  In distributed mode `batch_size` is the
    per-replica batch and every rank reads its own shard."""
    strategy = DistributionStrategy.strategy
    per_replica = opts.PER_REPLICA_BATCH if batch_size is None else batch_size
    if dataset_name.startswith("synthetic"):
        h, w = opts.get_img_shape("HW", "kitti_raw")
        steps = int(os.environ.get("XPT_SYNTHETIC_STEPS", 20 if split == "train" else 4))
        ds = SyntheticDataset(per_replica, steps, h, w, dataset_name.endswith("stereo"), device())
        return ds, ds.config, steps
    from ..tfrecords.tfrecord_reader import TfrecordReader, default_workers
    tfr_path = op.join(opts.DATAPATH_TFR, f"{dataset_name}_{split}")
    print("tfr path : ", tfr_path)
    assert op.isdir(tfr_path), tfr_path
    rank, world = (0, 1) if strategy is None else (strategy.rank, strategy.num_replicas_in_sync)
    # background read / decode / pinned staging / side-stream upload (tf.data's role, tfrecord_reader.py:61-108)
    reader = TfrecordReader(tfr_path, shuffle=shuffle, batch_size=per_replica, rank=rank, world_size=world,
                            device=device(), prefetch=int(getattr(opts, "READER_PREFETCH", 2)),
                            workers=int(getattr(opts, "READER_WORKERS", 0)) or default_workers())
    return reader.get_dataset(), reader.get_tfr_config(), reader.get_total_steps()


def save_model_weights(model, weights_suffix):
    """model_main.py:121-129."""
    model_dir_path = op.join(opts.DATAPATH_CKP, opts.CKPT_NAME, "ckpt")
    os.makedirs(model_dir_path, exist_ok=True)
    model.save_weights(model_dir_path, weights_suffix)


def save_log(epoch, dataset_name, results_train, results_val):
    """Resume contract of the reference's logger (model/model_util/logger.py:24-39, util_funcs.py:129-143):
    history.csv with an `epoch` column, ':' = training and '!' = validation columns."""
    row = {"epoch": epoch, "dataset": dataset_name[:7]}
    for prefix, (frame, hours) in ((":", results_train), ("!", results_val)):
        means = frame.mean(axis=0).to_dict()
        means["time"] = hours
        row.update({prefix + k: v for k, v in means.items()})
    filepath = op.join(opts.DATAPATH_CKP, opts.CKPT_NAME, "history.csv")
    if op.isfile(filepath):
        history = pd.read_csv(filepath, encoding="utf-8")
        history = pd.concat([history[history["epoch"] != epoch], pd.DataFrame([row])], ignore_index=True)
    else:
        history = pd.DataFrame([row])
    history.sort_values(by=["epoch"]).to_csv(filepath, encoding="utf-8", index=False, float_format="%.4f")


def predict_by_plan():
    """model_main.py:132-135."""
    set_configs()
    for net_names, dataset_name, save_keys, ckpt_name, weight_suffix in opts.TEST_PLAN:
        predict(net_names, dataset_name, save_keys, ckpt_name, weight_suffix)


def predict(net_names, dataset_name, save_keys, ckpt_name, weight_suffix):
    """model_main.py:138-160: predictions of the test split -> DATAPATH_PRD/{ckpt}/{dataset}_{suffix}.npz."""
    dataset, tfr_config, steps = get_dataset(dataset_name, "test", False)
    model = ModelFactory(tfr_config, net_names=net_names, global_batch=opts.BATCH_SIZE).get_model()
    model = try_load_weights(model, ckpt_name, weight_suffix)
    model.to(device())
    results = model.predict_dataset(dataset, save_keys, steps)
    out_dir = op.join(opts.DATAPATH_PRD, ckpt_name)
    os.makedirs(out_dir, exist_ok=True)
    np.savez(op.join(out_dir, f"{dataset_name}_{weight_suffix}.npz"), **results)
    print(f"predictions were saved to {out_dir}/{dataset_name}_{weight_suffix}.npz")


if __name__ == "__main__":
    train_by_plan()
