"""Options singleton `opts` -- same attribute names and meaning as the reference's config.py
(a user copy of config-example.py:16-299), so model_main.py / train_val.py read the same.

Differences that are deliberate ("Config / flags"):
  * no hard-coded data directory and no import-time `assert op.isdir(DATAPATH)`
    (config-example.py:184-186): DATAPATH comes from $XPT_DATAPATH (default ./xpt_data) and is
    created lazily by the entry points;
  * defaults follow BASELINE.json's benchmark configuration (NASNetMobile DepthNet, 128x416
    KITTI-raw snippets, batch 8 per GPU, bf16 convolutions) instead of the shipped
    EfficientNetB5 / HIGH_RES / batch 1 -- all of them remain plain attributes a user can change;
  * MI355X-specific switches are grouped at the end (CONV_DTYPE, USE_HIP_GRAPH, FUSED_LOSS, ...).
"""
import os
import os.path as op

import numpy as np

RAW_DATA_PATHS = {name: os.environ.get("XPT_RAW_" + name.upper(), "")
                  for name in ("kitti_raw", "kitti_odom", "cityscapes__sequence", "waymo", "a2d2")}


def _sizes(hw_by_name, mult=1):
    return {k: (h * mult, w * mult) for k, (h, w) in hw_by_name.items()}


class FixedOptions:
    # ---- data options (config-example.py:20-36)
    STEREO = True
    HIGH_RES = False
    SNIPPET_LEN = 5
    MIN_DEPTH = 1e-3
    MAX_DEPTH = 80
    _BASE_SIZES = {"kitti_raw": (128, 416), "kitti_odom": (128, 416), "cityscapes": (192, 512),
                   "waymo": (256, 384), "a2d2": (192, 384)}
    IMAGE_SIZES_SMALL = _sizes(_BASE_SIZES)
    IMAGE_SIZES_LARGE = _sizes(_BASE_SIZES, 2)
    IMAGE_SIZES = IMAGE_SIZES_LARGE if HIGH_RES else IMAGE_SIZES_SMALL

    # ---- training options (:41-46)
    PER_REPLICA_BATCH = 8
    BATCH_SIZE = PER_REPLICA_BATCH
    OPTIMIZER = "adam_constant"
    DEPTH_ACTIVATION = "InverseSigmoid"        # or "Exponential"
    PRETRAINED_WEIGHT = False                  # ImageNet weights are not obtainable offline

    # ---- network options (:51-65)
    JOINT_NET = {"depth": "NASNetMobile", "camera": "PoseNetImproved", "flow": "PWCNet"}
    RIGID_NET = {"depth": JOINT_NET["depth"], "camera": JOINT_NET["camera"]}
    FLOW_NET = {"flow": JOINT_NET["flow"]}
    _CONV_ARGS = {"activation": "leaky_relu", "activation_param": 0.1,
                  "kernel_initializer": "truncated_normal", "kernel_initializer_param": 0.025}
    DEPTH_CONV_ARGS = dict(_CONV_ARGS)
    POSE_CONV_ARGS = dict(_CONV_ARGS)
    FLOW_CONV_ARGS = dict(_CONV_ARGS)
    DEPTH_UPSAMPLE_INTERP = "nearest"

    # ---- loss constants (:67-71)
    IMAGE_GRADIENT_FACTOR = 4
    SMOOTHNESS_FACTOR = 20
    SSIM_RATIO = 0.5
    SCALE_WEIGHT_T1 = np.array([0.25, 0.25, 0.25, 0.25]) * 4.
    SCALE_WEIGHT_T2 = np.array([0.1, 0.2, 0.3, 0.4]) * 4.


def _photo_set(prefix, l1_gain=1., **extra):
    r = FixedOptions.SSIM_RATIO
    d = {prefix + "L1": (1. - r) * l1_gain, prefix + "L1_R": (1. - r) * l1_gain,
         prefix + "SSIM": r, prefix + "SSIM_R": r}
    d.update(extra)
    return d


class LossOptions(FixedOptions):
    """Loss-weight dictionaries, config-example.py:74-127 (same keys, same values)."""
    F = FixedOptions
    _R = F.SSIM_RATIO
    _SM = {"smoothe": F.SMOOTHNESS_FACTOR, "smoothe_R": F.SMOOTHNESS_FACTOR}
    _ST = {"stereoL1": 1. - _R, "stereoSSIM": _R}
    LOSS_RIGID_T1 = _photo_set("", **{"smoothe": 1., "smoothe_R": 1., "stereoL1": 0.01, "stereoSSIM": 0.01,
                                      "stereoPose": 1.})
    LOSS_RIGID_T2 = _photo_set("", **_SM, **_ST, stereoPose=1.)
    LOSS_RIGID_COMB = _photo_set("cmb", 10, **_SM, **_ST, stereoPose=1.)
    LOSS_RIGID_MOA = _photo_set("moa", 10, **_SM, stereoPose=1.)
    LOSS_RIGID_MOA_WST = _photo_set("moa", 10, **_SM, **_ST, stereoPose=1.)
    LOSS_FLOW = {"flowL2": 1., "flowL2_R": 1., "flow_reg": 4e-7}
    LOSS_RIGID_MD2 = _photo_set("md2", **{"smoothe": 1., "smoothe_R": 1.}, **_ST, stereoPose=1.)

    # training plans: rows of (net_names, dataset, epochs, learning_rate, loss_weights, scale_weights, save_ckpt)
    # -- config-example.py:128-174.  The flow-aided fine-tuning rows need FlowNet ("next" in SURVEY 8f).
    LOSS_PRETRAIN_STEP3 = LOSS_RIGID_T2
    LOSS_FINETUNE_STEP3 = LOSS_RIGID_COMB
    FINE_TUNE_NET = FixedOptions.JOINT_NET

    @staticmethod
    def _plan(rows):
        return [(net, ds, ep, lr, loss, FixedOptions.SCALE_WEIGHT_T1, True) for net, ds, ep, lr, loss in rows]

    _RN, _JN = FixedOptions.RIGID_NET, FixedOptions.JOINT_NET
    TRAINING_PLAN_28 = _plan.__func__(
        [(_RN, "kitti_raw", 5, 1e-5, LOSS_RIGID_T1), (_RN, "kitti_raw", 10, 1e-4, LOSS_RIGID_T2),
         (_RN, "a2d2", 10, 1e-4, LOSS_RIGID_T2), (_RN, "waymo", 10, 1e-4, LOSS_RIGID_T2),
         (_RN, "kitti_odom", 10, 1e-4, LOSS_RIGID_T2), (_RN, "cityscapes", 10, 1e-5, LOSS_RIGID_T2),
         (_RN, "kitti_raw", 5, 1e-4, LOSS_RIGID_T2),
         (_JN, "kitti_raw", 10, 1e-4, LOSS_RIGID_COMB), (_JN, "kitti_raw", 10, 1e-5, LOSS_RIGID_COMB),
         (_JN, "kitti_raw", 5, 1e-6, LOSS_RIGID_COMB)])
    TRAINING_PLAN_29 = [row for row in TRAINING_PLAN_28 if row[1] != "waymo"]
    TRAINING_PLAN_30 = _plan.__func__(
        [(_RN, "kitti_raw", 5, 1e-5, LOSS_RIGID_T1), (_RN, "kitti_raw", 10, 1e-4, LOSS_RIGID_T2),
         (_RN, "kitti_raw", 5, 1e-4, LOSS_RIGID_T2),
         (_JN, "kitti_raw", 10, 1e-4, LOSS_RIGID_COMB), (_JN, "kitti_raw", 10, 1e-5, LOSS_RIGID_COMB),
         (_JN, "kitti_raw", 5, 1e-6, LOSS_RIGID_COMB)])
    # KITTI-only rigid plan (the paper's Table 7 backbone ablation; no FlowNet needed)
    TRAINING_PLAN_KITTI_RIGID = TRAINING_PLAN_30[:3]


class VodeOptions(LossOptions):
    L = LossOptions
    # ---- path options (config-example.py:176-193)
    CKPT_NAME = os.environ.get("XPT_CKPT_NAME", "mde01")
    DEVICE = "cuda"
    DATAPATH = os.environ.get("XPT_DATAPATH", op.join(os.getcwd(), "xpt_data"))
    DATAPATH_SRC = op.join(DATAPATH, "srcdata")
    DATAPATH_TFR = op.join(DATAPATH, "tfrecords")
    DATAPATH_CKP = op.join(DATAPATH, "checkpts")
    DATAPATH_LOG = op.join(DATAPATH, "log")
    DATAPATH_PRD = op.join(DATAPATH, "prediction")
    DATAPATH_EVL = op.join(DATAPATH, "evaluation")
    PROJECT_ROOT = op.dirname(op.abspath(__file__))

    # ---- data options (:198-211)
    DATASETS_TO_PREPARE = {"cityscapes__sequence": ["train"], "waymo": ["train"], "a2d2": ["train"],
                           "kitti_raw": ["train", "test"], "kitti_odom": ["train", "test"]}
    FRAME_PER_DRIVE = 0
    TOTAL_FRAME_LIMIT = 0
    VALIDATION_FRAMES = 500
    # PoseNet on a side HIP stream next to DepthNet (forked / joined inside the hipGraph; the mono wrapper only).  It lost in
    # round 2 (20.97 vs 19.60 ms per step: the parallel branch cost more per node than it overlapped); with the step at 4.2 ms
    # it wins: 4.163 -> 4.11 ms at batch 8, 5.70 -> 5.60 at batch 16, the two-graph data-parallel step 4.45 -> 4.27; the
    # stereo wrappers (two calls per step) measured 9.24 -> 9.33 and keep one stream.  XPT_NET_STREAMS=0 / 1 overrides.
    NET_STREAMS = __import__("os").environ.get("XPT_NET_STREAMS", "1") == "1"
    # the graph trainer updates decoder / PoseNet parameters on the side stream while the encoder's backward still runs
    # (train_val.ModelTrainerGraph); XPT_EARLY_UPDATE=0 / 1 overrides
    EARLY_UPDATE = __import__("os").environ.get("XPT_EARLY_UPDATE", "0") == "1"
    AUGMENT_PROBS = {"CropAndResize": 0.2, "HorizontalFlip": 0.2, "ColorJitter": 0.2}

    # ---- training options (:216-253)
    TRAINING_PLAN = L.TRAINING_PLAN_KITTI_RIGID
    _cam = {"camera": "PoseNetImproved", "flow": "PWCNet"}
    RIGID_EF0 = dict(depth="EfficientNetB0", **_cam)
    RIGID_EF3 = dict(depth="EfficientNetB3", **_cam)
    RIGID_EF5 = dict(depth="EfficientNetB5", **_cam)
    RIGID_EF7 = dict(depth="EfficientNetB7", **_cam)
    RIGID_MOBILE = dict(depth="MobileNetV2", **_cam)
    RIGID_NASMOB = dict(depth="NASNetMobile", **_cam)
    TEST_PLAN_LOW = [(RIGID_NASMOB, "kitti_raw", ["depth"], "vode30_nasmob", "latest")]
    TEST_PLAN_HIGH = [(RIGID_NASMOB, "kitti_raw", ["depth"], "vode30_nasmob_2x", "latest")]
    TEST_PLAN = TEST_PLAN_HIGH if FixedOptions.HIGH_RES else TEST_PLAN_LOW

    # ---- other options (:258-266)
    ENABLE_SHAPE_DECOR = False
    LOG_LOSS = True
    READER_PREFETCH = 2      # batches the TFRecord reader keeps ready ahead of the step (0: synchronous generator)
    READER_WORKERS = 0       # decode threads of the prefetching reader (0: from os.sched_getaffinity, tfrecord_reader.default_workers)
    TRAIN_MODE = "graph"                      # "eager" | "graph" (hipGraph replay) | "distributed" (RCCL DP)
    # steps with library (MIOpen) convolutions on the path -- fp32 mode, PWC-Net: "audit" = captured when the node audit of
    # the captured graph finds no memset node (memset nodes replay wrongly on this runtime), True = captured regardless,
    # False = always eager (round 2's rule)
    CAPTURE_LIBRARY_STEPS = "audit"
    RAW_IMAGE_RES = {"kitti_raw": (375, 1242)}

    # ---- MI355X build switches (new; nothing to mirror in the reference)
    CONV_DTYPE = "bf16"                       # dtype of the DepthNet / PoseNet convolutions ("bf16" | "fp16" | "fp32")
    # "fp16" (BASELINE configs[4]: fp16 convolutions + fp32 loss accumulation): IEEE-half activations and packed weights through
    # libxpt_hip_f16.so, fp32 masters / losses / optimizer as always, and a STATIC loss scale -- the seed of the backward pass
    # is LOSS_SCALE_FP16 instead of 1 (per-pixel gradients of a mean over 4e5 pixels are ~1e-7, below half's 6e-8 .. 6e-5
    # subnormal range), taken out again by the optimizer's grad_scale (a power of two: exact)
    LOSS_SCALE_FP16 = float(__import__("os").environ.get("XPT_LOSS_SCALE_FP16", "32768"))
    CHANNELS_LAST = True                      # NHWC activations for MIOpen
    FUSED_LOSS = True                         # fused warp+L1+SSIM march kernels when the loss set allows it
    GRAD_BUCKETS = 1                          # flat gradient buckets per all-reduce (RCCL over xGMI)

    @classmethod
    def get_raw_data_path(cls, dataset_name):
        path = RAW_DATA_PATHS.get(dataset_name)
        assert path is not None, f"Invalid dataset name, available datasets are {list(RAW_DATA_PATHS.keys())}"
        assert op.exists(path), f"{path}"
        return path

    @classmethod
    def get_img_shape(cls, code="HW", dataset="kitti_raw", scale_div=1):
        """config-example.py:272-294."""
        h, w = cls.IMAGE_SIZES[dataset]
        hs, ws = h // scale_div, w // scale_div
        table = {"H": hs, "W": ws, "HW": (h, w), "WH": (ws, hs), "HWC": (hs, ws, 3),
                 "SHW": (cls.SNIPPET_LEN, hs, ws), "SHWC": (cls.SNIPPET_LEN, hs, ws, 3),
                 "BSHWC": (cls.BATCH_SIZE, cls.SNIPPET_LEN, hs, ws, 3),
                 "RSHWC": (cls.PER_REPLICA_BATCH, cls.SNIPPET_LEN, hs, ws, 3)}
        assert code in table, f"Invalid code: {code}"
        return table[code]


opts = VodeOptions()
