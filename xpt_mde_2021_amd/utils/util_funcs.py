"""Hot-path helpers of the reference's utils/util_funcs.py, re-created for torch tensors on MI355X."""
import json
import os.path as op
import sys

import torch
import torch.nn.functional as F

from ..config import opts
from ..hip import ops as _ops


def print_progress_status(status_msg):
    """util_funcs.py:13-18."""
    sys.stdout.write("\r" + status_msg)
    sys.stdout.flush()


def to_float_image(im_tensor):
    """util_funcs.py:79-80: uint8 [0,255] -> float [-1,1]."""
    return im_tensor.to(torch.float32) * (2.0 / 255.0) - 1.0


def to_uint8_image(im_tensor):
    """util_funcs.py:83-85."""
    im = (im_tensor.clamp(-1, 1) + 1.) / 2.
    return (im * 255.0 + 0.5).floor().clamp(0, 255).to(torch.uint8)


def safe_reciprocal_number(src_tensor):
    """util_funcs.py:157-160: (1/x) * [x > 1e-5]  (elementwise on the network output; the smoothness
    kernel can also fuse it, see hip.ops.smoothness(input_is_depth=True))."""
    mask = (src_tensor > 0.00001).to(src_tensor.dtype)
    return (1. / src_tensor) * mask


def safe_reciprocal_number_ms(src_ms):
    """util_funcs.py:146-154."""
    return [safe_reciprocal_number(src) for src in src_ms]


def resize_bilinear_tf(image_nhwc, size):
    """tf.image.resize(bilinear) of TF2 (half-pixel centres, no antialias) for arbitrary sizes, differentiable
    (used where a gradient is needed or the factor is not an exact integer down-scale)."""
    x = image_nhwc.permute(0, 3, 1, 2)
    x = F.interpolate(x, size=tuple(int(s) for s in size), mode="bilinear", align_corners=False, antialias=False)
    return x.permute(0, 2, 3, 1)


def multi_scale_like_depth(image, depth_ms):
    """util_funcs.py:163-175: image [B,H,W,3] resized (TF2 bilinear) to every depth scale.  Exact integer
    down-scales of an input image run the gfx950 pyramid kernel (K1)."""
    H, W = image.shape[1:3]
    out = []
    for depth in depth_ms:
        hs, ws = depth.shape[1:3]
        if H % hs == 0 and W % ws == 0 and H // hs == W // ws and (H // hs == 1 or (H // hs) % 2 == 0):
            out.append(_ops.resize_down(image, H // hs))
        else:
            out.append(resize_bilinear_tf(image, (hs, ws)))
    return out


def resize_like_size(image, hs, ws):
    """TF2 bilinear resize of input images [M,H,W,3] (no gradient) to (hs, ws): the gfx950 pyramid kernel for exact even
    integer down-scales of device images, the generic resize otherwise."""
    H, W = image.shape[1:3]
    if image.is_cuda and H % hs == 0 and W % ws == 0 and H // hs == W // ws and (H // hs == 1 or (H // hs) % 2 == 0):
        return _ops.resize_down(image, H // hs)
    return resize_bilinear_tf(image, (hs, ws))


def multi_scale_like_flow(image, flow_ms):
    """util_funcs.py:178-190: image [B,H,W,3] resized to the resolution of every flow [B,N,h,w,2]."""
    return [resize_like_size(image, flow.shape[2], flow.shape[3]) for flow in flow_ms]


def read_tfrecords_info(dataset_dir):
    """util_funcs.py:112-116."""
    with open(op.join(opts.DATAPATH_TFR, dataset_dir, "tfr_config.txt"), "r") as fr:
        return json.load(fr)


def read_previous_epoch(model_name):
    """util_funcs.py:129-143: resume point = last 'epoch' in history.csv + 1."""
    filename = op.join(opts.DATAPATH_CKP, model_name, "history.csv")
    if not op.isfile(filename):
        print("[read_previous_epoch] NO history")
        return 0
    import pandas as pd
    history = pd.read_csv(filename, encoding="utf-8", converters={"epoch": lambda c: int(c)})
    if history.empty:
        return 0
    prev_epoch = sorted(history["epoch"].tolist())[-1]
    print(f"[read_previous_epoch] start from epoch {prev_epoch + 1}")
    return prev_epoch + 1
