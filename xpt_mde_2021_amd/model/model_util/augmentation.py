"""Intrinsics-consistent augmentation with the reference's names (model/model_util/augmentation.py:5-219):
augmentation_factory, TotalAugment, CropAndResize, HorizontalFlip, ColorJitter -- on GPU tensors.

Everything is branch-free device arithmetic (random draws come from the device generator, the "apply with probability
p" decisions are torch.where selections), so the augmenter can sit inside the captured hipGraph step like the
reference's augmenter sits inside its @tf.function (train_val.py:79).
"""
import torch
import torch.nn.functional as F

from ...utils.util_class import WrongInputException


def augmentation_factory(augment_probs=None):
    """augmentation.py:5-19."""
    table = {"CropAndResize": CropAndResize, "HorizontalFlip": HorizontalFlip, "ColorJitter": ColorJitter}
    augmenters = []
    for key, prob in (augment_probs or {}).items():
        if key not in table:
            raise WrongInputException(f"Wrong augmentation type: {key}")
        augmenters.append(table[key](prob))
    return TotalAugment(augmenters)


_PIN_BUFFERS = {}      # device -> float[9] (flag + eight pinned uniforms), see TotalAugment._pin_buffer


class TotalAugment:
    def __init__(self, augment_objects=None):
        self.augment_objects = augment_objects or []
        self._pin = None             # device float[9]: flag + eight pinned uniforms (fused path; see pin_draws)

    # ---- pinned draws (the replay check of a captured step: replays and an eager run must see the SAME random draws)
    def can_pin(self):
        return ([type(a) for a in self.augment_objects] == [CropAndResize, HorizontalFlip, ColorJitter]
                and all(a.aug_prob > 0 for a in self.augment_objects))

    def _pin_buffer(self, device):
        # ONE buffer per device for the life of the process: captured graphs keep its address (xpt_augment_pin hands it to
        # every later launch), so it must never be freed or replaced
        from ...hip import lib as _lib
        buf = _PIN_BUFFERS.get(device)
        if buf is None:
            if torch.cuda.is_current_stream_capturing():
                raise WrongInputException("augmentation: run one eager step before capturing (the pin buffer is allocated on first use)")
            buf = _PIN_BUFFERS[device] = torch.zeros(9, device=device)
        _lib.check(_lib.load().xpt_augment_pin(buf.data_ptr()), "xpt_augment_pin")
        self._pin = buf
        return buf

    def pin_draws(self):
        """From now on the fused augmentation kernel repeats one set of draws (also inside already captured graphs).
        The parameter tensors the augmenters expose (`param`, `params`: outputs of the LAST call -- of the captured step,
        when this is called right after a capture) are remembered: eager calls made while pinned rebind them."""
        if self._pin is not None:
            self._pin[1:].uniform_()
            self._pin[0] = 1.0
            self._bound = (getattr(self, "params", None), [getattr(a, "param", None) for a in self.augment_objects])

    def unpin_draws(self):
        if self._pin is not None:
            self._pin[0] = 0.0
            bound = getattr(self, "_bound", None)
            if bound is not None:
                self.params = bound[0]
                for a, q in zip(self.augment_objects, bound[1]):
                    a.param = q
                self._bound = None

    def __call__(self, features):
        if self._fusable(features):
            return self._fused(features)
        feat_aug = dict(features)                      # never mutate the caller's dict (augmentation.py:33-38)
        for augmenter in self.augment_objects:
            feat_aug = augmenter(feat_aug)
        for sfx in ("", "_R"):
            if "image5d" + sfx in feat_aug:
                feat_aug["image5d" + sfx] = feat_aug["image5d" + sfx].contiguous()
        return feat_aug


    # ---- gfx950 fast path: the default chain [CropAndResize, HorizontalFlip, ColorJitter] on device tensors in ONE launch
    #      (csrc/xpt_augment.hip) instead of ~60 (0.49 ms of an 8.5 ms step)
    def _fusable(self, features):
        img = features.get("image5d")
        objs = self.augment_objects
        return (torch.is_tensor(img) and img.is_cuda and img.dtype == torch.float32 and img.dim() == 5 and img.shape[-1] == 3
                and "intrinsic" in features and [type(a) for a in objs] == [CropAndResize, HorizontalFlip, ColorJitter]
                and all(a.aug_prob > 0 for a in objs))

    def _fused(self, features, u=None):
        from ...hip import lib as _lib
        from ...hip import ops as _ops
        crop, flip, jit = self.augment_objects
        lib = _lib.load()
        img = features["image5d"].contiguous()
        b, s, h, w, _ = img.shape
        dev = img.device
        self._pin_buffer(dev)
        if u is None:
            u = torch.rand(8, device=dev)             # device generator: new draws on every replay of a captured step
        params = torch.empty(8, device=dev)
        out = dict(features)

        def pair(key, shape_tail):
            x = features.get(key)
            if x is None:
                return None, None
            x = x.contiguous().float()
            if tuple(x.shape[-len(shape_tail):]) != shape_tail:
                raise WrongInputException(f"augmentation: {key} has shape {tuple(x.shape)}")
            y = torch.empty_like(x)
            out[key] = y
            return x, y

        img0_out = torch.empty_like(img)
        out["image5d"] = img0_out
        img1, img1_out = pair("image5d_R", (h, w, 3))
        depth, depth_out = pair("depth_gt", (h, w, 1))
        k0, k0_out = pair("intrinsic", (3, 3))
        k1, k1_out = pair("intrinsic_R", (3, 3))
        p0, p0_out = pair("pose_gt", (4, 4))
        p1, p1_out = pair("pose_gt_R", (4, 4))
        st, st_out = pair("stereo_T_LR", (4, 4))
        if img1 is not None and img1.shape != img.shape:
            raise WrongInputException("augmentation: image5d_R must have the shape of image5d")
        n_pose = 0 if p0 is None else p0.numel() // 16
        if p1 is not None and p1.numel() // 16 != n_pose:
            raise WrongInputException("augmentation: pose_gt_R must have the shape of pose_gt")
        ptr = _ops._ptr
        _lib.check(lib.xpt_augment(ptr(u), ptr(params), ptr(img), ptr(img0_out), ptr(img1), ptr(img1_out), b * s, ptr(depth),
                                   ptr(depth_out), 0 if depth is None else depth.shape[0], ptr(k0), ptr(k0_out), ptr(k1),
                                   ptr(k1_out), b, ptr(p0), ptr(p0_out), ptr(p1), ptr(p1_out), n_pose, ptr(st), ptr(st_out),
                                   h, w, float(crop.aug_prob), float(flip.aug_prob), float(jit.aug_prob),
                                   float(crop.half_crop_ratio), _ops._stream()), "xpt_augment")
        crop.param = params[0:4]
        jit.param = params[5:8]                   # (applied 0/1, gamma, saturation) -- views, no extra launch
        self.params = params
        return out


class AugmentBase:
    def __init__(self, aug_prob=0.):
        self.aug_prob = aug_prob
        self.param = 0

    def __call__(self, features):
        raise NotImplementedError()


def _flat(image5d):
    b, s, h, w, c = image5d.shape
    return image5d.reshape(b * s, h, w, c)


def crop_and_resize(image_nhwc, box, crop_hw, method="bilinear"):
    """tf.image.crop_and_resize with ONE normalised box (y1, x1, y2, x2) for the whole batch: output pixel (i, j)
    samples y = y1 (H-1) + i (y2-y1)(H-1)/(ch-1) (corner aligned), zeros outside the image."""
    n, h, w, c = image_nhwc.shape
    ch, cw = crop_hw
    dev, dt = image_nhwc.device, image_nhwc.dtype
    ys = box[0] + (box[2] - box[0]) * torch.linspace(0, 1, ch, device=dev, dtype=dt)     # in [0,1] image coordinates
    xs = box[1] + (box[3] - box[1]) * torch.linspace(0, 1, cw, device=dev, dtype=dt)
    gy, gx = torch.meshgrid(ys * 2 - 1, xs * 2 - 1, indexing="ij")
    grid = torch.stack([gx, gy], dim=-1).unsqueeze(0).expand(n, ch, cw, 2)
    out = F.grid_sample(image_nhwc.permute(0, 3, 1, 2), grid, mode=method, padding_mode="zeros", align_corners=True)
    return out.permute(0, 2, 3, 1)


class CropAndResize(AugmentBase):
    """augmentation.py:66-129: one random box for the whole batch, images resized back, intrinsics adjusted."""

    def __init__(self, aug_prob=0.3):
        super().__init__(aug_prob)
        self.half_crop_ratio = 0.1

    def __call__(self, features):
        image5d = features["image5d"]
        b, s, height, width, _ = image5d.shape
        boxes = self.random_crop_boxes(b * s, image5d.device)
        box = self.param = boxes[0]
        for sfx in ("", "_R"):
            if "image5d" + sfx in features:
                img = crop_and_resize(_flat(features["image5d" + sfx]), box, (height, width))
                features["image5d" + sfx] = img.reshape(b, s, height, width, -1)
                features["intrinsic" + sfx] = self.adjust_intrinsic(features["intrinsic" + sfx], boxes, (height, width))
        if "depth_gt" in features:
            features["depth_gt"] = crop_and_resize(features["depth_gt"], box, (height, width), method="nearest")
        return features

    def random_crop_boxes(self, num_box, device="cpu"):
        """augmentation.py:94-109 -> [num_box, 4] copies of ONE random (y1, x1, y2, x2): each side is cropped with
        probability aug_prob by up to 10 %."""
        maxval1 = self.half_crop_ratio
        minval1 = -(1. - self.aug_prob) * self.half_crop_ratio / self.aug_prob
        y1x1 = (torch.rand(2, device=device) * (maxval1 - minval1) + minval1).clamp(0, 1)
        minval2, maxval2 = 1. - maxval1, 1. - minval1
        y2x2 = (torch.rand(2, device=device) * (maxval2 - minval2) + minval2).clamp(0, 1)
        assert (minval1 < maxval1) and (minval2 < maxval2)
        return torch.cat([y1x1, y2x2]).unsqueeze(0).repeat(num_box, 1)

    def adjust_intrinsic(self, intrinsic, boxes, imsize):
        """augmentation.py:111-129: intrinsic [batch,3,3], boxes [n,4] (row 0 is used), imsize (height, width):
        cx' = (cx - x1 W) / (x2 - x1), fx' = fx / (x2 - x1), same for y."""
        height, width = float(imsize[0]), float(imsize[1])
        box = boxes[0].to(intrinsic.device, intrinsic.dtype)
        shift = torch.zeros_like(intrinsic)
        shift[:, 0, 2] = box[1] * width
        shift[:, 1, 2] = box[0] * height
        crop = intrinsic - shift
        x_ratio = 1. / (box[3] - box[1])
        y_ratio = 1. / (box[2] - box[0])
        return torch.stack([crop[:, 0] * x_ratio, crop[:, 1] * y_ratio, crop[:, 2]], dim=1)


class HorizontalFlip(AugmentBase):
    """augmentation.py:132-186: with probability aug_prob flip images, intrinsics, gt poses and the stereo extrinsic."""

    def __init__(self, aug_prob=0.2):
        super().__init__(aug_prob)

    def __call__(self, features):
        apply = torch.rand((), device=features["image5d"].device) < self.aug_prob
        flipped = self.flip_features(features)
        return {k: (torch.where(apply, flipped[k], v) if k in flipped and torch.is_tensor(v) else v)
                for k, v in features.items()}

    def flip_features(self, features):
        out = {}
        b, s, h, w, _ = features["image5d"].shape
        imshape = (b * s, h, w, 3)
        for sfx in ("", "_R"):
            if "image5d" + sfx in features:
                out["image5d" + sfx] = torch.flip(features["image5d" + sfx], dims=[3])
            if "intrinsic" + sfx in features:
                out["intrinsic" + sfx] = self.flip_intrinsic(features["intrinsic" + sfx], imshape)
            if "pose_gt" + sfx in features:
                out["pose_gt" + sfx] = self.flip_gt_pose(features["pose_gt" + sfx])
        if "stereo_T_LR" in features:
            out["stereo_T_LR"] = self.flip_stereo_pose(features["stereo_T_LR"])
        return out

    def flip_intrinsic(self, intrinsic, imshape):
        """augmentation.py:169-173: |[[0,0,W],[0,0,0],[0,0,0]] - K|  (cx -> W - cx); imshape (n, height, width, 3)."""
        width = imshape[2]
        wh = torch.zeros_like(intrinsic)
        wh[:, 0, 2] = float(width)
        return torch.abs(wh - intrinsic)

    def flip_gt_pose(self, pose):
        """augmentation.py:175-186: T_flip @ pose @ inv(T_flip), T_flip = diag(-1, 1, 1, 1)."""
        # T_flip P T_flip negates row 0 and column 0 except their crossing; built with fills only (no host->device
        # copy) so that it can be captured into the step's hipGraph
        sign = torch.ones(4, 4, device=pose.device, dtype=pose.dtype)
        sign[0, 1:] = -1.
        sign[1:, 0] = -1.
        return pose * sign

    def flip_stereo_pose(self, pose):
        """augmentation.py:181-185, [batch, 4, 4]."""
        return self.flip_gt_pose(pose)


class ColorJitter(AugmentBase):
    """augmentation.py:189-219: with probability aug_prob, saturation x U(0.5,1.5) then gamma U(0.5,1.5) on [0,1]."""

    def __init__(self, aug_prob=0.2):
        super().__init__(aug_prob)

    def __call__(self, features):
        dev = features["image5d"].device
        apply = torch.rand((), device=dev) < self.aug_prob
        gamma = torch.rand((), device=dev) + 0.5
        saturation = torch.rand((), device=dev) + 0.5
        self.param = torch.stack([gamma, saturation]) * apply
        for sfx in ("", "_R"):
            if "image5d" + sfx in features:
                img = features["image5d" + sfx]
                features["image5d" + sfx] = torch.where(apply, self.jitter_color(img, gamma, saturation), img)
        return features

    def jitter_color(self, image, gamma, saturation):
        image = (image + 1.) / 2.
        image = adjust_saturation(image, saturation)
        image = image.clamp_min(0) ** gamma                      # tf.image.adjust_gamma(gain=1)
        return image * 2. - 1.


def adjust_saturation(rgb, factor):
    """tf.image.adjust_saturation: RGB -> HSV, S *= factor (clipped to [0,1]), HSV -> RGB.  With hue and value fixed
    this is a per-pixel linear blend towards the pixel's maximum: c' = v - (v - c) * s'/s."""
    v = rgb.max(dim=-1, keepdim=True).values
    mn = rgb.min(dim=-1, keepdim=True).values
    delta = v - mn
    s = torch.where(v > 0, delta / v.clamp_min(1e-12), torch.zeros_like(v))
    s_new = (s * factor).clamp(0, 1)
    ratio = torch.where(s > 0, s_new / s.clamp_min(1e-12), torch.zeros_like(s))
    return v - (v - rgb) * ratio
