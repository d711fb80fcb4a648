"""TotalLoss and the rigid (depth + pose) loss objects with the reference's names and call signatures
(model/loss_and_metric/losses.py).  All per-pixel arithmetic runs in gfx950 kernels; what is left here is the
orchestration of losses.py:26-140 (which tensors are synthesized, how scales / types are merged).

In scope: L1, SSIM, smoothe (+ _R), stereoL1, stereoSSIM, stereoPose, and the "next" row
md2*/moa* min-over-sources variants and the flow-aided losses (cmb*, flowL2, flow_reg; losses.py:235-279, 497-533).
"""
import numpy as np
import torch

from ...config import opts
from ...hip import ops as _ops
from ...utils import convert_pose as cp
from ...utils import util_funcs as uf
from ...utils.util_class import WrongInputException
from ..synthesize.flow_warping import FlowWarpMultiScale
from ..synthesize.synthesize_base import SynthesizeMultiScale, _FUSED_MULTI_SCALE
from . import loss_util as lsu


class TotalLoss:
    def __init__(self, loss_objects=None, loss_weights=None, stereo=False, batch_size=1):
        """
        :param loss_objects: dict of loss objects
        :param loss_weights: dict of weights of losses
        :param batch_size: GLOBAL batch size (losses.py:49: per-example losses are summed and divided by it,
                           so that data-parallel gradients are SUMMED across replicas)
        """
        self.loss_objects = loss_objects
        self.loss_weights = loss_weights
        self.stereo = stereo
        self.batch_size = batch_size
        self.fused = None              # None = decide per call from opts.FUSED_LOSS; True / False forces it

    def use_fused(self, tensor):
        """The fused warp+L1+SSIM march kernel replaces `synthesize -> photometric_loss_l1/_ssim` when every consumer
        of the synthesized views is one of those two reduce=True losses (L1 / SSIM / stereoL1 / stereoSSIM); the
        min-over-sources variants (md2*, moa*) need the images themselves and keep the unfused path."""
        want = bool(opts.FUSED_LOSS) if self.fused is None else bool(self.fused)
        plain = all(isinstance(o, (PhotometricLossMultiScale, StereoDepthLoss)) and o.method in ("L1", "SSIM")
                    for o in (self.loss_objects or {}).values() if isinstance(o, PhotometricLoss))
        return want and plain and tensor.is_cuda

    def _photo_grad_hint(self, cls, suffix, nscales):
        """(d total / d L1_s, d total / d SSIM_s) per scale for the fused photometric terms a loss object of class `cls`
        with this key suffix reads: weight of the loss type x scale weight / global batch (the coefficients __call__
        applies below), or None when they cannot be told in advance."""
        out = {"L1": [0.0] * nscales, "SSIM": [0.0] * nscales}
        # (the fp16 configuration seeds the backward pass with its static loss scale, train_val.ModelTrainer.loss_seed)
        seed = float(opts.LOSS_SCALE_FP16) if opts.CONV_DTYPE == "fp16" else 1.0
        try:
            for name, obj in (self.loss_objects or {}).items():
                if type(obj) is not cls or getattr(obj, "key_suffix", "") != suffix or obj.method not in out:
                    continue
                weights = [float(w) for w in np.asarray(obj.scale_weights, dtype=np.float64).reshape(-1)]
                if len(weights) < nscales:
                    return None
                for i in range(nscales):
                    out[obj.method][i] += seed * float(self.loss_weights[name]) * weights[i] / self.batch_size
        except (KeyError, TypeError, ValueError):
            return None
        return out["L1"], out["SSIM"]

    def __call__(self, predictions, features):
        """
        :param predictions: {"depth_ms": .., "disp_ms": .., "pose": ..}
        :param features: {"image5d": .., "intrinsic": .., ...}
        :return: (total_loss scalar, {loss name: unweighted per-type mean})     losses.py:26-55
        """
        augm_data = self.append_data(features, predictions)
        if self.stereo and ("image5d_R" in features):
            augm_data.update(self.append_data(features, predictions, "_R"))
            augm_data.update(self.synethesize_stereo(features, predictions, augm_data))

        # Every loss object hands back its raw per-scale [batch] terms with their host-side weights instead of
        # combining them with a dozen tiny tensor ops each; all terms of all loss types are then reduced together:
        #   rs = rowsum(stack(terms));  total = <c, rs>;  loss_by_type = A rs      (c, A: constant device tensors)
        # = 1 launch forward and 1 backward for the whole merge (xpt_merge_total_*) (losses.py:46-55 and :147-154 in one step).
        LossBase._collector = collected = []
        try:
            outputs = {name: obj(features, predictions, augm_data) for name, obj in self.loss_objects.items()}
        finally:
            LossBase._collector = None
        terms, coef, rows = [], [], []
        for loss_name, out in outputs.items():
            pairs = collected[out.index] if isinstance(out, _DeferredTerms) else [(out, 1.0)]
            w_type = float(self.loss_weights[loss_name])
            row = {}
            for tensor, w in pairs:
                row[len(terms)] = w / self.batch_size
                coef.append(w * w_type / self.batch_size)           # tf.nn.compute_average_loss, then the type weight
                terms.append(tensor.reshape(-1))
            rows.append(row)
        c_vec, a_mat = self._merge_constants(coef, rows, terms[0].device)
        if (terms[0].is_cuda and len(terms) <= 64 and len(rows) <= 64
                and all(t.dtype == torch.float32 and t.numel() == terms[0].numel() for t in terms)):
            total_loss, by_type = _ops.merge_total(c_vec, a_mat, terms)          # one launch (and one backward)
        else:
            total_loss, row_sums = _WeightedTotal.apply(c_vec, *terms)
            by_type = torch.mv(a_mat, row_sums)
        loss_by_type = {name: by_type[i] for i, name in enumerate(outputs)}
        return total_loss, loss_by_type

    def _merge_constants(self, coef, rows, device):
        """Constant operands of the merge, built on the first (eager, warm-up) call and cached: a host->device copy is
        not capturable into the hipGraph step."""
        key = (tuple(coef), tuple(tuple(sorted(r.items())) for r in rows), str(device))
        cached = getattr(self, "_merge_cache", None)
        if cached is None or cached[0] != key:
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise WrongInputException("loss configuration changed during graph capture")
            a = torch.zeros((len(rows), len(coef)), dtype=torch.float32)
            for i, r in enumerate(rows):
                for j, w in r.items():
                    a[i, j] = w
            self._merge_cache = (key, torch.tensor(coef, dtype=torch.float32).to(device), a.to(device))
        return self._merge_cache[1], self._merge_cache[2]

    def append_data(self, features, predictions, suffix=""):
        """losses.py:57-104: source/target split (TARGET FRAME LAST), multi-scale target, synthesized views."""
        image5d = features["image5d" + suffix]
        intrinsic = features["intrinsic" + suffix]
        rigid = ("depth_ms" + suffix in predictions) and ("pose" + suffix in predictions)
        scales = self._pyramid_scales(image5d, predictions["depth_ms" + suffix]) if rigid else None
        sources_ms = None
        if scales:
            # the dense copies of the source / target frames and their pyramids in one launch (xpt_image_pyramids)
            ready = predictions.get("image_pyramids" + suffix)      # (model_wrappers: issued on PoseNet's side stream)
            if ready is not None and set(scales) <= set(ready[0]):
                sources, targets = ready[1], ready[2]
            else:
                sources, targets = _ops.image_pyramids(image5d, scales)
            source_image, target_image = sources[1], targets[1]
            sources_ms, target_ms = [sources[s] for s in scales], [targets[s] for s in scales]
        else:
            # one dense copy each, made here once: every scale would otherwise re-copy the strided slices (4 x 25 MB)
            source_image = image5d[:, :-1].contiguous()
            target_image = image5d[:, -1].contiguous()
        augm_data = {"source" + suffix: source_image, "target" + suffix: target_image}
        if rigid:
            pred_depth_ms = predictions["depth_ms" + suffix]
            pred_pose = predictions["pose" + suffix]
            if not scales:
                target_ms = uf.multi_scale_like_depth(target_image, pred_depth_ms)
            augm_data["target_ms" + suffix] = target_ms
            if self.use_fused(image5d):
                augm_data["fused_photo_ms" + suffix] = SynthesizeMultiScale().photometric_losses(
                    source_image, intrinsic, pred_depth_ms, pred_pose, target_ms, sources_ms,
                    grad_hint=self._photo_grad_hint(PhotometricLossMultiScale, suffix, len(pred_depth_ms)))
            else:
                augm_data["synth_target_ms" + suffix] = SynthesizeMultiScale()(source_image, intrinsic,
                                                                                pred_depth_ms, pred_pose)
        if "flow_ms" + suffix in predictions:                     # losses.py:94-101
            pred_flow_ms = predictions["flow_ms" + suffix]
            # flows have a lower resolution than the depths: the targets are resized like the flows
            augm_data["flow_target_ms" + suffix] = uf.multi_scale_like_flow(target_image, pred_flow_ms)
            augm_data["warped_target_ms" + suffix] = FlowWarpMultiScale()(source_image, pred_flow_ms)
        return augm_data

    @staticmethod
    def _pyramid_scales(image5d, depth_ms):
        """Integer factors of the depth scales when every one is an exact even (or unit) down-scale of a dense float32
        device snippet batch (what xpt_image_pyramids serves), else None."""
        if not (_FUSED_MULTI_SCALE and image5d.is_cuda and image5d.dtype == torch.float32 and image5d.is_contiguous()
                and image5d.shape[1] >= 2 and image5d.shape[-1] == 3 and len(depth_ms) <= 4):
            return None
        H, W = image5d.shape[2:4]
        scales = []
        for depth in depth_ms:
            hs, ws = depth.shape[1:3]
            if H % hs or W % ws or H // hs != W // ws or (H // hs != 1 and (H // hs) % 2):
                return None
            scales.append(H // hs)
        return scales if len(set(scales)) == len(scales) else None

    def synethesize_stereo(self, features, predictions, augm_data):
        """losses.py:106-140 (the reference's spelling is kept: logger.py:211-216 calls it by this name)."""
        synth_stereo = dict()
        if ("stereo_T_LR" not in features) or ("depth_ms" not in predictions):
            return synth_stereo
        # left image from the right image: points move left -> right with inv(T_LR)
        pose_T_RL = cp.pose_matr2rvec_batch(cp.rigid_inverse(features["stereo_T_LR"]).unsqueeze(1))
        pose_T_LR = cp.pose_matr2rvec_batch(features["stereo_T_LR"].unsqueeze(1))
        left_src, right_src = augm_data["target_R"].unsqueeze(1), augm_data["target"].unsqueeze(1)
        if self.use_fused(left_src):
            hint = self._photo_grad_hint(StereoDepthLoss, "", len(predictions["depth_ms"]))   # left and right: same weights
            synth_stereo["fused_stereo_ms"] = SynthesizeMultiScale().photometric_losses(
                left_src, features["intrinsic"], predictions["depth_ms"], pose_T_RL, augm_data["target_ms"], grad_hint=hint)
            synth_stereo["fused_stereo_ms_R"] = SynthesizeMultiScale().photometric_losses(
                right_src, features["intrinsic"], predictions["depth_ms_R"], pose_T_LR, augm_data["target_ms_R"],
                grad_hint=hint)
            return synth_stereo
        synth_stereo["stereo_synth_ms"] = SynthesizeMultiScale()(left_src, features["intrinsic"],
                                                                 predictions["depth_ms"], pose_T_RL)
        synth_stereo["stereo_synth_ms_R"] = SynthesizeMultiScale()(right_src, features["intrinsic"],
                                                                   predictions["depth_ms_R"], pose_T_LR)
        return synth_stereo


class _DeferredTerms:
    """Placeholder a loss object returns while TotalLoss collects raw per-scale terms."""

    def __init__(self, index):
        self.index = index


class _WeightedTotal(torch.autograd.Function):
    """total = c . rowsum(stack(terms)) with a backward that hands every term ONE row of a dense [terms, batch] matrix
    (one multiply + one copy): autograd's own chain (dot, sum, stack) gives each term an expanded stride-0 gradient that
    the loss kernels then copy one by one (12 launches per step)."""

    @staticmethod
    def forward(ctx, c_vec, *terms):
        stacked = torch.stack(terms)                                  # [terms, batch]
        row_sums = stacked.sum(dim=1)
        ctx.save_for_backward(c_vec)
        ctx.shape = tuple(stacked.shape)
        ctx.mark_non_differentiable(row_sums)
        ctx.set_materialize_grads(False)         # no zero-filled gradient for the non-differentiable second output
        return torch.dot(c_vec, row_sums), row_sums

    @staticmethod
    def backward(ctx, g_total, _):
        c_vec, = ctx.saved_tensors
        n, batch = ctx.shape
        grads = (c_vec * g_total).unsqueeze(1).expand(n, batch).contiguous()
        return (None, *grads.unbind(0))


class LossBase:
    scale_weights = None
    _collector = None            # list filled by merge_multi_scale_losses while TotalLoss.__call__ runs

    def __call__(self, features, predictions, augm_data):
        raise NotImplementedError()

    def merge_multi_scale_losses(self, losses, name="", factors=None):
        """losses.py:147-154: [scales, batch]^T x scale_weights[scales, 1] -> [batch, 1].
        `losses` may hold several groups of per-scale terms (stereo: left + right); `factors` multiplies the scale
        weights (smoothness: 1 / scale)."""
        # the weights are host constants: fold them in as scalars (no H2D copy -> hipGraph-capturable)
        weights = [float(w) for w in np.asarray(self.scale_weights, dtype=np.float64).reshape(-1)]
        groups = losses if isinstance(losses[0], (list, tuple)) else [losses]
        pairs = []
        for group in groups:
            for i, (w, loss) in enumerate(zip(weights, group)):
                pairs.append((loss, w * (1.0 if factors is None else float(factors[i]))))
        if LossBase._collector is not None:
            LossBase._collector.append(pairs)
            return _DeferredTerms(len(LossBase._collector) - 1)
        merged = None
        for loss, w in pairs:
            merged = loss * w if merged is None else merged + loss * w
        return merged.unsqueeze(1)


class PhotometricLoss(LossBase):
    def __init__(self, method, scale_weights, key_suffix=""):
        table = {"L1": lsu.photometric_loss_l1, "L2": lsu.photometric_loss_l2, "SSIM": lsu.photometric_loss_ssim}
        if method not in table:
            raise WrongInputException("Wrong photometric loss name: " + method)
        self.method = method
        self.photometric_loss = table[method]
        self.key_suffix = key_suffix
        self.scale_weights = scale_weights


class PhotometricLossMultiScale(PhotometricLoss):
    def __call__(self, features, predictions, augm_data):
        """losses.py:179-195 -> photo_loss [batch, 1]"""
        fused = augm_data.get("fused_photo_ms" + self.key_suffix)
        if fused is not None:           # (l1, ssim) per scale from the fused march kernel
            return self.merge_multi_scale_losses([pair[0 if self.method == "L1" else 1] for pair in fused])
        target_ms = augm_data["target_ms" + self.key_suffix]
        synth_ms = augm_data["synth_target_ms" + self.key_suffix]
        losses = [self.photometric_loss(synt, orig) for synt, orig in zip(synth_ms, target_ms)]
        return self.merge_multi_scale_losses(losses)


def resize_bilinear(srcimg, dst_hw):
    """losses.py:377-383: [B,N,h,w,C] -> [B,N,Hd,Wd,C] (TF2 bilinear; differentiable up-sampling)."""
    B, N, Hs, Ws, C = srcimg.shape
    dst = uf.resize_bilinear_tf(srcimg.reshape(B * N, Hs, Ws, C), dst_hw)
    return dst.reshape(B, N, dst_hw[0], dst_hw[1], C)


class MonoDepth2LossMultiScale(PhotometricLoss):
    """losses.py:198-232: per-pixel loss at full resolution, minimum over the source views."""

    def __call__(self, features, predictions, augm_data):
        synth_ms = augm_data["synth_target_ms" + self.key_suffix]
        original_target = augm_data["target" + self.key_suffix]
        Ho, Wo = original_target.shape[1:3]
        losses = []
        for synt in synth_ms:
            loss = self.photometric_loss(resize_bilinear(synt, (Ho, Wo)), original_target, False)
            loss = torch.min(loss, dim=1).values
            losses.append(torch.mean(loss, dim=[1, 2, 3]))
        return self.merge_multi_scale_losses(losses)


class CombinedLossMultiScale(PhotometricLoss):
    """losses.py:235-279: the static (depth + pose) per-pixel loss counts only where it is BELOW the optical-flow
    loss of the finest flow scale (both compared at the original resolution); mean over everything."""

    def __call__(self, features, predictions, augm_data):
        synth_ms = augm_data["synth_target_ms" + self.key_suffix]
        warped_ms = augm_data["warped_target_ms" + self.key_suffix]
        original_target = augm_data["target" + self.key_suffix]
        Ho, Wo = original_target.shape[1:3]
        flow_loss = self.photometric_loss(resize_bilinear(warped_ms[0], (Ho, Wo)), original_target, False)
        losses = []
        for synt in synth_ms:
            static_loss = self.photometric_loss(resize_bilinear(synt, (Ho, Wo)), original_target, False)
            mask = (static_loss < flow_loss).to(static_loss.dtype)          # tf.cast(static < flow): no gradient
            losses.append(torch.mean(static_loss * mask, dim=[1, 2, 3, 4]))
        return self.merge_multi_scale_losses(losses)


class MD2CombLossMultiScale(PhotometricLoss):
    """losses.py:324-374 (a class the reference defines but does not register in loss_factory.py): per-pixel static loss;
    where it exceeds TWICE the optical-flow loss of the finest flow scale it is pushed out of the running by +1000; the
    minimum over the source views is then averaged over the pixels that stayed below 1000 -- per sample sum, divided by
    the number of kept elements of the WHOLE batch tensor (`tf.math.count_nonzero(mask)` without an axis, :369)."""

    def __call__(self, features, predictions, augm_data):
        synth_ms = augm_data["synth_target_ms" + self.key_suffix]
        warped_ms = augm_data["warped_target_ms" + self.key_suffix]
        original_target = augm_data["target" + self.key_suffix]
        Ho, Wo = original_target.shape[1:3]
        flow_loss = self.photometric_loss(resize_bilinear(warped_ms[0], (Ho, Wo)), original_target, False)
        losses = []
        for synt in synth_ms:
            static_loss = self.photometric_loss(resize_bilinear(synt, (Ho, Wo)), original_target, False)
            mask = (static_loss > flow_loss * 2.).to(static_loss.dtype)
            static_loss = static_loss + mask * 1000.
            static_loss = torch.min(static_loss, dim=1).values                # [B, H, W, 3]
            keep = (static_loss < 1000.).to(static_loss.dtype)
            losses.append(torch.sum(static_loss * keep, dim=[1, 2, 3]) / torch.count_nonzero(keep).to(static_loss.dtype))
        return self.merge_multi_scale_losses(losses)


class FlowWarpLossMultiScale(PhotometricLoss):
    """losses.py:497-519: photometric loss between the flow-warped sources and the target at every flow scale."""

    def __call__(self, features, predictions, augm_data):
        flow_target_ms = augm_data["flow_target_ms" + self.key_suffix]
        warped_target_ms = augm_data["warped_target_ms" + self.key_suffix]
        losses = [self.photometric_loss(warp, orig) for warp, orig in zip(warped_target_ms, flow_target_ms)]
        return self.merge_multi_scale_losses(losses)


class L2Regularizer(LossBase):
    """losses.py:522-533: sum over the given weights of tf.nn.l2_loss(w) = sum(w^2) / 2, tiled to [batch].

    The VALUE is computed here (two multi-tensor launches over the ~110 FlowNet weights); its gradient, weight * w, is
    added to the flat gradient buffer by the optimizer (`KerasAdam.add_l2`, wired up in
    model_main.create_training_parts) in one launch instead of ~110 per-weight autograd nodes -- and because the
    deferred parameter gradients (hip/ops.py GradSink) are written, not accumulated, into that buffer."""

    def __init__(self, weights_to_regularize):
        self.weights = list(weights_to_regularize)
        self.scale_weights = None

    def __call__(self, features, predictions, augm_data):
        with torch.no_grad():
            norms = torch._foreach_norm([w.detach() for w in self.weights])
            loss = torch.stack(norms).float().square().sum() * 0.5
        batch = features["image5d"].shape[0]
        return loss.reshape(1).expand(batch)


class MoALossMultiScale(PhotometricLoss):
    """losses.py:282-321: minimum over the temporal views and the stereo view."""

    def __call__(self, features, predictions, augm_data):
        temp_ms = augm_data["synth_target_ms" + self.key_suffix]
        stro_ms = augm_data["stereo_synth_ms"]                        # losses.py:295 (no suffix, as in the reference)
        original_target = augm_data["target" + self.key_suffix]
        Ho, Wo = original_target.shape[1:3]
        losses = []
        for temp_target, stro_target in zip(temp_ms, stro_ms):
            temp_loss = self.photometric_loss(resize_bilinear(temp_target, (Ho, Wo)), original_target, False)
            stro_loss = self.photometric_loss(resize_bilinear(stro_target, (Ho, Wo)), original_target, False)
            moa = torch.min(torch.cat([temp_loss, stro_loss], dim=1), dim=1).values
            losses.append(torch.mean(moa, dim=[1, 2, 3]))
        return self.merge_multi_scale_losses(losses)


class SmoothenessLossMultiScale(LossBase):
    def __init__(self, scale_weights, key_suffix=""):
        self.key_suffix = key_suffix
        self.scale_weights = scale_weights

    def __call__(self, features, predictions, augm_data):
        """losses.py:391-407: every scale's loss is divided by its scale factor."""
        disp_ms = predictions["disp_ms" + self.key_suffix]
        target_ms = augm_data["target_ms" + self.key_suffix]
        orig_width = target_ms[0].shape[2]
        factors = [image.shape[2] / orig_width for image in target_ms]   # each scale's loss divided by its scale factor
        if len(disp_ms) <= 4 and _FUSED_MULTI_SCALE:
            losses = _ops.smoothness_multi_scale(list(disp_ms), list(target_ms), float(opts.IMAGE_GRADIENT_FACTOR))
        else:
            losses = [self.smootheness_loss(disp, image) for disp, image in zip(disp_ms, target_ms)]
        return self.merge_multi_scale_losses(losses, factors=factors)

    def smootheness_loss(self, disp, image):
        """losses.py:409-440 -> [batch]"""
        return _ops.smoothness(disp, image, float(opts.IMAGE_GRADIENT_FACTOR))


class StereoDepthLoss(PhotometricLoss):
    def __init__(self, method, scale_weights):
        super().__init__(method, scale_weights)

    def __call__(self, features, predictions, augm_data):
        """losses.py:447-478: left-from-right + right-from-left photometric loss per scale."""
        if "fused_stereo_ms" in augm_data:
            k = 0 if self.method == "L1" else 1
            left = [pair[k] for pair in augm_data["fused_stereo_ms"]]
            right = [pair[k] for pair in augm_data["fused_stereo_ms_R"]]
            return self.merge_multi_scale_losses([left, right])
        left = self.stereo_photometric_loss(augm_data["stereo_synth_ms"], augm_data["target_ms"])
        right = self.stereo_photometric_loss(augm_data["stereo_synth_ms_R"], augm_data["target_ms_R"], "_R")
        return self.merge_multi_scale_losses([left, right])

    def stereo_photometric_loss(self, synth_target_ms, target_ms, suffix=""):
        return [self.photometric_loss(s, t) for s, t in zip(synth_target_ms, target_ms)]


class StereoPoseLoss(LossBase):
    def __call__(self, features, predictions, augm_data):
        """losses.py:481-494: MSE between the known stereo extrinsic (as twist) and the PoseNet's
        left<->right predictions, mean over numsrc -> [batch]"""
        pose_lr_true_mat = features["stereo_T_LR"].unsqueeze(1)
        pose_rl_true_mat = cp.rigid_inverse(pose_lr_true_mat)
        pose_lr_true = cp.pose_matr2rvec_batch(pose_lr_true_mat)
        pose_rl_true = cp.pose_matr2rvec_batch(pose_rl_true_mat)
        loss = torch.mean(torch.square(pose_lr_true - predictions["pose_LR"]), dim=-1) \
            + torch.mean(torch.square(pose_rl_true - predictions["pose_RL"]), dim=-1)
        return torch.mean(loss, dim=1)
