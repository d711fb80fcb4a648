"""SynthesizeMultiScale / SynthesizeSingleScale with the reference's call signatures
(model/synthesize/synthesize_base.py:10-58), backed by the gfx950 kernels:
pose twist -> matrix (K0), source pyramid (K1), projective warp + bilinear sampling (K2+K3)."""
from ...hip import ops as _ops
from ...utils.convert_pose import pose_rvec2matr_batch_tf


_FUSED_MULTI_SCALE = __import__("os").environ.get("XPT_DEBUG_PER_SCALE_LOSS", "0") != "1"     # A/B: one launch per scale


class SynthesizeMultiScale:
    def __call__(self, source_image, intrinsic, pred_depth_ms, pred_pose):
        """
        :param source_image: source images [batch, numsrc, height, width, 3]
        :param intrinsic: [batch, 3, 3]
        :param pred_depth_ms: predicted target depth in multi scale, list of [batch, height/scale, width/scale, 1]
        :param pred_pose: twist poses that transform target points to each source frame [batch, numsrc, 6]
        :return: reconstructed target view in multi scale, list of [batch, numsrc, height/scale, width/scale, 3]
        """
        poses_matr = pose_rvec2matr_batch_tf(pred_pose)
        return [SynthesizeSingleScale()(source_image, intrinsic, depth_sc, poses_matr) for depth_sc in pred_depth_ms]

    def photometric_losses(self, source_image, intrinsic, pred_depth_ms, pred_pose, target_ms, sources_ms=None,
                           grad_hint=None):
        """Fused fast path: per scale (photometric L1 [batch], photometric SSIM [batch]) of the synthesized views
        against target_ms, computed by the warp+L1+SSIM march kernel without materialising the views.
        sources_ms: the sources already resized to every scale (hip/ops.py image_pyramids), else resized here.
        grad_hint: (d total / d L1 per scale, d total / d SSIM per scale) when the caller knows its loss weights: the
        march then leaves the gradients in the same pass as the loss values (hip/ops.py _PhotoFusedMS)."""
        poses_matr = pose_rvec2matr_batch_tf(pred_pose)
        if source_image.shape[1] in (1, 4) and len(pred_depth_ms) <= 4 and _FUSED_MULTI_SCALE:
            # every scale in one march launch (csrc/xpt_march.hip)
            singles = [SynthesizeSingleScale() for _ in pred_depth_ms]
            sources = []
            for i, (single, depth_sc) in enumerate(zip(singles, pred_depth_ms)):
                single.read_shape(source_image, depth_sc)
                sources.append(sources_ms[i] if sources_ms is not None else single.resize_source_images(source_image))
            return _ops.photo_fused_multi_scale(sources, list(pred_depth_ms), poses_matr, intrinsic, list(target_ms),
                                                [s.scale for s in singles], grad_hint=grad_hint)
        return [SynthesizeSingleScale().photometric_losses(source_image, intrinsic, depth_sc, poses_matr, target_sc)
                for depth_sc, target_sc in zip(pred_depth_ms, target_ms)]


class SynthesizeSingleScale:
    def __init__(self, shape=(0, 0, 0), numsrc=0, scale=0):
        self.batch, self.height_sc, self.width_sc = shape
        self.numsrc = numsrc
        self.scale = scale

    def __call__(self, source_image, intrinsic, depth_sc, poses_matr):
        """source_image [batch, numsrc, height, width, 3] (full resolution), intrinsic [batch, 3, 3] (full
        resolution; scaled inside the kernel as scale_intrinsic does, :66-71), depth_sc
        [batch, height/scale, width/scale, 1], poses_matr [batch, numsrc, 4, 4]
        -> [batch, numsrc, height/scale, width/scale, 3]"""
        self.read_shape(source_image, depth_sc)
        source_images_sc = self.resize_source_images(source_image)
        return _ops.warp(source_images_sc, depth_sc, poses_matr, intrinsic, self.scale)

    def photometric_losses(self, source_image, intrinsic, depth_sc, poses_matr, target_sc):
        """Fused fast path (no reference counterpart as ONE call): photometric_loss_l1 and photometric_loss_ssim
        (loss_util.py:6-25, 52-96) of the view this object would synthesize, WITHOUT writing the synthesized image:
        -> (l1 [batch], ssim [batch]).  Same arguments as __call__ plus the scaled target [batch, h, w, 3]."""
        self.read_shape(source_image, depth_sc)
        source_images_sc = self.resize_source_images(source_image)
        return _ops.photo_fused(source_images_sc, depth_sc, poses_matr, intrinsic, target_sc, self.scale)

    def read_shape(self, source_image, depth_sc):
        _, self.numsrc, height_orig, _, _ = source_image.shape
        self.batch, self.height_sc, self.width_sc, _ = depth_sc.shape
        self.scale = int(height_orig // self.height_sc)          # synthesize_base.py:64

    def scale_intrinsic(self, intrinsic, scale):
        """synthesize_base.py:66-71 (kept for API parity; the warp kernel applies the same scaling itself)."""
        out = intrinsic.clone()
        out[:, :2, :] = intrinsic[:, :2, :] / scale
        out[:, 2, :] = out.new_tensor([0., 0., 1.])
        return out

    def resize_source_images(self, source_image):
        """synthesize_base.py:74-85: TF2 bilinear resize of every source to this scale."""
        batch, numsrc, height, width, ch = source_image.shape
        flat = source_image.reshape(batch * numsrc, height, width, ch)
        flat = _ops.resize_down(flat, self.scale)
        return flat.reshape(batch, numsrc, self.height_sc, self.width_sc, ch)
