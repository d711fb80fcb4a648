"""loss_factory with the reference's signature (model/loss_and_metric/loss_factory.py:6-74)."""
import numpy as np

from ...config import opts
from ...utils.util_class import WrongInputException
from . import losses as lm


def loss_factory(dataset_cfg, loss_weights, scale_weights, stereo=None, weights_to_regularize=None, batch_size=None):
    """Builds TotalLoss from a {name: weight} dict.  Zero-weight losses and losses whose dataset keys are missing
    are dropped exactly as loss_factory.py:41-47, 55-74 do.  `batch_size` is the GLOBAL batch."""
    stereo = opts.STEREO if stereo is None else stereo
    batch_size = opts.BATCH_SIZE if batch_size is None else batch_size
    sw = np.asarray(scale_weights, dtype=np.float32).reshape(-1, 1)

    def build(name):
        sfx = "_R" if name.endswith("_R") else ""
        base = name[:-2] if sfx else name
        if base in ("L1", "SSIM"):
            return lm.PhotometricLossMultiScale(base, sw, key_suffix=sfx)
        if base in ("md2L1", "md2SSIM"):
            return lm.MonoDepth2LossMultiScale(base[3:], sw, key_suffix=sfx)
        if base in ("moaL1", "moaSSIM"):
            return lm.MoALossMultiScale(base[3:], sw, key_suffix=sfx)
        if base == "smoothe":
            return lm.SmoothenessLossMultiScale(sw, key_suffix=sfx)
        if name in ("stereoL1", "stereoSSIM"):
            return lm.StereoDepthLoss(name[6:], sw)
        if name == "stereoPose":
            return lm.StereoPoseLoss()
        if base in ("cmbL1", "cmbSSIM"):
            return lm.CombinedLossMultiScale(base[3:], sw, key_suffix=sfx)
        if base == "flowL2":
            return lm.FlowWarpLossMultiScale("L2", sw, key_suffix=sfx)
        if name == "flow_reg":
            if not weights_to_regularize:
                raise WrongInputException("loss 'flow_reg' needs a FlowNet in the model (weights_to_regularize is empty)")
            return lm.L2Regularizer(weights_to_regularize)
        raise WrongInputException(f"unknown loss name '{name}'")

    losses, weights = dict(), dict()
    for name, weight in loss_weights.items():
        if weight == 0.:
            continue
        if not check_loss_dependency(name, dataset_cfg):
            continue
        losses[name] = build(name)
        weights[name] = weight
    print("[loss_factory] loss weights:", weights)
    print("[loss_factory] scale weights:", sw[:, 0])
    return lm.TotalLoss(losses, weights, stereo, batch_size)


_DEPENDENCY = [
    (("L1", "SSIM", "smoothe", "md2L1", "md2SSIM", "cmbL1", "cmbSSIM", "flowL2", "flow_reg"), ("image", "intrinsic")),
    (("L1_R", "SSIM_R", "smoothe_R", "md2L1_R", "md2SSIM_R", "cmbL1_R", "cmbSSIM_R", "flowL2_R"),
     ("image_R", "intrinsic_R")),
    (("stereoL1", "stereoSSIM", "stereoPose", "moaL1", "moaSSIM", "moaL1_R", "moaSSIM_R"),
     ("image", "intrinsic", "image_R", "intrinsic_R", "stereo_T_LR")),
]


def check_loss_dependency(loss_key, dataset_cfg):
    """loss_factory.py:55-74 (the moa* losses read stereo_synth_ms, so they are listed with the stereo group)."""
    for loss_names, data_names in _DEPENDENCY:
        if loss_key in loss_names:
            for dep in data_names:
                if dep not in dataset_cfg:
                    print(f"[check_loss_dependency] {loss_key} loss is excluded because {dep} is NOT in dataset")
                    return False
    return True
