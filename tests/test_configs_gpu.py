"""The BASELINE.json configurations that had no GPU coverage: configs[3] (256x832, batch 4, 4 scales: total-loss parity of
the fused kernels against the oracle at that size) and configs[4] (stereo + mono losses over the mixed dataset image sizes
of config-example.py:25-30, cycled per step: one captured hipGraph per input signature)."""
import pytest
import torch

from oracle import ref_loss
from tests.test_total_loss_gpu import fake_predictions, flip_safe_predictions, leaves
from tests.util import frac_close
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.utils import synthetic_data as sd

pytestmark = pytest.mark.gpu
MIXED_SHAPES = [(128, 512), (192, 512), (256, 384), (192, 384)]


def test_total_loss_c4_256x832_batch4(gpu_device):
    from xpt_mde_2021_amd.model.loss_and_metric.loss_factory import loss_factory
    B, H, W = 4, 256, 832
    feats = sd.make_features(B, H, W, 5, 31, False)
    total_loss = loss_factory(sd.tfr_config_for(feats), opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1, False, None, B)
    total_loss.fused = True
    raw = flip_safe_predictions(feats, fake_predictions(feats, 8, False), False)
    p_ref = leaves(raw, "cpu", torch.float64)
    tot_ref, by_ref = ref_loss.total_loss(p_ref, {k: v.double() for k, v in feats.items()}, dict(total_loss.loss_weights),
                                          opts.SCALE_WEIGHT_T1, False, B)
    tot_ref.backward()
    p = leaves(raw, gpu_device, torch.float32)
    tot, by = total_loss(p, {k: v.to(gpu_device) for k, v in feats.items()})
    tot.backward()
    torch.cuda.synchronize()
    for k in by:
        frac_close(by[k], by_ref[k], 2e-5, rtol=2e-4, what=f"loss {k}")
    frac_close(tot, tot_ref, 1e-4, rtol=2e-4, what="total loss")
    for i, (d, dr) in enumerate(zip(p["depth_ms"], p_ref["depth_ms"])):
        scale = dr.grad.abs().max().item()
        frac_close(d.grad, dr.grad, 1e-3 * scale, max_bad_frac=0.0, what=f"d depth_ms[{i}]")
    scale = p_ref["pose"].grad.abs().max().item()
    frac_close(p["pose"].grad, p_ref["pose"].grad, 1e-3 * scale, max_bad_frac=0.0, what="d pose")


def test_mixed_shape_stereo_steps_c5(gpu_device):
    """configs[4]: LOSS_RIGID_T2 (mono + stereo L1 / SSIM / smoothness + stereoPose) on a stereo feature dict whose image
    size changes every step; the graph trainer captures each size once and replays it on the second visit."""
    from xpt_mde_2021_amd.model import model_main as mm
    from xpt_mde_2021_amd.model import train_val as tv
    saved = (opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE, dict(opts.IMAGE_SIZES), opts.STEREO)
    opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 2
    # (XPT_HALF=fp16: this process runs the half-precision build -- configs[4] as written, "fp16 convs + fp32 loss
    #  accumulation"; tests/test_fp16_build_gpu.py starts this test that way)
    from xpt_mde_2021_amd.hip import lib as _xlib
    opts.CONV_DTYPE, opts.STEREO = _xlib.half_format(), True
    try:
        batches, cfg = [], None
        for hw in MIXED_SHAPES:
            opts.IMAGE_SIZES["kitti_raw"] = hw
            dataset, c, _ = mm.get_dataset("synthetic_stereo", "train", True)
            batches.append(dataset.batches[0])
            cfg = cfg or c
        torch.manual_seed(0)
        model, _, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T2, opts.SCALE_WEIGHT_T1,
                                                                   opts.RIGID_NET, ckpt_name="__test__")
        trainer, _ = tv.train_val_factory("graph", model, loss_object, 0, True, None, optimizer)
        captures = []
        original = trainer._graph._capture
        trainer._graph._capture = lambda f, s: (captures.append(s), original(f, s))[1]
        losses = []
        for _ in range(2):
            for feats in batches:
                _, loss, by_type = trainer.run_a_batch(feats)
                losses.append(float(loss))
        assert len(trainer._graph.cache) == len(MIXED_SHAPES) == len(set(captures)) == len(captures)
        assert all(l == l and abs(l) < 1e4 for l in losses), losses
        assert {"L1", "SSIM", "smoothe", "stereoL1", "stereoSSIM", "stereoPose"} <= set(by_type)
        assert not trainer._graph.eager_fallback
    finally:
        opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE = saved[:3]
        opts.IMAGE_SIZES.clear()
        opts.IMAGE_SIZES.update(saved[3])
        opts.STEREO = saved[4]
