"""GPU parity of the orchestration: TotalLoss (mono LOSS_RIGID_T1 and stereo LOSS_RIGID_T2 sets, 4 scales) through
the reference-named boundary classes against the oracle's restatement of losses.py, values and gradients."""
import pytest
import torch

from oracle import ref_loss
from tests.util import flip_safe_depth, frac_close
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.utils import synthetic_data as sd

pytestmark = pytest.mark.gpu


def fake_predictions(feats, seed, stereo):
    g = torch.Generator().manual_seed(seed)
    B, S, H, W, _ = feats["image5d"].shape
    preds = {}
    for sfx in ("", "_R") if stereo else ("",):
        preds["depth_ms" + sfx] = [sd.smooth_depth(B, H // s, W // s, g, lo=1.0, hi=60.0) for s in (1, 2, 4, 8)]
        preds["pose" + sfx] = sd.random_poses(B, S - 1, g) * 0.3
    if stereo:
        preds["pose_LR"] = sd.random_poses(B, S - 1, g) * 0.1
        preds["pose_RL"] = sd.random_poses(B, S - 1, g) * 0.1
    return preds


def flip_safe_predictions(feats, preds, stereo):
    """Depth predictions with the pixels moved (by a few per cent of their depth, on both sides of the comparison) whose projection into any
    of the views the losses synthesize -- the four temporal sources and, with stereo, the other camera -- lies within fp32
    rounding of an integer coordinate or of the validity border (tests/util.py risky_pixels, from the fp64 oracle)."""
    from oracle import ref_pose
    shares = []
    for sfx in ("", "_R") if stereo else ("",):
        views = [ref_pose.pose_rvec2matr_batch(preds["pose" + sfx].double())]
        if stereo:
            T_lr = feats["stereo_T_LR"].double().unsqueeze(1)
            views.append(torch.linalg.inv(T_lr) if sfx == "" else T_lr)             # losses.py:106-140 synethesize_stereo
        T = torch.cat(views, dim=1)
        H = feats["image5d"].shape[2]
        out = []
        for d in preds["depth_ms" + sfx]:
            d, share = flip_safe_depth(d, T, feats["intrinsic" + sfx], H // d.shape[1], nudge=True)   # (depth 0 would be NaN in the smoothness term)
            out.append(d)
            shares.append(share)
        preds["depth_ms" + sfx] = out
    assert max(shares) < 0.5, shares          # (a near-rectified stereo pair keeps many rows close to integer source rows)
    return preds


def leaves(preds, device, dtype):
    out = {}
    for k, v in preds.items():
        if isinstance(v, list):
            out[k] = [t.to(device=device, dtype=dtype).requires_grad_(True) for t in v]
        else:
            out[k] = v.to(device=device, dtype=dtype).requires_grad_(True)
    for sfx in ("", "_R"):
        if "depth_ms" + sfx in out:
            out["disp_ms" + sfx] = ref_loss.safe_reciprocal_number_ms(out["depth_ms" + sfx])
    return out


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("stereo,loss_set", [(False, "LOSS_RIGID_T1"), (True, "LOSS_RIGID_T2")])
def test_total_loss_matches_oracle(gpu_device, stereo, loss_set, fused):
    from xpt_mde_2021_amd.model.loss_and_metric.loss_factory import loss_factory
    B, H, W = 2, 64, 208
    feats = sd.make_features(B, H, W, 5, 99, stereo)
    if stereo:
        # A rectified pair (identity rotation, pure x translation) maps every target row EXACTLY onto an integer
        # source row, so the validity of the whole first / last row hinges on the last ulp of v' (bilinear_interp.py
        # :64-73 is strict) and no two fp32 implementations agree there.  Tilt the extrinsic slightly so that the
        # comparison is well conditioned; the rectified case is covered row-wise in test_rectified_stereo_interior.
        from oracle import ref_pose
        twist = torch.tensor([[[0.54, 0.013, 0.004, 0.004, -0.006, 0.003]]]).repeat(B, 1, 1)
        feats["stereo_T_LR"] = ref_pose.pose_rvec2matr_batch(twist)[:, 0].contiguous()
    cfg = sd.tfr_config_for(feats)
    weights = getattr(opts, loss_set)
    total_loss = loss_factory(cfg, weights, opts.SCALE_WEIGHT_T2, True, None, B)
    total_loss.fused = fused              # fused warp+L1+SSIM march kernels vs the separate synthesize / loss kernels
    raw = flip_safe_predictions(feats, fake_predictions(feats, 5, stereo), stereo)

    p_ref = leaves(raw, "cpu", torch.float64)
    f_ref = {k: v.double() for k, v in feats.items()}
    w_ref = {k: v for k, v in total_loss.loss_weights.items()}
    tot_ref, by_ref = ref_loss.total_loss(p_ref, f_ref, w_ref, opts.SCALE_WEIGHT_T2, True, B)
    tot_ref.backward()

    p = leaves(raw, gpu_device, torch.float32)
    f = {k: v.to(gpu_device) for k, v in feats.items()}
    tot, by = total_loss(p, f)
    tot.backward()

    assert set(by) == set(by_ref)
    for k in by:
        frac_close(by[k], by_ref[k], 2e-5, rtol=2e-4, what=f"loss {k}")
    frac_close(tot, tot_ref, 1e-4, rtol=2e-4, what="total loss")
    for sfx in ("", "_R") if stereo else ("",):
        for i, (d, dr) in enumerate(zip(p["depth_ms" + sfx], p_ref["depth_ms" + sfx])):
            scale = dr.grad.abs().max().item()
            frac_close(d.grad, dr.grad, 1e-3 * scale, max_bad_frac=0.0, what=f"d depth_ms{sfx}[{i}]")
        scale = p_ref["pose" + sfx].grad.abs().max().item()
        frac_close(p["pose" + sfx].grad, p_ref["pose" + sfx].grad, 1e-3 * scale, max_bad_frac=0.0, what=f"d pose{sfx}")
    if stereo:
        for k in ("pose_LR", "pose_RL"):
            frac_close(p[k].grad, p_ref[k].grad, 1e-6, rtol=1e-4, what=f"d {k}")


def test_rectified_stereo_interior(gpu_device):
    """Rectified stereo (R = I, t = (0.54, 0, 0)): rows 1..h-2 of the stereo-synthesized view must match the oracle;
    the first / last row is the documented ulp-level coin flip of the reference algorithm."""
    from oracle import ref_pose, ref_synthesize as rs
    from xpt_mde_2021_amd.model.synthesize.synthesize_base import SynthesizeMultiScale
    from xpt_mde_2021_amd.utils import convert_pose as cp
    B, H, W = 2, 64, 208
    feats = sd.make_features(B, H, W, 5, 99, True)
    g = torch.Generator().manual_seed(5)
    depth_ms = [sd.smooth_depth(B, H // s, W // s, g, lo=1.0, hi=60.0) for s in (1, 2, 4, 8)]
    T = feats["stereo_T_LR"]
    src = feats["image5d_R"][:, -1].unsqueeze(1)
    p_ref = ref_pose.pose_matr2rvec_batch(torch.linalg.inv(T.double()).unsqueeze(1))
    ref = rs.synthesize_multi_scale(src.double(), feats["intrinsic"].double(), [d.double() for d in depth_ms], p_ref)
    p = cp.pose_matr2rvec_batch(torch.linalg.inv(T.to(gpu_device)).unsqueeze(1))
    assert torch.equal(p.cpu(), p_ref.float())
    out = SynthesizeMultiScale()(src.to(gpu_device), feats["intrinsic"].to(gpu_device),
                                 [d.to(gpu_device) for d in depth_ms], p)
    for a, b in zip(out, ref):
        frac_close(a[:, :, 1:-1], b[:, :, 1:-1], 1e-4, max_bad_frac=2e-3, what=f"rectified stereo synth {tuple(a.shape)}")


def test_train_step_runs_and_learns(gpu_device):
    """Eager and hipGraph trainers: a few steps on one synthetic batch reduce the loss and agree with each other."""
    from xpt_mde_2021_amd.model import model_main as mm
    from xpt_mde_2021_amd.model import train_val as tv
    saved = (opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE, dict(opts.IMAGE_SIZES))
    opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 2
    opts.IMAGE_SIZES["kitti_raw"] = (64, 192)
    try:
        losses = {}
        for mode, dtype in (("eager", "fp32"), ("graph", "fp32"), ("graph", "bf16"), ("eager", "bf16")):
            opts.CONV_DTYPE = dtype
            torch.manual_seed(0)
            dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
            model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1,
                                                                          opts.SCALE_WEIGHT_T1, opts.RIGID_NET,
                                                                          ckpt_name="__test__")
            trainer, _ = tv.train_val_factory(mode, model, loss_object, 0, False, None, optimizer)   # no random augmentation
            feats = dataset.batches[0]
            hist = []
            for _ in range(8):
                _, loss, by_type = trainer.run_a_batch(feats)
                hist.append(float(loss))
            assert all(torch.isfinite(torch.tensor(hist))), hist
            assert hist[-1] < hist[0], (mode, dtype, hist)
            losses[(mode, dtype)] = hist
        a, b = losses[("eager", "fp32")], losses[("graph", "fp32")]
        # fp32 = the library path (MIOpen solvers with atomics: runs are not repeatable bit for bit, and 8 Adam steps
        # amplify the rounding differences -- Adam moves every weight by ~lr whatever the gradient's size)
        # (first step: the same weights through whatever solver MIOpen's find picked in each trainer's warm-up -- seen 2e-6 to
        #  2e-5 apart depending on the box)
        assert abs(a[0] - b[0]) < 1e-4 * max(1.0, abs(a[0])) and abs(a[-1] - b[-1]) < 5e-3 * abs(a[-1]), (a, b)
        c = losses[("graph", "bf16")]
        assert abs(a[0] - c[0]) < 5e-2 * abs(a[0]), (a, c)
        # bf16 = own deterministic kernels end to end: the captured trainer IS the eager trainer, bit for bit
        assert losses[("eager", "bf16")] == c, (losses[("eager", "bf16")], c)
    finally:
        opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE = saved[:3]
        opts.IMAGE_SIZES.clear()
        opts.IMAGE_SIZES.update(saved[3])


def test_step_leaves_no_autograd_graph_behind(gpu_device):
    """No tensor with a grad_fn may survive a training step (a module attribute holding one keeps the whole autograd
    graph and its AccumulateGrad nodes -- created on the warm-up stream -- alive into the hipGraph capture), the
    capture must not trip autograd's AccumulateGrad stream-mismatch warning, and the graph trainer must not have
    fallen back to eager execution."""
    import gc
    import warnings
    from xpt_mde_2021_amd.model import model_main as mm
    from xpt_mde_2021_amd.model import train_val as tv
    saved = (opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE, dict(opts.IMAGE_SIZES))
    opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 2
    opts.IMAGE_SIZES["kitti_raw"] = (64, 192)
    opts.CONV_DTYPE = "bf16"
    try:
        dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
        model, aug, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1,
                                                                      opts.SCALE_WEIGHT_T1, opts.RIGID_NET,
                                                                      ckpt_name="__test__")
        trainer, _ = tv.train_val_factory("eager", model, loss_object, 0, False, None, optimizer)
        gc.collect()
        trainer.run_a_batch(dataset.batches[0])
        torch.cuda.synchronize()
        alive = []
        for o in gc.get_objects():
            try:
                if torch.is_tensor(o) and o.grad_fn is not None:
                    alive.append((tuple(o.shape), type(o.grad_fn).__name__))
            except ReferenceError:
                pass
        assert not alive, f"tensors keeping the step's autograd graph alive: {alive[:8]}"

        trainer, _ = tv.train_val_factory("graph", model, loss_object, 0, False, None, optimizer)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for _ in range(3):
                trainer.run_a_batch(dataset.batches[0])
            torch.cuda.synchronize()
        stale = [str(w.message)[:120] for w in caught if "AccumulateGrad" in str(w.message)]
        assert not stale, stale
        assert not trainer._graph.eager_fallback
    finally:
        opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE = saved[:3]
        opts.IMAGE_SIZES.clear()
        opts.IMAGE_SIZES.update(saved[3])


def test_stereo_train_step_in_graph(gpu_device):
    """configs[4]-style step (stereo feature dict, LOSS_RIGID_T2: mono + stereo L1/SSIM + stereoPose) captured as a
    hipGraph: nothing in it may synchronise with the host (the pose inverses are closed-form), every layer is applied
    to the left and to the right snippet, so each deferred parameter gradient has two segments."""
    from xpt_mde_2021_amd.model import model_main as mm
    from xpt_mde_2021_amd.model import train_val as tv
    saved = (opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE, dict(opts.IMAGE_SIZES), opts.STEREO)
    opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 2
    opts.IMAGE_SIZES["kitti_raw"] = (64, 192)
    opts.STEREO = True
    try:
        losses = {}
        for mode in ("eager", "graph"):
            opts.CONV_DTYPE = "fp32"
            torch.manual_seed(0)
            dataset, cfg, _ = mm.get_dataset("synthetic_stereo", "train", True)
            model, _, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T2,
                                                                       opts.SCALE_WEIGHT_T1, opts.RIGID_NET,
                                                                       ckpt_name="__test__")
            trainer, _ = tv.train_val_factory(mode, model, loss_object, 0, True, None, optimizer)
            hist = [float(trainer.run_a_batch(dataset.batches[0])[1]) for _ in range(5)]
            assert all(h == h for h in hist) and hist[-1] < hist[0], (mode, hist)
            losses[mode] = hist
        a, b = losses["eager"], losses["graph"]
        # the synthetic pair is rectified: border rows sit exactly on the validity boundary, so the
        # two runs may disagree on a few rows through last-bit differences of the library convolutions
        assert abs(a[0] - b[0]) < 2e-3 * abs(a[0]) and abs(a[-1] - b[-1]) < 5e-2 * abs(a[-1]), (a, b)
    finally:
        opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE = saved[:3]
        opts.IMAGE_SIZES.clear()
        opts.IMAGE_SIZES.update(saved[3])
        opts.STEREO = saved[4]
