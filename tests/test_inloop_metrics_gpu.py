"""The per-step metrics of merge_results (model/train_val.py:157-236) as computed on the device against the host (numpy)
versions the reference's own known-answer tests pin (evaluate/eval_utils.py:109-131 valid_depth_filter, :134-154
compute_depth_metrics, :9-87 PoseMetricNumpy)."""
import numpy as np
import pytest
import torch

from xpt_mde_2021_amd.evaluate import eval_utils as eu
from xpt_mde_2021_amd.utils import synthetic_data as sd

pytestmark = pytest.mark.gpu


def test_depth_metric_matches_eval_utils(gpu_device):
    from xpt_mde_2021_amd.model import train_val as tv
    g = torch.Generator().manual_seed(4)
    B, H, W = 3, 128, 416
    gt = sd.smooth_depth(B, H, W, g)
    lidar = (torch.rand((B, H, W, 1), generator=g) < 0.05).float()
    gt = gt * lidar                                                   # sparse "lidar" ground truth, 0 = no return
    gt[1, 70:90, 100:300, 0] = 95.0                                   # beyond MAX_DEPTH: filtered out
    pred = (gt.clamp(min=1.0) * 0.4 + 3.0 * torch.rand((B, H, W, 1), generator=g)).clamp(min=0.5)   # off in scale
    feats = {"depth_gt": gt.to(gpu_device)}
    preds = {"depth_ms": [pred.to(gpu_device)]}
    got = float(tv.get_depth_metric(feats, preds))
    rows = []
    for p, t in zip(pred.numpy()[..., 0], gt.numpy()[..., 0]):
        pv, tv_ = eu.valid_depth_filter(p, t)
        rows.append(eu.compute_depth_metrics(pv, tv_)[0])
    assert abs(got - float(np.mean(rows))) < 2e-5 * float(np.mean(rows)) + 1e-7, (got, np.mean(rows))
    # center-depth read-out
    mt, mp = tv.get_center_depths(feats, preds)
    ys, xs = H // 4 * 3 - 10, W // 2 - 10
    win = gt[:, ys:ys + 20, xs:xs + 20, 0].numpy()
    expect = np.array([w[w > 0].mean() if (w > 0).any() else 0.0 for w in win])
    assert np.allclose(mt.cpu().numpy(), expect, rtol=1e-5)
    assert np.allclose(mp.cpu().numpy(), pred[:, ys:ys + 20, xs:xs + 20, :].mean(dim=(1, 2, 3)).numpy(), rtol=1e-5)


@pytest.mark.parametrize("H,W,density", [(128, 416, 0.05), (64, 208, 1.0), (37, 61, 0.3), (16, 52, 0.02)])
def test_depth_metric_kernel_edge_cases(gpu_device, H, W, density):
    """xpt_depth_metric (radix-selected medians) against the batched-sort formulation and numpy: even and odd mask counts,
    an empty mask, ties at the median, negative / zero predictions."""
    from xpt_mde_2021_amd.model import train_val as tv
    g = torch.Generator().manual_seed(H + W)
    B = 5
    gt = sd.smooth_depth(B, H, W, g) * (torch.rand((B, H, W, 1), generator=g) < density).float()
    pred = gt.clamp(min=1.0) * 0.7 + torch.randn((B, H, W, 1), generator=g)           # some values <= 0
    gt[1] = 0.0                                                     # sample 1: empty mask -> contributes 0
    gt[2] = torch.where(gt[2] > 0, torch.full_like(gt[2], 7.5), gt[2])      # ties: every valid gt equal
    pred[3] = torch.round(pred[3] * 2) / 2                           # many equal predictions around the median
    rows = gt[4, :, :, 0].nonzero()
    if len(rows) > 1:                                                # make sample 4's count differ in parity from sample 0's
        gt[4, rows[0, 0], rows[0, 1], 0] = 0.0
    feats = {"depth_gt": gt.to(gpu_device)}
    preds = {"depth_ms": [pred.to(gpu_device)]}
    got = float(tv.get_depth_metric(feats, preds))
    crop = (np.array([0.40810811 * H, 0.99189189 * H, 0.03594771 * W, 0.96405229 * W])).astype(np.int32)
    sorted_way = float(tv.depth_metric_batched(pred[..., 0].to(gpu_device), gt[..., 0].to(gpu_device), crop))
    assert abs(got - sorted_way) <= 1e-6 * abs(sorted_way) + 1e-9, (got, sorted_way)
    rows = []
    for p, t in zip(pred.numpy()[..., 0], gt.numpy()[..., 0]):
        pv, tv_ = eu.valid_depth_filter(p, t)
        rows.append(eu.compute_depth_metrics(pv, tv_)[0] if len(tv_) else 0.0)
    assert abs(got - float(np.mean(rows))) < 2e-5 * abs(float(np.mean(rows))) + 1e-7, (got, np.mean(rows))


def test_pose_metric_matches_eval_utils(gpu_device):
    from xpt_mde_2021_amd.model import train_val as tv
    g = torch.Generator().manual_seed(9)
    pose = sd.random_poses(5, 4, g)
    true_twist = pose + 0.05 * torch.randn(pose.shape, generator=g)
    true_mat = torch.from_numpy(eu.pose_rvec2matr_batch_np(true_twist.numpy())).float()
    ref = eu.PoseMetricNumpy()
    ref.compute_pose_errors(pose.numpy(), true_mat.numpy())
    expect = ref.get_mean_pose_error()
    got = tv.get_pose_metric({"pose": pose.to(gpu_device)}, {"pose_gt": true_mat.to(gpu_device)})
    for a, b in zip(got, expect):
        assert abs(float(a) - float(b)) < 2e-4 * abs(float(b)) + 1e-6, (float(a), float(b))
    # merge_results carries the three pose columns of the reference's history.csv (train_val.py:161-165)
    out = tv.merge_results({"pose_gt": true_mat.to(gpu_device)}, {"pose": pose.to(gpu_device)},
                           torch.zeros((), device=gpu_device), {}, False)
    assert {"trjabs", "trjrel", "roterr"} <= set(out)
    assert tv.get_pose_metric({"pose": pose.to(gpu_device)}, {})[0].item() == 0.0


def test_metrics_graph_matches_eager(gpu_device):
    """run_an_epoch's per-step record comes from a captured graph (train_val._MetricsGraph) that reads the training
    graph's static buffers: every replay must equal the eager merge_results on the same tensors."""
    from xpt_mde_2021_amd.config import opts
    from xpt_mde_2021_amd.model import model_main as mm, train_val as tv
    saved = (opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE, dict(opts.IMAGE_SIZES))
    opts.PER_REPLICA_BATCH = opts.BATCH_SIZE = 2
    opts.CONV_DTYPE = "bf16"
    opts.IMAGE_SIZES["kitti_raw"] = (64, 192)
    try:
        dataset, cfg, _ = mm.get_dataset("synthetic", "train", True)
        model, _, loss_object, optimizer = mm.create_training_parts(0, cfg, 1e-4, opts.LOSS_RIGID_T1, opts.SCALE_WEIGHT_T1,
                                                                   opts.RIGID_NET, ckpt_name="__test__")
        trainer, _ = tv.train_val_factory("graph", model, loss_object, 0, False, None, optimizer)
        for step in range(4):
            feats = dataset.batches[step % len(dataset.batches)]
            preds, loss, by_type = trainer.run_a_batch(feats)
            got = trainer.step_metrics(feats, preds, loss, by_type)
            ref = tv.merge_results(trainer._graph.static_in, preds, loss, by_type, False)
            assert set(got) == set(ref) and {"loss", "deprel", "trjabs", "trjrel", "roterr", "L1", "SSIM", "smoothe"} <= set(got)
            for k in ref:
                assert torch.allclose(got[k].float(), ref[k].float(), rtol=1e-6, atol=1e-7), (step, k, got[k], ref[k])
        frame, hours = trainer.run_an_epoch(dataset)
        assert len(frame) == dataset.steps and frame["loss"].notna().all()
    finally:
        opts.PER_REPLICA_BATCH, opts.BATCH_SIZE, opts.CONV_DTYPE = saved[:3]
        opts.IMAGE_SIZES.clear()
        opts.IMAGE_SIZES.update(saved[3])
