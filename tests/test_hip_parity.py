"""GPU parity: every entry point of libxpt_hip.so (through the ctypes C ABI) against the oracle.

Tolerance: 1e-4 absolute in fp32 (BASELINE.json north_star) on values in [-1,1]; gradients 1e-3 of their scale with
ZERO outliers.  The synthesis tests are flip-aware (tests/util.py): pixels whose projection lies within fp32 rounding of
an integer coordinate or of the validity border -- predicted from the fp64 oracle -- are invalidated on both sides
(depth 0), so that no allowance for floor() / validity flips is needed."""
import numpy as np
import pytest
import torch

from xpt_mde_2021_amd.hip.lib import half as _half_dtype

HALF = _half_dtype()      # 16-bit activation dtype of this process: bf16, or fp16 under XPT_HALF=fp16 (tests/test_fp16_build_gpu.py)

from oracle import ref_loss, ref_pose, ref_synthesize as rs
from tests.util import flip_safe_depth, frac_close
from xpt_mde_2021_amd.utils import synthetic_data as sd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(gpu_device):
    from xpt_mde_2021_amd.hip import ops as _ops
    return _ops


def gen(seed):
    return torch.Generator().manual_seed(seed)


def warp_inputs(B, N, h, w, seed, scale=1):
    g = gen(seed)
    src = torch.stack([sd.smooth_noise((B, h, w, 3), g) for _ in range(N)], dim=1).contiguous()
    depth = sd.smooth_depth(B, h, w, g)
    K = sd.kitti_like_intrinsic(B, h * scale, w * scale)
    pose = sd.random_poses(B, N, g)
    return src, depth, K, pose


# ------------------------------------------------------------------------------------------------ K0
def test_pose_fwd_bwd(ops, gpu_device):
    g = gen(11)
    pose = torch.rand((8, 4, 6), generator=g) * 2 - 1
    pose[0, 0, 3:] = 0.0            # exactly-zero rotation: where(|t|<1e-8, I, .) branch
    pose[0, 1, 3:] = torch.tensor([0., 0., np.pi / 3])
    dT = torch.randn((8, 4, 4, 4), generator=g)
    p_ref = pose[:, 1:].clone().double().requires_grad_(True)     # the zero-rotation row is NaN in the reference grad
    T_ref = ref_pose.pose_rvec2matr_batch(p_ref)
    T_ref.backward(dT[:, 1:].double())
    p = pose.to(gpu_device).requires_grad_(True)
    T = ops.pose_rvec2matr(p)
    T.backward(dT.to(gpu_device))
    frac_close(T[:, 1:], T_ref, 1e-5, what="pose T")
    assert torch.equal(T[0, 0, :3, :3].detach().cpu(), torch.eye(3))
    assert torch.equal(T[0, 0, :3, 3].detach().cpu(), pose[0, 0, :3])
    c, s = np.cos(np.pi / 3), np.sin(np.pi / 3)
    assert np.allclose(T[0, 1, :3, :3].detach().cpu().numpy(), [[c, s, 0], [-s, c, 0], [0, 0, 1]], atol=1e-6)
    frac_close(p.grad[:, 1:], p_ref.grad, 1e-4, rtol=1e-4, what="dpose")
    assert torch.isfinite(p.grad).all()


# ------------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("scale", [2, 4, 8])
def test_resize_down(ops, gpu_device, scale):
    g = gen(12)
    img = torch.rand((3, 32, 48, 3), generator=g) * 2 - 1
    out = ops.resize_down(img.to(gpu_device), scale)
    ref = rs.tf_resize_bilinear(img, (32 // scale, 48 // scale))
    frac_close(out, ref, 1e-6, what=f"resize/{scale}")


@pytest.mark.parametrize("shape,scales", [((2, 5, 32, 48, 3), [1, 2, 4, 8]), ((3, 3, 16, 24, 3), [2, 4]),
                                          ((1, 2, 6, 10, 3), [1, 2])])
def test_image_pyramids_equal_resize_down_of_the_dense_slices(ops, gpu_device, shape, scales):
    """xpt_image_pyramids (sources + target, every scale, one launch) == resize_down of image5d[:, :-1] / [:, -1],
    bit for bit; scale 1 (the dense copies) is always part of the answer."""
    g = gen(13)
    img = (torch.rand(shape, generator=g) * 2 - 1).to(gpu_device)
    B, S, H, W, _ = shape
    sources, targets = ops.image_pyramids(img, scales)
    assert sorted(sources) == sorted(set(scales) | {1})
    src_dense = img[:, :-1].contiguous()
    tgt_dense = img[:, -1].contiguous()
    for s in sources:
        es = ops.resize_down(src_dense.reshape(B * (S - 1), H, W, 3), s).reshape(B, S - 1, H // s, W // s, 3)
        et = ops.resize_down(tgt_dense, s)
        assert torch.equal(sources[s], es), s
        assert torch.equal(targets[s], et), s


# ------------------------------------------------------------------------------------------------ K2+K3
@pytest.mark.parametrize("B,N,h,w,scale", [(4, 4, 128, 416, 1), (2, 4, 32, 104, 4), (2, 1, 16, 52, 8), (1, 3, 5, 7, 1)])
def test_warp_fwd_bwd(ops, gpu_device, B, N, h, w, scale):
    src, depth, K, pose = warp_inputs(B, N, h, w, 100 + h, scale)
    depth[0, 1:3, 2:5] = 0.0                         # invalid depth -> masked pixels
    pose[0, 0, 0] = 40.0                             # a view that leaves the image almost entirely
    depth, masked = flip_safe_depth(depth, ref_pose.pose_rvec2matr_batch(pose.double()), K, scale)
    assert masked < 0.1
    g = gen(7)
    dsynth = torch.randn((B, N, h, w, 3), generator=g)
    # oracle (fp64 keeps the reference gradient free of its own fp32 noise)
    d_ref = depth.clone().double().requires_grad_(True)
    T_ref = ref_pose.pose_rvec2matr_batch(pose.double()).requires_grad_(True)
    K_sc = rs.scale_intrinsic(K.double(), scale)
    coords = rs.warp_pixel_coords(d_ref, T_ref, K_sc, h, w)
    synth_ref = rs.bilinear_interpolation(src.double(), coords, d_ref)
    synth_ref.backward(dsynth.double())
    # fp32 oracle for the forward value (the bar is the reference's fp32 CPU path)
    synth_ref32 = rs.bilinear_interpolation(src, rs.warp_pixel_coords(depth, ref_pose.pose_rvec2matr_batch(pose),
                                                                      rs.scale_intrinsic(K, scale), h, w), depth)
    d = depth.to(gpu_device).requires_grad_(True)
    T = ref_pose.pose_rvec2matr_batch(pose).to(gpu_device).requires_grad_(True)
    synth = ops.warp(src.to(gpu_device), d, T, K.to(gpu_device), scale)
    synth.backward(dsynth.to(gpu_device))
    frac_close(synth, synth_ref, 1e-4, max_bad_frac=0.0, what="synth vs fp64 oracle")
    frac_close(synth, synth_ref32, 2e-4, max_bad_frac=0.0, what="synth vs fp32 oracle")     # two fp32 chains: each 1e-4 from fp64
    gs = d_ref.grad.abs().max().item()
    frac_close(d.grad, d_ref.grad, 1e-3 * gs, max_bad_frac=0.0, what="ddepth")
    ts = T_ref.grad.abs().max().item()
    frac_close(T.grad, T_ref.grad, 1e-3 * ts, max_bad_frac=0.0, what="dT")
    assert torch.all(synth[0, :, 1:3, 2:5] == 0)


def test_warp_bwd_planar_image_tight(ops, gpu_device):
    # On an image that is affine in (u,v) the sampler's coordinate gradient is the same in every cell, so
    # floor() flips cannot change it: the pose / depth gradients must then match the fp64 oracle tightly.
    B, N, h, w = 2, 4, 64, 208
    _, depth, K, pose = warp_inputs(B, N, h, w, 77, 2)
    vv, uu = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    plane = torch.stack([0.003 * uu - 0.002 * vv, 0.001 * uu + 0.004 * vv - 0.3, -0.002 * uu + 0.1], dim=-1)
    src = plane.reshape(1, 1, h, w, 3).repeat(B, N, 1, 1, 1).contiguous()
    g = gen(8)
    dsynth = torch.randn((B, N, h, w, 3), generator=g)
    d_ref = depth.clone().double().requires_grad_(True)
    T_ref = ref_pose.pose_rvec2matr_batch(pose.double()).requires_grad_(True)
    coords = rs.warp_pixel_coords(d_ref, T_ref, rs.scale_intrinsic(K.double(), 2), h, w)
    rs.bilinear_interpolation(src.double(), coords, d_ref).backward(dsynth.double())
    d = depth.to(gpu_device).requires_grad_(True)
    T = ref_pose.pose_rvec2matr_batch(pose).to(gpu_device).requires_grad_(True)
    ops.warp(src.to(gpu_device), d, T, K.to(gpu_device), 2).backward(dsynth.to(gpu_device))
    frac_close(d.grad, d_ref.grad, 1e-4 * d_ref.grad.abs().max().item(), rtol=1e-3, max_bad_frac=1e-4, what="ddepth planar")
    frac_close(T.grad, T_ref.grad, 2e-4 * T_ref.grad.abs().max().item(), rtol=1e-3, what="dT planar")


def test_synthesis_known_answer_on_gpu(ops, gpu_device):
    # the reference's test_reconstruct_bilinear_interp (test_synthesizing.py:258-301) through the HIP sampler
    batch, numsrc, height, width = 8, 4, 5, 5
    pc = np.meshgrid(np.arange(0, height), np.arange(0, width))
    pc = np.stack(pc, axis=0).reshape((1, 1, 2, 5, 5)).astype(np.float32)
    pc[0, 0, 0] += 1.3
    pc = np.tile(pc, (batch, numsrc, 1, 1, 1)).reshape((batch, numsrc, 2, height * width))
    image = np.meshgrid(np.arange(0, height), np.arange(0, width))[0].reshape((1, 1, height, width, 1))
    image = np.tile(image, (batch, numsrc, 1, 1, 3)).astype(np.float32)
    depth = np.ones((batch, height, width, 1), dtype=np.float32)
    recon = ops.bilinear_sample(torch.tensor(image).to(gpu_device), torch.tensor(pc).to(gpu_device),
                                torch.tensor(depth).to(gpu_device))
    mask = np.zeros((batch, numsrc, height, width, 1))
    mask[:, :, :4, :3] = 1
    assert np.allclose(recon.cpu().numpy(), (image + 1.3) * mask, atol=1e-5)


@pytest.mark.parametrize("C,ncoord,use_mask", [(3, 3, True), (2, 2, False), (8, 2, True)])
def test_bilinear_sampler_fwd_bwd(ops, gpu_device, C, ncoord, use_mask):
    g = gen(21)
    B, N, h, w = 2, 3, 12, 20
    image = torch.rand((B, N, h, w, C), generator=g)
    coords = torch.rand((B, N, ncoord, h * w), generator=g)
    coords[:, :, 0] = coords[:, :, 0] * (w + 4) - 2
    coords[:, :, 1] = coords[:, :, 1] * (h + 4) - 2
    vm = (torch.rand((B, h, w, 1), generator=g) > 0.2).float() if use_mask else None
    dout = torch.randn((B, N, h, w, C), generator=g)
    c_ref = coords.clone().double().requires_grad_(True)
    out_ref = rs.bilinear_interpolation(image.double(), c_ref, None if vm is None else vm.double())
    out_ref.backward(dout.double())
    c = coords.to(gpu_device).requires_grad_(True)
    out = ops.bilinear_sample(image.to(gpu_device), c, None if vm is None else vm.to(gpu_device))
    out.backward(dout.to(gpu_device))
    frac_close(out, out_ref, 1e-5, what="bilinear out")
    frac_close(c.grad, c_ref.grad, 1e-4, rtol=1e-4, what="dcoords")


# ------------------------------------------------------------------------------------------------ K4/K5
def photo_inputs(B, N, h, w, seed):
    g = gen(seed)
    tgt = sd.smooth_noise((B, h, w, 3), g)
    synth = torch.stack([(tgt + 0.1 * sd.smooth_noise((B, h, w, 3), g)).clamp(-1, 1) for _ in range(N)], dim=1)
    synth[:, 0, : h // 4] = 0.0                       # black (invalid) band -> gray == 0 mask
    synth[:, -1, :, w - 3:] = 0.0
    return synth.contiguous(), tgt.contiguous(), g


@pytest.mark.parametrize("method", ["L1", "L2", "SSIM"])
@pytest.mark.parametrize("B,N,h,w", [(4, 4, 128, 416), (2, 1, 16, 52), (1, 2, 3, 5)])
def test_photometric_fwd_bwd(ops, gpu_device, method, B, N, h, w):
    synth, tgt, g = photo_inputs(B, N, h, w, 31 + h)
    fn = ref_loss.PHOTOMETRIC[method]
    for reduce in (True, False):
        s_ref = synth.clone().double().requires_grad_(True)
        out_ref = fn(s_ref, tgt.double(), reduce)
        out_ref32 = fn(synth, tgt, reduce)
        gout = torch.randn(out_ref.shape, generator=g)
        out_ref.backward(gout.double())
        s = synth.to(gpu_device).requires_grad_(True)
        out = ops.photometric(method, s, tgt.to(gpu_device), reduce)
        out.backward(gout.to(gpu_device))
        # per-pixel SSIM in fp32 is conditioned by c2 = 9e-4 in flat regions: compare against fp64 with 2e-4
        atol = 1e-5 if reduce else (2e-4 if method == "SSIM" else 1e-6)
        frac_close(out, out_ref, atol, what=f"{method} reduce={reduce} vs fp64")
        frac_close(out, out_ref32, max(atol, 1e-5) * (3 if method == "SSIM" and not reduce else 1),
                   what=f"{method} reduce={reduce} vs fp32")
        gs = s_ref.grad.abs().max().item()
        # SSIM: pixels whose (1-ssim)/2 sits on the clip boundary (synth ~ target) toggle their gradient
        frac_close(s.grad, s_ref.grad, 2e-4 * gs, rtol=2e-3, max_bad_frac=1e-4 if method == "SSIM" else 1e-5,
                   what=f"d{method} reduce={reduce}")


# ------------------------------------------------------------------------------------------------ K2-K5 fused march
@pytest.mark.parametrize("B,N,h,w,scale", [(4, 4, 128, 416, 1), (2, 4, 64, 208, 2), (2, 4, 16, 52, 8), (2, 1, 32, 104, 4),
                                           (1, 3, 9, 70, 1), (1, 4, 3, 5, 1), (1, 2, 37, 130, 1)])
def test_fused_warp_l1_ssim_fwd_bwd(ops, gpu_device, B, N, h, w, scale):
    """xpt_photo_fused_{fwd,bwd} == photometric_loss_l1 + photometric_loss_ssim of the synthesized views (oracle),
    values, synthesized images and gradients w.r.t. depth and pose matrices."""
    src, depth, K, pose = warp_inputs(B, N, h, w, 300 + h, scale)
    g = gen(9)
    tgt = (src[:, 0] * 0.7 + 0.3 * sd.smooth_noise((B, h, w, 3), g)).clamp(-1, 1).contiguous()
    if h > 4:
        depth[0, 1:3, 2:5] = 0.0                      # invalid depth -> black pixels inside the image
    pose[0, -1, 0] = 30.0                             # one view almost completely out of the image
    depth, masked = flip_safe_depth(depth, ref_pose.pose_rvec2matr_batch(pose.double()), K, scale)
    assert masked < 0.15
    gl1, gss = torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5
    d_ref = depth.clone().double().requires_grad_(True)
    T_ref = ref_pose.pose_rvec2matr_batch(pose.double()).requires_grad_(True)
    coords = rs.warp_pixel_coords(d_ref, T_ref, rs.scale_intrinsic(K.double(), scale), h, w)
    synth_ref = rs.bilinear_interpolation(src.double(), coords, d_ref)
    l1_ref = ref_loss.photometric_loss_l1(synth_ref, tgt.double())
    ss_ref = ref_loss.photometric_loss_ssim(synth_ref, tgt.double())
    ((l1_ref * gl1.double()).sum() + (ss_ref * gss.double()).sum()).backward()

    d = depth.to(gpu_device).requires_grad_(True)
    T = ref_pose.pose_rvec2matr_batch(pose).to(gpu_device).requires_grad_(True)
    dev = [t.to(gpu_device) for t in (src, K, tgt)]
    l1, ss = ops.photo_fused(dev[0], d, T, dev[1], dev[2], scale)
    ((l1 * gl1.to(gpu_device)).sum() + (ss * gss.to(gpu_device)).sum()).backward()
    frac_close(l1, l1_ref, 2e-5, rtol=1e-4, what="fused L1")
    frac_close(ss, ss_ref, 2e-5, rtol=1e-4, what="fused SSIM")
    l1b, ssb, synth = ops.photo_fused_with_synth(dev[0], d, T, dev[1], dev[2], scale)
    # (separately compiled instantiations of the same expressions: the L1 sums are bit-identical, the SSIM sums agree to
    # the compiler's FMA contraction, ~1e-7 relative)
    assert torch.equal(l1b, l1) and torch.allclose(ssb, ss, rtol=2e-6, atol=0)
    frac_close(synth, synth_ref, 1e-4, max_bad_frac=0.0, what="fused synth")
    # the compiler-scheduled and the hand-pipelined row loop are the same arithmetic in the same order
    from xpt_mde_2021_amd.hip import lib as hip_lib
    lib = hip_lib.load()
    try:
        for variant in (0, 1, 2):                   # 2 = neighbour texels staged in LDS per block of rows
            assert lib.xpt_photo_fused_variant(variant) == 0
            l1v, ssv = ops.photo_fused(dev[0], d.detach(), T.detach(), dev[1], dev[2], scale)
            assert torch.equal(l1v, l1) and torch.allclose(ssv, ss, rtol=2e-6, atol=0), f"forward variant {variant}"
    finally:
        lib.xpt_photo_fused_variant(1)
    assert lib.xpt_photo_fused_variant(3) != 0
    gs = d_ref.grad.abs().max().item()
    frac_close(d.grad, d_ref.grad, 1e-3 * gs, max_bad_frac=0.0, what="fused ddepth")
    ts = T_ref.grad.abs().max().item()
    frac_close(T.grad, T_ref.grad, 1e-3 * ts, max_bad_frac=0.0, what="fused dT")
    # and against the unfused HIP path (same arithmetic, different kernels)
    d2 = depth.to(gpu_device).requires_grad_(True)
    T2 = ref_pose.pose_rvec2matr_batch(pose).to(gpu_device).requires_grad_(True)
    synth2 = ops.warp(dev[0], d2, T2, dev[1], scale)
    l1u, ssu = ops.photometric("L1", synth2, dev[2]), ops.photometric("SSIM", synth2, dev[2])
    ((l1u * gl1.to(gpu_device)).sum() + (ssu * gss.to(gpu_device)).sum()).backward()
    frac_close(l1, l1u, 1e-6, rtol=1e-5, what="fused vs unfused L1")
    frac_close(ss, ssu, 1e-6, rtol=1e-5, what="fused vs unfused SSIM")
    frac_close(d.grad, d2.grad, 1e-4 * gs, max_bad_frac=0.0, what="fused vs unfused ddepth")
    frac_close(T.grad, T2.grad, 1e-4 * ts, rtol=1e-3, what="fused vs unfused dT")


def test_fused_planar_image_gradients_tight(ops, gpu_device):
    """The fused march on a source image that is affine in (u, v), against a target offset far enough that no L1 sign
    flips: the sampler's coordinate gradient is then the same in every cell (floor() flips cannot change it), so the
    depth / pose gradients of L1 + SSIM must match the fp64 oracle tightly -- the fused counterpart of
    test_warp_bwd_planar_image_tight."""
    B, N, h, w = 2, 4, 64, 208
    _, depth, K, pose = warp_inputs(B, N, h, w, 78, 2)
    vv, uu = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    plane = torch.stack([0.003 * uu - 0.002 * vv, 0.001 * uu + 0.004 * vv - 0.3, -0.002 * uu + 0.1], dim=-1)
    src = plane.reshape(1, 1, h, w, 3).repeat(B, N, 1, 1, 1).contiguous()
    g = gen(18)
    tgt = (plane.reshape(1, h, w, 3).repeat(B, 1, 1, 1) + 0.35 + 0.02 * sd.smooth_noise((B, h, w, 3), g)).contiguous()
    gl1, gss = torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5
    d_ref = depth.clone().double().requires_grad_(True)
    T_ref = ref_pose.pose_rvec2matr_batch(pose.double()).requires_grad_(True)
    coords = rs.warp_pixel_coords(d_ref, T_ref, rs.scale_intrinsic(K.double(), 2), h, w)
    synth_ref = rs.bilinear_interpolation(src.double(), coords, d_ref)
    l1_ref = ref_loss.photometric_loss_l1(synth_ref, tgt.double())
    ss_ref = ref_loss.photometric_loss_ssim(synth_ref, tgt.double())
    ((l1_ref * gl1.double()).sum() + (ss_ref * gss.double()).sum()).backward()
    d = depth.to(gpu_device).requires_grad_(True)
    T = ref_pose.pose_rvec2matr_batch(pose).to(gpu_device).requires_grad_(True)
    l1, ss = ops.photo_fused(src.to(gpu_device), d, T, K.to(gpu_device), tgt.to(gpu_device), 2)
    ((l1 * gl1.to(gpu_device)).sum() + (ss * gss.to(gpu_device)).sum()).backward()
    frac_close(l1, l1_ref, 1e-5, rtol=1e-4, what="fused L1 planar")
    frac_close(ss, ss_ref, 1e-5, rtol=1e-4, what="fused SSIM planar")
    frac_close(d.grad, d_ref.grad, 1e-4 * d_ref.grad.abs().max().item(), rtol=1e-3, max_bad_frac=2e-4, what="fused ddepth planar")
    frac_close(T.grad, T_ref.grad, 2e-4 * T_ref.grad.abs().max().item(), rtol=1e-3, what="fused dT planar")


@pytest.mark.parametrize("B,N,H,W,nscales", [(2, 4, 64, 208, 4), (3, 1, 32, 104, 3), (1, 4, 24, 70, 2)])
def test_fused_multi_scale_launch_equals_per_scale_calls(ops, gpu_device, B, N, H, W, nscales, monkeypatch):
    """xpt_photo_fused_ms_{fwd,bwd} (csrc/xpt_fused.hip, the first generation): every scale of the pyramid in one march
    launch runs the SAME device function per scale as the per-scale entry points -> losses and depth gradients
    bit-identical, pose gradient = the sum of the per-scale pose gradients (added in scale order by the finishing
    kernel: compared to rounding).  (The training path runs csrc/xpt_march.hip: tests/test_march_gpu.py.)"""
    monkeypatch.setattr(ops, "_MARCH_V1", True)
    g = gen(77 + H)
    srcs, depths, tgts, scales = [], [], [], []
    pose = None
    for k in range(nscales):
        sc = 2 ** k
        h, w = H // sc, W // sc
        src, depth, K, p = warp_inputs(B, N, h, w, 900 + k, sc)
        pose = p if pose is None else pose
        srcs.append(src.to(gpu_device))
        depths.append(depth.to(gpu_device))
        tgts.append((src[:, 0] * 0.6 + 0.4 * sd.smooth_noise((B, h, w, 3), g)).clamp(-1, 1).contiguous().to(gpu_device))
        scales.append(sc)
    Kd = K.to(gpu_device)
    weights = [(torch.rand(B, generator=g) + 0.5).to(gpu_device) for _ in range(2 * nscales)]

    def run(multi):
        ds = [d.clone().requires_grad_(True) for d in depths]
        T = ref_pose.pose_rvec2matr_batch(pose).to(gpu_device).requires_grad_(True)
        if multi:
            pairs = ops.photo_fused_multi_scale(srcs, ds, T, Kd, tgts, scales)
        else:
            pairs = [ops.photo_fused(srcs[k], ds[k], T, Kd, tgts[k], scales[k]) for k in range(nscales)]
        total = sum((l1 * weights[2 * k]).sum() + (ss * weights[2 * k + 1]).sum() for k, (l1, ss) in enumerate(pairs))
        total.backward()
        return pairs, ds, T

    pm, dm, Tm = run(True)
    ps, dsingle, Ts = run(False)
    for k in range(nscales):
        assert torch.equal(pm[k][0], ps[k][0]) and torch.equal(pm[k][1], ps[k][1]), f"losses of scale {k}"
        assert torch.equal(dm[k].grad, dsingle[k].grad), f"depth gradient of scale {k}"
    scale_t = Ts.grad.abs().max().item()
    assert (Tm.grad - Ts.grad).abs().max().item() <= 2e-6 * scale_t


# ------------------------------------------------------------------------------------------------ K6
@pytest.mark.parametrize("B,h,w", [(4, 128, 416), (2, 16, 52), (1, 2, 2)])
@pytest.mark.parametrize("is_depth", [False, True])
def test_smoothness_fwd_bwd(ops, gpu_device, B, h, w, is_depth):
    g = gen(41 + h)
    img = sd.smooth_noise((B, h, w, 3), g)
    depth = sd.smooth_depth(B, h, w, g, lo=0.99, hi=100.0)
    if is_depth:
        depth[0, 0, 0] = 1e-6                          # below the 1e-5 threshold of safe_reciprocal_number
    gl = torch.randn((B,), generator=g)
    d_ref = depth.clone().double().requires_grad_(True)
    disp_ref = ref_loss.safe_reciprocal_number(d_ref) if is_depth else d_ref
    loss_ref = ref_loss.smootheness_loss(disp_ref, img.double(), 4)
    loss_ref.backward(gl.double())
    d = depth.to(gpu_device).requires_grad_(True)
    loss = ops.smoothness(d, img.to(gpu_device), 4.0, is_depth)
    loss.backward(gl.to(gpu_device))
    frac_close(loss, loss_ref, 1e-6, rtol=1e-5, what="smooth loss")
    gs = d_ref.grad.abs().max().item()
    frac_close(d.grad, d_ref.grad, 1e-5 * gs, rtol=1e-4, max_bad_frac=1e-5, what="dsmooth")


@pytest.mark.parametrize("is_depth", [False, True])
def test_smoothness_multi_scale_equals_per_scale(ops, gpu_device, is_depth):
    """xpt_smooth_ms_* (all scales in one launch pair) == xpt_smooth_* scale by scale, bit for bit, with a scale
    whose gradient is absent (no zero-filled stand-in is needed upstream)."""
    g = gen(97)
    B = 3
    shapes = [(32, 104), (16, 52), (9, 27), (4, 13)]
    imgs = [sd.smooth_noise((B, h, w, 3), g).to(gpu_device) for h, w in shapes]
    depths = [sd.smooth_depth(B, h, w, g, lo=0.99, hi=100.0).to(gpu_device) for h, w in shapes]
    gls = [torch.randn((B,), generator=g).to(gpu_device) for _ in shapes]
    singles, grads = [], []
    for d, im, gl in zip(depths, imgs, gls):
        d = d.clone().requires_grad_(True)
        loss = ops.smoothness(d, im, 4.0, is_depth)
        loss.backward(gl)
        singles.append(loss.detach())
        grads.append(d.grad)
    ds = [d.clone().requires_grad_(True) for d in depths]
    multi = ops.smoothness_multi_scale(ds, imgs, 4.0, is_depth)
    for a, b in zip(multi, singles):
        assert torch.equal(a, b)
    torch.autograd.backward(multi, gls)
    for d, ge in zip(ds, grads):
        assert torch.equal(d.grad, ge)
    # one scale left out of the total: its depth gets an all-zero gradient, the others are unchanged
    ds = [d.clone().requires_grad_(True) for d in depths]
    multi = ops.smoothness_multi_scale(ds, imgs, 4.0, is_depth)
    (multi[0] * gls[0] + multi[2] * gls[2]).sum().backward()
    assert torch.equal(ds[0].grad, grads[0]) and torch.equal(ds[2].grad, grads[2])
    assert ds[1].grad.abs().max().item() == 0 and ds[3].grad.abs().max().item() == 0


# ------------------------------------------------------------------------------------------------ conv epilogues (a2/a3)
@pytest.mark.parametrize("C,shape", [(44, (2, 16, 26)), (1, (2, 32, 52)), (3, (1, 5, 7)), (130, (1, 9, 11))])
@pytest.mark.parametrize("dtype", [torch.float32, HALF])
def test_bias_act_and_batchnorm_epilogues(ops, gpu_device, C, shape, dtype):
    import torch.nn.functional as F
    g = gen(70 + C)
    B, H, W = shape
    x = torch.randn((B, C, H, W), generator=g)
    if dtype == HALF:
        x = x.to(HALF).float()
    gy = torch.randn((B, C, H, W), generator=g)
    if dtype == HALF:
        gy = gy.to(HALF).float()
    bias = torch.randn(C, generator=g) * 0.3
    gamma = torch.rand(C, generator=g) + 0.5
    mean = torch.randn(C, generator=g) * 0.2
    var = torch.rand(C, generator=g) + 0.3
    tol = 1e-5 if dtype == torch.float32 else 3e-2

    def dev(t):
        return t.to(gpu_device)

    for slope in (0.1, 1.0, 0.0):                       # LeakyReLU(0.1), linear, ReLU
        xr, br = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        yr = F.leaky_relu(xr + br.view(1, -1, 1, 1), slope) if slope != 1.0 else xr + br.view(1, -1, 1, 1)
        yr.backward(gy)
        xg = dev(x).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        bg = dev(bias).requires_grad_(True)
        y = ops.bias_act(xg, bg, slope)
        y.backward(dev(gy).to(dtype))
        frac_close(y.float(), yr, tol, rtol=tol, what=f"bias_act y slope={slope}")
        frac_close(xg.grad.float(), xr.grad, tol, rtol=tol, what=f"bias_act dx slope={slope}")
        frac_close(bg.grad, br.grad, 2e-3 * (1 + br.grad.abs().max().item()), rtol=2e-3, what=f"bias_act dbias slope={slope}")
    for relu_in in (False, True):
        xr = x.clone().requires_grad_(True)
        gr, br = gamma.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        yr = F.batch_norm(F.relu(xr) if relu_in else xr, mean, var, gr, br, False, 0.0, 1e-3)
        yr.backward(gy)
        xg = dev(x).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gg, bg = dev(gamma).requires_grad_(True), dev(bias).requires_grad_(True)
        y = ops.batchnorm_inference(xg, gg, bg, dev(mean), dev(var), 1e-3, relu_in)
        y.backward(dev(gy).to(dtype))
        frac_close(y.float(), yr, tol, rtol=tol, what=f"bn y relu_in={relu_in}")
        frac_close(xg.grad.float(), xr.grad, tol, rtol=tol, what=f"bn dx relu_in={relu_in}")
        frac_close(gg.grad, gr.grad, 2e-3 * (1 + gr.grad.abs().max().item()), rtol=2e-3, what=f"bn dgamma relu_in={relu_in}")
        frac_close(bg.grad, br.grad, 2e-3 * (1 + br.grad.abs().max().item()), rtol=2e-3, what=f"bn dbeta relu_in={relu_in}")


@pytest.mark.parametrize("C,shape", [(44, (2, 16, 52)), (88, (2, 8, 26)), (11, (1, 32, 104)), (32, (1, 9, 7))])
@pytest.mark.parametrize("dtype", [torch.float32, HALF])
def test_batchnorm_residual_and_sliced_gradient(ops, gpu_device, C, shape, dtype):
    """The cell's branch add fused into the BatchNorm epilogue, and a dy that is a channel slice of a wider
    (concatenated) gradient read in place through its row pitch."""
    import torch.nn.functional as F
    g = gen(170 + C)
    B, H, W = shape

    def rnd(*s):
        t = torch.randn(s, generator=g)
        return t.to(HALF).float() if dtype == HALF else t

    x, res = rnd(B, C, H, W), rnd(B, C, H, W)
    gy_wide = rnd(B, 3 * C + 8, H, W)
    off = C + 8
    gamma, bias = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    mean, var = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.3
    tol = 1e-5 if dtype == torch.float32 else 3e-2

    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yr = F.batch_norm(F.relu(xr), mean, var, gr, br, False, 0.0, 1e-3) + rr
    yr.backward(gy_wide[:, off:off + C])

    def dev(t):
        return t.to(gpu_device)

    xg = dev(x).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    rg = dev(res).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gg, bg = dev(gamma).requires_grad_(True), dev(bias).requires_grad_(True)
    y = ops.batchnorm_inference(xg, gg, bg, dev(mean), dev(var), 1e-3, True, residual=rg)
    wide = dev(gy_wide).to(dtype).contiguous(memory_format=torch.channels_last)
    y.backward(wide[:, off:off + C])                      # non-dense: row pitch 3C + 8
    frac_close(y.float(), yr, tol, rtol=tol, what="bn+res y")
    frac_close(xg.grad.float(), xr.grad, tol, rtol=tol, what="bn+res dx")
    frac_close(rg.grad.float(), rr.grad, tol, rtol=tol, what="bn+res dres")
    frac_close(gg.grad, gr.grad, 2e-3 * (1 + gr.grad.abs().max().item()), rtol=2e-3, what="bn+res dgamma")
    frac_close(bg.grad, br.grad, 2e-3 * (1 + br.grad.abs().max().item()), rtol=2e-3, what="bn+res dbeta")
    with pytest.raises(Exception):
        ops._AffineAct.apply(xg, None, bg, None, None, 0.0, 0.1, False, rg)      # residual needs a linear epilogue


# ------------------------------------------------------------------------------------------------ depthwise conv (a2)
DW_SMALL = (2, 44, 16, 26)          # scalar kernels (small maps)
DW_LARGE = (2, 40, 96, 280)         # >= 2^21 elements: the vectorised LDS-tap stencil (forward, stride-1 data gradient) and
                                    # the vectorised stride-2 data gradient
DW_LARGE_ODD = (1, 22, 301, 330)    # V = 2 vectors, odd extents


@pytest.mark.parametrize("k,stride,shape", [(3, 1, DW_SMALL), (5, 1, DW_SMALL), (7, 1, DW_SMALL), (3, 2, DW_SMALL),
                                            (5, 2, DW_SMALL), (7, 2, DW_SMALL), (3, 1, DW_LARGE), (5, 1, DW_LARGE),
                                            (7, 2, DW_LARGE), (3, 2, DW_LARGE), (5, 2, DW_LARGE_ODD), (7, 1, DW_LARGE_ODD)])
@pytest.mark.parametrize("dtype", [torch.float32, HALF])
@pytest.mark.parametrize("relu_in", [False, True])
def test_depthwise_conv_fwd_bwd(ops, gpu_device, k, stride, shape, dtype, relu_in):
    """xpt_dwconv_* against torch's fp32 grouped convolution on the CPU (TF SAME padding at stride 2)."""
    import torch.nn.functional as F
    from xpt_mde_2021_amd.model.model_util.layer_ops import same_pad
    g = gen(60 + k + stride)
    B, C, H, W = shape
    if stride == 2:
        (pt, pb), (pl, pr) = same_pad(H, k, 2), same_pad(W, k, 2)
    else:
        pt = pb = pl = pr = k // 2
    x = torch.randn((B, C, H, W), generator=g)
    w = torch.randn((C, 1, k, k), generator=g) * 0.2
    if dtype == HALF:
        x = x.to(HALF).float()                                   # same rounded inputs on both sides
    x_ref = x.clone().requires_grad_(True)
    w_ref = w.clone().requires_grad_(True)
    xin = F.relu(x_ref) if relu_in else x_ref
    y_ref = F.conv2d(F.pad(xin, (pl, pr, pt, pb)), w_ref, None, stride, 0, 1, C)
    gy = torch.randn(y_ref.shape, generator=g)
    if dtype == HALF:
        gy = gy.to(HALF).float()
    y_ref.backward(gy)
    xg = x.to(gpu_device, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = w.to(gpu_device).requires_grad_(True)
    y = ops.depthwise_conv2d(xg, wg, stride, (pt, pb, pl, pr), relu_in)
    assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(gy.to(gpu_device, dtype))
    tol = 1e-4 if dtype == torch.float32 else 3e-2                  # bf16 outputs carry 8 mantissa bits
    frac_close(y.float(), y_ref, tol, rtol=tol, what="dwconv y")
    frac_close(xg.grad.float(), x_ref.grad, tol, rtol=tol, what="dwconv dx")
    wscale = max(1.0, w_ref.grad.abs().max().item())
    frac_close(wg.grad, w_ref.grad, (1e-3 if dtype == torch.float32 else 2e-2) * wscale, rtol=1e-3, what="dwconv dw")


# ------------------------------------------------------------------------------- 1x1 conv weight gradient (split-K MFMA)
@pytest.mark.parametrize("M,cout,cin,pad_dy,pad_x", [
    (416, 1056, 1056, 0, 0),       # 4x13 maps, widest layer: 17x17 tiles, 3 splits
    (416, 176, 1056, 0, 0),
    (1664, 88, 528, 0, 0),
    (6656, 44, 264, 0, 220),       # x = channel slice of a 6-branch concat (row pitch 264+220)
    (6656, 264, 44, 88, 0),        # dy = channel slice
    (26624, 22, 44, 0, 0),
    (106496, 11, 32, 0, 0),        # stem level: one 32x32 tile, long reduction
    (106496, 32, 3, 0, 0),
    (417, 33, 65, 3, 5),           # ragged: odd row count, tiles with one valid row / column
    (1, 7, 5, 0, 0),
    (63, 64, 64, 0, 0),            # single split: direct write
])
def test_conv1x1_weight_grad(gpu_device, M, cout, cin, pad_dy, pad_x):
    from xpt_mde_2021_amd.hip import ops
    g = torch.Generator().manual_seed(M + cout)
    dy_full = torch.randn(M, cout + pad_dy, generator=g).to(gpu_device, HALF)
    x_full = torch.randn(M, cin + pad_x, generator=g).to(gpu_device, HALF)
    dy2 = dy_full[:, pad_dy // 2:pad_dy // 2 + cout]
    x2 = x_full[:, pad_x // 2:pad_x // 2 + cin]
    dw = ops.conv1x1_weight_grad(dy2, x2)
    ref = dy2.double().t() @ x2.double()
    assert dw.shape == (cout, cin) and dw.dtype == torch.float32
    # exact products, fp32 accumulation in a fixed order: error ~ sqrt(M) * 2^-24 * |terms|
    err = (dw.double() - ref).abs().max().item()
    assert err < 2e-6 * max(1.0, ref.abs().max().item()) * max(1.0, (M / 64) ** 0.5), err
    # deterministic, and the arrival counters are left reset: a second call gives the same bits
    dw2 = ops.conv1x1_weight_grad(dy2, x2)
    assert torch.equal(dw, dw2)
    assert int(ops._counters(dy2.device).abs().sum()) == 0


def test_as_rows_views(gpu_device):
    from xpt_mde_2021_amd.hip import ops
    t = torch.randn(2, 12, 3, 5, device=gpu_device).contiguous(memory_format=torch.channels_last)
    r = ops.as_rows(t)
    assert r.data_ptr() == t.data_ptr() and r.shape == (30, 12) and r.stride() == (12, 1)
    assert torch.equal(r, t.permute(0, 2, 3, 1).reshape(30, 12))
    s = t[:, 4:9]
    rs = ops.as_rows(s)
    assert rs.data_ptr() == s.data_ptr() and rs.stride() == (12, 1)
    assert torch.equal(rs, s.permute(0, 2, 3, 1).reshape(30, 5))
    n = torch.randn(2, 12, 3, 5, device=gpu_device)            # NCHW-contiguous: copied
    assert torch.equal(ops.as_rows(n), n.permute(0, 2, 3, 1).reshape(30, 12))
    one = torch.randn(4, 6, 1, 1, device=gpu_device)
    assert torch.equal(ops.as_rows(one), one.reshape(4, 6))


# ------------------------------------------------------------------------------- gradient fan-in
@pytest.mark.parametrize("dtype", [torch.float32, HALF])
def test_sum_rows_and_fan_out(ops, gpu_device, dtype):
    g = gen(300)
    B, C, H, W = 2, 44, 5, 7
    wide = torch.randn((B, 3 * C + 4, H, W), generator=g).to(gpu_device, dtype).contiguous(memory_format=torch.channels_last)
    parts = [torch.randn((B, C, H, W), generator=g).to(gpu_device, dtype).contiguous(memory_format=torch.channels_last)
             for _ in range(3)]
    parts.append(wide[:, 4:4 + C])                                   # channel slice: row pitch 3C + 4
    parts.append(torch.randn((B, C, H, W), generator=g).to(gpu_device, dtype))     # NCHW-contiguous operand
    out = ops.sum_rows(parts)
    ref = sum(p.float() for p in parts)
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(out.float(), ref, atol=tol * (1 + ref.abs().max().item()))
    # fan_out: same gradient as plain reuse of the tensor
    x = parts[0].detach().float().requires_grad_(True)
    w = [torch.randn((B, C, H, W), generator=g).to(gpu_device) for _ in range(4)]
    a, b, c, d = ops.fan_out(x, 4)
    (a * w[0] + torch.relu(b) * w[1] + c[:, :C] * w[2]).sum().backward()          # d unused: its gradient is None
    x2 = parts[0].detach().float().requires_grad_(True)
    (x2 * w[0] + torch.relu(x2) * w[1] + x2 * w[2]).sum().backward()
    assert torch.allclose(x.grad, x2.grad, atol=1e-5)
    assert ops.fan_out(parts[0], 3)[0] is parts[0]                   # no grad needed: plain aliases


# ------------------------------------------------------------------------------- 3x3 SAME average pooling
@pytest.mark.parametrize("dtype", [torch.float32, HALF])
@pytest.mark.parametrize("shape", [(2, 44, 6, 9), (1, 11, 1, 5), (2, 8, 3, 1)])
def test_avg_pool3_same(ops, gpu_device, dtype, shape):
    import torch.nn.functional as F
    g = gen(400 + shape[1])
    B, C, H, W = shape
    x = torch.randn(shape, generator=g)
    if dtype == HALF:
        x = x.to(HALF).float()
    gy_wide = torch.randn((B, 2 * C + 4, H, W), generator=g)
    if dtype == HALF:
        gy_wide = gy_wide.to(HALF).float()
    xr = x.clone().requires_grad_(True)
    yr = F.avg_pool2d(xr, 3, 1, 1, count_include_pad=False) * 2.0
    yr.backward(gy_wide[:, 4:4 + C])
    xg = x.to(gpu_device, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.avg_pool3_same(xg, 2.0)
    wide = gy_wide.to(gpu_device, dtype).contiguous(memory_format=torch.channels_last)
    y.backward(wide[:, 4:4 + C])                      # channel slice: read through its row pitch
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    frac_close(y.float(), yr, tol, rtol=tol, what="avgpool y")
    frac_close(xg.grad.float(), xr.grad, tol, rtol=tol, what="avgpool dx")


# ------------------------------------------------------------------------------- depth head activation
def test_inverse_sigmoid_depth(ops, gpu_device):
    """depth = safe_rcp(sigmoid(x) + 0.01), disp = safe_rcp(depth) and their joint backward vs the tensor-op chain
    (model_factory.py:134-138, util_funcs.py:157-160)."""
    from xpt_mde_2021_amd.utils import util_funcs as uf
    g = gen(500)
    x = torch.randn((2, 1, 9, 13), generator=g) * 4
    gd, gs = torch.randn(x.shape, generator=g), torch.randn(x.shape, generator=g)
    xr = x.clone().double().requires_grad_(True)
    depth_r = uf.safe_reciprocal_number(torch.sigmoid(xr) + 0.01)
    disp_r = uf.safe_reciprocal_number(depth_r)
    (depth_r * gd.double() + disp_r * gs.double()).sum().backward()
    xg = x.to(gpu_device).requires_grad_(True)
    depth, disp = ops.inverse_sigmoid_depth(xg)
    (depth * gd.to(gpu_device) + disp * gs.to(gpu_device)).sum().backward()
    assert torch.allclose(depth.cpu().double(), depth_r.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(disp.cpu().double(), disp_r.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(xg.grad.cpu().double(), xr.grad, rtol=1e-4, atol=1e-5)
    assert depth.min() > 0.99 and depth.max() < 100.0
    # only one of the two outputs used
    x2 = x.to(gpu_device).requires_grad_(True)
    ops.inverse_sigmoid_depth(x2)[1].sum().backward()
    x2r = x.clone().double().requires_grad_(True)
    uf.safe_reciprocal_number(uf.safe_reciprocal_number(torch.sigmoid(x2r) + 0.01)).sum().backward()
    assert torch.allclose(x2.grad.cpu().double(), x2r.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 1, 16, 52), (1, 1, 1, 1), (3, 1, 5, 7), (2, 3, 4, 6)])
@pytest.mark.parametrize("dtype", [torch.float32, HALF])
def test_upsample2x_matches_interpolate(ops, gpu_device, shape, dtype):
    """xpt_upsample2x_* (resize_image at an exact factor 2, layer_ops.py:43-50) vs F.interpolate(bilinear, half-pixel)
    forward and backward, with the gradient arriving as a channel slice of an NHWC tensor (read in place)."""
    import torch.nn.functional as F
    g = gen(502)
    B, C, h, w = shape
    x = torch.randn(shape, generator=g)
    xr = x.clone().double().requires_grad_(True)
    yr = F.interpolate(xr, size=(2 * h, 2 * w), mode="bilinear", align_corners=False)
    xg = x.to(gpu_device).requires_grad_(True)
    y = ops.upsample2x(xg, dtype)
    assert y.dtype == dtype and y.shape == yr.shape
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert torch.allclose(y.float().cpu().double(), yr.detach(), rtol=tol, atol=tol)
    # the consumer is a concatenation in NHWC: the gradient of this operand is a strided channel slice
    other = torch.randn((B, 5, 2 * h, 2 * w), generator=g).to(gpu_device).to(dtype).contiguous(memory_format=torch.channels_last)
    wgt = torch.randn((B, 5 + C, 2 * h, 2 * w), generator=g)
    cat = torch.cat([other, y.contiguous(memory_format=torch.channels_last)], dim=1)
    (cat.float() * wgt.to(gpu_device)).sum().backward()
    gy = wgt[:, 5:].to(dtype).double() if dtype == HALF else wgt[:, 5:].double()
    yr.backward(gy)
    assert torch.allclose(xg.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("dtype", [torch.float32, HALF])
def test_global_avg_pool(ops, gpu_device, dtype):
    """GlobalAveragePooling2D closing PoseNet (pose_net.py:45): value and gradient vs x.float().mean((2, 3))."""
    g = gen(503)
    x = torch.randn((3, 24, 2, 7), generator=g).to(dtype)
    gy = torch.randn((3, 24), generator=g)
    xr = x.double().requires_grad_(True)
    xr.mean(dim=(2, 3)).backward(gy.double())
    xg = x.to(gpu_device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.global_avg_pool(xg)
    y.backward(gy.to(gpu_device))
    assert y.dtype == torch.float32
    assert torch.allclose(y.cpu().double(), x.double().mean(dim=(2, 3)), rtol=1e-6, atol=1e-6)
    assert xg.grad.dtype == dtype
    assert torch.allclose(xg.grad.float().cpu().double(), xr.grad, rtol=1e-2 if dtype == HALF else 1e-6, atol=1e-7)


def test_merge_total_matches_the_tensor_op_chain(ops, gpu_device):
    """xpt_merge_total_* == dot(c, rowsum(stack(terms))) and A rowsum(terms), with every term's gradient = c[t] g."""
    g = gen(504)
    n, batch, types = 12, 8, 3
    terms = [torch.randn(batch, generator=g) for _ in range(n)]
    c = torch.rand(n, generator=g)
    a = torch.rand((types, n), generator=g)
    tr = [t.clone().double().requires_grad_(True) for t in terms]
    rs = torch.stack(tr).sum(dim=1)
    total_r = torch.dot(c.double(), rs)
    (total_r * 1.7).backward()
    tg = [t.to(gpu_device).requires_grad_(True) for t in terms]
    total, by_type = ops.merge_total(c.to(gpu_device), a.to(gpu_device), tg)
    (total * 1.7).backward()
    assert abs(total.item() - total_r.item()) < 1e-5
    assert torch.allclose(by_type.cpu().double(), a.double() @ rs.detach(), rtol=1e-5, atol=1e-5)
    for t, r in zip(tg, tr):
        assert torch.allclose(t.grad.cpu().double(), r.grad, rtol=1e-6, atol=1e-7)


def test_inverse_sigmoid_depth_multi_equals_per_scale(ops, gpu_device):
    """All prediction scales in one launch (xpt_depth_head_ms_*) == scale by scale, bit for bit, with outputs of some
    scales left unused (their gradient slots are NULL at the boundary)."""
    g = gen(501)
    xs = [(torch.randn(shape, generator=g) * 4).to(gpu_device) for shape in [(2, 1, 16, 52), (2, 1, 8, 26), (2, 1, 4, 13), (2, 1, 2, 7)]]
    gds = [torch.randn(x.shape, generator=g).to(gpu_device) for x in xs]
    gss = [torch.randn(x.shape, generator=g).to(gpu_device) for x in xs]
    use = [(True, True), (True, False), (False, True), (False, False)]
    singles = []
    for x, gd, gs, (ud, us) in zip(xs, gds, gss, use):
        x = x.clone().requires_grad_(True)
        depth, disp = ops.inverse_sigmoid_depth(x)
        total = (depth * gd).sum() * float(ud) + (disp * gs).sum() * float(us) if (ud and us) else \
            (depth * gd).sum() if ud else (disp * gs).sum() if us else None
        if total is not None:
            total.backward()
        singles.append((depth.detach(), disp.detach(), x.grad))
    xm = [x.clone().requires_grad_(True) for x in xs]
    depths, disps = ops.inverse_sigmoid_depth_multi(xm)
    total = 0
    for depth, disp, gd, gs, (ud, us) in zip(depths, disps, gds, gss, use):
        if ud:
            total = total + (depth * gd).sum()
        if us:
            total = total + (disp * gs).sum()
    total.backward()
    for x, depth, disp, (d1, s1, g1) in zip(xm, depths, disps, singles):
        assert torch.equal(depth, d1) and torch.equal(disp, s1)
        if g1 is None:
            assert x.grad.abs().max().item() == 0
        else:
            assert torch.equal(x.grad, g1)
