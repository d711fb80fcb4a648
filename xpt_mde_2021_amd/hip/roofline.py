"""Roofline pricing of the hand-written kernels with HIP events on the launch stream (bench.py's `roofline`).

ALGORITHMIC bytes (fp32; inputs read once, outputs written once; P = h*w target pixels, N source views):
  warp_fwd           : depth 4P + sources 12NP + synth 12NP                     = P (4 + 24 N)   per batch element
  photo_fwd (L1/SSIM): synth 12NP + target 12P (+ nothing written but B floats) = P (12 + 12 N)
  warp+L1+SSIM fused : depth 4P + target 12P + sources 12NP [+ synth 12NP]      = P (16 + 12 N [+ 12 N])
The "unit" is one warped pixel (target pixel x source view x scale).
"""
import torch


def _time_kernel(fn, repeats, warmup=5):
    """Average duration (ms) of `fn` launched back-to-back `repeats` times on torch's current stream, which is the
    stream the ctypes launch uses (ops._stream()), bracketed by HIP events recorded on that same stream."""
    for _ in range(warmup):
        fn()
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(repeats):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / repeats


def measure(ops, feats, repeats, hbm_peak_gbs):
    """feats: a feature dict resident on the GPU (image5d [B,5,H,W,3], intrinsic [B,3,3])."""
    from ..utils import synthetic_data as sd
    image5d = feats["image5d"]
    B, S, H, W, _ = image5d.shape
    N = S - 1
    dev = image5d.device
    g = torch.Generator().manual_seed(5)
    src = image5d[:, :-1].contiguous()
    tgt = image5d[:, -1].contiguous()
    depth = sd.smooth_depth(B, H, W, g).to(dev)
    T = ops.pose_rvec2matr(sd.random_poses(B, N, g).to(dev))
    K = feats["intrinsic"]
    P = H * W
    kernels = {}

    synth = ops.warp(src, depth, T, K, 1)
    ms = _time_kernel(lambda: ops.warp(src, depth, T, K, 1), repeats)
    kernels["warp_fwd_kernel"] = (ms, B * P * (4 + 24 * N))
    ms = _time_kernel(lambda: ops.photometric("SSIM", synth, tgt, True), repeats)
    kernels["photo_fwd_kernel<SSIM>"] = (ms, B * P * (12 + 12 * N))

    name = max(kernels, key=lambda k: kernels[k][0])
    ms, nbytes = kernels[name]
    achieved = nbytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": hbm_peak_gbs, "unit": "GB/s",
            "frac": round(achieved / hbm_peak_gbs, 4), "traffic": None,
            "launch_us": round(ms * 1e3, 3), "algorithmic_bytes_per_launch": int(nbytes),
            "shape": {"B": B, "N": N, "h": H, "w": W},
            "all": {k: {"launch_us": round(v[0] * 1e3, 3), "GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 2)}
                    for k, v in kernels.items()}}
