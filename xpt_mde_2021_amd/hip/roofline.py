"""Roofline pricing of the hand-written kernels with HIP events on the launch stream (bench.py's `roofline`).

ALGORITHMIC bytes (fp32; every input read once, every output written once; P = h*w target pixels, N source views;
unit = one warped pixel = target pixel x source view x scale):
  fused warp+L1+SSIM forward  (no synthesized image written): depth 4P + target 12P + sources 12NP = P (16 + 12 N)
  fused warp+L1+SSIM backward (views re-synthesized)        : the same reads + d_depth 4P         = P (20 + 12 N)
  per warped pixel at N = 4: 16 B forward, 17 B backward (33 B for the pair; the reference's op graph writes ~430 B).
The unfused kernels are priced too (warp_fwd P (4 + 24 N), photometric P (12 + 12 N)) for comparison.
"""
import torch


def _time_kernel(fn, repeats, warmup=5, rounds=3):
    """Average duration (ms) of `fn` launched back-to-back on torch's current stream -- the stream the ctypes launch
    uses (ops._stream()) -- bracketed by HIP events recorded on that same stream; best of `rounds` rounds (the first
    round after other work can include clock ramp-up / allocator effects)."""
    for _ in range(warmup):
        fn()
    best = float("inf")
    for _ in range(rounds):
        start = torch.cuda.Event(enable_timing=True)
        stop = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for _ in range(repeats):
            fn()
        stop.record()
        stop.synchronize()
        best = min(best, start.elapsed_time(stop) / repeats)
    return best


def _inputs(feats, batch=None):
    from ..utils import synthetic_data as sd
    image5d = feats["image5d"]
    K = feats["intrinsic"]
    if batch is not None and batch != image5d.shape[0]:
        reps = (batch + image5d.shape[0] - 1) // image5d.shape[0]
        image5d = image5d.repeat(reps, 1, 1, 1, 1)[:batch].contiguous()
        K = K.repeat(reps, 1, 1)[:batch].contiguous()
    B, S, H, W, _ = image5d.shape
    dev = image5d.device
    g = torch.Generator().manual_seed(5)
    return {"src": image5d[:, :-1].contiguous(), "tgt": image5d[:, -1].contiguous(),
            "depth": sd.smooth_depth(B, H, W, g).to(dev), "pose": sd.random_poses(B, S - 1, g).to(dev), "K": K,
            "B": B, "N": S - 1, "H": H, "W": W}


def measure_fused(ops, feats, repeats, batch=None):
    """(fwd ms, bwd ms, algorithmic fwd bytes, algorithmic bwd bytes, shape) of the fused kernels at full resolution."""
    from . import lib as _lib
    x = _inputs(feats, batch)
    lib = _lib.load()
    B, N, H, W = x["B"], x["N"], x["H"], x["W"]
    T = ops.pose_rvec2matr(x["pose"])
    l1 = torch.empty(B, device=T.device)
    ss = torch.empty(B, device=T.device)
    ddepth = torch.empty_like(x["depth"])
    dT = torch.empty_like(T)
    g1 = torch.ones(B, device=T.device)
    nws = lib.xpt_photo_fused_workspace_floats(B, N, H, W)
    ws = torch.empty(nws, device=T.device)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()  # noqa: E731

    def fwd():
        # loss pointers NULL: the main march kernel alone (the 5 us per-batch reduce launch is not part of it)
        _lib.check(lib.xpt_photo_fused_fwd(p(x["src"]), p(x["depth"]), p(T), p(x["K"]), p(x["tgt"]), None, None, None,
                                           p(ws), nws, B, N, H, W, 1.0, st), "fused fwd")

    def bwd():
        _lib.check(lib.xpt_photo_fused_bwd(p(x["src"]), p(x["depth"]), p(T), p(x["K"]), p(x["tgt"]), p(g1), p(g1),
                                           p(ddepth), p(dT), p(ws), nws, B, N, H, W, 1.0, st), "fused bwd")

    P = H * W
    return (_time_kernel(fwd, repeats), _time_kernel(bwd, repeats), B * P * (16 + 12 * N), B * P * (20 + 12 * N),
            {"B": B, "N": N, "h": H, "w": W})


def _pmc_traffic(kernel, shape):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json: separate
    FETCH_SIZE / WRITE_SIZE runs with the gfx950 x2 FETCH_SIZE correction); None when no pass exists for this shape."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles",
                        "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
        return table[kernel]["{B}x{N}x{h}x{w}".format(**shape)]["bytes"]
    except (OSError, KeyError, ValueError):
        return None


def measure(ops, feats, repeats, hbm_peak_gbs, large_batch=128):
    """bench.py's `roofline` object: the fused forward kernel at the step's own shape (dominant hand-written kernel of
    the loss path), plus the backward, the unfused kernels and a large-batch point that no longer fits the 256 MiB
    Infinity Cache (the in-step shape does: SURVEY 7 'roofline measurement honesty')."""
    x = _inputs(feats)
    B, N, H, W = x["B"], x["N"], x["H"], x["W"]
    P = H * W
    T = ops.pose_rvec2matr(x["pose"])
    extra = {}
    synth = ops.warp(x["src"], x["depth"], T, x["K"], 1)
    ms = _time_kernel(lambda: ops.warp(x["src"], x["depth"], T, x["K"], 1), repeats)
    extra["unfused warp_fwd_kernel"] = (ms, B * P * (4 + 24 * N))
    ms = _time_kernel(lambda: ops.photometric("SSIM", synth, x["tgt"], True), repeats)
    extra["unfused photo_fwd_kernel<SSIM>"] = (ms, B * P * (12 + 12 * N))
    f_ms, b_ms, f_bytes, b_bytes, shape = measure_fused(ops, feats, repeats)
    extra["fused_bwd_kernel"] = (b_ms, b_bytes)
    lf_ms, lb_ms, lf_bytes, lb_bytes, lshape = measure_fused(ops, feats, max(repeats // 5, 5), batch=large_batch)
    achieved = f_bytes / (f_ms * 1e-3) / 1e9
    traffic = _pmc_traffic("fused_fwd_kernel<false, true>", shape)

    def gbs(ms_, nbytes):
        return round(nbytes / (ms_ * 1e-3) / 1e9, 2)

    return {"bound": "hbm", "kernel": "fused_fwd_kernel<false, true> (warp + L1 + SSIM, scale 1; hand-pipelined row loop)",
            "achieved": round(achieved, 2), "peak": hbm_peak_gbs, "unit": "GB/s", "frac": round(achieved / hbm_peak_gbs, 4),
            "traffic": traffic, "launch_us": round(f_ms * 1e3, 3), "algorithmic_bytes_per_launch": int(f_bytes),
            "bytes_per_warped_pixel": round(f_bytes / (B * N * P), 3), "shape": shape,
            "all": {k: {"launch_us": round(v[0] * 1e3, 3), "GBps": gbs(*v)} for k, v in extra.items()},
            "large_batch": {"shape": lshape,
                            "fwd": {"launch_us": round(lf_ms * 1e3, 2), "GBps": gbs(lf_ms, lf_bytes),
                                    "frac": round(gbs(lf_ms, lf_bytes) / hbm_peak_gbs, 4)},
                            "bwd": {"launch_us": round(lb_ms * 1e3, 2), "GBps": gbs(lb_ms, lb_bytes),
                                    "frac": round(gbs(lb_ms, lb_bytes) / hbm_peak_gbs, 4)}}}
