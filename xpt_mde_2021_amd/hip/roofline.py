"""Roofline pricing of the hand-written kernels with HIP events on the launch stream (bench.py's `roofline`).

ALGORITHMIC bytes (fp32; every input read once, every output written once; P = h*w target pixels, N source views;
unit = one warped pixel = target pixel x source view x scale):
  fused warp+L1+SSIM forward  (no synthesized image written): depth 4P + target 12P + sources 12NP = P (16 + 12 N)
  fused warp+L1+SSIM backward (views re-synthesized)        : the same reads + d_depth 4P         = P (20 + 12 N)
  per warped pixel at N = 4: 16 B forward, 17 B backward (33 B for the pair; the reference's op graph writes ~430 B).
The unfused kernels are priced too (warp_fwd P (4 + 24 N), photometric P (12 + 12 N)) for comparison.
"""
import torch
import torch.utils._python_dispatch


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


def measure_fused_ms(ops, feats, repeats, batch=None, nscales=4, generation=2):
    """(fwd ms, bwd ms, algorithmic fwd bytes, algorithmic bwd bytes, shape) of the MULTI-SCALE march launches (what the
    training step runs: all pyramid scales in one launch).  generation 2 = csrc/xpt_march.hip (forward launch; and the
    one-pass launch that leaves losses AND gradients -- the training step's only march launch -- priced with the backward's
    bytes), generation 1 = csrc/xpt_fused.hip (forward launch, backward launch)."""
    import ctypes
    from . import lib as _lib
    x = _inputs(feats, batch)
    lib = _lib.load()
    B, N, H, W = x["B"], x["N"], x["H"], x["W"]
    T = ops.pose_rvec2matr(x["pose"])
    srcs, depths, tgts, hs, ws_, scales = [], [], [], [], [], []
    for k in range(nscales):
        sc = 2 ** k
        h, w = H // sc, W // sc
        srcs.append(x["src"] if sc == 1 else ops.resize_down(x["src"].reshape(B * N, H, W, 3), sc).reshape(B, N, h, w, 3))
        tgts.append(x["tgt"] if sc == 1 else ops.resize_down(x["tgt"], sc))
        depths.append(x["depth"] if sc == 1 else x["depth"].reshape(B, H, W)[:, ::sc, ::sc].contiguous())
        hs.append(h), ws_.append(w), scales.append(float(sc))
    nws = sum(lib.xpt_photo_fused_workspace_floats(B, N, h, w) for h, w in zip(hs, ws_))
    ws = torch.empty(nws, device=T.device)
    g1 = torch.ones(B, device=T.device)
    ddepths = [torch.empty_like(d) for d in depths]
    dT = torch.empty_like(T)
    losses = torch.empty((2 * nscales, B), device=T.device)
    st = torch.cuda.current_stream().cuda_stream
    P = ctypes.c_void_p * nscales
    ptrs = lambda ts: P(*[t.data_ptr() for t in ts])  # noqa: E731
    ci, cf = (ctypes.c_int * nscales), (ctypes.c_float * nscales)
    a_src, a_dep, a_tgt, a_g, a_dd = ptrs(srcs), ptrs(depths), ptrs(tgts), ptrs([g1] * nscales), ptrs(ddepths)
    a_h, a_w, a_s = ci(*hs), ci(*ws_), cf(*scales)
    f_fwd = lib.xpt_photo_march_ms_fwd if generation == 2 else lib.xpt_photo_fused_ms_fwd

    def fwd():     # losses NULL: the march launch alone
        _lib.check(f_fwd(nscales, a_src, a_dep, T.data_ptr(), x["K"].data_ptr(), a_tgt, None, ws.data_ptr(), nws, B, N, a_h,
                         a_w, a_s, st), "march ms fwd")

    def bwd():     # march + its finishing launch (pose gradient; generation 2: the loss values as well)
        if generation == 2:
            _lib.check(lib.xpt_photo_march_ms_fwdbwd(nscales, a_src, a_dep, T.data_ptr(), x["K"].data_ptr(), a_tgt, a_g, a_g,
                                                     losses.data_ptr(), a_dd, dT.data_ptr(), ws.data_ptr(), nws, B, N, a_h,
                                                     a_w, a_s, st), "march ms fwdbwd")
        else:
            _lib.check(lib.xpt_photo_fused_ms_bwd(nscales, a_src, a_dep, T.data_ptr(), x["K"].data_ptr(), a_tgt, a_g, a_g,
                                                  a_dd, dT.data_ptr(), ws.data_ptr(), nws, B, N, a_h, a_w, a_s, st),
                       "fused ms bwd")

    pixels = sum(h * w for h, w in zip(hs, ws_))
    return (_time_kernel(fwd, repeats), _time_kernel(bwd, repeats), B * pixels * (16 + 12 * N), B * pixels * (20 + 12 * N),
            {"B": B, "N": N, "h": H, "w": W, "scales": nscales})


def _pmc_entry(kernel, shape):
    """The committed PMC record (bytes, vector-instruction count) of a kernel at a shape, or None."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles",
                        "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)[kernel]["{B}x{N}x{h}x{w}".format(**shape)]
    except (OSError, KeyError, ValueError):
        return None


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
    """bench.py's `roofline` object: the march launches of the training step (csrc/xpt_march.hip: view synthesis + L1 +
    SSIM of the four pyramid scales) at the step's own shape -- `frac` = the forward launch, `bwd` = the one-pass launch
    that leaves losses and gradients (the only march launch a training step runs) -- plus the same at a batch that no
    longer fits the 256 MiB Infinity Cache and at configs[3]'s 256 x 832, the first-generation kernels and the unfused
    kernels for comparison."""
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
    f1_ms, b1_ms, f_bytes, b_bytes, shape = measure_fused_ms(ops, feats, repeats, generation=1)
    extra["first generation fused_fwd_ms_kernel (4 scales)"] = (f1_ms, f_bytes)
    extra["first generation fused_bwd_ms_kernel<0> (4 scales, + its finisher)"] = (b1_ms, b_bytes)
    f_ms, b_ms, f_bytes, b_bytes, shape = measure_fused_ms(ops, feats, repeats)
    from ..utils import synthetic_data as sd

    def point(feats_, reps, batch=None):
        f, b, fb, bb, shp = measure_fused_ms(ops, feats_, reps, batch=batch)
        gf, gb = fb / (f * 1e-3) / 1e9, bb / (b * 1e-3) / 1e9
        return {"shape": shp,
                "fwd": {"launch_us": round(f * 1e3, 2), "GBps": round(gf, 2), "frac": round(gf / hbm_peak_gbs, 4)},
                "bwd": {"launch_us": round(b * 1e3, 2), "GBps": round(gb, 2), "frac": round(gb / hbm_peak_gbs, 4)}}

    # configs[3]'s shape (256 x 832) at batch 4 and at batch 32 (beyond the Infinity Cache); batch 128 at 128 x 416
    hfeats = {k: v.to(x["src"].device) for k, v in sd.make_features(4, 2 * H, 2 * W, feats["image5d"].shape[1], 7).items()}
    achieved = f_bytes / (f_ms * 1e-3) / 1e9
    achieved_b = b_bytes / (b_ms * 1e-3) / 1e9
    traffic = _pmc_traffic("march_fwd_ms_kernel", shape)

    def gbs(ms_, nbytes):
        return round(nbytes / (ms_ * 1e-3) / 1e9, 2)

    # TOP LEVEL = the launch the training step actually runs: the one-pass march (losses AND gradients of the four scales)
    # with its finisher; the forward-only launch (evaluation / the two-pass path) is reported beside it under "fwd"
    step_kernel = "march_bwd_ms_p_kernel<0>" if N == 4 else "march_bwd_ms_p_kernel<1>"
    pmc = _pmc_entry(step_kernel, shape)
    valu = None
    if pmc is not None and pmc.get("valu_wave_instructions"):
        # vector-pipe ceiling of the launch: PMC instruction count over the chip's SIMDs at the measured class mix
        # (tools/lab/valu_rates.hip: 1.95 cycles full-rate, 3.2 DPP / select / convert, 6.2 reciprocal; ~2.5 on this mix)
        issue_us = pmc["valu_wave_instructions"] / 1024.0 * 2.5 / 2.4e3
        valu = {"wave_instructions_per_launch": int(pmc["valu_wave_instructions"]), "per_wave": round(pmc["valu_wave_instructions"] / pmc["waves"], 1),
                "issue_time_us": round(issue_us, 2), "frac_of_issue_ceiling": round(issue_us / (b_ms * 1e3), 4),
                "how": "SQ_INSTS_VALU of the committed PMC pass (profiles/pmc_traffic.json) / 1,024 SIMDs x 2.5 cycles / 2.4 GHz "
                       "against the launch time measured here"}
    return {"bound": "hbm",
            "kernel": step_kernel + " + finisher: view synthesis + L1 + SSIM of the 4 pyramid scales, losses AND gradients in one "
                      "pass -- the training step's only march launch (csrc/xpt_march.hip)",
            "achieved": round(achieved_b, 2), "peak": hbm_peak_gbs, "unit": "GB/s", "frac": round(achieved_b / hbm_peak_gbs, 4),
            "traffic": None if pmc is None else pmc["bytes"], "launch_us": round(b_ms * 1e3, 3),
            "algorithmic_bytes_per_launch": int(b_bytes), "bytes_per_warped_pixel": round((20 + 12 * N) / N, 3), "shape": shape,
            "valu": valu,
            "fwd": {"kernel": "march_fwd_ms_kernel (forward only: evaluation and the two-pass path; not launched by the training step)",
                    "launch_us": round(f_ms * 1e3, 3), "GBps": round(achieved, 2), "frac": round(achieved / hbm_peak_gbs, 4),
                    "algorithmic_bytes_per_launch": int(f_bytes), "bytes_per_warped_pixel": round((16 + 12 * N) / N, 3),
                    "traffic": traffic},
            "bwd": {"kernel": step_kernel + " + its finisher (= the top-level entry)",
                    "launch_us": round(b_ms * 1e3, 3), "GBps": round(achieved_b, 2), "frac": round(achieved_b / hbm_peak_gbs, 4),
                    "algorithmic_bytes_per_launch": int(b_bytes), "bytes_per_warped_pixel": round((20 + 12 * N) / N, 3),
                    "traffic": None if pmc is None else pmc["bytes"]},
            "valu_note": "the one-pass launch is bound by vector-instruction issue, not by bytes: PMC counts ~5,500 vector "
                         "instructions per wave (~290 per pixel row of a wave) issued at ~2.5 cycles each on the measured class mix; "
                         "see `valu` and DESIGN.md section 5",
            "all": {k: {"launch_us": round(v[0] * 1e3, 3), "GBps": gbs(*v)} for k, v in extra.items()},
            "hires": point(hfeats, repeats),
            "hires_large_batch": point(hfeats, max(repeats // 5, 5), batch=32),
            "large_batch": point(feats, max(repeats // 5, 5), batch=large_batch)}


# ------------------------------------------------------------------------------- convolution arithmetic of one step
class _MacCounter(torch.utils._python_dispatch.TorchDispatchMode):
    """Counts multiply-accumulates of every aten convolution / matrix product executed under it, by kind:
    dense (k x k, groups 1), pointwise (1x1 convolutions and matrix products), depthwise (groups > 1)."""

    def __init__(self):
        super().__init__()
        self.macs = {"dense": 0, "pointwise": 0, "depthwise": 0}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._overloadpacket.__name__
        if name in ("convolution", "_convolution", "conv2d"):
            w = args[1]
            groups = args[8] if len(args) > 8 else 1
            macs = out.numel() * w.shape[1] * w.shape[2] * w.shape[3]
            kind = "depthwise" if groups > 1 else ("pointwise" if w.shape[2] * w.shape[3] == 1 else "dense")
            self.macs[kind] += int(macs)
        elif name in ("mm", "addmm", "bmm"):
            a, b = (args[1], args[2]) if name == "addmm" else (args[0], args[1])
            self.macs["pointwise"] += int(a.numel() * b.shape[-1])
        return out


def count_forward_macs(height, width, stereo=False):
    """Multiply-accumulates of ONE snippet's forward through DepthNet + PoseNet, counted by a dispatch hook on the CPU
    instance of the same modules (library ops there; the arithmetic is the same the gfx950 kernels perform)."""
    from ..config import opts
    from ..model.build_model.model_factory import ModelFactory
    from ..utils import synthetic_data as sd
    feats = sd.make_features(1, height, width, opts.SNIPPET_LEN, 1, stereo)
    saved = opts.CONV_DTYPE
    opts.CONV_DTYPE = "fp32"
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            model = ModelFactory(sd.tfr_config_for(feats), global_batch=1, net_names=opts.RIGID_NET).get_model()
        with torch.no_grad(), _MacCounter() as counter:
            model(feats)
    finally:
        opts.CONV_DTYPE = saved
    return dict(counter.macs)


def mfma_utilisation(height, width, batch, step_seconds, peak_tflops, stereo=False):
    """bench.py's `mfma` object: matrix-core-eligible convolution FLOPs of a training step (dense + pointwise, forward +
    data gradient + weight gradient = 3 x forward, 2 FLOP per MAC) over the measured step time, against the bf16 dense
    peak.  Depthwise convolutions are VALU work and are reported separately."""
    macs = count_forward_macs(height, width, stereo)
    eligible = macs["dense"] + macs["pointwise"]
    flops_step = 2.0 * 3.0 * eligible * batch
    achieved = flops_step / step_seconds / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": round(achieved / peak_tflops, 5),
            "gmac_forward_per_snippet": {k: round(v / 1e9, 4) for k, v in macs.items()},
            "flops_counted_per_step": flops_step,
            "how": "dispatch-hook MAC count of DepthNet+PoseNet forward (CPU instance of the same modules) x 3 (forward, data "
                   "gradient, weight gradient) x 2 FLOP x batch, divided by the timed step; the step is launch / latency "
                   "bound at batch 8 (about 1,100 kernels of a few microseconds), which is what this fraction shows"}
