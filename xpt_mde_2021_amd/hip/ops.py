"""torch.autograd ops over the gfx950 kernels of libxpt_hip.so.

Every op requires contiguous float32 CUDA(HIP) tensors and launches on torch's current stream
(hipGraph-capturable).  No CPU path exists: a CPU tensor or a missing library raises.
"""
import torch

from . import lib as _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.XptHipError(f"{name}: expected a CUDA/HIP tensor (the xpt HIP ops have no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.XptHipError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------- K0 pose
class _PoseRvec2Matr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose):
        lib = _lib.load()
        pose = _dev(pose, "pose")
        M = pose.numel() // 6
        T = torch.empty(pose.shape[:-1] + (4, 4), dtype=torch.float32, device=pose.device)
        _lib.check(lib.xpt_pose_rvec2matr_fwd(_ptr(pose), _ptr(T), M, _stream()), "xpt_pose_rvec2matr_fwd")
        ctx.save_for_backward(pose)
        return T

    @staticmethod
    def backward(ctx, dT):
        lib = _lib.load()
        (pose,) = ctx.saved_tensors
        dT = _dev(dT, "dT")
        dpose = torch.empty_like(pose)
        _lib.check(lib.xpt_pose_rvec2matr_bwd(_ptr(pose), _ptr(dT), _ptr(dpose), pose.numel() // 6, _stream()),
                   "xpt_pose_rvec2matr_bwd")
        return dpose


def pose_rvec2matr(pose):
    """[..., 6] twist (tx,ty,tz,u1,u2,u3) -> [..., 4, 4]  (utils/convert_pose.py:32-71)."""
    return _PoseRvec2Matr.apply(pose)


# ------------------------------------------------------------------------------- K1 pyramid
def resize_down(img, scale):
    """TF2 half-pixel bilinear resize of [M,H,W,C] by an exact integer factor (no gradient: the
    pyramids are built from input images only, synthesize_base.py:74-85, util_funcs.py:163-175)."""
    lib = _lib.load()
    img = _dev(img.detach(), "img")
    M, H, W, C = img.shape
    scale = int(scale)
    if scale == 1:
        return img
    out = torch.empty((M, H // scale, W // scale, C), dtype=torch.float32, device=img.device)
    _lib.check(lib.xpt_resize_down_fwd(_ptr(img), _ptr(out), M, H, W, C, scale, _stream()), "xpt_resize_down_fwd")
    return out


def image_pyramids(image5d, scales):
    """Source frames (all but the last) and target frame (the last) of image5d [B,S,H,W,3] at every factor in `scales`
    (1 = dense copy), in ONE launch: -> ({scale: [B,S-1,h,w,3]}, {scale: [B,h,w,3]}); no gradient.  Same values as
    resize_down of the dense slices."""
    import ctypes
    lib = _lib.load()
    img = _dev(image5d.detach(), "image5d")
    if not img.is_contiguous():
        raise _lib.XptHipError("image_pyramids: image5d must be contiguous")
    B, S, H, W, C = img.shape
    if C != 3 or S < 2:
        raise _lib.XptHipError(f"image_pyramids: expected [B, S>=2, H, W, 3], got {tuple(img.shape)}")
    scales = sorted({int(s) for s in scales} | {1})
    sources, targets, first, count, factor, outs = {}, {}, [], [], [], []
    for s in scales:
        sources[s] = torch.empty((B, S - 1, H // s, W // s, 3), dtype=torch.float32, device=img.device)
        targets[s] = torch.empty((B, H // s, W // s, 3), dtype=torch.float32, device=img.device)
        first += [0, S - 1]
        count += [S - 1, 1]
        factor += [s, s]
        outs += [sources[s], targets[s]]
    n = len(outs)
    _lib.check(lib.xpt_image_pyramids(_ptr(img), B, S, H, W, n, (ctypes.c_int * n)(*first), (ctypes.c_int * n)(*count),
                                      (ctypes.c_int * n)(*factor), (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs]),
                                      _stream()), "xpt_image_pyramids")
    return sources, targets


# ------------------------------------------------------------------------------- K2+K3 warp
class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, depth, T, K, scale):
        lib = _lib.load()
        src, depth, T, K = _dev(src, "src"), _dev(depth, "depth"), _dev(T, "T"), _dev(K, "K")
        B, N, h, w, C = src.shape
        if C != 3 or depth.numel() != B * h * w or T.numel() != B * N * 16 or K.numel() != B * 9:
            raise _lib.XptHipError(f"warp: inconsistent shapes src{tuple(src.shape)} depth{tuple(depth.shape)} "
                                   f"T{tuple(T.shape)} K{tuple(K.shape)}")
        synth = torch.empty_like(src)
        _lib.check(lib.xpt_warp_fwd(_ptr(src), _ptr(depth), _ptr(T), _ptr(K), _ptr(synth), B, N, h, w, float(scale),
                                    _stream()), "xpt_warp_fwd")
        ctx.save_for_backward(src, depth, T, K)
        ctx.scale = float(scale)
        return synth

    @staticmethod
    def backward(ctx, dsynth):
        lib = _lib.load()
        src, depth, T, K = ctx.saved_tensors
        dsynth = _dev(dsynth, "dsynth")
        B, N, h, w, _ = src.shape
        ddepth = torch.empty_like(depth)
        dT = torch.empty_like(T)
        nws = lib.xpt_warp_bwd_workspace_floats(B, N, h, w)
        ws = torch.empty(nws, dtype=torch.float32, device=src.device)
        _lib.check(lib.xpt_warp_bwd(_ptr(src), _ptr(depth), _ptr(T), _ptr(K), _ptr(dsynth), _ptr(ddepth), _ptr(dT),
                                    _ptr(ws), nws, B, N, h, w, ctx.scale, _stream()), "xpt_warp_bwd")
        return None, ddepth, dT, None, None


def warp(src, depth, T, K, scale):
    """src [B,N,h,w,3] (already at this scale), depth [B,h,w,1], T [B,N,4,4], K [B,3,3] unscaled
    -> synthesized target views [B,N,h,w,3]; differentiable w.r.t. depth and T."""
    return _Warp.apply(src, depth, T, K, scale)


# ------------------------------------------------------------------------------- K3 sampler
class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, coords, valid_mask):
        lib = _lib.load()
        image, coords = _dev(image, "image"), _dev(coords, "pixel_coords")
        vm = None if valid_mask is None else _dev(valid_mask, "valid_mask")
        B, N, h, w, C = image.shape
        ncoord = coords.shape[2]
        if coords.shape[0] != B or coords.shape[1] != N or coords.shape[3] != h * w:
            raise _lib.XptHipError(f"bilinear: coords {tuple(coords.shape)} do not match image {tuple(image.shape)}")
        if vm is not None and vm.numel() != B * h * w:
            raise _lib.XptHipError("bilinear: valid_mask must be [B,h,w,1]")
        out = torch.empty_like(image)
        _lib.check(lib.xpt_bilinear_fwd(_ptr(image), _ptr(coords), _ptr(vm), _ptr(out), B, N, h, w, C, ncoord, _stream()),
                   "xpt_bilinear_fwd")
        ctx.save_for_backward(image, coords, vm)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        image, coords, vm = ctx.saved_tensors
        dout = _dev(dout, "dout")
        B, N, h, w, C = image.shape
        dcoords = torch.empty_like(coords)
        _lib.check(lib.xpt_bilinear_bwd(_ptr(image), _ptr(coords), _ptr(vm), _ptr(dout), _ptr(dcoords), B, N, h, w, C,
                                        coords.shape[2], _stream()), "xpt_bilinear_bwd")
        return None, dcoords, None


def bilinear_sample(image, pixel_coords, valid_mask=None):
    """BilinearInterpolation.__call__ (bilinear_interp.py:7-32); gradient w.r.t. pixel_coords."""
    return _Bilinear.apply(image, pixel_coords, valid_mask)


# ------------------------------------------------------------------------------- K4/K5 photometric
class _Photo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, synth, target, method, reduce):
        lib = _lib.load()
        synth, target = _dev(synth, "synt_target"), _dev(target, "orig_target")
        B, N, h, w, C = synth.shape
        if C != 3 or tuple(target.shape) != (B, h, w, 3):
            raise _lib.XptHipError(f"photometric: synth {tuple(synth.shape)} vs target {tuple(target.shape)}")
        m = _lib.PHOTO_METHODS[method]
        if reduce:
            out = torch.empty((B,), dtype=torch.float32, device=synth.device)
            nws = B * N * ((h * w + 255) // 256)
            ws = torch.empty(nws, dtype=torch.float32, device=synth.device)
            _lib.check(lib.xpt_photo_fwd(m, _ptr(synth), _ptr(target), None, _ptr(out), _ptr(ws), nws, B, N, h, w,
                                         _stream()), "xpt_photo_fwd")
        else:
            out = torch.empty_like(synth)
            _lib.check(lib.xpt_photo_fwd(m, _ptr(synth), _ptr(target), _ptr(out), None, None, 0, B, N, h, w, _stream()),
                       "xpt_photo_fwd")
        ctx.save_for_backward(synth, target)
        ctx.method, ctx.reduce = m, reduce
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        synth, target = ctx.saved_tensors
        g = _dev(g, "grad")
        B, N, h, w, _ = synth.shape
        dsynth = torch.empty_like(synth)
        ws, nws = None, 0
        if ctx.method == _lib.XPT_PHOTO_SSIM:
            nws = lib.xpt_photo_workspace_floats(B, N, h, w)
            ws = torch.empty(nws, dtype=torch.float32, device=synth.device)
        gloss, gmap = (g, None) if ctx.reduce else (None, g)
        _lib.check(lib.xpt_photo_bwd(ctx.method, _ptr(synth), _ptr(target), _ptr(gloss), _ptr(gmap), _ptr(dsynth),
                                     _ptr(ws), nws, B, N, h, w, _stream()), "xpt_photo_bwd")
        return dsynth, None, None, None


def photometric(method, synt_target, orig_target, reduce=True):
    """photometric_loss_l1/_l2/_ssim (loss_util.py:6-96): [B] if reduce else [B,N,h,w,3]."""
    return _Photo.apply(synt_target, orig_target, method, bool(reduce))


# ------------------------------------------------------------------------------- K2-K5 fused march
class _PhotoFused(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, depth, T, K, target, scale):
        lib = _lib.load()
        src, depth, T, K, target = (_dev(src, "src"), _dev(depth, "depth"), _dev(T, "T"), _dev(K, "K"),
                                    _dev(target, "target"))
        B, N, h, w, C = src.shape
        if C != 3 or depth.numel() != B * h * w or T.numel() != B * N * 16 or K.numel() != B * 9 \
                or tuple(target.shape) != (B, h, w, 3):
            raise _lib.XptHipError(f"photo_fused: inconsistent shapes src{tuple(src.shape)} depth{tuple(depth.shape)} "
                                   f"T{tuple(T.shape)} K{tuple(K.shape)} target{tuple(target.shape)}")
        l1 = torch.empty((B,), dtype=torch.float32, device=src.device)
        ss = torch.empty((B,), dtype=torch.float32, device=src.device)
        nws = lib.xpt_photo_fused_workspace_floats(B, N, h, w)
        ws = torch.empty(nws, dtype=torch.float32, device=src.device)
        _lib.check(lib.xpt_photo_fused_fwd(_ptr(src), _ptr(depth), _ptr(T), _ptr(K), _ptr(target), None, _ptr(l1),
                                           _ptr(ss), _ptr(ws), nws, B, N, h, w, float(scale), _stream()),
                   "xpt_photo_fused_fwd")
        ctx.save_for_backward(src, depth, T, K, target)
        ctx.scale = float(scale)
        return l1, ss

    @staticmethod
    def backward(ctx, g_l1, g_ss):
        lib = _lib.load()
        src, depth, T, K, target = ctx.saved_tensors
        B, N, h, w, _ = src.shape
        g_l1 = _dev(g_l1, "g_l1") if g_l1 is not None else torch.zeros(B, device=src.device)
        g_ss = _dev(g_ss, "g_ssim") if g_ss is not None else torch.zeros(B, device=src.device)
        ddepth = torch.empty_like(depth)
        dT = torch.empty_like(T)
        nws = lib.xpt_photo_fused_workspace_floats(B, N, h, w)
        ws = torch.empty(nws, dtype=torch.float32, device=src.device)
        _lib.check(lib.xpt_photo_fused_bwd(_ptr(src), _ptr(depth), _ptr(T), _ptr(K), _ptr(target), _ptr(g_l1), _ptr(g_ss),
                                           _ptr(ddepth), _ptr(dT), _ptr(ws), nws, B, N, h, w, ctx.scale, _stream()),
                   "xpt_photo_fused_bwd")
        return None, ddepth, dT, None, None, None


def photo_fused(src, depth, T, K, target, scale):
    """Fused view synthesis + photometric L1 + SSIM of one scale (no synthesized image is written):
    src [B,N,h,w,3] at this scale, depth [B,h,w,1], T [B,N,4,4], K [B,3,3] unscaled, target [B,h,w,3]
    -> (loss_l1 [B], loss_ssim [B]) = (photometric_loss_l1, photometric_loss_ssim)(synth, target) of the reference;
    differentiable w.r.t. depth and T."""
    return _PhotoFused.apply(src, depth, T, K, target, scale)


_MARCH_V1 = __import__("os").environ.get("XPT_DEBUG_MARCH_V1", "0") == "1"      # A/B: the first-generation march kernels (xpt_fused.hip)
_ONE_PASS = __import__("os").environ.get("XPT_DEBUG_TWO_PASS_LOSS", "0") != "1"   # A/B: forward and backward march as two launches
_grad_constants = {}


def _constant_rows(values, batch, device):
    """[len(values), batch] float32 rows, row i filled with values[i]: the upstream gradients a loss module announces
    for its fused terms.  Built once per configuration (outside graph capture when the trainer warms up eagerly)."""
    key = (tuple(values), int(batch), str(device))
    t = _grad_constants.get(key)
    if t is None:
        t = torch.tensor(values, dtype=torch.float32).reshape(-1, 1).repeat(1, batch).to(device)
        _grad_constants[key] = t
    return t


class _PhotoFusedMS(torch.autograd.Function):
    """photo_fused for every scale of the pyramid in ONE march launch (+ one finishing launch) forward and backward;
    the pose gradient comes back already summed over the scales.
    args = (T, K, scales, grad_hint, src_0.., depth_0.., target_0..).

    grad_hint (2 n floats or None): the gradients the caller's loss will send back for (l1 of every scale, ssim of every
    scale) -- they are loss weights over the batch size (TotalLoss.__call__, losses.py:44-55), known before the forward
    runs.  With a hint the forward IS the backward: xpt_photo_march_ms_fwdbwd leaves losses, d_depth and dT in one pass
    (csrc/xpt_march.hip) and backward() hands the stored gradients out after checking the hint against what actually
    arrived (outside graph capture; a mismatch recomputes with the real gradients)."""

    @staticmethod
    def forward(ctx, T, K, scales, grad_hint, *tensors):
        import ctypes
        lib = _lib.load()
        n = len(scales)
        srcs = [_dev(t, "src") for t in tensors[:n]]
        depths = [_dev(t, "depth") for t in tensors[n:2 * n]]
        targets = [_dev(t, "target") for t in tensors[2 * n:3 * n]]
        T, K = _dev(T, "T"), _dev(K, "K")
        B, N = srcs[0].shape[:2]
        hs, ws_ = [s.shape[2] for s in srcs], [s.shape[3] for s in srcs]
        for s, d, t in zip(srcs, depths, targets):
            if s.shape[:2] != (B, N) or s.shape[4] != 3 or d.numel() != B * s.shape[2] * s.shape[3] \
                    or tuple(t.shape) != (B, s.shape[2], s.shape[3], 3):
                raise _lib.XptHipError(f"photo_fused_ms: inconsistent shapes src{tuple(s.shape)} depth{tuple(d.shape)} "
                                       f"target{tuple(t.shape)}")
        if T.numel() != B * N * 16 or K.numel() != B * 9:
            raise _lib.XptHipError(f"photo_fused_ms: T{tuple(T.shape)} K{tuple(K.shape)}")
        losses = torch.empty((2 * n, B), dtype=torch.float32, device=T.device)
        nws = sum(lib.xpt_photo_fused_workspace_floats(B, N, h, w) for h, w in zip(hs, ws_))
        ws = torch.empty(nws, dtype=torch.float32, device=T.device)
        P = ctypes.c_void_p * n
        ptrs = lambda ts: P(*[t.data_ptr() for t in ts])       # noqa: E731
        ci, cf = (ctypes.c_int * n)(*hs), (ctypes.c_float * n)(*[float(s) for s in scales])
        cw = (ctypes.c_int * n)(*ws_)
        needs_grad = any(ctx.needs_input_grad[i] for i in (0,) + tuple(range(4 + n, 4 + 2 * n)))
        one_pass = (grad_hint is not None and needs_grad and _ONE_PASS and not _MARCH_V1 and len(grad_hint) == 2 * n)
        ctx.stash = None
        if one_pass:
            hint = _constant_rows(grad_hint, B, T.device)
            ddepths = [torch.empty_like(d) for d in depths]
            dT = torch.empty_like(T)
            _lib.check(lib.xpt_photo_march_ms_fwdbwd(n, ptrs(srcs), ptrs(depths), _ptr(T), _ptr(K), ptrs(targets),
                                                     ptrs([hint[i] for i in range(n)]),
                                                     ptrs([hint[n + i] for i in range(n)]), _ptr(losses), ptrs(ddepths),
                                                     _ptr(dT), _ptr(ws), nws, B, N, ci, cw, cf, _stream()),
                       "xpt_photo_march_ms_fwdbwd")
            ctx.stash = (hint, ddepths, dT)
        else:
            fwd = lib.xpt_photo_fused_ms_fwd if _MARCH_V1 else lib.xpt_photo_march_ms_fwd
            _lib.check(fwd(n, ptrs(srcs), ptrs(depths), _ptr(T), _ptr(K), ptrs(targets), _ptr(losses), _ptr(ws), nws, B, N,
                           ci, cw, cf, _stream()), "xpt_photo_march_ms_fwd")
        ctx.save_for_backward(T, K, *srcs, *depths, *targets)
        ctx.cfg = (n, tuple(float(s) for s in scales), hs, ws_, nws)
        ctx.set_materialize_grads(False)
        return tuple(losses[i] for i in range(2 * n))          # l1 of every scale, then ssim of every scale

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        lib = _lib.load()
        n, scales, hs, ws_, nws = ctx.cfg
        saved = ctx.saved_tensors
        T, K = saved[0], saved[1]
        srcs, depths, targets = saved[2:2 + n], saved[2 + n:2 + 2 * n], saved[2 + 2 * n:2 + 3 * n]
        B, N = srcs[0].shape[:2]
        if ctx.stash is not None:
            hint, ddepths, dT = ctx.stash
            ctx.stash = None
            # the announced gradients against the ones that arrived (a device comparison + one fetch: only outside graph
            # capture; the trainers' eager warm-up steps run it before every capture).  A capture trusts the announcement only
            # while no eager step of this process has seen it miss: after a miss (an upstream scaling the loss object does not
            # know about -- backward(gradient=...), a re-weighted total) captured steps take the two-pass path below as well
            if (torch.cuda.is_current_stream_capturing() and not PHOTO_HINT_MISSES) or PHOTO_HINT_CHECK is False or (
                    not torch.cuda.is_current_stream_capturing()) and all(
                    (g is None and not bool(hint[i].any())) or (g is not None and torch.equal(g.reshape(-1), hint[i]))
                    for i, g in enumerate(grads)):
                return (dT, None, None, None, *([None] * n), *ddepths, *([None] * n))
            if not torch.cuda.is_current_stream_capturing():
                PHOTO_HINT_MISSES.append(tuple(None if g is None else float(g.reshape(-1)[0]) for g in grads))
        zero = None
        gs = []
        for g in grads:
            if g is None:
                if zero is None:
                    zero = torch.zeros(B, dtype=torch.float32, device=T.device)
                g = zero
            gs.append(_dev(g, "grad"))
        ddepths = [torch.empty_like(d) for d in depths]
        dT = torch.empty_like(T)
        ws = torch.empty(nws, dtype=torch.float32, device=T.device)
        P = ctypes.c_void_p * n
        ptrs = lambda ts: P(*[t.data_ptr() for t in ts])       # noqa: E731
        bwd = lib.xpt_photo_fused_ms_bwd if _MARCH_V1 else lib.xpt_photo_march_ms_bwd
        _lib.check(bwd(n, ptrs(srcs), ptrs(depths), _ptr(T), _ptr(K), ptrs(targets), ptrs(gs[:n]), ptrs(gs[n:]),
                       ptrs(ddepths), _ptr(dT), _ptr(ws), nws, B, N, (ctypes.c_int * n)(*hs), (ctypes.c_int * n)(*ws_),
                       (ctypes.c_float * n)(*scales), _stream()), "xpt_photo_march_ms_bwd")
        return (dT, None, None, None, *([None] * n), *ddepths, *([None] * n))


PHOTO_HINT_CHECK = True       # False: trust the announced gradients (no device fetch in eager steps)
PHOTO_HINT_MISSES = []        # announced-vs-actual mismatches seen by backward() (each one fell back to the two-pass path)


def photo_fused_multi_scale(srcs, depths, T, K, targets, scales, grad_hint=None):
    """[(l1 [B], ssim [B]) per scale] of photo_fused, all scales in one launch (N must be 4 or 1, at most 4 scales).
    grad_hint: ([d total / d l1_s] per scale, [d total / d ssim_s] per scale) as floats when the caller knows them (loss
    weights over the batch size): forward and backward then run as ONE pass (see _PhotoFusedMS)."""
    n = len(scales)
    hint = None
    if grad_hint is not None:
        hint = tuple(float(v) for v in grad_hint[0]) + tuple(float(v) for v in grad_hint[1])
    out = _PhotoFusedMS.apply(T, K, tuple(scales), hint, *srcs, *depths, *targets)
    return [(out[i], out[n + i]) for i in range(n)]


def photo_fused_with_synth(src, depth, T, K, target, scale):
    """Forward only (no autograd): also returns the synthesized views [B,N,h,w,3] (for image logging / tests)."""
    lib = _lib.load()
    src, depth, T, K, target = (_dev(src, "src"), _dev(depth.detach(), "depth"), _dev(T.detach(), "T"), _dev(K, "K"),
                                _dev(target, "target"))
    B, N, h, w, _ = src.shape
    l1 = torch.empty((B,), dtype=torch.float32, device=src.device)
    ss = torch.empty((B,), dtype=torch.float32, device=src.device)
    synth = torch.empty_like(src)
    nws = lib.xpt_photo_fused_workspace_floats(B, N, h, w)
    ws = torch.empty(nws, dtype=torch.float32, device=src.device)
    _lib.check(lib.xpt_photo_fused_fwd(_ptr(src), _ptr(depth), _ptr(T), _ptr(K), _ptr(target), _ptr(synth), _ptr(l1),
                                       _ptr(ss), _ptr(ws), nws, B, N, h, w, float(scale), _stream()), "xpt_photo_fused_fwd")
    return l1, ss, synth


# ------------------------------------------------------------------------------- K6 smoothness
class _Smooth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, image, grad_factor, input_is_depth):
        lib = _lib.load()
        disp, image = _dev(disp, "disp"), _dev(image, "image")
        B, h, w = image.shape[:3]
        if disp.numel() != B * h * w or image.shape[3] != 3:
            raise _lib.XptHipError(f"smoothness: disp {tuple(disp.shape)} vs image {tuple(image.shape)}")
        loss = torch.empty((B,), dtype=torch.float32, device=disp.device)
        nws = lib.xpt_smooth_workspace_floats(B, h, w)
        ws = torch.empty(nws, dtype=torch.float32, device=disp.device)
        _lib.check(lib.xpt_smooth_fwd(_ptr(disp), _ptr(image), _ptr(loss), _ptr(ws), nws, B, h, w, float(grad_factor),
                                      int(input_is_depth), _stream()), "xpt_smooth_fwd")
        ctx.save_for_backward(disp, image)
        ctx.gf, ctx.is_depth = float(grad_factor), int(input_is_depth)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        disp, image = ctx.saved_tensors
        g = _dev(g, "grad")
        B, h, w = image.shape[:3]
        dinput = torch.empty_like(disp)
        _lib.check(lib.xpt_smooth_bwd(_ptr(disp), _ptr(image), _ptr(g), _ptr(dinput), B, h, w, ctx.gf, ctx.is_depth,
                                      _stream()), "xpt_smooth_bwd")
        return dinput, None, None, None


def smoothness(disp, image, grad_factor, input_is_depth=False):
    """smootheness_loss (losses.py:409-440) for one scale -> [B]; gradient w.r.t. disp
    (or depth when input_is_depth: disp = safe_reciprocal_number(depth) fused, util_funcs.py:157-160)."""
    return _Smooth.apply(disp, image, grad_factor, input_is_depth)


class _MergeTotal(torch.autograd.Function):
    """(total, by_type) of the total-loss merge in one launch; the backward hands every term one row of a dense
    [terms, batch] matrix written by one launch."""

    @staticmethod
    def forward(ctx, c_vec, a_mat, *terms):
        import ctypes
        lib = _lib.load()
        n, batch = len(terms), terms[0].numel()
        terms = [_dev(t, "term") for t in terms]
        if any(t.numel() != batch for t in terms) or a_mat.shape[1] != n or c_vec.numel() != n:
            raise _lib.XptHipError("merge_total: inconsistent term / coefficient shapes")
        c_vec, a_mat = _dev(c_vec, "c"), _dev(a_mat, "a")
        total = torch.empty((), dtype=torch.float32, device=c_vec.device)
        by_type = torch.empty((a_mat.shape[0],), dtype=torch.float32, device=c_vec.device)
        _lib.check(lib.xpt_merge_total_fwd(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms]), _ptr(c_vec),
                                           _ptr(a_mat), _ptr(total), _ptr(by_type), batch, a_mat.shape[0], _stream()),
                   "xpt_merge_total_fwd")
        ctx.save_for_backward(c_vec)
        ctx.shape = (n, batch)
        ctx.mark_non_differentiable(by_type)
        ctx.set_materialize_grads(False)
        return total, by_type

    @staticmethod
    def backward(ctx, g_total, _):
        lib = _lib.load()
        c_vec, = ctx.saved_tensors
        n, batch = ctx.shape
        g_total = _dev(g_total, "grad")
        grads = torch.empty((n, batch), dtype=torch.float32, device=c_vec.device)
        _lib.check(lib.xpt_merge_total_bwd(n, _ptr(c_vec), _ptr(g_total), _ptr(grads), batch, _stream()),
                   "xpt_merge_total_bwd")
        return (None, None, *grads.unbind(0))


def graph_census(graph):
    """{"kernel", "memcpy", "memset", "host", "other", "total"} node counts of a captured torch.cuda.CUDAGraph that was
    created with keep_graph=True (xpt_graph_node_census walks the hipGraph_t, child graphs included)."""
    import ctypes
    counts = (ctypes.c_int * 6)()
    _lib.check(_lib.load().xpt_graph_node_census(ctypes.c_void_p(graph.raw_cuda_graph()), counts), "xpt_graph_node_census")
    return dict(zip(("kernel", "memcpy", "memset", "host", "other", "total"), [int(c) for c in counts]))


def merge_total(c_vec, a_mat, terms):
    """total = c . rowsum(terms), by_type = A rowsum(terms) (at most 64 float32 terms [batch] and 64 types)."""
    return _MergeTotal.apply(c_vec, a_mat, *terms)


class _SmoothMS(torch.autograd.Function):
    """All scales of the smoothness loss in one forward pair and one backward launch (xpt_smooth_ms_*)."""

    @staticmethod
    def forward(ctx, grad_factor, input_is_depth, n, *tensors):
        import ctypes
        lib = _lib.load()
        disps = [_dev(t, "disp") for t in tensors[:n]]
        images = [_dev(t, "image") for t in tensors[n:]]
        B = images[0].shape[0]
        hs, ws_ = [int(t.shape[1]) for t in images], [int(t.shape[2]) for t in images]
        for d, im in zip(disps, images):
            if im.shape[0] != B or im.shape[3] != 3 or d.numel() != B * im.shape[1] * im.shape[2]:
                raise _lib.XptHipError(f"smoothness_ms: disp {tuple(d.shape)} vs image {tuple(im.shape)}")
        losses = torch.empty((n, B), dtype=torch.float32, device=images[0].device)
        nws = sum(lib.xpt_smooth_workspace_floats(B, h, w) for h, w in zip(hs, ws_))
        ws = torch.empty(nws, dtype=torch.float32, device=images[0].device)
        P = ctypes.c_void_p * n
        ptrs = lambda ts: P(*[t.data_ptr() for t in ts])       # noqa: E731
        _lib.check(lib.xpt_smooth_ms_fwd(n, ptrs(disps), ptrs(images), _ptr(losses), _ptr(ws), nws, B,
                                         (ctypes.c_int * n)(*hs), (ctypes.c_int * n)(*ws_), float(grad_factor),
                                         int(input_is_depth), _stream()), "xpt_smooth_ms_fwd")
        ctx.save_for_backward(*disps, *images)
        ctx.cfg = (n, float(grad_factor), int(input_is_depth), hs, ws_)
        ctx.set_materialize_grads(False)
        return tuple(losses[i] for i in range(n))

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        lib = _lib.load()
        n, gf, is_depth, hs, ws_ = ctx.cfg
        disps, images = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        B = images[0].shape[0]
        zero = None
        gs = []
        for g in grads:
            if g is None:
                if zero is None:
                    zero = torch.zeros(B, dtype=torch.float32, device=images[0].device)
                g = zero
            gs.append(_dev(g, "grad"))
        dinputs = [torch.empty_like(d) for d in disps]
        P = ctypes.c_void_p * n
        ptrs = lambda ts: P(*[t.data_ptr() for t in ts])       # noqa: E731
        _lib.check(lib.xpt_smooth_ms_bwd(n, ptrs(disps), ptrs(images), ptrs(gs), ptrs(dinputs), B,
                                         (ctypes.c_int * n)(*hs), (ctypes.c_int * n)(*ws_), gf, is_depth, _stream()),
                   "xpt_smooth_ms_bwd")
        return (None, None, None, *dinputs, *([None] * n))


def smoothness_multi_scale(disps, images, grad_factor, input_is_depth=False):
    """[smoothness(disp_s, image_s) for every scale] in one launch pair (at most 4 scales)."""
    n = len(disps)
    return list(_SmoothMS.apply(grad_factor, input_is_depth, n, *disps, *images))


# ------------------------------------------------------------------------------- deferred parameter gradients
class GradSink:
    """Collects the split-K / per-workgroup partial sums of parameter gradients during one backward pass and finishes
    them all with ONE launch (xpt_reduce_partials) straight into the flat gradient buffer.

    A parameter takes part when it carries a `flat_grad` attribute: the persistent, contiguous float32 destination
    (FlatParameters sets it to the parameter's view of the flat gradient buffer).  Layers then call
    `partials(param, tag, nfloats)` for a persistent workspace, launch their *_partials kernel into it and `add()`
    the job; autograd gets None for that parameter.  `flush()` runs after loss.backward(): the job table is built on
    the first step (and rebuilt whenever the set of jobs changes, which it does not for a static model) and lives on
    the device, so a captured hipGraph replays it unchanged.  A layer applied several times per step contributes one
    segment per use (at most 4)."""

    def __init__(self):
        self.enabled = True
        self.buffers = {}        # (id(dst), tag, use) -> workspace tensor
        self.uses = {}           # (id(dst), tag) -> uses so far this step
        self.pending = []        # (dst, src tensor, n, nsplit, stride)
        self.signature = None
        self.table = None        # [(jobs tensor, blockmap tensor, nblocks)] per finishing launch
        self.scratch = None
        # Captured hipGraphs keep raw pointers to the job tables, the scratch matrix and the partial-sum workspaces.  A
        # trainer holds one captured step per input signature (mixed image sizes), so nothing a capture may have seen is
        # ever freed: tables are cached per job signature, outgrown workspaces are retired, not released.
        self.tables = {}         # job signature -> (table, scratch)
        self.retired = []
        self.join_streams = set()   # side streams that still produce partials of this step (hip/conv.py weight gradients)
        self.pre_flush = []         # callables run first thing in flush()

    def wants(self, param):
        return self.enabled and param is not None and getattr(param, "flat_grad", None) is not None

    def partials(self, dst, tag, nfloats):
        use = self.uses.get((id(dst), tag), 0)
        self.uses[(id(dst), tag)] = use + 1
        key = (id(dst), tag, use)
        buf = self.buffers.get(key)
        if buf is None or buf.numel() < nfloats:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.XptHipError("GradSink: new workspace requested during graph capture (run an eager step first)")
            if buf is not None:
                self.retired.append(buf)           # an earlier capture may still write here
            buf = self.buffers[key] = torch.empty(nfloats, dtype=torch.float32, device=dst.device)
        return buf

    def add(self, dst, src, offset, n, nsplit, stride):
        if not dst.is_contiguous() or dst.dtype != torch.float32 or dst.numel() != n:
            raise _lib.XptHipError(f"GradSink: destination must be a contiguous float32 tensor of {n} elements")
        self.pending.append((dst, src, int(offset), int(n), int(nsplit), int(stride)))

    def flush(self):
        for hook in self.pre_flush:                    # (hip/conv.py: queued weight-gradient launches nobody issued)
            hook()
        pending, self.pending = self.pending, []
        self.uses.clear()
        here = torch.cuda.current_stream()
        for side in self.join_streams:                 # partials produced off the main stream must have landed
            if side != here:                           # (finishing ON the producing stream: already in order)
                here.wait_stream(side)
        self.join_streams.clear()
        if not pending:
            return
        lib = _lib.load()
        sig = tuple((d.data_ptr(), s.data_ptr() + 4 * off, n, ns, st) for d, s, off, n, ns, st in pending)
        if sig != self.signature:
            cached = self.tables.get(sig)
            if cached is None:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.XptHipError("GradSink: the set of deferred gradients changed during graph capture")
                self.scratch = None
                cached = self.tables[sig] = (self._build(lib, pending), self.scratch)
            self.table, self.scratch = cached
            self.signature = sig
        for jobs, blockmap, nblocks in self.table:
            _lib.check(lib.xpt_reduce_partials(_ptr(jobs), _ptr(blockmap), nblocks, _stream()), "xpt_reduce_partials")

    MAX_SPLITS = int(__import__('os').environ.get('XPT_SINK_MAX_SPLITS', '256'))   # per job and pass; more are folded by a first pass into groups of GROUP splits
    GROUP = int(__import__('os').environ.get('XPT_SINK_GROUP', '256'))
    FLAT_MAX = int(__import__('os').environ.get('XPT_SINK_FLAT_MAX', '32'))     # most splits served 2048 outputs per workgroup

    def _build(self, lib, pending):
        """Job tables of the finishing launches: [pass 1 (only when some job has > MAX_SPLITS splits), pass 2]."""
        import ctypes
        assert lib.xpt_reduce_job_bytes() == ctypes.sizeof(_lib.ReduceJob)
        dev = pending[0][0].device
        by_dst, order = {}, []
        for d, s, off, n, ns, st in pending:
            key = d.data_ptr()
            if key not in by_dst:
                by_dst[key] = (d, n, [])
                order.append(key)
            by_dst[key][2].append((s.data_ptr() + 4 * off, ns, st))
        first, final = [], []          # entries: (dst_ptr, n, [(src_ptr, nsplit, stride), ...])
        scratch_need = sum(n * -(-ns // self.GROUP) for key in order for _, ns, _ in by_dst[key][2]
                           if ns > self.MAX_SPLITS for n in [by_dst[key][1]])
        if scratch_need:
            self.scratch = torch.empty(scratch_need, dtype=torch.float32, device=dev)
        used = 0
        for key in order:
            d, n, segs = by_dst[key]
            if len(segs) > 4:
                raise _lib.XptHipError("GradSink: a parameter is used more than 4 times per step")
            folded = []
            for ptr, ns, st in segs:
                if ns <= self.MAX_SPLITS:
                    folded.append((ptr, ns, st))
                    continue
                groups = -(-ns // self.GROUP)
                base = self.scratch.data_ptr() + 4 * used
                for g in range(groups):
                    first.append((base + 4 * g * n, n, [(ptr + 4 * g * self.GROUP * st, min(self.GROUP, ns - g * self.GROUP), st)]))
                folded.append((base, groups, n))
                used += groups * n
            final.append((d.data_ptr(), n, folded))
        return [self._table(entries, dev) for entries in (first, final) if entries]

    @staticmethod
    def _table(entries, dev):
        import numpy as np
        jobs = (_lib.ReduceJob * len(entries))()
        blockmap = []
        for j, (dst, n, segs) in enumerate(entries):
            waves = 1 if max(ns for _, ns, _ in segs) <= 8 else 4
            # wide mode (csrc/xpt_reduce.hip reduce_wide): whole 1 KiB rows per wave when every row is 16-byte aligned
            wide = n >= 256 and n % 4 == 0 and dst % 16 == 0 and all(ptr % 16 == 0 and st % 4 == 0 for ptr, _, st in segs)
            if wide:
                # <= 8 splits per segment (the large weights; a single split is a plain copy): 2048 outputs per
                # workgroup (reduce_flat), the launch being bound by the workgroup dispatch rate when a workgroup
                # moves only a few KiB
                waves = 32 if max(ns for _, ns, _ in segs) <= GradSink.FLAT_MAX else 16
            job = jobs[j]
            job.dst, job.n, job.nseg, job.split_waves = dst, n, len(segs), waves
            for g, (ptr, ns, st) in enumerate(segs):
                job.src[g], job.nsplit[g], job.stride[g] = ptr, ns, st
            blockmap.extend((j, first) for first in range(0, n, {4: 64, 32: 2048}.get(waves, 256)))
        jobs_t = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        map_t = torch.from_numpy(np.asarray(blockmap, dtype=np.int32).reshape(-1, 2)).to(dev)
        return jobs_t, map_t, len(blockmap)


grad_sink = GradSink()


# ------------------------------------------------------------------------------- depthwise conv (NASNet separable convs)
def _nhwc(t, name):
    if not t.is_cuda:
        raise _lib.XptHipError(f"{name}: expected a CUDA/HIP tensor (the xpt HIP ops have no CPU fallback)")
    if t.dtype not in (torch.float32, _lib.half()):
        raise _lib.XptHipError(f"{name}: expected float32 or bfloat16, got {t.dtype}")
    return t.contiguous(memory_format=torch.channels_last)


class _DepthwiseConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, stride, pad_t, pad_b, pad_l, pad_r, relu_in):
        lib = _lib.load()
        x = _nhwc(x, "x")
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_cuda:
            raise _lib.XptHipError("depthwise weight must be a float32 CUDA tensor")
        w = w.contiguous()
        B, C, H, W = x.shape
        k = w.shape[-1]
        if w.numel() != C * k * k:
            raise _lib.XptHipError(f"depthwise weight {tuple(w.shape)} does not match {C} channels")
        OH = (H + pad_t + pad_b - k) // stride + 1
        OW = (W + pad_l + pad_r - k) // stride + 1
        y = torch.empty((B, C, OH, OW), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dt = 0 if x.dtype == torch.float32 else 1
        _lib.check(lib.xpt_dwconv_fwd(_ptr(x), _ptr(w), _ptr(y), B, H, W, C, k, stride, pad_t, pad_l, OH, OW,
                                      int(relu_in), dt, _stream()), "xpt_dwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad_t, pad_l, OH, OW, int(relu_in), dt, weight.shape)
        ctx.sink_dst = weight.flat_grad if grad_sink.wants(weight) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        stride, pad_t, pad_l, OH, OW, relu_in, dt, wshape = ctx.cfg
        dy = _nhwc(dy.to(x.dtype), "dy")
        B, C, H, W = x.shape
        k = w.shape[-1]
        dx = dw = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.sink_dst is not None and x.numel() < (1 << 21):
            # small maps (everything but the stem at batch 8): both gradients in one launch
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            nchunk = lib.xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, stride)
            n = C * k * k
            ws = grad_sink.partials(ctx.sink_dst, "dw", nchunk * n)
            _lib.check(lib.xpt_dwconv_bwd_both(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(ws), ws.numel(), B, H, W, C, k,
                                               stride, pad_t, pad_l, OH, OW, relu_in, dt, _stream()), "xpt_dwconv_bwd_both")
            grad_sink.add(ctx.sink_dst, ws, 0, n, nchunk, n)
            return dx, None, None, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x, memory_format=torch.channels_last)
            _lib.check(lib.xpt_dwconv_bwd_data(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), B, H, W, C, k, stride, pad_t,
                                               pad_l, OH, OW, relu_in, dt, _stream()), "xpt_dwconv_bwd_data")
        if ctx.needs_input_grad[1] and ctx.sink_dst is not None:
            nchunk = lib.xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, stride)
            n = C * k * k
            ws = grad_sink.partials(ctx.sink_dst, "dw", nchunk * n)
            _lib.check(lib.xpt_dwconv_bwd_weight_partials(_ptr(x), _ptr(dy), _ptr(ws), ws.numel(), B, H, W, C, k, stride,
                                                          pad_t, pad_l, OH, OW, relu_in, dt, _stream()),
                       "xpt_dwconv_bwd_weight_partials")
            grad_sink.add(ctx.sink_dst, ws, 0, n, nchunk, n)
        elif ctx.needs_input_grad[1]:
            dw = torch.empty(wshape, dtype=torch.float32, device=x.device)
            nws = lib.xpt_dwconv_bwd_weight_workspace_floats(B, OH, OW, C, k)
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            _lib.check(lib.xpt_dwconv_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), nws, B, H, W, C, k, stride,
                                                 pad_t, pad_l, OH, OW, relu_in, dt, _stream()), "xpt_dwconv_bwd_weight")
        return dx, dw, None, None, None, None, None, None


def depthwise_conv2d(x, weight, stride=1, padding=(0, 0, 0, 0), relu_in=False):
    """Depthwise k x k convolution of an NCHW-indexed, channels_last-stored tensor.  weight [C,1,k,k] float32;
    padding = (top, bottom, left, right) zeros; relu_in fuses the preceding ReLU.  x may be float32 or bfloat16
    (fp32 accumulation); the result has x's dtype.  Differentiable w.r.t. x and weight."""
    pt, pb, pl, pr = (int(p) for p in padding)
    return _DepthwiseConv.apply(x, weight, int(stride), pt, pb, pl, pr, bool(relu_in))


class _MultiDepthwise(torch.autograd.Function):
    """n depthwise convolutions of one activation shape and stride in one launch forward and one backward
    (xpt_dwconv_multi_fwd / _bwd).  args = (relu_in, stride, pads, x_0..x_{n-1}, w_0..w_{n-1}); pads = per-layer
    (top, bottom, left, right); inputs may repeat: the gradient of a repeated input is returned once, already summed
    over its jobs."""

    precomputed = None      # outputs a fused launch already produced (pretrained_nets.fused_sep_stage): consumed by the next forward

    @staticmethod
    def forward(ctx, relu_in, stride, pads, *tensors):
        import ctypes
        lib = _lib.load()
        n = len(tensors) // 2
        xs = [_nhwc(t, "x") for t in tensors[:n]]
        ws = [t.detach().contiguous() for t in tensors[n:]]
        pre, _MultiDepthwise.precomputed = _MultiDepthwise.precomputed, None
        B, C, H, W = xs[0].shape
        ks = [int(w.shape[-1]) for w in ws]
        outs = {((H + pt + pb - k) // stride + 1, (W + pl + pr - k) // stride + 1) for k, (pt, pb, pl, pr) in zip(ks, pads)}
        for x, w in zip(xs, ws):
            if x.shape != xs[0].shape or x.dtype != xs[0].dtype or w.dtype != torch.float32 or w.numel() != C * w.shape[-1] ** 2:
                raise _lib.XptHipError("multi_depthwise: all inputs need one shape / dtype, weights [C,1,k,k] float32")
        if len(outs) != 1:
            raise _lib.XptHipError(f"multi_depthwise: the layers disagree on the output size: {outs}")
        OH, OW = outs.pop()
        pts, pls = [int(p[0]) for p in pads], [int(p[2]) for p in pads]
        dt = 0 if xs[0].dtype == torch.float32 else 1
        if pre is not None:                    # the fused branch-stage launch wrote the depthwise outputs already
            ys = list(pre)
        else:
            ys = [torch.empty((B, C, OH, OW), dtype=xs[0].dtype, device=xs[0].device, memory_format=torch.channels_last)
                  for _ in range(n)]
            P, I = ctypes.c_void_p * n, ctypes.c_int * n
            _lib.check(lib.xpt_dwconv_multi_fwd(P(*[x.data_ptr() for x in xs]), P(*[w.data_ptr() for w in ws]),
                                                P(*[y.data_ptr() for y in ys]), I(*ks), I(*pts), I(*pls), n, B, H, W, C,
                                                int(stride), OH, OW, int(relu_in), dt, _stream()), "xpt_dwconv_multi_fwd")
        # distinct inputs (by storage), in order of first use
        ptrs, input_of = [], []
        for x in xs:
            if x.data_ptr() not in ptrs:
                ptrs.append(x.data_ptr())
            input_of.append(ptrs.index(x.data_ptr()))
        first = [input_of.index(u) for u in range(len(ptrs))]
        ctx.save_for_backward(*[xs[j] for j in first], *ws)
        ctx.cfg = (n, ks, pts, pls, input_of, first, int(relu_in), int(stride), dt, (B, C, H, W, OH, OW))
        ctx.sinks = [t.flat_grad if grad_sink.wants(t) else None for t in tensors[n:]]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        import ctypes
        lib = _lib.load()
        n, ks, pts, pls, input_of, first, relu_in, stride, dt, (B, C, H, W, OH, OW) = ctx.cfg
        nu = len(first)
        saved = ctx.saved_tensors
        xin, ws = saved[:nu], saved[nu:]
        dtype = xin[0].dtype
        zero = None
        prepared = []
        for dy in dys:
            if dy is None:
                if zero is None:
                    zero = torch.zeros((B, C, OH, OW), dtype=dtype, device=xin[0].device, memory_format=torch.channels_last)
                prepared.append(zero)
            else:
                prepared.append(_nhwc(dy.to(dtype), "dy"))
        dxin = [torch.empty_like(x, memory_format=torch.channels_last) for x in xin]
        parts, later = [], []
        for j in range(n):
            nchunk = lib.xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, ks[j], stride)
            nfl = C * ks[j] * ks[j]
            if ctx.sinks[j] is not None:
                buf = grad_sink.partials(ctx.sinks[j], "dw", nchunk * nfl)
                grad_sink.add(ctx.sinks[j], buf, 0, nfl, nchunk, nfl)
            else:
                buf = torch.empty(nchunk * nfl, dtype=torch.float32, device=xin[0].device)
                later.append((j, buf, nchunk, nfl))
            parts.append(buf)
        PU, PN, IN_ = ctypes.c_void_p * nu, ctypes.c_void_p * n, ctypes.c_int * n
        _lib.check(lib.xpt_dwconv_multi_bwd(PU(*[x.data_ptr() for x in xin]), PU(*[d.data_ptr() for d in dxin]), nu,
                                            PN(*[d.data_ptr() for d in prepared]), PN(*[w.data_ptr() for w in ws]),
                                            PN(*[p.data_ptr() for p in parts]), IN_(*ks), IN_(*pts), IN_(*pls),
                                            IN_(*input_of), n, B, H, W, C, stride, OH, OW, relu_in, dt, _stream()),
                   "xpt_dwconv_multi_bwd")
        gx = [None] * n
        for u, j in enumerate(first):
            gx[j] = dxin[u]                       # repeated inputs: autograd adds None for the other uses
        gw = [None] * n
        for j, buf, nchunk, nfl in later:         # weights outside the gradient sink: finish here
            gw[j] = buf.view(nchunk, nfl).sum(0).view(ws[j].shape)
        return (None, None, None, *gx, *gw)


def multi_depthwise(inputs, weights, relu_in=True, stride=1, pads=None):
    """[dwconv(f(x_j), w_j)] for up to 6 (input, weight) pairs of one activation shape and stride, kernel sizes 3 / 5 / 7,
    in one launch; `inputs` may contain the same tensor several times.  pads: per-layer (top, bottom, left, right),
    default SAME for stride 1 (k // 2 on every side)."""
    if not 1 <= len(inputs) <= 6 or len(inputs) != len(weights):
        raise _lib.XptHipError("multi_depthwise: 1..6 (input, weight) pairs")
    if pads is None:
        if stride != 1:
            raise _lib.XptHipError("multi_depthwise: explicit pads needed for stride 2")
        pads = [(int(w.shape[-1]) // 2,) * 4 for w in weights]
    return list(_MultiDepthwise.apply(relu_in, int(stride), tuple(tuple(p) for p in pads), *inputs, *weights))


# ------------------------------------------------------------------------------- per-channel conv epilogues
def _rows_with_pitch(t):
    """NCHW-indexed tensor -> (tensor, row pitch in elements) readable as [B*H*W rows, C] with unit channel stride:
    dense channels_last tensors and channel slices of them as they are, anything else after a channels_last copy."""
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if (sc == 1 and sh == W * sw and sb == H * sh and sw >= C) and B * H * W > 1:
        return t, sw
    return t.contiguous(memory_format=torch.channels_last), C


class _AffineAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, slope, relu_in, residual):
        lib = _lib.load()
        x = _nhwc(x, "x")
        B, C, H, W = x.shape
        rows = B * H * W
        for name, t in (("gamma", gamma), ("beta", beta), ("mean", mean), ("var", var)):
            if t is not None and (t.dtype != torch.float32 or not t.is_cuda or t.numel() != C):
                raise _lib.XptHipError(f"affine_act: {name} must be a float32 CUDA vector of {C} elements")
        if residual is not None:
            if residual.shape != x.shape or slope != 1.0:
                raise _lib.XptHipError("affine_act: residual needs the shape of x and a linear epilogue")
            residual = _nhwc(residual.to(x.dtype), "residual")
        y = torch.empty_like(x, memory_format=torch.channels_last)
        dt = 0 if x.dtype == torch.float32 else 1
        g_, b_ = (None if gamma is None else gamma.detach().contiguous()), beta.detach().contiguous()
        _lib.check(lib.xpt_affine_act_fwd(_ptr(x), _ptr(g_), _ptr(b_), _ptr(mean), _ptr(var), float(eps), _ptr(residual),
                                          _ptr(y), rows, C, float(slope), int(relu_in), dt, _stream()), "xpt_affine_act_fwd")
        need_x = (gamma is not None) or relu_in
        ctx.save_for_backward(x if need_x else None, y if slope != 1.0 else None, g_, b_, mean, var)
        ctx.cfg = (float(eps), float(slope), int(relu_in), dt, rows, C, x.dtype)
        # deferred parameter gradients: both destinations (or the bias alone) must be flat-gradient views
        ctx.sink_dst = None
        if grad_sink.wants(beta) and (gamma is None or grad_sink.wants(gamma)):
            ctx.sink_dst = (None if gamma is None else gamma.flat_grad, beta.flat_grad)
        return y

    @staticmethod
    def backward(ctx, dy_in):
        lib = _lib.load()
        x, y, gamma, beta, mean, var = ctx.saved_tensors
        eps, slope, relu_in, dt, rows, C, dtype = ctx.cfg
        dy, pitch = _rows_with_pitch(dy_in.to(dtype))
        dres = dy_in if ctx.needs_input_grad[8] else None          # d(residual) = dy: no kernel
        dx = torch.empty(dy.shape, dtype=dtype, device=dy.device,
                         memory_format=torch.channels_last) if ctx.needs_input_grad[0] else None
        if ctx.sink_dst is not None:
            dst_gamma, dst_beta = ctx.sink_dst
            nblk = lib.xpt_affine_act_bwd_blocks(rows, C)
            ws = grad_sink.partials(dst_beta, "affine", nblk * 2 * C)
            _lib.check(lib.xpt_affine_act_bwd_partials(_ptr(x), _ptr(y), _ptr(dy), pitch, _ptr(gamma), _ptr(beta),
                                                       _ptr(mean), _ptr(var), eps, _ptr(dx), _ptr(ws), ws.numel(), rows,
                                                       C, slope, relu_in, dt, _stream()), "xpt_affine_act_bwd_partials")
            grad_sink.add(dst_beta, ws, 0, C, nblk, 2 * C)
            if dst_gamma is not None:
                grad_sink.add(dst_gamma, ws, C, C, nblk, 2 * C)
            return dx, None, None, None, None, None, None, None, dres
        dbeta = torch.empty(C, dtype=torch.float32, device=dy.device)
        dgamma = torch.empty(C, dtype=torch.float32, device=dy.device) if gamma is not None else None
        nws = lib.xpt_affine_act_bwd_workspace_floats(rows, C)
        ws = torch.empty(nws, dtype=torch.float32, device=dy.device)
        _lib.check(lib.xpt_affine_act_bwd(_ptr(x), _ptr(y), _ptr(dy), pitch, _ptr(gamma), _ptr(beta), _ptr(mean),
                                          _ptr(var), eps, _ptr(dx), _ptr(dbeta), _ptr(dgamma), _ptr(ws), nws, rows, C,
                                          slope, relu_in, dt, _stream()), "xpt_affine_act_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, dres


def bias_act(x, bias, slope=1.0):
    """y = LeakyReLU_slope(x + bias[c]) on an NCHW-indexed channels_last tensor (slope 1 = linear): the epilogue of
    CustomConv2D (layer_ops.py:31-35).  x float32 or bfloat16; bias float32; differentiable w.r.t. both."""
    return _AffineAct.apply(x, None, bias, None, None, 0.0, slope, False, None)


def batchnorm_inference(x, gamma, beta, running_mean, running_var, eps, relu_in=False, residual=None, slope=1.0):
    """keras BatchNormalization in inference mode (moving statistics, trainable gamma / beta), optionally with the
    preceding ReLU fused and with `residual` added to the result (the cell's layers.add); differentiable w.r.t. x,
    gamma, beta and residual.  slope != 1: LeakyReLU_slope of the result in the same launch (0 = the Activation('relu')
    that follows the stem's BatchNorm)."""
    return _AffineAct.apply(x, gamma, beta, running_mean, running_var, eps, float(slope), relu_in, residual)


# ------------------------------------------------------------------------------- depth head activation
class _DepthHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        x = _dev(x, "x")
        depth, disp = torch.empty_like(x), torch.empty_like(x)
        _lib.check(lib.xpt_depth_head_fwd(_ptr(x), _ptr(depth), _ptr(disp), x.numel(), _stream()), "xpt_depth_head_fwd")
        ctx.save_for_backward(x)
        return depth, disp

    @staticmethod
    def backward(ctx, g_depth, g_disp):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        if g_depth is None and g_disp is None:
            return None
        gd = None if g_depth is None else g_depth.contiguous().float()
        gs = None if g_disp is None else g_disp.contiguous().float()
        gx = torch.empty_like(x)
        _lib.check(lib.xpt_depth_head_bwd(_ptr(x), _ptr(gd), _ptr(gs), _ptr(gx), x.numel(), _stream()), "xpt_depth_head_bwd")
        return gx


class _GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _nhwc(x, "x")
        B, C, H, W = x.shape
        y = torch.empty((B, C), dtype=torch.float32, device=x.device)
        dt = 0 if x.dtype == torch.float32 else 1
        _lib.check(lib.xpt_global_avgpool_fwd(_ptr(x), _ptr(y), B, H * W, C, dt, _stream()), "xpt_global_avgpool_fwd")
        ctx.cfg = (B, C, H, W, x.dtype, dt)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, C, H, W, dtype, dt = ctx.cfg
        g = _dev(g, "grad")
        dx = torch.empty((B, C, H, W), dtype=dtype, device=g.device, memory_format=torch.channels_last)
        _lib.check(lib.xpt_global_avgpool_bwd(_ptr(g), _ptr(dx), B, H * W, C, dt, _stream()), "xpt_global_avgpool_bwd")
        return dx


def global_avg_pool(x):
    """GlobalAveragePooling2D of a channels_last [B,C,H,W] float32 / bfloat16 map -> float32 [B,C], one launch each way."""
    return _GlobalAvgPool.apply(x)


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        lib = _lib.load()
        x = _dev(x, "x")
        B, C, h, w = x.shape
        out = torch.empty((B, C, 2 * h, 2 * w), dtype=dtype, device=x.device)
        _lib.check(lib.xpt_upsample2x_fwd(_ptr(x), _ptr(out), B * C, h, w, 0 if dtype == torch.float32 else 1, _stream()),
                   "xpt_upsample2x_fwd")
        ctx.shape = (B, C, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        B, C, h, w = ctx.shape
        if g.dtype not in (torch.float32, _lib.half()):
            g = g.float()
        H, W = 2 * h, 2 * w
        pitch = g.stride(3)
        # a channel slice of an NHWC gradient (one map: C == 1) is read in place through its pixel pitch
        if not (C == 1 and pitch >= 1 and g.stride(2) == pitch * W and g.stride(0) == pitch * W * H):
            g, pitch = g.contiguous(), 1
        dx = torch.empty((B, C, h, w), dtype=torch.float32, device=g.device)
        _lib.check(lib.xpt_upsample2x_bwd(_ptr(g), pitch, _ptr(dx), B * C, h, w, 0 if g.dtype == torch.float32 else 1,
                                          _stream()), "xpt_upsample2x_bwd")
        return dx, None


class _Upsample2xSplit(torch.autograd.Function):
    """(x itself, its exact 2x bilinear up-sampling): the two consumers of a raw prediction (depth_net.py:87-92: the depth
    activation and the next decoder level) as ONE autograd node, so that their gradients meet inside the up-sampling's backward
    launch (xpt_upsample2x_bwd_add) instead of in a separate fan-in launch."""

    @staticmethod
    def forward(ctx, x, dtype):
        lib = _lib.load()
        x = _dev(x, "x")
        B, C, h, w = x.shape
        out = torch.empty((B, C, 2 * h, 2 * w), dtype=dtype, device=x.device)
        _lib.check(lib.xpt_upsample2x_fwd(_ptr(x), _ptr(out), B * C, h, w, 0 if dtype == torch.float32 else 1, _stream()),
                   "xpt_upsample2x_fwd")
        ctx.shape = (B, C, h, w)
        ctx.set_materialize_grads(False)
        return x.view_as(x), out

    @staticmethod
    def backward(ctx, g_self, g):
        lib = _lib.load()
        if g is None:
            return g_self, None
        B, C, h, w = ctx.shape
        if g.dtype not in (torch.float32, _lib.half()):
            g = g.float()
        H, W = 2 * h, 2 * w
        pitch = g.stride(3)
        if not (C == 1 and pitch >= 1 and g.stride(2) == pitch * W and g.stride(0) == pitch * W * H):
            g, pitch = g.contiguous(), 1
        add = None
        if g_self is not None:
            add = g_self.contiguous().float()
        dx = torch.empty((B, C, h, w), dtype=torch.float32, device=g.device)
        _lib.check(lib.xpt_upsample2x_bwd_add(_ptr(g), pitch, None if add is None else _ptr(add), _ptr(dx), B * C, h, w,
                                              0 if g.dtype == torch.float32 else 1, _stream()), "xpt_upsample2x_bwd_add")
        return dx, None


def upsample2x_split(x, dtype=torch.float32):
    """(x as an alias for its other consumer, upsample2x(x, dtype)); see _Upsample2xSplit."""
    return _Upsample2xSplit.apply(x, dtype)


def upsample2x(x, dtype=torch.float32):
    """Bilinear (half-pixel centres) 2x up-sampling of float32 [B,C,h,w] -> [B,C,2h,2w] of `dtype` (float32 / bfloat16)
    in one launch; the backward gathers (one launch, no zero fill, reads a strided one-channel gradient in place)."""
    return _Upsample2x.apply(x, dtype)


class _DepthHeadMS(torch.autograd.Function):
    """The depth activation of every prediction scale in one launch (and one for the backward: the gradients of all
    scales come from the loss, so they are all there when the first decoder level is reached)."""

    @staticmethod
    def forward(ctx, *xs):
        import ctypes
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        xs = [_dev(x, "x") for x in xs]
        n = len(xs)
        depths, disps = [torch.empty_like(x) for x in xs], [torch.empty_like(x) for x in xs]
        P = ctypes.c_void_p * n
        ptrs = lambda ts: P(*[t.data_ptr() for t in ts])       # noqa: E731
        _lib.check(lib.xpt_depth_head_ms_fwd(n, ptrs(xs), ptrs(depths), ptrs(disps),
                                             (ctypes.c_longlong * n)(*[x.numel() for x in xs]), _stream()),
                   "xpt_depth_head_ms_fwd")
        ctx.save_for_backward(*xs)
        return (*depths, *disps)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        lib = _lib.load()
        xs = ctx.saved_tensors
        n = len(xs)
        if all(g is None for g in grads):
            return (None,) * n
        gs = [None if g is None else g.contiguous().float() for g in grads]
        gxs = [torch.empty_like(x) for x in xs]
        P = ctypes.c_void_p * n
        ptrs = lambda ts: P(*[None if t is None else t.data_ptr() for t in ts])       # noqa: E731
        _lib.check(lib.xpt_depth_head_ms_bwd(n, ptrs(xs), ptrs(gs[:n]), ptrs(gs[n:]), ptrs(gxs),
                                             (ctypes.c_longlong * n)(*[x.numel() for x in xs]), _stream()),
                   "xpt_depth_head_ms_bwd")
        return tuple(gxs)


def inverse_sigmoid_depth_multi(xs):
    """([depth_s], [disp_s]) of inverse_sigmoid_depth for up to 4 prediction maps in one launch."""
    n = len(xs)
    out = _DepthHeadMS.apply(*xs)
    return list(out[:n]), list(out[n:])


def inverse_sigmoid_depth(x):
    """(depth, disp) = (safe_rcp(sigmoid(x) + 0.01), safe_rcp(depth)) in one launch (one more for the backward)."""
    ctx_mat = _DepthHead.apply(x)
    return ctx_mat[0], ctx_mat[1]


# ------------------------------------------------------------------------------- 3x3 SAME average pooling
class _AvgPool3Same(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        lib = _lib.load()
        xr, pitch = _rows_with_pitch(_nhwc_any(x))
        B, C, H, W = x.shape
        y = torch.empty((B, C, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        dt = 0 if x.dtype == torch.float32 else 1
        _lib.check(lib.xpt_avgpool3_same(_ptr(xr), pitch, _ptr(y), B, H, W, C, float(scale), 0, dt, _stream()),
                   "xpt_avgpool3_same")
        ctx.cfg = (float(scale), dt, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        scale, dt, dtype = ctx.cfg
        dyr, pitch = _rows_with_pitch(dy.to(dtype))
        B, C, H, W = dy.shape
        dx = torch.empty((B, C, H, W), dtype=dtype, device=dy.device, memory_format=torch.channels_last)
        _lib.check(lib.xpt_avgpool3_same(_ptr(dyr), pitch, _ptr(dx), B, H, W, C, scale, 1, dt, _stream()),
                   "xpt_avgpool3_same")
        return dx, None


def _nhwc_any(t):
    if not t.is_cuda or t.dtype not in (torch.float32, _lib.half()):
        raise _lib.XptHipError("avg_pool3_same: expected a float32 / bfloat16 CUDA/HIP tensor (no CPU fallback)")
    return t


def avg_pool3_same(x, scale=1.0):
    """scale * AveragePooling2D((3,3), strides 1, padding='same')(x) with the divisor excluding the padding; one launch
    forward, one backward (its transposed stencil), channel slices read in place."""
    return _AvgPool3Same.apply(x, scale)


# ------------------------------------------------------------------------------- the elementwise tail of a NASNet cell
class _CellTail(torch.autograd.Function):
    """relu(concat_s(sum_k scale * [avgpool3same](in))) in one launch; n_alias aliases of the result, one per consumer,
    so that the backward sees the consumers' gradients SEPARATELY and sums, masks, un-pools and splits them in one
    launch of its own (csrc/xpt_celltail.hip).  spec = tuple over slices of tuples of (input index, pooled, scale)."""

    @staticmethod
    def forward(ctx, spec, n_alias, *inputs):
        import ctypes
        lib = _lib.load()
        first = inputs[0]
        B, F, H, W = first.shape
        dtype = first.dtype
        if any(t.shape != first.shape for t in inputs):
            raise _lib.XptHipError("cell_tail: the inputs must share one shape")
        prepared = [_rows_with_pitch(_nhwc_any(t if t.dtype == dtype else t.to(dtype))) for t in inputs]
        ns = len(spec)
        out = torch.empty((B, ns * F, H, W), dtype=dtype, device=first.device, memory_format=torch.channels_last)
        nterms = (ctypes.c_int * ns)(*[len(s) for s in spec])
        src = (ctypes.c_void_p * (2 * ns))()
        pitch = (ctypes.c_longlong * (2 * ns))()
        pooled = (ctypes.c_int * (2 * ns))()
        scale = (ctypes.c_float * (2 * ns))()
        for s, terms in enumerate(spec):
            for k, (i, pl, sc) in enumerate(terms):
                t, p = prepared[i]
                src[2 * s + k], pitch[2 * s + k], pooled[2 * s + k], scale[2 * s + k] = t.data_ptr(), p, int(pl), float(sc)
        dt = 0 if dtype == torch.float32 else 1
        _lib.check(lib.xpt_cell_tail_fwd(ns, nterms, src, pitch, pooled, scale, out.data_ptr(), ns * F, B, H, W, F, dt,
                                         _stream()), "xpt_cell_tail_fwd")
        ctx.save_for_backward(out)
        ctx.cfg = (spec, len(inputs), F, dt)
        ctx.set_materialize_grads(False)
        # every alias is a view of the (never returned) buffer: were the first alias the base of the others, the list of
        # aliases the caller hangs on it would close a reference cycle (alias._base -> head -> list -> alias)
        return tuple(out.view_as(out) for _ in range(n_alias))

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        lib = _lib.load()
        out, = ctx.saved_tensors
        spec, n_in, F, dt = ctx.cfg
        live = [g for g in grads if g is not None]
        if not live:
            return (None, None) + (None,) * n_in
        while len(live) > 4:                                   # more consumers than the kernel sums: fold the rest first
            live = live[:3] + [sum_rows(live[3:])]
        B, C, H, W = out.shape
        ns = len(spec)
        prepared = [_rows_with_pitch(g if g.dtype == out.dtype else g.to(out.dtype)) for g in live]
        gm = torch.empty((B, C, H, W), dtype=out.dtype, device=out.device, memory_format=torch.channels_last)
        # terms per input; an input with one un-pooled, unscaled term takes its gradient as a channel slice of gm
        uses = [[] for _ in range(n_in)]
        for s, terms in enumerate(spec):
            for i, pl, sc in terms:
                uses[i].append((s, int(pl), float(sc)))
        need = ctx.needs_input_grad[2:]
        dense = [i for i in range(n_in) if need[i] and uses[i] and not (len(uses[i]) == 1 and not uses[i][0][1]
                                                                         and uses[i][0][2] == 1.0)]
        if len(dense) > 4 or any(len(uses[i]) > 3 for i in dense):
            raise _lib.XptHipError("cell_tail: at most 4 dense input gradients of at most 3 terms each")
        nd = len(dense)
        douts = [torch.empty((B, F, H, W), dtype=out.dtype, device=out.device, memory_format=torch.channels_last)
                 for _ in dense]
        n3 = max(3 * nd, 1)
        bterms = (ctypes.c_int * max(nd, 1))(*[len(uses[i]) for i in dense])
        slice_ = (ctypes.c_int * n3)()
        pooled = (ctypes.c_int * n3)()
        scale = (ctypes.c_float * n3)()
        for j, i in enumerate(dense):
            for k, (s, pl, sc) in enumerate(uses[i]):
                slice_[3 * j + k], pooled[3 * j + k], scale[3 * j + k] = s, pl, sc
        P = ctypes.c_void_p
        _lib.check(lib.xpt_cell_tail_bwd(len(prepared), (P * len(prepared))(*[t.data_ptr() for t, _ in prepared]),
                                         (ctypes.c_longlong * len(prepared))(*[p for _, p in prepared]), out.data_ptr(),
                                         C, gm.data_ptr(), ns, nd, (P * max(nd, 1))(*[d.data_ptr() for d in douts]),
                                         bterms, slice_, pooled, scale, B, H, W, F, dt, _stream()), "xpt_cell_tail_bwd")
        result = [None] * n_in
        for j, i in enumerate(dense):
            result[i] = douts[j]
        for i in range(n_in):
            if need[i] and result[i] is None and uses[i]:
                s = uses[i][0][0]
                result[i] = gm[:, s * F:(s + 1) * F]
        return (None, None, *result)


def cell_tail(spec, inputs, n_alias=3):
    """-> list of n_alias aliases of relu(concat(slices)), marked as already rectified (pretrained_nets.shared_relu hands
    them to the consumers one by one)."""
    outs = _CellTail.apply(tuple(tuple(t) for t in spec), n_alias, *inputs)
    return list(outs)


class _AdjustGather(torch.autograd.Function):
    """p [B,C,H,W] -> (p[:, :, ::2, ::2], p shifted by (1, 1) then [::2, ::2] with zeros past the border) as the two channel
    halves of ONE [B, 2C, H/2, W/2] buffer: one gather launch, and one scatter launch for the gradient of p."""

    @staticmethod
    def forward(ctx, p):
        lib = _lib.load()
        pr, pitch = _rows_with_pitch(_nhwc_any(p))
        B, C, H, W = p.shape
        H2, W2 = (H + 1) // 2, (W + 1) // 2
        out = torch.empty((B, 2 * C, H2, W2), dtype=p.dtype, device=p.device, memory_format=torch.channels_last)
        dt = 0 if p.dtype == torch.float32 else 1
        _lib.check(lib.xpt_adjust_gather(_ptr(pr), pitch, _ptr(out), B, H, W, C, dt, _stream()), "xpt_adjust_gather")
        ctx.cfg = (B, C, H, W, dt, p.dtype)
        ctx.set_materialize_grads(False)
        return out[:, :C], out[:, C:]

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None and g2 is None:
            return None
        lib = _lib.load()
        B, C, H, W, dt, dtype = ctx.cfg
        (r1, p1) = _rows_with_pitch(g1.to(dtype)) if g1 is not None else (None, 0)
        (r2, p2) = _rows_with_pitch(g2.to(dtype)) if g2 is not None else (None, 0)
        like = g1 if g1 is not None else g2
        dp = torch.empty((B, C, H, W), dtype=dtype, device=like.device, memory_format=torch.channels_last)
        _lib.check(lib.xpt_adjust_scatter(_ptr(r1), p1, _ptr(r2), p2, _ptr(dp), B, H, W, C, dt, _stream()),
                   "xpt_adjust_scatter")
        return dp


def adjust_gather(p):
    return _AdjustGather.apply(p)


class _PoolPair(torch.autograd.Function):
    """(max_pool2d, avg_pool2d)(zero_pad(h), 3, stride 2) in one launch; the backward routes both gradients in one."""

    @staticmethod
    def forward(ctx, h, pad_t, pad_b, pad_l, pad_r, split_mp=False):
        """split_mp: (mp, mp', ap) with mp' an alias of mp for a second consumer -- the backward adds their gradients on load."""
        lib = _lib.load()
        hr, pitch = _rows_with_pitch(_nhwc_any(h))
        B, C, H, W = h.shape
        OH, OW = (H + pad_t + pad_b - 3) // 2 + 1, (W + pad_l + pad_r - 3) // 2 + 1
        mp = torch.empty((B, C, OH, OW), dtype=h.dtype, device=h.device, memory_format=torch.channels_last)
        ap = torch.empty_like(mp)
        arg = torch.empty((B, OH, OW, C), dtype=torch.uint8, device=h.device)
        dt = 0 if h.dtype == torch.float32 else 1
        _lib.check(lib.xpt_pool_pair_fwd(_ptr(hr), pitch, _ptr(mp), _ptr(ap), _ptr(arg), B, H, W, C, OH, OW, pad_t, pad_l, dt,
                                         _stream()), "xpt_pool_pair_fwd")
        ctx.save_for_backward(arg)
        ctx.cfg = (B, C, H, W, OH, OW, pad_t, pad_l, dt, h.dtype)
        ctx.split = bool(split_mp)
        ctx.set_materialize_grads(False)
        return (mp, mp.view_as(mp), ap) if split_mp else (mp, ap)

    @staticmethod
    def backward(ctx, *grads):
        gm, gm2, ga = grads if ctx.split else (grads[0], None, grads[1])
        if gm is None and gm2 is not None:
            gm, gm2 = gm2, None
        if gm is None and ga is None:
            return (None,) * 6
        lib = _lib.load()
        arg, = ctx.saved_tensors
        B, C, H, W, OH, OW, pad_t, pad_l, dt, dtype = ctx.cfg
        (rm, pm) = _rows_with_pitch(gm.to(dtype)) if gm is not None else (None, 0)
        (rm2, pm2) = _rows_with_pitch(gm2.to(dtype)) if gm2 is not None else (None, 0)
        (ra, pa) = _rows_with_pitch(ga.to(dtype)) if ga is not None else (None, 0)
        dh = torch.empty((B, C, H, W), dtype=dtype, device=arg.device, memory_format=torch.channels_last)
        _lib.check(lib.xpt_pool_pair_bwd2(_ptr(rm), pm, _ptr(rm2), pm2, _ptr(ra), pa, _ptr(arg), _ptr(dh), B, H, W, C, OH, OW, pad_t,
                                          pad_l, dt, _stream()), "xpt_pool_pair_bwd2")
        return (dh,) + (None,) * 5


def pool_pair(h, pads, split_mp=False):
    """pads = ((top, bottom), (left, right)) of the zero padding -> (max pooled, average pooled), 3x3 windows, stride 2.
    split_mp: (max pooled, an alias of it for a second consumer, average pooled): no gradient fan-in launch."""
    (pt, pb), (pl, pr) = pads
    return _PoolPair.apply(h, pt, pb, pl, pr, bool(split_mp))


# ------------------------------------------------------------------------------- gradient fan-in
def sum_rows(tensors):
    """Sum of 2..8 NCHW-indexed tensors of one shape / dtype in ONE launch (dense channels_last result); operands may
    be channel slices of wider tensors (read in place through their row pitch)."""
    import ctypes
    lib = _lib.load()
    first = tensors[0]
    B, C, H, W = first.shape
    prepared = [_rows_with_pitch(t if t.dtype == first.dtype else t.to(first.dtype)) for t in tensors]
    out = torch.empty((B, C, H, W), dtype=first.dtype, device=first.device, memory_format=torch.channels_last)
    n = len(prepared)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _ in prepared])
    pitches = (ctypes.c_longlong * n)(*[p for _, p in prepared])
    _lib.check(lib.xpt_sum_rows(ptrs, pitches, n, _ptr(out), B * H * W, C, 0 if first.dtype == torch.float32 else 1,
                                _stream()), "xpt_sum_rows")
    return out


class _FanOut(torch.autograd.Function):
    """n aliases of x for n consumers; the backward adds the n incoming gradients with one kernel instead of the
    n - 1 pairwise adds of autograd's accumulation."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        live = [g for g in grads if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        if len(live) > 8 or live[0].dim() != 4 or not live[0].is_cuda or live[0].dtype not in (torch.float32, _lib.half()):
            total = live[0]
            for g in live[1:]:
                total = total + g
            return total, None
        return sum_rows(live), None


def fan_out(x, n):
    """x -> n aliases, one per consumer (see _FanOut); a no-op for tensors that do not require grad."""
    if n == 1 or not (x.requires_grad and torch.is_grad_enabled()) or not x.is_cuda:
        return (x,) * n
    return _FanOut.apply(x, n)


# ------------------------------------------------------------------------------- pointwise (1x1) convolution pieces
_COUNTERS = {}
_N_COUNTERS = 1024


def _counters(device):
    """Zero-initialised arrival counters of the split-K kernels (they leave them zero); one array per device, shared
    by every call because calls on a stream run in order.  Allocated outside any graph capture (warm-up step)."""
    buf = _COUNTERS.get(device)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.XptHipError("run one eager step before capturing: the split-K counters are allocated on first use")
        buf = _COUNTERS[device] = torch.zeros(_N_COUNTERS, dtype=torch.int32, device=device)
    return buf


def as_rows(t):
    """[B,C,H,W] NCHW-indexed tensor -> ([B*H*W, C] view, copied?) with unit channel stride and a constant row pitch.
    Dense channels_last tensors AND channel slices of them (what torch.cat's backward hands out) are consumed in
    place; anything else is first made channels_last-contiguous."""
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if not (sc == 1 and sh == W * sw and sb == H * sh and sw >= C) and not (B * H * W == 1):
        t = t.contiguous(memory_format=torch.channels_last)
        sw = C
    return t.as_strided((B * H * W, C), (sw, 1), t.storage_offset())


def conv1x1_weight_grad(dy2, x2):
    """dW [cout, cin] float32 = dy2^T @ x2 for bf16 row matrices dy2 [M, cout], x2 [M, cin] (unit column stride, any
    row pitch): the split-K matrix-core kernel of csrc/xpt_gemm.hip."""
    lib = _lib.load()
    if not (dy2.is_cuda and x2.is_cuda) or dy2.dtype != _lib.half() or x2.dtype != _lib.half():
        raise _lib.XptHipError("conv1x1_weight_grad: expected bfloat16 CUDA/HIP matrices (no CPU fallback)")
    if dy2.dim() != 2 or x2.dim() != 2 or dy2.shape[0] != x2.shape[0] or dy2.stride(1) != 1 or x2.stride(1) != 1:
        raise _lib.XptHipError(f"conv1x1_weight_grad: bad operands {tuple(dy2.shape)} {dy2.stride()} / "
                               f"{tuple(x2.shape)} {x2.stride()}")
    M, cout = dy2.shape
    cin = x2.shape[1]
    pitch_dy = dy2.stride(0) if M > 1 else cout
    pitch_x = x2.stride(0) if M > 1 else cin
    dw = torch.empty((cout, cin), dtype=torch.float32, device=dy2.device)
    nws = lib.xpt_conv1x1_bwd_weight_workspace_floats(M, cout, cin)
    ws = torch.empty(nws, dtype=torch.float32, device=dy2.device)
    cnt = _counters(dy2.device)
    _lib.check(lib.xpt_conv1x1_bwd_weight(_ptr(dy2), _ptr(x2), _ptr(dw), _ptr(ws), nws, _ptr(cnt), _N_COUNTERS, M, cout,
                                          cin, pitch_dy, pitch_x, _stream()), "xpt_conv1x1_bwd_weight")
    return dw


def conv1x1_weight_grad_deferred(dy2, x2, dst, w=None):
    """Same product as conv1x1_weight_grad, left as split-K partials for grad_sink.flush() to add into `dst`
    (the weight's flat-gradient view, [cout, cin(,1,1)] contiguous).  w (the bf16 weight [cout, cin], dense): also
    returns the data gradient dx = dy2 w [M, cin] bf16, computed by extra workgroups of the same launch."""
    lib = _lib.load()
    if not (dy2.is_cuda and x2.is_cuda) or dy2.dtype != _lib.half() or x2.dtype != _lib.half():
        raise _lib.XptHipError("conv1x1_weight_grad: expected bfloat16 CUDA/HIP matrices (no CPU fallback)")
    if dy2.dim() != 2 or x2.dim() != 2 or dy2.shape[0] != x2.shape[0] or dy2.stride(1) != 1 or x2.stride(1) != 1:
        raise _lib.XptHipError(f"conv1x1_weight_grad: bad operands {tuple(dy2.shape)} {dy2.stride()} / "
                               f"{tuple(x2.shape)} {x2.stride()}")
    M, cout = dy2.shape
    cin = x2.shape[1]
    pitch_dy = dy2.stride(0) if M > 1 else cout
    pitch_x = x2.stride(0) if M > 1 else cin
    nsplit = lib.xpt_conv1x1_bwd_weight_splits(M, cout, cin)
    ws = grad_sink.partials(dst, "conv1x1", nsplit * cout * cin)
    dx = None
    if w is not None:
        dx = torch.empty((M, cin), dtype=_lib.half(), device=dy2.device)
        _lib.check(lib.xpt_conv1x1_bwd_fused(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(dx), _ptr(ws), ws.numel(), M, cout, cin,
                                             pitch_dy, pitch_x, _stream()), "xpt_conv1x1_bwd_fused")
    else:
        _lib.check(lib.xpt_conv1x1_bwd_weight_partials(_ptr(dy2), _ptr(x2), _ptr(ws), ws.numel(), M, cout, cin, pitch_dy,
                                                       pitch_x, _stream()), "xpt_conv1x1_bwd_weight_partials")
    grad_sink.add(dst, ws, 0, cout * cin, nsplit, cout * cin)
    return dx


def multi_copy(dsts, srcs):
    """dst_i.copy_(src_i) for up to 8 pairs of contiguous same-shape, same-dtype device tensors per launch."""
    import ctypes
    lib = _lib.load()
    for d, s_ in zip(dsts, srcs):
        if not (d.is_cuda and s_.is_cuda) or d.dtype != s_.dtype or d.shape != s_.shape \
                or not d.is_contiguous() or not s_.is_contiguous():
            raise _lib.XptHipError("multi_copy: pairs of contiguous device tensors of one shape and dtype expected")
    for i in range(0, len(dsts), 8):
        dd, ss = dsts[i:i + 8], srcs[i:i + 8]
        n = len(dd)
        P, LL = ctypes.c_void_p * n, ctypes.c_longlong * n
        _lib.check(lib.xpt_multi_copy(P(*[t.data_ptr() for t in ss]), P(*[t.data_ptr() for t in dd]),
                                      LL(*[t.numel() * t.element_size() for t in dd]), n, _stream()), "xpt_multi_copy")


class _ConcatChannels(torch.autograd.Function):
    """concat(parts, channel axis) of bf16 NHWC activations plus the zero channels that pad the result to a multiple of 8:
    one launch (xpt_concat_channels); the backward hands out channel slices of the incoming gradient (views, no launch)."""

    @staticmethod
    def forward(ctx, *parts):
        import ctypes
        lib = _lib.load()
        B, _, H, W = parts[0].shape
        rows = [as_rows(p) for p in parts]
        chans = [p.shape[1] for p in parts]
        total = sum(chans)
        ct = -(-total // 8) * 8
        out = torch.empty((B, ct, H, W), dtype=_lib.half(), device=parts[0].device, memory_format=torch.channels_last)
        n = len(parts)
        M = B * H * W
        P, LL, I = ctypes.c_void_p * n, ctypes.c_longlong * n, ctypes.c_int * n
        _lib.check(lib.xpt_concat_channels(P(*[r.data_ptr() for r in rows]), LL(*[(r.stride(0) if M > 1 else c) for r, c in zip(rows, chans)]),
                                           I(*chans), n, _ptr(out), M, ct, _stream()), "xpt_concat_channels")
        ctx.chans = chans
        return out

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for i, c in enumerate(ctx.chans):
            outs.append(g[:, off:off + c] if ctx.needs_input_grad[i] else None)
            off += c
        return tuple(outs)


def concat_channels(parts):
    """torch.cat(parts, dim=1) for bf16 NCHW-indexed (channels_last) tensors, zero-padded to a multiple of 8 channels."""
    if len(parts) > 4 or any((not p.is_cuda) or p.dtype != _lib.half() for p in parts):
        raise _lib.XptHipError("concat_channels: expected up to four bfloat16 CUDA/HIP tensors (no CPU fallback)")
    return _ConcatChannels.apply(*parts)


def vector_rows(t, channels):
    """Rows of a [M, channels] bf16 view start on 4-byte boundaries and hold an even number of channels (what the fused
    data gradient of the pointwise backward kernels needs: their scalar-staged instantiations carry no such code)."""
    pitch = t.stride(0) if t.shape[0] > 1 else channels
    return channels % 2 == 0 and pitch % 2 == 0 and t.data_ptr() % 4 == 0


# ------------------------------------------------------------------------------- PWC-Net correlation cost volume
class _CorrelationCost(torch.autograd.Function):
    @staticmethod
    def forward(ctx, left, right, max_disp, stride2):
        lib = _lib.load()
        left, right = _nhwc(left, "left"), _nhwc(right, "right")
        if left.shape != right.shape or left.dtype != right.dtype:
            raise _lib.XptHipError(f"correlation_cost: left {tuple(left.shape)} {left.dtype} vs right "
                                   f"{tuple(right.shape)} {right.dtype}")
        B, C, H, W = left.shape
        DD = lib.xpt_corr_cost_channels(max_disp, stride2)
        out = torch.empty((B, DD, H, W), dtype=left.dtype, device=left.device, memory_format=torch.channels_last)
        dt = 0 if left.dtype == torch.float32 else 1
        _lib.check(lib.xpt_corr_cost_fwd(_ptr(left), _ptr(right), _ptr(out), B, H, W, C, max_disp, stride2, dt, _stream()),
                   "xpt_corr_cost_fwd")
        ctx.save_for_backward(left, right)
        ctx.cfg = (max_disp, stride2, dt)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        left, right = ctx.saved_tensors
        max_disp, stride2, dt = ctx.cfg
        B, C, H, W = left.shape
        gout = _nhwc(gout.to(left.dtype), "gout")
        dleft, dright = torch.empty_like(left), torch.empty_like(right)       # channels_last like the inputs
        _lib.check(lib.xpt_corr_cost_bwd(_ptr(left), _ptr(right), _ptr(gout), _ptr(dleft), _ptr(dright), B, H, W, C,
                                         max_disp, stride2, dt, _stream()), "xpt_corr_cost_bwd")
        return dleft, dright, None, None


def correlation_cost(left, right, max_disp, stride2):
    """tfa.layers.CorrelationCost(kernel_size=1, max_displacement, stride_1=1, stride_2, pad=max_displacement) on
    NCHW-indexed (channels_last) feature maps -> [B, (2 (max_disp // stride2) + 1)^2, H, W]  (flow_net.py:181-196)."""
    return _CorrelationCost.apply(left, right, int(max_disp), int(stride2))
