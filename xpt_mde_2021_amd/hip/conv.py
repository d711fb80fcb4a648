"""Dense k x k convolutions on the gfx950 matrix cores (csrc/xpt_conv.hip, csrc/xpt_conv_wgrad.hip) as one autograd op:
keras Conv2D(padding="same") + bias + LeakyReLU of CustomConv2D (model/model_util/layer_ops.py:5-36), optionally reading
its input through UpSampling2D(2, "nearest") (model/build_model/depth_net.py:76-84) without materialising it.

Nothing here goes through MIOpen: no library workspace, no find step, hipGraph-replay-safe by construction (every buffer
a kernel touches is either a tensor of the step or a persistent partial-sum workspace of hip/ops.py GradSink).
"""
import contextlib
import ctypes
import math
import weakref

import torch

from . import lib as _lib
from . import ops as _ops


def _stream():
    return torch.cuda.current_stream().cuda_stream


def round_up(n, m):
    return (n + m - 1) // m * m


def same_pad(n, k, s):
    """TF "SAME" (model/model_util/layer_ops.py padding="same"): total = max((ceil(n/s)-1)*s + k - n, 0); before = total // 2."""
    total = max((math.ceil(n / s) - 1) * s + k - n, 0)
    return total // 2, total - total // 2


# ------------------------------------------------------------------------------- packed bf16 weights
class ConvWeightPacker:
    """Keeps, for every registered convolution weight, the two bf16 operand layouts of the kernels (forward
    [N][T][Cp], data gradient [Cp][T][Np]) and refreshes ALL of them with one launch (`pack()`, called once per model
    forward, i.e. once per training step and captured with it).  The job table lives on the device and is rebuilt when a
    weight is added or its master storage moves (FlatParameters re-homes the parameters once)."""

    def __init__(self):
        self.entries = {}          # id(weight) -> dict(ref, fwd, bwd, geometry)
        self.signature = None
        self.table = None
        self.retired = []
        self.nblocks = 0

    def get(self, weight, need_bwd=True):
        e = self.entries.get(id(weight))
        if e is not None and e["ref"]() is weight:
            if need_bwd and e["bwd"] is None:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.XptHipError("ConvWeightPacker: new operand layout requested during graph capture")
                e["bwd"] = torch.zeros((e["Cp"], e["T"], e["Np"]), dtype=_lib.half(), device=weight.device)
                self.signature = None
                self._pack_now()
            return e
        if torch.cuda.is_current_stream_capturing():
            raise _lib.XptHipError("ConvWeightPacker: a convolution weight was first used during graph capture "
                                   "(run one eager step first)")
        N, C, KH, KW = weight.shape
        Cp, Np, T = round_up(C, 8), round_up(N, 8), KH * KW
        e = {"ref": weakref.ref(weight), "N": N, "C": C, "KH": KH, "KW": KW, "T": T, "Cp": Cp, "Np": Np,
             "fwd": torch.zeros((N, T, Cp), dtype=_lib.half(), device=weight.device),
             "bwd": torch.zeros((Cp, T, Np), dtype=_lib.half(), device=weight.device) if need_bwd else None}
        self.entries[id(weight)] = e
        self.signature = None
        self._pack_now()
        return e

    def _live(self):
        dead = [k for k, e in self.entries.items() if e["ref"]() is None]
        for k in dead:
            del self.entries[k]
        if dead:
            self.signature = None
        return [(e["ref"](), e) for e in self.entries.values()]

    def _pack_now(self):
        if not torch.cuda.is_current_stream_capturing():
            self.pack()

    def pack(self):
        live = self._live()
        live = [(w, e) for w, e in live if w is not None and w.is_cuda]
        if not live:
            return
        lib = _lib.load()
        sig = tuple((w.data_ptr(), w.stride(), e["fwd"].data_ptr(), 0 if e["bwd"] is None else e["bwd"].data_ptr())
                    for w, e in live)
        if sig != self.signature:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.XptHipError("ConvWeightPacker: the set of convolution weights changed during graph capture")
            assert lib.xpt_conv_pack_job_bytes() == ctypes.sizeof(_lib.ConvPackJob)
            jobs = (_lib.ConvPackJob * len(live))()
            first = 0
            for j, (w, e) in enumerate(live):
                if w.dtype != torch.float32:
                    raise _lib.XptHipError("ConvWeightPacker: master weights must be float32")
                sn, sc, sh, sw = w.stride()
                job = jobs[j]
                job.src, job.fwd, job.bwd = w.data_ptr(), e["fwd"].data_ptr(), (None if e["bwd"] is None else e["bwd"].data_ptr())
                job.sn, job.sc, job.sh, job.sw = sn, sc, sh, sw
                job.N, job.T, job.KW, job.C, job.Cp, job.Np = e["N"], e["T"], e["KW"], e["C"], e["Cp"], e["Np"]
                job.first_block = first
                # one workgroup per (tap, 64 output channels, 64 input channels) tile (csrc/xpt_conv.hip conv_pack_kernel)
                first += e["T"] * ((max(e["N"], e["Np"]) + 63) // 64) * ((e["Cp"] + 63) // 64)
            if self.table is not None:
                self.retired.append(self.table)     # a captured step may still launch with the old table
            self.table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(live[0][0].device)
            self.nblocks = first
            self.signature = sig
        _lib.check(lib.xpt_conv_pack_weights(self.table.data_ptr(), len(live), self.nblocks, _stream()),
                   "xpt_conv_pack_weights")


packer = ConvWeightPacker()


def _apply_env_tuning():
    """XPT_WGRAD_TUNE="max partial MiB per layer,target workgroups" (benchmarking / diagnostics)."""
    import os
    spec = os.environ.get("XPT_WGRAD_TUNE")
    if spec:
        mib, blocks = (int(v) for v in spec.split(","))
        _lib.load().xpt_conv2d_bwd_weight_tune(mib, blocks)
    spec = os.environ.get("XPT_PW_WGRAD_TUNE")             # pointwise weight gradient: waves, row pairs per wave, max workgroups, KiB
    if spec:
        _lib.load().xpt_conv1x1_bwd_weight_tune(*[int(v) for v in spec.split(",")])
    if os.environ.get("XPT_XCD_AFFINITY"):                 # 0: workgroups in launch order (A/B of the image-to-XCD numbering)
        _lib.load().xpt_set_xcd_affinity(int(os.environ["XPT_XCD_AFFINITY"]))
    for code in os.environ.get("XPT_DW_TUNE", "").split(","):     # depthwise knobs (xpt_dwconv_tune codes, see xpt_hip.h)
        if code:
            _lib.load().xpt_dwconv_tune(int(code))
    plan = os.environ.get("XPT_CONV_TUNE")                 # dense convolution plan (xpt_conv2d_tune: -1000 - n = LDS-kernel threshold)
    if plan:
        for code in plan.split(","):
            _lib.load().xpt_conv2d_tune(int(code))
    spec = os.environ.get("XPT_SPLITK_TUNE")               # deep-layer split-K path: enable, forced slices, min K, max pixels
    if spec:
        _lib.load().xpt_conv2d_splitk_tune(*[int(v) for v in spec.split(",")])
    spec = os.environ.get("XPT_STREAM_TUNE")               # persistent 3 x 3 path: enable, min tiles, workgroups per CU, LDS KiB
    if spec:
        _lib.load().xpt_conv2d_stream_tune(*[int(v) for v in spec.split(",")])
    spec = os.environ.get("XPT_PWCONV_TUNE")               # fused pointwise forward: k split from this cin on, up to this many tiles
    if spec:
        _lib.load().xpt_pwconv_tune(*[int(v) for v in spec.split(",")])
    spec = os.environ.get("XPT_HEAD_TUNE")                 # depth-head convolution backward: pixel passes per workgroup, most workgroups
    if spec:
        _lib.load().xpt_headconv_tune(*[int(v) for v in spec.split(",")])
    cap = os.environ.get("XPT_PW_DEFER_CAP_MIB")           # pointwise weight gradient: MiB of split partials per layer
    if cap:
        _lib.load().xpt_conv1x1_bwd_weight_defer_cap(int(cap))


# ------------------------------------------------------------------------------- activations as (pointer, pitch)
def nhwc_view(t, channels=None):
    """NCHW-indexed bf16 tensor -> (tensor, pixel pitch in elements): dense channels_last tensors and channel slices of
    them are used in place, anything else is made channels_last first."""
    if t.dtype != _lib.half() or not t.is_cuda:
        raise _lib.XptHipError("conv: expected a bfloat16 CUDA/HIP tensor (no CPU fallback)")
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    ok = sc == 1 and sh == W * sw and sb == H * sh and sw >= C and sw % 8 == 0 and t.data_ptr() % 16 == 0
    if not ok or B * H * W == 1:
        t = t.contiguous(memory_format=torch.channels_last)
        if t.stride(1) != 1:                       # [B,C,1,1] and friends: force the NHWC order
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        sw = C
        if C % 8 != 0:
            raise _lib.XptHipError(f"conv: {C} channels are not a multiple of 8 (pad the tensor)")
    return t, sw


_tuned = False


class _Conv2dSame(torch.autograd.Function):
    """y = LeakyReLU_slope(conv2d_same(x [nearest-2x up-sampled], weight, stride) + bias)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, slope, upsample, valid):
        lib = _lib.load()
        global _tuned
        if not _tuned:
            _apply_env_tuning()
            _tuned = True
        x, xpitch = nhwc_view(x)
        B, Cx, PH, PW = x.shape
        need_dx = ctx.needs_input_grad[0]
        e = packer.get(weight, need_bwd=need_dx)
        N, C, KH, KW, Cp = e["N"], e["C"], e["KH"], e["KW"], e["Cp"]
        if Cx != Cp:
            raise _lib.XptHipError(f"conv: input has {Cx} channels, the packed weight expects {Cp} (= {C} padded to 8)")
        H, W = PH << upsample, PW << upsample
        if valid:                                   # keras padding="valid" (the NASNet stem)
            pt = pl = 0
            OH, OW = (H - KH) // stride + 1, (W - KW) // stride + 1
        else:
            (pt, _), (pl, _) = same_pad(H, KH, stride), same_pad(W, KW, stride)
            OH, OW = -(-H // stride), -(-W // stride)
        y = torch.empty((B, N, OH, OW), dtype=_lib.half(), device=x.device, memory_format=torch.channels_last)
        b_ = None if bias is None else bias.detach()
        if b_ is not None and (b_.dtype != torch.float32 or not b_.is_contiguous()):
            raise _lib.XptHipError("conv: bias must be a contiguous float32 vector")
        wsf = 0 if valid else lib.xpt_conv2d_splitk_workspace_floats(B * OH * OW, N, Cp, KH * KW, stride)
        if wsf:
            # deep layers on the small maps: LDS-tiled implicit GEMM with a deterministic split of the reduction axis
            ws = torch.empty(wsf, dtype=torch.float32, device=x.device)
            _lib.check(lib.xpt_conv2d_fwd_splitk(x.data_ptr(), e["fwd"].data_ptr(), None if b_ is None else b_.data_ptr(),
                                                 y.data_ptr(), B, PH, PW, Cp, xpitch, N, KH, KW, pt, pl, OH, OW, N,
                                                 int(upsample), float(slope), ws.data_ptr(), wsf, _stream()),
                       "xpt_conv2d_fwd_splitk")
        elif not valid and KH == 5 and lib.xpt_conv2d_stream_serves(B, OH, OW, N, Cp, KH, KW, stride, int(upsample)):
            # PoseNet's 5 x 5 stride-2 layers on the large maps: the same persistent kernel, specialised
            _lib.check(lib.xpt_conv2d_fwd_stream_k5s2(x.data_ptr(), e["fwd"].data_ptr(), None if b_ is None else b_.data_ptr(),
                                                      y.data_ptr(), B, PH, PW, Cp, xpitch, N, pt, pl, OH, OW, N, float(slope),
                                                      _stream()), "xpt_conv2d_fwd_stream_k5s2")
        elif not valid and KH == 3 and lib.xpt_conv2d_stream_serves(B, OH, OW, N, Cp, KH, KW, stride, int(upsample)):
            # 3 x 3 layers of the half- / full-resolution levels: persistent workgroups, weights staged once
            _lib.check(lib.xpt_conv2d_fwd_stream(x.data_ptr(), e["fwd"].data_ptr(), None if b_ is None else b_.data_ptr(),
                                                 y.data_ptr(), B, PH, PW, Cp, xpitch, N, pt, pl, OH, OW, N, int(upsample),
                                                 float(slope), _stream()), "xpt_conv2d_fwd_stream")
        else:
            _lib.check(lib.xpt_conv2d_fwd(x.data_ptr(), e["fwd"].data_ptr(), None if b_ is None else b_.data_ptr(),
                                          y.data_ptr(), B, PH, PW, Cp, xpitch, N, KH, KW, stride, pt, pl, OH, OW, N,
                                          int(upsample), float(slope), _stream()), "xpt_conv2d_fwd")
        ctx.save_for_backward(x, y if slope != 1.0 else None, b_)
        ctx.geom = (B, PH, PW, Cp, C, xpitch, N, KH, KW, stride, pt, pl, OH, OW, int(upsample), float(slope))
        ctx.weight = weight
        ctx.sink_w = weight.flat_grad if (_ops.grad_sink.wants(weight) and N % 8 == 0) else None
        ctx.sink_b = bias.flat_grad if (bias is not None and _ops.grad_sink.wants(bias)) else None
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy_in):
        lib = _lib.load()
        x, y, bias = ctx.saved_tensors
        B, PH, PW, Cp, C, xpitch, N, KH, KW, stride, pt, pl, OH, OW, ups, slope = ctx.geom
        weight = ctx.weight
        rows = B * OH * OW
        # ---- activation / bias backward: g = dy * act'(y) (dense bf16 [rows, N]) and the bias gradient
        dy, dpitch = _ops._rows_with_pitch(dy_in.to(_lib.half()))
        dbias = None
        if ctx.has_bias or slope != 1.0:
            g = torch.empty((B, N, OH, OW), dtype=_lib.half(), device=dy.device, memory_format=torch.channels_last)
            if ctx.sink_b is not None:
                nblk = lib.xpt_affine_act_bwd_blocks(rows, N)
                ws = _ops.grad_sink.partials(ctx.sink_b, "affine", nblk * 2 * N)
                _lib.check(lib.xpt_affine_act_bwd_partials(None, None if y is None else y.data_ptr(), dy.data_ptr(), dpitch,
                                                           None, bias.data_ptr(), None, None, 0.0, g.data_ptr(),
                                                           ws.data_ptr(), ws.numel(), rows, N, slope, 0, 1, _stream()),
                           "xpt_affine_act_bwd_partials")
                _ops.grad_sink.add(ctx.sink_b, ws, 0, N, nblk, 2 * N)
            else:
                beta = bias if bias is not None else torch.zeros(N, dtype=torch.float32, device=dy.device)
                dbias = torch.empty(N, dtype=torch.float32, device=dy.device)
                nws = lib.xpt_affine_act_bwd_workspace_floats(rows, N)
                ws = torch.empty(nws, dtype=torch.float32, device=dy.device)
                _lib.check(lib.xpt_affine_act_bwd(None, None if y is None else y.data_ptr(), dy.data_ptr(), dpitch, None,
                                                  beta.data_ptr(), None, None, 0.0, g.data_ptr(), dbias.data_ptr(), None,
                                                  ws.data_ptr(), nws, rows, N, slope, 0, 1, _stream()), "xpt_affine_act_bwd")
                if not ctx.has_bias:
                    dbias = None
            gpitch = N
        else:
            g, gpitch = nhwc_view(dy_in.to(_lib.half()))
        if N % 8 != 0:
            raise _lib.XptHipError(f"conv backward: {N} output channels are not a multiple of 8")
        dx = dw = None
        if ctx.needs_input_grad[1]:
            # the weight gradient is off the critical path (nothing downstream of it until the step's finishing launch):
            # with a deferred destination it is forked onto a side stream BEFORE the data gradient is issued and runs next
            # to the data-gradient chain -- inside a captured step that is a fork / join of the graph
            if ctx.sink_w is not None and WGRAD_DEFER and getattr(weight, "defer_wgrad", False):
                # (the DepthNet decoder: issued later, all at once, next to the encoder's backward -- _flush_deferred)
                _deferred.append((ctx, g, gpitch, x))
                side = None
            else:
                side = _wgrad_side_stream(g.device) if (ctx.sink_w is not None and WGRAD_SIDE_STREAM) else None
            if _deferred and _deferred[-1][0] is ctx:
                pass
            elif side is not None:
                side.wait_stream(torch.cuda.current_stream())
                g.record_stream(side)
                x.record_stream(side)
                _ops.grad_sink.join_streams.add(side)
            if not (_deferred and _deferred[-1][0] is ctx):
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    dw = _weight_grad(ctx, lib, g, gpitch, x)
        if ctx.needs_input_grad[0]:
            e = packer.get(weight, need_bwd=True)
            dx = torch.empty((B, Cp, PH, PW), dtype=_lib.half(), device=g.device, memory_format=torch.channels_last)
            wsf = lib.xpt_conv2d_splitk_workspace_floats(B * (PH << ups) * (PW << ups), Cp, e["Np"], KH * KW, stride)
            if wsf:
                ws = torch.empty(wsf, dtype=torch.float32, device=g.device)
                _lib.check(lib.xpt_conv2d_bwd_data_splitk(g.data_ptr(), e["bwd"].data_ptr(), dx.data_ptr(), B, OH, OW, e["Np"],
                                                          gpitch, Cp, KH, KW, pt, pl, PH, PW, Cp, ups, ws.data_ptr(), wsf,
                                                          _stream()), "xpt_conv2d_bwd_data_splitk")
            elif KH == 3 and lib.xpt_conv2d_stream_serves(B, PH << ups, PW << ups, Cp, e["Np"], KH, KW, stride, ups):
                _lib.check(lib.xpt_conv2d_bwd_data_stream(g.data_ptr(), e["bwd"].data_ptr(), dx.data_ptr(), B, OH, OW, e["Np"],
                                                          gpitch, Cp, pt, pl, PH, PW, Cp, ups, _stream()),
                           "xpt_conv2d_bwd_data_stream")
            else:
                _lib.check(lib.xpt_conv2d_bwd_data(g.data_ptr(), e["bwd"].data_ptr(), dx.data_ptr(), B, OH, OW, e["Np"], gpitch,
                                                   Cp, KH, KW, stride, pt, pl, PH, PW, Cp, ups, _stream()), "xpt_conv2d_bwd_data")
        if _deferred and getattr(weight, "flush_wgrads", False):
            _flush_deferred(True)
        return dx, dw, dbias, None, None, None, None


# The decoder's weight gradients as ONE parallel branch of the step: a fork per layer costs more than it returns (below), so the
# layers tagged `weight.defer_wgrad` only queue (ctx, g, x) during their backward; the layer tagged `weight.flush_wgrads` (the
# decoder's first: its backward runs last) issues the whole queue on the side stream behind ONE fork edge, next to the encoder's
# launch-latency-bound backward, and the gradient sink joins the stream before its finishing launch.
# Measured (round 4, bit-identical losses): 4.11 -> 4.20 ms/step -- the persistent weight-gradient workgroups hold the CUs the
# encoder's short launches are waiting for, the critical chain loses more than the branch saves.  Opt-in: XPT_WGRAD_DEFER=1.
WGRAD_DEFER = __import__("os").environ.get("XPT_WGRAD_DEFER", "0") == "1"
_deferred = []


def _flush_deferred(to_side):
    """Issue the queued weight-gradient launches (on the side stream, or -- the sink's safety net -- where we are)."""
    global _deferred
    queue, _deferred = _deferred, []
    if not queue:
        return
    lib = _lib.load()
    side = _wgrad_side_stream(queue[0][1].device) if to_side else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        _ops.grad_sink.join_streams.add(side)
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        for ctx, g, gpitch, x in queue:
            if side is not None:
                g.record_stream(side)
                x.record_stream(side)
            _weight_grad(ctx, lib, g, gpitch, x)


# measured: the fork / join edges cost more than the overlap returns inside the captured step (10.53 -> 11.9 ms), so off
WGRAD_SIDE_STREAM = __import__("os").environ.get("XPT_WGRAD_SIDE_STREAM", "0") == "1"
_SIDE = {}


def _wgrad_side_stream(device):
    if device not in _SIDE:
        _SIDE[device] = torch.cuda.Stream(device=device)
    return _SIDE[device]


def _weight_grad(ctx, lib, g, gpitch, x):
    B, PH, PW, Cp, C, xpitch, N, KH, KW, stride, pt, pl, OH, OW, ups, slope = ctx.geom
    nsplit = lib.xpt_conv2d_bwd_weight_splits(B, Cp, N, KH, KW, stride, OH, OW)
    if nsplit < 1:
        raise _lib.XptHipError(f"xpt_conv2d_bwd_weight_splits failed: {nsplit}")
    n = N * KH * KW * C
    if ctx.sink_w is not None:
        ws = _ops.grad_sink.partials(ctx.sink_w, "convw", nsplit * n)
    else:
        ws = torch.empty(nsplit * n, dtype=torch.float32, device=g.device)
    _lib.check(lib.xpt_conv2d_bwd_weight_partials(g.data_ptr(), x.data_ptr(), ws.data_ptr(), ws.numel(), B, PH, PW, Cp, C,
                                                  xpitch, N, gpitch, KH, KW, stride, pt, pl, OH, OW, ups, _stream()),
               "xpt_conv2d_bwd_weight_partials")
    if ctx.sink_w is not None:
        _ops.grad_sink.add(ctx.sink_w, ws, 0, n, nsplit, n)
        return None
    # no deferred destination (a weight outside FlatParameters, i.e. no optimizer bound: eager unit tests).  torch.sum must
    # not end up inside a captured step -- its multi-block mode zeroes a semaphore with a memset node, which this runtime
    # replays as garbage -- so a capture without a deferred destination is refused
    if torch.cuda.is_current_stream_capturing():
        raise _lib.XptHipError("conv weight gradient without a flat-gradient destination inside a graph capture")
    return ws[:nsplit * n].view(nsplit, N, KH, KW, C).sum(0).permute(0, 3, 1, 2)


def conv2d_same(x, weight, bias, stride=1, slope=1.0, upsample=False, valid=False):
    """x [B,Cp,H,W] bf16 (NCHW-indexed, NHWC-stored; Cp = weight's input channels rounded up to 8, pad channels zero),
    weight [N,C,KH,KW] float32 master, bias [N] float32 or None -> [B,N,ceil(2^u H / stride),ceil(2^u W / stride)] bf16."""
    return _Conv2dSame.apply(x, weight, bias, int(stride), float(slope), bool(upsample), bool(valid))


def usable(x, conv, slope):
    """Can this Conv2DSame call run on the matrix-core kernels?  (bf16 activations on the GPU, dense, undilated, at most
    13 taps per wave group: k <= 5; heads with a single output channel go through their own kernels.)"""
    if not x.is_cuda or slope is None or conv.groups != 1 or conv.dilation != (1, 1):
        return False
    dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
    k = conv.kernel_size
    return dtype == _lib.half() and k[0] == k[1] and k[0] in (1, 3, 5) and conv.out_channels % 8 == 0 \
        and conv.weight.dtype == torch.float32


def stem_input_usable(image):
    """[B,3,H,W] float32 view of NHWC frames (channel stride 1, pixel stride 3) on the GPU under bf16 autocast."""
    if not (torch.is_tensor(image) and image.is_cuda and image.dtype == torch.float32 and image.dim() == 4
            and image.shape[1] == 3 and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == _lib.half()):
        return False
    sb, sc, sh, sw = image.stride()
    return sc == 1 and sw == 3 and sh == 3 * image.shape[3] and sb >= 3 * image.shape[2] * image.shape[3]


def stem_input(image):
    """PretrainedModel's preprocessing + the stem convolution's input layout in one launch (csrc/xpt_augment.hip
    stem_input_kernel): [B,3,H,W] float32 (NHWC view) -> [B,8,H+2,W+2] bf16 channels_last, channels 3..7 zero."""
    lib = _lib.load()
    B, _, H, W = image.shape
    out = torch.empty((B, 8, H + 2, W + 2), dtype=_lib.half(), device=image.device, memory_format=torch.channels_last)
    _lib.check(lib.xpt_stem_input(image.data_ptr(), image.stride(0), out.data_ptr(), B, H, W, _stream()), "xpt_stem_input")
    return out


def restack_bf16(image5d, channels_padded):
    """restack_on_channels (pose_net.py:44-50) + cast: [B,S,H,W,3] float32 -> [B,Cp,H,W] bf16 (NHWC storage)."""
    lib = _lib.load()
    if not image5d.is_cuda or image5d.dtype != torch.float32:
        raise _lib.XptHipError("restack_bf16: expected a float32 CUDA/HIP tensor")
    image5d = image5d.contiguous()
    B, S, H, W, C = image5d.shape
    if C != 3:
        raise _lib.XptHipError("restack_bf16: expected 3 channels per frame")
    out = torch.empty((B, H, W, channels_padded), dtype=_lib.half(), device=image5d.device)
    _lib.check(lib.xpt_restack_bf16(image5d.data_ptr(), out.data_ptr(), B, S, H, W, channels_padded, _stream()),
               "xpt_restack_bf16")
    return out.permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------- decoder prediction heads (1 output channel)
class _HeadConv(torch.autograd.Function):
    """pre [B,1,H,W] fp32 = conv3x3_same(x bf16, weight [1,C,3,3]) + bias (csrc/xpt_headconv.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x, xpitch = nhwc_view(x)
        B, C, H, W = x.shape
        w = weight.detach()
        if w.dtype != torch.float32 or tuple(w.shape) != (1, C, 3, 3):
            raise _lib.XptHipError(f"head_conv: weight must be float32 [1, {C}, 3, 3], got {tuple(w.shape)} {w.dtype}")
        if not (w.stride(1) == 1 and w.stride(3) == C and w.stride(2) == 3 * C):     # [kh][kw][C] in memory
            w = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        b_ = None if bias is None else bias.detach()
        pre = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
        _lib.check(lib.xpt_headconv_fwd(x.data_ptr(), xpitch, w.data_ptr(), None if b_ is None else b_.data_ptr(),
                                        pre.data_ptr(), B, H, W, C, _stream()), "xpt_headconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.geom = (B, C, H, W, xpitch)
        ctx.sink = None
        if _ops.grad_sink.wants(weight) and weight.flat_grad.numel() == 9 * C and (bias is None or _ops.grad_sink.wants(bias)):
            ctx.sink = (weight.flat_grad, None if bias is None else bias.flat_grad)
        ctx.has_bias = bias is not None
        return pre

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        B, C, H, W, xpitch = ctx.geom
        g = g.contiguous().float()
        dx = torch.empty((B, C, H, W), dtype=_lib.half(), device=g.device, memory_format=torch.channels_last)
        nblk = lib.xpt_headconv_bwd_blocks(B, H, W, C)
        row = 9 * C + 1
        if ctx.sink is not None:
            ws = _ops.grad_sink.partials(ctx.sink[0], "head", nblk * row)
        else:
            ws = torch.empty(nblk * row, dtype=torch.float32, device=g.device)
        _lib.check(lib.xpt_headconv_bwd(x.data_ptr(), xpitch, w.data_ptr(), g.data_ptr(), dx.data_ptr(), ws.data_ptr(),
                                        ws.numel(), B, H, W, C, _stream()), "xpt_headconv_bwd")
        if ctx.sink is not None:
            _ops.grad_sink.add(ctx.sink[0], ws, 0, 9 * C, nblk, row)
            if ctx.sink[1] is not None:
                _ops.grad_sink.add(ctx.sink[1], ws, 9 * C, 1, nblk, row)
            return dx, None, None
        tot = ws[:nblk * row].view(nblk, row).sum(0)
        dw = tot[:9 * C].view(1, 3, 3, C).permute(0, 3, 1, 2)
        return dx, dw, (tot[9 * C:].clone() if ctx.has_bias else None)


class _HeadConvSplit(_HeadConv):
    """(pre, x itself): the decoder's feature map feeds its prediction head AND the next decoder level (depth_net.py:137-167);
    as one autograd node the next level's gradient is added inside the head's backward launch (xpt_headconv_bwd_add)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        pre = _HeadConv.forward(ctx, x, weight, bias)
        ctx.set_materialize_grads(False)
        return pre, x.view_as(x)

    @staticmethod
    def backward(ctx, g, g_x):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        B, C, H, W, xpitch = ctx.geom
        if g is None:
            return g_x, None, None
        g = g.contiguous().float()
        add, apitch = (None, 0)
        if g_x is not None:
            add, apitch = nhwc_view(g_x.to(_lib.half()))
        dx = torch.empty((B, C, H, W), dtype=_lib.half(), device=g.device, memory_format=torch.channels_last)
        nblk = lib.xpt_headconv_bwd_blocks(B, H, W, C)
        row = 9 * C + 1
        if ctx.sink is not None:
            ws = _ops.grad_sink.partials(ctx.sink[0], "head", nblk * row)
        else:
            ws = torch.empty(nblk * row, dtype=torch.float32, device=g.device)
        _lib.check(lib.xpt_headconv_bwd_add(x.data_ptr(), xpitch, w.data_ptr(), g.data_ptr(), None if add is None else add.data_ptr(),
                                            apitch, dx.data_ptr(), ws.data_ptr(), ws.numel(), B, H, W, C, _stream()),
                   "xpt_headconv_bwd_add")
        if ctx.sink is not None:
            _ops.grad_sink.add(ctx.sink[0], ws, 0, 9 * C, nblk, row)
            if ctx.sink[1] is not None:
                _ops.grad_sink.add(ctx.sink[1], ws, 9 * C, 1, nblk, row)
            return dx, None, None
        tot = ws[:nblk * row].view(nblk, row).sum(0)
        dw = tot[:9 * C].view(1, 3, 3, C).permute(0, 3, 1, 2)
        return dx, dw, (tot[9 * C:].clone() if ctx.has_bias else None)


def head_conv_split(x, weight, bias):
    """(head_conv(x, weight, bias), x as an alias for the feature map's other consumer); see _HeadConvSplit."""
    return _HeadConvSplit.apply(x, weight, bias)


def head_conv(x, weight, bias):
    """Conv2D(1, 3, padding="same", linear) of get_scaled_depth (depth_net.py:87-92): x [B,C,H,W] bf16 (NHWC storage,
    C in 16/32/64/128) -> [B,1,H,W] float32."""
    return _HeadConv.apply(x, weight, bias)


def head_usable(x, conv):
    return x.is_cuda and x.dtype == _lib.half() and conv.in_channels in (16, 32, 64, 128) and conv.out_channels == 1 \
        and conv.kernel_size == (3, 3) and conv.weight.dtype == torch.float32 and conv.dilation == (1, 1) and conv.stride == (1, 1)


_ops.grad_sink.pre_flush.append(lambda: _flush_deferred(False))
