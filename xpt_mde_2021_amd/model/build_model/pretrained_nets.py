"""Multi-scale encoder behind DepthNetPretrained (reference: model/build_model/pretrained_nets.py:11-117).

The reference takes `tf.keras.applications.NASNetMobile(include_top=False)` from tensorflow==2.4.1 (a
third-party dependency that is NOT part of the reference checkout) and taps the five layers listed in
scaled_layers.json ("NASNetMobile": activation_7 / _18 / _77 / _136 / _187 at 1/2 ... 1/32).  This file
restates the published NASNet-A (4 @ 1056) architecture of that Keras application as torch modules.

PARITY UNPINNED against TensorFlow itself: no golden activations or ImageNet weights exist offline.  What is pinned:
(i) creation-order index of the unnamed Activation layers reproduces exactly the five tap names above with the spatial
sizes scaled_layers.json records; (ii) the parameter count equals Keras' published 4,269,716 (include_top=False)
(tests/test_nasnet_pins.py); (iii) an independent second restatement in the framework's own conventions
(oracle/ref_nasnet.py: NHWC, HWIO, Keras variable names) yields the same five taps on the same weights, on the CPU and
through the gfx950 kernels (tests/test_ref_nasnet.py); (iv) every Keras variable of the no-top model lands on exactly one
parameter / buffer of this module (load_keras_weights below, tests/golden/nasnet_mobile_manifest.json).

Bug-compatible details kept (SURVEY 7 "Hard parts" h, i):
  * `preprocess_input` (x/127.5 - 1) is applied to images that are ALREADY in [-1, 1] (pretrained_nets.py:40),
    then the image is resized to (H+2, W+2) so that the VALID 3x3/2 stem conv yields H/2 x W/2 (:32, :41);
  * the model is called without training=True (train_val.py:82), so every BatchNormalization uses its moving
    statistics (never updated) while gamma / beta are trained: FrozenBatchNorm below.
"""
import torch
import torch.nn as nn
from ...hip.lib import half as _half      # torch dtype of the 16-bit activations (bf16 | fp16 build of the library)

import torch.nn.functional as F

from ...hip import ops as _ops
from ...utils.util_class import WrongInputException
from ...hip import conv as _conv
from ..model_util.layer_ops import conv2d_library, same_pad

_FUSED_STEM_RELU = __import__("os").environ.get("XPT_DEBUG_STEM_RELU", "1") == "1"         # A/B: the stem's ReLU inside its BatchNorm launch
_LIBRARY_WGRAD = __import__("os").environ.get("XPT_DEBUG_LIBRARY_WGRAD", "0") == "1"     # A/B switch: rocBLAS weight gradient
_DISABLE_HIP_DWCONV = bool(int(__import__("os").environ.get("XPT_DEBUG_MIOPEN_DWCONV", "0")))   # A/B debugging only
BN_EPS = 1e-3          # keras_applications nasnet: BatchNormalization(momentum=0.9997, epsilon=1e-3)


class FrozenBatchNorm(nn.Module):
    """Keras BatchNormalization in inference mode with trainable gamma / beta."""

    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))

    def forward(self, x, residual=None, relu_out=False):
        """residual: added to the normalised output -- the `layers.add` that joins two branches of a NASNet cell,
        fused into the epilogue kernel of the branch that ends in this BatchNorm.  relu_out: the Activation('relu') that
        follows rides in the same launch (forward and backward)."""
        if x.is_cuda:      # one gfx950 streaming pass (and one for dx / dgamma / dbeta in the backward)
            return _ops.batchnorm_inference(x, self.weight, self.bias, self.running_mean, self.running_var, BN_EPS,
                                            residual=residual, slope=0.0 if relu_out else 1.0)
        y = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, BN_EPS)
        y = y if residual is None else y + residual
        return F.relu(y) if relu_out else y


def shared_relu(x):
    """relu(x), computed once per tensor: a cell output is the `ip` of the next cell and the `p` of the one after,
    and both start with Activation('relu') on it."""
    pool = getattr(x, "_xpt_relu_aliases", None)
    if pool is not None:               # a cell output produced already rectified (cell_tail): one alias per consumer
        return pool.pop() if pool else x
    cached = getattr(x, "_xpt_relu", None)
    if cached is None:
        cached = F.relu(x)
        # (never on a leaf that requires grad: relu(x) -> ReluBackward -> AccumulateGrad -> x -> its attribute -> relu(x) is a
        #  cycle through C++ that neither side's collector sees -- the graph of every step would stay alive)
        if x.is_cuda and not (x.requires_grad and x.grad_fn is None):
            x._xpt_relu = cached
    return cached


class _Conv1x1Bf16(torch.autograd.Function):
    """1x1 convolution as GEMMs on the bf16 shadow copy of the weight (kept current by the fused Adam kernel):
    y = x W^T and dx = dy W in bf16 through rocBLAS; dW = dy^T x by the split-K matrix-core kernel of
    csrc/xpt_gemm.hip (exact products, fp32 accumulation and output).  Operands are [pixels, channels] views with a
    row pitch, so channel slices (torch.cat's backward) are consumed without a copy."""

    @staticmethod
    def forward(ctx, x, weight):
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        ws = weight.shadow_bf16.reshape(cout, cin)
        x2 = _ops.as_rows(x)
        y2 = torch.mm(x2, ws.t())
        ctx.save_for_backward(x2, ws)
        ctx.dims = (B, cin, H, W, cout, weight.shape)
        ctx.sink_dst = weight.flat_grad if (_ops.grad_sink.wants(weight) and not _LIBRARY_WGRAD) else None
        return y2.view(B, H, W, cout).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        x2, ws = ctx.saved_tensors
        B, cin, H, W, cout, wshape = ctx.dims
        dy2 = _ops.as_rows(dy.to(_half()))
        dx = dw = None
        if (_FUSED_DGRAD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and ctx.sink_dst is not None
                and ws.is_contiguous() and _ops.vector_rows(dy2, cout) and _ops.vector_rows(x2, cin)):
            # weight-gradient partials and the data gradient in ONE launch (extra workgroups; no GEMM launch)
            dx = _ops.conv1x1_weight_grad_deferred(dy2, x2, ctx.sink_dst, ws).view(B, H, W, cin).permute(0, 3, 1, 2)
            return dx, None
        if ctx.needs_input_grad[0]:
            dx = torch.mm(dy2, ws).view(B, H, W, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1] and ctx.sink_dst is not None:
            _ops.conv1x1_weight_grad_deferred(dy2, x2, ctx.sink_dst)      # finished by grad_sink.flush()
        elif ctx.needs_input_grad[1]:
            if _LIBRARY_WGRAD:
                dw = torch.mm(dy2.t(), x2, out_dtype=torch.float32).view(wshape)
            else:
                dw = _ops.conv1x1_weight_grad(dy2, x2).view(wshape)
        return dx, dw


def conv1x1(x, weight):
    """1x1 convolution (no bias) of an NCHW-indexed tensor.  On the GPU the channels_last tensor IS a row-major
    [B*H*W, Cin] matrix, so the convolution is one GEMM (rocBLAS on MFMA) on a zero-copy view and its weight
    gradient one more GEMM -- no MIOpen convolution (nor its zero / cast helper launches) involved."""
    if not x.is_cuda:
        return F.conv2d(x, weight)
    if x.dtype == _half() and hasattr(weight, "shadow_bf16") and torch.is_autocast_enabled():
        return _Conv1x1Bf16.apply(x, weight)
    xv = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)            # [B,H,W,Cin] view
    y = F.linear(xv, weight.reshape(weight.shape[0], weight.shape[1]))                  # [B,H,W,Cout]
    return y.permute(0, 3, 1, 2)                                                        # NCHW-indexed, channels_last


class _Conv1x1BnBf16(torch.autograd.Function):
    """BatchNorm(conv1x1(x)) [+ residual] with the BatchNorm backward folded into the convolution's weight-gradient
    launch (csrc/xpt_gemm.hip, BnFuse): backward = ONE gfx950 launch (g = dy * s for the data-gradient GEMM, split-K
    partials of dW, dgamma, dbeta -> GradSink) + ONE rocBLAS GEMM, instead of BN-backward kernel + weight-gradient
    kernel + GEMM.  Only used when all three parameters are deferred-gradient views (training on flat buffers)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mean, var, eps, residual, n_alias=1):
        """n_alias > 1 (no residual): n aliases of y, one per consumer -- the backward then sees the consumers' gradients
        separately and the weight-gradient kernel adds them on load (no gradient fan-in launch)."""
        lib = _ops._lib.load()
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        ws = weight.shadow_bf16.reshape(cout, cin)
        x2 = _ops.as_rows(x)
        M = x2.shape[0]
        pitch_x = x2.stride(0) if M > 1 else cin
        y = torch.empty((B, cout, H, W), dtype=_half(), device=x.device, memory_format=torch.channels_last)
        if residual is not None:
            residual = residual.to(_half()).contiguous(memory_format=torch.channels_last)
        g_, b_ = gamma.detach(), beta.detach()
        # deep reductions on small maps (cin >= 256, few tiles) split the k loop over the 4 waves of a workgroup inside
        # the kernel (xpt_pwconv.hip, KW = 4): with that the fused launch beats GEMM + epilogue launch up to the widest
        # layer (1056 channels) too
        if not _LIBRARY_PWCONV and cin <= _PWCONV_MAX_CIN:
            # GEMM + BatchNorm (+ branch add) in one gfx950 launch (csrc/xpt_pwconv.hip)
            ypre = torch.empty((M, cout), dtype=_half(), device=x.device)
            _ops._lib.check(lib.xpt_pwconv_bn_fwd(x2.data_ptr(), ws.data_ptr(), g_.data_ptr(), b_.data_ptr(),
                                                  mean.data_ptr(), var.data_ptr(), float(eps), _ops._ptr(residual),
                                                  ypre.data_ptr(), y.data_ptr(), M, cin, cout, pitch_x, _ops._stream()),
                            "xpt_pwconv_bn_fwd")
        else:
            ypre = torch.mm(x2, ws.t())                                 # [M, cout] bf16, dense
            _ops._lib.check(lib.xpt_affine_act_fwd(ypre.data_ptr(), g_.data_ptr(), b_.data_ptr(), mean.data_ptr(),
                                                   var.data_ptr(), float(eps), _ops._ptr(residual), y.data_ptr(), M,
                                                   cout, 1.0, 0, 1, _ops._stream()), "xpt_affine_act_fwd")
        ctx.save_for_backward(x2, ws, ypre, g_, mean, var)
        ctx.dims = (B, cin, H, W, cout, float(eps))
        ctx.dst = (weight.flat_grad, gamma.flat_grad, beta.flat_grad)
        if n_alias > 1:
            if residual is not None:
                raise WrongInputException("conv1x1_bn: aliases and a residual cannot be combined")
            ctx.set_materialize_grads(False)
            return tuple(y.view_as(y) for _ in range(n_alias))
        return y

    @staticmethod
    def backward(ctx, *dys):
        x2, ws, ypre, gamma, mean, var = ctx.saved_tensors
        live = [d for d in dys if d is not None]
        if not live:
            return (None,) * 9
        if len(live) > 3:                                     # the kernel adds up to three pieces: more go through
            live = [_ops.sum_rows(live)]                      # the fan-in kernel (same single rounding)
        dx = _conv_bn_backward(x2, ws, ypre, gamma, mean, var, ctx.dims, ctx.dst, live[0], ctx.needs_input_grad[0],
                               extra=live[1:])
        dres = live[0] if ctx.needs_input_grad[7] else None
        return dx, None, None, None, None, None, None, dres, None


def _conv_bn_backward(x2, ws, ypre, gamma, mean, var, dims, dst, dy, need_dx, extra=()):
    """One gfx950 launch (g = dy * s, split-K partials of dW / dgamma / dbeta -> GradSink) + one rocBLAS GEMM (dx).
    extra: up to two more pieces of the output gradient (added to dy on load by the kernel)."""
    lib = _ops._lib.load()
    B, cin, H, W, cout, eps = dims
    w_dst, g_dst, b_dst = dst
    dy2 = _ops.as_rows(dy.to(_half()))
    more = [_ops.as_rows(e.to(_half())) for e in extra]
    M = dy2.shape[0]
    nsplit = lib.xpt_conv1x1_bwd_weight_splits(M, cout, cin)
    sink = _ops.grad_sink
    wpart = sink.partials(w_dst, "conv1x1", nsplit * cout * cin)
    bpart = sink.partials(b_dst, "bnfuse", nsplit * 2 * cout)
    pitch_dy = dy2.stride(0) if M > 1 else cout
    pitch_x = x2.stride(0) if M > 1 else cin
    pitch_of = lambda t: t.stride(0) if M > 1 else cout           # noqa: E731
    e1 = more[0] if len(more) > 0 else None
    e2 = more[1] if len(more) > 1 else None
    even = lambda t, pitch: pitch % 2 == 0 and t.data_ptr() % 4 == 0           # noqa: E731
    vector_rows = (cout % 2 == 0 and cin % 2 == 0 and even(dy2, pitch_dy) and even(x2, pitch_x)
                   and all(even(e, pitch_of(e)) for e in more))
    if _FUSED_DGRAD and ws.is_contiguous() and vector_rows:     # (odd widths: scalar staging, no data-gradient workgroups)
        # the data gradient rides in the same launch (extra workgroups of the weight-gradient kernel): no g, no GEMM launch
        dx = torch.empty((M, cin), dtype=_half(), device=dy.device) if need_dx else None
        _ops._lib.check(lib.xpt_conv1x1_bn_bwd_fused(dy2.data_ptr(), _ops._ptr(e1), _ops._ptr(e2), ypre.data_ptr(),
                                                     x2.data_ptr(), ws.data_ptr(), gamma.data_ptr(), var.data_ptr(),
                                                     mean.data_ptr(), eps, _ops._ptr(dx), wpart.data_ptr(), wpart.numel(),
                                                     bpart.data_ptr(), bpart.numel(), M, cout, cin, pitch_dy,
                                                     0 if e1 is None else pitch_of(e1), 0 if e2 is None else pitch_of(e2),
                                                     pitch_x, _ops._stream()), "xpt_conv1x1_bn_bwd_fused")
        sink.add(w_dst, wpart, 0, cout * cin, nsplit, cout * cin)
        sink.add(b_dst, bpart, 0, cout, nsplit, 2 * cout)
        sink.add(g_dst, bpart, cout, cout, nsplit, 2 * cout)
        return dx.view(B, H, W, cin).permute(0, 3, 1, 2) if need_dx else None
    g = torch.empty((M, cout), dtype=_half(), device=dy.device)
    _ops._lib.check(lib.xpt_conv1x1_bn_bwd_partials_sum(dy2.data_ptr(), _ops._ptr(e1), _ops._ptr(e2), ypre.data_ptr(),
                                                        x2.data_ptr(), gamma.data_ptr(), var.data_ptr(), mean.data_ptr(),
                                                        eps, g.data_ptr(), wpart.data_ptr(), wpart.numel(),
                                                        bpart.data_ptr(), bpart.numel(), M, cout, cin, pitch_dy,
                                                        0 if e1 is None else pitch_of(e1), 0 if e2 is None else pitch_of(e2),
                                                        pitch_x, _ops._stream()), "xpt_conv1x1_bn_bwd_partials_sum")
    sink.add(w_dst, wpart, 0, cout * cin, nsplit, cout * cin)
    sink.add(b_dst, bpart, 0, cout, nsplit, 2 * cout)
    sink.add(g_dst, bpart, cout, cout, nsplit, 2 * cout)
    return torch.mm(g, ws).view(B, H, W, cin).permute(0, 3, 1, 2) if need_dx else None


class _MultiConv1x1Bn(torch.autograd.Function):
    """n independent conv1x1 + BatchNorm (+ residual) layers of one shape: ONE forward launch (xpt_pwconv_bn_multi_fwd) and
    ONE backward launch.  Optionally some of them have a SIBLING: a further layer of the same shape whose BatchNorm output is
    added to theirs (the two operands of a NASNet cell's `add`), evaluated inside the same launch
    (xpt_pwconv_bn_multi_fwd_sib); the backward then runs all n + ns layers as one launch, a sibling with its main layer's
    output gradient.  args = (n, eps, sib_of, xs..., weights..., gammas..., betas..., means..., vars... (n + ns each, the
    siblings behind the main layers), residuals-or-None... (n)); sib_of[k] = the main layer sibling k adds to."""

    precomputed = None      # (ypres, ys) a fused branch-stage launch already produced (fused_sep_stage): consumed by the next forward

    @staticmethod
    def forward(ctx, n, eps, sib_of, *a):
        import ctypes
        lib = _ops._lib.load()
        pre, _MultiConv1x1Bn.precomputed = _MultiConv1x1Bn.precomputed, None
        L = n + len(sib_of)
        xs, ws_, gs, bs, ms, vs = (a[i * L:(i + 1) * L] for i in range(6))
        rs = a[6 * L:6 * L + n]
        B, cin, H, W = xs[0].shape
        cout = ws_[0].shape[0]
        x2s = [_ops.as_rows(x) for x in xs]
        M = x2s[0].shape[0]
        pitch = x2s[0].stride(0) if M > 1 else cin
        shadows = [w.shadow_bf16.reshape(cout, cin) for w in ws_]
        res = [None if r is None else r.to(_half()).contiguous(memory_format=torch.channels_last) for r in rs]
        if pre is not None:                    # the fused branch-stage launch computed this stage already
            ypres, ys = list(pre[0]), list(pre[1])
        else:
            ypres = [torch.empty((M, cout), dtype=_half(), device=xs[0].device) for _ in range(L)]
            ys = [torch.empty((B, cout, H, W), dtype=_half(), device=xs[0].device, memory_format=torch.channels_last)
                  for _ in range(n)]
            P = ctypes.c_void_p * n
            ptr = lambda ts: P(*[None if t is None else t.data_ptr() for t in ts])           # noqa: E731
            det = lambda ts: [None if t is None else t.detach() for t in ts]                # noqa: E731
            main = (ptr(x2s[:n]), ptr(shadows[:n]), ptr(det(gs[:n])), ptr(det(bs[:n])), ptr(ms[:n]), ptr(vs[:n]), float(eps),
                    ptr(res), ptr(ypres[:n]), ptr(ys))
            if sib_of:
                at = {j: n + k for k, j in enumerate(sib_of)}                                # main layer -> its sibling
                pick = lambda ts: [ts[at[j]] if j in at else None for j in range(n)]         # noqa: E731
                _ops._lib.check(lib.xpt_pwconv_bn_multi_fwd_sib(
                    n, *main, ptr(pick(x2s)), ptr(pick(shadows)), ptr(det(pick(gs))), ptr(det(pick(bs))), ptr(pick(ms)),
                    ptr(pick(vs)), ptr(pick(ypres)), M, cin, cout, pitch, 0, _ops._stream()), "xpt_pwconv_bn_multi_fwd_sib")
            else:
                _ops._lib.check(lib.xpt_pwconv_bn_multi_fwd(n, *main, M, cin, cout, pitch, _ops._stream()),
                                "xpt_pwconv_bn_multi_fwd")
        ctx.save_for_backward(*x2s, *shadows, *ypres, *[g.detach() for g in gs], *ms, *vs)
        ctx.n, ctx.sib_of = n, tuple(sib_of)
        ctx.dims = (B, cin, H, W, cout, float(eps))
        ctx.dsts = [(w.flat_grad, g.flat_grad, b.flat_grad) for w, g, b in zip(ws_, gs, bs)]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        import ctypes
        n, sib_of = ctx.n, ctx.sib_of
        L = n + len(sib_of)
        t = ctx.saved_tensors
        x2s, shadows, ypres, gs, ms, vs = (t[i * L:(i + 1) * L] for i in range(6))
        B, cin, H, W, cout, eps = ctx.dims
        none = [None] * L
        dres = [dys[j] if (dys[j] is not None and ctx.needs_input_grad[3 + 6 * L + j]) else None for j in range(n)]
        dys = list(dys) + [dys[j] for j in sib_of]            # a sibling sees its main layer's output gradient
        need = [ctx.needs_input_grad[3 + j] for j in range(L)]
        if any(d is None for d in dys):                      # a branch without gradient: per-layer path
            dxs = [None if dys[j] is None else
                   _conv_bn_backward(x2s[j], shadows[j], ypres[j], gs[j], ms[j], vs[j], ctx.dims, ctx.dsts[j], dys[j],
                                     need[j]) for j in range(L)]
            return (None, None, None, *dxs, *none, *none, *none, *none, *none, *dres)
        # ONE launch for the L layers: g_j = dy_j * s_j, split-K partials of dW_j / dgamma_j / dbeta_j
        lib = _ops._lib.load()
        sink = _ops.grad_sink
        dy2s = [_ops.as_rows(d.to(_half())) for d in dys]
        M = dy2s[0].shape[0]
        nsplit = lib.xpt_conv1x1_bwd_weight_splits(M, cout, cin)
        wparts, bparts = [], []
        for w_dst, g_dst, b_dst in ctx.dsts:
            wparts.append(sink.partials(w_dst, "conv1x1", nsplit * cout * cin))
            bparts.append(sink.partials(b_dst, "bnfuse", nsplit * 2 * cout))
        P, LL = ctypes.c_void_p * L, ctypes.c_longlong * L
        ptr = lambda ts: P(*[x.data_ptr() for x in ts])       # noqa: E731
        pitch_x = x2s[0].stride(0) if M > 1 else cin
        vector_rows = (cout % 2 == 0 and cin % 2 == 0 and pitch_x % 2 == 0
                       and all(d.data_ptr() % 4 == 0 and (d.stride(0) if M > 1 else cout) % 2 == 0 for d in dy2s)
                       and all(x.data_ptr() % 4 == 0 for x in x2s))
        if _FUSED_DGRAD and vector_rows and all(s.is_contiguous() for s in shadows):
            # ... and the L data gradients in the same launch (extra workgroups): no g, no batched GEMM launch
            dx_all = torch.empty((L, M, cin), dtype=_half(), device=dys[0].device)
            _ops._lib.check(lib.xpt_conv1x1_bn_multi_bwd_fused(
                L, ptr(dy2s), LL(*[(d.stride(0) if M > 1 else cout) for d in dy2s]), ptr(ypres), ptr(x2s), ptr(shadows),
                ptr(gs), ptr(vs), ptr(ms), eps, P(*[dx_all[j].data_ptr() if need[j] else None for j in range(L)]),
                ptr(wparts), ptr(bparts), wparts[0].numel(), bparts[0].numel(), M, cout, cin, pitch_x, _ops._stream()),
                "xpt_conv1x1_bn_multi_bwd_fused")
            for j, (w_dst, g_dst, b_dst) in enumerate(ctx.dsts):
                sink.add(w_dst, wparts[j], 0, cout * cin, nsplit, cout * cin)
                sink.add(b_dst, bparts[j], 0, cout, nsplit, 2 * cout)
                sink.add(g_dst, bparts[j], cout, cout, nsplit, 2 * cout)
            dxs = [dx_all[j].view(B, H, W, cin).permute(0, 3, 1, 2) if need[j] else None for j in range(L)]
            return (None, None, None, *dxs, *none, *none, *none, *none, *none, *dres)
        g_all = torch.empty((L, M, cout), dtype=_half(), device=dys[0].device)
        _ops._lib.check(lib.xpt_conv1x1_bn_multi_bwd_partials(
            L, ptr(dy2s), LL(*[(d.stride(0) if M > 1 else cout) for d in dy2s]), ptr(ypres), ptr(x2s), ptr(gs), ptr(vs),
            ptr(ms), eps, P(*[g_all[j].data_ptr() for j in range(L)]), ptr(wparts), ptr(bparts), wparts[0].numel(),
            bparts[0].numel(), M, cout, cin, pitch_x, _ops._stream()), "xpt_conv1x1_bn_multi_bwd_partials")
        for j, (w_dst, g_dst, b_dst) in enumerate(ctx.dsts):
            sink.add(w_dst, wparts[j], 0, cout * cin, nsplit, cout * cin)
            sink.add(b_dst, bparts[j], 0, cout, nsplit, 2 * cout)
            sink.add(g_dst, bparts[j], cout, cout, nsplit, 2 * cout)
        if L >= 2 and all(need) and (L >= 3 or _is_stacked(shadows)):
            # data gradients of all layers as ONE strided-batched GEMM: the g_j are already one [L, M, cout] buffer and the
            # weights sit equally spaced in the flat shadow buffer (FlatParameters groups them: stack_groups()), so the
            # [L, cout, cin] operand is a strided view; one stack launch otherwise
            dx_all = torch.bmm(g_all, _stacked_view(shadows))              # [L, M, cin]
            dxs = [dx_all[j].view(B, H, W, cin).permute(0, 3, 1, 2) for j in range(L)]
        else:
            dxs = [torch.mm(g_all[j], shadows[j]).view(B, H, W, cin).permute(0, 3, 1, 2) if need[j] else None
                   for j in range(L)]
        return (None, None, None, *dxs, *none, *none, *none, *none, *none, *dres)


def _is_stacked(mats):
    """True when the matrices sit equally spaced in one storage (no copy needed for the batched GEMM operand)."""
    first = mats[0]
    step = mats[1].data_ptr() - first.data_ptr()
    return (step > 0 and step % first.element_size() == 0
            and all(m.is_contiguous() and m.shape == first.shape for m in mats)
            and all(m.data_ptr() - first.data_ptr() == j * step for j, m in enumerate(mats))
            and all(m.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() for m in mats))


def _stacked_view(mats):
    """[n, r, c] over n equally spaced contiguous [r, c] matrices of one storage as a strided view (no copy)."""
    first = mats[0]
    step = mats[1].data_ptr() - first.data_ptr()
    esz = first.element_size()
    if (step > 0 and step % esz == 0 and all(m.is_contiguous() and m.shape == first.shape for m in mats)
            and all(m.data_ptr() - first.data_ptr() == j * step for j, m in enumerate(mats))
            and all(m.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() for m in mats)):
        return torch.as_strided(first, (len(mats), *first.shape), (step // esz, first.shape[1], 1))
    return torch.stack(list(mats))


def multi_conv1x1_bn(xs, weights, bns, residuals=None, siblings=None):
    """[bn_j(conv1x1(x_j, w_j)) (+ residual_j)] for layers of one shape: one forward launch when the fused path applies
    (same conditions as conv1x1_bn), the per-layer calls otherwise.  siblings[j] = (x, weight, bn) or None: a further layer
    of the same shape whose result is added to layer j's (evaluated inside the same launches, forward and backward)."""
    n = len(xs)
    residuals = [None] * n if residuals is None else residuals
    siblings = [None] * n if siblings is None else siblings
    sib_of = tuple(j for j in range(n) if siblings[j] is not None)
    sibs = [siblings[j] for j in sib_of]
    all_x = list(xs) + [q[0] for q in sibs]
    all_w = list(weights) + [q[1] for q in sibs]
    all_bn = list(bns) + [q[2] for q in sibs]
    sink = _ops.grad_sink
    x0, w0 = xs[0], weights[0]
    cin = w0.shape[1]
    ok = (_FUSE_CONV_BN and not _LIBRARY_PWCONV and not _LIBRARY_WGRAD and 1 < len(all_x) <= 6 and x0.is_cuda
          and x0.dtype == _half() and torch.is_autocast_enabled() and torch.is_grad_enabled()
          and cin <= _PWCONV_MAX_CIN
          and all(x.shape == x0.shape and x.dtype == x0.dtype for x in all_x)
          and all(w.shape == w0.shape and hasattr(w, "shadow_bf16") and sink.wants(w) for w in all_w)
          and all(sink.wants(b.weight) and sink.wants(b.bias) for b in all_bn))
    if ok:
        rows = [_ops.as_rows(x) for x in all_x]
        ok = all(r.stride(0) == rows[0].stride(0) for r in rows)
    if ok and sib_of and _MultiConv1x1Bn.precomputed is not None:
        ok = False
    if not ok:
        if sib_of:           # the siblings first, their outputs as residuals of the main layers (the same sums)
            if any(residuals[j] is not None for j in sib_of):
                raise WrongInputException("multi_conv1x1_bn: a layer with a sibling cannot take a residual as well")
            outs = multi_conv1x1_bn([q[0] for q in sibs], [q[1] for q in sibs], [q[2] for q in sibs]) if len(sibs) > 1 \
                else [conv1x1_bn(*sibs[0])]
            residuals = list(residuals)
            for j, r in zip(sib_of, outs):
                residuals[j] = r
            return multi_conv1x1_bn(xs, weights, bns, residuals)
        if n == 1:
            return [conv1x1_bn(xs[0], weights[0], bns[0], residuals[0])]
        return [conv1x1_bn(x, w, b, r) for x, w, b, r in zip(xs, weights, bns, residuals)]
    if any(residuals[j] is not None for j in sib_of):
        raise WrongInputException("multi_conv1x1_bn: a layer with a sibling cannot take a residual as well")
    return list(_MultiConv1x1Bn.apply(n, BN_EPS, sib_of, *all_x, *all_w, *[b.weight for b in all_bn], *[b.bias for b in all_bn],
                                      *[b.running_mean for b in all_bn], *[b.running_var for b in all_bn], *residuals))


class _PairConv1x1BnFan(torch.autograd.Function):
    """The two 1x1 convolution + BatchNorm heads of a NASNet normal cell (the squeeze of the cell input and the projection
    of the input of two cells ago: keras nasnet._normal_a_cell / _adjust_block) when both read tensors of one shape: ONE
    forward launch for the two layers and ONE backward launch.  Each output goes to several branches of the cell: it is
    handed out as aliases (one per consumer), and the consumers' gradients are added inside the backward launch
    (xpt_conv1x1_bn_multi_bwd_fused_fan) -- no fan-in launch.  args = (n_alias_a, n_alias_b, eps, xa, xb, wa, wb, gammas,
    betas, means, vars ...)."""

    @staticmethod
    def forward(ctx, na, nb, eps, xa, xb, wa, wb, ga, gb, ba, bb, ma, mb, va, vb):
        import ctypes
        lib = _ops._lib.load()
        B, cin, H, W = xa.shape
        cout = wa.shape[0]
        x2s = [_ops.as_rows(xa), _ops.as_rows(xb)]
        M = x2s[0].shape[0]
        pitch = x2s[0].stride(0) if M > 1 else cin
        shadows = [w.shadow_bf16.reshape(cout, cin) for w in (wa, wb)]
        ypres = [torch.empty((M, cout), dtype=_half(), device=xa.device) for _ in range(2)]
        ys = [torch.empty((B, cout, H, W), dtype=_half(), device=xa.device, memory_format=torch.channels_last)
              for _ in range(2)]
        P = ctypes.c_void_p * 2
        ptr = lambda ts: P(*[t.data_ptr() for t in ts])           # noqa: E731
        gs, bs = [ga.detach(), gb.detach()], [ba.detach(), bb.detach()]
        _ops._lib.check(lib.xpt_pwconv_bn_multi_fwd(2, ptr(x2s), ptr(shadows), ptr(gs), ptr(bs), ptr([ma, mb]), ptr([va, vb]),
                                                    float(eps), P(None, None), ptr(ypres), ptr(ys), M, cin, cout, pitch,
                                                    _ops._stream()), "xpt_pwconv_bn_multi_fwd")
        ctx.save_for_backward(*x2s, *shadows, *ypres, *gs, ma, mb, va, vb)
        ctx.na, ctx.nb = na, nb
        ctx.dims = (B, cin, H, W, cout, float(eps))
        ctx.dsts = [(wa.flat_grad, ga.flat_grad, ba.flat_grad), (wb.flat_grad, gb.flat_grad, bb.flat_grad)]
        ctx.set_materialize_grads(False)
        return tuple(ys[0].view_as(ys[0]) for _ in range(na)) + tuple(ys[1].view_as(ys[1]) for _ in range(nb))

    @staticmethod
    def backward(ctx, *dys):
        import ctypes
        t = ctx.saved_tensors
        x2s, shadows, ypres, gs, ms, vs = t[0:2], t[2:4], t[4:6], t[6:8], t[8:10], t[10:12]
        B, cin, H, W, cout, eps = ctx.dims
        pieces = [[d for d in dys[:ctx.na] if d is not None], [d for d in dys[ctx.na:] if d is not None]]
        none = (None,) * 15
        if not pieces[0] and not pieces[1]:
            return none
        need = [ctx.needs_input_grad[3], ctx.needs_input_grad[4]]
        if not pieces[0] or not pieces[1]:                   # one head without gradient: the per-layer launch for the other
            j = 0 if pieces[0] else 1
            live = pieces[j] if len(pieces[j]) <= 3 else [_ops.sum_rows(pieces[j])]
            dx = _conv_bn_backward(x2s[j], shadows[j], ypres[j], gs[j], ms[j], vs[j], ctx.dims, ctx.dsts[j], live[0], need[j],
                                   extra=live[1:])
            return (None, None, None) + ((dx, None) if j == 0 else (None, dx)) + (None,) * 10
        lib = _ops._lib.load()
        sink = _ops.grad_sink
        rows = []
        for j in range(2):
            live = pieces[j] if len(pieces[j]) <= 3 else [_ops.sum_rows(pieces[j])]   # (the kernel adds up to three pieces)
            rows.append([_ops.as_rows(d.to(_half())) for d in live])
        M = rows[0][0].shape[0]
        nsplit = lib.xpt_conv1x1_bwd_weight_splits(M, cout, cin)
        wparts = [sink.partials(d[0], "conv1x1", nsplit * cout * cin) for d in ctx.dsts]
        bparts = [sink.partials(d[2], "bnfuse", nsplit * 2 * cout) for d in ctx.dsts]
        P, LL = ctypes.c_void_p * 2, ctypes.c_longlong * 2
        pitch_of = lambda r: r.stride(0) if M > 1 else cout      # noqa: E731
        piece = lambda k: (P(*[(r[k].data_ptr() if len(r) > k else None) for r in rows]),      # noqa: E731
                           LL(*[(pitch_of(r[k]) if len(r) > k else 0) for r in rows]))
        pitch_x = x2s[0].stride(0) if M > 1 else cin
        dx_all = torch.empty((2, M, cin), dtype=_half(), device=rows[0][0].device)
        ptr = lambda ts: P(*[q.data_ptr() for q in ts])          # noqa: E731
        _ops._lib.check(lib.xpt_conv1x1_bn_multi_bwd_fused_fan(
            2, *piece(0), *piece(1), *piece(2), ptr(ypres), ptr(x2s), ptr(shadows), ptr(gs), ptr(vs), ptr(ms), eps,
            P(*[dx_all[j].data_ptr() if need[j] else None for j in range(2)]), ptr(wparts), ptr(bparts), wparts[0].numel(),
            bparts[0].numel(), M, cout, cin, pitch_x, _ops._stream()), "xpt_conv1x1_bn_multi_bwd_fused_fan")
        for j, (w_dst, g_dst, b_dst) in enumerate(ctx.dsts):
            sink.add(w_dst, wparts[j], 0, cout * cin, nsplit, cout * cin)
            sink.add(b_dst, bparts[j], 0, cout, nsplit, 2 * cout)
            sink.add(g_dst, bparts[j], cout, cout, nsplit, 2 * cout)
        dxs = [dx_all[j].view(B, H, W, cin).permute(0, 3, 1, 2) if need[j] else None for j in range(2)]
        return (None, None, None, dxs[0], dxs[1]) + (None,) * 10


_PAIR_HEADS = __import__("os").environ.get("XPT_DEBUG_SEPARATE_HEADS", "0") != "1"     # A/B: squeeze and projection as two launches


def pair_conv1x1_bn_usable(xa, xb, wa, wb, bna, bnb):
    sink = _ops.grad_sink
    if not (_PAIR_HEADS and _FUSE_CONV_BN and _FUSED_DGRAD and _FUSE_FAN_IN and not _LIBRARY_PWCONV and not _LIBRARY_WGRAD):
        return False
    if not (xa.is_cuda and xa.dtype == _half() and xb.dtype == _half() and xa.shape == xb.shape
            and torch.is_autocast_enabled() and torch.is_grad_enabled() and wa.shape == wb.shape):
        return False
    cout, cin = wa.shape[0], wa.shape[1]
    if cout % 2 or cin % 2 or cin > _PWCONV_MAX_CIN:
        return False
    if not all(hasattr(w, "shadow_bf16") and sink.wants(w) and w.shadow_bf16.is_contiguous() for w in (wa, wb)):
        return False
    if not all(sink.wants(b.weight) and sink.wants(b.bias) for b in (bna, bnb)):
        return False
    ra, rb = _ops.as_rows(xa), _ops.as_rows(xb)
    return ra.stride(0) == rb.stride(0) and ra.stride(0) % 2 == 0 and ra.data_ptr() % 4 == 0 and rb.data_ptr() % 4 == 0


class _SpatialAdjustBn(torch.autograd.Function):
    """bn(concat([conv1x1(p1, w1), conv1x1(p2, w2)])) of the "spatial" _adjust_block in ONE launch each way: the two
    convolutions are two jobs of the multi-layer pointwise kernel that write the two channel halves of one tensor (their
    BatchNorm parameters are the halves of the block's BatchNorm) -- instead of two GEMMs, a concat and a BatchNorm launch
    forward, and a BatchNorm-backward launch plus two weight-gradient launches backward."""

    @staticmethod
    def forward(ctx, p1, p2, w1, w2, gamma, beta, mean, var, eps, halves):
        import ctypes
        lib = _ops._lib.load()
        B, cin, H, W = p1.shape
        half = w1.shape[0]
        x2s = [_ops.as_rows(p1), _ops.as_rows(p2)]
        M = x2s[0].shape[0]
        pitch = x2s[0].stride(0) if M > 1 else cin
        shadows = [w.shadow_bf16.reshape(half, cin) for w in (w1, w2)]
        y = torch.empty((B, 2 * half, H, W), dtype=_half(), device=p1.device, memory_format=torch.channels_last)
        ypres = [torch.empty((M, half), dtype=_half(), device=p1.device) for _ in range(2)]
        g_, b_ = gamma.detach(), beta.detach()
        P = ctypes.c_void_p * 2
        sl = lambda t: P(t[:half].data_ptr(), t[half:].data_ptr())       # noqa: E731
        _ops._lib.check(lib.xpt_pwconv_bn_multi_fwd_sib(
            2, P(*[x.data_ptr() for x in x2s]), P(*[q.data_ptr() for q in shadows]), sl(g_), sl(b_), sl(mean), sl(var), float(eps),
            P(None, None), P(*[q.data_ptr() for q in ypres]), P(y.data_ptr(), y.data_ptr() + 2 * half), None, None, None, None,
            None, None, None, M, cin, half, pitch, 2 * half, _ops._stream()), "xpt_pwconv_bn_multi_fwd_sib")
        ctx.save_for_backward(*x2s, *shadows, *ypres, g_, mean, var)
        ctx.dims = (B, cin, H, W, half, float(eps))
        ctx.dsts = [(w1.flat_grad,) + halves[0], (w2.flat_grad,) + halves[1]]
        return y

    @staticmethod
    def backward(ctx, dy):
        import ctypes
        t = ctx.saved_tensors
        x2s, shadows, ypres, gamma, mean, var = t[0:2], t[2:4], t[4:6], t[6], t[7], t[8]
        B, cin, H, W, half, eps = ctx.dims
        lib = _ops._lib.load()
        sink = _ops.grad_sink
        rows = _ops.as_rows(dy.to(_half()))                        # [M, 2 half], unit channel stride
        M = rows.shape[0]
        pitch_dy = rows.stride(0) if M > 1 else 2 * half
        nsplit = lib.xpt_conv1x1_bwd_weight_splits(M, half, cin)
        wparts = [sink.partials(d[0], "conv1x1", nsplit * half * cin) for d in ctx.dsts]
        bparts = [sink.partials(d[2], "bnfuse", nsplit * 2 * half) for d in ctx.dsts]
        P, LL = ctypes.c_void_p * 2, ctypes.c_longlong * 2
        sl = lambda q: P(q[:half].data_ptr(), q[half:].data_ptr())       # noqa: E731
        pitch_x = x2s[0].stride(0) if M > 1 else cin
        dx_all = torch.empty((2, M, cin), dtype=_half(), device=dy.device)
        _ops._lib.check(lib.xpt_conv1x1_bn_multi_bwd_fused(
            2, P(rows.data_ptr(), rows.data_ptr() + 2 * half), LL(pitch_dy, pitch_dy), P(*[q.data_ptr() for q in ypres]),
            P(*[q.data_ptr() for q in x2s]), P(*[q.data_ptr() for q in shadows]), sl(gamma), sl(var), sl(mean), eps,
            P(dx_all[0].data_ptr(), dx_all[1].data_ptr()), P(*[q.data_ptr() for q in wparts]),
            P(*[q.data_ptr() for q in bparts]), wparts[0].numel(), bparts[0].numel(), M, half, cin, pitch_x, _ops._stream()),
            "xpt_conv1x1_bn_multi_bwd_fused")
        for j, (w_dst, g_dst, b_dst) in enumerate(ctx.dsts):
            sink.add(w_dst, wparts[j], 0, half * cin, nsplit, half * cin)
            sink.add(b_dst, bparts[j], 0, half, nsplit, 2 * half)
            sink.add(g_dst, bparts[j], half, half, nsplit, 2 * half)
        dxs = [dx_all[j].view(B, H, W, cin).permute(0, 3, 1, 2) for j in range(2)]
        return (dxs[0], dxs[1]) + (None,) * 8


def _bn_grad_halves(bn, half):
    """Persistent views of the two halves of a BatchNorm's deferred-gradient destinations (the gradient sink keys its
    workspaces by the destination OBJECT: a fresh slice per step would look like a new parameter every time)."""
    key = (bn.weight.flat_grad.data_ptr(), bn.bias.flat_grad.data_ptr(), half)
    cached = getattr(bn, "_xpt_grad_halves", None)
    if cached is None or cached[0] != key:
        gw, gb = bn.weight.flat_grad, bn.bias.flat_grad
        cached = bn._xpt_grad_halves = (key, ((gw[:half], gb[:half]), (gw[half:], gb[half:])))
    return cached[1]


_FUSED_SPATIAL_ADJUST = __import__("os").environ.get("XPT_DEBUG_UNFUSED_SPATIAL_ADJUST", "0") != "1"     # A/B: two GEMMs + concat + BatchNorm launches


_FUSE_CONV_BN = __import__("os").environ.get("XPT_DEBUG_UNFUSED_CONV_BN", "0") != "1"
_WIDE_CELL = __import__("os").environ.get("XPT_DEBUG_NARROW_CELL", "0") != "1"          # A/B: per-branch depthwise launches
_STEM2_FILTERS = int(__import__("os").environ.get("XPT_STEM2_FILTERS", "24"))     # physical filters of the second cell (22 logical; 22 = no padding)
_STEM1_FILTERS = int(__import__("os").environ.get("XPT_STEM1_FILTERS", "16"))     # physical filters of the first cell (11 logical; 11 = no padding)
_WIDE_STEM = __import__("os").environ.get("XPT_DEBUG_NARROW_STEM", "0") != "1"          # A/B: the first reduction cell branch by branch
_FUSE_FAN_IN = __import__("os").environ.get("XPT_DEBUG_SEPARATE_FAN_IN", "0") != "1"     # A/B: gradient fan-in as its own launch
_FUSED_DGRAD = __import__("os").environ.get("XPT_DEBUG_GEMM_DGRAD", "0") != "1"     # A/B: data gradient of conv1x1+BN as a library GEMM launch
_CELL_TAIL = __import__("os").environ.get("XPT_DEBUG_UNFUSED_CELL_TAIL", "0") != "1"     # A/B: pools / add / concat / relu as separate launches
_SIBLING_PW = __import__("os").environ.get("XPT_DEBUG_SEPARATE_SIBLINGS", "0") != "1"     # A/B: right branches' last pointwise layers as their own launch


def _rectified_concat(spec, inputs):
    """relu(concat(...)) of a cell in one launch (csrc/xpt_celltail.hip).  Every consumer of a cell output starts with
    Activation('relu') (shared_relu), so the cell hands out the rectified tensor; its aliases carry one gradient edge per
    consumer (next cell's squeeze convolution, the adjust block after it, a decoder tap)."""
    outs = _ops.cell_tail(spec, inputs, 3)
    head = outs[0]
    head._xpt_relu_aliases = outs[1:]        # (not the head itself: a tensor -> list -> tensor cycle would outlive the step)
    return head
_PWCONV_MAX_CIN = int(__import__("os").environ.get("XPT_PWCONV_MAX_CIN", "1056"))    # deeper reductions: library GEMM + epilogue launch
_LIBRARY_PWCONV = __import__("os").environ.get("XPT_DEBUG_LIBRARY_PWCONV", "0") == "1"    # A/B: rocBLAS GEMM + epilogue launch


def conv1x1_bn(x, weight, bn, residual=None, fan_out=1):
    """bn(conv1x1(x, weight)) [+ residual]: the fused-backward path when training on flat (deferred-gradient)
    parameters in bf16, the two separate ops otherwise.  fan_out = n > 1 (no residual): a tuple of n aliases of the
    result, one per consumer; on the fused path their gradients are added inside the weight-gradient launch."""
    sink = _ops.grad_sink
    if (_FUSE_CONV_BN and x.is_cuda and x.dtype == _half() and torch.is_autocast_enabled()
            and torch.is_grad_enabled() and hasattr(weight, "shadow_bf16") and not _LIBRARY_WGRAD
            and sink.wants(weight) and sink.wants(bn.weight) and sink.wants(bn.bias)):
        if fan_out > 1 and residual is None and _FUSE_FAN_IN:
            return _Conv1x1BnBf16.apply(x, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, BN_EPS, None, fan_out)
        y = _Conv1x1BnBf16.apply(x, weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, BN_EPS, residual)
    else:
        y = bn(conv1x1(x, weight), residual)
    return y if fan_out == 1 else _ops.fan_out(y, fan_out)


def correct_pad(h, w, k):
    """keras imagenet_utils.correct_pad for a stride-2 VALID conv: ((k//2 - adj_h, k//2), (k//2 - adj_w, k//2))
    with adj = 1 for an even extent -- identical to TF SAME padding at stride 2."""
    return same_pad(h, k, 2), same_pad(w, k, 2)


def zero_pad(x, pad_hw):
    (pt, pb), (pl, pr) = pad_hw
    return F.pad(x, (pl, pr, pt, pb))


class SeparableConv(nn.Module):
    """keras SeparableConv2D(use_bias=False): depthwise k x k (multiplier 1) then pointwise 1x1."""

    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.k, self.stride = k, stride
        self.depthwise = nn.Conv2d(cin, cin, k, stride, padding=0 if stride == 2 else k // 2, groups=cin, bias=False)
        self.pointwise = nn.Conv2d(cin, cout, 1, bias=False)
        nn.init.kaiming_normal_(self.depthwise.weight, mode="fan_in", nonlinearity="relu")
        nn.init.kaiming_normal_(self.pointwise.weight, mode="fan_in", nonlinearity="relu")

    def forward(self, x, relu_in=False, bn=None, residual=None):
        """relu_in: apply the ReLU that precedes the convolution in _separable_conv_block (fused into the gfx950
        depthwise kernel on the GPU).  bn / residual: the BatchNorm (and branch add) that follow the pointwise half,
        applied here so that their backward can ride in the pointwise weight-gradient launch."""
        if x.is_cuda and not _DISABLE_HIP_DWCONV:
            if self.stride == 2:
                (pt, pb), (pl, pr) = correct_pad(x.shape[2], x.shape[3], self.k)
            else:
                pt = pb = pl = pr = self.k // 2
            y = _ops.depthwise_conv2d(x, self.depthwise.weight, self.stride, (pt, pb, pl, pr), relu_in)
            if bn is not None:
                return conv1x1_bn(y, self.pointwise.weight, bn, residual)
            return conv1x1(y, self.pointwise.weight)
        if relu_in:                       # host tensors (CPU baseline / unit tests): library convolution
            x = F.relu(x)
        if self.stride == 2:
            x = zero_pad(x, correct_pad(x.shape[2], x.shape[3], self.k))
        y = self.pointwise(self.depthwise(x))
        return y if bn is None else bn(y, residual)


class SepConvBlock(nn.Module):
    """_separable_conv_block: relu -> sepconv(k, stride) -> BN -> relu -> sepconv(k, 1) -> BN.
    Creates TWO unnamed Activation layers (counted by the tap bookkeeping)."""

    def __init__(self, net, cin, filters, k=3, stride=1):
        super().__init__()
        self.act_id1 = net.new_activation()
        self.conv1 = SeparableConv(cin, filters, k, stride)
        self.bn1 = FrozenBatchNorm(filters)
        self.act_id2 = net.new_activation()
        self.conv2 = SeparableConv(filters, filters, k, 1)
        self.bn2 = FrozenBatchNorm(filters)

    def forward(self, x, taps, residual=None, rectified=False):
        """residual: the other operand of the cell's `add` (fused into the last BatchNorm's epilogue).
        rectified: x has been through the ReLU already (the very first cell shares one ReLU among all its branches)."""
        if taps.wants(self.act_id1) or taps.wants(self.act_id2):      # a tapped activation must be materialised
            if not rectified:
                x = F.relu(x)
            taps.offer(self.act_id1, x)
            x = self.conv1(x, bn=self.bn1)
            if not taps.wants(self.act_id2):      # (NASNet-Mobile's skip taps are all first activations of a block)
                return self.conv2(x, relu_in=True, bn=self.bn2, residual=residual)
            x = F.relu(x)
            taps.offer(self.act_id2, x)
            return self.conv2(x, bn=self.bn2, residual=residual)
        x = self.conv1(x, relu_in=True, bn=self.bn1)
        return self.conv2(x, relu_in=True, bn=self.bn2, residual=residual)


class AdjustBlock(nn.Module):
    """_adjust_block: brings `p` (the cell input of two cells ago) to the spatial size / channel count of the
    current cell.  The spatial branch uses a NAMED relu (not counted); the projection branch an unnamed one."""

    def __init__(self, net, p_channels, p_reduction, ip_reduction, filters, p_is_none=False):
        super().__init__()
        self.mode = "none"
        if p_is_none:                                   # first cell: `p = ip`, nothing else (if / elif chain)
            pass
        elif p_reduction != ip_reduction:
            self.mode = "spatial"
            self.conv1 = nn.Conv2d(p_channels, filters // 2, 1, bias=False)
            self.conv2 = nn.Conv2d(p_channels, filters // 2, 1, bias=False)
            self.bn = FrozenBatchNorm(2 * (filters // 2))
        elif p_channels != filters:
            self.mode = "project"
            self.act_id = net.new_activation()
            self.conv = nn.Conv2d(p_channels, filters, 1, bias=False)
            self.bn = FrozenBatchNorm(filters)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")

    def forward(self, p, taps, fan_out=1):
        """fan_out = n > 1: a tuple of n aliases of the result, one per consumer (conv1x1_bn adds their gradients inside
        its weight-gradient launch; the other modes go through the fan-in kernel)."""
        if self.mode == "project":
            p = shared_relu(p)
            taps.offer(self.act_id, p)
            return conv1x1_bn(p, self.conv.weight, self.bn, fan_out=fan_out)
        if self.mode == "spatial":
            p = shared_relu(p)
            if p.is_cuda and _CELL_TAIL and p.dtype in (torch.float32, _half()):
                p1, p2 = _ops.adjust_gather(p)                  # both sub-sampled copies in one launch (one scatter backward)
                w1, w2, bn = self.conv1.weight, self.conv2.weight, self.bn
                half, cin = w1.shape[0], w1.shape[1]
                sink = _ops.grad_sink
                if (_FUSED_SPATIAL_ADJUST and _FUSE_CONV_BN and _FUSED_DGRAD and not _LIBRARY_PWCONV and not _LIBRARY_WGRAD
                        and p.dtype == _half() and torch.is_autocast_enabled() and torch.is_grad_enabled()
                        and half % 2 == 0 and cin % 2 == 0 and cin <= _PWCONV_MAX_CIN
                        and all(hasattr(w, "shadow_bf16") and sink.wants(w) and w.shadow_bf16.is_contiguous() for w in (w1, w2))
                        and sink.wants(bn.weight) and sink.wants(bn.bias)):
                    # both convolutions and the BatchNorm over their concatenation: one launch (and one backward launch)
                    p = _SpatialAdjustBn.apply(p1, p2, w1, w2, bn.weight, bn.bias, bn.running_mean, bn.running_var, BN_EPS,
                                               _bn_grad_halves(bn, half))
                else:
                    p = self.bn(torch.cat([conv1x1(p1, w1), conv1x1(p2, w2)], dim=1))
            else:
                p1 = conv1x1(p[:, :, ::2, ::2], self.conv1.weight)                   # AveragePooling2D((1,1), strides 2)
                p2 = F.pad(p, (0, 1, 0, 1))[:, :, 1:, 1:]            # ZeroPadding2D(((0,1),(0,1))) + Cropping2D(((1,0),(1,0)))
                p2 = conv1x1(p2[:, :, ::2, ::2], self.conv2.weight)
                p = self.bn(torch.cat([p1, p2], dim=1))
        return p if fan_out == 1 else _ops.fan_out(p, fan_out)


def avg_pool_same(x, scale=1.0):
    """scale * AveragePooling2D((3,3), strides 1, padding='same')(x): the divisor excludes the padding."""
    if x.is_cuda:
        return _ops.avg_pool3_same(x, scale)
    y = F.avg_pool2d(x, 3, 1, 1, count_include_pad=False)
    return y if scale == 1.0 else y * scale


# The one-launch branch stage (csrc/xpt_sepconv.hip) is bit-identical to the two / three launches it replaces and removes 36
# launches from the step, but it is SLOWER at batch 8 (measured, profiles/r03_c_*: 6.67 against 6.56 ms per step): with 32
# pixels per workgroup a stage is 39-1040 workgroups whose critical path -- taps to LDS, the depthwise loads, the filter
# rows, the stores: four dependent memory round trips and four barriers -- is as long as the separate launches' paths
# together, and on the 4 x 13 maps only 39-65 of the 256 CUs have work.  Off by default; XPT_FUSED_SEPCONV=1 turns it on.
_FUSED_SEP_STAGE = __import__("os").environ.get("XPT_FUSED_SEPCONV", "0") == "1"


def _sep_stage_usable(xs, seps, bns):
    """The one-launch branch stage (csrc/xpt_sepconv.hip) applies: training in bf16 on flat (deferred-gradient) parameters,
    stride-1 separable convolutions of one activation shape with equally many input and output channels."""
    sink = _ops.grad_sink
    x0 = xs[0]
    return (_FUSED_SEP_STAGE and _FUSE_CONV_BN and _WIDE_CELL and not _LIBRARY_PWCONV and not _LIBRARY_WGRAD
            and not _DISABLE_HIP_DWCONV and x0.is_cuda and x0.dtype == _half() and torch.is_autocast_enabled()
            and torch.is_grad_enabled() and x0.shape[1] % 2 == 0
            and all(x.shape == x0.shape and x.dtype == x0.dtype and x.is_contiguous(memory_format=torch.channels_last) for x in xs)
            and all(sp.stride == 1 and sp.k in (3, 5, 7) and sp.pointwise.weight.shape[:2] == (x0.shape[1], x0.shape[1])
                    and hasattr(sp.pointwise.weight, "shadow_bf16") and sink.wants(sp.pointwise.weight)
                    and sp.depthwise.weight.dtype == torch.float32 for sp in seps)
            and all(sink.wants(b.weight) and sink.wants(b.bias) for b in bns))


def fused_sep_stage(main, siblings=None, residuals=None):
    """One branch stage of a cell in ONE forward launch.  main / siblings: lists of (x, SeparableConv, FrozenBatchNorm);
    job j computes  BN_j(pw_j(dw_j(relu(x_j))))  [+ BN(pw(dw(relu(x)))) of siblings[j]]  [+ residuals[j]].
    Returns (results per job, the siblings' own outputs or None).  The autograd graph is that of the unfused path
    (_MultiDepthwise followed by _MultiConv1x1Bn, the sibling branches first): the launch only pre-computes their outputs,
    their backward kernels run unchanged."""
    import ctypes
    n = len(main)
    siblings = siblings or [None] * n
    residuals = residuals or [None] * n
    lib = _ops._lib.load()
    x0 = main[0][0]
    B, C, H, W = x0.shape
    M = B * H * W
    dev = x0.device
    new_map = lambda: torch.empty((B, C, H, W), dtype=_half(), device=dev, memory_format=torch.channels_last)   # noqa: E731
    new_rows = lambda: torch.empty((M, C), dtype=_half(), device=dev)                                           # noqa: E731
    sib = [j for j in range(n) if siblings[j] is not None]
    ydw_a, ypre_a, y_a = [new_map() for _ in range(n)], [new_rows() for _ in range(n)], [new_map() for _ in range(n)]
    ydw_b, ypre_b, y_b = {j: new_map() for j in sib}, {j: new_rows() for j in sib}, {j: new_map() for j in sib}
    res = [None if r is None else r.to(_half()).contiguous(memory_format=torch.channels_last) for r in residuals]
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    ptr = lambda ts: P(*[None if t is None else t.data_ptr() for t in ts])                                             # noqa: E731

    def columns(entries):
        xs = [None if e is None else e[0] for e in entries]
        sp = [None if e is None else e[1] for e in entries]
        bn = [None if e is None else e[2] for e in entries]
        pick = lambda f: [None if e is None else f(e) for e in entries]                                                 # noqa: E731
        return (ptr(xs), ptr(pick(lambda e: e[1].depthwise.weight.detach())),
                ptr(pick(lambda e: e[1].pointwise.weight.shadow_bf16)), ptr(pick(lambda e: e[2].weight.detach())),
                ptr(pick(lambda e: e[2].bias.detach())), ptr(pick(lambda e: e[2].running_mean)),
                ptr(pick(lambda e: e[2].running_var)), I(*[0 if e is None else int(e[1].k) for e in entries]))

    a, b = columns(main), columns(siblings)
    _ops._lib.check(lib.xpt_sepconv_bn_multi_fwd(
        n, *a, ptr(ydw_a), ptr(ypre_a), *b, ptr([ydw_b.get(j) for j in range(n)]), ptr([ypre_b.get(j) for j in range(n)]),
        ptr([y_b.get(j) for j in range(n)]), ptr(res), ptr(y_a), float(BN_EPS), B, H, W, C, C, _ops._stream()),
        "xpt_sepconv_bn_multi_fwd")
    # ---- the unfused path's autograd graph around the pre-computed outputs
    entries = list(main) + [siblings[j] for j in sib]
    _ops._MultiDepthwise.precomputed = ydw_a + [ydw_b[j] for j in sib]
    ys_dw = _ops.multi_depthwise([e[0] for e in entries], [e[1].depthwise.weight for e in entries])
    sib_out = {}
    if sib:
        _MultiConv1x1Bn.precomputed = ([ypre_b[j] for j in sib], [y_b[j] for j in sib])
        outs = _MultiConv1x1Bn.apply(len(sib), BN_EPS, (), *ys_dw[n:], *[siblings[j][1].pointwise.weight for j in sib],
                                     *[siblings[j][2].weight for j in sib], *[siblings[j][2].bias for j in sib],
                                     *[siblings[j][2].running_mean for j in sib], *[siblings[j][2].running_var for j in sib],
                                     *([None] * len(sib)))
        sib_out = dict(zip(sib, outs))
    adds = [sib_out.get(j, residuals[j]) for j in range(n)]
    _MultiConv1x1Bn.precomputed = (ypre_a, y_a)
    outs = _MultiConv1x1Bn.apply(n, BN_EPS, (), *ys_dw[:n], *[e[1].pointwise.weight for e in main], *[e[2].weight for e in main],
                                 *[e[2].bias for e in main], *[e[2].running_mean for e in main],
                                 *[e[2].running_var for e in main], *adds)
    return list(outs), [sib_out.get(j) for j in range(n)]


class NormalCell(nn.Module):
    """_normal_a_cell -> concat([p, x1, x2, x3, x4, x5]) = 6 * filters channels."""

    def __init__(self, net, ip_channels, p_channels, ip_reduction, p_reduction, filters, p_is_none=False):
        super().__init__()
        self.adjust = AdjustBlock(net, p_channels, p_reduction, ip_reduction, filters, p_is_none)
        self.act_id = net.new_activation()
        self.conv = nn.Conv2d(ip_channels, filters, 1, bias=False)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_in", nonlinearity="relu")
        self.bn = FrozenBatchNorm(filters)
        p_ch = filters if self.adjust.mode != "none" else p_channels
        self.left1 = SepConvBlock(net, filters, filters, 5)
        self.right1 = SepConvBlock(net, p_ch, filters, 3)
        self.left2 = SepConvBlock(net, p_ch, filters, 5)
        self.right2 = SepConvBlock(net, p_ch, filters, 3)
        self.left5 = SepConvBlock(net, filters, filters, 3)
        self.out_channels = p_ch + 5 * filters
        self.f_sel = self.ip_sel = self.p_sel = None

    def stack_groups(self):
        """Pointwise weights the wide cell consumes as one strided batch (order = the multi_conv1x1_bn calls below)."""
        blocks = (self.left1, self.left5, self.right1, self.left2, self.right2)
        return [[b.conv1.pointwise.weight for b in blocks],
                [self.left1.conv2.pointwise.weight, self.left2.conv2.pointwise.weight, self.left5.conv2.pointwise.weight,
                 self.right1.conv2.pointwise.weight, self.right2.conv2.pointwise.weight]]

    def forward(self, ip, p, taps):
        blocks = (self.left1, self.left5, self.right1, self.left2, self.right2)
        wide = ip.is_cuda and not _DISABLE_HIP_DWCONV and _WIDE_CELL and not any(
            taps.wants(b.act_id1) or taps.wants(b.act_id2) for b in blocks)
        # h and p feed several branches each: one alias per consumer; where they come out of a fused pointwise + BatchNorm
        # layer their gradients are added inside that layer's weight-gradient launch, elsewhere by one fan-in kernel
        n_h, n_p = (3, 2 if _CELL_TAIL else 4) if wide else (4, 6)
        h = shared_relu(ip)
        taps.offer(self.act_id, h)
        ps = None
        if wide and self.adjust.mode == "project":
            # (usability is judged on p itself -- its rectified aliases share its shape, dtype and strides --: shared_relu
            #  hands out one alias of the producing cell's output per CALL, so it must only be called by a real consumer)
            if pair_conv1x1_bn_usable(h, p, self.conv.weight, self.adjust.conv.weight, self.bn, self.adjust.bn):
                # the squeeze of the cell input and the projection of p read tensors of one shape: one launch for both heads
                # (and one backward launch, which also adds the gradients their consumers send back)
                pr = shared_relu(p)
                taps.offer(self.adjust.act_id, pr)
                a, b = self.bn, self.adjust.bn
                outs = _PairConv1x1BnFan.apply(n_h, n_p, BN_EPS, h, pr, self.conv.weight, self.adjust.conv.weight, a.weight,
                                               b.weight, a.bias, b.bias, a.running_mean, b.running_mean, a.running_var,
                                               b.running_var)
                hs, ps = outs[:n_h], outs[n_h:]
        if ps is None:
            ps = self.adjust(p, taps, fan_out=n_p)
            hs = conv1x1_bn(h, self.conv.weight, self.bn, fan_out=n_h)
        if wide:
            # The five separable-conv branches are mutually independent and of one shape: their depthwise halves run as
            # ONE launch per stage (and one backward launch each, which also sums the gradients of h and p over the
            # branches); the pointwise + BatchNorm halves follow per branch, the branch adds in their epilogues.
            fused_tail = _CELL_TAIL and hs[0].dtype == ps[0].dtype and hs[0].shape == ps[0].shape
            if not fused_tail and len(ps) < 4:
                ps = (ps[0],) + tuple(_ops.fan_out(ps[1], 3))
            stage1_in = [hs[0], hs[0], ps[0], ps[0], ps[0]]
            if _sep_stage_usable(stage1_in, [b.conv1 for b in blocks] + [b.conv2 for b in blocks],
                                 [b.bn1 for b in blocks] + [b.bn2 for b in blocks]):
                # each stage of the five branches is ONE launch (csrc/xpt_sepconv.hip): depthwise tile -> LDS -> matrix
                # cores -> BatchNorm; the second stage adds the sibling branches (x1 = left1 + right1, x2 = left2 + right2)
                # and h (x5 = left5 + h) in its epilogue
                z1, _ = fused_sep_stage([(x, b.conv1, b.bn1) for x, b in zip(stage1_in, blocks)])
                (x1, x2, x5), _ = fused_sep_stage(
                    [(z1[0], self.left1.conv2, self.left1.bn2), (z1[3], self.left2.conv2, self.left2.bn2),
                     (z1[1], self.left5.conv2, self.left5.bn2)],
                    siblings=[(z1[2], self.right1.conv2, self.right1.bn2), (z1[4], self.right2.conv2, self.right2.bn2), None],
                    residuals=[None, None, hs[2]])
            else:
                y1 = _ops.multi_depthwise(stage1_in, [b.conv1.depthwise.weight for b in blocks])
                z1 = multi_conv1x1_bn(y1, [b.conv1.pointwise.weight for b in blocks], [b.bn1 for b in blocks])
                y2 = _ops.multi_depthwise(z1, [b.conv2.depthwise.weight for b in blocks])
                # second pointwise stage in ONE launch: the right branches are the siblings of the left ones they are added
                # to (x1 = left1 + right1, x2 = left2 + right2), x5 = left5 + h takes h as a residual
                lefts = (self.left1, self.left2, self.left5)
                rights = [(y2[2], self.right1.conv2.pointwise.weight, self.right1.bn2),
                          (y2[4], self.right2.conv2.pointwise.weight, self.right2.bn2), None]
                if not _SIBLING_PW:                      # A/B: the right branches first, as a launch of their own
                    r1, r2 = multi_conv1x1_bn([q[0] for q in rights[:2]], [q[1] for q in rights[:2]],
                                              [q[2] for q in rights[:2]])
                x1, x2, x5 = multi_conv1x1_bn([y2[0], y2[3], y2[1]], [b.conv2.pointwise.weight for b in lefts],
                                              [b.bn2 for b in lefts], [None, None, hs[2]] if _SIBLING_PW else [r1, r2, hs[2]],
                                              siblings=rights if _SIBLING_PW else None)
            if fused_tail:
                # concat([p, x1, x2, avg(h) + p, avg(p) + avg(p), x5]) and the consumers' relu: inputs (p, x1, x2, h, x5)
                spec = (((0, 0, 1.0),), ((1, 0, 1.0),), ((2, 0, 1.0),), ((3, 1, 1.0), (0, 0, 1.0)), ((0, 1, 2.0),),
                        ((4, 0, 1.0),))
                return _rectified_concat(spec, [ps[1], x1, x2, hs[1], x5]), ip
            x3 = avg_pool_same(hs[1]) + ps[1]
            x4 = avg_pool_same(ps[2], 2.0)                   # add([avg(p), avg(p)]): x + x == 2 x exactly
            return torch.cat([ps[3], x1, x2, x3, x4, x5], dim=1), ip
        # every `add` whose operand ends in a BatchNorm rides in that BatchNorm's epilogue kernel; h and p feed several
        # branches each: their gradients are summed by one fan-in kernel instead of a chain of pairwise adds
        h1, h2, h3, h4 = hs
        p1, p2, p3, p4, p5, p6 = ps
        x1 = self.left1(h1, taps, residual=self.right1(p1, taps))
        x2 = self.left2(p2, taps, residual=self.right2(p3, taps))
        x3 = avg_pool_same(h2) + p4
        x4 = avg_pool_same(p5, 2.0)                      # add([avg(p), avg(p)]): x + x == 2 x exactly
        x5 = self.left5(h3, taps, residual=h4)
        p = p6
        return torch.cat([p, x1, x2, x3, x4, x5], dim=1), ip


_SEL_ON_DEVICE = {}


def _logical_channels(x, sel):
    """The logical channels of a tensor that carries structurally-zero ones (sel None: x itself)."""
    if sel is None:
        return x
    rows = _ops.as_rows(x) if x.is_cuda else x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])
    key = (id(sel), x.device)
    if key not in _SEL_ON_DEVICE:                # (uploaded by the eager warm-up steps, never inside a capture)
        _SEL_ON_DEVICE[key] = (sel, sel.to(x.device))
    picked = rows[:, _SEL_ON_DEVICE[key][1]]
    return picked.view(x.shape[0], x.shape[2], x.shape[3], len(sel)).permute(0, 3, 1, 2)


class ReductionCell(nn.Module):
    """_reduction_a_cell -> concat([x2, x3, x4, x5]) = 4 * filters channels at half the resolution."""

    def __init__(self, net, ip_channels, p_channels, ip_reduction, p_reduction, filters, p_is_none=False):
        super().__init__()
        self.adjust = AdjustBlock(net, p_channels, p_reduction, ip_reduction, filters, p_is_none)
        self.act_id = net.new_activation()
        self.conv = nn.Conv2d(ip_channels, filters, 1, bias=False)
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_in", nonlinearity="relu")
        self.bn = FrozenBatchNorm(filters)
        p_ch = filters if self.adjust.mode != "none" else p_channels
        self.left1 = SepConvBlock(net, filters, filters, 5, 2)
        self.right1 = SepConvBlock(net, p_ch, filters, 7, 2)
        self.right2 = SepConvBlock(net, p_ch, filters, 7, 2)
        self.right3 = SepConvBlock(net, p_ch, filters, 5, 2)
        self.left4 = SepConvBlock(net, filters, filters, 3, 1)
        self.out_channels = 4 * filters
        self.f_sel = self.ip_sel = self.p_sel = None       # logical channel indices (NASNetMobileEncoder: structural zeros)

    def stack_groups(self):
        return [[b.conv1.pointwise.weight for b in (self.left1, self.right1, self.right2)],
                [self.right1.conv2.pointwise.weight, self.right2.conv2.pointwise.weight]]

    def forward(self, ip, p, taps):
        # h feeds the pooling pair and the left branch, p the stride-2 branches and the tapped right3 block: two aliases
        # each (their gradients are added inside the producing layer's weight-gradient launch where that layer is fused)
        rectified = self.adjust.mode == "none" and p is ip and ip.is_cuda
        tap_sel = self.f_sel if self.adjust.mode != "none" else self.p_sel      # channel layout of what right3 reads (its tap)
        if rectified:
            # the very first cell: p IS ip and every branch starts with a ReLU, so one ReLU serves them all (the
            # in-kernel ReLU of the separable convolutions is idempotent on it) and the raw stem output has ONE consumer
            h, p, p_tap = _ops.fan_out(shared_relu(ip), 3)
        else:
            p, p_tap = self.adjust(p, taps, fan_out=2)
            h = shared_relu(ip)
        taps.offer(self.act_id, h)
        h, h_pool = conv1x1_bn(h, self.conv.weight, self.bn, fan_out=2)
        if h.is_cuda and _CELL_TAIL and h.dtype in (torch.float32, _half()):
            # both poolings of the zero-padded h in one launch (and one backward launch)
            # (the max-pooled tensor feeds x2 and x5: two aliases, their gradients added inside the pooling backward)
            mp1, mp2, ap_h = _ops.pool_pair(h_pool, correct_pad(h.shape[2], h.shape[3], 3), split_mp=True)
        else:
            h3 = zero_pad(h_pool, correct_pad(h.shape[2], h.shape[3], 3))
            mp = F.max_pool2d(h3, 3, 2)                      # MaxPooling2D of h feeds x2 and x5: pooled once
            ap_h = None
            mp1, mp2 = _ops.fan_out(mp, 2)
        def tapped(b):
            return taps.wants(b.act_id1) or taps.wants(b.act_id2)

        # (the very first cell: h has `filters` channels, p the stem convolution's 32 -- its first stage runs the left branch
        #  and the right branches as two groups, everything behind it is of one shape again)
        mixed = _WIDE_STEM and h.shape[1] != p.shape[1] and h.shape[0] == p.shape[0] and h.shape[2:] == p.shape[2:]
        if (h.is_cuda and not _DISABLE_HIP_DWCONV and _WIDE_CELL and (h.shape == p.shape or mixed)
                and not any(tapped(b) for b in (self.left1, self.right1, self.right2, self.left4))):
            # the stride-2 separable-conv branches stage by stage, as in the normal cell (left4 follows on x1); right3
            # carries one of the decoder's skip taps in every reduction cell and then runs on its own
            wide3 = not tapped(self.right3)
            if not wide3 and _WIDE_STEM and not taps.wants(self.right3.act_id2):
                # the tapped activation is relu(p) -- in the first cell the (already rectified) input itself --: hand it out and
                # let the block join the stage launches on p (the in-kernel ReLU of its depthwise layer gives the same numbers)
                taps.offer(self.right3.act_id1, p_tap if rectified else F.relu(_logical_channels(p_tap, None if taps.physical else tap_sel)))
                wide3 = True
            blocks = (self.left1, self.right1, self.right2) + ((self.right3,) if wide3 else ())
            H, W = h.shape[2], h.shape[3]
            pads = []
            for b in blocks:
                (pt, pb), (pl, pr) = correct_pad(H, W, b.conv1.k)
                pads.append((pt, pb, pl, pr))
            if mixed:
                rights = blocks[1:]
                zl = self.left1.conv1(h, relu_in=True, bn=self.left1.bn1)
                yr = _ops.multi_depthwise([p] * len(rights), [b.conv1.depthwise.weight for b in rights], stride=2, pads=pads[1:])
                z1 = [zl] + list(multi_conv1x1_bn(yr, [b.conv1.pointwise.weight for b in rights], [b.bn1 for b in rights]))
            else:
                y1 = _ops.multi_depthwise([h] + [p] * (len(blocks) - 1), [b.conv1.depthwise.weight for b in blocks],
                                          stride=2, pads=pads)
                z1 = multi_conv1x1_bn(y1, [b.conv1.pointwise.weight for b in blocks], [b.bn1 for b in blocks])
            y2 = _ops.multi_depthwise(z1, [b.conv2.depthwise.weight for b in blocks])
            ap = ap_h if ap_h is not None else F.avg_pool2d(h3, 3, 2)
            if _SIBLING_PW:
                # x1 = left1 + right1 (right1 as the sibling of left1), x2 = right2 + max-pool, x3 = right3 + avg-pool: ONE launch
                mains = [blocks[0], blocks[2]] + ([blocks[3]] if wide3 else [])
                outs = multi_conv1x1_bn([y2[0], y2[2]] + ([y2[3]] if wide3 else []), [b.conv2.pointwise.weight for b in mains],
                                        [b.bn2 for b in mains], [None, mp1] + ([ap] if wide3 else []),
                                        siblings=[(y2[1], self.right1.conv2.pointwise.weight, self.right1.bn2), None]
                                        + ([None] if wide3 else []))
                x1, x2 = outs[0], outs[1]
                x3 = outs[2] if wide3 else self.right3(p_tap, taps, residual=ap, rectified=rectified)
            else:
                outs = multi_conv1x1_bn(y2[1:], [b.conv2.pointwise.weight for b in blocks[1:]], [b.bn2 for b in blocks[1:]],
                                        [None, mp1] + ([ap] if wide3 else []))
                r1, x2 = outs[0], outs[1]
                x3 = outs[2] if wide3 else self.right3(p_tap, taps, residual=ap, rectified=rectified)
                x1 = conv1x1_bn(y2[0], self.left1.conv2.pointwise.weight, self.left1.bn2, residual=r1)
            x1a, x1b = _ops.fan_out(x1, 2)
            if _CELL_TAIL and x1.dtype == x2.dtype == x3.dtype:
                # concat([x2, x3, x2 + avg(x1), x5]) and the consumers' relu: inputs (x2, x3, x1, x5)
                x5 = self.left4(x1b, taps, residual=mp2)
                spec = (((0, 0, 1.0),), ((1, 0, 1.0),), ((0, 0, 1.0), (2, 1, 1.0)), ((3, 0, 1.0),))
                return _rectified_concat(spec, [x2, x3, x1a, x5]), ip
            x2a, x2b = _ops.fan_out(x2, 2)
            x4 = x2a + avg_pool_same(x1a)
            x5 = self.left4(x1b, taps, residual=mp2)
            return torch.cat([x2b, x3, x4, x5], dim=1), ip
        p1, p2 = _ops.fan_out(p, 2)
        p3 = p_tap
        x1 = self.left1(h, taps, residual=self.right1(p1, taps))
        x1a, x1b = _ops.fan_out(x1, 2)
        x2 = self.right2(p2, taps, residual=mp1)
        x2a, x2b = _ops.fan_out(x2, 2)
        x3 = self.right3(p3, taps, residual=ap_h if ap_h is not None else F.avg_pool2d(h3, 3, 2), rectified=rectified)
        if tap_sel is not None and not taps.physical and self.right3.act_id1 in taps.found:
            taps.found[self.right3.act_id1] = _logical_channels(taps.found[self.right3.act_id1], tap_sel)
        x4 = x2a + avg_pool_same(x1a)
        x5 = self.left4(x1b, taps, residual=mp2)
        x2 = x2b
        return torch.cat([x2, x3, x4, x5], dim=1), ip


class _Taps:
    def __init__(self, wanted, physical=False):
        self.wanted = wanted
        self.found = {}
        self.physical = physical       # hand out tapped tensors with their structurally-zero channels (no gather)

    def wants(self, act_id):
        return act_id in self.wanted

    def offer(self, act_id, tensor):
        if act_id in self.wanted:
            self.found[act_id] = tensor


def cell_phys(encoder, act_id):
    """Physical channel count of the tensor tapped at act_id (a reduction cell's right3 input)."""
    for cell in encoder.cells:
        if isinstance(cell, ReductionCell) and cell.right3.act_id1 == act_id:
            return cell.right3.conv1.depthwise.weight.shape[0]
    raise WrongInputException(f"no tapped block at activation {act_id}")


class NASNetMobileEncoder(nn.Module):
    """NASNet-A Mobile (penultimate_filters=1056, num_blocks=4, stem_block_filters=32, filter_multiplier=2,
    skip_reduction=False), include_top=False, with the five taps of scaled_layers.json.

    forward(image NCHW in [-1,1]) -> [c1 (1/2, 32 ch), c2 (1/4, 22), c3 (1/8, 88), c4 (1/16, 176), c5 (1/32, 1056)]."""
    TAP_ACTIVATIONS = (7, 18, 77, 136, 187)          # scaled_layers.json "NASNetMobile": activation_<k>
    TAP_CHANNELS = (32, 22, 88, 176, 1056)

    def __init__(self, penultimate_filters=1056, num_blocks=4, stem_block_filters=32, filter_multiplier=2):
        super().__init__()
        self._n_act = 0
        filters = penultimate_filters // 24
        fm = filter_multiplier
        self.stem_conv = nn.Conv2d(3, stem_block_filters, 3, 2, 0, bias=False)       # padding="valid"
        nn.init.kaiming_normal_(self.stem_conv.weight, mode="fan_in", nonlinearity="relu")
        self.stem_bn = FrozenBatchNorm(stem_block_filters)
        cells = []
        # (channels, log2 reduction) of x (current) and p (previous cell input)
        x_ch, x_red = stem_block_filters, 1
        p_ch, p_red = None, None

        # Physical channel indices of the logical channels of x / p (None: all of them).  The first cell has 11 filters: every
        # tensor in it is 22 bytes per pixel, which no vector load, MFMA fragment or fused data gradient of the kernels takes
        # (scalar paths, 13 library GEMMs, a third of the backward's tail).  It is built with 16 filters of which 5 are
        # STRUCTURALLY ZERO: their weights, BatchNorm parameters and the columns its two consumers read them through are zero,
        # so their activations and every gradient that reaches those entries are exactly zero, Adam leaves them at zero, and
        # the other 11 channels compute what the 11-filter cell computes (structural_pads() lists the tensors; the Keras
        # variable map addresses the logical entries only).
        x_sel = p_sel = None

        def add(kind, f, pad_to=None):
            nonlocal x_ch, x_red, p_ch, p_red, x_sel, p_sel
            pc, pr = (x_ch, x_red) if p_ch is None else (p_ch, p_red)
            cls = ReductionCell if kind == "R" else NormalCell
            phys = pad_to if (pad_to is not None and pad_to > f) else f
            cell = cls(self, x_ch, pc, x_red, pr, phys, p_is_none=p_ch is None)
            cell.ip_sel, cell.p_sel = x_sel, (x_sel if p_ch is None else p_sel)
            cell.f_sel = None
            if phys != f:
                if kind != "R":
                    raise WrongInputException("structurally padded filters: reduction cells only")
                if cell.adjust.mode == "spatial":          # two halves of phys / 2 channels, f / 2 logical ones each
                    cell.f_sel = torch.cat([torch.arange(f // 2), phys // 2 + torch.arange(f // 2)])
                else:
                    cell.f_sel = torch.arange(f)
            cells.append(cell)
            p_ch, p_red, p_sel = x_ch, x_red, x_sel        # the cell returns (x, ip): p <- ip
            x_ch = cell.out_channels
            x_sel = torch.cat([g * phys + cell.f_sel for g in range(4)]) if phys != f else None
            if kind == "R":
                x_red += 1

        add("R", filters // (fm ** 2), pad_to=_STEM1_FILTERS)    # stem_1
        add("R", filters // fm, pad_to=_STEM2_FILTERS)     # stem_2
        for _ in range(num_blocks):
            add("N", filters)
        add("R", filters * fm)                             # reduce_4  (skip_reduction=False: p <- p0 = its input)
        for _ in range(num_blocks):
            add("N", filters * fm)
        add("R", filters * fm ** 2)                        # reduce_8
        for _ in range(num_blocks):
            add("N", filters * fm ** 2)
        self.cells = nn.ModuleList(cells)
        self.final_act_id = self.new_activation()
        self.out_channels = x_ch
        self.num_activations = self._n_act
        self.apply_structural_zeros(rescale=True)

    def structural_pads(self):
        """[(tensor, out_sel, in_sel)]: the tensors that carry structurally-zero entries and the indices of their LOGICAL
        entries along the output (first) and input (second) axis (None: the whole axis)."""
        out = []

        def bn(m, sel):
            out.extend((t, sel, None) for t in (m.weight, m.bias, m.running_mean, m.running_var))

        def sep(block, in_sel, f):
            out.append((block.conv1.depthwise.weight, in_sel, None))
            out.append((block.conv1.pointwise.weight, f, in_sel))
            bn(block.bn1, f)
            out.append((block.conv2.depthwise.weight, f, None))
            out.append((block.conv2.pointwise.weight, f, f))
            bn(block.bn2, f)

        for cell in self.cells:
            f, ip, ps = cell.f_sel, cell.ip_sel, cell.p_sel
            if f is None and ip is None and ps is None:
                continue
            adj = cell.adjust
            if adj.mode == "spatial":
                half = None if f is None else f[:len(f) // 2]          # the logical rows of each half
                out.extend([(adj.conv1.weight, half, ps), (adj.conv2.weight, half, ps)])
                if f is not None:
                    bn(adj.bn, f)
            elif adj.mode == "project":
                out.append((adj.conv.weight, f, ps))
                if f is not None:
                    bn(adj.bn, f)
            out.append((cell.conv.weight, f, ip))
            if f is not None:
                bn(cell.bn, f)
                p_in = ps if adj.mode == "none" else f         # what the right branches read: raw p or the adjusted one
                for name, in_sel in (("left1", f), ("right1", p_in), ("right2", p_in), ("right3", p_in), ("left4", f)):
                    sep(getattr(cell, name), in_sel, f)
        return [(t, o, i) for t, o, i in out if o is not None or i is not None]

    def apply_structural_zeros(self, rescale=False):
        """Zeroes the structural entries (running variances: 1).  rescale: once, right after the random initialisation --
        the He scale of a weight follows its LOGICAL fan-in."""
        with torch.no_grad():
            for t, o, i in self.structural_pads():
                keep = torch.zeros(t.shape[:2] if (t.dim() > 1 and i is not None) else t.shape[:1], dtype=torch.bool)
                oo = o if o is not None else torch.arange(t.shape[0])
                if keep.dim() == 2:
                    keep[oo[:, None], i[None, :]] = True
                else:
                    keep[oo] = True
                keep = keep.to(t.device).view(*keep.shape, *([1] * (t.dim() - keep.dim())))
                is_var = any(t is m.running_var for m in self.modules() if isinstance(m, FrozenBatchNorm))
                t.copy_(torch.where(keep, t, torch.ones_like(t) if is_var else torch.zeros_like(t)))
                if rescale and t.dim() == 4 and i is not None:
                    t.mul_((t.shape[1] / len(i)) ** 0.5)

    def new_activation(self):
        """Index this unnamed keras Activation layer would get ('activation', 'activation_1', ...)."""
        k = self._n_act
        self._n_act += 1
        return k

    def preprocess(self, image):
        """pretrained_nets.py:36-43."""
        x = image / 127.5 - 1.0
        h, w = image.shape[2:]
        return F.interpolate(x, size=(h + 2, w + 2), mode="bilinear", align_corners=False, antialias=False)

    def tap_layout(self):
        """[(physical channels, logical index set or None)] of the five taps as forward(image, physical_taps=True) returns them."""
        out = []
        for k, ch in zip(self.TAP_ACTIVATIONS, self.TAP_CHANNELS):
            sel = None
            for cell in self.cells:
                if isinstance(cell, ReductionCell) and cell.right3.act_id1 == k:
                    sel = cell.f_sel if cell.adjust.mode != "none" else cell.p_sel
            out.append((ch if sel is None else (cell_phys(self, k)), sel))
        return out

    def forward(self, image, physical_taps=False):
        """physical_taps: a tap that lies inside a cell with structurally-zero filters comes with them (tap_layout());
        otherwise its logical channels are gathered (TAP_CHANNELS)."""
        taps = _Taps(self.TAP_ACTIVATIONS, physical_taps)
        if _conv.stem_input_usable(image) and self.stem_conv.weight.dtype == torch.float32:
            # preprocessing, resize, cast and channel padding in one launch, then the matrix-core stem convolution
            x = _conv.conv2d_same(_conv.stem_input(image), self.stem_conv.weight, None, 2, 1.0, valid=True)
            first = self.cells[0]
            if _FUSED_STEM_RELU and isinstance(first, ReductionCell) and first.adjust.mode == "none":
                # the stem's BatchNorm output has ONE reader, the first cell's Activation('relu') (ReductionCell.forward,
                # `rectified`): the ReLU rides in the BatchNorm launch, forward and backward (one aten clamp and one
                # threshold-backward launch per step less); shared_relu() hands the tensor out as it is
                x = self.stem_bn(x, relu_out=True)
                x._xpt_relu_aliases = []
                return self._cells(x, taps)
            return self._cells(self.stem_bn(x), taps)
        x = self.preprocess(image)
        if _conv.usable(x, self.stem_conv, 1.0):          # keras Conv2D(32, 3, strides 2, padding="valid") on the matrix cores
            x = F.pad(x.to(_half()), (0, 0, 0, 0, 0, 5))                   # 3 -> 8 channels (16-byte pixel rows)
            x = _conv.conv2d_same(x, self.stem_conv.weight, None, 2, 1.0, valid=True)
        else:
            x = conv2d_library(x, self.stem_conv.weight, 2, (0, 0))
        return self._cells(self.stem_bn(x), taps)

    def _cells(self, x, taps):
        p = None
        for cell in self.cells:
            x, p = cell(x, x if p is None else p, taps)
        x = shared_relu(x)
        taps.offer(self.final_act_id, x)
        return [taps.found[k] for k in self.TAP_ACTIVATIONS]


# ------------------------------------------------------------------------------------------ Keras weights
# tf.keras.applications.NASNetMobile addresses its variables by LAYER NAME / VARIABLE NAME (stem_conv1/kernel,
# separable_conv_1_normal_left1_0/depthwise_kernel, separable_conv_1_bn_normal_left1_0/moving_mean, ...).  The map below
# gives every parameter / buffer of NASNetMobileEncoder its Keras name and layout; tests/golden/nasnet_mobile_manifest.json
# (names + shapes of the published architecture, written by tools/make_nasnet_manifest.py from the independent
# restatement oracle/ref_nasnet.py) must be covered exactly: every variable lands on one tensor, every tensor is filled.
CELL_BLOCK_IDS = ("stem_1", "stem_2", "0", "1", "2", "3", "reduce_4", "5", "6", "7", "8", "reduce_8", "9", "10", "11", "12")


def _to_keras(kind, t):
    """torch layout -> keras layout: conv OIHW -> HWIO, depthwise [C,1,k,k] -> [k,k,C,1]."""
    if kind == "conv":
        return t.permute(2, 3, 1, 0)
    if kind == "depthwise":
        return t.permute(2, 3, 0, 1)
    return t


def _from_keras(kind, a):
    if kind == "conv":
        return a.permute(3, 2, 0, 1)
    if kind == "depthwise":
        return a.permute(2, 3, 0, 1)
    return a


def keras_variable_map(encoder):
    """{keras variable name: (tensor of the encoder, kind)} with kind in {"conv", "depthwise", "vector"}."""
    out = {}

    def conv(name, module):
        out[f"{name}/kernel"] = (module.weight, "conv")

    def bn(name, module):
        out[f"{name}/gamma"] = (module.weight, "vector")
        out[f"{name}/beta"] = (module.bias, "vector")
        out[f"{name}/moving_mean"] = (module.running_mean, "vector")
        out[f"{name}/moving_variance"] = (module.running_var, "vector")

    def sep_block(block_id, block):
        for k, (sep, norm) in enumerate(((block.conv1, block.bn1), (block.conv2, block.bn2)), start=1):
            out[f"separable_conv_{k}_{block_id}/depthwise_kernel"] = (sep.depthwise.weight, "depthwise")
            out[f"separable_conv_{k}_{block_id}/pointwise_kernel"] = (sep.pointwise.weight, "conv")
            bn(f"separable_conv_{k}_bn_{block_id}", norm)

    conv("stem_conv1", encoder.stem_conv)
    bn("stem_bn1", encoder.stem_bn)
    if len(encoder.cells) != len(CELL_BLOCK_IDS):
        raise WrongInputException("keras_variable_map: not the NASNet-Mobile cell sequence")
    for bid, cell in zip(CELL_BLOCK_IDS, encoder.cells):
        adj = cell.adjust
        if adj.mode == "spatial":
            conv(f"adjust_conv_1_{bid}", adj.conv1)
            conv(f"adjust_conv_2_{bid}", adj.conv2)
            bn(f"adjust_bn_{bid}", adj.bn)
        elif adj.mode == "project":
            conv(f"adjust_conv_projection_{bid}", adj.conv)
            bn(f"adjust_bn_{bid}", adj.bn)
        if isinstance(cell, NormalCell):
            conv(f"normal_conv_1_{bid}", cell.conv)
            bn(f"normal_bn_1_{bid}", cell.bn)
            for side in ("left1", "right1", "left2", "right2", "left5"):
                sep_block(f"normal_{side}_{bid}", getattr(cell, side))
        else:
            conv(f"reduction_conv_1_{bid}", cell.conv)
            bn(f"reduction_bn_1_{bid}", cell.bn)
            for side in ("left1", "right1", "right2", "right3", "left4"):
                sep_block(f"reduction_{side}_{bid}", getattr(cell, side))
    return out


def logical_entries(encoder):
    """{id(tensor): (out_sel, in_sel)} for the tensors of the encoder that carry structurally-zero entries
    (NASNetMobileEncoder.structural_pads): the Keras variables address their logical entries only."""
    return {id(t): (o, i) for t, o, i in encoder.structural_pads()}


def logical_view(t, sel):
    """The logical entries of t (a copy when t carries structural zeros, t itself otherwise)."""
    if sel is None:
        return t
    o, i = sel
    if o is not None:
        t = t.index_select(0, o.to(t.device))
    if i is not None:
        t = t.index_select(1, i.to(t.device))
    return t


def _store_logical(t, sel, value, fill=0.0):
    if sel is None:
        t.copy_(value.to(device=t.device, dtype=t.dtype))
        return
    o, i = sel
    full = torch.full_like(t, fill)
    oo = (o if o is not None else torch.arange(t.shape[0])).to(t.device)
    v = value.to(device=t.device, dtype=t.dtype)
    if i is not None:
        full[oo[:, None], i.to(t.device)[None, :]] = v
    else:
        full[oo] = v
    t.copy_(full)


def export_keras_weights(encoder):
    """{keras variable name: float32 array in the keras layout} of the encoder's current weights."""
    sel = logical_entries(encoder)
    return {name: _to_keras(kind, logical_view(t.detach(), sel.get(id(t)))).contiguous().float().cpu()
            for name, (t, kind) in keras_variable_map(encoder).items()}


def read_keras_weight_file(path):
    """{variable name: numpy array} from an .npz keyed by keras variable names (INTEGRATION.md has the one-line export to
    run where Keras is installed) or from Keras' own NASNet-mobile-no-top.h5 when h5py is importable."""
    import numpy as np
    if str(path).endswith(".npz"):
        with np.load(path) as z:
            return {k[:-2] if k.endswith(":0") else k: z[k] for k in z.files}
    if str(path).endswith((".h5", ".hdf5")):
        try:
            import h5py
        except ImportError as e:
            raise WrongInputException("reading a Keras .h5 file needs h5py (not installed): convert it to .npz, "
                                      "INTEGRATION.md") from e
        out = {}
        with h5py.File(path, "r") as f:
            root = f["model_weights"] if "model_weights" in f else f

            def visit(name, obj):
                if isinstance(obj, h5py.Dataset):
                    parts = name.split("/")
                    var = parts[-1][:-2] if parts[-1].endswith(":0") else parts[-1]
                    out[f"{parts[-2]}/{var}"] = obj[()]
            root.visititems(visit)
        return out
    raise WrongInputException(f"unknown weight file type: {path}")


def load_keras_weights(encoder, weights):
    """Fills the encoder from Keras NASNetMobile(include_top=False) variables (a path or a {name: array} dict).  Strict: a
    missing, unknown or mis-shaped variable raises; nothing is loaded partially."""
    if not isinstance(weights, dict):
        weights = read_keras_weight_file(weights)
    table = keras_variable_map(encoder)
    missing = sorted(set(table) - set(weights))
    unknown = sorted(set(weights) - set(table))
    if missing or unknown:
        raise WrongInputException(f"NASNet-Mobile weights: {len(missing)} variables missing (e.g. {missing[:3]}), "
                                  f"{len(unknown)} not part of the no-top model (e.g. {unknown[:3]})")
    staged = {}
    sel = logical_entries(encoder)
    for name, (t, kind) in table.items():
        a = torch.as_tensor(weights[name])
        want = tuple(_to_keras(kind, logical_view(t, sel.get(id(t)))).shape)
        if tuple(a.shape) != want:
            raise WrongInputException(f"{name}: file has shape {tuple(a.shape)}, the model expects {want}")
        staged[name] = _from_keras(kind, a)
    with torch.no_grad():
        for name, (t, kind) in table.items():
            _store_logical(t, sel.get(id(t)), staged[name], fill=1.0 if name.endswith("/moving_variance") else 0.0)
    return len(staged)


class PretrainedModel:
    """pretrained_nets.py:11-117 interface: PretrainedModel(net_name, use_pt_weight).encoder() builds the module
    whose forward is the reference's `.encode(input_image)`."""
    SUPPORTED = ("NASNetMobile",)

    def __init__(self, net_name, use_pt_weight):
        if net_name not in self.SUPPORTED:
            raise WrongInputException(f"Pretrained backbone '{net_name}' is outside this build's hot path "
                                      f"(available: {self.SUPPORTED})")
        self.weight_file = None
        if use_pt_weight:
            # weights="imagenet" (pretrained_nets.py:23) downloads from the Keras storage bucket; offline the user
            # supplies the same variables as a file (one-line export where Keras is installed)
            self.weight_file = __import__("os").environ.get("XPT_NASNET_WEIGHTS", "")
            if not self.weight_file:
                raise WrongInputException("ImageNet weights (Keras storage bucket download, pretrained_nets.py:23) are not "
                                          "obtainable offline: point XPT_NASNET_WEIGHTS at an .npz / .h5 of the Keras "
                                          "NASNetMobile(include_top=False) variables, or set opts.PRETRAINED_WEIGHT = False")
        self.net_name = net_name

    def encoder(self):
        net = NASNetMobileEncoder()
        if self.weight_file:
            load_keras_weights(net, self.weight_file)
        return net
