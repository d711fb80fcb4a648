"""optimizer_factory with the reference's signature (model/model_util/optimizers.py:7-13) on flat HBM buffers.

MI355X-first layout: every trainable parameter is a view into ONE flat fp32 buffer, every gradient a view into
a second one (288 GB of HBM makes the duplication irrelevant).  The flat gradient buffer is at the same time the
RCCL all-reduce bucket of the data-parallel step, and Adam is one fused gfx950 kernel over the four flat streams
(xpt_adam_step).  Keras semantics are kept: epsilon = 1e-7 added to the UNcorrected sqrt(v); no weight decay, no
clipping, constant learning rate; a fresh optimizer state per training-plan row (model_main.py:95).
"""
import torch

from ...hip import lib as _lib
from ...utils.util_class import WrongInputException


class FlatParameters:
    """Re-homes `params` (list of nn.Parameter) into flat data / grad buffers, preserving each parameter's memory
    format (channels_last conv kernels stay channels_last views)."""

    def __init__(self, params, align=8, groups=None):   # 8 elements: 16-byte aligned bf16 shadow rows for the matrix-core kernels
        params = [p for p in params if p.requires_grad]
        if not params:
            raise WrongInputException("no trainable parameters")
        params = self._grouped(params, groups)
        self.params = params
        dev, dtype = params[0].device, torch.float32
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + align - 1) // align * align
        self.numel = total
        self.data = torch.zeros(total, dtype=dtype, device=dev)
        self.grad = torch.zeros(total, dtype=dtype, device=dev)
        self.grad_views = []
        # bf16 copy of every weight, refreshed by the fused Adam kernel: the bf16 GEMMs / convolutions read it directly
        # instead of launching one fp32->bf16 cast per layer and step (and one bf16->fp32 cast per gradient)
        self.shadow = torch.zeros(total, dtype=_lib.half(), device=dev) if dev.type == "cuda" else None
        for p, off in zip(params, offsets):
            dview, gview = self._view(self.data, p, off), self._view(self.grad, p, off)
            dview.copy_(p.data)
            p.data = dview
            p.grad = None            # autograd then hands over ("steals") each gradient tensor without an add kernel
            self.grad_views.append(gview)
            if self.shadow is not None:
                p.shadow_bf16 = self._view(self.shadow, p, off)
                kxk = p.dim() == 4 and p.shape[2] * p.shape[3] > 1
                if gview.is_contiguous() and (not kxk or gview.is_contiguous(memory_format=torch.channels_last)):
                    # destination of the deferred gradient finishing (hip/ops.py GradSink): the gfx950 backward kernels
                    # of this parameter leave partial sums behind and ONE launch per step adds them into this view.
                    # (A k x k kernel stored plain NCHW gets NO sink: the dense convolution's partials are laid out
                    # [cout][kh][kw][cin] (hip/conv.py) and would land permuted; its gradient takes the non-deferred path.)
                    p.flat_grad = gview
                elif p.dim() == 4 and gview.is_contiguous(memory_format=torch.channels_last):
                    # channels_last k x k kernels: memory order [cout][kh][kw][cin] -- exactly the layout of the dense
                    # convolution's weight-gradient partials (hip/conv.py), so the destination is the flat chunk itself
                    p.flat_grad = self.grad[off:off + p.numel()]
        if self.shadow is not None:
            self.shadow.copy_(self.data)
        self.offsets = offsets

    @staticmethod
    def _grouped(params, groups):
        """Lays the members of every group (same-shape weights a layer consumes as ONE strided batch, e.g. the pointwise
        weights of a NASNet cell stage) next to each other, in group order, at the position of the group's first
        member: equal shapes then mean equal spacing, so [n, cout, cin] is a strided VIEW of the flat buffers."""
        if not groups:
            return params
        present = {id(p) for p in params}
        leader, member = {}, set()
        for group in groups:
            group = [p for p in group if id(p) in present]
            if len(group) < 2 or any(p.shape != group[0].shape for p in group) or any(id(p) in member for p in group):
                continue
            first = min(group, key=lambda q: next(i for i, r in enumerate(params) if r is q))
            leader[id(first)] = group
            member.update(id(p) for p in group)
        out = []
        for p in params:
            if id(p) in leader:
                out.extend(leader[id(p)])
            elif id(p) not in member:
                out.append(p)
        return out

    def gather_grads(self):
        """Moves the per-parameter gradients autograd produced into the flat buffer with one multi-tensor copy
        (instead of ~700 accumulate-into-view kernels per step) and releases them.  Parameters that received no
        gradient keep the zeros the optimizer left behind."""
        srcs, dsts = [], []
        for p, gv in zip(self.params, self.grad_views):
            if p.grad is not None:
                srcs.append(p.grad)
                dsts.append(gv)
                p.grad = None
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    @staticmethod
    def _view(flat, p, off):
        n = p.numel()
        chunk = flat[off:off + n]
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            co, ci, kh, kw = p.shape
            return chunk.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return chunk.view(p.shape)

    def zero_grad(self):
        self.grad.zero_()



class KerasAdam:
    """tf.optimizers.Adam(learning_rate) over FlatParameters."""

    def __init__(self, learning_rate, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), beta_1, beta_2, epsilon
        self.flat = None
        self.m = self.v = self.step_count = None
        self.l2_terms = []            # (parameters, coefficient): gradient coefficient * w added before the update
        self._l2_ranges = None

    def add_l2(self, params, coefficient):
        """Gradient of coefficient * sum_w l2_loss(w) (= coefficient * w) for a run of parameters that is contiguous in
        the flat buffers (one network): added to the flat gradient by ONE launch per step, just before the update
        (the `flow_reg` loss, losses.py:522-533; its value is reported by L2Regularizer)."""
        params = [p for p in params if p.requires_grad]
        if params and coefficient:
            self.l2_terms.append((params, float(coefficient)))
            self._l2_ranges = None

    def _ranges(self):
        if self._l2_ranges is None:
            f = self.flat
            index = {id(p): i for i, p in enumerate(f.params)}
            ranges = []
            for params, coef in self.l2_terms:
                idx = sorted(index[id(p)] for p in params)
                if idx != list(range(idx[0], idx[0] + len(idx))):
                    raise WrongInputException("add_l2: the parameters are not a contiguous run of the flat buffer")
                last = f.params[idx[-1]]
                ranges.append((f.offsets[idx[0]], f.offsets[idx[-1]] + last.numel(), coef))   # alignment gaps hold zeros
            self._l2_ranges = ranges
        return self._l2_ranges

    def bind(self, params, groups=None):
        self.flat = params if isinstance(params, FlatParameters) else FlatParameters(list(params), groups=groups)
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=self.flat.data.device)
        return self.flat

    def begin_step(self):
        """The step counter of a step that is applied in pieces (apply_gradients(..., lo, hi, bump=False))."""
        self.step_count += 1

    def _piece(self, lo, hi):
        f = self.flat
        hi = f.numel if hi is None else hi
        if not (0 <= lo < hi <= f.numel) or lo % 4:       # (the update kernels move 16-byte vectors from the piece's base)
            raise WrongInputException(f"apply_gradients: bad range [{lo}, {hi}) of {f.numel}")
        if (lo, hi) != (0, f.numel) and self._ranges():
            raise WrongInputException("apply_gradients: an L2 term needs the whole buffer in one piece")
        return lo, hi

    def apply_gradients(self, grad_scale=1.0, zero_grad=True, lo=0, hi=None, bump=True):
        """optimizer.apply_gradients (train_val.py:86) on the flat buffers; hipGraph-capturable.  [lo, hi): a piece of the
        buffers (element-wise update: pieces are independent; the trainer that applies a step in pieces counts the step once,
        begin_step(), and passes bump=False)."""
        f = self.flat
        lo, hi = self._piece(lo, hi)
        if bump:
            self.step_count += 1
        for a, b, coef in self._ranges():
            f.grad[a:b].add_(f.data[a:b], alpha=coef / float(grad_scale))
        if f.data.is_cuda:
            lib = _lib.load()
            _lib.check(lib.xpt_adam_step(f.data.data_ptr() + 4 * lo, f.grad.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                                         self.v.data_ptr() + 4 * lo, hi - lo, self.step_count.data_ptr(), self.lr, self.b1,
                                         self.b2, self.eps, float(grad_scale), int(zero_grad),
                                         None if f.shadow is None else f.shadow.data_ptr() + f.shadow.element_size() * lo,
                                         torch.cuda.current_stream().cuda_stream),
                       "xpt_adam_step")
            return
        if (lo, hi) != (0, f.numel):
            raise WrongInputException("apply_gradients: pieces are a device feature")
        # host tensors (CPU-only unit tests of the data-parallel host logic): same arithmetic with tensor ops
        with torch.no_grad():
            t = self.step_count
            lr_t = self.lr * torch.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
            g = f.grad * grad_scale
            self.m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            f.data.sub_(lr_t * self.m / (torch.sqrt(self.v) + self.eps))
            if zero_grad:
                f.grad.zero_()


class KerasSGD(KerasAdam):
    """tf.optimizers.SGD(learning_rate) (optimizers.py:10-11: momentum 0, constant rate) over the same flat buffers; the
    trainers see the KerasAdam interface (bind / add_l2 / apply_gradients; `m` and `v` are empty)."""

    def __init__(self, learning_rate):
        super().__init__(learning_rate)

    def bind(self, params, groups=None):
        self.flat = params if isinstance(params, FlatParameters) else FlatParameters(list(params), groups=groups)
        self.m = torch.zeros(0, dtype=torch.float32, device=self.flat.data.device)
        self.v = torch.zeros(0, dtype=torch.float32, device=self.flat.data.device)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=self.flat.data.device)
        return self.flat

    def apply_gradients(self, grad_scale=1.0, zero_grad=True, lo=0, hi=None, bump=True):
        f = self.flat
        lo, hi = self._piece(lo, hi)
        if bump:
            self.step_count += 1
        for a, b, coef in self._ranges():
            f.grad[a:b].add_(f.data[a:b], alpha=coef / float(grad_scale))
        if f.data.is_cuda:
            lib = _lib.load()
            _lib.check(lib.xpt_sgd_step(f.data.data_ptr() + 4 * lo, f.grad.data_ptr() + 4 * lo, hi - lo, self.lr, float(grad_scale),
                                        int(zero_grad),
                                        None if f.shadow is None else f.shadow.data_ptr() + f.shadow.element_size() * lo,
                                        torch.cuda.current_stream().cuda_stream), "xpt_sgd_step")
            return
        if (lo, hi) != (0, f.numel):
            raise WrongInputException("apply_gradients: pieces are a device feature")
        with torch.no_grad():
            f.data.sub_(f.grad, alpha=self.lr * float(grad_scale))
            if zero_grad:
                f.grad.zero_()


def optimizer_factory(opt_name, basic_lr, epoch=0):
    if opt_name == "adam_constant":
        return KerasAdam(learning_rate=basic_lr)
    if opt_name == "sgd_constant":
        return KerasSGD(learning_rate=basic_lr)
    raise WrongInputException(f"{opt_name} is NOT an available optimizer name")
