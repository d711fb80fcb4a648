"""Train / validation loop with the reference's public names (model/train_val.py:12-264):
train_val_factory(mode, model, loss_object, steps_per_epoch, stereo, augmenter, optimizer) -> (trainer, validater),
`.run_an_epoch(dataset) -> (DataFrame, hours)`, `.run_a_batch(features)`, `.train_a_step(features)`.

MI355X-first differences:
  * "graph" mode = the whole training step (nets fwd, HIP synthesis + loss kernels, backward, fused Adam) captured
    ONCE into a hipGraph and replayed on static input buffers -- the counterpart of the reference's @tf.function
    (train_val.py:100-102) without a tracing compiler;
  * "distributed" mode = one process per GPU; the flat gradient buffer is all-reduced (SUM) with RCCL between the
    backward and the optimizer (the reference's implicit MirroredStrategy all-reduce, train_val.py:86,114);
  * per-step metrics stay on the device and are fetched once per epoch (the reference syncs the host every step in
    merge_results, train_val.py:157-177).
"""
import os

import numpy as np
import pandas as pd
import torch

from ..config import opts
from ..hip import ops as _ops
from ..utils import util_class as uc
from ..utils import util_funcs as uf
from .model_util.distributer import DistributionStrategy
from .model_util.optimizers import KerasAdam


def train_val_factory(mode_sel, model, loss_object, steps_per_epoch, stereo, augmenter, optimizer):
    table = {"eager": (ModelTrainer, ModelValidater), "graph": (ModelTrainerGraph, ModelValidaterGraph),
             "distributed": (ModelTrainerDistrib, ModelValidaterDistrib)}
    if mode_sel not in table:
        raise uc.WrongInputException(f"training mode '{mode_sel}' is NOT available")
    configure_backend()
    trainer_cls, validater_cls = table[mode_sel]
    trainer = trainer_cls(model, loss_object, steps_per_epoch, stereo, augmenter, optimizer)
    validater = validater_cls(model, loss_object, steps_per_epoch, stereo)
    return trainer, validater


def detach_tree(obj):
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, dict):
        return {k: detach_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(detach_tree(v) for v in obj)
    return obj


def configure_backend():
    """Library-convolution settings for every trainer.

    * MIOpen "find" is on in its FAST mode (MIOPEN_FIND_MODE=2 unless the environment says otherwise): only the ~30 dense
      3x3 / 5x5 / 7x7 convolutions of the decoder and PoseNet still go through MIOpen (depthwise and pointwise layers are
      gfx950 kernels of this repo), so the search takes seconds, not the minutes of the exhaustive find over all ~230
      NASNet layer shapes, and the kernels it picks are 1.15 ms per step faster than the immediate-mode heuristics.
    * MIOpen's bf16 weight-gradient solvers return garbage from the second replay of a captured hipGraph on ROCm 7.0 /
      torch 2.10 (tests/test_graph_replay.py); weight gradients therefore run in fp32 (layer_ops._ConvFp32WeightGrad).
      XPT_MIOPEN_DETERMINISTIC=1 selects the (25x slower) deterministic solvers instead."""
    import os
    os.environ.setdefault("MIOPEN_FIND_MODE", "2")
    torch.backends.cudnn.deterministic = bool(int(os.environ.get("XPT_MIOPEN_DETERMINISTIC", "0")))
    torch.backends.cudnn.benchmark = bool(getattr(opts, "MIOPEN_FIND", True))
    # GEMMs (the 1x1 convolutions) through rocBLAS: hipBLASLt launches cost ~2 ms EACH when replayed from a hipGraph
    # on this stack (1.18 s / step measured), rocBLAS replays at kernel speed
    if torch.cuda.is_available():
        torch.backends.cuda.preferred_blas_library("cublas")


class TrainValBase:
    def __init__(self, model, loss_object, steps_per_epoch, stereo, augmenter=None, optimizer=None):
        self.model = model
        self.augmenter = augmenter
        self.loss_object = loss_object
        self.train_val_name = "train_val"
        self.steps_per_epoch = steps_per_epoch
        self.stereo = stereo
        self.optimizer = optimizer
        self.print_stride = max(int(getattr(opts, "PRINT_STRIDE", 50)), 1)

    def set_name(self, name):
        self.train_val_name = name

    def run_an_epoch(self, dataset):
        """train_val.py:43-64 -> (DataFrame of per-step results, hours)."""
        results = []
        with uc.DurationTime() as epoch_time:
            for step, features in enumerate(dataset):
                preds, loss, loss_by_type = self.run_a_batch(features)
                results.append(self.step_metrics(features, preds, loss, loss_by_type))
                if step % self.print_stride == 0:
                    uf.print_progress_status(f"    {self.train_val_name} {step}/{self.steps_per_epoch} steps, "
                                             f"loss = {float(results[-1]['loss']):1.4f}...")
                if self.steps_per_epoch and step + 1 >= self.steps_per_epoch:
                    break
        print("")
        results = pd.DataFrame(fetch_results(results))
        mean_results = results.mean(axis=0).to_dict()
        message = f"[{self.train_val_name} Epoch MEAN], result: " + ", ".join(f"{k}={v:1.4f}" for k, v in mean_results.items())
        print(message, "\n\n")
        return results, epoch_time.duration / 3600.

    def run_a_batch(self, features):
        raise NotImplementedError()

    def step_metrics(self, features, preds, loss, loss_by_type):
        """The per-step record of run_an_epoch (train_val.py:157-177 merge_results)."""
        return merge_results(features, preds, loss, loss_by_type, self.stereo)


class ModelTrainer(TrainValBase):
    def __init__(self, model, loss_object, steps_per_epoch, stereo, augmenter, optimizer):
        super().__init__(model, loss_object, steps_per_epoch, stereo, augmenter, optimizer)
        self.set_name("Train (eager)")
        if isinstance(optimizer, KerasAdam) and optimizer.flat is None:
            groups = model.weight_groups() if hasattr(model, "weight_groups") else None
            optimizer.bind(model.trainable_weights(), groups=groups)

    def run_a_batch(self, features):
        return self.train_a_step(features)

    def loss_seed(self, total_loss):
        """Seed of the backward pass: None (= 1), or the static loss scale of the fp16 configuration (config.LOSS_SCALE_FP16;
        a persistent tensor: no launch inside the captured step).  grad_unscale() takes it out again in the optimizer."""
        forced = os.environ.get("XPT_TEST_FORCE_SEED")      # tests: a seed the loss object's gradient hint does not know about
        if forced and total_loss.is_cuda:
            seed = getattr(self, "_loss_seed", None)
            if seed is None:
                seed = self._loss_seed = torch.full_like(total_loss, float(forced))
            return seed
        if opts.CONV_DTYPE != "fp16" or not total_loss.is_cuda:
            return None
        seed = getattr(self, "_loss_seed", None)
        if seed is None or seed.device != total_loss.device or seed.dtype != total_loss.dtype:
            seed = self._loss_seed = torch.full_like(total_loss, float(opts.LOSS_SCALE_FP16))
        return seed

    @staticmethod
    def grad_unscale():
        forced = os.environ.get("XPT_TEST_FORCE_SEED")
        if forced:
            return 1.0 / float(forced)
        return 1.0 / float(opts.LOSS_SCALE_FP16) if opts.CONV_DTYPE == "fp16" else 1.0

    def forward_backward(self, features):
        if self.augmenter is not None:
            features = self.augmenter(features)
        preds = self.model(features)
        total_loss, loss_by_type = self.loss_object(preds, features)
        total_loss.backward(gradient=self.loss_seed(total_loss))
        if total_loss.is_cuda:
            _ops.grad_sink.flush()       # one launch finishes every deferred parameter gradient into the flat buffer
        if self.optimizer is not None and getattr(self.optimizer, "flat", None) is not None:
            self.optimizer.flat.gather_grads()
        # hand back detached values only: a live autograd graph would keep its AccumulateGrad nodes (and their
        # stream) alive across iterations, which breaks hipGraph capture of the next step
        return detach_tree(preds), total_loss.detach(), {k: v.detach() for k, v in loss_by_type.items()}

    # ---- the backward pass in two phases (cut between the decoder and the encoder): the data-parallel trainer starts the
    # all-reduce of the first phase's gradients between them, the graph trainer their optimizer update (ModelTrainerGraph)
    def _find_early_bucket(self):
        """Offset in the flat buffers where the early gradients start: the encoders' parameters (finished by the second
        backward phase) must form the head of the buffer, everything else its tail; None when there is no such cut."""
        flat = getattr(self.optimizer, "flat", None)
        if flat is None or not hasattr(self.model, "set_backward_cut"):
            return None
        late = {id(p) for p in self.model.set_backward_cut(True)}
        flags = [id(p) in late for p in flat.params]
        k = sum(flags)
        if k == 0 or k == len(flags) or not all(flags[:k]):
            self.model.set_backward_cut(False)
            return None
        return flat.offsets[k]

    def backward_first(self, features, finish=True):
        """Forward pass and the backward of everything behind the encoders -> (carry for backward_second, outputs).
        finish=False: the caller finishes the deferred gradients of this phase itself (on another stream)."""
        if self.augmenter is not None:
            features = self.augmenter(features)
        preds = self.model(features)
        total_loss, loss_by_type = self.loss_object(preds, features)
        total_loss.backward(gradient=self.loss_seed(total_loss))
        carry = self.model.take_backward_cuts()
        if finish:
            if total_loss.is_cuda:
                _ops.grad_sink.flush()
            self.optimizer.flat.gather_grads()
        return carry, (detach_tree(preds), total_loss.detach(), {k: v.detach() for k, v in loss_by_type.items()})

    def backward_second(self, carry):
        """The encoders' backward, resumed from the gradients the first phase left at the cut."""
        if carry:
            torch.autograd.backward([t for t, _ in carry], [g for _, g in carry])
            if carry[0][0].is_cuda:
                _ops.grad_sink.flush()
            self.optimizer.flat.gather_grads()

    def reduce_gradients(self):
        pass

    def repair_flagged(self):
        """Called when a captured step failed the replay check: the 4-D convolution weights among the parameters the
        check named get their weight gradient from the patch-gather GEMM (layer_ops.unfolded_weight_grad) instead of a
        library solver from now on.  Returns the number of newly marked weights (0: nothing left to try)."""
        marked = 0
        for p in getattr(self, "_flagged_params", []):
            if p.dim() == 4 and p.shape[2] * p.shape[3] > 1 and not getattr(p, "xpt_safe_wgrad", False):
                p.xpt_safe_wgrad = True
                marked += 1
        self._flagged_params = []
        return marked

    def state_segments(self):
        """Lengths (alignment gaps included) of the parameters inside the flat buffers, for the replay check."""
        flat = self.optimizer.flat
        ends = list(flat.offsets[1:]) + [flat.numel]
        return torch.tensor([e - o for o, e in zip(flat.offsets, ends)], dtype=torch.int64, device=flat.data.device)

    def describe_state(self, index, bad):
        """Names the parameters behind the flagged elements of a flat state tensor (for the replay-check message)."""
        flat = self.optimizer.flat
        if bad.numel() != flat.numel:
            return ""
        import bisect
        names = {id(q): f"{net}.{n}" for net, m in self.model.models.items() for n, q in m.named_parameters()}
        label = ["value", "gradient", "first moment", "second moment"][index] if index < 4 else f"state {index}"
        hits = {}
        self._flagged_params = []
        for off in torch.nonzero(bad.reshape(-1))[:, 0].tolist()[::max(int(bad.sum()) // 4096, 1)]:
            i = bisect.bisect_right(flat.offsets, off) - 1
            key = names.get(id(flat.params[i]), f"param {i}")
            if key not in hits:
                self._flagged_params.append(flat.params[i])
            hits[key] = hits.get(key, 0) + 1
        return f" [{label} of " + ", ".join(f"{k}{tuple(self._shape(k, names, flat))}" for k in list(hits)[:6]) + \
               (" ..." if len(hits) > 6 else "") + "]"

    @staticmethod
    def _shape(name, names, flat):
        for p in flat.params:
            if names.get(id(p)) == name:
                return p.shape
        return ()

    def _pin_hooks(self):
        """(enter, leave) for _StepGraph's replay check when the augmenter can repeat its draws (the fused default chain)."""
        aug = self.augmenter
        if aug is not None and hasattr(aug, "can_pin") and aug.can_pin():
            return (aug.pin_draws, aug.unpin_draws)
        return None

    def optimizer_state(self):
        opt = self.optimizer
        state = [opt.flat.data, opt.flat.grad, opt.m, opt.v, opt.step_count]
        if getattr(opt.flat, "shadow", None) is not None:
            state.append(opt.flat.shadow)
        return state

    def train_a_step(self, features):
        """train_val.py:78-92: augment -> model -> loss -> gradients -> optimizer.apply_gradients."""
        out = self.forward_backward(features)
        self.reduce_gradients()
        self.optimizer.apply_gradients(grad_scale=self.grad_unscale())
        return out


_DEBUG_REPLAY = __import__("os").environ.get("XPT_DEBUG_REPLAY", "0") == "1"
_OWN_MULTI_COPY = __import__("os").environ.get("XPT_DEBUG_FOREACH_COPY", "0") != "1"     # A/B: torch._foreach_copy_ for the batch refresh


class _GraphPair:
    """Two hipGraphs replayed back to back, with a host callback between them."""

    def __init__(self, first, second):
        self.first, self.second = first, second

    def replay(self, between=None):
        self.first.replay()
        if between is not None:
            between()
        self.second.replay()


class _StepGraph:
    """Captures fn(static_features) into a hipGraph; replays it after copying a new batch into the static buffers."""

    def __init__(self, fn, warmup=3, state=None, describe=None, segments=None, repair=None, reference=False,
                 phases=None, between=None, pin=None):
        self.fn = fn
        # phases = (first, second): the step is captured as TWO graphs sharing one memory pool, first(features) ->
        # (carry, outputs) and second(carry); `between` runs between their replays (the data-parallel trainer launches
        # the all-reduce of the gradients the first phase finished there).  fn must be second(first(.)) without it.
        self.phases = phases
        self.between = between
        self.warmup = warmup
        self.state = state                     # callable -> list of tensors the warm-up runs must not change
        self.describe = describe               # (state index, bad-element mask) -> text for the replay-check message
        self.segments = segments               # callable -> int64 lengths of the parameters inside a flat state tensor
        self.repair = repair                   # callable -> number of layers switched to replay-safe gradients
        self.repairs = 0
        self.reference = reference             # the step draws no random numbers: compare the replays with an eager step
        # pin = (enter, leave): make the step's random draws repeat / draw freshly again (the fused augmentation kernel reads
        # pinned uniforms while a device flag is set): during the replay check an augmented step is then held to the same
        # replay-vs-replay and replay-vs-eager comparisons as a step without random numbers
        self.pin = pin
        self.graph = None
        self.static_in = None
        self.static_out = None
        self.signature = None
        self.cache = {}                        # input signature -> (graph, static inputs, static outputs)
        self.eager_fallback = False            # set when no capture of the step survives the replay check
        self.library_path = False              # the step contains library (MIOpen) convolutions: executed eagerly by design

    @staticmethod
    def _sig(features):
        return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in features.items() if torch.is_tensor(v)))

    def __call__(self, features):
        if self.eager_fallback:
            return self.fn(features)
        sig = self._sig(features)
        entry = self.cache.get(sig)
        if entry is None:
            # one captured graph per input signature (the mixed pretrain stream cycles through a handful of image sizes,
            # config-example.py:25-30): captured on first sight, replayed ever after
            self._capture(features, sig)
            if self.eager_fallback:
                return self.fn(features)
            self.cache[sig] = (self.graph, self.static_in, self.static_out)
        else:
            self.graph, self.static_in, self.static_out = entry
            self.signature = sig
            # the new batch into the graph's static inputs: ONE multi-tensor copy launch instead of one copy per feature
            fits = lambda k, v: (features[k].is_cuda and features[k].dtype == v.dtype and features[k].shape == v.shape      # noqa: E731
                                 and features[k].is_contiguous() and v.is_contiguous() and v.numel() > 0)
            dst = [v for k, v in self.static_in.items() if fits(k, v)]
            src = [features[k] for k, v in self.static_in.items() if fits(k, v)]
            if len(dst) > 1 and _OWN_MULTI_COPY:
                _ops.multi_copy(dst, src)                       # (xpt_multi_copy: 16 MB + six small tensors in one launch)
            elif len(dst) > 1:
                torch._foreach_copy_(dst, src)
            rest = [k for k, v in self.static_in.items() if not fits(k, v)]
            for k in (rest if len(dst) > 1 else self.static_in):
                self.static_in[k].copy_(features[k], non_blocking=True)
        if isinstance(self.graph, _GraphPair):
            self.graph.replay(self.between)
        else:
            self.graph.replay()
        return self.static_out

    def _capture(self, features, sig):
        self.static_in = {k: v.clone() for k, v in features.items() if torch.is_tensor(v)}
        state = self.state() if self.state is not None else []
        saved = [t.clone() for t in state]     # warm-up executes real steps: roll the weights / moments back after it
        from .model_util import layer_ops as _lo
        library_before = _lo.LIBRARY_CONV_CALLS[0]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):           # MIOpen find / workspace allocation happens here, not in capture
                self.fn(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.library_path = _lo.LIBRARY_CONV_CALLS[0] > library_before
        policy = getattr(opts, "CAPTURE_LIBRARY_STEPS", "audit")

        def run_eagerly(why):
            import sys
            for t, s in zip(state, saved):
                t.copy_(s)
            print(f"[StepGraph] the step contains library (MIOpen) convolutions {why}: it is executed eagerly, not captured",
                  file=sys.stderr, flush=True)
            self.eager_fallback = True
            self.graph = None

        if self.library_path and policy is False:
            # The step went through MIOpen (fp32 mode, PWC-Net's dilated / transposed convolutions).  Solvers that zero a
            # workspace do it with a memset node, and memset nodes of a captured graph write garbage from the second replay on
            # with this runtime; which solver the library picks varies from run to run.  Round 2 therefore
            # never captured such steps; since round 3 the captured graph is AUDITED (below) and a library step is captured
            # exactly when its graph holds no memset node ("audit", the default).  False restores round 2's rule, True
            # captures whatever the audit says.
            run_eagerly("and opts.CAPTURE_LIBRARY_STEPS is False")
            return
        import torch.distributed as _dist
        # With a process group alive the collective library's watchdog thread polls events of (completed) collectives at its
        # own pace: in the default "global" capture mode such a query from another thread invalidates the capture.  Capture
        # thread-locally instead (only THIS thread's calls are held to the capture rules) after draining the device.
        mode = "global"
        if _dist.is_available() and _dist.is_initialized():
            torch.cuda.synchronize()
            mode = "thread_local"
        # keep_graph: the hipGraph_t stays available, so that the capture can be AUDITED before it is instantiated
        if self.phases is None:
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.static_out = self.fn(self.static_in)
            captured = [self.graph]
        else:
            first, second = torch.cuda.CUDAGraph(keep_graph=True), torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(first, capture_error_mode=mode):
                carry, self.static_out = self.phases[0](self.static_in)
            with torch.cuda.graph(second, pool=first.pool(), capture_error_mode=mode):   # same capture stream (torch's default), same pool
                self.phases[1](carry)
            del carry
            self.graph = _GraphPair(first, second)
            captured = [first, second]
        # Node audit: a memset node replays wrongly on this runtime -- the invariant "nothing inside a
        # captured step depends on a memset" is ENFORCED here, not remembered: one torch.zeros / multi-block sum / library
        # workspace clear inside the step and the capture is refused.
        self.census = {}
        for g in captured:
            for k, v in _ops.graph_census(g).items():
                self.census[k] = self.census.get(k, 0) + v
        if self.census.get("memset", 0) and self.library_path and policy != True:      # noqa: E712
            # a library solver that clears its workspace with a memset was picked this time: this step is not replayable
            del captured
            self.static_out = None
            run_eagerly(f"whose capture holds {self.census['memset']} memset node(s) ({self.census})")
            return
        if self.census.get("memset", 0) and not self.library_path and not getattr(opts, "ALLOW_MEMSET_NODES", False):
            for t, s_ in zip(state, saved):
                t.copy_(s_)
            self.graph = None
            raise RuntimeError(f"[StepGraph] the captured step contains {self.census['memset']} memset node(s) "
                               f"({self.census}): memset nodes of a hipGraph write garbage from the second replay on with "
                               "this runtime -- replace the zero-fill (torch.zeros / sum over several blocks / library "
                               "workspace clear) by a kernel, or run the step eagerly")
        for g in captured:
            g.instantiate()
        for t, s in zip(state, saved):
            t.copy_(s)
        self.signature = sig
        check = state and __import__("os").environ.get("XPT_REPLAY_CHECK", "1") != "0"      # 0: diagnostics only
        report = None
        if check:
            was_reference = self.reference
            if self.pin is not None:
                self.pin[0]()
                self.reference = True
            try:
                report = self._replay_report(state, saved)
            finally:
                if self.pin is not None:
                    self.pin[1]()
                self.reference = was_reference
        if report is not None:
            # Some library convolution solvers return garbage from the second replay of a captured graph on this stack
            # (DESIGN.md section 6); which solver MIOpen's find picks can vary from process to process.  Fall back to
            # the immediate-mode heuristics (whose choices are covered by tests/test_graph_replay.py) and recapture;
            # if that capture fails the check too, run the step eagerly rather than train on garbage.
            import sys
            marked = self.repair() if (self.repair is not None and self.repairs < 6) else 0
            if marked:
                self.repairs += 1
                print(f"[StepGraph] captured step fails the replay check ({report}): {marked} convolution(s) switched "
                      f"to the GEMM weight gradient, recapturing", file=sys.stderr, flush=True)
                self.graph = None
                return self._capture(features, sig)
            if torch.backends.cudnn.benchmark:
                print(f"[StepGraph] captured step fails the replay check ({report}): MIOpen find off, recapturing",
                      file=sys.stderr, flush=True)
                torch.backends.cudnn.benchmark = False
                opts.MIOPEN_FIND = False
                self.graph = None
                return self._capture(features, sig)
            self.graph = None
            if not self.library_path and not bool(int(__import__("os").environ.get("XPT_ALLOW_EAGER_FALLBACK", "0"))):
                # a captured step that does not reproduce its own eager execution is a defect, not an operating mode:
                # stop (tests and bench.py fail on this); XPT_ALLOW_EAGER_FALLBACK=1 trades that for a ~4x slower run
                raise RuntimeError(f"[StepGraph] the captured training step fails the replay check: {report}")
            why = "library convolutions on the path" if self.library_path else "XPT_ALLOW_EAGER_FALLBACK=1"
            print(f"[StepGraph] captured step fails the replay check again ({report}): {why}, running the step EAGERLY",
                  file=sys.stderr, flush=True)
            self.eager_fallback = True

    def _replay_report(self, state, saved, replays=3):
        """Replays the fresh graph a few times from the saved state and checks that
        (a) parameters, moments and gradients stay finite and bounded, (b) every later replay reproduces the first one
        parameter by parameter: the largest deviation inside a parameter must stay below half of that parameter's largest
        gradient magnitude (the failures this guards against are correct on the first replay and yield garbage -- 1e25 ...
        inf, sometimes finite tiles -- from the second on; honest run-to-run noise is orders of magnitude below the bar:
        the hand-written kernels are bit-repeatable, rocBLAS split-K atomics move single elements by ~1e-3 relative, and
        for a rectified stereo pair ulp-level depth differences flip the sampler's validity of border rows (DESIGN.md
        section 8), worth a few percent of a parameter's largest gradient).
        Returns None when all is well, else a short description of what went wrong."""
        report, first, last, last_loss = None, None, None, None
        lengths = self.segments() if self.segments is not None else None
        for rep in range(replays):
            self.graph.replay()
            torch.cuda.synchronize()
            for i, t in enumerate(state):
                if report is None and t.is_floating_point():
                    bad = ~torch.isfinite(t) | (t.abs() >= 1e8)
                    if bool(bad.any()):
                        report = f"replay {rep}: {int(bad.sum())} of {t.numel()} elements of state tensor {i} non-finite"
                        if self.describe is not None:
                            report += self.describe(i, bad)
            if report is None and first is None:
                first = [t.clone() for t in state]
            elif report is None and self.reference:
                # (steps with an augmenter draw new random crops / flips / colours on every replay: their gradients differ
                # from replay to replay by design, so only (a) applies to them -- the kernels they run are the ones
                # tests/test_graph_replay.py holds to bit-equality with eager on the un-augmented step)
                report = self._compare_replays(f"replay {rep} differs from replay 0", first, state, lengths)
                last, last_loss = [t.clone() for t in state], self._scalar_loss(self.static_out)
            for t, s in zip(state, saved):
                t.copy_(s)
        if report is None and last is not None and self.reference:
            # (c) steps without random draws (no augmenter): the same step executed EAGERLY from the same state must give
            # the same loss and gradients of the same magnitude -- a captured step whose replays agree with each other can
            # still be wrong, consistently.  (With an augmenter the draws of a replay and of an eager run only line up if
            # the generator state is reset in between -- tools/rng_align_probe.py -- and torch.cuda.set_rng_state() after a
            # capture leaves the captured graph drawing the SAME numbers on every later replay, so that is not done.)
            out = self.fn(self.static_in)
            torch.cuda.synchronize()
            ref_loss = self._scalar_loss(out)
            if _DEBUG_REPLAY:
                import sys
                print(f"[StepGraph debug] replay loss {last_loss}, eager loss {ref_loss}, benchmark "
                      f"{torch.backends.cudnn.benchmark}, weights checksum {float(saved[0].double().abs().sum()):.6f}",
                      file=sys.stderr, flush=True)
            if ref_loss is not None and last_loss is not None and \
                    not abs(last_loss - ref_loss) <= 2e-2 * abs(ref_loss) + 1e-6:
                report = f"replayed loss {last_loss:.6g} but eager loss {ref_loss:.6g} from the same state"
            else:
                report = self._compare_replays("replay differs from the eager step", state, last, lengths)
            del out
            for t, s in zip(state, saved):
                t.copy_(s)
        torch.cuda.synchronize()
        return report

    @staticmethod
    def _scalar_loss(out):
        if isinstance(out, (tuple, list)) and len(out) > 1 and torch.is_tensor(out[1]) and out[1].numel() == 1:
            return float(out[1])
        return None

    def _compare_replays(self, what, first, state, lengths, rtol=None):
        if rtol is None:
            # bf16 steps run on this repo's kernels only (bit-repeatable: replays are expected to be IDENTICAL, the bar
            # leaves room for a rocBLAS split-K atomic or two); fp32 steps go through MIOpen's atomically accumulating
            # solvers, whose run-to-run noise the rectified-stereo border flips can turn into
            # several-fold changes of single tiny gradients -- there only gross garbage is caught
            rtol = 0.05 if opts.CONV_DTYPE in ("bf16", "fp16") else 8.0
        # absolute floor relative to the largest gradient of the whole model: a one-element bias whose gradient is a
        # nearly cancelling sum has no meaningful relative error under the library path's atomics
        floor = 1e-5 if opts.CONV_DTYPE in ("bf16", "fp16") else 1e-3
        for i, (a, b) in enumerate(zip(first, state)):
            # gradients and first moments only (indices 1, 2 of the optimizer state): Adam turns a rounding-noise
            # gradient into a +-lr step, so VALUES of parameters with a ~zero gradient legitimately differ between runs
            if a.dtype != torch.float32 or a.numel() < 2 or (len(first) > 3 and i not in (1, 2)):
                continue
            diff, mag = (a - b).abs().reshape(-1), a.abs().reshape(-1)
            if lengths is not None and int(lengths.sum()) == a.numel():      # per parameter of the flat buffers
                seg_diff = torch.segment_reduce(diff, "max", lengths=lengths)
                seg_mag = torch.segment_reduce(mag, "max", lengths=lengths)
                bad_seg = seg_diff > rtol * seg_mag + floor * seg_mag.max() + 1e-12
                if bool(bad_seg.any()):
                    bad = torch.repeat_interleave(bad_seg, lengths) & (diff > 0)
                    text = f"{what} in {int(bad_seg.sum())} parameters of state tensor {i}"
                    return text + (self.describe(i, bad) if self.describe is not None else "")
            elif bool((diff > rtol * mag.max() + 1e-12).any()):
                return f"{what} in state tensor {i}"
        return None


class _MetricsGraph:
    """merge_results (abs-rel with its two batched sorts, centre depths, pose errors: ~70 tiny launches) captured into a
    hipGraph of its own, one per captured training step: it reads the step graph's STATIC inputs and outputs, so a
    replay right behind the step's replay yields this step's record without ~70 eager launches (6 ms of host time per
    step at batch 8).  The record is one stacked tensor, cloned per step and fetched once per epoch."""

    def __init__(self, stereo):
        self.stereo = stereo
        self.cache = {}

    def __call__(self, static_in, preds, loss, loss_by_type):
        key = (loss.data_ptr(), static_in["image5d"].data_ptr())
        entry = self.cache.get(key)
        if entry is None:
            first = merge_results(static_in, preds, loss, loss_by_type, self.stereo)        # eager: allocations, values
            keys = list(first)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                merge_results(static_in, preds, loss, loss_by_type, self.stereo)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph(keep_graph=True)
            with torch.cuda.graph(graph):
                res = merge_results(static_in, preds, loss, loss_by_type, self.stereo)
                out = torch.stack([res[k].reshape(()).float() for k in keys])
            census = _ops.graph_census(graph)          # the same audit as the training step's graph (no memset nodes)
            if census["memset"]:
                # memset nodes do not survive replay on this stack.  The metrics are auxiliary and their
                # eager values are in hand: this signature computes them eagerly from now on (the step graph itself is held
                # to the hard rule)
                import sys
                print(f"[MetricsGraph] the captured per-step metrics contain memset nodes ({census}): computed eagerly for "
                      f"this input signature", file=sys.stderr, flush=True)
                del graph
                self.cache[key] = "eager"
                return first
            graph.instantiate()
            self.cache[key] = (graph, keys, out)
            return first
        if entry == "eager":
            return merge_results(static_in, preds, loss, loss_by_type, self.stereo)
        graph, keys, out = entry
        graph.replay()
        vals = out.clone()
        return {k: vals[i] for i, k in enumerate(keys)}


class ModelTrainerGraph(ModelTrainer):
    def __init__(self, model, loss_object, steps_per_epoch, stereo, augmenter, optimizer):
        super().__init__(model, loss_object, steps_per_epoch, stereo, augmenter, optimizer)
        self.set_name("Train (graph)")
        self._metrics = _MetricsGraph(stereo)
        # Steps that TRAIN PWC-Net (flowL2 / flow_reg) run MIOpen's backward solvers on its small pyramid levels.  Round 2 kept
        # them out of capture altogether (some solvers clear a workspace with a memset node, which does not survive replay on
        # this stack: DESIGN.md section 6).  Since round 3 they are captured like every other step and it is the graph AUDIT
        # (_StepGraph._capture: no memset node) plus the replay check that decide; opts.CAPTURE_LIBRARY_STEPS = False restores
        # the eager rule.  Measured at batch 8, 128x384: 15.6 ms captured (932 kernel nodes, no memset) against 17.3 ms eager.
        weights = getattr(loss_object, "loss_weights", None) or {}
        self.trains_flow_net = ("flownet" in getattr(model, "models", {}) and any(k.startswith("flow") for k in weights)
                                and getattr(opts, "CAPTURE_LIBRARY_STEPS", "audit") is False)
        # Early optimizer update (config.EARLY_UPDATE): the backward pass is cut between decoder and encoder; the finishing
        # launch of the first phase's deferred gradients (decoder, heads, PoseNet: ~60 % of the partial bytes) and the Adam
        # update of those parameters run on the side stream while the encoder's backward occupies the main one -- 0.1 ms of
        # memory streaming off the critical chain of the captured step.  Element-wise update, disjoint buffer pieces, the
        # same partial sums in the same order: bit-identical to the one-piece step (tests/test_graph_replay.py).
        self._early_start = None
        if (getattr(opts, "EARLY_UPDATE", False) and not self.trains_flow_net and getattr(optimizer, "flat", None) is not None
                and optimizer.flat.data.is_cuda and not getattr(optimizer, "l2_terms", None) and type(model).__name__ == "ModelWrapper"):
            self._early_start = self._find_early_bucket()
            if self._early_start is not None and self._early_start % 4:
                self.model.set_backward_cut(False)
                self._early_start = None
        self._update_stream = None
        self._graph = _StepGraph(self.train_a_step, state=self.optimizer_state, describe=self.describe_state,
                                 segments=self.state_segments, repair=self.repair_flagged,
                                 reference=self.augmenter is None, pin=self._pin_hooks())

    def train_a_step(self, features):
        if self._early_start is None or not features["image5d"].is_cuda:
            if self._early_start is not None:          # (host tensors: the plain step, no cut)
                return self._plain_step_with_cut(features)
            return super().train_a_step(features)
        opt, scale = self.optimizer, self.grad_unscale()
        carry, out = self.backward_first(features, finish=False)
        opt.flat.gather_grads()                        # (gradients autograd left on parameters: copied on their own stream)
        opt.begin_step()
        main = torch.cuda.current_stream()
        if self._update_stream is None:
            streams = getattr(self.model, "_side_streams", None)
            self._update_stream = streams[0] if streams else torch.cuda.Stream()
        side = self._update_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _ops.grad_sink.flush()
            opt.apply_gradients(grad_scale=scale, lo=self._early_start, hi=None, bump=False)
        self.backward_second(carry)
        opt.apply_gradients(grad_scale=scale, lo=0, hi=self._early_start, bump=False)
        main.wait_stream(side)
        return out

    def _plain_step_with_cut(self, features):
        carry, out = self.backward_first(features)
        self.backward_second(carry)
        self.optimizer.apply_gradients(grad_scale=self.grad_unscale())
        return out

    def run_a_batch(self, features):
        if not features["image5d"].is_cuda or self.trains_flow_net:
            return self.train_a_step(features)
        return self._graph(features)

    def step_metrics(self, features, preds, loss, loss_by_type):
        graph = self._graph
        if not features["image5d"].is_cuda or self.trains_flow_net or graph.eager_fallback or graph.static_out is None \
                or loss is not graph.static_out[1]:
            return merge_results(features, preds, loss, loss_by_type, self.stereo)
        return self._metrics(graph.static_in, preds, loss, loss_by_type)


class ModelTrainerDistrib(ModelTrainer):
    """One process per GPU.  forward+backward is a hipGraph; the gradient all-reduce (RCCL) and the fused Adam run
    behind it on the same stream."""

    def __init__(self, model, loss_object, steps_per_epoch, stereo, augmenter, optimizer):
        super().__init__(model, loss_object, steps_per_epoch, stereo, augmenter, optimizer)
        self.set_name("Train (distributed)")
        self.strategy = DistributionStrategy.get_strategy()
        if self.strategy is not None:
            flat = self.optimizer.flat
            self.strategy.broadcast_parameters(flat.data)
            if flat.shadow is not None:          # the bf16 copies the kernels read follow the broadcast weights at once
                flat.shadow.copy_(flat.data)     # (otherwise only the first Adam step would refresh them)
        weights = getattr(loss_object, "loss_weights", None) or {}           # see ModelTrainerGraph
        self.trains_flow_net = "flownet" in getattr(model, "models", {}) and any(k.startswith("flow") for k in weights)
        # Overlap of the gradient exchange with the backward pass: the backward is cut between decoder and encoder; the
        # gradients of everything behind the encoder (decoder, PoseNet: 60% of the bytes) are complete after the first
        # phase and their all-reduce runs on RCCL's stream while the encoder's backward (the second phase) computes.
        self._early_start = self._find_early_bucket() if self._overlap_wanted() else None
        self._early_work = None
        phases = (self.backward_first, self.backward_second) if self._early_start is not None else None
        self._graph = _StepGraph(self.forward_backward, state=self.optimizer_state, describe=self.describe_state,
                                 segments=self.state_segments, repair=self.repair_flagged,
                                 reference=self.augmenter is None, pin=self._pin_hooks(), phases=phases,
                                 between=self.reduce_early if phases else None) \
            if getattr(opts, "DISTRIB_GRAPH", True) else None

    def _overlap_wanted(self):
        import os
        forced = os.environ.get("XPT_DP_OVERLAP")
        if forced is not None:
            return forced == "1"
        return self.strategy is not None and self.strategy.num_replicas_in_sync > 1 and not self.trains_flow_net

    def forward_backward(self, features):
        if self._early_start is None:
            return super().forward_backward(features)
        carry, out = self.backward_first(features)
        self.backward_second(carry)
        return out

    def reduce_early(self):
        """Between the two phases: start the all-reduce of the gradients the first phase completed."""
        if self.strategy is not None and self._early_start is not None:
            grad = self.optimizer.flat.grad
            self._early_work = self.strategy.all_reduce_range(grad, self._early_start, grad.numel(), async_op=True)

    def reduce_gradients(self):
        if self.strategy is None:
            return
        grad = self.optimizer.flat.grad
        if self._early_work is not None:
            rest = self.strategy.all_reduce_range(grad, 0, self._early_start, async_op=True)
            self._early_work.wait()
            self._early_work = None
            if rest is not None:
                rest.wait()
        else:
            self.strategy.all_reduce_gradients(grad)

    def run_a_batch(self, features):
        if self._graph is not None and features["image5d"].is_cuda and not self.trains_flow_net:
            out = self._graph(features)
        elif self._early_start is not None:
            carry, out = self.backward_first(features)
            self.reduce_early()
            self.backward_second(carry)
        else:
            out = self.forward_backward(features)
        self.reduce_gradients()
        self.optimizer.apply_gradients(grad_scale=self.grad_unscale())
        return out


class ModelValidater(TrainValBase):
    def __init__(self, model, loss_object, steps_per_epoch, stereo):
        super().__init__(model, loss_object, steps_per_epoch, stereo)
        self.set_name("Validate (eager)")

    def run_a_batch(self, features):
        return self.validate_a_step(features)

    def validate_a_step(self, features):
        """train_val.py:127-130."""
        with torch.no_grad():
            preds = self.model(features)
            total_loss, loss_by_type = self.loss_object(preds, features)
        return preds, total_loss, loss_by_type


class ModelValidaterGraph(ModelValidater):
    def __init__(self, model, loss_object, steps_per_epoch, stereo):
        super().__init__(model, loss_object, steps_per_epoch, stereo)
        self.set_name("Validate (graph)")
        self._graph = _StepGraph(self.validate_a_step)

    def run_a_batch(self, features):
        if not features["image5d"].is_cuda:
            return self.validate_a_step(features)
        return self._graph(features)


class ModelValidaterDistrib(ModelValidaterGraph):
    def __init__(self, model, loss_object, steps_per_epoch, stereo):
        super().__init__(model, loss_object, steps_per_epoch, stereo)
        self.set_name("Validate (distributed)")


# ---------------------------------------------------------------------------------------------- per-step results
def merge_results(features, preds, loss, loss_by_type, stereo):
    """train_val.py:157-177, kept on the device: {"loss", "deprel", "gtdepth", "prdepth", <per-type losses>} as
    0-dim tensors (cloned, so hipGraph replays do not overwrite them); fetch_results() moves a whole epoch to the
    host at once."""
    batch_result = {"loss": loss.detach().clone()}
    if "pose" in preds:
        trjabs, trjrel, roterr = get_pose_metric(preds, features)
        batch_result["trjabs"], batch_result["trjrel"], batch_result["roterr"] = trjabs, trjrel, roterr
    if "depth_ms" in preds and "depth_gt" in features:
        batch_result["deprel"] = get_depth_metric(features, preds)
        gtdepth, prdepth = get_center_depths(features, preds)
        batch_result["gtdepth"] = gtdepth[0]
        batch_result["prdepth"] = prdepth[0]
    batch_result.update({key: val.detach().clone() for key, val in loss_by_type.items()})
    return batch_result


def fetch_results(results):
    if not results:
        return []
    keys = list(results[0].keys())
    stacked = torch.stack([torch.stack([torch.as_tensor(r[k], dtype=torch.float32, device=results[0]["loss"].device)
                                        for k in keys]) for r in results]).cpu().numpy()
    return [dict(zip(keys, row)) for row in stacked]


def get_depth_metric(features, preds):
    """train_val.py:180-200 + evaluate/eval_utils.py:109-131 (valid_depth_filter: 1e-3 < gt < 80, Garg crop,
    median scaling, clip) -> mean abs-rel over the batch.  On the GPU ONE launch for the whole batch
    (csrc/xpt_metric.hip: exact medians by radix selection; the library formulation below needs two full sorts, ~110
    launches and 0.5 ms per step); host tensors take the batched torch formulation."""
    depth_pred = preds["depth_ms"][0].detach()[..., 0].float()
    depth_true = features["depth_gt"][..., 0]
    B, h, w = depth_true.shape
    crop = (np.array([0.40810811 * h, 0.99189189 * h, 0.03594771 * w, 0.96405229 * w])).astype(np.int32)
    if depth_true.is_cuda:
        from ..hip import lib as _hip_lib
        from ..hip import ops as _hip_ops
        lib = _hip_lib.load()
        pred_c, true_c = depth_pred.contiguous(), depth_true.float().contiguous()
        per_sample = torch.empty(B, dtype=torch.float32, device=depth_true.device)
        _hip_lib.check(lib.xpt_depth_metric(pred_c.data_ptr(), true_c.data_ptr(), per_sample.data_ptr(), B, h, w,
                                            int(crop[0]), int(crop[1]), int(crop[2]), int(crop[3]), float(opts.MIN_DEPTH),
                                            float(opts.MAX_DEPTH), _hip_ops._stream()), "xpt_depth_metric")
        return per_sample.mean()
    return depth_metric_batched(depth_pred, depth_true, crop)


def depth_metric_batched(depth_pred, depth_true, crop):
    """The same metric with torch ops on the whole batch (two batched sorts): host tensors, and the cross-check of the
    kernel in tests/test_inloop_metrics_gpu.py."""
    B, h, w = depth_true.shape
    crop_mask = torch.zeros((h, w), dtype=torch.bool, device=depth_true.device)
    crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = True
    gt, pr = depth_true.reshape(B, -1), depth_pred.reshape(B, -1)
    mask = (gt > opts.MIN_DEPTH) & (gt < opts.MAX_DEPTH) & crop_mask.reshape(1, -1)
    cnt = mask.sum(dim=1)
    inf = torch.full_like(gt, float("inf"))
    gt_sorted = torch.sort(torch.where(mask, gt, inf), dim=1).values
    pr_sorted = torch.sort(torch.where(mask, pr, inf), dim=1).values
    k_lo = torch.clamp((cnt - 1) // 2, min=0).unsqueeze(1)                  # np.median: mean of the two middle values
    k_hi = torch.clamp(cnt // 2, min=0, max=gt.shape[1] - 1).unsqueeze(1)
    med_gt = 0.5 * (gt_sorted.gather(1, k_lo) + gt_sorted.gather(1, k_hi))
    med_pr = 0.5 * (pr_sorted.gather(1, k_lo) + pr_sorted.gather(1, k_hi))
    scaled = torch.clamp(pr * (med_gt / med_pr), opts.MIN_DEPTH, opts.MAX_DEPTH)
    err = torch.where(mask, torch.abs(gt - scaled) / torch.where(mask, gt, torch.ones_like(gt)), torch.zeros_like(gt))
    per_sample = torch.where(cnt > 0, err.sum(dim=1) / cnt.clamp(min=1), torch.zeros_like(med_gt[:, 0]))
    return per_sample.mean()


def get_pose_metric(preds, features):
    """train_val.py:203-210 + evaluate/eval_utils.py:15-87 (PoseMetricNumpy) on the device: mean absolute-scale trajectory
    error, mean scale-aligned trajectory error (metres) and mean rotational error (radians) of the snippet re-based on
    its first frame.  Same arithmetic as evaluate/eval_utils.PoseMetricNumpy (the host version the reference's
    known-answer tests pin); zeros when the dataset has no pose_gt (:209-210)."""
    pose = preds["pose"].detach().float()
    zero = torch.zeros((), device=pose.device)
    if "pose_gt" not in features:
        return zero, zero, zero
    from ..utils import convert_pose as cp
    pred = cp.pose_rvec2matr_batch_tf(pose)                # [B, N, 4, 4] (utils/convert_pose.py:32-71; the gfx950 kernel)
    true = features["pose_gt"].float()
    if pred.is_cuda:                                       # one launch for the three means (csrc/xpt_metric.hip)
        from ..hip import lib as _hip_lib
        from ..hip import ops as _hip_ops
        out = torch.empty(3, dtype=torch.float32, device=pred.device)
        pred_c, true_c = pred.contiguous(), true.contiguous()
        _hip_lib.check(_hip_lib.load().xpt_pose_metric(pred_c.data_ptr(), true_c.data_ptr(), out.data_ptr(), pred.shape[0],
                                                       pred.shape[1], _hip_ops._stream()), "xpt_pose_metric")
        return out[0], out[1], out[2]

    def from_first(poses):                       # [B, N, 4, 4] -> [B, N + 1, 4, 4], target (identity) in the middle
        eye = torch.eye(4, device=poses.device).expand(poses.shape[0], 1, 4, 4)
        mats = torch.cat([poses[:, :2], eye, poses[:, 2:]], dim=1)
        return cp.rigid_inverse(mats[:, 0:1]) @ mats

    pred, true = from_first(pred), from_first(true)
    xyz_p, xyz_t = pred[:, :, :3, 3], true[:, :, :3, 3]
    abs_err = (xyz_t - xyz_p).norm(dim=2)[:, 1:]
    scale = (xyz_t * xyz_p).sum(dim=2) / (xyz_p ** 2).sum(dim=2)            # 0 / 0 only at the dropped origin frame
    rel_err = (xyz_t - xyz_p * scale[..., None]).norm(dim=2)[:, 1:]
    rel = pred[:, :, :3, :3].transpose(-1, -2) @ true[:, :, :3, :3]
    cosine = ((rel.diagonal(dim1=-2, dim2=-1).sum(-1) - 1.0) / 2.0).clamp(-1.0, 1.0)
    rot_err = torch.acos(cosine)[:, 1:]
    return abs_err.mean(), rel_err.mean(), rot_err.mean()


def get_center_depths(features, preds):
    """train_val.py:213-236: mean true (positive only) / predicted depth in a 20x20 window at 3/4 height."""
    depth_pred = preds["depth_ms"][0].detach().float()
    depth_true = features["depth_gt"]
    _, height, width, _ = depth_pred.shape
    xs, xe = width // 2 - 10, width // 2 + 10
    ys, ye = height // 4 * 3 - 10, height // 4 * 3 + 10
    win = depth_true[:, ys:ye, xs:xe, :]
    pos = (win > 0).to(win.dtype)
    mean_true = (win * pos).sum(dim=(1, 2, 3)) / pos.sum(dim=(1, 2, 3)).clamp(min=1)
    mean_pred = depth_pred[:, ys:ye, xs:xe, :].mean(dim=(1, 2, 3))
    return mean_true, mean_pred
