"""ModelWrapper / StereoModelWrapper / StereoPoseModelWrapper with the reference's interface
(model/build_model/model_wrappers.py:10-177) over torch modules on MI355X."""
import os.path as op

import numpy as np
import torch

from ...config import opts
from ...utils import util_funcs as uf


def _structural_pads(model):
    """{id(tensor): (out_sel, in_sel)} over every sub-module that declares structurally-zero entries."""
    pads = {}
    for m in model.modules():
        fn = getattr(m, "structural_pads", None)
        if callable(fn):
            for t, o, i in fn():
                pads[id(t)] = (o, i)
    return pads


def logical_state_dict(model):
    """state_dict() with every structurally padded tensor cut down to its logical entries: the on-disk shapes are the
    reference's and do not depend on XPT_STEM1_FILTERS / XPT_STEM2_FILTERS."""
    from .pretrained_nets import logical_view
    pads = _structural_pads(model)
    return {k: logical_view(v.detach(), pads.get(id(v))).clone() for k, v in model.state_dict(keep_vars=True).items()}


def load_logical_state_dict(model, state):
    """Strict load of a logical_state_dict() file into a model of ANY physical padding; files of earlier rounds that hold
    the physical (padded) shapes are accepted too.  The structural zeros are re-applied afterwards, so a file with non-zero
    padded entries cannot break the 'same function' invariant."""
    from .pretrained_nets import _store_logical, logical_view
    pads = _structural_pads(model)
    own = model.state_dict(keep_vars=True)
    missing, unexpected = sorted(set(own) - set(state)), sorted(set(state) - set(own))
    if missing or unexpected:
        raise RuntimeError(f"checkpoint does not match the model: missing {missing[:4]} ({len(missing)}), "
                           f"unexpected {unexpected[:4]} ({len(unexpected)})")
    with torch.no_grad():
        for k, t in own.items():
            v, sel = state[k], pads.get(id(t))
            if sel is not None and tuple(v.shape) == tuple(logical_view(t, sel).shape) and tuple(v.shape) != tuple(t.shape):
                _store_logical(t, sel, v, fill=1.0 if k.endswith("running_var") else 0.0)
            elif tuple(v.shape) == tuple(t.shape):
                t.copy_(v.to(device=t.device, dtype=t.dtype))
            else:
                raise RuntimeError(f"checkpoint tensor {k}: shape {tuple(v.shape)}, the model expects {tuple(t.shape)}"
                                   + (f" or its logical {tuple(logical_view(t, sel).shape)}" if sel is not None else ""))
        for m in model.modules():
            fn = getattr(m, "apply_structural_zeros", None)
            if callable(fn):
                fn()


class ModelWrapper:
    def __init__(self, models):
        self.models = models                      # {"depthnet": nn.Module, "posenet": nn.Module}
        # (BASELINE configs[4] names fp16 convolutions: this build computes them in bf16 -- same matrix-core rate on gfx950,
        #  fp32 accumulation, no loss scaling; DESIGN.md section 7 -- so "fp16" is not an accepted value)
        if opts.CONV_DTYPE in ("bf16", "fp16"):
            # the 16-bit format is a property of the loaded library (hip/lib.py): one per process
            from ...hip import lib as _lib
            _lib.set_half_format(opts.CONV_DTYPE)
        self.conv_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[opts.CONV_DTYPE]

    def __call__(self, features):
        return self.predict_batch(features)

    # ------------------------------------------------------------------ device / format plumbing
    def to(self, device, channels_last=None):
        channels_last = opts.CHANNELS_LAST if channels_last is None else channels_last
        for model in self.models.values():
            model.to(device)
            if channels_last:
                model.to(memory_format=torch.channels_last)
        return self

    def _run(self, model, image5d):
        if self.conv_dtype is None or not image5d.is_cuda:
            return model(image5d)
        with torch.autocast(device_type="cuda", dtype=self.conv_dtype):
            return model(image5d)

    # ------------------------------------------------------------------ model_wrappers.py:41-51
    def predict_batch(self, features, suffix=""):
        predictions = dict()
        image5d = features["image5d" + suffix]
        if image5d.is_cuda and suffix == "":
            # one launch refreshes the bf16 operand copies of every dense convolution weight from the fp32 masters
            # (first thing of every step: captured with it, so a replayed step sees the weights Adam just wrote)
            from ...hip import conv as _conv
            _conv.packer.pack()
        nets = list(self.models.values())
        if image5d.is_cuda and len(nets) > 1 and getattr(opts, "NET_STREAMS", False) and type(self) is ModelWrapper:
            # DepthNet and PoseNet are independent until the loss: the small PoseNet runs on a side HIP stream next
            # to the (launch-latency-bound) encoder, forward and -- because autograd replays every node on the stream
            # of its forward -- backward.  Inside a hipGraph capture this forks / joins the captured graph.
            main = torch.cuda.current_stream()
            if not hasattr(self, "_side_streams"):
                self._side_streams = [torch.cuda.Stream() for _ in nets[1:]]
            side_out = []
            from ...hip import ops as _ops
            for model, side in zip(nets[1:], self._side_streams):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    extra = {}
                    if (not side_out and image5d.dtype == torch.float32 and image5d.is_contiguous() and image5d.dim() == 5
                            and image5d.shape[-1] == 3 and image5d.shape[1] >= 2
                            and image5d.shape[2] % 8 == 0 and image5d.shape[3] % 8 == 0):
                        # the loss's image pyramids (dense source / target copies at 1, 1/2, 1/4, 1/8: losses.append_data) depend
                        # on the snippets alone: issued here, next to the encoder, instead of between the nets and the march
                        extra["image_pyramids"] = ((1, 2, 4, 8), *_ops.image_pyramids(image5d, (1, 2, 4, 8)))
                    out = self._run(model, image5d)
                    out.update(extra)
                    side_out.append(out)
                if torch.is_grad_enabled():
                    # the deferred parameter-gradient partials of this net are written on `side` during backward and are no
                    # autograd outputs: the sink's finishing launch (main stream, after backward) waits for the stream itself
                    _ops.grad_sink.join_streams.add(side)
            predictions.update(self._run(nets[0], image5d))
            for out, side in zip(side_out, self._side_streams):
                main.wait_stream(side)
                for value in out.values():
                    for t in (value if isinstance(value, (list, tuple)) else [value]):
                        for u in (t.values() if isinstance(t, dict) else [t]):
                            if torch.is_tensor(u):
                                u.record_stream(main)
                predictions.update(out)
        else:
            for model in nets:
                predictions.update(self._run(model, image5d))
        if "depth_ms" in predictions and "disp_ms" not in predictions:
            predictions["disp_ms"] = uf.safe_reciprocal_number_ms(predictions["depth_ms"])
        return {key + suffix: value for key, value in predictions.items()}

    def predict_dataset(self, dataset, save_keys, total_steps):
        """model_wrappers.py:18-31."""
        outputs = self.init_output_structure(save_keys)
        with torch.no_grad():
            for step, features in enumerate(dataset):
                predictions = self.predict_batch(features)
                outputs = self.append_outputs(features, predictions, outputs)
                uf.print_progress_status(f"Progress: {step} / {total_steps}")
        print("")
        return {key: np.concatenate(data, axis=0) for key, data in outputs.items() if data}

    def init_output_structure(self, keys):
        outputs = {"image": []}
        outputs.update({key: [] for key in keys})
        outputs.update({key + "_gt": [] for key in keys})
        if "depth" in keys:
            outputs["intrinsic"] = []
        return outputs

    def append_outputs(self, features, predictions, outputs, suffix=""):
        """model_wrappers.py:53-79: target image (uint8), pose(+gt), full-resolution depth(+gt, intrinsic)."""
        image = features["image5d" + suffix][:, -1]
        outputs["image" + suffix].append(uf.to_uint8_image(image).cpu().numpy())
        if "pose" + suffix in outputs:
            outputs["pose_gt" + suffix].append(features["pose_gt" + suffix].cpu().numpy())
            outputs["pose" + suffix].append(predictions["pose" + suffix].float().cpu().numpy())
        if "depth" + suffix in outputs:
            outputs["depth_gt" + suffix].append(features["depth_gt" + suffix].cpu().numpy())
            outputs["depth" + suffix].append(predictions["depth_ms" + suffix][0].float().cpu().numpy())
            outputs["intrinsic" + suffix].append(features["intrinsic" + suffix].cpu().numpy())
        if "flow" + suffix in outputs:                                       # model_wrappers.py:72-78 (no flow_gt yet)
            outputs["flow" + suffix].append(predictions["flow_ms" + suffix][0].float().cpu().numpy())
        return outputs

    def set_trainable(self, name, trainable):
        for p in self.models[name].parameters():
            p.requires_grad_(trainable)
        print(f"[ModelWrapper] set {name} trainable {trainable}")

    def trainable_weights(self):
        """model_wrappers.py:89-93."""
        return [p for model in self.models.values() for p in model.parameters() if p.requires_grad]

    def weight_groups(self):
        """Same-shape weights that a layer consumes as one strided batch (modules expose `stack_groups()`): the flat
        parameter buffers place them next to each other (optimizers.FlatParameters)."""
        groups = []
        for model in self.models.values():
            for module in model.modules():
                if hasattr(module, "stack_groups"):
                    groups.extend(module.stack_groups())
        return groups

    # ------------------------------------------------------------------ two-phase backward (data-parallel overlap)
    def set_backward_cut(self, enabled):
        """Cut every depth network's backward between decoder and encoder (depth_net.DepthNetPretrained.forward).
        Returns the parameters whose gradients are only finished by the second phase (the encoders')."""
        late = []
        for model in self.models.values():
            if hasattr(model, "cut_backward") and hasattr(model, "encoder"):
                model.cut_backward = bool(enabled)
                model.cuts = []
                late.extend(p for p in model.encoder.parameters() if p.requires_grad)
        return late

    def take_backward_cuts(self):
        """[(tensor, gradient)] to resume the backward pass from, after the first phase; clears the record."""
        pairs = []
        for model in self.models.values():
            for taps, leaves in getattr(model, "cuts", []):
                pairs.extend((t, leaf.grad) for t, leaf in zip(taps, leaves) if leaf.grad is not None)
            if hasattr(model, "cuts"):
                model.cuts = []
        return pairs

    def weights_to_regularize(self):
        """model_wrappers.py:95-99: the FlowNet's trainable weights (for the flow_reg L2 term), else None."""
        if "flownet" in self.models:
            return [p for p in self.models["flownet"].parameters() if p.requires_grad]
        return None

    def save_weights(self, ckpt_dir_path, suffix):
        """model_wrappers.py:101-105 ({netname}_{suffix}; torch state_dict instead of Keras H5).  The file holds the
        LOGICAL tensors -- the reference's shapes (11 / 22-filter stem cells, 87 input channels of up2.conv2) -- whatever
        physical padding this build runs with (logical_state_dict)."""
        for netname, model in self.models.items():
            save_path = op.join(ckpt_dir_path, f"{netname}_{suffix}.pt")
            torch.save(logical_state_dict(model), save_path)
            print(f"===== {netname} weights are saved to", save_path)

    def load_weights(self, ckpt_dir_path, suffix):
        """model_wrappers.py:107-117."""
        for netname, model in self.models.items():
            ckpt_file = op.join(ckpt_dir_path, f"{netname}_{suffix}.pt")
            if op.isfile(ckpt_file):
                load_logical_state_dict(model, torch.load(ckpt_file, map_location="cpu"))
                print(f"===== {netname} weights loaded from", ckpt_file)
                print(f"      {netname} num params:", sum(p.numel() for p in model.parameters()))
            else:
                print(f"===== Failed to load weights of {netname}, train from scratch ...")
                print("      tried to load file:", ckpt_file)

    def summary(self, **kwargs):
        for netname, model in self.models.items():
            n = sum(p.numel() for p in model.parameters())
            print(f"{netname}: {n / 1e6:.2f} M parameters")


class StereoModelWrapper(ModelWrapper):
    def __call__(self, features):
        """model_wrappers.py:141-145."""
        predictions = self.predict_batch(features)
        predictions.update(self.predict_batch(features, "_R"))
        return predictions


class StereoPoseModelWrapper(StereoModelWrapper):
    def __call__(self, features):
        """model_wrappers.py:152-159."""
        predictions = self.predict_batch(features)
        predictions.update(self.predict_batch(features, "_R"))
        if "posenet" in self.models:
            predictions.update(self.predict_stereo_pose(features))
        return predictions

    def predict_stereo_pose(self, features):
        """model_wrappers.py:161-177: PoseNet on [R,R,R,R,L] -> pose_LR and [L,L,L,L,R] -> pose_RL."""
        posenet = self.models["posenet"]
        left_target = features["image5d"][:, -1]
        right_target = features["image5d_R"][:, -1]
        numsrc = opts.SNIPPET_LEN - 1
        lr_input = torch.stack([right_target] * numsrc + [left_target], dim=1)
        rl_input = torch.stack([left_target] * numsrc + [right_target], dim=1)
        pose_lr = self._run(posenet, lr_input)
        pose_rl = self._run(posenet, rl_input)
        return {"pose_LR": pose_lr["pose"], "pose_RL": pose_rl["pose"]}
