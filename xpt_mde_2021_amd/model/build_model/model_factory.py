"""ModelFactory with the reference's interface (model/build_model/model_factory.py:18-145)."""
import torch

from ...config import opts
from ...utils import util_funcs as uf
from ...utils.util_class import WrongInputException
from ..model_util import layer_ops as lo
from . import depth_net as dn
from . import flow_net as fn
from . import model_wrappers as mw
from . import pose_net as pn

PRETRAINED_MODELS = ["MobileNetV2", "NASNetMobile", "DenseNet121", "VGG16", "Xception", "ResNet50V2", "NASNetLarge",
                     "EfficientNetB0", "EfficientNetB3", "EfficientNetB5", "EfficientNetB7"]


class ModelFactory:
    def __init__(self, dataset_cfg, global_batch=None, net_names=None, depth_activation=None, pretrained_weight=None,
                 stereo=None, high_res=None):
        self.global_batch = opts.BATCH_SIZE if global_batch is None else global_batch
        self.dataset_cfg = dataset_cfg
        self.bshwc_shape = [self.global_batch] + list(dataset_cfg["imshape"])
        self.net_names = opts.JOINT_NET if net_names is None else net_names
        self.activation = opts.DEPTH_ACTIVATION if depth_activation is None else depth_activation
        self.pretrained_weight = opts.PRETRAINED_WEIGHT if pretrained_weight is None else pretrained_weight
        self.stereo = opts.STEREO if stereo is None else stereo
        self.high_res = opts.HIGH_RES if high_res is None else high_res
        print("[ModelFactory] net names:", self.net_names)

    def get_model(self):
        """model_factory.py:36-64."""
        models = dict()
        if "depth" in self.net_names:
            conv_depth = self.conv2d_factory(opts.DEPTH_CONV_ARGS)
            models["depthnet"] = self.depth_net_factory(self.net_names["depth"], conv_depth,
                                                        self.activation_factory(self.activation),
                                                        opts.DEPTH_UPSAMPLE_INTERP)
        if "camera" in self.net_names:
            models["posenet"] = self.pose_net_factory(self.net_names["camera"], self.conv2d_factory(opts.POSE_CONV_ARGS))
        if "flow" in self.net_names and self.net_names.get("flow"):
            models["flownet"] = self.flow_net_factory(self.net_names["flow"], self.conv2d_factory(opts.FLOW_CONV_ARGS))

        if ("stereo_T_LR" in self.dataset_cfg) and ("depth" in self.net_names):
            return mw.StereoPoseModelWrapper(models)
        if ("image_R" in self.dataset_cfg) and self.stereo:
            return mw.StereoModelWrapper(models)
        return mw.ModelWrapper(models)

    def activation_factory(self, activ_name):
        if activ_name == "InverseSigmoid":
            return InverseSigmoidActivation()
        if activ_name == "Exponential":
            return ExponentialActivation()
        raise WrongInputException("[activation_factory] wrong activation name: " + activ_name)

    def conv2d_factory(self, src_args):
        """model_factory.py:74-98: string options -> convolution factory."""
        args = {}
        if "activation" in src_args:
            args["activation"] = "leaky_relu" if src_args["activation"] == "leaky_relu" else "relu"
            args["activation_param"] = src_args.get("activation_param")
        if "kernel_initializer" in src_args:
            tn = src_args["kernel_initializer"] == "truncated_normal"
            args["kernel_initializer"] = "truncated_normal" if tn else "glorot_uniform"
            args["kernel_initializer_param"] = src_args.get("kernel_initializer_param")
        return lo.CustomConv2D(**args)

    def depth_net_factory(self, net_name, conv2d_d, pred_activ, upsample_interp):
        if net_name in PRETRAINED_MODELS:
            return dn.DepthNetPretrained(self.bshwc_shape, self.global_batch, conv2d_d, pred_activ, upsample_interp,
                                         net_name, self.pretrained_weight, self.high_res)
        raise WrongInputException("[depth_net_factory] depth net outside this build's hot path: " + net_name)

    def pose_net_factory(self, net_name, conv2d_p):
        if net_name == "PoseNetBasic":
            return pn.PoseNetBasic(self.bshwc_shape, self.global_batch, conv2d_p, self.high_res)
        if net_name == "PoseNetImproved":
            return pn.PoseNetImproved(self.bshwc_shape, self.global_batch, conv2d_p, self.high_res)
        raise WrongInputException("[pose_net_factory] pose net outside this build's hot path: " + net_name)

    def flow_net_factory(self, net_name, conv2d_f):
        """model_factory.py:126-131."""
        if net_name == "PWCNet":
            return fn.PWCNet(self.bshwc_shape, self.global_batch, conv2d_f)
        raise WrongInputException("[flow_net_factory] wrong flow net name: " + net_name)


class InverseSigmoidActivation:
    """model_factory.py:134-138: depth = 1 / (sigmoid(x) + 0.01)  in (0.99, 100)."""

    def __call__(self, x):
        return uf.safe_reciprocal_number(torch.sigmoid(x) + 0.01)

    def with_disparity(self, x):
        """(depth, disp = safe_reciprocal_number(depth)): one gfx950 launch on the GPU instead of ~10 elementwise ones."""
        if x.is_cuda and x.dtype == torch.float32:
            from ...hip import ops as _ops
            return _ops.inverse_sigmoid_depth(x)
        depth = self(x)
        return depth, uf.safe_reciprocal_number(depth)

    def with_disparity_multi(self, xs):
        """with_disparity for every prediction scale of the decoder at once: ([depth_s], [disp_s])."""
        if len(xs) <= 4 and all(x.is_cuda and x.dtype == torch.float32 for x in xs):
            from ...hip import ops as _ops
            return _ops.inverse_sigmoid_depth_multi(list(xs))
        pairs = [self.with_disparity(x) for x in xs]
        return [p[0] for p in pairs], [p[1] for p in pairs]


class ExponentialActivation:
    """model_factory.py:141-145: depth = exp(sigmoid(x + 1) * 10 - 5)."""

    def __call__(self, x):
        return torch.exp(torch.sigmoid(x + 1.) * 10. - 5.)
