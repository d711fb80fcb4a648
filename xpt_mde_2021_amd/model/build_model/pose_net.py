"""PoseNetImproved / PoseNetBasic (reference: model/build_model/pose_net.py:8-91) as torch modules."""
import torch
import torch.nn as nn
from ...hip.lib import half as _half      # torch dtype of the 16-bit activations (bf16 | fp16 build of the library)



def restack_on_channels(image5d):
    """pose_net.py:44-50: [batch, snippet, H, W, C] -> frames stacked on channels.  The reference builds
    [batch, H, W, snippet*C] (channel index = frame*C + c); here the same channel order in NCHW."""
    b, s, h, w, c = image5d.shape
    return image5d.permute(0, 1, 4, 2, 3).reshape(b, s * c, h, w)


class PoseNetBasic(nn.Module):
    """pose_net.py:8-50 (sfmlearner-style)."""
    SPEC = [(16, 7, 2), (32, 5, 2), (64, 3, 2), (128, 3, 2), (256, 3, 2), (256, 3, 2), (256, 3, 2)]

    def __init__(self, input_shape, global_batch, conv2d, high_res):
        super().__init__()
        self.input_shape = tuple(input_shape)
        self.high_res = high_res
        _, snippet, _, _, channel = self.input_shape
        self.numsrc = snippet - 1
        cin = snippet * channel
        layers = []
        for filters, k, s in self.SPEC:
            layers.append(conv2d(cin, filters, k, strides=s))
            cin = filters
        cin, extra = self.extra_layers(conv2d, cin)
        self.convs = nn.Sequential(*layers, *extra)
        self.head = conv2d(cin, self.numsrc * 6, 1, activation="linear")

    def extra_layers(self, conv2d, cin):
        return cin, []

    def forward(self, image5d):
        if image5d.is_cuda and image5d.dtype == torch.float32 and torch.is_autocast_enabled() \
                and torch.get_autocast_dtype("cuda") == _half():
            from ...hip import conv as _conv           # one launch: restack + cast + zero pad channel(s) for the MFMA conv
            x = _conv.restack_bf16(image5d, _conv.round_up(image5d.shape[1] * image5d.shape[4], 8))
        else:
            x = restack_on_channels(image5d)
        x = self.head(self.convs(x))
        if x.is_cuda and x.dtype in (torch.float32, _half()):
            from ...hip import ops as _ops
            poses = _ops.global_avg_pool(x)                    # GlobalAveragePooling2D (cast included), one launch
        else:
            poses = x.float().mean(dim=(2, 3))                 # GlobalAveragePooling2D
        return {"pose": poses.reshape(-1, self.numsrc, 6)}


class PoseNetImproved(PoseNetBasic):
    """pose_net.py:53-91: (32,5,2) (32,5,2) (64,3,2) (128,3,2) (256,3,2) (256,3,2) (256,3,1) (256,3,1)
    [+ (512,3,2) (512,3,1) (512,3,1) when high_res] -> 1x1 conv to numsrc*6 -> global average."""
    SPEC = [(32, 5, 2), (32, 5, 2), (64, 3, 2), (128, 3, 2), (256, 3, 2), (256, 3, 2), (256, 3, 1), (256, 3, 1)]

    def extra_layers(self, conv2d, cin):
        if not self.high_res:
            return cin, []
        return 512, [conv2d(cin, 512, 3, strides=2), conv2d(512, 512, 3), conv2d(512, 512, 3)]
