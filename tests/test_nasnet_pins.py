"""Structural pins of the NASNet-A-Mobile restatement (model/build_model/pretrained_nets.py): the arithmetic of
tf.keras.applications.NASNetMobile (tensorflow==2.4.1, not part of the reference checkout) is parity-unpinned, but these
facts are fixed by Keras' published architecture and by the reference's own data file:

* Keras reports 4,269,716 parameters for NASNetMobile(include_top=False): 4,232,978 trainable + 36,738 non-trainable
  (BatchNorm moving statistics);
* model/build_model/scaled_layers.json "NASNetMobile" (read by pretrained_nets.py:103-117) taps the Activation layers
  activation_7 / _18 / _77 / _136 / _187 and records their sizes for a 256x384 input: 128x192, 64x96, 32x48, 16x24, 8x12;
* the unnamed keras Activation layers are numbered in creation order: 188 of them (activation .. activation_187).
"""
import torch

# the "NASNetMobile" entry of the reference's scaled_layers.json: [layer index, layer name, height, width] at 256x384
SCALED_LAYERS_NASNET_MOBILE = [[22, "activation_7", 128, 192], [79, "activation_18", 64, 96], [316, "activation_77", 32, 48],
                               [553, "activation_136", 16, 24], [768, "activation_187", 8, 12]]


def test_nasnet_mobile_structure():
    from xpt_mde_2021_amd.model.build_model.pretrained_nets import NASNetMobileEncoder
    torch.manual_seed(0)
    enc = NASNetMobileEncoder().eval()
    # (logical entries: the first cell carries 5 structurally-zero filters, pretrained_nets.NASNetMobileEncoder.structural_pads)
    from xpt_mde_2021_amd.model.build_model.pretrained_nets import logical_entries, logical_view
    sel = logical_entries(enc)
    count = lambda t: logical_view(t, sel.get(id(t))).numel()          # noqa: E731
    trainable = sum(count(p) for p in enc.parameters() if p.requires_grad)
    stats = sum(count(b) for n, b in enc.named_buffers() if n.endswith("running_mean") or n.endswith("running_var"))
    assert trainable == 4_232_978
    assert stats == 36_738
    assert trainable + stats == 4_269_716
    assert enc.num_activations == 188
    assert [f"activation_{k}" for k in enc.TAP_ACTIVATIONS] == [row[1] for row in SCALED_LAYERS_NASNET_MOBILE]
    with torch.no_grad():
        taps = enc(torch.rand(1, 3, 256, 384) * 2 - 1)
    assert [tuple(t.shape[2:]) for t in taps] == [(row[2], row[3]) for row in SCALED_LAYERS_NASNET_MOBILE]
    assert tuple(t.shape[1] for t in taps) == tuple(enc.TAP_CHANNELS) == (32, 22, 88, 176, 1056)
    assert all(bool(torch.isfinite(t).all()) for t in taps)
