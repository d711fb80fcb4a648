"""NASNet-Mobile encoder against an INDEPENDENT restatement and the Keras variable manifest.

oracle/ref_nasnet.py writes tf.keras.applications.NASNetMobile(include_top=False) a second time -- NHWC, HWIO kernels,
weights addressed by Keras variable names, plain pad / conv2d / pool calls, no code shared with
xpt_mde_2021_amd/model/build_model/pretrained_nets.py.  Here:
  * the committed manifest (names + shapes) equals what the restatement declares, and carries the published structural
    facts (4,269,716 elements, 188 unnamed activations, tap shapes of scaled_layers.json);
  * the product's Keras weight loader consumes exactly the manifest (strict), round-trips, and refuses partial files;
  * on the same weights the product encoder (CPU, fp32) and the restatement produce the same five taps: a mis-wired
    adjust block, a wrong correct_pad or a swapped branch shows here (reference call site pretrained_nets.py:36-44,
    taps :103-117 / scaled_layers.json "NASNetMobile").
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_nasnet
from xpt_mde_2021_amd.model.build_model import pretrained_nets as pn
from xpt_mde_2021_amd.utils.util_class import WrongInputException

MANIFEST = os.path.join(os.path.dirname(__file__), "golden", "nasnet_mobile_manifest.json")


@pytest.fixture(scope="module")
def manifest():
    with open(MANIFEST) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def weights():
    return ref_nasnet.random_weights(seed=7)


@pytest.fixture(scope="module")
def encoder(weights):
    torch.manual_seed(0)
    net = pn.NASNetMobileEncoder().float().eval()
    pn.load_keras_weights(net, {k: v.numpy() for k, v in weights.items()})
    return net


def test_manifest_is_what_the_restatement_declares(manifest):
    variables, info = ref_nasnet.manifest(128, 416)
    assert [[n, list(s)] for n, s in variables.items()] == manifest["variables"]
    assert info["total_elements"] == manifest["total_elements"] == 4269716       # Keras: NASNetMobile, include_top=False
    assert info["unnamed_activations"] == manifest["unnamed_activations"] == 188
    # scaled_layers.json "NASNetMobile": activation_7 / 18 / 77 / 136 / 187 at 1/2 ... 1/32 of 128 x 416
    assert manifest["tap_shapes"] == {"activation_7": [64, 208, 32], "activation_18": [32, 104, 22],
                                      "activation_77": [16, 52, 88], "activation_136": [8, 26, 176],
                                      "activation_187": [4, 13, 1056]}


def test_every_manifest_variable_lands_on_exactly_one_tensor_of_the_encoder(manifest):
    net = pn.NASNetMobileEncoder()
    table = pn.keras_variable_map(net)
    assert sorted(table) == sorted(n for n, _ in manifest["variables"])
    seen = set()
    # (the first cell is built with 16 filters of which 5 are structurally zero: a variable addresses the LOGICAL entries of its
    #  tensor -- pretrained_nets.NASNetMobileEncoder.structural_pads)
    sel = pn.logical_entries(net)
    for name, shape in manifest["variables"]:
        tensor, kind = table[name]
        assert list(pn._to_keras(kind, pn.logical_view(tensor, sel.get(id(tensor)))).shape) == shape, name
        assert tensor.data_ptr() not in seen, f"{name} shares storage with another variable"
        seen.add(tensor.data_ptr())
    # ... and nothing of the encoder is left unfilled (BatchNorm's num_batches_tracked does not exist: frozen statistics)
    own = {t.data_ptr() for t in list(net.parameters()) + list(net.buffers())}
    assert own == seen
    tensors = list(net.parameters()) + list(net.buffers())
    assert sum(pn.logical_view(t, sel.get(id(t))).numel() for t in tensors) == manifest["total_elements"]


def test_structural_zeros_stay_zero_and_change_nothing(weights):
    """The 16-filter first cell with 5 structurally-zero filters computes what the 11-filter cell computes: same taps (CPU,
    fp32) from the same Keras variables; the zero entries are zero after loading, and the gradient of a loss reaches them as
    exact zeros (so the optimiser leaves them there)."""
    import importlib
    x = torch.rand((1, 3, 64, 96), generator=torch.Generator().manual_seed(3)) * 255
    arrays = {k: v.numpy() for k, v in weights.items()}
    outs = {}
    try:
        for filters in (11, 16):
            pn._STEM1_FILTERS, pn._STEM2_FILTERS = filters, (22 if filters == 11 else 24)
            net = pn.NASNetMobileEncoder().float().eval()
            pn.load_keras_weights(net, arrays)
            if filters == 16:
                assert net.cells[0].conv.weight.shape[0] == 16 and net.cells[1].conv.weight.shape[0] == 24 and len(net.structural_pads()) > 120
                before = [(t.clone(), o, i) for t, o, i in net.structural_pads()]
                net.apply_structural_zeros()
                assert all(torch.equal(a, t) for (a, _, _), (t, _, _) in zip(before, net.structural_pads()))
                for q in net.parameters():
                    q.requires_grad_(True)
                ys = net(x)
                sum((y * y).mean() for y in ys).backward()
                sel = pn.logical_entries(net)
                for q in net.parameters():
                    if id(q) in sel:
                        o, i = sel[id(q)]
                        keep = torch.zeros(q.shape[:2] if i is not None else q.shape[:1], dtype=torch.bool)
                        oo = o if o is not None else torch.arange(q.shape[0])
                        if i is not None:
                            keep[oo[:, None], i[None, :]] = True
                        else:
                            keep[oo] = True
                        pad_grad = q.grad[~keep]
                        assert float(pad_grad.abs().max()) == 0.0 if pad_grad.numel() else True
                outs[filters] = [y.detach() for y in ys]
            else:
                assert not net.structural_pads()
                with torch.no_grad():
                    outs[filters] = net(x)
    finally:
        pn._STEM1_FILTERS, pn._STEM2_FILTERS = 16, 24
    for a, b in zip(outs[11], outs[16]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))


def test_loader_round_trips_and_is_strict(weights, tmp_path):
    net = pn.NASNetMobileEncoder()
    arrays = {k: v.numpy() for k, v in weights.items()}
    path = tmp_path / "nasnet_mobile.npz"
    np.savez(path, **arrays)
    assert pn.load_keras_weights(net, str(path)) == len(arrays)
    back = pn.export_keras_weights(net)
    for k, v in arrays.items():
        assert np.array_equal(back[k].numpy(), v), k
    short = dict(arrays)
    short.pop("normal_bn_1_5/gamma")
    with pytest.raises(WrongInputException, match="1 variables missing"):
        pn.load_keras_weights(pn.NASNetMobileEncoder(), short)
    extra = dict(arrays, **{"predictions/kernel": np.zeros((1056, 1000), np.float32)})
    with pytest.raises(WrongInputException, match="not part of the no-top model"):
        pn.load_keras_weights(pn.NASNetMobileEncoder(), extra)
    bad = dict(arrays)
    bad["stem_conv1/kernel"] = np.zeros((3, 3, 32, 3), np.float32)            # OIHW-like order instead of HWIO
    with pytest.raises(WrongInputException, match="stem_conv1/kernel"):
        pn.load_keras_weights(pn.NASNetMobileEncoder(), bad)


def test_pretrained_model_loads_from_the_environment(weights, tmp_path, monkeypatch):
    path = tmp_path / "w.npz"
    np.savez(path, **{k: v.numpy() for k, v in weights.items()})
    monkeypatch.delenv("XPT_NASNET_WEIGHTS", raising=False)
    with pytest.raises(WrongInputException, match="XPT_NASNET_WEIGHTS"):
        pn.PretrainedModel("NASNetMobile", True)
    monkeypatch.setenv("XPT_NASNET_WEIGHTS", str(path))
    net = pn.PretrainedModel("NASNetMobile", True).encoder()
    assert torch.equal(net.stem_bn.running_var, weights["stem_bn1/moving_variance"])


@pytest.mark.parametrize("hw", [(128, 416), (64, 96)])
def test_product_encoder_equals_the_independent_restatement(encoder, weights, hw):
    g = torch.Generator().manual_seed(11)
    image = torch.rand((2, hw[0], hw[1], 3), generator=g) * 2 - 1                  # NHWC in [-1, 1] (tfrecord_reader.py:93)
    with torch.no_grad():
        ref = ref_nasnet.forward(weights, image)
        got = encoder(image.permute(0, 3, 1, 2).contiguous())
    assert [tuple(t.shape[1:]) for t in ref] == [(hw[0] >> k, hw[1] >> k, c) for k, c in zip(range(1, 6), pn.NASNetMobileEncoder.TAP_CHANNELS)]
    for k, (r, o) in enumerate(zip(ref, got)):
        o = o.permute(0, 2, 3, 1)
        scale = float(r.abs().max())
        assert scale > 1e-3, f"tap {k} is degenerate"
        assert float((o - r).abs().max()) <= 1e-5 * max(scale, 1.0) + 2e-5 * scale, (k, float((o - r).abs().max()), scale)


def test_restatement_catches_a_rewired_cell(encoder, weights):
    """The comparison has teeth: swapping two branches of one cell in the PRODUCT changes a tap by far more than the bar."""
    g = torch.Generator().manual_seed(3)
    image = torch.rand((1, 64, 96, 3), generator=g) * 2 - 1
    cell = encoder.cells[3]
    cell.left1, cell.left2 = cell.left2, cell.left1
    try:
        with torch.no_grad():
            ref = ref_nasnet.forward(weights, image)
            got = encoder(image.permute(0, 3, 1, 2).contiguous())
    finally:
        cell.left1, cell.left2 = cell.left2, cell.left1
    worst = max(float((o.permute(0, 2, 3, 1) - r).abs().max()) / float(r.abs().max()) for r, o in zip(ref, got))
    assert worst > 1e-3


@pytest.mark.gpu
def test_gfx950_encoder_equals_the_independent_restatement(gpu_device, weights):
    """The fp32 HIP path (own depthwise / pointwise / cell-tail kernels) against oracle/ref_nasnet.py evaluated in fp64 on
    the same Keras variables: the five taps, and the gradient of a random functional of each tap w.r.t. the input image.

    The network is piecewise linear between its ReLU / max-pool switches, and the gradient of the DEEPEST tap (1/32, behind
    all 188 activations) moves by ~3 % of its elements when ONE early pre-activation that lies within fp32 rounding of zero
    lands on the other side -- which it does or does not depending on the summation order of whoever evaluates the network
    in fp32 (tools/lab/nasnet_grad_probe.py: the restatement in fp32, the product on the CPU and the product on the GPU
    disagree with the fp64 restatement in exactly that way, differently on an 8-thread and a 128-thread host).  Such an
    image says nothing about the kernels, so: every image must agree to the loose bar a mis-wired cell cannot meet (a swapped
    branch moves ALL elements), and at least one of up to four images must agree to the tight one: at most 0.2 % of the
    gradient elements of any tap off by more than 2e-3 of the scale (single late switches still move ~0.1 % on some boxes --
    the library GEMMs / the stem convolution of the fp32 path do not sum in the same order everywhere --, a wrong border or
    pad rule moves the 4 % of the elements that sit on the perimeter)."""
    from tests.util import frac_close
    net = pn.NASNetMobileEncoder().float().eval()
    pn.load_keras_weights(net, {k: v.numpy() for k, v in weights.items()})
    net = net.to(gpu_device)
    w64 = {k: v.double() for k, v in weights.items()}
    tight = False
    for seed in (5, 6, 7, 8):
        image = (torch.rand((2, 64, 192, 3), generator=torch.Generator().manual_seed(seed)) * 2 - 1)
        x_ref = image.double().requires_grad_(True)
        x_dev = image.permute(0, 3, 1, 2).contiguous().to(gpu_device).requires_grad_(True)
        ref = ref_nasnet.forward(w64, x_ref)
        got = net(x_dev)
        for k, (r, o) in enumerate(zip(ref, got)):
            frac_close(o.permute(0, 2, 3, 1), r, 2e-4 * float(r.abs().max()), what=f"image {seed}, tap {k}")
        probes = [torch.randn(r.shape, generator=torch.Generator().manual_seed(20 + k)) for k, r in enumerate(ref)]
        shares = []
        for k in range(5):
            g_ref, = torch.autograd.grad((ref[k] * probes[k].double()).sum(), x_ref, retain_graph=True)
            g_dev, = torch.autograd.grad((got[k].permute(0, 2, 3, 1) * probes[k].to(gpu_device)).sum(), x_dev, retain_graph=True)
            scale = float(g_ref.abs().max())
            err = (g_dev.permute(0, 2, 3, 1).double().cpu() - g_ref).abs()
            share = float((err > 2e-3 * scale).double().mean())
            shares.append(share)
            # loose bar, every image and tap: a few per cent of the elements behind one flipped switch, never more
            assert share <= 0.1 and float(err.max()) <= 0.2 * scale, (seed, k, share, float(err.max()) / scale)
        if max(shares) <= 2e-3:
            tight = True
            break
    assert tight, f"no image of four agreed with the fp64 restatement to 0.2 % of the gradient elements (last: {shares})"


def test_checkpoints_hold_logical_shapes_and_cross_load_between_paddings(tmp_path):
    """model_wrappers.save_weights / load_weights (model_wrappers.py:101-117): the file carries the reference's LOGICAL
    shapes (11 / 22-filter stem cells) whatever padding the running build uses, loads into a model of the other padding
    with the same function, re-applies the structural zeros, and still accepts a physical-shaped file of an earlier round."""
    from xpt_mde_2021_amd.model.build_model import model_wrappers as mw
    x = torch.rand((1, 3, 64, 96), generator=torch.Generator().manual_seed(5)) * 255
    try:
        pn._STEM1_FILTERS, pn._STEM2_FILTERS = 16, 24
        torch.manual_seed(11)
        padded = pn.NASNetMobileEncoder().float().eval()
        state = mw.logical_state_dict(padded)
        assert state["cells.0.conv.weight"].shape[0] == 11 and state["cells.1.conv.weight"].shape[0] == 22
        pn._STEM1_FILTERS, pn._STEM2_FILTERS = 11, 22
        plain = pn.NASNetMobileEncoder().float().eval()
        assert {k: tuple(v.shape) for k, v in plain.state_dict().items()} == {k: tuple(v.shape) for k, v in state.items()}
        mw.load_logical_state_dict(plain, state)
        with torch.no_grad():
            for a, b in zip(padded(x), plain(x)):
                assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))
        # back into a padded model, through a file; poisoned pads in a physical-shaped file are zeroed again
        pn._STEM1_FILTERS, pn._STEM2_FILTERS = 16, 24
        again = pn.NASNetMobileEncoder().float().eval()
        torch.save(mw.logical_state_dict(plain), tmp_path / "enc.pt")
        mw.load_logical_state_dict(again, torch.load(tmp_path / "enc.pt"))
        for (k, a), (_, b) in zip(padded.state_dict().items(), again.state_dict().items()):
            assert torch.equal(a, b), k
        physical = {k: v.clone() for k, v in padded.state_dict().items()}
        physical["cells.0.conv.weight"] += 1.0                       # non-zero structural entries
        mw.load_logical_state_dict(again, physical)
        w = again.cells[0].conv.weight.detach()
        assert float(w[11:].abs().max()) == 0.0 and torch.equal(w[:11], physical["cells.0.conv.weight"][:11])
        with pytest.raises(RuntimeError, match="shape"):
            mw.load_logical_state_dict(again, dict(state, **{"stem_conv.weight": torch.zeros(3, 3)}))
    finally:
        pn._STEM1_FILTERS, pn._STEM2_FILTERS = 16, 24


def test_wrapper_checkpoint_file_has_the_reference_shapes(tmp_path):
    """ModelWrapper.save_weights -> load_weights through files: depthnet_*.pt holds 87 input channels for up2.conv2 (64 upconv
    + 22 skip + 1 prediction, depth_net.py:101-109) and 11 / 22-filter stem cells; reloading restores the physical model."""
    from xpt_mde_2021_amd.config import opts
    from xpt_mde_2021_amd.model.build_model.model_factory import ModelFactory
    model = ModelFactory({"imshape": (5, 64, 96, 3)}, global_batch=1, net_names=opts.RIGID_NET).get_model()
    dn = model.models["depthnet"]
    assert dn.up2.conv2.conv.weight.shape[1] == 89
    model.save_weights(str(tmp_path), "ep00")
    disk = torch.load(tmp_path / "depthnet_ep00.pt")
    assert disk["up2.conv2.conv.weight"].shape[1] == 87
    assert disk["encoder.cells.0.conv.weight"].shape[0] == 11 and disk["encoder.cells.1.conv.weight"].shape[0] == 22
    before = {k: v.clone() for k, v in dn.state_dict().items()}
    with torch.no_grad():
        for p in dn.parameters():
            p.add_(1.0)
    model.load_weights(str(tmp_path), "ep00")
    for k, v in dn.state_dict().items():
        assert torch.equal(v, before[k]), k
