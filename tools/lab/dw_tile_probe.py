"""Lab: the stride-2 tile kernels of xpt_dwconv.hip (input region of a tile in LDS) against the kernels they replace
(xpt_dwconv_tune(-20000) / (-30000) switch them off): outputs bit for bit, gradients to rounding, and hot times per launch
at the model's stride-2 shapes (batch 8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from xpt_mde_2021_amd.hip import lib as _lib, ops  # noqa: E402
from xpt_mde_2021_amd.model.model_util.layer_ops import same_pad  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
FWD_ON, BWD_ON = int(os.environ.get("TILE_FWD", "1616")), int(os.environ.get("TILE_BWD", "1"))
lib.xpt_dwconv_tune(-50000 - int(os.environ.get("TILE_FWD_MIN_LOG2", "0")))     # the check below: every shape through the tiles


BWD1_ON = int(os.environ.get("TILE_BWD1", "1"))


def tiles(on):
    lib.xpt_dwconv_tune(-20000 - (FWD_ON if on else 0))
    lib.xpt_dwconv_tune(-30000 - (BWD_ON if on else 0))
    lib.xpt_dwconv_tune(-70000 - (BWD1_ON if on else 0))
    lib.xpt_dwconv_tune(-80001)                                  # odd channel counts too


def run(C, ks, B, H, W, on, multi, relu=True, deferred=True, stride=2):
    tiles(on)
    g = torch.Generator().manual_seed(C * 7 + H)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    h, p = mk(B, C, H, W).requires_grad_(True), mk(B, C, H, W).requires_grad_(True)
    OH, OW = ((H + 1) // 2, (W + 1) // 2) if stride == 2 else (H, W)
    params = [torch.nn.Parameter((torch.randn(C, 1, k, k, generator=g) * 0.2).to(dev)) for k in ks]
    if deferred:
        for q in params:
            q.flat_grad = torch.zeros_like(q)
    gys = [mk(B, C, OH, OW) for _ in ks]
    pads = [same_pad(H, k, 2) + same_pad(W, k, 2) if stride == 2 else (k // 2,) * 4 for k in ks]
    ins = [h, h, p, p, p][:len(ks)]
    if multi:
        ys = ops.multi_depthwise(ins, params, relu_in=relu, stride=stride, pads=pads)
    else:
        ys = [ops.depthwise_conv2d(x, q, stride, pd, relu) for x, q, pd in zip(ins, params, pads)]
    torch.autograd.backward(ys, gys)
    ops.grad_sink.flush()
    torch.cuda.synchronize()
    gw = [q.flat_grad if deferred else q.grad for q in params]
    return [y.detach().float() for y in ys], [h.grad.float(), p.grad.float() if len(ks) > 2 else None], gw


bad = 0
CASES = () if os.environ.get("SKIP_CHECK") else ((32, [7, 5, 7], 2, 64, 208), (22, [5, 7, 7, 5, 3], 2, 32, 104), (88, [5, 7, 7, 5, 3], 2, 16, 52),
                       (176, [5, 7, 5], 2, 8, 26), (44, [5, 7, 7, 5, 3], 2, 10, 14), (22, [5, 7, 7, 5, 3], 2, 13, 27),
                       (8, [3, 5], 1, 5, 7), (12, [7], 3, 9, 9), (11, [5, 7, 3], 2, 32, 104), (11, [5, 7, 3], 2, 9, 15),
                       (11, [5, 7, 3], 2, 32, 104, 1), (44, [5, 3, 3, 5, 3], 2, 16, 52, 1), (22, [3, 5, 3], 2, 18, 21, 1),
                       (88, [5, 7, 3], 1, 17, 19, 1))
for case in CASES:
    C, ks, B, H, W = case[:5]
    stride = case[5] if len(case) > 5 else 2
    for multi in (True, False):
        for relu in (True, False):
            ya, ga, wa = run(C, ks, B, H, W, True, multi, relu, stride=stride)
            yb, gb, wb = run(C, ks, B, H, W, False, multi, relu, stride=stride)
            ny = sum(int((a != b).sum()) for a, b in zip(ya, yb))
            gerr = max(float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(ga, gb) if a is not None)
            werr = max(float((a - b).abs().max()) / max(1.0, float(b.abs().max())) for a, b in zip(wa, wb))
            flag = "" if ny == 0 and gerr < 2e-2 and werr < 1e-4 else "  <-- MISMATCH"
            bad += bool(flag)
            print(f"C={C} ks={ks} {B}x{H}x{W} s{stride} multi={multi} relu={relu}: y mismatches {ny}, dx err {gerr:.2e}, dw err {werr:.2e}{flag}",
                  flush=True)
print("MISMATCHES:", bad)


def timeit(fn, n=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * n)


import ctypes  # noqa: E402

print("\nhot us per launch (batch 8, C ABI): forward | backward (dx + weight-gradient partials), tiles off -> on")
B = 8
for C, ks, H, W, multi, S in ((32, [7], 64, 208, False, 2), (32, [5], 64, 208, False, 2), (22, [5, 7, 5], 32, 104, True, 2),
                              (88, [5, 7, 5], 16, 52, True, 2), (176, [5, 7, 5], 8, 26, True, 2), (88, [5], 16, 52, False, 2),
                              (176, [5], 8, 26, False, 2), (22, [5], 32, 104, False, 2), (11, [5], 64, 208, False, 2),
                              (11, [7], 32, 104, False, 1), (11, [5], 32, 104, False, 1), (11, [3], 32, 104, False, 1),
                              (22, [3, 5, 3], 16, 52, True, 1), (44, [5, 3, 5], 16, 52, True, 1), (44, [3, 3], 16, 52, True, 1)):
    g = torch.Generator().manual_seed(1)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16).contiguous(memory_format=torch.channels_last)  # noqa: E731
    n, OH, OW = len(ks), (H + 1) // 2 if S == 2 else H, (W + 1) // 2 if S == 2 else W
    h, p = mk(B, C, H, W), mk(B, C, H, W)
    ins = [h, p, p][:n]
    ws = [(torch.randn(C, 1, k, k, generator=g) * 0.2).to(dev) for k in ks]
    ys, dys = [mk(B, C, OH, OW) for _ in ks], [mk(B, C, OH, OW) for _ in ks]
    dxs = [torch.empty_like(h), torch.empty_like(p)][:min(n, 2)]
    pads = [same_pad(H, k, 2) + same_pad(W, k, 2) if S == 2 else (k // 2,) * 4 for k in ks]
    pts, pls = [pd[0] for pd in pads], [pd[2] for pd in pads]
    parts = [torch.empty(lib.xpt_dwconv_bwd_weight_chunks(B, OH, OW, C, k, S) * C * k * k, device=dev) for k in ks]
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    PU = ctypes.c_void_p * len(dxs)
    input_of = [0, 1, 1][:n]
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731

    def fwd():
        if multi:
            rc = lib.xpt_dwconv_multi_fwd(P(*[x.data_ptr() for x in ins]), P(*[w.data_ptr() for w in ws]), P(*[y.data_ptr() for y in ys]),
                                          I(*ks), I(*pts), I(*pls), n, B, H, W, C, S, OH, OW, 1, 1, st())
        else:
            rc = lib.xpt_dwconv_fwd(h.data_ptr(), ws[0].data_ptr(), ys[0].data_ptr(), B, H, W, C, ks[0], S, pts[0], pls[0], OH, OW, 1, 1, st())
        assert rc == 0, rc

    def bwd():
        if multi:
            rc = lib.xpt_dwconv_multi_bwd(PU(*[x.data_ptr() for x in [h, p][:len(dxs)]]), PU(*[d.data_ptr() for d in dxs]), len(dxs),
                                          P(*[d.data_ptr() for d in dys]), P(*[w.data_ptr() for w in ws]),
                                          P(*[q.data_ptr() for q in parts]), I(*ks), I(*pts), I(*pls), I(*input_of), n, B, H, W, C, S,
                                          OH, OW, 1, 1, st())
        else:
            rc = lib.xpt_dwconv_bwd_both(h.data_ptr(), ws[0].data_ptr(), dys[0].data_ptr(), dxs[0].data_ptr(), parts[0].data_ptr(),
                                         parts[0].numel(), B, H, W, C, ks[0], S, pts[0], pls[0], OH, OW, 1, 1, st())
        assert rc == 0, rc

    def bwd_data():
        rc = lib.xpt_dwconv_bwd_data(h.data_ptr(), ws[0].data_ptr(), dys[0].data_ptr(), dxs[0].data_ptr(), B, H, W, C, ks[0], S, pts[0],
                                     pls[0], OH, OW, 1, 1, st())
        assert rc == 0, rc

    res = []
    for on in (False, True):
        tiles(on)
        res.append((timeit(fwd), timeit(bwd), 0.0 if multi else timeit(bwd_data)))
    print(f"C={C} ks={ks} {H}x{W} s{S}: fwd {res[0][0]:6.1f} -> {res[1][0]:6.1f} | bwd {res[0][1]:6.1f} -> {res[1][1]:6.1f}"
          f" | bwd_data {res[0][2]:6.1f} -> {res[1][2]:6.1f}", flush=True)
tiles(True)
