"""Per-shape timing of the pointwise conv + BN forward: xpt_pwconv_bn_fwd vs rocBLAS GEMM + affine epilogue launch."""
import sys, collections, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from xpt_mde_2021_amd.hip import lib as _lib

lib = _lib.load()
shapes = [(416, 176, 176, 50), (416, 1056, 176, 5), (416, 704, 176, 2), (416, 528, 88, 2), (1664, 88, 88, 50), (1664, 528, 88, 5),
          (1664, 352, 88, 2), (1664, 264, 44, 2), (6656, 44, 44, 40), (6656, 264, 44, 5), (6656, 88, 44, 2), (6656, 264, 88, 2),
          (26624, 44, 22, 1), (106496, 32, 32, 1)]


def timeit(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = collections.defaultdict(float)
print("M cin cout n | own | rocblas mm | mm + affine")
for M, cin, cout, n in shapes:
    x = torch.randn(M, cin, device="cuda").bfloat16(); w = torch.randn(cout, cin, device="cuda").bfloat16()
    g_, b_, mu, var = (torch.rand(cout, device="cuda") + .5 for _ in range(4))
    ypre = torch.empty(M, cout, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(ypre)
    S = lambda: torch.cuda.current_stream().cuda_stream
    own = timeit(lambda: lib.xpt_pwconv_bn_fwd(x.data_ptr(), w.data_ptr(), g_.data_ptr(), b_.data_ptr(), mu.data_ptr(), var.data_ptr(), 1e-3, None, ypre.data_ptr(), y.data_ptr(), M, cin, cout, cin, S()))
    mm = timeit(lambda: torch.mm(x, w.t(), out=ypre))
    def both():
        torch.mm(x, w.t(), out=ypre)
        lib.xpt_affine_act_fwd(ypre.data_ptr(), g_.data_ptr(), b_.data_ptr(), mu.data_ptr(), var.data_ptr(), 1e-3, None, y.data_ptr(), M, cout, 1.0, 0, 1, S())
    lib_t = timeit(both)
    tot["own"] += own * n; tot["lib"] += lib_t * n
    print(f"{M:7d} {cin:5d} {cout:5d} x{n:3d} | {own:6.1f} | {mm:6.1f} | {lib_t:6.1f}", flush=True)
print({k: round(v) for k, v in tot.items()})
