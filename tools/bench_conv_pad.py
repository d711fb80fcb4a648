"""Does padding PoseNet's 15-channel input to 16 channels help the library convolutions (fwd / fp32 wgrad)?"""
import torch, torch.nn.functional as F
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for cin in (15, 16):
    x = torch.randn(8, cin, 128, 416, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(32, cin, 5, 5, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    y = F.conv2d(F.pad(x, (1, 2, 1, 2)), w, None, 2)
    dy = torch.randn_like(y)
    xp = F.pad(x, (1, 2, 1, 2))
    f = timeit(lambda: F.conv2d(xp, w, None, 2))
    xf, wf, dyf = xp.float(), w.float(), dy.float()
    wg = timeit(lambda: torch.ops.aten.convolution_backward(dyf, xf, wf, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))
    print(f"cin={cin}: fwd {f:.1f} us, fp32 wgrad {wg:.1f} us")
