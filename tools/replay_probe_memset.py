"""Is a hipMemsetAsync captured into a hipGraph executed again on every replay?  (PyTorch's multi-block reductions and some
library solvers zero their semaphores / workspaces that way.)  y starts at 5; graph = [memset(y, 0); y += 1]; every replay
must leave y == 1."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
for nbytes in (4, 64, 4096, 1 << 20):
    y = torch.full((nbytes // 4,), 5.0, device="cuda")
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y.add_(1)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = hip.hipMemsetAsync(y.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        y.add_(1)
    vals = []
    for it in range(4):
        g.replay(); torch.cuda.synchronize()
        vals.append((float(y.min()), float(y.max())))
    print(f"[memset] {nbytes} bytes (rc {rc}): y after replays {vals}", flush=True)
