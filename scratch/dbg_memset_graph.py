import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
hip.hipMemsetD32Async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetD32Async.restype = ctypes.c_int
for n in (16, 1 << 20):
  for fn_name in ("hipMemsetAsync", "hipMemsetD32Async", "zero_"):
    buf = torch.full((n,), 5.0, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        if fn_name == "hipMemsetAsync":
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, n * 4, st)
        elif fn_name == "hipMemsetD32Async":
            rc = hip.hipMemsetD32Async(buf.data_ptr(), 0, n, st)
        else:
            buf.zero_(); rc = 0
        buf += 1.0
    vals = []
    for _ in range(3):
        g.replay(); torch.cuda.synchronize(); vals.append((float(buf[0]), float(buf[-1])))
    print(f"n={n} {fn_name} rc={rc} values after replays {vals}")
print(torch.__version__, torch.version.hip)
