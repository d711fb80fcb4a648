import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from xpt_mde_2021_amd.hip import ops, roofline as rf, lib as _lib
from xpt_mde_2021_amd.utils import synthetic_data as sd
lib = _lib.load()
feats = {k: v.cuda() for k, v in sd.make_features(8, 128, 416).items()}
for fw, bw, mr, pipe in ((4096, 1536, 8, 0), (4096, 1536, 8, 1), (4096, 1536, 8, 2), (4096, 1536, 8, 10), (2048, 1024, 8, 1), (4096, 3072, 8, 1)):
    assert lib.xpt_photo_fused_tune(fw, bw, mr) == 0
    assert lib.xpt_photo_fused_variant(pipe) == 0
    out = []
    for batch in (8, 32, 128):
        f, b, fb, bb, shape = rf.measure_fused(ops, feats, 20, batch=batch)
        out.append(f"B={batch}: fwd {f*1e3:6.1f} us ({fb/f/1e6:5.0f} GB/s) bwd {b*1e3:6.1f} us ({bb/b/1e6:5.0f} GB/s)")
    print(f"min_waves fwd {fw} bwd {bw} min_rows {mr} pipe {pipe} | " + " | ".join(out), flush=True)
