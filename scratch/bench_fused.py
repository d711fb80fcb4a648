import sys, torch
sys.path.insert(0, "/root/repo")
from xpt_mde_2021_amd.hip import ops, roofline as rf
from xpt_mde_2021_amd.utils import synthetic_data as sd
feats = {k: v.cuda() for k, v in sd.make_features(8, 128, 416).items()}
for batch in (8, 32, 128):
    f, b, fb, bb, shape = rf.measure_fused(ops, feats, 20, batch=batch)
    print(f"B={batch}: fwd {f*1e3:.1f} us ({fb/f/1e6:.0f} GB/s)  bwd {b*1e3:.1f} us ({bb/b/1e6:.0f} GB/s)")
