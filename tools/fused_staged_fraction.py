"""How many row blocks of the LDS-staged fused forward (variant 2) take the staged path on the bench's synthetic data?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xpt_mde_2021_amd.hip import ops, roofline as rf, lib as _lib
from xpt_mde_2021_amd.utils import synthetic_data as sd
lib = _lib.load()
lib.xpt_photo_fused_variant(2)
feats = {k: v.cuda() for k, v in sd.make_features(8, 128, 416).items()}
x = rf._inputs(feats, None)
B, N, H, W = x["B"], x["N"], x["H"], x["W"]
T = ops.pose_rvec2matr(x["pose"])
nws = lib.xpt_photo_fused_workspace_floats(B, N, H, W)
ws = torch.zeros(nws, device="cuda")
p = lambda t: t.data_ptr()
_lib.check(lib.xpt_photo_fused_fwd(p(x["src"]), p(x["depth"]), p(T), p(x["K"]), p(x["tgt"]), None, None, None, p(ws), nws, B, N, H, W, 1.0,
                                   torch.cuda.current_stream().cuda_stream), "fwd")
torch.cuda.synchronize()
w = ws.view(-1, 16)
print(f"[staged] blocks with valid pixels {float(w[:, 3].sum()):.0f}, staged {float(w[:, 2].sum()):.0f}")
print("[staged] pose", x["pose"][0].tolist(), "depth range", float(x["depth"].min()), float(x["depth"].max()))
