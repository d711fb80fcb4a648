"""Source footprint of the fused march's target blocks on the roofline leg's synthetic inputs (CPU, numpy):
for blocks of 64 columns x FB rows of one view: width / height of the bounding box of the touched source texels,
and the share of blocks (weighted by valid pixels) that fit an LDS tile of TW texels x TH rows."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from xpt_mde_2021_amd.utils import synthetic_data as sd
from oracle import ref_pose


def coords(B, H, W, scale=1):
    g = torch.Generator().manual_seed(5)
    depth = sd.smooth_depth(B, H, W, g)[..., 0].double()
    pose = sd.random_poses(B, 4, g)
    K = sd.kitti_like_intrinsic(B, H, W).double()
    T = ref_pose.pose_rvec2matr_batch(pose.double())
    if scale > 1:
        depth = depth[:, ::scale, ::scale]
        K = K.clone(); K[:, :2] /= scale
        H, W = H // scale, W // scale
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    pix = torch.stack([u, v, torch.ones_like(u)], 0).reshape(3, -1)
    out = []
    for b in range(B):
        X = (torch.linalg.inv(K[b]) @ pix) * depth[b].reshape(1, -1)
        for n in range(4):
            Xs = T[b, n, :3, :3] @ X + T[b, n, :3, 3:4]
            q = K[b] @ Xs
            up, vp = q[0] / (q[2] + 1e-10), q[1] / (q[2] + 1e-10)
            out.append((up.reshape(H, W).numpy(), vp.reshape(H, W).numpy()))
    return out, H, W


def stats(B=8, H=128, W=416, scale=1, FB=8, strip=62):
    views, h, w = coords(B, H, W, scale)
    recs = []
    for up, vp in views:
        fu, fv = np.floor(up), np.floor(vp)
        ok = (fu >= 0) & (fu <= w - 2) & (fv >= 0) & (fv <= h - 2)
        for r0 in range(0, h, FB):
            for c0 in range(0, w, strip):
                sl = (slice(max(r0 - 1, 0), min(r0 + FB + 1, h)), slice(max(c0 - 1, 0), min(c0 + strip + 1, w)))
                m = ok[sl]
                if not m.any():
                    recs.append((0, 0, 0)); continue
                a, b_ = fu[sl][m], fv[sl][m]
                recs.append((a.max() - a.min() + 2, b_.max() - b_.min() + 2, m.sum()))
    r = np.array(recs)
    return r


for scale in (1, 2, 4, 8):
    for FB in (6, 8):
        r = stats(scale=scale, FB=FB)
        tot = r[:, 2].sum()
        line = f"scale {scale} FB {FB}: blocks {len(r)}, valid px share {tot / (8*4*(128//scale)*(416//scale)):.3f} | "
        for TW, TH in ((84, FB + 4), (84, FB + 6), (128, FB + 4), (128, FB + 6), (128, FB + 8), (170, FB + 8)):
            fit = (r[:, 0] <= TW) & (r[:, 1] <= TH)
            line += f"{TW}x{TH}: {100 * r[fit, 2].sum() / tot:5.1f}%  "
        print(line, flush=True)
    nz = r[r[:, 2] > 0]
    print("   width pct 50/90/99:", np.percentile(nz[:, 0], [50, 90, 99]), " height:", np.percentile(nz[:, 1], [50, 90, 99]))
