import numpy as np
import torch


def frac_close(actual, expected, atol, rtol=0.0, max_bad_frac=0.0, what=""):
    """assert |a-e| <= atol + rtol*|e| for all but `max_bad_frac` of the elements (bilinear validity /
    floor() flips at exact-integer coordinates are measure-zero events that fp32 rounding can toggle)."""
    a = actual.detach().double().cpu().numpy()
    e = expected.detach().double().cpu().numpy()
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    err = np.abs(a - e)
    bad = err > (atol + rtol * np.abs(e))
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_bad_frac, (f"{what}: {bad.sum()} / {bad.size} elements off (frac {frac:.2e} > {max_bad_frac:.1e}); "
                                  f"max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}; "
                                  f"ref scale {np.abs(e).max():.3e}")
    return float(err.max())


# ------------------------------------------------------------------------------------------------ flip-aware parity
# The bilinear validity test is strict and its coordinate gradient is piecewise constant (bilinear_interp.py:41-49,
# 60-75): a projected coordinate within fp32 rounding of an integer (floor() picks another cell) or of the validity
# border (0, w-1, h-1) legitimately lands on the other side in any two fp32 implementations.  Those pixels are
# PREDICTABLE from the fp64 oracle, so the tight-bar tests take them out on BOTH sides -- their depth is set to 0, which
# bilinear_interp.py:69-73 treats as invalid in every view -- and hold everything else to the north-star tolerance with
# zero outliers.
def risky_pixels(depth, T, K, scale, eps=1e-3, eps_z=2e-2):
    """[B,h,w] bool: target pixels whose projection into ANY view lies within eps_px of an integer coordinate (the
    validity borders 0, w-1, h-1 are integers too), from the fp64 oracle.  eps_px = max(eps, eps_z / |z'|): the fp32
    rounding of the homogeneous coordinates (terms of magnitude ~ column x depth) moves u' = x / z' by about 1e-2 / z'
    pixels, in TensorFlow's fp32 chain (synthesize_base.py:126-178) as much as in the kernels.
    depth [B,h,w,1] or [B,h,w], T [B,N,4,4], K [B,3,3] (full resolution)."""
    from oracle import ref_synthesize as rs
    B, h, w = depth.shape[:3]
    d = depth.detach().double().reshape(B, h, w, 1)
    Ks = rs.scale_intrinsic(K.detach().double(), scale)
    cam = rs.transform_to_source(rs.pixel2cam(rs.pixel_meshgrid(h, w, dtype=torch.float64), d, Ks), T.detach().double())
    z = cam[:, :, 2]                                           # [B,N,P]: depth of the point in the source camera
    coords = rs.cam2pixel(cam, Ks)
    u, v = coords[:, :, 0], coords[:, :, 1]
    eps_px = torch.clamp(eps_z / z.abs().clamp_min(1e-12), min=eps)
    near_int = ((u - torch.round(u)).abs() < eps_px) | ((v - torch.round(v)).abs() < eps_px)
    inside = (u > -1.0) & (u < w) & (v > -1.0) & (v < h)       # far outside the image nothing can flip
    risky = (near_int & inside & torch.isfinite(u) & torch.isfinite(v)).any(dim=1)
    return risky.reshape(B, h, w)


def flip_safe_depth(depth, T, K, scale, eps=1e-3, nudge=False):
    """depth with the risky pixels (see risky_pixels) set to 0 = invalid on both sides; also returns their share.
    nudge=True: instead of invalidating them, the risky pixels' depths are scaled by a per cent or two until their
    projections are clear of every integer coordinate (for comparisons that also take 1 / depth: the reference's
    safe_reciprocal_number is NaN at depth 0, util_funcs.py:157-160)."""
    out = depth.clone()
    risky = risky_pixels(out, T, K, scale, eps)
    share = float(risky.float().mean())
    if not nudge:
        out.reshape(risky.shape)[risky] = 0.0
        return out, share
    for k in range(12):
        if not bool(risky.any()):
            return out, share
        out.reshape(risky.shape)[risky] *= 1.0 + 0.0137 * (1 + k % 3)
        risky = risky_pixels(out, T, K, scale, eps)
    # what is left does not move with its depth (no parallax along that axis): its d_depth term is as insensitive as its
    # projection, and it is one pixel in a sum for the pose gradient
    return out, share
