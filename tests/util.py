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
