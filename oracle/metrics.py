"""Error measures shared by the parity tests and bench.py's parity leg (TEST INFRASTRUCTURE ONLY)."""
import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def per_gaussian_err(a, b, eps=1e-3):
    """max over Gaussians i of |a_i - b_i| / (|b_i| + eps * max_j |b_j|), |.| = Euclidean norm of the Gaussian's row.

    A global relative L2 hides an error that sits on splats with a small gradient; this measure is relative to each
    Gaussian's OWN gradient, with a floor of eps x the largest gradient so that rows whose reference is (near) zero
    by cancellation do not divide by nothing.  `a`, `b`: (P, ...) arrays, b the reference."""
    a = np.asarray(a, np.float64).reshape(np.shape(a)[0], -1)
    b = np.asarray(b, np.float64).reshape(np.shape(b)[0], -1)
    d = np.linalg.norm(a - b, axis=1)
    r = np.linalg.norm(b, axis=1)
    floor = eps * max(float(r.max()) if r.size else 0.0, 1e-30)
    return float((d / (r + floor)).max()) if d.size else 0.0
