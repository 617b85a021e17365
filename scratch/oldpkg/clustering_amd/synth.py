"""Normative synthetic input of the bench/tests: SURVEY.md section 8(d).

3 isotropic Gaussian blobs in D dims; centres differ only in the first two columns:
(-1.0,-0.5), (0.0,+0.5), (+1.0,-0.5); label ~ U{0,1,2}; x = centre + N(0, sigma^2 I),
sigma = 0.08; float64 -> float32; numpy default_rng(seed); draw order: all labels
(integers(0,3,N)) then normal(0, sigma, (N, D)).
"""
import numpy as np

CENTRES = np.array([[-1.0, -0.5], [0.0, 0.5], [1.0, -0.5]], dtype=np.float64)
SIGMA = 0.08
SEED = 20240


def gaussian_blobs(n_rows, n_cols, seed=SEED, sigma=SIGMA):
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, 3, n_rows)
    x = rng.normal(0.0, sigma, (n_rows, n_cols))
    ncen = min(2, n_cols)
    x[:, :ncen] += CENTRES[labels, :ncen]
    return np.ascontiguousarray(x.astype(np.float32))
