"""utils.Statistics — maximum mean discrepancy used as the posterior parity metric
(reference: src/utils/Statistics.py:13-84; the biased estimator `MMDb` with an RBF kernel of
bandwidth sigma is what the reference's evaluation scripts report,
example/slam/small_range_gaussian_problem/icra_paper/mmd_rmse_time_da_plot_grid.py:167,245)."""
import numpy as np


def _rbf_gram(x: np.ndarray, y: np.ndarray, sigma: float) -> np.ndarray:
    d2 = (x * x).sum(1)[:, None] + (y * y).sum(1)[None, :] - 2.0 * x @ y.T
    return np.exp(-np.maximum(d2, 0.0) / (2.0 * sigma ** 2))


def MMDb(x: np.ndarray, y: np.ndarray, sigma: float = None) -> float:
    """Biased MMD estimate sqrt(mean k(x,x) + mean k(y,y) - 2 mean k(x,y)); sigma defaults to sqrt(dim)."""
    x, y = np.atleast_2d(x), np.atleast_2d(y)
    if sigma is None:
        sigma = np.sqrt(x.shape[1])
    v = _rbf_gram(x, x, sigma).mean() + _rbf_gram(y, y, sigma).mean() - 2.0 * _rbf_gram(x, y, sigma).mean()
    return float(np.sqrt(max(v, 0.0)))


def MMDu2(x: np.ndarray, y: np.ndarray, sigma: float = None) -> float:
    """Unbiased estimate of MMD^2 (may be negative)."""
    x, y = np.atleast_2d(x), np.atleast_2d(y)
    if sigma is None:
        sigma = np.sqrt(x.shape[1])
    m, n = x.shape[0], y.shape[0]
    kxx, kyy = _rbf_gram(x, x, sigma), _rbf_gram(y, y, sigma)
    return float((kxx.sum() - np.trace(kxx)) / (m * (m - 1)) + (kyy.sum() - np.trace(kyy)) / (n * (n - 1)) -
                 2.0 * _rbf_gram(x, y, sigma).mean())


def mmd(samples1: np.ndarray, samples2: np.ndarray, k_sigma2: float = 1.0) -> np.ndarray:
    """The reference's `mmd` (src/utils/Statistics.py:13-45; what icra_paper/compute_mmd.py writes to `run1/mmd`):
    sqrt(E1 + E2 - 2 E3) with the Gaussian kernel k(d) = N(d; 0, k_sigma2 I) / N(0; 0, k_sigma2 I) = exp(-|d|^2 / (2 k_sigma2)),
    E1 / E2 without the diagonal (means over i != j), E3 over all pairs.  Returns a one-element array like the
    reference (callers index `[0]`); NaN when the unbiased combination is negative, as there.  Vectorised."""
    x, y = np.atleast_2d(samples1).astype(np.float64), np.atleast_2d(samples2).astype(np.float64)
    m, n = x.shape[0], y.shape[0]
    s = np.sqrt(k_sigma2)
    kxx, kyy = _rbf_gram(x, x, s), _rbf_gram(y, y, s)
    e1 = (kxx.sum() - np.trace(kxx)) / (m * (m - 1))
    e2 = (kyy.sum() - np.trace(kyy)) / (n * (n - 1))
    e3 = _rbf_gram(x, y, s).mean()
    with np.errstate(invalid="ignore"):
        return np.sqrt(np.array([e1 + e2 - 2.0 * e3]))
