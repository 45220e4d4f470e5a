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


mmd = MMDb
