"""Host-side helpers exported by the reference package (gpsa/__init__.py:3-10, gpsa/util/util.py).

Only the small, dependency-free ones that user training scripts import next to the model are
provided; the count-data preprocessing and plotting callbacks of the reference are out of the hot
path's scope.
"""
import numpy as np


def rbf_kernel_numpy(x, xp, kernel_params):
    """numpy RBF used by the data simulators: params = [log variance, log lengthscale(s)]."""
    var = np.exp(kernel_params[0])
    ell = np.exp(kernel_params[1:])
    d = (x / ell)[:, None, :] - (xp / ell)[None, :, :]
    return var * np.exp(-0.5 * np.sum(d * d, axis=2))


def polar_warp(X, r, theta):
    """shift 2-D points by polar offsets (r, theta)"""
    return np.stack([X[:, 0] + r * np.cos(theta), X[:, 1] + r * np.sin(theta)], axis=1)


def get_st_coordinates(df):
    """'AxB'-style spot names of an ST data frame index -> float coordinate array"""
    return np.array([[float(t) for t in str(name).split("x")] for name in df.index])


def compute_distance(X1, X2):
    """mean Euclidean distance between matched rows"""
    return np.mean(np.sqrt(np.sum((X1 - X2) ** 2, axis=1)))


class ConvergenceChecker:
    """Cubic-polynomial smoothing of the last ``span`` loss values; relative-change stopping rule."""

    def __init__(self, span, dtp="float64"):
        self.span = span
        t = np.arange(span, dtype=dtp)
        t -= t.mean()
        basis = np.column_stack([np.ones_like(t), t, t**2, t**3])
        self.U = np.linalg.svd(basis, full_matrices=False)[0]

    def smooth(self, y):
        return self.U @ (self.U.T @ y)

    def subset(self, y, idx=-1):
        lo = idx - self.U.shape[0] + 1
        return y[lo:] if idx == -1 else y[lo : idx + 1]

    def relative_change(self, y, idx=-1, smooth=True):
        y = self.subset(y, idx=idx)
        if smooth:
            y = self.smooth(y)
        return (y[-1] - y[-2]) / (0.1 + abs(y[-2]))

    def converged(self, y, tol=1e-4, **kwargs):
        return abs(self.relative_change(y, **kwargs)) < tol

    def relative_change_all(self, y, smooth=True):
        out = np.full(len(y), np.nan)
        for i in range(self.U.shape[0], len(y)):
            out[i] = self.relative_change(y, idx=i, smooth=smooth)
        return out

    def converged_all(self, y, tol=1e-4, smooth=True):
        return np.abs(self.relative_change_all(y, smooth=smooth)) < tol


class LossNotDecreasingChecker:
    """True once the windowed mean decrease of the loss trace falls below ``atol``."""

    def __init__(self, max_epochs, atol=1e-2, window_size=10):
        self.max_epochs = max_epochs
        self.atol = atol
        self.window_size = window_size
        self.decrease_in_loss = np.zeros(max_epochs)
        self.average_decrease_in_loss = np.zeros(max_epochs)

    def check_loss(self, iternum, loss_trace):
        if iternum < 1:
            return False
        self.decrease_in_loss[iternum] = loss_trace[iternum - 1] - loss_trace[iternum]
        if iternum < self.window_size:
            return False
        w = self.decrease_in_loss[iternum - self.window_size + 1 : iternum]
        self.average_decrease_in_loss[iternum] = np.mean(w)
        return bool(self.average_decrease_in_loss[iternum] < self.atol)
