"""Host-side helpers that user training scripts import next to the model (the reference exports them
from gpsa/__init__.py:3-10).  Small, dependency-free re-implementations; the count-data preprocessing
and the plotting callbacks of the reference are outside the hot path's scope.
"""
import numpy as np


def rbf_kernel_numpy(x, xp, kernel_params):
    """numpy RBF of the data simulators; ``kernel_params = [log variance, log lengthscale(s)...]``"""
    scale = 1.0 / np.exp(np.asarray(kernel_params[1:]))
    a, b = x * scale, xp * scale
    sq = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
    return np.exp(kernel_params[0]) * np.exp(-0.5 * np.maximum(sq, 0.0))


def polar_warp(X, r, theta):
    """displace 2-D points by the polar offsets (r, theta)"""
    out = np.array(X[:, :2], dtype=float, copy=True)
    out[:, 0] += r * np.cos(theta)
    out[:, 1] += r * np.sin(theta)
    return out


def get_st_coordinates(df):
    """spot names of the form 'AxB' (index of an ST data frame) -> float array [[A, B], ...]"""
    return np.asarray([tuple(map(float, str(label).split("x"))) for label in df.index])


def compute_distance(X1, X2):
    """mean Euclidean distance between matched rows"""
    return float(np.linalg.norm(np.asarray(X1) - np.asarray(X2), axis=1).mean())


class ConvergenceChecker:
    """Fits a cubic to the last ``span`` loss values (least squares through an orthonormal basis) and
    reports the relative change between the last two fitted values."""

    def __init__(self, span, dtp="float64"):
        self.span = span
        t = np.arange(span, dtype=dtp) - (span - 1) / 2.0
        self.U, _, _ = np.linalg.svd(np.vander(t, 4, increasing=True), full_matrices=False)

    def smooth(self, y):
        return self.U @ (self.U.T @ y)

    def subset(self, y, idx=-1):
        first = idx - self.span + 1
        return y[first:] if idx == -1 else y[first : idx + 1]

    def relative_change(self, y, idx=-1, smooth=True):
        window = self.subset(y, idx=idx)
        fitted = self.smooth(window) if smooth else window
        return (fitted[-1] - fitted[-2]) / (0.1 + abs(fitted[-2]))

    def converged(self, y, tol=1e-4, **kwargs):
        return abs(self.relative_change(y, **kwargs)) < tol

    def relative_change_all(self, y, smooth=True):
        changes = np.full(len(y), np.nan)
        for i in range(self.span, len(y)):
            changes[i] = self.relative_change(y, idx=i, smooth=smooth)
        return changes

    def converged_all(self, y, tol=1e-4, smooth=True):
        return np.abs(self.relative_change_all(y, smooth=smooth)) < tol


class LossNotDecreasingChecker:
    """``check_loss(i, trace)`` turns True once the mean per-step decrease of the loss over the last
    ``window_size - 1`` steps drops below ``atol``."""

    def __init__(self, max_epochs, atol=1e-2, window_size=10):
        self.max_epochs, self.atol, self.window_size = max_epochs, atol, window_size
        self.decrease_in_loss = np.zeros(max_epochs)
        self.average_decrease_in_loss = np.zeros(max_epochs)

    def check_loss(self, iternum, loss_trace):
        if iternum < 1:
            return False
        self.decrease_in_loss[iternum] = loss_trace[iternum - 1] - loss_trace[iternum]
        if iternum < self.window_size:
            return False
        recent = self.decrease_in_loss[iternum - self.window_size + 1 : iternum]
        self.average_decrease_in_loss[iternum] = recent.mean()
        return bool(self.average_decrease_in_loss[iternum] < self.atol)
