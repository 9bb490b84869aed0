"""CPU oracle for the on-device k-means initialisation.  TEST INFRASTRUCTURE ONLY.

Plain numpy Lloyd iterations from given initial centres: nearest centre by squared Euclidean distance
(ties -> lowest index), centres <- mean of their points, an empty cluster keeps its centre.  This is
the algorithm scikit-learn's ``KMeans(algorithm="lloyd")`` runs after its initialisation, which is what
the reference calls at gpsa/models/vgpsa.py:74-76, 90-92 (``KMeans(n_clusters=...).fit(X)``; sklearn is
a third-party dependency of the reference, pinned 1.0.2 in its requirements.txt:9, 1.7.2 in this image).
Pinned in tests/test_kmeans.py against sklearn itself run from the same initial centres.
"""
import numpy as np


def lloyd(X, centres, iters):
    X = np.asarray(X, dtype=np.float64)
    C = np.asarray(centres, dtype=np.float64).copy()
    assign = None
    for _ in range(iters):
        d2 = ((X[:, None, :] - C[None, :, :]) ** 2).sum(-1)
        assign = d2.argmin(1)
        for k in range(C.shape[0]):
            m = assign == k
            if m.any():
                C[k] = X[m].mean(0)
    return C, assign
