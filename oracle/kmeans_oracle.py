"""CPU restatement of the reference's masked K-means (code/models/kmeans/kmeans.py) -- TEST INFRASTRUCTURE ONLY.

Pinned against tests/golden/kmeans.npz: starting centroids, every iteration's assignments and the final state the
reference itself produced under ``random.seed`` (tests/golden/make_golden.py::make_kmeans).  The product class is
``bnmtf_amd.kmeans.KMeans`` (device passes through libbnmtf_hip.so); this file is its checker and is imported by
tests only.

Follows, method by method:
  __init__           kmeans.py:7-41   (unobserved columns are dropped, fully unobserved rows rejected)
  initialise         kmeans.py:45-57  (uniform between each coordinate's observed min and max, one random.uniform per
                                       coordinate, cluster by cluster: the reference's call order)
  assignment         kmeans.py:88-119 (MSE over the coordinates point and centroid both know; None = no overlap;
                                       the first centroid, or a later one that is the first DEFINED one, or a smaller
                                       defined one wins)
  update/_cluster    kmeans.py:126-163 ('singleton': an empty cluster takes the point furthest from its centroid, whose
                                       old cluster is then updated again).  AS WRITTEN the refilled centroid is not a copy
                                       of the point but the row of X itself (`self.centroids[c] = self.X[index]`, :141, a
                                       NumPy view), so every later mean written into that centroid (:158-163) overwrites
                                       the data point; the reference's results contain this, and so do these.
  cluster            kmeans.py:70-84
"""
import random

import numpy as np

max_iterations = 200


class KMeansOracle(object):
    def __init__(self, X, M, K, resolve_empty='singleton'):
        self.X = np.array(X, dtype=float)
        self.M = np.array(M, dtype=float)
        self.K = K
        self.resolve_empty = resolve_empty
        assert len(self.X.shape) == 2, "Input matrix X is not a two-dimensional array, but instead %s-dimensional." % len(self.X.shape)
        assert self.X.shape == self.M.shape, "Input matrix X is not of the same size as the indicator matrix M: %s and %s respectively." % (self.X.shape, self.M.shape)
        assert self.K > 0, "K should be greater than 0."
        self.no_unique_points = len(set(tuple(l) for l in self.X.tolist()))      # counted before columns are dropped (:16)
        for i, c in enumerate(self.M.sum(axis=1)):
            assert c != 0, "Fully unobserved row in X, row %s." % i
        keep = self.M.sum(axis=0) > 0
        self.X, self.M = self.X[:, keep], self.M[:, keep]
        (self.no_points, self.no_coordinates) = self.X.shape
        self.distances = np.zeros(self.no_points)

    def initialise(self, seed=None):
        if seed is not None:
            random.seed(seed)
        self.mins = [self.X[self.M[:, j] > 0, j].min() for j in range(self.no_coordinates)]
        self.maxs = [self.X[self.M[:, j] > 0, j].max() for j in range(self.no_coordinates)]
        self.centroids = [np.array(self._random_centroid()) for _ in range(self.K)]     # a list: an entry may become a VIEW of a row of X
        self.cluster_assignments = -np.ones(self.no_points, dtype=int)
        self.mask_centroids = np.ones((self.K, self.no_coordinates))

    def _random_centroid(self):
        return [random.uniform(self.mins[j], self.maxs[j]) for j in range(self.no_coordinates)]

    def assignment(self):
        change = False
        for d in range(self.no_points):
            best, best_mse = None, None
            for c in range(self.K):
                both = self.M[d] * self.mask_centroids[c]
                n = both.sum()
                mse = None if n == 0 else (both * (self.X[d] - self.centroids[c]) ** 2).sum() / float(n)
                if best_mse is None or (mse is not None and mse < best_mse):
                    best, best_mse = c, mse
            self.distances[d] = np.nan if best_mse is None else best_mse
            change = change or (self.cluster_assignments[d] != best)
            self.cluster_assignments[d] = best
        return change

    def update(self):
        for c in range(self.K):
            self._update_cluster(c)

    def _update_cluster(self, c):
        members = np.nonzero(self.cluster_assignments == c)[0]
        if len(members) == 0:
            if self.no_unique_points >= self.K:
                if self.resolve_empty == 'singleton':
                    far = int(np.argmax(self.distances))
                    old = int(self.cluster_assignments[far])
                    self.centroids[c] = self.X[far]            # a view, as in the reference: see the header
                    self.mask_centroids[c] = self.M[far]
                    self.distances[far] = 0.0
                    self.cluster_assignments[far] = c
                    self._update_cluster(old)
                else:
                    self.centroids[c] = self._random_centroid()
                    self.mask_centroids[c] = np.ones(self.no_coordinates)
            return
        cnt = self.M[members].sum(axis=0)
        tot = np.array([sum(self.X[d, j] for d in members if self.M[d, j]) for j in range(self.no_coordinates)])   # in member order, like sum() of the reference's list
        self.centroids[c][:] = np.where(cnt > 0, tot / np.maximum(cnt, 1), 0.0)       # in place (through a view into X if it is one)
        self.mask_centroids[c] = (cnt > 0).astype(float)

    def cluster(self):
        iteration = 1
        change = True
        self.assign_hist = []
        while change:
            iteration += 1
            change = self.assignment()
            self.update()
            self.assign_hist.append(self.cluster_assignments.copy())
            if iteration >= max_iterations:
                break
        self.clustering_results = np.zeros((self.no_points, self.K))
        self.clustering_results[np.arange(self.no_points), self.cluster_assignments] = 1.0
