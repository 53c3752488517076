"""Host-side helper for initialise(init_FG='kmeans'): K-means on the rows of a matrix with missing
values (code/models/kmeans/kmeans.py), vectorised with NumPy.  Runs once before sampling; not part
of the device hot path.  Semantics kept from the reference: centroids start uniformly between each
coordinate's observed min and max (same `random.uniform` call order, so `random.seed` reproduces the
reference's starting centroids), distance = mean squared difference over the coordinates both the
point and the centroid know (no overlap = infinitely far, ties go to the lowest index), centroid
coordinates without observed members are masked out, an empty cluster takes the point currently
furthest from its centroid ('singleton'), at most 200 iterations."""
import random

import numpy as np

max_iterations = 200


class KMeans(object):
    def __init__(self, X, M, K, resolve_empty='singleton'):
        self.X = np.array(X, dtype=float)
        self.M = np.array(M, dtype=float)
        self.K = K
        self.resolve_empty = resolve_empty
        assert len(self.X.shape) == 2, "Input matrix X is not a two-dimensional array, but instead %s-dimensional." % len(self.X.shape)
        assert self.X.shape == self.M.shape, "Input matrix X is not of the same size as the indicator matrix M: %s and %s respectively." % (self.X.shape, self.M.shape)
        assert self.K > 0, "K should be greater than 0."
        for i, c in enumerate(self.M.sum(axis=1)):
            assert c != 0, "Fully unobserved row in X, row %s." % i
        keep = self.M.sum(axis=0) > 0                    # unobserved columns do not influence the clustering
        self.X, self.M = self.X[:, keep], self.M[:, keep]
        (self.no_points, self.no_coordinates) = self.X.shape
        self.no_unique_points = len(set(tuple(l) for l in self.X.tolist()))
        self.distances = np.zeros(self.no_points)

    def initialise(self, seed=None):
        if seed is not None:
            random.seed(seed)
        big = np.where(self.M > 0, self.X, np.inf); small = np.where(self.M > 0, self.X, -np.inf)
        self.mins, self.maxs = big.min(axis=0), small.max(axis=0)
        self.centroids = np.array([[random.uniform(self.mins[j], self.maxs[j]) for j in range(self.no_coordinates)]
                                   for _ in range(self.K)])
        self.cluster_assignments = -np.ones(self.no_points, dtype=int)
        self.mask_centroids = np.ones((self.K, self.no_coordinates))

    def _distances_to_centroids(self):
        # sum_j M_dj Mc_cj (x_dj - c_cj)^2 = (M x^2) Mc^T - 2 (M x)(Mc c)^T + M (Mc c^2)^T
        Mx, Mc = self.M, self.mask_centroids
        Xm = self.X * Mx; Cm = self.centroids * Mc
        num = (Xm * self.X) @ Mc.T - 2.0 * Xm @ Cm.T + Mx @ (Cm * self.centroids).T
        overlap = Mx @ Mc.T
        with np.errstate(all='ignore'):
            d = np.where(overlap > 0, num / overlap, np.inf)
        return d

    def assignment(self):
        d = self._distances_to_centroids()
        new = d.argmin(axis=1)
        self.distances = d[np.arange(self.no_points), new]
        change = bool((new != self.cluster_assignments).any())
        self.cluster_assignments = new
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
                    self.centroids[c] = self.X[far]; self.mask_centroids[c] = self.M[far]
                    self.distances[far] = 0.0
                    self.cluster_assignments[far] = c
                    if old != c and old >= 0:
                        self._update_cluster(old)
                else:
                    self.centroids[c] = [random.uniform(self.mins[j], self.maxs[j]) for j in range(self.no_coordinates)]
                    self.mask_centroids[c] = np.ones(self.no_coordinates)
            return
        cnt = self.M[members].sum(axis=0)
        tot = (self.X[members] * self.M[members]).sum(axis=0)
        with np.errstate(all='ignore'):
            self.centroids[c] = np.where(cnt > 0, tot / np.maximum(cnt, 1), 0.0)
        self.mask_centroids[c] = (cnt > 0).astype(float)

    def cluster(self):
        iteration = 1
        change = True
        while change:
            iteration += 1
            change = self.assignment()
            self.update()
            if iteration >= max_iterations:
                break
        self.create_matrix()

    def create_matrix(self):
        self.clustering_results = np.zeros((self.no_points, self.K))
        self.clustering_results[np.arange(self.no_points), self.cluster_assignments] = 1.0
