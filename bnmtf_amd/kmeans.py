"""initialise(init_FG='kmeans'): K-means on the rows of a matrix with missing values (code/models/kmeans/kmeans.py).
The two O(points x coordinates x K) passes of an iteration -- assignment distances and per-cluster sums -- run on the
GPU (KMeans(..., device=<ordinal>), bnmtf_kmeans_* in libbnmtf_hip.so; what the model classes use) or vectorised in
NumPy (device=None); the O(K x coordinates) centroid division and the empty-cluster rule are host code either way.
Semantics kept from the reference: centroids start uniformly between each
coordinate's observed min and max (same `random.uniform` call order, so `random.seed` reproduces the
reference's starting centroids), distance = mean squared difference over the coordinates both the
point and the centroid know (no overlap = infinitely far, ties go to the lowest index), centroid
coordinates without observed members are masked out, an empty cluster takes the point currently
furthest from its centroid ('singleton'), at most 200 iterations."""
import random

import numpy as np

max_iterations = 200


class KMeans(object):
    def __init__(self, X, M, K, resolve_empty='singleton', *, device=None):
        """device=None: NumPy on the host; device=<GPU ordinal>: the two O(points x coordinates x K) passes of an iteration
        (assignment distances, per-cluster sums) run in libbnmtf_hip.so (bnmtf_kmeans_*), the O(K x coordinates) rest here."""
        self._device = device
        self._dh = None
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

    _sums = None

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

    def _device_handle(self):
        if self._dh is None:
            import ctypes as C
            from . import _lib
            self._keep = (np.ascontiguousarray(self.X, dtype=np.float32), np.ascontiguousarray(self.M != 0, dtype=np.uint8))
            h = C.c_void_p()
            _lib.check(_lib.lib().bnmtf_kmeans_create(_lib.ptr(self._keep[0]), _lib.ptr(self._keep[1]), self.no_points, self.no_coordinates,
                                                      int(self.K), int(self._device), C.byref(h)))
            self._dh = h
        return self._dh

    def close(self):
        if getattr(self, "_dh", None) is not None:
            from . import _lib
            _lib.lib().bnmtf_kmeans_destroy(self._dh)
            self._dh = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def assignment(self):
        if self._device is not None:
            from . import _lib
            new = np.zeros(self.no_points, dtype=np.int32); dist = np.zeros(self.no_points)
            C32 = np.ascontiguousarray(self.centroids, dtype=np.float32); Mc8 = np.ascontiguousarray(self.mask_centroids != 0, dtype=np.uint8)
            _lib.check(_lib.lib().bnmtf_kmeans_assign(self._device_handle(), _lib.ptr(C32), _lib.ptr(Mc8), _lib.ptr(new), _lib.ptr(dist)))
            new = new.astype(int)
            self.distances = dist
            change = bool((new != self.cluster_assignments).any())
            self.cluster_assignments = new
            return change
        d = self._distances_to_centroids()
        new = d.argmin(axis=1)
        new[~np.isfinite(d).any(axis=1)] = self.K - 1     # kmeans.py:107-113: a point that overlaps no centroid ends in the last cluster
        self.distances = d[np.arange(self.no_points), new]
        change = bool((new != self.cluster_assignments).any())
        self.cluster_assignments = new
        return change

    def update(self):
        self._sums = None
        if self._device is not None and self.K <= 40:
            from . import _lib
            cnt = np.zeros((self.K, self.no_coordinates)); tot = np.zeros((self.K, self.no_coordinates))
            a32 = np.ascontiguousarray(self.cluster_assignments, dtype=np.int32)
            _lib.check(_lib.lib().bnmtf_kmeans_sums(self._device_handle(), _lib.ptr(a32), _lib.ptr(cnt), _lib.ptr(tot)))
            self._sums = (cnt, tot, self.cluster_assignments.copy())
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
        if self._sums is not None and np.array_equal(self._sums[2] == c, self.cluster_assignments == c):
            cnt, tot = self._sums[0][c], self._sums[1][c]      # from the device pass (membership unchanged by an empty-cluster move)
        else:
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
