"""initialise(init_FG='kmeans'): K-means on the rows of a matrix with missing values (code/models/kmeans/kmeans.py).

The two O(points x coordinates x K) passes of an iteration -- the assignment distances and the per-cluster sums -- run
on the GPU in fp64 (bnmtf_kmeans_* of libbnmtf_hip.so); the O(K x coordinates) centroid division, the masks and the
empty-cluster rule are host code.  There is no CPU compute path here: without the library or a GPU the class raises
(the NumPy restatement used to check it lives in oracle/kmeans_oracle.py, test infrastructure).

Semantics kept from the reference, all pinned by tests/golden/kmeans.npz (reference runs under `random.seed`):
centroids start uniformly between each coordinate's observed min and max (same `random.uniform` call order);
distance = mean squared difference over the coordinates both the point and the centroid know (no overlap = infinitely
far, the first centroid / the first defined one / a smaller defined one wins, kmeans.py:104-113); centroid coordinates
without observed members are masked out; an empty cluster takes the point currently furthest from its centroid
('singleton', :137-152) -- and, as written in the reference, that centroid then IS the row of X (`self.centroids[c] =
self.X[index]` is a NumPy view), so later means written into the centroid overwrite the data point; at most 200
iterations."""
import ctypes as C
import random

import numpy as np

from . import _lib

max_iterations = 200


class KMeans(object):
    def __init__(self, X, M, K, resolve_empty='singleton', *, device=0):
        self._device = 0 if device is None else int(device)
        self._dh = None
        self.X = np.array(X, dtype=float)
        self.M = np.array(M, dtype=float)
        self.K = K
        self.resolve_empty = resolve_empty
        assert len(self.X.shape) == 2, "Input matrix X is not a two-dimensional array, but instead %s-dimensional." % len(self.X.shape)
        assert self.X.shape == self.M.shape, "Input matrix X is not of the same size as the indicator matrix M: %s and %s respectively." % (self.X.shape, self.M.shape)
        assert self.K > 0, "K should be greater than 0."
        self._X_as_given = self.X                        # (no_unique_points: counted when an empty cluster first asks -- a model search builds two of these per model)
        for i, c in enumerate(self.M.sum(axis=1)):
            assert c != 0, "Fully unobserved row in X, row %s." % i
        keep = self.M.sum(axis=0) > 0                    # unobserved columns do not influence the clustering
        self.X, self.M = np.ascontiguousarray(self.X[:, keep]), np.ascontiguousarray(self.M[:, keep])
        (self.no_points, self.no_coordinates) = self.X.shape
        self.distances = np.zeros(self.no_points)

    @property
    def no_unique_points(self):
        """kmeans.py:47 (the reference counts in its constructor; here: the first time the empty-cluster rule needs the number)."""
        if getattr(self, "_no_unique_points", None) is None:
            self._no_unique_points = len(set(tuple(l) for l in self._X_as_given.tolist()))
        return self._no_unique_points

    def initialise(self, seed=None):
        if seed is not None:
            random.seed(seed)
        big = np.where(self.M > 0, self.X, np.inf); small = np.where(self.M > 0, self.X, -np.inf)
        self.mins, self.maxs = big.min(axis=0), small.max(axis=0)
        # a list, as in the reference: an entry may become a view of a row of X (see the header)
        self.centroids = [np.array([random.uniform(self.mins[j], self.maxs[j]) for j in range(self.no_coordinates)]) for _ in range(self.K)]
        self.cluster_assignments = -np.ones(self.no_points, dtype=int)
        self.mask_centroids = np.ones((self.K, self.no_coordinates))
        self._alias = [None] * self.K                    # centroid c is the row _alias[c] of X

    # ---- device side
    def _device_handle(self):
        if self._dh is None:
            self._M8 = np.ascontiguousarray(self.M != 0, dtype=np.uint8)
            h = C.c_void_p()
            _lib.check(_lib.lib().bnmtf_kmeans_create(_lib.ptr(self.X), _lib.ptr(self._M8), self.no_points, self.no_coordinates,
                                                      int(self.K), self._device, C.byref(h)))
            self._dh = h
        return self._dh

    def close(self):
        if getattr(self, "_dh", None) is not None:
            _lib.lib().bnmtf_kmeans_destroy(self._dh)
            self._dh = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _device_sums(self):
        cnt = np.zeros((self.K, self.no_coordinates)); tot = np.zeros((self.K, self.no_coordinates))
        a32 = np.ascontiguousarray(self.cluster_assignments, dtype=np.int32)
        _lib.check(_lib.lib().bnmtf_kmeans_sums(self._device_handle(), _lib.ptr(a32), _lib.ptr(cnt), _lib.ptr(tot)))
        self._sums = (cnt, tot)

    # ---- the reference's loop
    def assignment(self):
        new = np.zeros(self.no_points, dtype=np.int32); dist = np.zeros(self.no_points)
        Cm = np.ascontiguousarray(np.array([np.asarray(c, dtype=float) for c in self.centroids]))
        Mc8 = np.ascontiguousarray(self.mask_centroids != 0, dtype=np.uint8)
        _lib.check(_lib.lib().bnmtf_kmeans_assign(self._device_handle(), _lib.ptr(Cm), _lib.ptr(Mc8), _lib.ptr(new), _lib.ptr(dist)))
        new = new.astype(int)
        self.distances = dist                            # +inf where the reference stores None (no shared coordinate)
        change = bool((new != self.cluster_assignments).any())
        self.cluster_assignments = new
        return change

    def update(self):
        self._device_sums()
        for c in range(self.K):
            self._update_cluster(c)

    def _update_cluster(self, c):
        if not (self.cluster_assignments == c).any():
            if self.no_unique_points >= self.K:
                if self.resolve_empty == 'singleton':
                    far = int(np.argmax(self.distances))
                    old = int(self.cluster_assignments[far])
                    self.centroids[c] = self.X[far]; self._alias[c] = far        # a view of the row: kmeans.py:141
                    self.mask_centroids[c] = self.M[far]
                    self.distances[far] = 0.0
                    self.cluster_assignments[far] = c
                    self._device_sums()                   # memberships changed: the sums of the clusters still to come
                    self._update_cluster(old)
                else:
                    self.centroids[c] = np.array([random.uniform(self.mins[j], self.maxs[j]) for j in range(self.no_coordinates)])
                    self._alias[c] = None
                    self.mask_centroids[c] = np.ones(self.no_coordinates)
            return
        if self._sums is None:
            self._device_sums()
        cnt, tot = self._sums[0][c], self._sums[1][c]
        with np.errstate(all='ignore'):
            self.centroids[c][:] = np.where(cnt > 0, tot / np.maximum(cnt, 1), 0.0)
        self.mask_centroids[c] = (cnt > 0).astype(float)
        if self._alias[c] is not None:                    # the mean went into the data point itself: the device copy follows,
            _lib.check(_lib.lib().bnmtf_kmeans_set_row(self._device_handle(), int(self._alias[c]), _lib.ptr(np.ascontiguousarray(self.X[self._alias[c]]))))
            self._sums = None                             # and the point may by now belong to a cluster that is still to be updated

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
        self.create_matrix()

    def create_matrix(self):
        self.clustering_results = np.zeros((self.no_points, self.K))
        self.clustering_results[np.arange(self.no_points), self.cluster_assignments] = 1.0
