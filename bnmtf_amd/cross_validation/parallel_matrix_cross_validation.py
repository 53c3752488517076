"""code/cross_validation/parallel_matrix_cross_validation.py (class ParallelMatrixCrossValidation): the folds of a
parameter setting run side by side -- in the reference on `multiprocessing.Pool(P)` CPU workers (:52-65), here on P
replica slots over the visible GPUs (devices i mod n_gpus), one worker process per slot."""
import numpy

from .matrix_cross_validation import MatrixCrossValidation
from .replicas import ReplicaPool, _accepts


def run_fold(job, shared):
    """:22-33 (one fold: construct, train, predict), on the worker's GPU."""
    method = job["method"]
    kw = dict(job["parameters"])
    if _accepts(method.__init__, "device"):
        kw["device"] = job.get("device", 0)
    if _accepts(method.__init__, "verbose"):
        kw.setdefault("verbose", False)
    model = method(shared["X"] if job.get("X") is None else job["X"], job["train"], **kw)
    model.train(**job["train_config"])
    out = model.predict(job["test"])
    if hasattr(model, "close"):
        model.close()
    return out


class ParallelMatrixCrossValidation(MatrixCrossValidation):
    def __init__(self, method, X, M, K, parameter_search, train_config, file_performance, P, *, devices=None):
        MatrixCrossValidation.__init__(self, method, X, M, K, parameter_search, train_config, file_performance)
        self.P = P
        self.devices = devices

    def run_folds(self, folds_training, folds_test, parameters):
        devices = self.devices
        if devices is None:
            from .replicas import visible_devices
            n = max(visible_devices(), 1)
            devices = [p % n for p in range(self.P)]
        with ReplicaPool(devices=devices, shared={"X": numpy.copy(self.X)}) as pool:
            jobs = [dict(method=self.method, parameters=parameters, train=train, test=test, train_config=self.train_config)
                    for train, test in zip(folds_training, folds_test)]
            return pool.map(run_fold, jobs)
