"""code/cross_validation/parallel_matrix_cross_validation.py (class ParallelMatrixCrossValidation): in the reference the
folds of a setting run on `multiprocessing.Pool(P)` CPU workers (:52-65); here P is the number of replica slots over the
visible GPUs (devices i mod n_gpus, one worker process per slot) that MatrixCrossValidation's job batch is dealt to."""
from .matrix_cross_validation import MatrixCrossValidation


class ParallelMatrixCrossValidation(MatrixCrossValidation):
    def __init__(self, method, X, M, K, parameter_search, train_config, file_performance, P, *, devices=None, seed=None):
        if devices is None:
            from .replicas import visible_devices
            n = max(visible_devices(), 1)
            devices = [p % n for p in range(P)]
        MatrixCrossValidation.__init__(self, method, X, M, K, parameter_search, train_config, file_performance, devices=devices, seed=seed)
        self.P = P
