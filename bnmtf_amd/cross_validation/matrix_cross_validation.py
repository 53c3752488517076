"""K-fold cross-validation of a method over a list of parameter settings, as one batch of device jobs.

Mirrors the contract of the reference's MatrixCrossValidation (code/cross_validation/matrix_cross_validation.py:47-148:
constructor arguments, `run()`, `find_best_parameters(criterion, low_better)`, the attributes `all_performances`,
`average_performances`, `performances`, `best_*`, and the three kinds of log line) -- but the work is organised around
what is new here: a fold is an independent model, so `run()` first PLANS every (parameter setting, fold) pair -- drawing
each setting's folds from Python's `random` stream in the order the reference draws them -- then EXECUTES the whole plan
as one job list on a ReplicaPool (all settings and folds side by side on the GPU slots, not one setting after the
other), and only then RECORDS the results setting by setting, writing the reference's log lines in its order.  A setting
whose folds cannot be drawn, or any of whose models raises, is logged with the reference's "Tried parameters ... but got
exception" line (:79-81) and skipped, like there.

`ParallelMatrixCrossValidation` (parallel_matrix_cross_validation.py) is this class with P slots.
"""
import json

import numpy

from . import mask
from .replicas import ReplicaError, ReplicaPool, _accepts

attempts_generate_M = 1000


def fold_job(job, shared):
    """One fold on the worker's GPU: `method(X, train, **parameters).train(**train_config).predict(test)` (:85-88)."""
    method = job["method"]
    kw = dict(job["parameters"])
    if job.get("seed") is not None:              # (an extension: the reference's folds draw from the process's global streams, :85-88)
        import random
        numpy.random.seed(job["seed"] % (2 ** 32)); random.seed(job["seed"])
        if _accepts(method.__init__, "seed"):
            kw.setdefault("seed", job["seed"])
    if _accepts(method.__init__, "device"):
        kw["device"] = job.get("device", 0)
    if _accepts(method.__init__, "verbose"):
        kw.setdefault("verbose", False)
    model = method(shared["X"], job["train"], **kw)
    try:
        model.train(**job["train_config"])
        return model.predict(job["test"])
    finally:
        if hasattr(model, "close"):
            model.close()


class _Setting(object):
    """One entry of parameter_search in the plan: its key, its folds (or why it has none), later its fold results."""
    def __init__(self, parameters, key):
        self.parameters, self.key = parameters, key
        self.folds, self.error, self.results = [], None, None


class MatrixCrossValidation(object):
    def __init__(self, method, X, M, K, parameter_search, train_config, file_performance, *, devices=None, seed=None):
        """seed (not in the reference): every fold job seeds NumPy's and Python's global streams -- and the model's sampler -- with
        seed + its index before it builds its model; the folds run in worker processes, whose streams are otherwise their own."""
        self.method = method
        self.seed = seed
        self.X = numpy.array(X, dtype=float)
        self.M = numpy.array(M)
        self.K = K
        self.train_config = train_config
        self.parameter_search = parameter_search
        self.fout = open(file_performance, 'w')
        (self.I, self.J) = self.X.shape
        assert (self.X.shape == self.M.shape), "X and M are of different shapes: %s and %s respectively." % (self.X.shape, self.M.shape)
        self.devices = [0] if devices is None else list(devices)      # replica slots (ParallelMatrixCrossValidation: P of them)
        self.all_performances = {}          # JSON(parameters) -> {measure: [value per fold]}
        self.average_performances = {}      # JSON(parameters) -> {measure: mean over the folds}
        self.performances = {}              # measure -> [mean, one per recorded setting, in parameter_search order]

    # ------------------------------------------------------------------ plan -> execute -> record
    def run(self):
        plan = self._plan()
        self._execute(plan)
        for setting in plan:
            self._record(setting)

    def _plan(self):
        plan = []
        for parameters in self.parameter_search:
            setting = _Setting(parameters, self.JSON(parameters))
            try:
                tests = mask.compute_folds_attempts(I=self.I, J=self.J, no_folds=self.K, attempts=attempts_generate_M, M=self.M)
                setting.folds = list(zip(mask.compute_Ms(tests), tests))
            except Exception as e:      # noqa: BLE001 -- reported by _record in the reference's words
                setting.error = e
            plan.append(setting)
        return plan

    def _execute(self, plan):
        jobs, owner = [], []
        for si, setting in enumerate(plan):
            for train, test in setting.folds:
                jobs.append(dict(method=self.method, parameters=setting.parameters, train=train, test=test, train_config=self.train_config,
                                 seed=None if self.seed is None else self.seed + 7919 * len(jobs)))
                owner.append(si)
        if not jobs:
            return
        with ReplicaPool(devices=self.devices, shared={"X": self.X}) as pool:
            out = pool.map(fold_job, jobs, errors="return")
        for setting in plan:
            setting.results = []
        for si, res in zip(owner, out):
            plan[si].results.append(res)

    def _record(self, setting):
        failed = setting.error
        if failed is None:
            failed = next((r for r in setting.results or [] if isinstance(r, ReplicaError)), None)
        if failed is not None:
            text = failed.first_line.split(": ", 1)[-1] if isinstance(failed, ReplicaError) else failed
            self.fout.write("Tried parameters %s but got exception: %s. \n" % (setting.parameters, text))
            self.fout.flush()
            return
        self.all_performances[setting.key] = {}
        for fold_result in setting.results:
            self.store_performances(fold_result, setting.parameters)
        self.log(setting.parameters)

    def run_model(self, train, test, parameters):
        """One fold in this process on the first slot (the reference's per-fold entry point)."""
        return fold_job(dict(method=self.method, parameters=parameters, train=train, test=test, train_config=self.train_config, device=self.devices[0]),
                        {"X": self.X})

    # ------------------------------------------------------------------ bookkeeping (names and formats are the contract)
    def JSON(self, d):
        return json.dumps({k: (v.tolist() if isinstance(v, numpy.ndarray) else v) for k, v in d.items()}, sort_keys=True)

    def store_performances(self, performance_dict, parameters):
        per_measure = self.all_performances[self.JSON(parameters)]
        for measure, value in performance_dict.items():
            per_measure.setdefault(measure, []).append(value)

    def compute_average_performances(self, parameters):
        key = self.JSON(parameters)
        means = {measure: sum(vals) / float(len(vals)) for measure, vals in self.all_performances[key].items()}
        self.average_performances[key] = means
        for measure, mean in means.items():
            self.performances.setdefault(measure, []).append(mean)

    def find_best_parameters(self, evaluation_criterion, low_better):
        scores = self.performances[evaluation_criterion]
        self.best_performance = min(scores) if low_better else max(scores)
        position = scores.index(self.best_performance)
        self.best_parameters = self.parameter_search[position]
        self.best_performances_all = self.average_performances[self.JSON(self.best_parameters)]
        self.log_best(position)
        return (self.best_parameters, self.best_performance)

    def log(self, parameters):
        self.compute_average_performances(parameters)
        key = self.JSON(parameters)
        self.fout.write("Tried parameters %s. Average performances: %s. \nAll performances: %s. \n" % (
            parameters, self.average_performances[key], self.all_performances[key]))
        self.fout.flush()

    def log_best(self, index_best):
        self.fout.write("Best performances: %s. Best parameters: %s. \n" % (self.best_performances_all, self.best_parameters))
        self.fout.flush()
