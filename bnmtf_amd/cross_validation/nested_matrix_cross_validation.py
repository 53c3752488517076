"""Nested cross-validation of a matrix-prediction method: an outer K-fold split measures the method, and inside every outer
training set an inner K-fold cross-validation over `parameter_search` picks the parameters that outer fold is fitted with.

The contract of the reference's MatrixNestedCrossValidation (code/cross_validation/nested_matrix_cross_validation.py:55-133:
constructor arguments, `run()`, `run_model`, `store_performances`, `compute_average_performances`, `log`, the attributes
`all_performances` / `average_performances`, one inner log file per outer fold, the closing "Average performances" line).  The
inner cross-validations are ParallelMatrixCrossValidation objects of this package -- each a single batch of device jobs on P
replica slots -- and the K outer models are fitted as one more batch after all inner searches have named their parameters."""
import numpy

from . import mask
from .matrix_cross_validation import fold_job
from .parallel_matrix_cross_validation import ParallelMatrixCrossValidation
from .replicas import ReplicaPool

attempts_generate_M = 1000


class MatrixNestedCrossValidation(object):
    def __init__(self, method, X, M, K, P, parameter_search, train_config, file_performance, files_nested_performances, *, devices=None):
        self.method = method
        self.X = numpy.array(X, dtype=float)
        self.M = numpy.array(M)
        self.K = K
        self.P = P
        self.train_config = train_config
        self.parameter_search = parameter_search
        self.files_nested_performances = files_nested_performances
        self.fout = open(file_performance, 'w')
        (self.I, self.J) = self.X.shape
        assert (self.X.shape == self.M.shape), "X and M are of different shapes: %s and %s respectively." % (self.X.shape, self.M.shape)
        self.devices = devices
        self.all_performances = {}      # criterion -> list over the outer folds
        self.average_performances = {}  # criterion -> mean over the outer folds

    def run(self):
        """:78-112.  Outer folds drawn first (the reference's draw order: outer split, then each inner cross-validation's own
        splits as it runs), every inner search in turn, then the K outer models side by side.
        Deviation from the reference's loop, on purpose: it fits a fold's outer model inside the fold loop (run_model, :104); here
        the K outer fits wait for all inner searches and run as ONE batch on the replica slots (they are independent once their
        parameters are chosen) -- so a model that draws from the GLOBAL NumPy stream sees the inner searches' draws first.  A subclass
        that overrides run_model() is honoured: its fits then run fold by fold through it, after the searches."""
        folds_test = mask.compute_folds_attempts(I=self.I, J=self.J, no_folds=self.K, attempts=attempts_generate_M, M=self.M)
        folds_training = mask.compute_Ms(folds_test)
        chosen = []
        devices = self.devices
        for i, (train, test) in enumerate(zip(folds_training, folds_test)):
            print("Fold %s of nested cross-validation." % (i + 1))
            crossval = ParallelMatrixCrossValidation(method=self.method, X=self.X, M=train, K=self.K, parameter_search=self.parameter_search,
                                                     train_config=self.train_config, file_performance=self.files_nested_performances[i],
                                                     P=self.P, devices=self.devices)
            crossval.run()
            try:
                (best_parameters, _) = crossval.find_best_parameters(evaluation_criterion='MSE', low_better=True)
                print("Best parameters for fold %s were %s." % (i + 1, best_parameters))
            except KeyError:
                best_parameters = self.parameter_search[0]
                print("Found no performances, dataset too sparse? Use first values instead for fold %s, %s." % (i + 1, best_parameters))
            chosen.append(best_parameters)
            if devices is None:
                devices = crossval.devices                       # (what the slots resolved to: once, from the first inner run)
        # the outer models: independent of one another once their parameters are known -- one batch
        jobs = [dict(method=self.method, parameters=p, train=train, test=test, train_config=self.train_config)
                for p, train, test in zip(chosen, folds_training, folds_test)]
        if type(self).run_model is not MatrixNestedCrossValidation.run_model:
            results = [self.run_model(train, test, p) for p, train, test in zip(chosen, folds_training, folds_test)]
        else:
            with ReplicaPool(devices=devices, shared={"X": self.X}) as pool:
                results = pool.map(fold_job, jobs)
        for i, performance_dict in enumerate(results):
            self.store_performances(performance_dict)
            print("Finished fold %s, with performances %s." % (i + 1, performance_dict))
        self.log()

    def run_model(self, train, test, parameters):
        """:116-119."""
        return fold_job(dict(method=self.method, parameters=parameters, train=train, test=test, train_config=self.train_config,
                             device=(self.devices or [0])[0]), {"X": self.X})

    def store_performances(self, performance_dict):
        """:122-127."""
        for name in performance_dict:
            self.all_performances.setdefault(name, []).append(performance_dict[name])

    def compute_average_performances(self):
        """:130-133."""
        self.average_performances = {name: (sum(values) / float(len(values))) for (name, values) in self.all_performances.items()}

    def log(self):
        """:136-140."""
        self.compute_average_performances()
        self.fout.write("Average performances: %s. \nAll performances: %s. \n" % (self.average_performances, self.all_performances))
        self.fout.flush()
        self.fout.close()
