"""code/cross_validation/matrix_cross_validation.py (class MatrixCrossValidation): K-fold cross-validation of a method
over a list of parameter settings; a fold is `method(X, train, **parameters).train(**train_config).predict(test)`."""
import json

import numpy

from . import mask

attempts_generate_M = 1000


class MatrixCrossValidation:
    def __init__(self, method, X, M, K, parameter_search, train_config, file_performance):
        self.method = method
        self.X = numpy.array(X, dtype=float)
        self.M = numpy.array(M)
        self.K = K
        self.train_config = train_config
        self.parameter_search = parameter_search
        self.fout = open(file_performance, 'w')
        (self.I, self.J) = self.X.shape
        assert (self.X.shape == self.M.shape), "X and M are of different shapes: %s and %s respectively." % (self.X.shape, self.M.shape)
        self.all_performances = {}
        self.average_performances = {}
        self.performances = {}

    def run(self):
        """:62-81."""
        for parameters in self.parameter_search:
            try:
                folds_test = mask.compute_folds_attempts(I=self.I, J=self.J, no_folds=self.K, attempts=attempts_generate_M, M=self.M)
                folds_training = mask.compute_Ms(folds_test)
                self.all_performances[self.JSON(parameters)] = {}
                for performance_dict in self.run_folds(folds_training, folds_test, parameters):
                    self.store_performances(performance_dict, parameters)
                self.log(parameters)
            except Exception as e:      # noqa: BLE001 -- the reference logs and carries on (:79-81)
                self.fout.write("Tried parameters %s but got exception: %s. \n" % (parameters, e))
                self.fout.flush()

    def run_folds(self, folds_training, folds_test, parameters):
        return [self.run_model(train, test, parameters) for train, test in zip(folds_training, folds_test)]

    def run_model(self, train, test, parameters):
        """:85-88."""
        model = self.method(self.X, train, **parameters)
        model.train(**self.train_config)
        return model.predict(test)

    def JSON(self, d):
        """:91-98."""
        d_copy = d.copy()
        for key, val in d.items():
            if isinstance(val, numpy.ndarray):
                d_copy[key] = val.tolist()
        return json.dumps(d_copy, sort_keys=True)

    def store_performances(self, performance_dict, parameters):
        """:101-106."""
        for name in performance_dict:
            self.all_performances[self.JSON(parameters)].setdefault(name, []).append(performance_dict[name])

    def compute_average_performances(self, parameters):
        """:109-119."""
        performances = self.all_performances[self.JSON(parameters)]
        average_performances = {name: (sum(values) / float(len(values))) for (name, values) in performances.items()}
        self.average_performances[self.JSON(parameters)] = average_performances
        for (name, avr_perf) in average_performances.items():
            self.performances.setdefault(name, []).append(avr_perf)

    def find_best_parameters(self, evaluation_criterion, low_better):
        """:122-131."""
        min_or_max = min if low_better else max
        self.best_performance = min_or_max(self.performances[evaluation_criterion])
        index_best = self.performances[evaluation_criterion].index(self.best_performance)
        self.best_parameters = self.parameter_search[index_best]
        self.best_performances_all = self.average_performances[self.JSON(self.best_parameters)]
        self.log_best(index_best)
        return (self.best_parameters, self.best_performance)

    def log(self, parameters):
        """:134-138."""
        self.compute_average_performances(parameters)
        message = "Tried parameters %s. Average performances: %s. \nAll performances: %s. \n" % (
            parameters, self.average_performances[self.JSON(parameters)], self.all_performances[self.JSON(parameters)])
        self.fout.write(message)
        self.fout.flush()

    def log_best(self, index_best):
        """:141-148."""
        message = "Best performances: %s. Best parameters: %s. \n" % (self.best_performances_all, self.best_parameters)
        self.fout.write(message)
        self.fout.flush()
