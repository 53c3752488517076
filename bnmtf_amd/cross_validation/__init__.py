"""Model selection and cross-validation drivers of code/cross_validation/ on device replicas: every candidate
(fold x rank x restart) is an independent model, so they are spread over the visible GPUs -- one worker process per
GPU, each building its own handles -- and a candidate's posterior means are accumulated on its device
(run(..., expectation=(burn_in, thinning), store_samples=False)): no iterations x I x K sample array is ever built.

Same class names, constructor arguments, methods and log-file lines as the reference
(line_search_bnmf.LineSearch, grid_search_bnmtf.GridSearch, greedy_search_bnmtf.GreedySearch,
line_search_cross_validation.LineSearchCrossValidation, greedy_search_cross_validation.GreedySearchCrossValidation,
matrix_cross_validation.MatrixCrossValidation,
parallel_matrix_cross_validation.ParallelMatrixCrossValidation,
nested_matrix_cross_validation.MatrixNestedCrossValidation, mask.*); build-only extras are keyword-only
(`pool=`, a ReplicaPool)."""
from . import mask
from .replicas import ReplicaPool
from .line_search_bnmf import LineSearch
from .grid_search_bnmtf import GridSearch
from .greedy_search_bnmtf import GreedySearch
from .line_search_cross_validation import LineSearchCrossValidation
from .greedy_search_cross_validation import GreedySearchCrossValidation
from .matrix_cross_validation import MatrixCrossValidation
from .parallel_matrix_cross_validation import ParallelMatrixCrossValidation
from .nested_matrix_cross_validation import MatrixNestedCrossValidation

__all__ = ["mask", "ReplicaPool", "LineSearch", "GridSearch", "GreedySearch", "LineSearchCrossValidation", "GreedySearchCrossValidation",
           "MatrixCrossValidation", "ParallelMatrixCrossValidation", "MatrixNestedCrossValidation"]
