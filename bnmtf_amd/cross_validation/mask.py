"""code/cross_validation/mask.py with the interpreted O(I*J) loops vectorised.  The functions that draw (generate_M,
generate_M_from_M, compute_folds) consume Python's `random` stream exactly as the reference does -- random.sample /
random.shuffle on a list of the same length -- so `random.seed(s)` gives the reference's masks and folds."""
import random

import numpy as np


def generate_M(I, J, fraction):
    """mask.py:9-14."""
    M = np.ones([I, J])
    values = random.sample(range(0, I * J), int(I * J * fraction))
    M.flat[np.asarray(values, dtype=np.int64)] = 0
    return M


def nonzero_indices(M):
    """mask.py:147-149: row-major list of (i, j) with M[i][j] != 0."""
    ii, jj = np.nonzero(np.asarray(M))
    return list(zip(ii.tolist(), jj.tolist()))


def _shuffled_flat_indices(M):
    """random.shuffle of the row-major non-zero positions: the permutation only depends on the list's length, so
    shuffling flat indices draws what the reference's shuffle of (i, j) tuples draws."""
    flat = np.flatnonzero(np.asarray(M)).tolist()
    random.shuffle(flat)
    return np.asarray(flat, dtype=np.int64)


def generate_M_from_M(M, fraction):
    """mask.py:18-37."""
    M = np.asarray(M)
    I, J = M.shape
    no_elements = int(np.count_nonzero(M))
    no_missing_total = I * J - no_elements
    assert no_missing_total < I * J * fraction, "Specified %s fraction missing, so %s entries missing, but there are already %s missing by default!" % \
        (fraction, I * J * fraction, no_missing_total)
    flat = _shuffled_flat_indices(M)
    index_last_observed = int(I * J * (1 - fraction))
    M_train, M_test = np.zeros((I, J)), np.zeros((I, J))
    M_train.flat[flat[:index_last_observed]] = 1
    M_test.flat[flat[index_last_observed:]] = 1
    assert np.array_equal(M, M_train + M_test), "Tried splitting M into M_test and M_train but something went wrong."
    return M_train, M_test


def try_generate_M_from_M(M, fraction, attempts):
    """mask.py:39-44."""
    for i in range(0, attempts):
        M_train, M_test = generate_M_from_M(M, fraction)
        if check_empty_rows_columns(M_train):
            return M_train, M_test
    assert False, "Failed to generate folds for training and test data, %s attempts, fraction %s." % (attempts, fraction)


def compute_folds(I, J, no_folds, M=None):
    """mask.py:48-68."""
    M = np.ones((I, J)) if M is None else np.array(M)
    flat = _shuffled_flat_indices(M)
    no_elements = len(flat)
    split_places = [int(i * no_elements / no_folds) for i in range(0, no_folds + 1)]
    folds_M = []
    for f in range(no_folds):
        Mf = np.zeros((I, J))
        Mf.flat[flat[split_places[f]:split_places[f + 1]]] = 1
        folds_M.append(Mf)
    return folds_M


def compute_folds_attempts(I, J, no_folds, attempts, M=None):
    """mask.py:71-82."""
    Mfull = np.ones((I, J)) if M is None else np.asarray(M)
    for i in range(0, attempts):
        folds_M = compute_folds(I=I, J=J, no_folds=no_folds, M=M)
        if all(check_empty_rows_columns(Mfull - M_test) for M_test in folds_M):
            return folds_M
    assert False, "Failed to generate folds for training and test data, %s attempts." % attempts


def compute_crossval_folds_rows_attempts(M, no_rows, no_folds, attempts):
    """mask.py:87-105."""
    M = np.asarray(M)
    I, J = M.shape
    M_rows, M_rest = M[:no_rows], M[no_rows:]
    out = []
    for test_rows in compute_folds_attempts(no_rows, J, no_folds, attempts, M_rows):
        out.append((np.concatenate((M_rows - test_rows, M_rest), axis=0),
                    np.concatenate((test_rows, np.zeros((I - no_rows, J))), axis=0)))
    return out


def compute_crossval_folds_columns_attempts(M, no_columns, no_folds, attempts):
    """mask.py:107-125."""
    M = np.asarray(M)
    I, J = M.shape
    M_cols, M_rest = M[:, :no_columns], M[:, no_columns:]
    out = []
    for test_cols in compute_folds_attempts(I, no_columns, no_folds, attempts, M_cols):
        out.append((np.concatenate((M_cols - test_cols, M_rest), axis=1),
                    np.concatenate((test_cols, np.zeros((I, J - no_columns))), axis=1)))
    return out


def check_empty_rows_columns(M):
    """mask.py:128-139: True if every row and column has an observation."""
    M = np.asarray(M)
    return bool((M.sum(axis=0) != 0).all() and (M.sum(axis=1) != 0).all())


def compute_Ms(folds_M):
    """mask.py:142-145: the training mask of fold f = the sum of the other folds."""
    folds_M = [np.array(f) for f in folds_M]
    total = sum(folds_M)
    return [total - f for f in folds_M]


def calc_inverse_M(M):
    """mask.py:147-154."""
    return np.where(np.asarray(M) == 1, 0.0, 1.0)


def nonzero_row_indices(M):
    """mask.py:160-162."""
    return [np.flatnonzero(row).tolist() for row in np.asarray(M)]


def nonzero_column_indices(M):
    """mask.py:164-167."""
    return [np.flatnonzero(col).tolist() for col in np.asarray(M).T]


def recover_predictions(M, X_true, X_pred):
    """mask.py:170-177 (as written: the pairs at the ZERO entries of M)."""
    M = np.asarray(M)
    ii, jj = np.nonzero(M == 0)
    Xt, Xp = np.asarray(X_true), np.asarray(X_pred)
    return [(Xt[i][j], Xp[i][j]) for i, j in zip(ii.tolist(), jj.tolist())]
