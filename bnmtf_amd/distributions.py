"""Drop-ins for code/models/distributions/{truncated_normal_vector,truncated_normal,
gamma,exponential}.py.  Draws and TN moments run on the device through the stand-alone
C-ABI hooks; the closed-form Gamma moments are scalar host arithmetic as in the reference."""
import ctypes as C
import math

import numpy as np

from . import _lib

_counter = [0]


def _seed(seed):
    return int(np.random.randint(0, 2 ** 62)) if seed is None else int(seed)


def TN_vector_draw(mus, taus, seed=None, it=0, col=0, elem0=0, device=0):
    """truncated_normal_vector.py:37-50 -> Python list of draws (tau == 0 -> 0)."""
    mu = _lib.f64(mus); tau = _lib.f64(taus)
    out = np.zeros(mu.shape[0])
    _lib.check(_lib.lib().bnmtf_tn_sample(_lib.ptr(mu), _lib.ptr(tau), mu.shape[0], _seed(seed), int(it), int(col),
                                           int(elem0), int(device), _lib.ptr(out)))
    return list(out)


def TN_draw(mu, tau, seed=None, it=0, col=0, device=0):
    """truncated_normal.py:37-44."""
    return TN_vector_draw([mu], [tau], seed, it, col, 0, device)[0]


def _moments(mus, taus, device=0):
    mu = _lib.f64(mus); tau = _lib.f64(taus)
    e = np.zeros(mu.shape[0]); v = np.zeros(mu.shape[0])
    _lib.check(_lib.lib().bnmtf_tn_moments(_lib.ptr(mu), _lib.ptr(tau), mu.shape[0], int(device), _lib.ptr(e), _lib.ptr(v)))
    return e, v


def TN_vector_expectation(mus, taus):
    """truncated_normal_vector.py:53-61."""
    return list(_moments(mus, taus)[0])


def TN_vector_variance(mus, taus):
    """truncated_normal_vector.py:64-73."""
    return list(_moments(mus, taus)[1])


def TN_vector_mode(mus):
    """truncated_normal_vector.py:76-78."""
    return np.maximum(np.zeros(len(mus)), mus)


def TN_expectation(mu, tau):
    return _moments([mu], [tau])[0][0]


def TN_variance(mu, tau):
    return _moments([mu], [tau])[1][0]


def TN_mode(mu):
    return max(0.0, mu)


def gamma_draw(alpha, beta, seed=None, it=0, device=0):
    """gamma.py:11-14 (shape alpha, scale 1/beta)."""
    out = C.c_double()
    _lib.check(_lib.lib().bnmtf_gamma_sample(float(alpha), float(beta), _seed(seed), int(it), int(device), C.byref(out)))
    return out.value


def gamma_expectation(alpha, beta):
    """gamma.py:17-19."""
    return float(alpha) / float(beta)


def gamma_expectation_log(alpha, beta):
    """gamma.py:22-24."""
    from scipy.special import psi
    return float(psi(float(alpha))) - math.log(float(beta))


def gamma_mode(alpha, beta):
    """gamma.py:27-29."""
    return (float(alpha) - 1) / float(beta)


def exponential_draw(lambdax):
    """exponential.py:7-9 (host: initialisation only)."""
    return np.random.exponential(scale=1.0 / lambdax, size=None)
