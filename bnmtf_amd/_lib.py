"""ctypes binding of libbnmtf_hip.so (include/bnmtf_hip.h).  No PyTorch, no fallback:
if the HIP library is missing or a call fails, an exception is raised."""
import ctypes as C
import atexit
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BNMTF_LIB: load another build of the library (tools/: the phase-timing or an A/B build) without touching the shipped one
LIB_PATH = os.environ.get("BNMTF_LIB") or os.path.join(_HERE, "lib", "libbnmtf_hip.so")

KERNEL_GEMM_ROWS, KERNEL_GEMM_COLS, KERNEL_SWEEP_ROWS, KERNEL_SWEEP_COLS, KERNEL_SWEEP_S = 0, 1, 2, 3, 4
UPDATE_DRAW, UPDATE_MODE, UPDATE_ICM = 0, 1, 2


class BnmtfError(RuntimeError):
    pass


class Problem(C.Structure):
    _fields_ = [("I", C.c_int32), ("J", C.c_int32), ("K", C.c_int32), ("L", C.c_int32),
                ("R", C.c_void_p), ("M", C.c_void_p),
                ("lambda_rows", C.c_void_p), ("lambda_cols", C.c_void_p), ("lambda_S", C.c_void_p),
                ("alpha", C.c_double), ("beta", C.c_double), ("seed", C.c_uint64),
                ("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32),
                ("comm_id", C.c_void_p)]


_P = C.c_void_p
_SIGS = {
    "bnmtf_version": ([], C.c_int),
    "bnmtf_last_error": ([], C.c_char_p),
    "bnmtf_device_count": ([C.POINTER(C.c_int)], C.c_int),
    "bnmtf_comm_unique_id": ([_P], C.c_int),
    "bnmtf_shard_range": ([C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int),
    "bnmtf_create": ([C.POINTER(Problem), C.POINTER(_P)], C.c_int),
    "bnmtf_destroy": ([_P], C.c_int),
    "bnmtf_sync": ([_P], C.c_int),
    "bnmtf_host_alloc": ([C.c_size_t, C.POINTER(_P)], C.c_int),
    "bnmtf_host_free": ([_P], C.c_int),
    "bnmtf_omega_counts": ([_P, C.POINTER(C.c_uint64), _P, _P], C.c_int),
    "bnmtf_set_expectation": ([_P, C.c_int, C.c_int], C.c_int),
    "bnmtf_get_expectation": ([_P, _P, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_uint64)], C.c_int),
    "bnmtf_set_iteration": ([_P, C.c_uint64], C.c_int),
    "bnmtf_get_iteration": ([_P, C.POINTER(C.c_uint64)], C.c_int),
    "bnmf_set_state": ([_P, _P, _P, C.c_double], C.c_int),
    "bnmf_get_state": ([_P, _P, _P, C.POINTER(C.c_double)], C.c_int),
    "bnmf_cond_params": ([_P, C.c_int, C.c_int, _P, _P], C.c_int),
    "bnmtf_beta_s": ([_P, C.POINTER(C.c_double)], C.c_int),
    "bnmf_gibbs_run": ([_P, C.c_int, C.c_int, _P, _P, _P, _P, _P], C.c_int),
    "bnmf_gibbs_run_many": ([_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "bnmtf_comm_info": ([_P, C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "bnmtf_has_experiments": ([], C.c_int),
    "bnmtf_set_small_path": ([_P, C.c_int], C.c_int),
    "bnmtf_is_small": ([_P, C.POINTER(C.c_int)], C.c_int),
    "bnmtf_set_state": ([_P, _P, _P, _P, C.c_double], C.c_int),
    "bnmtf_get_state": ([_P, _P, _P, _P, C.POINTER(C.c_double)], C.c_int),
    "bnmtf_gibbs_run_many": ([_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "bnmtf_cond_params": ([_P, C.c_int, C.c_int, C.c_int, _P, _P], C.c_int),
    "bnmtf_gibbs_run": ([_P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P], C.c_int),
    "bnmf_vb_set_state": ([_P] + [_P] * 8 + [C.c_double], C.c_int),
    "bnmf_vb_get_state": ([_P] + [_P] * 8, C.c_int),
    "bnmf_vb_update": ([_P, C.c_int, C.c_int, C.c_int], C.c_int),
    "bnmf_vb_exp_square_diff": ([_P, C.POINTER(C.c_double)], C.c_int),
    "bnmf_vb_masked_sums": ([_P, C.c_int, _P, _P], C.c_int),
    "bnmf_vb_run": ([_P, C.c_int, _P, _P, _P, _P], C.c_int),
    "bnmf_vb_run_many": ([_P, C.c_int, C.c_int, _P, _P, _P, _P, _P], C.c_int),
    "bnmtf_vb_set_state": ([_P] + [_P] * 12 + [C.c_double], C.c_int),
    "bnmtf_vb_get_state": ([_P] + [_P] * 12, C.c_int),
    "bnmtf_vb_update": ([_P, C.c_int, C.c_int, C.c_int, C.c_int], C.c_int),
    "bnmtf_vb_exp_square_diff": ([_P, C.POINTER(C.c_double), _P], C.c_int),
    "bnmtf_vb_run": ([_P, C.c_int, _P, _P, _P, _P, _P], C.c_int),
    "bnmtf_kmeans_create": ([_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)], C.c_int),
    "bnmtf_kmeans_destroy": ([_P], C.c_int),
    "bnmtf_kmeans_assign": ([_P, _P, _P, _P, _P], C.c_int),
    "bnmtf_kmeans_sums": ([_P, _P, _P, _P], C.c_int),
    "bnmtf_kmeans_set_row": ([_P, C.c_int, _P], C.c_int),
    "bnmtf_metric_sums": ([_P, _P, _P, _P, _P, _P], C.c_int),
    "bnmtf_metric_sums_wide": ([_P, _P, _P, _P, C.c_int, _P], C.c_int),
    "bnmtf_set_tau": ([_P, C.c_double], C.c_int),
    "bnmf_vb_half_sweep": ([_P, C.c_int], C.c_int),
    "bnmf_vb_esd_terms": ([_P, _P], C.c_int),
    "bnmf_set_column_block": ([_P, C.c_int], C.c_int),
    "bnmf_set_residual_data": ([_P, _P, C.c_int], C.c_int),
    "bnmf_half_sweep": ([_P, C.c_int, C.c_int], C.c_int),
    "bnmtf_set_s_block": ([_P, C.c_int, C.c_int, C.c_int], C.c_int),
    "bnmtf_s_rows": ([_P, C.c_int, C.c_int, C.c_int], C.c_int),
    "bnmtf_tn_sample": ([_P, _P, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _P], C.c_int),
    "bnmtf_tn_moments": ([_P, _P, C.c_size_t, C.c_int, _P, _P], C.c_int),
    "bnmtf_gamma_sample": ([C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_double)], C.c_int),
    "bnmtf_set_profiling": ([_P, C.c_int], C.c_int),
    "bnmtf_set_minimum_tn": ([_P, C.c_double], C.c_int),
    "bnmtf_set_sweep_path": ([_P, C.c_int], C.c_int),
    "bnmtf_kernel_stats": ([_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)], C.c_int),
    "bnmtf_describe": ([_P, C.c_char_p, C.c_size_t], C.c_int),
}
EXPORTS = tuple(_SIGS)

_lib = None


def lib():
    """The loaded library; raises BnmtfError when it has not been built (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BnmtfError("HIP library %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "or `make -C bnmtf_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (args, res) in _SIGS.items():
            f = getattr(l, name)
            f.argtypes, f.restype = args, res
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise BnmtfError("libbnmtf_hip error %d: %s" % (rc, lib().bnmtf_last_error().decode("utf-8", "replace")))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    n = C.c_int(0)
    check(lib().bnmtf_device_count(C.byref(n)))
    return n.value


# Page-locking is slow (35-40 ms per 200 MB here, and the first device write to a fresh page costs again): a block whose last
# array has gone away is kept for the next sample_buffer() of the same size instead of being unpinned -- run(n) called again and
# again (model-selection drivers, chains continued in pieces) then allocates nothing after its second call.  At most
# BNMTF_PIN_POOL_MB (default 1 024 MB) wait in the pool; 0 switches it off.
_PIN_POOL = {}          # nbytes -> [address, ...]
_PIN_POOL_BYTES = [0]


def _pin_pool_cap():
    return int(os.environ.get("BNMTF_PIN_POOL_MB", "1024")) * (1 << 20)


def _pin_pool_clear():
    for addrs in _PIN_POOL.values():
        for a in addrs:
            try:
                lib().bnmtf_host_free(C.c_void_p(a))
            except Exception:
                pass
    _PIN_POOL.clear()
    _PIN_POOL_BYTES[0] = 0


class _PinnedBlock(object):
    """Owner of one bnmtf_host_alloc block (back to the pool, or freed, when the last array viewing it goes away)."""

    def __init__(self, nbytes):
        pooled = _PIN_POOL.get(nbytes)
        if pooled:
            self.ptr, self.nbytes = pooled.pop(), nbytes
            _PIN_POOL_BYTES[0] -= nbytes
            return
        p = C.c_void_p()
        check(lib().bnmtf_host_alloc(nbytes, C.byref(p)))
        self.ptr, self.nbytes = p.value, nbytes

    def __del__(self):
        try:
            if self.ptr:
                if _PIN_POOL_BYTES[0] + self.nbytes <= _pin_pool_cap():
                    _PIN_POOL.setdefault(self.nbytes, []).append(self.ptr)
                    _PIN_POOL_BYTES[0] += self.nbytes
                else:
                    lib().bnmtf_host_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def sample_buffer(shape, dtype=np.float32):
    """A zero-copy NumPy array over page-locked memory for run()'s sample outputs (asynchronous device-to-host
    copies land in it while the next iterations compute); ordinary pageable memory when pinning fails."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    # page-locking gigabytes takes the host seconds and memory it may not have (1 000 samples of an 8192 x 64 factor are
    # 2 GB): above the cap the array is pageable and run() goes through its small pinned ring instead (BNMTF_PIN_CAP_MB)
    if n > int(os.environ.get("BNMTF_PIN_CAP_MB", "1024")) * (1 << 20):
        return np.zeros(shape, dtype=dtype)
    try:
        blk = _PinnedBlock(max(n, 1))
    except BnmtfError:
        return np.zeros(shape, dtype=dtype)
    buf = (C.c_char * max(n, 1)).from_address(blk.ptr)
    buf._owner = blk                      # the ctypes buffer (kept alive by the array's .base chain) keeps the block
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


atexit.register(_pin_pool_clear)
