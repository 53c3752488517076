"""bnmtf_amd -- MI355X-native Gibbs / VB inference for Bayesian non-negative matrix
(tri-)factorisation, behind the class API of ThomasBrouwer/BNMTF.

Python host code + ctypes -> libbnmtf_hip.so (hand-written HIP for gfx950).  There is
no CPU fallback: every model method that computes goes through the library."""
from ._lib import BnmtfError, device_count, lib, LIB_PATH, EXPORTS
from .bnmf_gibbs import bnmf_gibbs_optimised, bnmf_gibbs
from .bnmtf_gibbs import bnmtf_gibbs_optimised, bnmtf_gibbs
from .bnmf_vb import bnmf_vb_optimised, bnmf_vb
from .bnmtf_vb import bnmtf_vb_optimised, bnmtf_vb
from .nmf_icm import nmf_icm
from .nmtf_icm import nmtf_icm
from .batch import run_many

__all__ = ["bnmf_gibbs_optimised", "bnmf_gibbs", "bnmtf_gibbs_optimised", "bnmtf_gibbs", "bnmf_vb_optimised", "bnmf_vb", "bnmtf_vb_optimised", "bnmtf_vb", "nmf_icm", "nmtf_icm", "run_many", "device_count", "BnmtfError", "lib", "LIB_PATH", "EXPORTS"]
