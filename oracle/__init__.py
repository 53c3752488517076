"""CPU oracle for the BNMTF Gibbs / VB hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``bnmtf_amd``) never imports this package and fails loudly
when the HIP library is missing.

Parity status: the conditional-posterior parameters, metrics, model-quality
numbers, TN moments and the deterministic VB trajectory are PINNED against
vectors produced by importing the reference itself (``tests/golden/*.npz``,
generator ``tests/golden/make_golden.py``) and against the reference's own
known-answer tests.  Random draws: the reference's tests pin nothing about
them (only ``>= 0``); the oracle's sampler is pinned *distributionally*
against quantiles of 10^6 reference ``TN_vector_draw`` samples per (mu,tau)
pair and against the exact truncated-normal CDF.
"""
