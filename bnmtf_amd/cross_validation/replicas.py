"""Independent models on device replicas: a list of jobs (one model fit each) is dealt to worker processes, one per
GPU slot; a worker builds its models with device=<its GPU>.  The parent never touches a GPU.

    pool = ReplicaPool()                 # one worker per visible GPU (in-process when there is a single slot)
    results = pool.map(fit_model, jobs)  # jobs: list of dicts (see fit_model); results in job order

The reference runs its folds in `multiprocessing.Pool(P)` on CPU cores (parallel_matrix_cross_validation.py:52-65);
here the unit of parallelism is a GPU."""
import inspect
import multiprocessing as mp
import threading
import os
import traceback

import numpy as np

_worker_device = [0]
_worker_shared = {}


def visible_devices():
    """Number of GPUs, asked of a short-lived child so that this process never initialises one."""
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as p:
        return p.apply(_count_devices)


def _count_devices():
    import bnmtf_amd
    return bnmtf_amd.device_count()


def _init_worker(device_queue, shared):
    _worker_device[0] = device_queue.get()
    _worker_shared.clear()
    _worker_shared.update(shared)


def _call(args):
    fn, job = args
    try:
        return ("ok", fn(dict(job, device=_worker_device[0]), _worker_shared))
    except Exception as e:      # noqa: BLE001 -- re-raised in the parent with the worker's traceback
        return ("error", "%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc()))


class ReplicaError(Exception):
    """A job's exception as it comes back from a worker: str() is "<Type>: <message>" followed by the worker's traceback;
    `.first_line` is the part a log line wants."""
    @property
    def first_line(self):
        return str(self).split("\n", 1)[0]


SMALL_TRI_DENSE_RANK = 10      # the one-launch kernel's dense S step (csrc/kernel_small.hip: small_tri_dense): K, L <= 10


class ReplicaPool(object):
    def __init__(self, devices=None, shared=None, batched=False):
        """devices: list of device ordinals, one worker each (repeat an ordinal to run several models on one GPU at
        once); default: every visible GPU once.  shared: dict handed to every worker once (e.g. the data matrix R),
        so that jobs only carry what differs between them.  batched: a slot fits ALL the model-fit jobs it is dealt in
        one device call (bnmtf_amd.run_many: small models share a launch, one block each) instead of one after the other."""
        if devices is None:
            n = visible_devices()
            devices = list(range(max(n, 1)))
        self.devices = list(devices)
        self.shared = dict(shared or {})
        # several slots on one GPU: their models run side by side.  A tri-factorisation that fills a CU is 1.3-1.6 x slower on the
        # one-launch path than alone on the multi-launch path -- four of them side by side are ahead there (measured: the greedy
        # search's job 9.4 -> 7.5 s with four slots); a two-factor model is 2.5 x slower and is not (1.6 -> 4.2 s): tri models only
        if len(set(self.devices)) < len(self.devices):
            self.shared.setdefault("_small_path_tri", "always")
        self.batched = bool(batched)
        self._pool = None
        self._lock = threading.Lock()           # (map() may be called from several threads: the folds of a greedy-search cross-validation)

    def _start(self):
        with self._lock:
            if self._pool is None and len(self.devices) > 1:
                ctx = mp.get_context("spawn")
                q = ctx.Queue()
                for d in self.devices:
                    q.put(d)
                self._pool = ctx.Pool(len(self.devices), initializer=_init_worker, initargs=(q, self.shared))

    def map(self, fn, jobs, errors="raise"):
        """fn(job, shared) -> result for every job, results in job order; job gets a 'device' entry.  An exception in
        any job is raised here (after all jobs have run) with the worker's traceback -- or, with errors="return", comes
        back in the job's place as a ReplicaError (a caller that logs a failed setting and carries on)."""
        jobs = list(jobs)
        if self.batched and fn is fit_model and jobs:
            out = self._map_batched(jobs)
        elif len(self.devices) <= 1:                    # single slot: in this process, like the reference's serial loops
            _worker_device[0] = self.devices[0] if self.devices else 0
            _worker_shared.clear(); _worker_shared.update(self.shared)
            out = [_call((fn, j)) for j in jobs]
        else:
            self._start()
            out = self._pool.map(_call, [(fn, j) for j in jobs], chunksize=1)
        errs = [r[1] for r in out if r[0] == "error"]
        if errs and errors == "raise":
            raise RuntimeError("%d of %d replica jobs failed; first:\n%s" % (len(errs), len(jobs), errs[0]))
        return [ReplicaError(r[1]) if r[0] == "error" else r[1] for r in out]

    def _map_batched(self, jobs):
        """Every slot gets a contiguous share of the jobs and fits it as one batch."""
        ns = max(len(self.devices), 1)
        shares = [list(range(len(jobs)))[i::ns] for i in range(ns)]
        if ns == 1:
            _worker_device[0] = self.devices[0] if self.devices else 0
            _worker_shared.clear(); _worker_shared.update(self.shared)
            parts = [_call_batch([jobs[i] for i in shares[0]])]
        else:
            self._start()
            parts = self._pool.map(_call_batch, [[jobs[i] for i in sh] for sh in shares], chunksize=1)
        out = [None] * len(jobs)
        for sh, part in zip(shares, parts):
            for i, r in zip(sh, part):
                out[i] = r
        return out

    def joint(self, participants):
        """A view of this pool for `participants` threads that each walk a chain of dependent map() calls (the folds of a
        greedy-search cross-validation): the calls the threads have open at the same time are run as ONE map() of this pool --
        on a batched pool one device call, a block per model -- and every thread gets its own jobs' results.  A thread that has
        no further call says so with .leave()."""
        return _Joint(self, participants)

    def close(self):
        if self._pool is not None:
            self._pool.close(); self._pool.join()
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class _Joint(object):
    def __init__(self, pool, participants):
        self.pool, self.active = pool, int(participants)
        self.shared, self.devices, self.batched = pool.shared, pool.devices, pool.batched
        self._cv = threading.Condition()
        self._open = []                     # [fn, jobs, results] of the calls waiting for the others

    def map(self, fn, jobs, errors="raise"):
        call = [fn, list(jobs), None]
        with self._cv:
            self._open.append(call)
            if len(self._open) >= self.active:
                self._run()
            while call[2] is None:
                self._cv.wait()
        out = call[2]
        if isinstance(out, BaseException):
            raise out
        errs = [r for r in out if isinstance(r, ReplicaError)]
        if errs and errors == "raise":
            raise RuntimeError("%d of %d replica jobs failed; first:\n%s" % (len(errs), len(out), errs[0]))
        return out

    def leave(self):
        with self._cv:
            self.active -= 1
            if self._open and len(self._open) >= self.active:
                self._run()

    def _run(self):
        """(called with the lock held: everybody else is waiting for this very call)"""
        calls, self._open = self._open, []
        try:
            by_fn = {}
            for c in calls:
                by_fn.setdefault(c[0], []).append(c)
            for fn, cs in by_fn.items():
                res = self.pool.map(fn, [j for c in cs for j in c[1]], errors="return")
                at = 0
                for c in cs:
                    c[2] = res[at:at + len(c[1])]; at += len(c[1])
        except BaseException as e:          # noqa: BLE001 -- handed to every waiting caller
            for c in calls:
                if c[2] is None:
                    c[2] = e
        self._cv.notify_all()

    def close(self):
        pass


def _call_batch(jobs):
    try:
        return [("ok", r) for r in fit_models([dict(j, device=_worker_device[0]) for j in jobs], _worker_shared)]
    except Exception as e:      # noqa: BLE001 -- one failure fails the share: every job of it reports it
        msg = "%s: %s\n%s" % (type(e).__name__, e, traceback.format_exc())
        return [("error", msg) for _ in jobs]


def _accepts(fn, name):
    try:
        return name in inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False


def _build(job, shared):
    """The model of one job: constructed and initialised (the job's seeds set first)."""
    import random
    R = job["R"] if job.get("R") is not None else shared["R"]
    cls = job["classifier"]
    kw = {}
    if _accepts(cls.__init__, "device"):
        kw["device"] = job.get("device", 0)
    if _accepts(cls.__init__, "verbose"):
        kw["verbose"] = False
    if job.get("seed") is not None:
        np.random.seed(job["seed"] % (2 ** 32)); random.seed(job["seed"])
        if _accepts(cls.__init__, "seed"):
            kw["seed"] = job["seed"]
    model = cls(R, np.asarray(job["M"], dtype=float), *job["args"], **kw)
    model.initialise(**job["init"])
    # (after initialise: set_small_path creates the device handle, and a model built without a seed draws its Philox key from
    # NumPy's global stream at that moment -- the initial factors must see the same stream as on the single-slot path.)
    # Forced only where it was measured to pay: the DENSE S step of the one-launch kernel (K, L <= 10: 9.4 -> 7.5 s for the greedy
    # search's job on four slots).  Wider ranks walk S sequentially on the block (K = L = 32: 0.6 k it/s against 4.3 k alone on the
    # multi-launch path): they keep the library's own rule ('auto': small_wanted's batch threshold).
    if (shared.get("_small_path_tri") is not None and getattr(model, "L", 0) and hasattr(model, "set_small_path")
            and max(int(model.K), int(model.L)) <= SMALL_TRI_DENSE_RANK):
        model.set_small_path(shared["_small_path_tri"])
    return model


def _run_kw(job, model):
    burn_in, thinning = job.get("burn_in"), job.get("thinning")
    sampled = burn_in is not None and thinning is not None
    run_kw = {"iterations": job["iterations"]}
    if job.get("minimum_TN") is not None:
        run_kw["minimum_TN"] = job["minimum_TN"]
    if sampled and _accepts(model.run, "expectation"):          # posterior means on the device, no sample hand-off
        run_kw["expectation"] = (burn_in, thinning); run_kw["store_samples"] = False
    return run_kw, sampled


def _score(job, model, sampled):
    q_args = (job.get("burn_in"), job.get("thinning")) if sampled else ()
    quality = {m: model.quality(m, *q_args) for m in job.get("metrics", ["loglikelihood"])}
    perf = model.predict(np.asarray(job["test"], dtype=float), *q_args) if job.get("test") is not None else None
    if hasattr(model, "close"):
        model.close()
    return {"quality": quality, "performance": perf}


def fit_model(job, shared):
    """One candidate model, start to finish, on the worker's GPU.  job:
        classifier   the model class (bnmf_gibbs_optimised, nmf_icm, bnmtf_vb_optimised, ...)
        args         positional arguments after (R, M): (K, priors) or (K, L, priors)
        init         kwargs of initialise()
        iterations, burn_in, thinning, minimum_TN (the last three may be None)
        M            training mask;  test: mask to predict on, or None;  R: data matrix, or absent (then shared['R'])
        seed         optional (numpy / random / sampler seed of this candidate)
    Returns {'quality': {metric: value}, 'performance': predict(test) or None}."""
    model = _build(job, shared)
    run_kw, sampled = _run_kw(job, model)
    model.run(**run_kw)
    return _score(job, model, sampled)


def fit_models(jobs, shared):
    """The same for a list of jobs at once: all models are built first, those that bnmtf_amd.run_many takes (BNMF Gibbs) and
    that ask for the same run (iterations, expectation) go to the device as ONE call -- small models share a launch, one
    block each --, then every model is scored.  Results in job order, each what fit_model(job) returns."""
    from ..batch import run_many, takes
    models = [_build(j, shared) for j in jobs]
    kws = [_run_kw(j, m) for j, m in zip(jobs, models)]
    groups = {}
    for i, (m, (kw, _)) in enumerate(zip(models, kws)):
        if takes(m) and "minimum_TN" not in kw:                     # (nmf_icm inherits _run_prepare but not the Gibbs run(): one by one)
            groups.setdefault((kw["iterations"], kw.get("expectation"), kw.get("store_samples", True)), []).append(i)
        else:
            m.run(**kw)
    for (iterations, expectation, store), idx in groups.items():
        run_many([models[i] for i in idx], iterations, store_samples=store, expectation=expectation)
    return [_score(j, m, sampled) for j, m, (_, sampled) in zip(jobs, models, kws)]
