"""The torch-free control plane of the multi-GPU path (bnmtf_amd/comm.py): rank 0 <-> ranks over TCP on 127.0.0.1 --
broadcast of the 128-byte communicator id, barrier, max-reduction of the timings -- and spawn_local(), the launcher
bench.py uses when it is called as `python bench.py --gpus N` without an external launcher.  No GPU, no library call."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, %r)
    from bnmtf_amd.comm import ControlPlane
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert int(os.environ["LOCAL_RANK"]) == rank
    cp = ControlPlane(rank, world)
    cid = cp.broadcast(bytes(range(128)) if rank == 0 else None)
    assert cid == bytes(range(128))
    time.sleep(0.05 * rank)                       # ranks arrive at different times
    cp.barrier()
    m = cp.allreduce_max(1.5 + rank)
    assert m == 1.5 + (world - 1), m
    for i in range(20):
        assert cp.allreduce_max(float(i * (rank + 1))) == float(i * world)
    cp.barrier(); cp.close()
    print("rank %%d ok" %% rank)
    sys.exit(7 if (rank == 1 and os.environ.get("FAIL_ONE")) else 0)
""") % ROOT


def _spawn(n, extra_env=None):
    code = "import sys; sys.path.insert(0, %r); from bnmtf_amd import comm; sys.exit(comm.spawn_local(%d, ['-c', %r]))" % (ROOT, n, CHILD)
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)


def test_three_ranks_exchange_id_barrier_and_max():
    r = _spawn(3)
    assert r.returncode == 0, r.stderr
    assert sorted(r.stdout.split("\n")[:3]) == ["rank 0 ok", "rank 1 ok", "rank 2 ok"]


def test_spawn_local_reports_a_failing_rank():
    r = _spawn(2, {"FAIL_ONE": "1"})
    assert r.returncode == 7


def test_single_rank_needs_no_sockets():
    sys.path.insert(0, ROOT)
    from bnmtf_amd.comm import ControlPlane
    cp = ControlPlane(0, 1)
    assert cp.broadcast(b"x") == b"x" and cp.allreduce_max(2.0) == 2.0
    cp.barrier(); cp.close()


def test_bench_spawns_its_own_ranks_when_called_directly():
    """`python bench.py --gpus 2` (no launcher): the parent starts two children before touching a GPU; here, without a GPU,
    both children reach the library call that needs one and fail loudly -- there is no CPU fallback -- and the parent
    returns their exit code instead of raising `needs torch.distributed.run`."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "bnmf_1024_k16", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert "torch.distributed.run" not in r.stderr
    try:
        import bnmtf_amd
        has_gpu = bnmtf_amd.device_count() > 0
    except Exception:
        has_gpu = False
    if not has_gpu:
        assert r.returncode != 0 and "BnmtfError" in r.stderr
