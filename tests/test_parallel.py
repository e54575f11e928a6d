"""
CPU tests of the N > 1 path: seed sharding and the single summed cost/gradient all-reduce,
exercised with world_size = 2 over gloo (torch.distributed is test plumbing here; the product
communicator is RCCL through libqocx, qoc_amd.parallel.RcclComm).
"""

import os
import socket
import sys

import numpy as np
import pytest

from qoc_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for total in (1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_comm_is_identity():
    cost = np.array([1.0, 2.0, 3.0])
    grads = np.arange(24, dtype=np.float64).reshape(3, 4, 2)
    total, g = parallel.summed_cost_and_gradient(cost, grads, parallel.SingleComm())
    assert total == 6.0 and np.array_equal(g, grads.sum(axis=0))
    total, g = parallel.summed_cost_and_gradient(cost, None, parallel.SingleComm())
    assert total == 6.0 and g is None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import qoc_numpy as onp
    from tests import cases as cases_mod
    from tests.helpers import oracle_problem

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class GlooComm(object):
        def __init__(self):
            self.rank, self.world = rank, world

        def allreduce_sum(self, array):
            import torch
            t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.numpy()

        def barrier(self):
            dist.barrier()

    case = cases_mod.case_by_name("scaled_n8")
    seeds = np.concatenate([case.controls, 0.5 * case.controls, -case.controls])  # 6 seeds
    lo, hi = parallel.shard_bounds(len(seeds), rank, world)
    problem = oracle_problem(case)
    cost, grads = [], []
    for u in seeds[lo:hi]:  # the oracle stands in for the per-rank HIP evaluation
        e, g, _ = onp.evaluate_with_grad(problem, u)
        cost.append(e)
        grads.append(g)
    total, grad = parallel.summed_cost_and_gradient(np.array(cost), np.stack(grads), GlooComm())
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), total=total, grad=grad, lo=lo, hi=hi)
    dist.destroy_process_group()


def test_world_size_two_gloo(tmp_path):
    import torch.multiprocessing as mp
    from oracle import qoc_numpy as onp
    from tests import cases as cases_mod
    from tests.helpers import oracle_problem

    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    case = cases_mod.case_by_name("scaled_n8")
    seeds = np.concatenate([case.controls, 0.5 * case.controls, -case.controls])
    problem = oracle_problem(case)
    ref_total, ref_grad = 0.0, 0.0
    for u in seeds:
        e, g, _ = onp.evaluate_with_grad(problem, u)
        ref_total, ref_grad = ref_total + e, ref_grad + g
    covered = []
    for r in range(world):
        out = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert abs(float(out["total"]) - ref_total) < 1e-12
        assert np.max(np.abs(out["grad"] - ref_grad)) < 1e-12
        covered.append((int(out["lo"]), int(out["hi"])))
    assert covered == [(0, 3), (3, 6)]


def test_rendezvous_file_roundtrip(tmp_path, monkeypatch):
    """The file rendezvous used to pass rank 0's ncclUniqueId (no RCCL call involved)."""
    monkeypatch.setenv("QOCX_RDZV_FILE", str(tmp_path / "uid"))
    assert parallel._rendezvous_path() == str(tmp_path / "uid")
    monkeypatch.delenv("QOCX_RDZV_FILE")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29512")
    path = parallel._rendezvous_path()
    assert "127.0.0.1_29512_" in path and str(os.getppid()) in path
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert parallel.env_world() == (3, 8, 3)


def test_rendezvous_ignores_files_of_a_crashed_launch(tmp_path):
    """a stale id file (same key) left by a crashed launch must never reach
    ncclCommInitRank; the nonce/token handshake makes every rank end with rank 0's fresh id."""
    import threading
    from qoc_amd import parallel

    path = str(tmp_path / "qocx_rdzv_test")
    world = 3
    stale_uid, stale_nonce = b"\x01" * 128, b"\x02" * 16
    with open(path, "wb") as f:                      # crashed launch: id published ...
        f.write(stale_uid + stale_nonce)
    with open(path + ".go", "wb") as f:              # ... and even released
        f.write(stale_nonce + b"\x03" * 16 * (world - 1))
    fresh = bytes(range(128))
    got = {}

    def run(rank, delay):
        import time
        time.sleep(delay)
        got[rank] = parallel.exchange_unique_id(path, rank, world, lambda: fresh, timeout=20.0)

    # the other ranks start BEFORE rank 0 and see the stale files first
    threads = [threading.Thread(target=run, args=(1, 0.0)),
               threading.Thread(target=run, args=(2, 0.05)),
               threading.Thread(target=run, args=(0, 0.4))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(30)
    assert got == {0: fresh, 1: fresh, 2: fresh}
    parallel.cleanup_rendezvous(path, world)
    assert not any(name.startswith("qocx_rdzv_test") for name in os.listdir(str(tmp_path)))
