"""
parallel.py - seed sharding across the GPUs of one node.

Independent control seeds shard embarrassingly: rank r evaluates its own block of seeds with
its own context on its own GPU; the only exchange is ONE all-reduce of
[sum_b cost_b, sum_b d cost_b / d controls] (1 + Nc*K doubles) through RCCL over xGMI.
The reference has no counterpart (single process).

The communicator is abstract so the sharding logic can be exercised on CPU with gloo
(tests/test_parallel.py); the product communicator is RCCL through libqocx (RcclComm).
"""

import os
import time

import numpy as np


def env_world():
    """(rank, world, local_rank) from the torchrun-style environment (defaults 0, 1, 0)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local_rank


def shard_bounds(total, rank, world):
    """Contiguous block [lo, hi) of `total` seeds owned by `rank`; sizes differ by at most 1."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


class SingleComm(object):
    rank, world = 0, 1

    def allreduce_sum(self, array):
        return np.asarray(array, dtype=np.float64)

    def allreduce_max(self, array):
        return np.asarray(array, dtype=np.float64)

    def barrier(self):
        return None


def _rendezvous_path():
    explicit = os.environ.get("QOCX_RDZV_FILE")
    if explicit:
        return explicit
    key = "{}_{}_{}".format(os.environ.get("MASTER_ADDR", "local"),
                            os.environ.get("MASTER_PORT", "0"), os.getppid())
    base = os.environ.get("QOCX_RDZV_DIR", "/tmp")
    return os.path.join(base, "qocx_rdzv_" + key.replace("/", "_"))


class RcclComm(object):
    """
    RCCL communicator owned by an Engine context. The 128-byte ncclUniqueId of rank 0 reaches
    the other ranks of the node through a file keyed by (MASTER_ADDR, MASTER_PORT, parent pid);
    all ranks of one torchrun launch share that parent.
    """

    def __init__(self, engine, rank=None, world=None, timeout=300.0):
        env_rank, env_world_size, _ = env_world()
        self.rank = env_rank if rank is None else rank
        self.world = env_world_size if world is None else world
        self.engine = engine
        path = _rendezvous_path()
        if self.rank == 0:
            uid = engine.comm_unique_id()
            tmp = "{}.{}.tmp".format(path, os.getpid())
            with open(tmp, "wb") as f:
                f.write(uid)
            os.replace(tmp, path)
        else:
            deadline = time.time() + timeout
            uid = None
            while time.time() < deadline:
                try:
                    with open(path, "rb") as f:
                        data = f.read()
                    if len(data) == 128:
                        uid = data
                        break
                except OSError:
                    pass
                time.sleep(0.05)
            if uid is None:
                raise RuntimeError("qocx rendezvous: no unique id at {}".format(path))
        engine.comm_init(uid, self.rank, self.world)
        self.barrier()
        if self.rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass

    def allreduce_sum(self, array):
        return self.engine.comm_allreduce_sum(array)

    def allreduce_max(self, array):
        return self.engine.comm_allreduce_max(array)

    def barrier(self):
        self.engine.comm_barrier()


def summed_cost_and_gradient(cost, grads, comm):
    """
    The path's single collective: per-seed results of this rank -> [sum cost, sum gradient]
    over every seed of every rank.
    """
    cost = np.asarray(cost, dtype=np.float64)
    if grads is None:
        packed = np.array([cost.sum()])
    else:
        grads = np.asarray(grads, dtype=np.float64)
        packed = np.concatenate([[cost.sum()], grads.sum(axis=0).ravel()])
    packed = comm.allreduce_sum(packed)
    total_cost = float(packed[0])
    total_grad = None if grads is None else packed[1:].reshape(grads.shape[1:])
    return total_cost, total_grad
