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


def _read(path):
    try:
        with open(path, "rb") as f:
            return f.read()
    except OSError:
        return b""


def _write_atomic(path, data):
    tmp = "{}.{}.tmp".format(path, os.getpid())
    with open(tmp, "wb") as f:
        f.write(data)
    os.replace(tmp, path)


def _remove(path):
    try:
        os.remove(path)
    except OSError:
        pass


_UID, _NONCE = 128, 16


def exchange_unique_id(path, rank, world, make_uid, timeout=300.0, poll=0.02):
    """
    Hand rank 0's 128-byte id to the other ranks through files next to `path`, safe against a
    file a crashed launch with the same key left behind: rank 0 publishes id + a fresh nonce;
    every other rank acknowledges the nonce it read together with a random token of its own and
    proceeds only once rank 0's "go" file carries that nonce AND that token - which a stale
    file cannot. Returns the id (every rank) after which all ranks hold the same one.
    """
    go, deadline = path + ".go", time.time() + timeout
    if rank == 0:
        for stale in [path, go] + ["{}.ack.{}".format(path, r) for r in range(1, world)]:
            _remove(stale)
        uid, nonce = bytes(make_uid()), os.urandom(_NONCE)
        assert len(uid) == _UID
        _write_atomic(path, uid + nonce)
        tokens = [None] * (world - 1)
        while any(t is None for t in tokens):
            if time.time() > deadline:
                raise RuntimeError("qocx rendezvous: ranks {} never acknowledged {}".format(
                    [r + 1 for r, t in enumerate(tokens) if t is None], path))
            for r in range(1, world):
                ack = _read("{}.ack.{}".format(path, r))
                if len(ack) == 2 * _NONCE and ack[:_NONCE] == nonce:
                    tokens[r - 1] = ack[_NONCE:]
            time.sleep(poll)
        _write_atomic(go, nonce + b"".join(tokens))
        return uid
    token, acked = os.urandom(_NONCE), None
    while time.time() < deadline:
        data = _read(path)
        if len(data) == _UID + _NONCE:
            nonce = data[_UID:]
            if nonce != acked:
                _write_atomic("{}.ack.{}".format(path, rank), nonce + token)
                acked = nonce
            ready = _read(go)
            mine = ready[_NONCE * rank:_NONCE * (rank + 1)]
            if ready[:_NONCE] == nonce and mine == token:
                return data[:_UID]
        time.sleep(poll)
    raise RuntimeError("qocx rendezvous: no unique id at {}".format(path))


def cleanup_rendezvous(path, world):
    for name in [path, path + ".go"] + ["{}.ack.{}".format(path, r) for r in range(1, world)]:
        _remove(name)


class RcclComm(object):
    """
    RCCL communicator owned by an Engine context. The 128-byte ncclUniqueId of rank 0 reaches
    the other ranks of the node through files keyed by (MASTER_ADDR, MASTER_PORT, parent pid) -
    all ranks of one torchrun launch share that parent - with the nonce / token handshake of
    exchange_unique_id, so a file left by a crashed launch is never taken for the current one.
    """

    def __init__(self, engine, rank=None, world=None, timeout=300.0):
        env_rank, env_world_size, _ = env_world()
        self.rank = env_rank if rank is None else rank
        self.world = env_world_size if world is None else world
        self.engine = engine
        path = _rendezvous_path()
        uid = exchange_unique_id(path, self.rank, self.world, engine.comm_unique_id, timeout)
        engine.comm_init(uid, self.rank, self.world)
        self.barrier()
        if self.rank == 0:
            cleanup_rendezvous(path, self.world)

    def allreduce_sum(self, array):
        return self.engine.comm_allreduce_sum(array)

    def allreduce_max(self, array):
        return self.engine.comm_allreduce_max(array)

    def barrier(self):
        self.engine.comm_barrier()


def summed_results_resident(engine, comm, want_grad=True):
    """
    The path's single collective on the DEVICE: the per-seed costs and gradients of the engine's
    last evaluation are summed by a kernel, all-reduced over the ranks by one ncclAllReduce on the
    device buffer (RcclComm), and 8 (1 + Nc K) bytes reach the host - instead of 4 MB of per-seed
    gradients, a NumPy sum and a host-staged all-reduce. Communicators that do not live in the
    engine (tests on CPU) take summed_cost_and_gradient.
    """
    if isinstance(comm, RcclComm) and comm.engine is engine:
        return engine.reduce_results(allreduce=True, want_grad=want_grad)
    if isinstance(comm, SingleComm):
        return engine.reduce_results(allreduce=False, want_grad=want_grad)
    cost, grads, _ = engine.download_results(want_grad=want_grad, want_final=False)
    return summed_cost_and_gradient(cost, grads, comm)


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
