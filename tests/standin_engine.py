"""
standin_engine.py - TEST INFRASTRUCTURE: a CPU stand-in for qoc_amd.engine.Engine that bench.py
loads when its test-only flag --standin-engine names this module (tests/test_bench_main.py). It rehearses the
multi-rank CONTROL FLOW of bench.py on a box without GPUs - torchrun environment, the real file
rendezvous of qoc_amd.parallel.RcclComm, the --gpus / WORLD_SIZE check, the barrier + max-over-ranks
timing, rank-0-only printing - with gloo in the place of RCCL. It computes nothing of the hot
path: "cost" of a seed = sum of its squared controls. Never imported by the product.
"""
import os
import time

import numpy as np

KERNEL_NAMES = ("pade_pq", "sweep", "krylov_grad", "scatter", "lu", "lindblad", "lindblad_combine")
COST_TARGET_COHERENT = 0


class Engine(object):
    created = 0

    def __init__(self, device=-1):
        Engine.created += 1
        self.device = device
        self._controls = None
        self._cost = self._grads = None
        self._launches = 0
        self._group = None
        self._nc = self._k = 0
        self.log = []

    # -- problem / evaluation ---------------------------------------------------------------
    def set_schroedinger_problem(self, n, s, k, nc, n_eval, evolution_time, h0, g, psi0, costs=(),
                                 **kwargs):
        self._nc, self._k = nc, k
        self.n_eval = n_eval

    def set_pipeline(self, segments):
        self.log.append(("pipeline", segments))

    def upload_controls(self, controls):
        self._controls = np.array(controls, dtype=np.float64)

    def eval_resident(self, want_grad=True):
        time.sleep(0.002)  # something for the clock to see
        self._cost = np.sum(self._controls ** 2, axis=(1, 2))
        self._grads = 2.0 * self._controls if want_grad else None
        self._launches += 8

    def evaluate(self, controls, want_grad=True):
        self.upload_controls(controls)
        self.eval_resident(want_grad)
        return self._cost, self._grads, np.zeros((len(self._cost), 1, 2), dtype=np.complex128)

    def reduce_results(self, allreduce=False, want_grad=True):
        packed = np.concatenate([[self._cost.sum()], self._grads.sum(axis=0).ravel()])
        if allreduce:
            packed = self.comm_allreduce_sum(packed)
        return float(packed[0]), packed[1:].reshape(self._nc, self._k)

    def download_results(self, want_grad=True, want_final=True):
        return self._cost, self._grads, None

    def synchronize(self):
        return None

    def pade_orders(self):
        return {3: 0, 5: len(self._cost) * (self.n_eval - 1), 7: 0, 9: 0, 13: 0}

    def set_timing(self, enable, only=None):
        return None

    def reset_timing(self):
        self._launches = 0

    def timing(self):
        out = {name: (0, 0.0) for name in KERNEL_NAMES}
        out["pade_pq"] = (self._launches, 0.8 * self._launches)
        return out

    def set_knob(self, name, value):
        self.log.append((name, value))

    def mfma_peak(self, waves_per_simd=1, iters=20000):
        return 47.0

    def close(self):
        if self._group is not None:
            import torch.distributed as dist
            dist.barrier()
            dist.destroy_process_group()
            self._group = None

    # -- the communicator (gloo where the product has RCCL) ------------------------------------
    @staticmethod
    def comm_unique_id():
        return os.urandom(128)

    def comm_init(self, unique_id, rank, world):
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
        self._group = True
        # every rank must have received rank 0's id through the file rendezvous
        mine = torch.tensor(list(bytes(unique_id)), dtype=torch.int64)
        ref = mine.clone()
        dist.broadcast(ref, src=0)
        if not bool((mine == ref).all()):
            raise RuntimeError("stand-in: rank {} holds a different unique id than rank 0".format(rank))

    def _allreduce(self, array, op):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.array(array, dtype=np.float64).ravel())
        dist.all_reduce(t, op=op)
        return t.numpy().reshape(np.shape(array))

    def comm_allreduce_sum(self, array):
        import torch.distributed as dist
        return self._allreduce(array, dist.ReduceOp.SUM)

    def comm_allreduce_max(self, array):
        import torch.distributed as dist
        return self._allreduce(array, dist.ReduceOp.MAX)

    def comm_barrier(self):
        import torch.distributed as dist
        dist.barrier()
