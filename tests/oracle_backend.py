"""
oracle_backend.py - TEST INFRASTRUCTURE: an object with the Engine's method surface whose
arithmetic is the CPU oracle. Injected through tests.helpers.set_backend_factory so that
the host logic (entry points, structure extraction, control layouts, optimizers, bookkeeping)
can be tested without a GPU. Never shipped, never selected by the product.
"""

import numpy as np

from oracle import qoc_numpy as onp


class _DescriptorCost(onp.OracleCost):
    """Oracle-side evaluation of a device cost descriptor (include/qocx.h qocx_cost_desc)."""

    def __init__(self, desc, state_count, n):
        super().__init__(1.0)
        self.kind = desc["kind"]
        self.requires_step_evaluation = bool(desc["step_cost"])
        self.scale = desc["scale"]
        vec = np.asarray(desc["vectors"], dtype=np.complex128).reshape(-1, n)
        self.S = state_count
        if self.kind == 2:
            counts = list(desc["counts"])
            self.sets, base = [], 0
            for c in counts:
                self.sets.append(vec[base:base + c])
                base += c
        else:
            self.targets = vec

    def cost(self, controls, states, step):
        psi = np.asarray(states)[:, :, 0]
        if self.kind == 0:
            tot = np.sum(np.sum(np.conj(self.targets) * psi, axis=1))
            return self.scale * (1 - np.abs(tot) ** 2 / self.S ** 2)
        if self.kind == 1:
            ip = np.sum(np.conj(self.targets) * psi, axis=1)
            return self.scale * (1 - np.sum(np.abs(ip) ** 2) / self.S)
        total = 0
        for s, fs in enumerate(self.sets):
            ip = fs.conj() @ psi[s]
            total += np.sum(np.abs(ip) ** 2) / len(fs)
        return self.scale * total

    def states_bar(self, controls, states, step):
        psi = np.asarray(states)[:, :, 0]
        out = np.zeros_like(psi)
        if self.kind == 0:
            tot = np.sum(np.sum(np.conj(self.targets) * psi, axis=1))
            out = -(2 * self.scale / self.S ** 2) * tot * self.targets
        elif self.kind == 1:
            ip = np.sum(np.conj(self.targets) * psi, axis=1)
            out = -(2 * self.scale / self.S) * ip[:, None] * self.targets
        else:
            for s, fs in enumerate(self.sets):
                ip = fs.conj() @ psi[s]
                out[s] = (2 * self.scale / len(fs)) * (ip[:, None] * fs).sum(axis=0)
        return out[:, :, None]


class _InjectedCotangent(onp.OracleCost):
    """Host-supplied state cotangents of one seed (qocx_set_state_cotangents): contributes no
    cost, only states_bar at the given system steps."""

    def __init__(self, by_step, step_cost):
        super().__init__(1.0)
        self.requires_step_evaluation = step_cost
        self.by_step = by_step

    def cost(self, controls, states, step):
        return 0.0

    def states_bar(self, controls, states, step):
        bar = self.by_step.get(step)
        return None if bar is None else bar[:, :, None]


class _DensityDescriptorCost(object):
    """Oracle-side evaluation of a density cost descriptor (qocx.h kinds 3 and 4)."""

    def __init__(self, desc, density_count, n):
        self.kind = desc["kind"]
        self.requires_step_evaluation = bool(desc["step_cost"])
        self.scale = desc["scale"]
        mats = np.asarray(desc["vectors"], dtype=np.complex128).reshape(-1, n, n)
        self.S, self.n = density_count, n
        if self.kind == 4:
            self.sets, base = [], 0
            for c in desc["counts"]:
                self.sets.append(mats[base:base + c])
                base += c
        else:
            self.targets = mats

    def cost(self, controls, densities, step):
        if self.kind == 3:
            ip = np.einsum("sij,sij->s", self.targets.conj(), densities)
            return self.scale * (1 - np.sum(np.abs(ip)) / (self.S * self.n))
        total = 0
        for s, fs in enumerate(self.sets):
            ip = np.einsum("fij,ij->f", fs.conj(), densities[s]) / self.n
            total += np.sum(np.abs(ip) ** 2) / len(fs)
        return self.scale * total

    def states_bar(self, controls, densities, step):
        out = np.zeros_like(np.asarray(densities, dtype=np.complex128))
        if self.kind == 3:
            ip = np.einsum("sij,sij->s", self.targets.conj(), densities)
            for s in range(self.S):
                if abs(ip[s]) > 0:
                    out[s] = -(self.scale / (self.S * self.n)) * (ip[s] / abs(ip[s])) * self.targets[s]
        else:
            for s, fs in enumerate(self.sets):
                ip = np.einsum("fij,ij->f", fs.conj(), densities[s]) / self.n
                out[s] = (2 * self.scale / (len(fs) * self.n)) * np.einsum("f,fij->ij", ip, fs)
        return out


class OracleBackend(object):
    def __init__(self):
        self.keep = False
        self.calls = 0

    def set_schroedinger_problem(self, n, S, K, Nc, N, T, h0, g, psi0, costs=(),
                                 cost_eval_step=1, magnus_policy="M2"):
        h0 = np.asarray(h0, dtype=np.complex128).reshape(-1, n, n)
        nt = h0.shape[0]
        g = np.asarray(g if K > 0 else np.zeros((nt, 0, n, n)),
                       dtype=np.complex128).reshape(nt, K, n, n)
        dt = T / (N - 1)
        nodes = {"M2": (0.5,), "M4": (0.5 - 3 ** 0.5 / 6, 0.5 + 3 ** 0.5 / 6),
                 "M6": (0.5 - 15 ** 0.5 / 10, 0.5, 0.5 + 15 ** 0.5 / 10)}[magnus_policy]
        sample_times = np.array([step * dt + dt * c for step in range(N - 1) for c in nodes])

        def hamiltonian(u, t):
            j = 0 if nt == 1 else int(np.argmin(np.abs(sample_times - t)))
            h = h0[j]
            for k in range(K):
                h = h + u[k] * g[j, k]
            return h

        ocosts = [_DescriptorCost(c, S, n) for c in costs]
        self.problem = onp.SchroedingerProblem(
            T, hamiltonian, np.asarray(psi0, dtype=np.complex128).reshape(S, n, 1), N,
            control_eval_count=Nc, costs=ocosts, cost_eval_step=cost_eval_step,
            magnus_policy=magnus_policy, complex_controls=False, control_count=K)
        self.dims = (n, S, K, Nc, N)

    def set_keep_step_states(self, keep):
        self.keep = bool(keep)

    def set_state_cotangents(self, steps, bars):
        if steps is None or len(steps) == 0:
            self.inj = None
            return
        n, S, K, Nc, N = self.dims
        self.inj = (list(int(x) for x in steps),
                    np.asarray(bars, dtype=np.complex128).reshape(-1, len(steps), S, n))

    def _problem_for_seed(self, b):
        inj = getattr(self, "inj", None)
        if inj is None:
            return self.problem
        import copy
        n, S, K, Nc, N = self.dims
        steps, bars = inj
        assert bars.shape[0] == self.batch
        ces = self.problem.cost_eval_step
        on_grid = {st: bars[b, r] for r, st in enumerate(steps) if st % ces == 0}
        off_grid = {st: bars[b, r] for r, st in enumerate(steps) if st % ces != 0}
        assert all(st == N - 1 for st in off_grid), "off-grid cotangents only at the final step"
        p = copy.copy(self.problem)
        p.costs = list(self.problem.costs) + [_InjectedCotangent(on_grid, True),
                                              _InjectedCotangent(off_grid, False)]
        p.step_costs = [c for c in p.costs if c.requires_step_evaluation]
        return p

    def upload_controls(self, controls):
        n, S, K, Nc, N = self.dims
        if K == 0:
            self.controls = [None] * (1 if controls is None else int(controls))
        else:
            self.controls = list(np.asarray(controls, dtype=np.float64).reshape(-1, Nc, K))
        self.batch = len(self.controls)

    def eval_resident(self, want_grad=True):
        n, S, K, Nc, N = self.dims
        self.calls += 1
        self.cost, self.grads, self.final, self.steps = [], [], [], []
        for b, u in enumerate(self.controls):
            if want_grad and K > 0:
                err, gr, fin = onp.evaluate_with_grad(self._problem_for_seed(b), u)
                self.grads.append(gr)
            else:
                err, fin = onp.evaluate(self.problem, u)
            if self.keep:
                inter = []
                onp.evaluate(self.problem, u, intermediate=inter)
                self.steps.append(np.stack(inter)[:, :, :, 0])
            self.cost.append(err)
            self.final.append(np.asarray(fin)[:, :, 0])

    def download_results(self, want_grad=True, want_final=True):
        n, S, K, Nc, N = self.dims
        grads = np.stack(self.grads) if (want_grad and K > 0 and self.grads) else None
        return (np.array(self.cost, dtype=np.float64), grads,
                np.stack(self.final) if want_final else None)

    def download_step_states(self):
        return np.stack(self.steps)

    # -- Lindblad: the NumPy model of the device algorithm (tests/lindblad_model.py) -----------
    @staticmethod
    def lindblad_stage_times(*args):
        from qoc_amd.engine import Engine  # a host-only function of libqocx, no GPU involved
        return Engine.lindblad_stage_times(*args)

    def set_lindblad_problem(self, n, S, K, Nc, N, T, h0, g, dissipators, operators,
                             initial_densities, costs=(), cost_eval_step=1, fixed_subdivision=0,
                             h0_stages=None, g_stages=None, diss_stages=None, op_stages=None):
        from tests import lindblad_model as lm
        g = np.asarray(g if K > 0 else np.zeros((0, n, n)), dtype=np.complex128).reshape(K, n, n)
        h0_of_t = g_of_t = data_of_t = None
        self.lb_subdivision = None
        if fixed_subdivision:
            times = self.lindblad_stage_times(T, N, Nc, K, fixed_subdivision)
            hs = np.asarray(h0_stages, dtype=np.complex128).reshape(len(times), n, n)
            h0_of_t = lambda t: hs[int(np.argmin(np.abs(times - t)))]
            if g_stages is not None:
                gs = np.asarray(g_stages, dtype=np.complex128).reshape(len(times), K, n, n)
                g_of_t = lambda t: list(gs[int(np.argmin(np.abs(times - t)))])
                g = np.max(np.abs(gs), axis=0) * 0 + gs[np.argmax(
                    [np.linalg.norm(x.reshape(K * n, n), 1) for x in gs])]
            h0 = hs[int(np.argmax([np.linalg.norm(x, 1) for x in hs]))]
            if op_stages is not None:
                ds = np.asarray(diss_stages, dtype=np.float64).reshape(len(times), -1)
                os_ = np.asarray(op_stages, dtype=np.complex128).reshape(len(times), -1, n, n)
                data_of_t = lambda t: (ds[int(np.argmin(np.abs(times - t)))],
                                       os_[int(np.argmin(np.abs(times - t)))])
                worst = int(np.argmax([sum(g_ * np.linalg.norm(o, 1) * np.linalg.norm(o, np.inf)
                                           for g_, o in zip(d, o_)) for d, o_ in zip(ds, os_)]))
                dissipators, operators = ds[worst], os_[worst]
            self.lb_subdivision = int(fixed_subdivision)
        self.lb_system = lm.StructuredLindblad(np.asarray(h0).reshape(n, n), list(g),
                                               dissipators, operators, h0_of_t, g_of_t, data_of_t)
        self.lb_costs = [_DensityDescriptorCost(c, S, n) for c in costs]
        self.lb = dict(n=n, S=S, K=K, Nc=Nc, N=N, T=T, ces=cost_eval_step,
                       rho0=np.asarray(initial_densities, dtype=np.complex128).reshape(S, n, n))

    def set_density_cotangents(self, steps, bars):
        if steps is None or len(steps) == 0:
            self.lb_inj = None
            return
        p = self.lb
        self.lb_inj = (list(int(x) for x in steps), np.asarray(bars, dtype=np.complex128).reshape(
            -1, len(steps), p["S"], p["n"], p["n"]))

    def _lindblad_costs_for_seed(self, b):
        inj = getattr(self, "lb_inj", None)
        if inj is None:
            return self.lb_costs
        steps, bars = inj
        p = self.lb
        ces = p["ces"]

        class Injected(object):
            def __init__(self, by_step, step_cost):
                self.by_step, self.requires_step_evaluation = by_step, step_cost

            def cost(self, controls, densities, step):
                return 0.0

            def states_bar(self, controls, densities, step):
                return self.by_step.get(step, np.zeros_like(densities))

        on_grid = {st: bars[b, r] for r, st in enumerate(steps) if st % ces == 0}
        off_grid = {st: bars[b, r] for r, st in enumerate(steps) if st % ces != 0}
        assert all(st == p["N"] - 1 for st in off_grid)
        return list(self.lb_costs) + [Injected(on_grid, True), Injected(off_grid, False)]

    def evaluate_lindblad(self, controls, want_grad=True, want_final=True):
        from tests import lindblad_model as lm
        p = self.lb
        if p["K"] == 0:
            batch = [np.zeros((2, 0))] * (1 if controls is None else int(controls))
        else:
            batch = list(np.asarray(controls, dtype=np.float64).reshape(-1, p["Nc"], p["K"]))
        want_grad = want_grad and p["K"] > 0
        cost, grads, final, self.lb_steps = [], [], [], []
        self.calls += 1
        for b, u in enumerate(batch):
            costs = self._lindblad_costs_for_seed(b) if want_grad else self.lb_costs
            err, gr, fin = lm.evaluate_with_grad(self.lb_system, u, p["rho0"], p["T"], p["N"],
                                                 costs, p["ces"], want_grad=want_grad,
                                                 subdivision=self.lb_subdivision)
            cost.append(err)
            grads.append(gr)
            final.append(fin)
            if self.keep:
                steps = [p["rho0"]]
                for step in range(1, p["N"]):
                    _, _, rho = lm.evaluate_with_grad(
                        self.lb_system, u, p["rho0"], p["T"], p["N"], [], 1, want_grad=False,
                        stop_step=step, subdivision=self.lb_subdivision)
                    steps.append(rho)
                self.lb_steps.append(np.stack(steps))
        return (np.array(cost, dtype=np.float64), np.stack(grads) if want_grad else None,
                np.stack(final))

    def download_step_densities(self):
        return np.stack(self.lb_steps)

    def close(self):
        pass
