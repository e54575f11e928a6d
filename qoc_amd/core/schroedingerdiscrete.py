"""
schroedingerdiscrete.py - evolve_schroedinger_discrete and grape_schroedinger_discrete.

Same positional/keyword signatures, result objects, side effects (stdout table, save file) and
optimizer-callback protocol as qoc/core/schroedingerdiscrete.py:28-353. The time loop, the
matrix exponential and the gradient (the reference's :356-502 traced by autograd) run on the
MI355X through qoc_amd.core.device.SchroedingerEvaluator.
"""

import numpy as np

from qoc_amd.core.common import (clip_control_norms, initialize_controls, slap_controls,
                                 strip_controls)
from qoc_amd.core.device import SchroedingerEvaluator
from qoc_amd.models import (Dummy, EvolveSchroedingerDiscreteState, EvolveSchroedingerResult,
                            GrapeSchroedingerDiscreteState, GrapeSchroedingerResult,
                            InterpolationPolicy, MagnusPolicy)
from qoc_amd.standard.optimizers import SGD, Adam


def evolve_schroedinger_discrete(evolution_time, hamiltonian, initial_states, system_eval_count,
                                 controls=None, cost_eval_step=1, costs=list(),
                                 interpolation_policy=InterpolationPolicy.LINEAR,
                                 magnus_policy=MagnusPolicy.M2, save_file_path=None,
                                 save_intermediate_states=False):
    """
    Evolve state vectors under the Schroedinger equation and compute the optimization error.

    Arguments as in the reference (schroedingerdiscrete.py:28-84):
    evolution_time :: float; hamiltonian :: (controls (control_count), time) -> (n x n);
    initial_states :: (state_count x n x 1); system_eval_count :: int >= 2;
    controls :: (control_eval_count x control_count) or None; costs :: iterable(Cost).
    Returns EvolveSchroedingerResult{error, final_states}.
    """
    if controls is not None:
        controls = np.asarray(controls)
        control_eval_count, control_count = controls.shape[0], controls.shape[1]
    else:
        control_eval_count, control_count = 0, 0
    pstate = EvolveSchroedingerDiscreteState(control_eval_count, cost_eval_step, costs,
                                             evolution_time, hamiltonian, initial_states,
                                             interpolation_policy, magnus_policy, save_file_path,
                                             save_intermediate_states, system_eval_count)
    pstate.save_initial(controls)
    evaluator = SchroedingerEvaluator(
        evolution_time, hamiltonian, initial_states, system_eval_count,
        control_count=control_count, control_eval_count=control_eval_count,
        complex_controls=controls is not None and np.iscomplexobj(controls), costs=costs,
        cost_eval_step=cost_eval_step, interpolation_policy=interpolation_policy,
        magnus_policy=magnus_policy, need_gradients=False, latency_mode=True)
    error, _, final_states, step_states = evaluator.evaluate(
        controls, want_grad=False, want_step_states=pstate.save_intermediate_states_)
    if pstate.save_intermediate_states_:
        pstate.save_all_intermediate_states(0, step_states)
    return EvolveSchroedingerResult(error=error, final_states=final_states)


def grape_schroedinger_discrete(control_count, control_eval_count, costs, evolution_time,
                                hamiltonian, initial_states, system_eval_count,
                                complex_controls=False, cost_eval_step=1,
                                impose_control_conditions=None, initial_controls=None,
                                interpolation_policy=InterpolationPolicy.LINEAR,
                                iteration_count=1000, log_iteration_step=10,
                                magnus_policy=MagnusPolicy.M2, max_control_norms=None,
                                min_error=0, optimizer=Adam(), save_file_path=None,
                                save_intermediate_states=False, save_iteration_step=0):
    """
    Optimize time-discrete controls for the evolution of a set of states (GRAPE).
    Arguments as in the reference (schroedingerdiscrete.py:106-212).
    Returns GrapeSchroedingerResult{best_controls, best_error, best_final_states, best_iteration}.
    """
    initial_controls, max_control_norms = initialize_controls(
        complex_controls, control_count, control_eval_count, evolution_time, initial_controls,
        max_control_norms)
    pstate = GrapeSchroedingerDiscreteState(
        complex_controls, control_count, control_eval_count, cost_eval_step, costs,
        evolution_time, hamiltonian, impose_control_conditions, initial_controls, initial_states,
        interpolation_policy, iteration_count, log_iteration_step, max_control_norms,
        magnus_policy, min_error, optimizer, save_file_path, save_intermediate_states,
        save_iteration_step, system_eval_count)
    pstate.evaluator = SchroedingerEvaluator(
        evolution_time, hamiltonian, initial_states, system_eval_count,
        control_count=control_count, control_eval_count=control_eval_count,
        complex_controls=complex_controls, costs=costs, cost_eval_step=cost_eval_step,
        interpolation_policy=interpolation_policy, magnus_policy=magnus_policy,
        need_gradients=True, latency_mode=True)
    pstate.log_and_save_initial()
    reporter = Dummy()
    reporter.iteration = 0
    result = GrapeSchroedingerResult()
    flat_controls = strip_controls(pstate.complex_controls, pstate.initial_controls)
    pstate.optimizer.run(_esd_wrap, pstate.iteration_count, flat_controls, _esdj_wrap,
                         args=(pstate, reporter, result))
    return result


def _cost_format(flat_controls, pstate):
    """optimizer format -> clipped, conditioned cost-function format (:308-315)."""
    controls = slap_controls(pstate.complex_controls, flat_controls, pstate.controls_shape)
    clip_control_norms(controls, pstate.max_control_norms)  # in place, aliases real params
    if pstate.impose_control_conditions is not None:
        controls = pstate.impose_control_conditions(controls)
    return controls


def _esd_wrap(controls, pstate, reporter, result):
    """function(params, *args) -> (error, terminate); used by optimizers that ask for values."""
    controls = _cost_format(controls, pstate)
    error, _, final_states, _ = pstate.evaluator.evaluate(controls, want_grad=False)
    reporter.error = error
    reporter.final_states = final_states
    return error, bool(error <= pstate.min_error)


def _esdj_wrap(controls, pstate, reporter, result):
    """jacobian(params, *args) -> (grads, terminate); one device evaluation per call."""
    controls = _cost_format(controls, pstate)
    save_states = pstate.save_intermediate_states_
    error, grads, final_states, step_states = pstate.evaluator.evaluate(
        controls, want_grad=True, want_step_states=save_states)
    reporter.error = error
    reporter.final_states = final_states
    if save_states:
        pstate.save_all_intermediate_states(reporter.iteration, step_states)
    if error < result.best_error:  # strict, as schroedingerdiscrete.py:333
        result.best_controls = controls
        result.best_error = error
        result.best_final_states = final_states
        result.best_iteration = reporter.iteration
    pstate.log_and_save(controls, error, final_states, grads, reporter.iteration)
    reporter.iteration += 1
    return strip_controls(pstate.complex_controls, grads), bool(error <= pstate.min_error)


# ---- multi-start GRAPE: B independent optimisations in lock step ---------------

class GrapeSchroedingerBatchResult(object):
    """Per-seed bests of grape_schroedinger_discrete_batch; `best` is the overall winner as a
    GrapeSchroedingerResult. With a communicator the arrays hold this rank's seeds and
    `global_best_error` the minimum over all ranks."""

    def __init__(self, seed_count):
        self.best_controls = [None] * seed_count
        self.best_error = np.repeat(np.finfo(np.float64).max, seed_count)
        self.best_final_states = [None] * seed_count
        self.best_iteration = np.full(seed_count, -1, dtype=np.int64)
        self.iterations_run = np.zeros(seed_count, dtype=np.int64)
        self.global_best_error = None

    @property
    def best(self):
        b = int(np.argmin(self.best_error))
        return GrapeSchroedingerResult(
            best_controls=self.best_controls[b], best_error=float(self.best_error[b]),
            best_final_states=self.best_final_states[b], best_iteration=int(self.best_iteration[b]))


def _optimizer_clone(optimizer, flat_controls):
    """A private copy of a step-wise optimizer plugin, initialised as its run() would."""
    import copy
    if not hasattr(optimizer, "update"):
        raise NotImplementedError(
            "grape_schroedinger_discrete_batch drives the optimizer step by step and needs its "
            "update(grads, params) (Adam, SGD); {} only offers run().".format(optimizer))
    clone = copy.deepcopy(optimizer)
    if hasattr(clone, "gradient_moment"):  # Adam.run(), adam.py:83-88 of the reference
        clone.iteration_count = 0
        clone.gradient_moment = np.zeros_like(flat_controls)
        clone.gradient_square_moment = np.zeros_like(flat_controls)
    return clone


# ---- the B optimizer states as [B, P] arrays ------------------------------------------------------
# The per-seed loop (one plugin object per seed: slap / clip / strip / update = ~60 small NumPy
# calls per seed and iteration) costs 13 ms of host time per iteration at B = 256, Nc = 1001,
# K = 2 - as long as the device evaluation it wraps. For the built-in Adam and
# SGD the same arithmetic runs on [rows, P] blocks: every operation is elementwise and in the
# reference's order (adam.py:110-165, sgd.py), so seed b still walks exactly its single-seed
# trajectory, bit for bit. Row blocks keep the temporaries cache resident.
_ROW_BLOCK = 16
_POOL = None


def _row_blocks(rows):
    """rows (sorted index array) as a list of slices where they are consecutive, index arrays
    otherwise, at most _ROW_BLOCK rows each."""
    out = []
    for lo in range(0, len(rows), _ROW_BLOCK):
        r = rows[lo:lo + _ROW_BLOCK]
        out.append(slice(int(r[0]), int(r[-1]) + 1) if r[-1] - r[0] + 1 == len(r) else r)
    return out


def _for_blocks(work, blocks):
    """work(block) for every block, on a small thread pool (NumPy releases the GIL inside its
    loops; the blocks are disjoint rows)."""
    global _POOL
    if len(blocks) < 4:
        for blk in blocks:
            work(blk)
        return
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) - 1)))
    list(_POOL.map(work, blocks))


_NATIVE_HOST = None


def _native_ok(*arrays):
    """The [B, P] host routines of libqocx (clip, Adam / SGD on host threads) can take these
    arrays. Without the library (a CPU box running the host logic against the oracle backend)
    the NumPy block path below each call site does the same arithmetic."""
    global _NATIVE_HOST
    if _NATIVE_HOST is None:
        from qoc_amd import engine
        try:
            engine.load_library()
            _NATIVE_HOST = True
        except ImportError:  # the library FILE is absent: a CPU box running the host logic only
            _NATIVE_HOST = False
        # (a library that is present but does not load, or lacks a declared symbol, is a broken
        # build: OSError / AttributeError propagate instead of silently taking the NumPy path)
    return _NATIVE_HOST and all(a.dtype == np.float64 and a.flags.c_contiguous for a in arrays)


class _BatchedSGD(object):
    def __init__(self, optimizer, params):
        self.learning_rate = optimizer.learning_rate

    def update(self, grads, params, rows):
        """params[rows] <- SGD.update(grads[rows], params[rows]), in place."""
        if _native_ok(grads, params):
            from qoc_amd import engine
            engine.host_optimizer_update(0, params, grads, None, None, rows, self.learning_rate)
            return

        def work(r):
            params[r] = params[r] - self.learning_rate * grads[r]
        _for_blocks(work, _row_blocks(rows))


class _BatchedAdam(object):
    def __init__(self, optimizer, params):
        self.o = optimizer
        self.m = np.zeros_like(params)
        self.v = np.zeros_like(params)
        self.count = np.zeros(params.shape[0], dtype=np.int64)

    def update(self, grads, params, rows):
        """params[rows] <- Adam.update(grads[rows], params[rows]) with every seed's own moments and
        step counter, in place: the operations of adam.py:110-165 in their order, element by
        element (a * x + b * y is formed as (a * x) + (b * y) there and here)."""
        o = self.o
        for count in np.unique(self.count[rows]):  # (seeds that stopped earlier keep their own)
            sel = rows[self.count[rows] == count]
            if o.apply_learning_rate_decay:
                learning_rate = (o.initial_learning_rate
                                 * np.exp(-np.divide(count, o.learning_rate_decay)))
            else:
                learning_rate = o.initial_learning_rate
            step = count + 1
            corr_1 = 1 - np.power(o.beta_1, step)
            corr_2 = 1 - np.power(o.beta_2, step)
            if not o.apply_scale_grads and _native_ok(grads, params, self.m, self.v):
                # the same operations on host threads of libqocx (qocx_host_optimizer_update)
                from qoc_amd import engine
                engine.host_optimizer_update(
                    1, params, grads, self.m, self.v, sel, learning_rate, o.beta_1, o.beta_2,
                    o.epsilon, corr_1, corr_2, o.clip_grads if o.apply_clip_grads else None)
                self.count[sel] = step
                continue

            def work(r):
                g = grads[r]
                if o.apply_scale_grads:
                    norms = np.array([np.linalg.norm(row) for row in g])
                    g = (g / norms[:, None]) * o.scale_grads
                if o.apply_clip_grads:
                    g = np.clip(g, -o.clip_grads, o.clip_grads)
                m, v = self.m[r], self.v[r]  # views for a slice, copies for an index array
                t = np.multiply(g, 1 - o.beta_1)
                np.multiply(m, o.beta_1, out=m)
                m += t                                    # gradient_moment
                np.square(g, out=t)
                t *= (1 - o.beta_2)
                np.multiply(v, o.beta_2, out=v)
                v += t                                    # gradient_square_moment
                if not isinstance(r, slice):
                    self.m[r] = m
                    self.v[r] = v
                u = np.divide(v, corr_2)                  # square_hat
                np.sqrt(u, out=u)
                u += o.epsilon
                np.divide(m, corr_1, out=t)               # moment_hat
                np.divide(t, u, out=t)
                t *= learning_rate
                if isinstance(r, slice):
                    p = params[r]
                    p -= t
                else:
                    params[r] = params[r] - t
            _for_blocks(work, _row_blocks(sel))
            self.count[sel] = step


def _batched_stepper(optimizer, params):
    """The [B, P] form of the built-in step-wise optimizers; None for any other plugin (those keep
    one deep copy per seed and their own update())."""
    if type(optimizer) is Adam:
        return _BatchedAdam(optimizer, params)
    if type(optimizer) is SGD:
        return _BatchedSGD(optimizer, params)
    return None


def _cost_format_batch(params, pstate):
    """_cost_format on all rows of params [B, P] at once: (B x Nc x K) controls, clipped in place
    (a view of params for real controls, as in the single-seed driver), conditions per seed."""
    B = params.shape[0]
    shape = (B,) + tuple(pstate.controls_shape)
    if pstate.complex_controls:
        half = params.shape[1] // 2
        controls = (params[:, :half] + 1j * params[:, half:]).reshape(shape)
    else:
        controls = params.reshape(shape)
        if _native_ok(params) and controls.base is not None:
            from qoc_amd import engine
            engine.host_clip_controls(controls, pstate.max_control_norms)
            if pstate.impose_control_conditions is not None:
                controls = np.stack([pstate.impose_control_conditions(controls[b])
                                     for b in range(B)])
            return controls
    for i, max_norm in enumerate(pstate.max_control_norms):  # clip_control_norms, all seeds
        column = controls[:, :, i]
        moduli = np.abs(column)
        over = np.less(max_norm, moduli)
        if over.any():
            column[over] = (column[over] / moduli[over]) * max_norm
    if pstate.impose_control_conditions is not None:
        controls = np.stack([pstate.impose_control_conditions(controls[b]) for b in range(B)])
    return controls


def _strip_batch(complex_controls, arrays):
    """strip_controls on every row: (B x Nc x K) -> [B, P]."""
    flat = np.reshape(arrays, (arrays.shape[0], -1))
    if complex_controls:
        flat = np.hstack((np.real(flat), np.imag(flat)))
    return flat


def _grape_batch_resident(engine, optimizer, params, shape, max_control_norms, iteration_count,
                          log_iteration_step, min_error, comm, result):
    """The loop of grape_schroedinger_discrete_batch with everything but the decisions on the
    device: engine.opt_clip -> eval_resident -> B costs to the host -> engine.opt_step."""
    B = params.shape[0]
    is_adam = type(optimizer) is Adam
    engine.upload_controls(params.reshape((B,) + tuple(shape)))
    engine.opt_begin()
    active = np.ones(B, dtype=bool)
    count = 0  # optimizer steps taken so far (every active seed has taken all of them)
    should_log = log_iteration_step != 0
    for iteration in range(iteration_count):
        engine.opt_clip(max_control_norms)
        engine.eval_resident(True)
        errors = engine.download_costs()
        result.iterations_run[active] = iteration + 1
        improved = active & (errors < result.best_error)  # strict, as :333 of the reference
        result.best_error[improved] = errors[improved]
        result.best_iteration[improved] = iteration
        if should_log and (iteration % log_iteration_step == 0 or iteration == iteration_count - 1):
            total = comm.allreduce_sum(np.array([float(np.sum(errors[active]))]))[0]
            low = -comm.allreduce_max(np.array([-float(np.min(errors))]))[0]
            seeds = comm.allreduce_sum(np.array([float(np.sum(active))]))[0]
            if comm.rank == 0:
                print("{:^6d} | {:^1.8e} | {:^1.8e} | {:^6d}".format(iteration, total, low,
                                                                     int(seeds)))
        active &= ~(errors <= min_error)
        if is_adam:
            o = optimizer
            if o.apply_learning_rate_decay:
                learning_rate = (o.initial_learning_rate
                                 * np.exp(-np.divide(count, o.learning_rate_decay)))
            else:
                learning_rate = o.initial_learning_rate
            step = count + 1
            engine.opt_step(1, improved, active, learning_rate, o.beta_1, o.beta_2, o.epsilon,
                            1 - np.power(o.beta_1, step), 1 - np.power(o.beta_2, step),
                            o.clip_grads if o.apply_clip_grads else None)
        else:
            engine.opt_step(0, improved, active, optimizer.learning_rate)
        count += 1
        still = comm.allreduce_sum(np.array([float(np.sum(active))]))[0]
        if still == 0:
            break
    best_controls, best_finals = engine.opt_download_best()
    for b in range(B):
        if result.best_iteration[b] >= 0:
            result.best_controls[b] = best_controls[b]
            result.best_final_states[b] = best_finals[b][..., None]
    best_local = float(np.min(result.best_error)) if B else np.inf
    result.global_best_error = float(-comm.allreduce_max(np.array([-best_local]))[0])
    return result


def grape_schroedinger_discrete_batch(control_count, control_eval_count, costs, evolution_time,
                                      hamiltonian, initial_states, system_eval_count,
                                      initial_controls, complex_controls=False, cost_eval_step=1,
                                      impose_control_conditions=None,
                                      interpolation_policy=InterpolationPolicy.LINEAR,
                                      iteration_count=1000, log_iteration_step=10,
                                      magnus_policy=MagnusPolicy.M2, max_control_norms=None,
                                      min_error=0, optimizer=Adam(), comm=None):
    """
    Multi-start GRAPE: B = len(initial_controls) independent optimisations of the same problem,
    one batched device evaluation per iteration (the engine's batch axis; the reference runs one
    control set per process). Seed b follows EXACTLY the iteration of
    grape_schroedinger_discrete (reference :293-353 per seed): clip -> conditions -> evaluate ->
    best-so-far (strict <) -> optimizer update, its own optimizer state (a deep copy of
    `optimizer`), its own termination at error <= min_error (a finished seed is frozen and no
    longer updated; the batch ends when every seed has finished or after iteration_count
    iterations).

    initial_controls :: (B x control_eval_count x control_count), each conforming to
    max_control_norms. comm (qoc_amd.parallel communicator, optional): the seed axis is sharded
    over its ranks, every rank optimises its own block, and the logged error is the all-reduced
    sum (the path's single collective); result arrays are rank local.
    Returns GrapeSchroedingerBatchResult.
    """
    from qoc_amd import parallel
    initial_controls = np.asarray(initial_controls)
    if initial_controls.ndim != 3:
        raise ValueError("initial_controls must be (seed_count x control_eval_count x "
                         "control_count), got shape {}".format(initial_controls.shape))
    comm = comm if comm is not None else parallel.SingleComm()
    lo, hi = parallel.shard_bounds(initial_controls.shape[0], comm.rank, comm.world)
    if max_control_norms is None:  # the default of initialize_controls, also for a rank without seeds
        max_control_norms = np.ones(control_count)
    seeds = []
    for b in range(lo, hi):
        controls_b, max_control_norms = initialize_controls(
            complex_controls, control_count, control_eval_count, evolution_time,
            initial_controls[b], max_control_norms)
        seeds.append(np.array(controls_b))
    B = len(seeds)
    shape = (control_eval_count, control_count)
    pstate = Dummy()
    pstate.complex_controls = complex_controls
    pstate.controls_shape = shape
    pstate.max_control_norms = max_control_norms
    pstate.impose_control_conditions = impose_control_conditions
    evaluator = SchroedingerEvaluator(
        evolution_time, hamiltonian, initial_states, system_eval_count,
        control_count=control_count, control_eval_count=control_eval_count,
        complex_controls=complex_controls, costs=costs, cost_eval_step=cost_eval_step,
        interpolation_policy=interpolation_policy, magnus_policy=magnus_policy,
        need_gradients=True, latency_mode=B <= 128)
    params = (np.stack([strip_controls(complex_controls, c) for c in seeds]) if B
              else np.zeros((0, control_eval_count * control_count)))
    params = np.array(params, dtype=np.float64)
    stepper = _batched_stepper(optimizer, params)
    optimizers = None if stepper is not None else [_optimizer_clone(optimizer, p) for p in params]
    active = np.ones(B, dtype=bool)
    result = GrapeSchroedingerBatchResult(B)
    best_controls = best_finals = None
    should_log = log_iteration_step != 0
    if should_log and comm.rank == 0:
        print("iter   |  summed error  |   min error    |  active seeds \n"
              "===========================================================")
    # Device-resident form: controls, gradients, Adam moments and the best
    # so far stay in HBM; per iteration the host sees B costs and sends 2 B flags. The same
    # iteration, the same arithmetic (qocx_optim.hip: IEEE operations in the reference's order).
    resident = (B > 0 and stepper is not None and impose_control_conditions is None
                and not getattr(optimizer, "apply_scale_grads", False)
                and hasattr(evaluator, "resident_capable") and evaluator.resident_capable())
    if resident:
        return _grape_batch_resident(evaluator.backend, optimizer, params, shape, max_control_norms,
                                     iteration_count, log_iteration_step, min_error, comm, result)
    for iteration in range(iteration_count):
        # cost-function format of every seed (clipping acts in place on the optimizer's params
        # for real controls, exactly as in the single-seed driver)
        controls = _cost_format_batch(params, pstate)
        if B > 0:
            errors, grads, finals, _ = evaluator.evaluate_batch(controls, want_grad=True)
        else:
            errors, grads, finals = np.zeros(0), np.zeros((0,) + shape), np.zeros(0)
        errors = np.asarray(errors)
        result.iterations_run[active] = iteration + 1
        improved = active & (errors < result.best_error)  # strict, as :333 of the reference
        if improved.any():
            if best_controls is None:
                best_controls = np.zeros_like(controls)
                best_finals = np.zeros_like(finals)
            best_controls[improved] = controls[improved]
            best_finals[improved] = finals[improved]
            result.best_error[improved] = errors[improved]
            result.best_iteration[improved] = iteration
        if should_log and (iteration % log_iteration_step == 0 or iteration == iteration_count - 1):
            local = np.array([float(np.sum(errors[active])) if B else 0.0])
            total = comm.allreduce_sum(local)[0]
            low = -comm.allreduce_max(np.array([-float(np.min(errors)) if B else -np.inf]))[0]
            count = comm.allreduce_sum(np.array([float(np.sum(active))]))[0]
            if comm.rank == 0:
                print("{:^6d} | {:^1.8e} | {:^1.8e} | {:^6d}".format(iteration, total, low,
                                                                     int(count)))
        # a seed at error <= min_error ends its optimizer loop (terminate = True): frozen from here
        active &= ~(errors <= min_error)
        rows = np.nonzero(active)[0]
        if len(rows):
            flat_grads = _strip_batch(complex_controls, np.asarray(grads))
            if stepper is not None:
                stepper.update(flat_grads, params, rows)
            else:
                for b in rows:
                    params[b] = optimizers[b].update(flat_grads[b], params[b])
        still = comm.allreduce_sum(np.array([float(np.sum(active))]))[0]
        if still == 0:
            break
    for b in range(B):
        if result.best_iteration[b] >= 0:
            result.best_controls[b] = best_controls[b]
            result.best_final_states[b] = best_finals[b]
    best_local = float(np.min(result.best_error)) if B else np.inf
    result.global_best_error = float(-comm.allreduce_max(np.array([-best_local]))[0])
    return result
