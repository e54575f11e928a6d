"""
engine.py - thin ctypes binding of libqocx.so (include/qocx.h): NumPy in, NumPy out.

This is the only way the package reaches the hot path, and there is no CPU fallback: if the
HIP library is missing or no MI355X is visible, construction fails loudly.
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, "libqocx.so")
# the measurement build (make -C qoc_amd/csrc diag): loaded by the scripts under tools/ only
DIAG_LIBRARY_PATH = os.path.join(_HERE, "libqocx_diag.so")

COST_TARGET_COHERENT = 0
COST_TARGET_INCOHERENT = 1
COST_FORBID = 2
COST_TARGET_DENSITY = 3
COST_FORBID_DENSITY = 4

MAGNUS_CODES = {"M2": 2, "M4": 4, "M6": 6}

ERR_SINGULAR = -4

KERNEL_NAMES = ("pade_pq", "sweep", "krylov_grad", "scatter", "lu", "lindblad", "lindblad_combine")

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int_p = ctypes.POINTER(ctypes.c_int32)


def host_clip_controls(controls, max_norms):
    """clip_control_norms (qoc/core/common.py:8-30) on a C-contiguous float64 (B x Nc x K) array,
    IN PLACE, on host threads (qocx_host_clip_controls; no GPU involved)."""
    lib = load_library()
    max_norms = np.ascontiguousarray(max_norms, dtype=np.float64)
    rc = lib.qocx_host_clip_controls(controls.ctypes.data_as(_c_double_p), controls.shape[0],
                                     controls.shape[1], controls.shape[2],
                                     max_norms.ctypes.data_as(_c_double_p))
    if rc != 0:
        raise QocxError(rc, "qocx_host_clip_controls")


def host_optimizer_update(kind, params, grads, moment, square_moment, rows, learning_rate,
                          beta_1=0.0, beta_2=0.0, epsilon=0.0, corr_1=1.0, corr_2=1.0,
                          clip_grads=None):
    """Adam (kind 1) / SGD (kind 0) update of the rows `rows` of C-contiguous float64 [B][P]
    arrays, in place, in the reference's operation order (qocx_host_optimizer_update)."""
    lib = load_library()
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    none = ctypes.cast(None, _c_double_p)
    rc = lib.qocx_host_optimizer_update(
        kind, params.ctypes.data_as(_c_double_p), grads.ctypes.data_as(_c_double_p),
        moment.ctypes.data_as(_c_double_p) if moment is not None else none,
        square_moment.ctypes.data_as(_c_double_p) if square_moment is not None else none,
        params.shape[1], rows.ctypes.data_as(ctypes.POINTER(_I64)), len(rows),
        float(learning_rate), float(beta_1), float(beta_2), float(epsilon), float(corr_1),
        float(corr_2), 0 if clip_grads is None else 1, 0.0 if clip_grads is None else float(clip_grads))
    if rc != 0:
        raise QocxError(rc, "qocx_host_optimizer_update")


class QocxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libqocx error {}: {}".format(code, message))
        self.code = code
        self.message = message


class _CostDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("step_cost", ctypes.c_int32),
                ("scale", ctypes.c_double), ("vectors", _c_double_p), ("counts", _c_int_p)]


class _SchroedingerProblem(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32),
                ("hilbert_size", ctypes.c_int32), ("state_count", ctypes.c_int32),
                ("control_count", ctypes.c_int32), ("control_eval_count", ctypes.c_int32),
                ("system_eval_count", ctypes.c_int32), ("cost_eval_step", ctypes.c_int32),
                ("magnus_policy", ctypes.c_int32), ("nt", ctypes.c_int32),
                ("evolution_time", ctypes.c_double), ("h0", _c_double_p), ("g", _c_double_p),
                ("initial_states", _c_double_p), ("cost_count", ctypes.c_int32),
                ("costs", ctypes.POINTER(_CostDesc))]


class _LindbladProblem(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_int32),
                ("hilbert_size", ctypes.c_int32), ("density_count", ctypes.c_int32),
                ("control_count", ctypes.c_int32), ("control_eval_count", ctypes.c_int32),
                ("system_eval_count", ctypes.c_int32), ("cost_eval_step", ctypes.c_int32),
                ("operator_count", ctypes.c_int32), ("evolution_time", ctypes.c_double),
                ("h0", _c_double_p), ("g", _c_double_p), ("dissipators", _c_double_p),
                ("operators", _c_double_p), ("initial_densities", _c_double_p),
                ("cost_count", ctypes.c_int32), ("costs", ctypes.POINTER(_CostDesc)),
                ("fixed_subdivision", ctypes.c_int32), ("h0_stages", _c_double_p),
                ("g_stages", _c_double_p), ("diss_stages", _c_double_p),
                ("op_stages", _c_double_p)]


# name -> (restype, argtypes); every symbol declared in include/qocx.h
_VP = ctypes.c_void_p
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64
_U8P = ctypes.POINTER(ctypes.c_uint8)
SIGNATURES = {
    "qocx_last_error": (ctypes.c_char_p, []),
    "qocx_version": (ctypes.c_int, []),
    "qocx_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "qocx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_VP)]),
    "qocx_destroy": (ctypes.c_int, [_VP]),
    "qocx_synchronize": (ctypes.c_int, [_VP]),
    "qocx_set_schroedinger_problem": (ctypes.c_int, [_VP, ctypes.POINTER(_SchroedingerProblem)]),
    "qocx_eval_schroedinger": (ctypes.c_int, [_VP, _I32, _c_double_p, _I32, _c_double_p,
                                              _c_double_p, _c_double_p]),
    "qocx_upload_controls": (ctypes.c_int, [_VP, _I32, _c_double_p]),
    "qocx_eval_resident": (ctypes.c_int, [_VP, _I32]),
    "qocx_download_results": (ctypes.c_int, [_VP, _c_double_p, _c_double_p, _c_double_p]),
    "qocx_upload_generators": (ctypes.c_int, [_VP, _I32, _c_double_p]),
    "qocx_download_generator_cotangents": (ctypes.c_int, [_VP, _c_double_p]),
    "qocx_set_keep_step_states": (ctypes.c_int, [_VP, _I32]),
    "qocx_download_step_states": (ctypes.c_int, [_VP, _c_double_p]),
    "qocx_set_lindblad_problem": (ctypes.c_int, [_VP, ctypes.POINTER(_LindbladProblem)]),
    "qocx_lindblad_stage_times": (ctypes.c_int, [ctypes.c_double, _I32, _I32, _I32, _I32,
                                                 _c_double_p, _I64, ctypes.POINTER(_I64)]),
    "qocx_eval_lindblad": (ctypes.c_int, [_VP, _I32, _c_double_p, _I32, _c_double_p,
                                          _c_double_p, _c_double_p]),
    "qocx_download_step_densities": (ctypes.c_int, [_VP, _c_double_p]),
    "qocx_lindblad_last_subintervals": (ctypes.c_int, [_VP, ctypes.POINTER(_I64)]),
    "qocx_set_density_cotangents": (ctypes.c_int, [_VP, _I32, _I32, _c_int_p, _c_double_p]),
    "qocx_set_state_cotangents": (ctypes.c_int, [_VP, _I32, _I32, _c_int_p, _c_double_p]),
    "qocx_set_timing": (ctypes.c_int, [_VP, _I32]),
    "qocx_get_timing": (ctypes.c_int, [_VP, _I32, ctypes.POINTER(_I64), _c_double_p]),
    "qocx_reset_timing": (ctypes.c_int, [_VP]),
    "qocx_set_chunk": (ctypes.c_int, [_VP, _I32]),
    "qocx_set_pipeline": (ctypes.c_int, [_VP, _I32]),
    "qocx_comm_unique_id": (ctypes.c_int, [_U8P]),
    "qocx_comm_init": (ctypes.c_int, [_VP, _U8P, _I32, _I32]),
    "qocx_reduce_results": (ctypes.c_int, [_VP, _I32, _c_double_p, _I64]),
    "qocx_comm_allreduce_sum": (ctypes.c_int, [_VP, _c_double_p, _I64]),
    "qocx_comm_allreduce_max": (ctypes.c_int, [_VP, _c_double_p, _I64]),
    "qocx_comm_barrier": (ctypes.c_int, [_VP]),
    "qocx_comm_destroy": (ctypes.c_int, [_VP]),
    "qocx_debug_pade_factor": (ctypes.c_int, [_VP, _I32, _I32, _c_double_p, _c_double_p,
                                              _c_double_p, _c_int_p, _c_double_p, _c_int_p]),
    "qocx_debug_selftest": (ctypes.c_int, [_VP, _c_int_p, ctypes.c_char_p, _I32]),
    "qocx_debug_lindblad_knobs": (ctypes.c_int, [_VP, _I64, _I32, _I32]),
    "qocx_debug_set_knob": (ctypes.c_int, [_VP, ctypes.c_char_p, _I64]),
    "qocx_knob_kind": (ctypes.c_int, [ctypes.c_char_p]),
    "qocx_build_is_diag": (ctypes.c_int, []),
    "qocx_debug_read_stamps": (ctypes.c_int, [_VP, ctypes.POINTER(ctypes.c_uint64), _I64]),
    "qocx_debug_timeline": (ctypes.c_int, [_VP, _c_double_p, _I64, ctypes.POINTER(_I64)]),
    "qocx_pade_orders": (ctypes.c_int, [_VP, ctypes.POINTER(_I64)]),
    "qocx_lu_fallbacks": (ctypes.c_int, [_VP, ctypes.POINTER(_I64)]),
    "qocx_debug_mfma_peak": (ctypes.c_int, [_VP, _I32, _I32, _c_double_p]),
    "qocx_opt_begin": (ctypes.c_int, [_VP]),
    "qocx_opt_clip": (ctypes.c_int, [_VP, _c_double_p]),
    "qocx_download_costs": (ctypes.c_int, [_VP, _c_double_p]),
    "qocx_opt_step": (ctypes.c_int, [
        _VP, _I32, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_uint8), ctypes.c_double,
        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I32,
        ctypes.c_double]),
    "qocx_opt_download_best": (ctypes.c_int, [_VP, _c_double_p, _c_double_p]),
    "qocx_host_clip_controls": (ctypes.c_int, [_c_double_p, _I64, _I64, _I32, _c_double_p]),
    "qocx_host_optimizer_update": (ctypes.c_int, [
        _I32, _c_double_p, _c_double_p, _c_double_p, _c_double_p, _I64, ctypes.POINTER(_I64), _I64,
        ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
        ctypes.c_double, _I32, ctypes.c_double]),
}

_lib = None


def load_library(path=None):
    """Load libqocx.so and bind every declared symbol. Raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIBRARY_PATH
    if not os.path.exists(path):
        raise ImportError(
            "qoc_amd: the HIP engine {} is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.".format(path))
    # Multi-process GPU work on this platform needs dmabuf IPC (RCCL between the ranks of a node
    # fails with hipIpcGetMemHandle: invalid argument otherwise). The HSA runtime reads
    # HSA_ENABLE_IPC_MODE_LEGACY when it initialises - which dlopen of libqocx does not trigger, the
    # first qocx_create does - so it is defaulted here, before any context exists: a value the user
    # (or the launcher, qoc_amd/parallel.py documents it) has set is kept, a process in which
    # another library initialised HSA earlier is not affected, and child processes inherit it.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _dp(array):
    return array.ctypes.data_as(_c_double_p)


def _as_complex(array, shape=None):
    out = np.ascontiguousarray(array, dtype=np.complex128)
    if shape is not None:
        out = out.reshape(shape)
    return out


class Engine(object):
    """One context on one MI355X."""

    @staticmethod
    def device_count():
        """HIP devices visible to this process (qocx_device_count)."""
        lib = load_library()
        count = ctypes.c_int(0)
        code = lib.qocx_device_count(ctypes.byref(count))
        if code != 0:
            raise QocxError(code, lib.qocx_last_error().decode("utf-8", "replace"))
        return int(count.value)

    def __init__(self, device=-1):
        self._lib = load_library()
        self._ctx = _VP()
        self._check(self._lib.qocx_create(int(device), ctypes.byref(self._ctx)))
        self._problem = None
        self._keepalive = []
        self.batch = 0

    # -- plumbing ----------------------------------------------------------------------------
    def _check(self, code):
        if code != 0:
            msg = self._lib.qocx_last_error().decode("utf-8", "replace")
            if code == ERR_SINGULAR:
                raise np.linalg.LinAlgError(msg)
            raise QocxError(code, msg)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value is not None:
            self._lib.qocx_destroy(self._ctx)
            self._ctx = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._check(self._lib.qocx_synchronize(self._ctx))

    # -- problem -------------------------------------------------------------------------------
    def set_schroedinger_problem(self, hilbert_size, state_count, control_count,
                                 control_eval_count, system_eval_count, evolution_time,
                                 h0, g, initial_states, costs=(), cost_eval_step=1,
                                 magnus_policy="M2"):
        """
        h0 :: (nt, n, n) complex, g :: (nt, K, n, n) complex, initial_states :: (S, n) complex,
        costs :: iterable of dicts {kind, step_cost, scale, vectors, counts(optional)}.
        """
        n, S, K = int(hilbert_size), int(state_count), int(control_count)
        h0 = _as_complex(h0)
        if h0.ndim == 2:
            h0 = h0[None]
        nt = h0.shape[0]
        h0 = _as_complex(h0, (nt, n, n))
        g = _as_complex(g if K > 0 else np.zeros((nt, 0, n, n)), (nt, K, n, n))
        psi = _as_complex(initial_states, (S, n))
        keep = [h0, g, psi]
        descs = (_CostDesc * max(1, len(costs)))()
        for i, c in enumerate(costs):
            vec = _as_complex(c["vectors"])
            vec = vec.reshape(-1, n)
            keep.append(vec)
            descs[i].kind = int(c["kind"])
            descs[i].step_cost = int(bool(c["step_cost"]))
            descs[i].scale = float(c["scale"])
            descs[i].vectors = _dp(vec)
            if c.get("counts") is not None:
                cnt = np.ascontiguousarray(c["counts"], dtype=np.int32)
                keep.append(cnt)
                descs[i].counts = cnt.ctypes.data_as(_c_int_p)
        p = _SchroedingerProblem()
        p.struct_size = ctypes.sizeof(_SchroedingerProblem)
        p.hilbert_size, p.state_count, p.control_count = n, S, K
        p.control_eval_count, p.system_eval_count = int(control_eval_count), int(system_eval_count)
        p.cost_eval_step = int(cost_eval_step)
        p.magnus_policy = MAGNUS_CODES[magnus_policy]
        p.nt = nt
        p.evolution_time = float(evolution_time)
        p.h0, p.g, p.initial_states = _dp(h0), _dp(g), _dp(psi)
        p.cost_count = len(costs)
        p.costs = descs
        self._check(self._lib.qocx_set_schroedinger_problem(self._ctx, ctypes.byref(p)))
        self._problem = dict(n=n, S=S, K=K, Nc=int(control_eval_count), N=int(system_eval_count))
        self.batch = 0

    # -- evaluation ----------------------------------------------------------------------------
    def upload_controls(self, controls):
        pr = self._problem
        if pr["K"] > 0:
            controls = np.ascontiguousarray(controls, dtype=np.float64)
            controls = controls.reshape(-1, pr["Nc"], pr["K"])
            batch = controls.shape[0]
            self._check(self._lib.qocx_upload_controls(self._ctx, batch, _dp(controls)))
        else:
            batch = 1 if controls is None else int(controls)
            self._check(self._lib.qocx_upload_controls(self._ctx, batch, None))
        self.batch = batch

    def upload_generators(self, generators):
        """generators :: (B, N-1, n, n) complex step generators M_j = -i dt H (opaque Hamiltonians);
        the problem must have been set with control_count = 0 and magnus_policy M2."""
        pr = self._problem
        gens = _as_complex(generators).reshape(-1, pr["N"] - 1, pr["n"], pr["n"])
        self._check(self._lib.qocx_upload_generators(self._ctx, gens.shape[0], _dp(gens)))
        self.batch = gens.shape[0]

    def download_generator_cotangents(self):
        pr = self._problem
        out = np.empty((self.batch, pr["N"] - 1, pr["n"], pr["n"]), dtype=np.complex128)
        self._check(self._lib.qocx_download_generator_cotangents(self._ctx, _dp(out)))
        return out

    def eval_resident(self, want_grad=True):
        self._check(self._lib.qocx_eval_resident(self._ctx, int(bool(want_grad))))

    def download_results(self, want_grad=True, want_final=True):
        pr, B = self._problem, self.batch
        cost = np.empty(B, dtype=np.float64)
        want_grad = want_grad and pr["K"] > 0
        grads = np.empty((B, pr["Nc"], pr["K"]), dtype=np.float64) if want_grad else None
        final = np.empty((B, pr["S"], pr["n"]), dtype=np.complex128) if want_final else None
        self._check(self._lib.qocx_download_results(
            self._ctx, _dp(cost), _dp(grads) if want_grad else None,
            _dp(final) if want_final else None))
        return cost, grads, final

    def evaluate(self, controls, want_grad=True):
        """(cost[B], grads[B,Nc,K] or None, final_states[B,S,n]) for controls[B,Nc,K]."""
        self.upload_controls(controls)
        self.eval_resident(want_grad)
        return self.download_results(want_grad)

    def set_keep_step_states(self, keep):
        self._check(self._lib.qocx_set_keep_step_states(self._ctx, int(bool(keep))))

    def download_step_states(self):
        pr, B = self._problem, self.batch
        out = np.empty((B, pr["N"], pr["S"], pr["n"]), dtype=np.complex128)
        self._check(self._lib.qocx_download_step_states(self._ctx, _dp(out)))
        return out

    # -- Lindblad -------------------------------------------------------------------------------
    @staticmethod
    def lindblad_stage_times(evolution_time, system_eval_count, control_eval_count, control_count,
                             subdivision):
        """Times at which a time-dependent Hamiltonian is sampled for `subdivision` pieces per
        system step (no GPU needed)."""
        lib = load_library()
        count = _I64(0)
        args = (float(evolution_time), int(system_eval_count), int(control_eval_count),
                int(control_count), int(subdivision))
        code = lib.qocx_lindblad_stage_times(*args, None, 0, ctypes.byref(count))
        if code:
            raise QocxError(code, lib.qocx_last_error().decode("utf-8", "replace"))
        times = np.empty(count.value, dtype=np.float64)
        code = lib.qocx_lindblad_stage_times(*args, _dp(times), count.value, ctypes.byref(count))
        if code:
            raise QocxError(code, lib.qocx_last_error().decode("utf-8", "replace"))
        return times

    def set_lindblad_problem(self, hilbert_size, density_count, control_count,
                             control_eval_count, system_eval_count, evolution_time,
                             h0, g, dissipators, operators, initial_densities, costs=(),
                             cost_eval_step=1, fixed_subdivision=0, h0_stages=None,
                             g_stages=None, diss_stages=None, op_stages=None):
        """
        h0 :: (n, n), g :: (K, n, n), dissipators :: (L,), operators :: (L, n, n),
        initial_densities :: (S, n, n); costs :: dicts {kind (3|4), step_cost, scale,
        vectors (matrices), counts(optional)}.
        """
        n, S, K = int(hilbert_size), int(density_count), int(control_count)
        h0 = _as_complex(h0, (n, n))
        g = _as_complex(g if K > 0 else np.zeros((0, n, n)), (K, n, n))
        L = 0 if operators is None else len(operators)
        ops = _as_complex(operators if L > 0 else np.zeros((0, n, n)), (L, n, n))
        gam = np.ascontiguousarray(dissipators if L > 0 else np.zeros(0), dtype=np.float64)
        gam = gam.reshape(L)
        rho = _as_complex(initial_densities, (S, n, n))
        keep = [h0, g, ops, gam, rho]
        descs = (_CostDesc * max(1, len(costs)))()
        for i, c in enumerate(costs):
            mats = _as_complex(c["vectors"]).reshape(-1, n, n)
            keep.append(mats)
            descs[i].kind = int(c["kind"])
            descs[i].step_cost = int(bool(c["step_cost"]))
            descs[i].scale = float(c["scale"])
            descs[i].vectors = _dp(mats)
            if c.get("counts") is not None:
                cnt = np.ascontiguousarray(c["counts"], dtype=np.int32)
                keep.append(cnt)
                descs[i].counts = cnt.ctypes.data_as(_c_int_p)
        p = _LindbladProblem()
        p.struct_size = ctypes.sizeof(_LindbladProblem)
        p.hilbert_size, p.density_count, p.control_count = n, S, K
        p.control_eval_count, p.system_eval_count = int(control_eval_count), int(system_eval_count)
        p.cost_eval_step = int(cost_eval_step)
        p.operator_count = L
        p.evolution_time = float(evolution_time)
        p.h0, p.g, p.initial_densities = _dp(h0), _dp(g), _dp(rho)
        p.dissipators, p.operators = _dp(gam), _dp(ops)
        p.cost_count = len(costs)
        p.costs = descs
        p.fixed_subdivision = int(fixed_subdivision)
        if fixed_subdivision:
            hs = _as_complex(h0_stages).reshape(-1, n, n)
            keep.append(hs)
            p.h0_stages = _dp(hs)
            if g_stages is not None and K > 0:
                gs = _as_complex(g_stages).reshape(-1, K, n, n)
                keep.append(gs)
                p.g_stages = _dp(gs)
            if op_stages is not None and L > 0:  # time-dependent lindblad_data
                ds = np.ascontiguousarray(diss_stages, dtype=np.float64).reshape(-1, L)
                os_ = _as_complex(op_stages).reshape(-1, L, n, n)
                keep.extend([ds, os_])
                p.diss_stages, p.op_stages = _dp(ds), _dp(os_)
        self._check(self._lib.qocx_set_lindblad_problem(self._ctx, ctypes.byref(p)))
        self._lindblad = dict(n=n, S=S, K=K, Nc=int(control_eval_count),
                              N=int(system_eval_count))
        self._lindblad_batch = 0

    def evaluate_lindblad(self, controls, want_grad=True, want_final=True):
        """(cost[B], grads[B,Nc,K] or None, final_densities[B,S,n,n]) for controls[B,Nc,K]."""
        pr = self._lindblad
        if pr["K"] > 0:
            controls = np.ascontiguousarray(controls, dtype=np.float64)
            controls = controls.reshape(-1, pr["Nc"], pr["K"])
            B = controls.shape[0]
        else:
            B = 1 if controls is None else int(controls)
            controls = None
        cost = np.empty(B, dtype=np.float64)
        want_grad = bool(want_grad) and pr["K"] > 0
        grads = np.empty((B, pr["Nc"], pr["K"]), dtype=np.float64) if want_grad else None
        final = (np.empty((B, pr["S"], pr["n"], pr["n"]), dtype=np.complex128)
                 if want_final else None)
        self._check(self._lib.qocx_eval_lindblad(
            self._ctx, B, _dp(controls) if controls is not None else None, int(want_grad),
            _dp(cost), _dp(grads) if want_grad else None, _dp(final) if want_final else None))
        self._lindblad_batch = B
        return cost, grads, final

    def lindblad_last_subintervals(self):
        """DOP853 sub-intervals of the last evaluate_lindblad, summed over its seeds."""
        total = _I64(0)
        self._check(self._lib.qocx_lindblad_last_subintervals(self._ctx, ctypes.byref(total)))
        return total.value

    def set_density_cotangents(self, steps, bars):
        """bars :: (B, len(steps), S, n, n) complex cotangents of the densities at `steps`."""
        if steps is None or len(steps) == 0:
            self._check(self._lib.qocx_set_density_cotangents(self._ctx, 0, 0, None, None))
            return
        pr = self._lindblad
        steps = np.ascontiguousarray(steps, dtype=np.int32)
        bars = _as_complex(bars).reshape(-1, len(steps), pr["S"], pr["n"], pr["n"])
        self._check(self._lib.qocx_set_density_cotangents(
            self._ctx, bars.shape[0], len(steps), steps.ctypes.data_as(_c_int_p), _dp(bars)))

    def download_step_densities(self):
        pr, B = self._lindblad, self._lindblad_batch
        out = np.empty((B, pr["N"], pr["S"], pr["n"], pr["n"]), dtype=np.complex128)
        self._check(self._lib.qocx_download_step_densities(self._ctx, _dp(out)))
        return out

    def set_state_cotangents(self, steps, bars):
        """bars :: (B, len(steps), S, n) complex cotangents of the states at system steps `steps`;
        steps = None or empty clears them."""
        if steps is None or len(steps) == 0:
            self._check(self._lib.qocx_set_state_cotangents(self._ctx, 0, 0, None, None))
            return
        pr = self._problem
        steps = np.ascontiguousarray(steps, dtype=np.int32)
        bars = _as_complex(bars).reshape(-1, len(steps), pr["S"], pr["n"])
        self._check(self._lib.qocx_set_state_cotangents(
            self._ctx, bars.shape[0], len(steps), steps.ctypes.data_as(_c_int_p), _dp(bars)))

    def set_chunk(self, seeds_per_chunk):
        self._check(self._lib.qocx_set_chunk(self._ctx, int(seeds_per_chunk)))

    def set_pipeline(self, time_segments):
        """Number of time segments the evaluation pipeline is cut into (0 = automatic)."""
        self._check(self._lib.qocx_set_pipeline(self._ctx, int(time_segments)))

    def pade_orders(self):
        """{order: propagator steps} of the last evaluation (include/qocx.h: qocx_pade_orders)."""
        counts = (_I64 * 5)()
        self._check(self._lib.qocx_pade_orders(self._ctx, counts))
        return {order: int(counts[i]) for i, order in enumerate((3, 5, 7, 9, 13))}

    def lu_fallbacks(self):
        """Matrices of the last evaluation / debug_pade_factor call whose factorisation left the
        diagonal-pivot MFMA form for the general elimination (include/qocx.h: qocx_lu_fallbacks)."""
        count = _I64(0)
        self._check(self._lib.qocx_lu_fallbacks(self._ctx, ctypes.byref(count)))
        return int(count.value)

    def timeline(self, capacity=4096):
        """(which, start_ms, end_ms) of the last evaluation's kernel launches (timing on)."""
        out = np.zeros((capacity, 3))
        count = _I64(0)
        self._check(self._lib.qocx_debug_timeline(
            self._ctx, out.ctypes.data_as(_c_double_p), capacity, ctypes.byref(count)))
        return out[:min(capacity, count.value)]

    # -- timing --------------------------------------------------------------------------------
    def set_timing(self, enable, only=None):
        """enable: every launch carries HIP events; only = a name of KERNEL_NAMES: that kernel only."""
        mode = (2 + KERNEL_NAMES.index(only)) if (enable and only) else int(bool(enable))
        self._check(self._lib.qocx_set_timing(self._ctx, mode))

    def reset_timing(self):
        self._check(self._lib.qocx_reset_timing(self._ctx))

    def timing(self):
        out = {}
        for which, name in enumerate(KERNEL_NAMES):
            launches, ms = _I64(0), ctypes.c_double(0)
            self._check(self._lib.qocx_get_timing(self._ctx, which, ctypes.byref(launches),
                                                  ctypes.byref(ms)))
            out[name] = (launches.value, ms.value)
        return out

    # -- RCCL ----------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = (ctypes.c_uint8 * 128)()
        code = lib.qocx_comm_unique_id(buf)
        if code != 0:
            raise QocxError(code, lib.qocx_last_error().decode("utf-8", "replace"))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.qocx_comm_init(self._ctx, buf, int(rank), int(world)))

    # -- multi-start driver with the optimizer states on the device -----------------------------
    def opt_begin(self):
        self._check(self._lib.qocx_opt_begin(self._ctx))

    def opt_clip(self, max_norms):
        max_norms = np.ascontiguousarray(max_norms, dtype=np.float64)
        self._check(self._lib.qocx_opt_clip(self._ctx, _dp(max_norms)))

    def download_costs(self):
        cost = np.empty(self.batch, dtype=np.float64)
        self._check(self._lib.qocx_download_costs(self._ctx, _dp(cost)))
        return cost

    def opt_step(self, kind, improved, update, learning_rate, beta_1=0.0, beta_2=0.0, epsilon=0.0,
                 corr_1=1.0, corr_2=1.0, clip_grads=None):
        improved = np.ascontiguousarray(improved, dtype=np.uint8)
        update = np.ascontiguousarray(update, dtype=np.uint8)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        self._check(self._lib.qocx_opt_step(
            self._ctx, int(kind), improved.ctypes.data_as(u8), update.ctypes.data_as(u8),
            float(learning_rate), float(beta_1), float(beta_2), float(epsilon), float(corr_1),
            float(corr_2), 0 if clip_grads is None else 1,
            0.0 if clip_grads is None else float(clip_grads)))

    def opt_download_best(self):
        pr, B = self._problem, self.batch
        controls = np.empty((B, pr["Nc"], pr["K"]), dtype=np.float64)
        final = np.empty((B, pr["S"], pr["n"]), dtype=np.complex128)
        self._check(self._lib.qocx_opt_download_best(self._ctx, _dp(controls), _dp(final)))
        return controls, final

    def reduce_results(self, allreduce=False, want_grad=True):
        """(sum of the costs, sum of the gradients [Nc x K] or None) over the seeds of the last
        evaluation, summed on the device and - allreduce=True - over the ranks of the communicator
        by one ncclAllReduce on the device buffer (qocx_reduce_results)."""
        nc, k = self._problem["Nc"], self._problem["K"]
        want_grad = want_grad and k > 0
        count = 1 + (nc * k if want_grad else 0)
        out = np.zeros(count)
        self._check(self._lib.qocx_reduce_results(self._ctx, 1 if allreduce else 0, _dp(out), count))
        return float(out[0]), (out[1:].reshape(nc, k) if want_grad else None)

    def comm_allreduce_sum(self, array):
        array = np.ascontiguousarray(array, dtype=np.float64)
        self._check(self._lib.qocx_comm_allreduce_sum(self._ctx, _dp(array), array.size))
        return array

    def comm_allreduce_max(self, array):
        array = np.ascontiguousarray(array, dtype=np.float64)
        self._check(self._lib.qocx_comm_allreduce_max(self._ctx, _dp(array), array.size))
        return array

    def comm_barrier(self):
        self._check(self._lib.qocx_comm_barrier(self._ctx))

    def comm_destroy(self):
        self._check(self._lib.qocx_comm_destroy(self._ctx))

    # -- debug ---------------------------------------------------------------------------------
    def debug_pade_factor(self, a):
        a = _as_complex(a)
        if a.ndim == 2:
            a = a[None]
        count, n = a.shape[0], a.shape[1]
        q = np.empty((count, n, n), dtype=np.complex128)
        lu = np.empty((count, n, n), dtype=np.complex128)
        perm = np.empty((count, n), dtype=np.int32)
        dinv = np.empty((count, n), dtype=np.complex128)
        s = np.empty(count, dtype=np.int32)
        self._check(self._lib.qocx_debug_pade_factor(
            self._ctx, count, n, _dp(a), _dp(q), _dp(lu), perm.ctypes.data_as(_c_int_p),
            _dp(dinv), s.ctypes.data_as(_c_int_p)))
        # entry of the step table: squarings in bits 0..7, Pade order in bits 8..15 (0: 13)
        order = (s >> 8) & 0xff
        return dict(q=q, lu=lu, perm=perm, dinv=dinv, s=s & 0xff, order=np.where(order == 0, 13, order))

    def set_knob(self, name, value):
        """Kernel-variant switch (include/qocx.h: qocx_debug_set_knob)."""
        self._check(self._lib.qocx_debug_set_knob(self._ctx, name.encode(), int(value)))

    def read_stamps(self, batch, roles=4):
        """[batch][roles][8] cycle sums of a stamped diagnostic kernel build (knobs sweep3_stamps:
        4 roles; lindblad_stamps: 6 wavefronts)."""
        out = np.zeros((batch, roles, 8), dtype=np.uint64)
        self._check(self._lib.qocx_debug_read_stamps(
            self._ctx, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), out.size))
        return out

    def debug_lindblad_knobs(self, stage_budget_seeds=0, min_piece=256, wave_mode=0):
        """Force the Lindblad launch variants (piece-wise, recompute, one / several waves per
        seed); see include/qocx.h. Defaults restore the automatic choice."""
        self._check(self._lib.qocx_debug_lindblad_knobs(
            self._ctx, int(stage_budget_seeds), int(min_piece), int(wave_mode)))

    def mfma_peak(self, waves_per_simd=1, iters=20000):
        """Sustained FP64 MFMA TFLOP/s of a register-only MFMA loop (roofline calibration)."""
        out = ctypes.c_double(0)
        self._check(self._lib.qocx_debug_mfma_peak(self._ctx, int(waves_per_simd), int(iters),
                                                   ctypes.byref(out)))
        return out.value

    def selftest(self):
        failures = ctypes.c_int32(0)
        report = ctypes.create_string_buffer(4096)
        self._check(self._lib.qocx_debug_selftest(self._ctx, ctypes.byref(failures), report, 4096))
        return failures.value, report.value.decode("utf-8", "replace")
