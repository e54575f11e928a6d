// qocx_api.hip - host side of the C ABI declared in include/qocx.h: context, device memory,
// problem upload (what GrapeSchroedingerDiscreteState prepares for the evolve loop,
// qoc/models/programstate.py:33-61), the batched evaluation driver and the RCCL shim.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../../include/qocx.h"
#include "dop853_tableau.h"
#include "qocx_device.h"
#include "qocx_diag.h"

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(QOCX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t count = 0;
    int ensure(size_t n) {
        if (n <= count && p != nullptr) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
        if (n == 0) return 0;
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e != hipSuccess) {
            g_error = std::string("hipMalloc(") + std::to_string(n * sizeof(T)) +
                      " bytes): " + hipGetErrorString(e);
            return QOCX_ERR_HIP;
        }
        count = n;
        return 0;
    }
    int upload(const std::vector<T>& v, hipStream_t st) {
        int rc = ensure(v.size());
        if (rc) return rc;
        if (v.empty()) return 0;
        hipError_t e = hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return fail(QOCX_ERR_HIP, hipGetErrorString(e));
        e = hipStreamSynchronize(st);  // v may be a temporary
        if (e != hipSuccess) return fail(QOCX_ERR_HIP, hipGetErrorString(e));
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        count = 0;
    }
};

const double THETA13 = 5.371920351148152;

int pade_scale_count(double norm1) {
    int s = 0;
    double th = THETA13;
    while (norm1 > th && s < 1000) {
        th *= 2.0;
        ++s;
    }
    return s;
}

double one_norm(const double* m, int n) {  // complex row-major
    double best = 0;
    for (int c = 0; c < n; ++c) {
        double s = 0;
        for (int r = 0; r < n; ++r) s += hypot(m[2 * ((size_t)r * n + c)], m[2 * ((size_t)r * n + c) + 1]);
        best = std::max(best, s);
    }
    return best;
}

// Largest singular value of a complex n x n matrix (row-major, interleaved), for the step-size
// rule of the Lindblad integrator: power iteration on A^H A from a fixed start vector, stopped at
// 1e-4 relative change; the estimate comes from below, so 2 % are added, and it never exceeds the
// rigorous bound sqrt(||A||_1 ||A||_inf). (The 1-norm used before over-estimates the 2-norm of a
// dense Hermitian matrix by 2-3x, i.e. made the integrator take 2-3x more sub-intervals than the
// same threshold on the operator norm asks for.)
double two_norm(const double* m, int n) {
    double n1 = one_norm(m, n), ninf = 0;
    for (int r = 0; r < n; ++r) {
        double sum = 0;
        for (int c = 0; c < n; ++c) sum += hypot(m[2 * ((size_t)r * n + c)], m[2 * ((size_t)r * n + c) + 1]);
        ninf = std::max(ninf, sum);
    }
    const double upper = std::sqrt(n1 * ninf);
    if (!(upper > 0) || !(upper < 1e300)) return upper;
    std::vector<double> v(2 * n), w(2 * n);
    double nv = 0;
    for (int i = 0; i < n; ++i) {
        v[2 * i] = 1.0 + 0.37 * i / n;
        v[2 * i + 1] = 0.11 * ((i * 7) % 5);
        nv += v[2 * i] * v[2 * i] + v[2 * i + 1] * v[2 * i + 1];
    }
    nv = std::sqrt(nv);
    for (auto& e : v) e /= nv;
    double sigma = 0, prev = -1;
    for (int it = 0; it < 200; ++it) {
        double nw = 0;
        for (int r = 0; r < n; ++r) {  // w = A v
            double re = 0, im = 0;
            for (int c = 0; c < n; ++c) {
                const double ar = m[2 * ((size_t)r * n + c)], ai = m[2 * ((size_t)r * n + c) + 1];
                re += ar * v[2 * c] - ai * v[2 * c + 1];
                im += ar * v[2 * c + 1] + ai * v[2 * c];
            }
            w[2 * r] = re; w[2 * r + 1] = im;
            nw += re * re + im * im;
        }
        sigma = std::sqrt(nw);  // ||A v||, ||v|| = 1
        if (!(sigma > 0)) break;
        if (it >= 6 && std::fabs(sigma - prev) <= 1e-4 * sigma) break;
        prev = sigma;
        double nn = 0;
        for (int c = 0; c < n; ++c) {  // v = A^H w, normalised
            double re = 0, im = 0;
            for (int r = 0; r < n; ++r) {
                const double ar = m[2 * ((size_t)r * n + c)], ai = m[2 * ((size_t)r * n + c) + 1];
                re += ar * w[2 * r] + ai * w[2 * r + 1];
                im += ar * w[2 * r + 1] - ai * w[2 * r];
            }
            v[2 * c] = re; v[2 * c + 1] = im;
            nn += re * re + im * im;
        }
        nn = std::sqrt(nn);
        if (!(nn > 0)) break;
        for (auto& e : v) e /= nn;
    }
    return std::min(1.02 * sigma, upper);
}

struct TimingRec {
    int which;
    hipEvent_t a, b;
};

// RCCL entry points, resolved lazily so that single-GPU use never loads librccl.
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

}  // namespace

struct ncclUniqueIdBytes {
    char internal[128];
};

struct qocx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // ---- problem ----
    bool has_problem = false;
    int n = 0, nb = 0, np = 0, S = 0, K = 0, nc = 0, N = 0, nsteps = 0, ces = 1, nt = 1;
    double T = 0, dt = 0;
    int has_step_costs = 0, cost_count = 0;
    double h0_norm_max = 0;
    std::vector<double> g_norm_max;
    DevBuf<double2> h0_cimg, g_cimg, h0_rimg, g_rimg, h0_timg, g_timg, psi0, cost_vectors;
    DevBuf<qocx::StepInterp> interp;
    DevBuf<qocx::DevCost> costs;
    DevBuf<int> cost_counts, row_ptr, col_step;
    DevBuf<double> weight;
    // ---- evaluation state ----
    int B = 0;
    int sbound = 0;
    int last_chunk = 0;         // seeds of the last memory chunk of the last evaluation (its step table is in s_arr)
    double norm_bound = 1e300;  // host bound of ||step generator||_1 of the uploaded controls / generators
    // M2, control knots at the system times (Nc = N): a bound of the step generators at their MIDPOINTS,
    // where u is the mean of two knots - what the step table's per-step bound can reach at most; 1e300 when
    // it does not apply. Decides only whether the two-wave K1a is launched beside the three-wave one.
    double norm_bound_mid = 1e300;
    size_t slot_cap = 0;
    int chunk_user = 0;
    int pipe_user = 0;
    std::vector<hipStream_t> sweep_streams;
    hipStream_t lu_stream = nullptr;       // K1b of a segment beside K1a of the next one (n > 32)
    std::vector<hipEvent_t> ev_pq;         // K1a of segment i has finished
    std::vector<hipEvent_t> ev_factored, ev_swept, ev_fwd;
    bool unit_ok = false;          // the only cost is one separable final cost (qocx_sweep_common.h)
    // multi-start driver on the device (qocx_opt_*)
    DevBuf<double> opt_m, opt_v, opt_best_controls, opt_max_norms;
    DevBuf<double2> opt_best_final;
    DevBuf<unsigned char> opt_flags;  // [2][B]: improved | update
    int opt_batch = 0;
    DevBuf<double2> lam_scale;     // unit adjoint: [B][S]
    DevBuf<int> offs_x;            // unit adjoint: [chunk][nsteps + 1]
    int keep_step_states = 0;
    bool have_results = false, have_grads = false, have_step_states = false;
    DevBuf<double> controls, cost_out, grads, gstep;
    double* pin_controls = nullptr;  // pinned staging of the controls (qocx_upload_controls)
    size_t pin_controls_cap = 0;
    DevBuf<double2> final_out, step_states;
    DevBuf<double2> q_img, lu_img, dinv, states, xs;
    DevBuf<double2> qt_img;  // one control set (sweep_umode): the transposed propagator images
    DevBuf<int> perm, iperm, s_arr, offs, status;
    DevBuf<int> lu_fallbacks;  // [1] matrices that left the diagonal-pivot MFMA factorisation (qocx_lu_fallbacks)
    DevBuf<int> lu_redo;  // 33 <= n <= 64: matrices the MFMA factorisation hands to the general one (LuArgs::redo)
    // ---- Lindblad problem / evaluation state ----
    // ---- Magnus M4/M6 ----
    int nodes = 1;
    int cu_count = 256;
    int hermitian = 0;  // every h0[t], g[t][k] equals its conjugate transpose bit for bit
    bool general_path = false;  // the evaluation runs on qocx_general.hip (n > 64, or S beyond the sweep's LDS)
    DevBuf<double2> m_rm, mbar_rm, magnus_scratch, lam_buf;
    // M4 with time-independent H0 / G_k as a linear problem in Ke effective controls (M4LinArgs)
    int m4lin_Ke = 0;  // 0: not available for this problem
    DevBuf<double2> ge_cimg, ge_rimg, ge_timg;
    DevBuf<qocx::StepInterp> interp_id;
    DevBuf<double> veff, gnode;
    DevBuf<double> ustep, g_norm_dev;  // step table (launch_step_table): u_k(t_mid) per step; ||G_k||_1
    // explicit-generator mode (qocx_upload_generators): opaque Hamiltonians sampled by the host
    bool explicit_mode = false;
    int explicit_hermitian = 0;
    DevBuf<double2> gen_rm, genbar_rm;  // [B][nsteps] row-major padded generators / cotangents
    // ---- host-supplied state cotangents ----
    int inj_count = 0, inj_batch = 0;
    DevBuf<int> inj_index;
    DevBuf<double2> inj_bars;
    struct Lindblad {
        bool has_problem = false, have_results = false, have_grads = false, have_steps = false;
        int n = 0, S = 0, K = 0, nc = 0, N = 0, nsteps = 0, ces = 1, nops = 0;
        double T = 0, dt = 0, h0_norm = 0, diss_norm = 0, l0_norm = 0;
        std::vector<double> g_norm;
        int has_step_costs = 0, cost_count = 0;
        DevBuf<double2> a0l, a0r, a0ld, a0rd, gp, gpd, gpt, ops, rho0, cost_matrices;
        DevBuf<double> gammas;
        DevBuf<qocx::DevCost> costs;
        DevBuf<int> cost_counts;
        // sub-interval tables, by sub-division count
        struct Grid {
            int nsub = 0;
            DevBuf<qocx::SubStep> substeps;
            DevBuf<int> row_ptr, col;
            DevBuf<double> weight;
        };
        std::map<int, Grid> grids;
        // per evaluation
        int B = 0;
        std::vector<int> order;  // device position -> seed
        // host-supplied density cotangents
        int inj_count = 0, inj_batch = 0;
        std::vector<int> inj_steps;
        std::vector<double> inj_host;  // [B][count][S][n][n] complex
        DevBuf<int> inj_index;
        DevBuf<double2> inj_bars;
        DevBuf<double> gsub, cost_out, grads, controls;
        DevBuf<double2> checkpoints, final_out, step_densities, ystages, scratch;
        DevBuf<double2> kbstages, lam_scale;  // two-sided evaluation (LindbladArgs::phase)
        bool unit_ok = false;                 // one final TargetDensityInfidelity, one density
        bool hermitian = false;               // H0, G_k, sum gamma L^H L, initial densities and cost matrices
                                              // are Hermitian: so is every density and every cotangent
        bool ops_real = false;                // every Lindblad operator has a zero imaginary part
        int global_scratch = 0, multi_wave = 0, cache_gen = 0;
        int pad_op = 0;  // L = 1: a zero second operator behind the real one, for the four-wave launches
        int fixed_ksub = 0;              // > 0: time-dependent Hamiltonian sampled for this grid
        // qocx_debug_lindblad_knobs (tests force the kernel variants large batches / little HBM use)
        int64_t last_subintervals = 0;   // sum over the seeds of the last evaluation
        int64_t dbg_stage_seeds = 0;     // seeds whose stage values may be kept; 0: 45 % of free HBM
        int dbg_min_piece = 256;         // below this many seeds per piece the adjoint recomputes
        int dbg_wave_mode = 0;           // 0 auto, 1 one wave per seed, 2 several whenever built for
        DevBuf<double2> a0_tab, gp_tab, op_tab;
        DevBuf<double> gamma_tab;
    } lb;
    // ---- qocx_debug_set_knob: kernel-variant switches for A/B measurements and tests ----
    std::map<std::string, int64_t> knobs;
    DevBuf<unsigned long long> stamps;  // sweep3 diagnostic build
    int64_t knob(const char* name, int64_t dflt) const {
        auto it = knobs.find(name);
        return it == knobs.end() ? dflt : it->second;
    }
    // ---- timing ----
    int timing = 0;            // 0 off, 1 every launch, 2 + k the launches of kernel k only
    bool time_active = false;  // the launch between the last time_begin / time_end is being timed
    std::vector<TimingRec> pending;
    std::vector<double> timeline;  // (which, start, end) of the last evaluation's launches
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    int64_t t_launch[7] = {0, 0, 0, 0, 0, 0, 0};
    double t_ms[7] = {0, 0, 0, 0, 0, 0, 0};
    // ---- comm ----
    Rccl rccl;
    void* comm = nullptr;
    DevBuf<double> comm_buf;
};

namespace {

void c_image(const double* m, int n, int nb, double2* out) {
    for (int ti = 0; ti < nb; ++ti)
        for (int tj = 0; tj < nb; ++tj)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int q = lane >> 4, c = lane & 15;
                    const int row = 16 * ti + 4 * r + q, col = 16 * tj + c;
                    double2 e = make_double2(0, 0);
                    if (row < n && col < n) {
                        e.x = m[2 * ((size_t)row * n + col)];
                        e.y = m[2 * ((size_t)row * n + col) + 1];
                    }
                    out[((ti * nb + tj) * 4 + r) * 64 + lane] = e;
                }
}

// Column-major NP x NP image of a row-major n x n matrix (or of its transpose), zero padded.
void r_image(const double* m, int n, int np, bool transpose, double2* out) {
    for (int col = 0; col < np; ++col)
        for (int row = 0; row < np; ++row) {
            int r = row, c = col;
            if (transpose) std::swap(r, c);
            double2 e = make_double2(0, 0);
            if (r < n && c < n) {
                e.x = m[2 * ((size_t)r * n + c)];
                e.y = m[2 * ((size_t)r * n + c) + 1];
            }
            out[(size_t)col * np + row] = e;
        }
}

// column-major image -> row-major n x n complex; row_map (optional) gives the image row of each
// output row (the LU factors are stored in original row order: row_map = perm)
void from_image(const double2* img, int n, int np, const int* row_map, double* out) {
    for (int row = 0; row < n; ++row)
        for (int col = 0; col < n; ++col) {
            const int src = row_map ? row_map[row] : row;
            out[2 * ((size_t)row * n + col)] = img[(size_t)col * np + src].x;
            out[2 * ((size_t)row * n + col) + 1] = img[(size_t)col * np + src].y;
        }
}

// timing events come from a grow-only pool: creating and destroying ~100 events per evaluation
// makes the runtime stall for tens of milliseconds every few evaluations
hipEvent_t pooled_event(qocx_ctx* ctx) {
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[ctx->ev_used++];
}

void time_begin(qocx_ctx* ctx, int which, hipStream_t st) {
    // timing 1: every launch; 2 + k: the launches of kernel k only (qocx_set_timing)
    ctx->time_active = ctx->timing == 1 || (ctx->timing >= 2 && which == ctx->timing - 2);
    if (!ctx->time_active) return;
    TimingRec r;
    r.which = which;
    r.a = pooled_event(ctx);
    r.b = pooled_event(ctx);
    (void)hipEventRecord(r.a, st);
    ctx->pending.push_back(r);
}

void time_end(qocx_ctx* ctx, hipStream_t st) {
    if (!ctx->time_active) return;
    (void)hipEventRecord(ctx->pending.back().b, st);
}

void time_collect(qocx_ctx* ctx) {
    ctx->timeline.clear();
    for (auto& r : ctx->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ctx->t_ms[r.which] += ms;
            ctx->t_launch[r.which] += 1;
            float t0 = 0;
            if (hipEventElapsedTime(&t0, ctx->pending.front().a, r.a) == hipSuccess) {
                ctx->timeline.push_back((double)r.which);
                ctx->timeline.push_back((double)t0);
                ctx->timeline.push_back((double)t0 + ms);
            }
        }
    }
    ctx->pending.clear();
    ctx->ev_used = 0;
}

int load_rccl(qocx_ctx* ctx) {
    if (ctx->rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* nm : names) {
        lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return fail(QOCX_ERR_RCCL, std::string("cannot load librccl: ") + dlerror());
    Rccl& r = ctx->rccl;
    r.lib = lib;
    *(void**)(&r.GetUniqueId) = dlsym(lib, "ncclGetUniqueId");
    *(void**)(&r.AllReduce) = dlsym(lib, "ncclAllReduce");
    *(void**)(&r.CommDestroy) = dlsym(lib, "ncclCommDestroy");
    *(void**)(&r.GetErrorString) = dlsym(lib, "ncclGetErrorString");
    if (!r.GetUniqueId || !dlsym(lib, "ncclCommInitRank") || !r.AllReduce || !r.CommDestroy)
        return fail(QOCX_ERR_RCCL, "librccl lacks the expected nccl* symbols");
    return 0;
}

}  // namespace

// ---- host-side helpers of the multi-start GRAPE driver (include/qocx.h) ---------------------------
namespace {
template <class F>
void host_parallel_rows(int64_t count, F f) {
    const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
    const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>((int64_t)hw, count / 8));
    if (nthreads <= 1) {
        f(0, count);
        return;
    }
    std::vector<std::thread> pool;
    const int64_t per = (count + nthreads - 1) / nthreads;
    for (int64_t t = 1; t < nthreads; ++t) {
        const int64_t lo = t * per, hi = std::min(count, lo + per);
        if (lo < hi) pool.emplace_back([=] { f(lo, hi); });
    }
    f(0, std::min(count, per));  // the calling thread takes the first share
    for (auto& th : pool) th.join();
}

// One row of Adam.update. Every product and sum is rounded on its own, as NumPy's array
// operations are (no contraction into fused multiply-adds); division and square root are the
// IEEE ones in scalar and in vector form alike, so the AVX2 clone gives the same bits.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))
#endif
void adam_row(double* __restrict x, const double* __restrict g, double* __restrict m,
              double* __restrict v, int64_t p, double learning_rate, double beta_1, double beta_2,
              double one_m_b1, double one_m_b2, double epsilon, double corr_1, double corr_2,
              int apply_clip, double clip) {
#pragma clang fp contract(off)
    for (int64_t i = 0; i < p; ++i) {
        double gi = g[i];
        if (apply_clip) gi = gi < -clip ? -clip : (gi > clip ? clip : gi);
        const double a = beta_1 * m[i], b = one_m_b1 * gi;
        const double mi = a + b;
        const double sq = gi * gi;
        const double c = beta_2 * v[i], d = one_m_b2 * sq;
        const double vi = c + d;
        m[i] = mi;
        v[i] = vi;
        const double mh = mi / corr_1, vh = vi / corr_2;
        const double den = sqrt(vh) + epsilon;
        const double q = mh / den;
        const double s = learning_rate * q;
        x[i] = x[i] - s;
    }
}
}  // namespace

extern "C" {

const char* qocx_last_error(void) { return g_error.c_str(); }

int qocx_version(void) { return 100; }

int qocx_device_count(int* count) {
    if (!count) return fail(QOCX_ERR_ARG, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return 0;
}

int qocx_create(int device, qocx_ctx** out) {
    if (!out) return fail(QOCX_ERR_ARG, "out is NULL");
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) return fail(QOCX_ERR_HIP, "no HIP device visible");
    if (device < 0) {
        const char* lr = getenv("LOCAL_RANK");
        device = lr ? atoi(lr) : 0;
        device %= count;
    }
    if (device >= count) return fail(QOCX_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    qocx_ctx* ctx = new qocx_ctx();
    ctx->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
            cus > 0)
            ctx->cu_count = cus;
    }
    hipError_t e = hipStreamCreate(&ctx->stream);
    if (e != hipSuccess) {
        delete ctx;
        return fail(QOCX_ERR_HIP, hipGetErrorString(e));
    }
    if (ctx->status.ensure(1)) {
        qocx_destroy(ctx);
        return QOCX_ERR_HIP;
    }
    // side streams of the latency-bound sweeps get the highest priority, so that their few
    // waves are placed as soon as a SIMD frees up under the compute stream's big grids
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (hipStreamCreateWithFlags(&ctx->lu_stream, hipStreamNonBlocking) != hipSuccess) {
        ctx->lu_stream = nullptr;
        qocx_destroy(ctx);
        return fail(QOCX_ERR_HIP, "cannot create the pipeline streams");
    }
    for (int i = 0; i < 32; ++i) {
        hipEvent_t e0;
        if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess) {
            qocx_destroy(ctx);
            return fail(QOCX_ERR_HIP, "cannot create the pipeline streams");
        }
        ctx->ev_pq.push_back(e0);
    }
    for (int i = 0; i < 32; ++i) {
        hipStream_t st;
        hipEvent_t e1, e2, e3;
        if ((i < 2 &&
             hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_greatest) != hipSuccess) ||
            hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e3, hipEventDisableTiming) != hipSuccess) {
            qocx_destroy(ctx);  // releases what has been created so far
            return fail(QOCX_ERR_HIP, "cannot create the pipeline streams");
        }
        if (i < 2) ctx->sweep_streams.push_back(st);  // forward | adjoint (two-sided pipeline)
        ctx->ev_factored.push_back(e1);
        ctx->ev_swept.push_back(e2);
        ctx->ev_fwd.push_back(e3);
    }
    // (diagnostic build only, qocx_diag.h: fuzz runs of the whole suite on another sweep)
    if (const char* env = qocx::diag_getenv("QOCX_SWEEP_IMPL")) ctx->knobs["sweep_impl"] = atoi(env);
    if (const char* env = qocx::diag_getenv("QOCX_SWEEP_LOADER")) ctx->knobs["sweep_loader"] = atoi(env);
    *out = ctx;
    return 0;
}

int qocx_destroy(qocx_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    time_collect(ctx);
    if (ctx->comm && ctx->rccl.CommDestroy) ctx->rccl.CommDestroy(ctx->comm);
    if (ctx->pin_controls) (void)hipHostFree(ctx->pin_controls);
    ctx->pin_controls = nullptr;
    ctx->m_rm.release();
    ctx->lam_buf.release();
    ctx->inj_index.release();
    ctx->inj_bars.release();
    ctx->mbar_rm.release();
    ctx->magnus_scratch.release();
    ctx->lam_scale.release();
    ctx->offs_x.release();
    ctx->ge_cimg.release(); ctx->ge_rimg.release(); ctx->ge_timg.release();
    ctx->interp_id.release(); ctx->veff.release(); ctx->gnode.release();
    ctx->ustep.release(); ctx->g_norm_dev.release(); ctx->lu_redo.release(); ctx->lu_fallbacks.release();
    ctx->opt_m.release(); ctx->opt_v.release(); ctx->opt_best_controls.release();
    ctx->opt_max_norms.release(); ctx->opt_best_final.release(); ctx->opt_flags.release();
    ctx->gen_rm.release(); ctx->genbar_rm.release(); ctx->stamps.release();
    DevBuf<double2>* b2[] = {&ctx->h0_cimg, &ctx->g_cimg, &ctx->h0_rimg, &ctx->g_rimg, &ctx->h0_timg,
                             &ctx->g_timg, &ctx->psi0, &ctx->cost_vectors, &ctx->final_out,
                             &ctx->step_states, &ctx->q_img, &ctx->qt_img, &ctx->lu_img, &ctx->dinv,
                             &ctx->states, &ctx->xs};
    for (auto* b : b2) b->release();
    DevBuf<double>* b1[] = {&ctx->weight, &ctx->controls, &ctx->cost_out, &ctx->grads, &ctx->gstep,
                            &ctx->comm_buf};
    for (auto* b : b1) b->release();
    DevBuf<int>* bi[] = {&ctx->cost_counts, &ctx->row_ptr, &ctx->col_step, &ctx->perm, &ctx->iperm, &ctx->s_arr,
                         &ctx->offs, &ctx->status};
    for (auto* b : bi) b->release();
    ctx->interp.release();
    ctx->costs.release();
    {
        auto& lb = ctx->lb;
        DevBuf<double2>* l2[] = {&lb.a0l, &lb.a0r, &lb.a0ld, &lb.a0rd, &lb.gp, &lb.gpd, &lb.gpt,
                                 &lb.ops, &lb.rho0, &lb.cost_matrices, &lb.checkpoints,
                                 &lb.final_out, &lb.step_densities};
        for (auto* b : l2) b->release();
        DevBuf<double>* l1[] = {&lb.gammas, &lb.gsub, &lb.cost_out, &lb.grads, &lb.controls};
        for (auto* b : l1) b->release();
        lb.costs.release();
        lb.cost_counts.release();
        for (auto& kv : lb.grids) {
            kv.second.substeps.release();
            kv.second.row_ptr.release();
            kv.second.col.release();
            kv.second.weight.release();
        }
        lb.inj_index.release();
        lb.inj_bars.release();
        lb.ystages.release();
        lb.kbstages.release();
        lb.lam_scale.release();
        lb.scratch.release();
        lb.a0_tab.release();
        lb.gp_tab.release();
    }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto st : ctx->sweep_streams) (void)hipStreamDestroy(st);
    if (ctx->lu_stream) (void)hipStreamDestroy(ctx->lu_stream);
    for (auto e : ctx->ev_pq) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_factored) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_swept) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_fwd) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int qocx_synchronize(qocx_ctx* ctx) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int qocx_set_schroedinger_problem(qocx_ctx* ctx, const qocx_schroedinger_problem* p) {
    if (!ctx || !p) return fail(QOCX_ERR_ARG, "NULL argument");
    if (p->struct_size != (int32_t)sizeof(qocx_schroedinger_problem))
        return fail(QOCX_ERR_ARG, "qocx_schroedinger_problem.struct_size does not match this "
                                  "library's header (stale binding?)");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = p->hilbert_size, S = p->state_count, K = p->control_count;
    const int N = p->system_eval_count, nc = p->control_eval_count;
    if (n < 1 || n > 1024)
        return fail(QOCX_ERR_ARG, "hilbert_size must be in 1..1024 (1..64: the wavefront kernels; 65..1024: the general "
                                  "path of qocx_general.hip)");

    // (a full propagator has n states: up to 256 of them on the general path)
    if (S < 1 || S > (n > 64 ? 1024 : 64)) return fail(QOCX_ERR_ARG, "state_count must be in 1..64 (1..1024 above hilbert_size 64)");
    if (K < 0 || K > 64) return fail(QOCX_ERR_ARG, "control_count must be in 0..64");
    if (N < 2) return fail(QOCX_ERR_ARG, "system_eval_count must be >= 2");
    if (K > 0 && nc < 2) return fail(QOCX_ERR_ARG, "control_eval_count must be >= 2");
    if (p->cost_eval_step < 1) return fail(QOCX_ERR_ARG, "cost_eval_step must be >= 1");
    if (p->magnus_policy != QOCX_MAGNUS_M2 && p->magnus_policy != QOCX_MAGNUS_M4 &&
        p->magnus_policy != QOCX_MAGNUS_M6)
        return fail(QOCX_ERR_ARG, "unknown magnus_policy");
    const int nsteps = N - 1;
    const int nodes = p->magnus_policy / 2;  // quadrature nodes per step: 1, 2, 3
    if (p->nt != 1 && p->nt != nsteps * nodes)
        return fail(QOCX_ERR_ARG, "nt must be 1 or (N-1) * (quadrature nodes of the policy)");
    if (!p->h0 || !p->initial_states || (K > 0 && !p->g))
        return fail(QOCX_ERR_ARG, "h0 / g / initial_states missing");
    if (p->cost_count < 0 || (p->cost_count > 0 && !p->costs))
        return fail(QOCX_ERR_ARG, "costs missing");

    // matrices are padded to 16, 32 or 64: one, four or sixteen MFMA tiles (33 <= n <= 64 runs on
    // the four-wave K1a of qocx_pade4.hip and the NB = 4 forms of K1b / K2 / K3)
    // (n > 64: nb = ceil(n / 16) > 4 selects the general path; it reads the row-major padded matrices that
    // h0_timg / g_timg hold - the column-major images of the transposes)
    const int nb = (n <= 16) ? 1 : (n <= 32 ? 2 : (n <= 64 ? 4 : (n + 15) / 16)), np = 16 * nb, mat = np * np, nt = p->nt;
    ctx->has_problem = false;
    ctx->n = n; ctx->nb = nb; ctx->np = np; ctx->S = S; ctx->K = K; ctx->nc = nc; ctx->N = N;
    ctx->nsteps = nsteps; ctx->ces = p->cost_eval_step; ctx->nt = nt; ctx->nodes = nodes;
    ctx->T = p->evolution_time;
    ctx->dt = p->evolution_time / (N - 1);  // programstate.py:44

    {
        auto is_hermitian = [n](const double* m) {
            for (int r = 0; r < n; ++r)
                for (int c = r; c < n; ++c)
                    if (m[2 * ((size_t)r * n + c)] != m[2 * ((size_t)c * n + r)] ||
                        m[2 * ((size_t)r * n + c) + 1] != -m[2 * ((size_t)c * n + r) + 1])
                        return false;
            return true;
        };
        bool herm = true;
        for (int t = 0; t < p->nt && herm; ++t) {
            herm = is_hermitian(p->h0 + (size_t)t * n * n * 2);
            for (int k = 0; k < K && herm; ++k)
                herm = is_hermitian(p->g + ((size_t)t * K + k) * n * n * 2);
        }
        ctx->hermitian = herm ? 1 : 0;
    }
    // Hamiltonian images + norms for the squaring bound
    std::vector<double2> img((size_t)nt * mat);
    ctx->h0_norm_max = 0;
    for (int t = 0; t < nt; ++t) {
        const double* m = p->h0 + (size_t)t * n * n * 2;
        ctx->h0_norm_max = std::max(ctx->h0_norm_max, one_norm(m, n));
        c_image(m, n, nb, img.data() + (size_t)t * mat);
    }
    if (ctx->h0_cimg.upload(img, ctx->stream)) return QOCX_ERR_HIP;
    for (int t = 0; t < nt; ++t) r_image(p->h0 + (size_t)t * n * n * 2, n, np, false, img.data() + (size_t)t * mat);
    if (ctx->h0_rimg.upload(img, ctx->stream)) return QOCX_ERR_HIP;
    for (int t = 0; t < nt; ++t) r_image(p->h0 + (size_t)t * n * n * 2, n, np, true, img.data() + (size_t)t * mat);
    if (ctx->h0_timg.upload(img, ctx->stream)) return QOCX_ERR_HIP;
    ctx->g_norm_max.assign(K, 0.0);
    std::vector<double2> gimg((size_t)nt * K * mat);
    for (int pass = 0; pass < 3; ++pass) {
        for (int t = 0; t < nt; ++t)
            for (int k = 0; k < K; ++k) {
                const double* m = p->g + ((size_t)t * K + k) * n * n * 2;
                double2* dst = gimg.data() + ((size_t)t * K + k) * mat;
                if (pass == 0) {
                    ctx->g_norm_max[k] = std::max(ctx->g_norm_max[k], one_norm(m, n));
                    c_image(m, n, nb, dst);
                } else {
                    r_image(m, n, np, pass == 2, dst);
                }
            }
        DevBuf<double2>& dst = pass == 0 ? ctx->g_cimg : (pass == 1 ? ctx->g_rimg : ctx->g_timg);
        if (dst.upload(gimg, ctx->stream)) return QOCX_ERR_HIP;
    }
    if (K > 0 && ctx->g_norm_dev.upload(ctx->g_norm_max, ctx->stream)) return QOCX_ERR_HIP;  // step table

    // M4, time-independent H0 / G_k: the commutators leave the time loop (M4LinArgs). Constant
    // matrices G_k, A_k = -i [G_k, H0], B_kl = -i [G_k, G_l] (k < l); Hermitian when H0 and the
    // G_k are (made so bit for bit, the Hermitian kernel forms rely on it).
    ctx->m4lin_Ke = 0;
    if (nodes == 2 && nt == 1 && K >= 1 && K <= QOCX_M4LIN_MAX_K) {
        const int Ke = 2 * K + K * (K - 1) / 2;
        const size_t nn = (size_t)n * n;
        std::vector<double> ge((size_t)Ke * nn * 2);
        std::copy(p->g, p->g + (size_t)K * nn * 2, ge.begin());
        auto neg_i_commutator = [&](const double* x, const double* y, double* out) {
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c) {
                    double re = 0, im = 0;  // (x y - y x)[r][c]
                    for (int q = 0; q < n; ++q) {
                        const double* xa = x + 2 * ((size_t)r * n + q);
                        const double* yb = y + 2 * ((size_t)q * n + c);
                        const double* ya = y + 2 * ((size_t)r * n + q);
                        const double* xb = x + 2 * ((size_t)q * n + c);
                        re += xa[0] * yb[0] - xa[1] * yb[1] - (ya[0] * xb[0] - ya[1] * xb[1]);
                        im += xa[0] * yb[1] + xa[1] * yb[0] - (ya[0] * xb[1] + ya[1] * xb[0]);
                    }
                    out[2 * ((size_t)r * n + c)] = im;       // -i (re + i im) = im - i re
                    out[2 * ((size_t)r * n + c) + 1] = -re;
                }
            if (ctx->hermitian)
                for (int r = 0; r < n; ++r)
                    for (int c = r; c < n; ++c) {
                        double* a = out + 2 * ((size_t)r * n + c);
                        double* b = out + 2 * ((size_t)c * n + r);
                        const double re = 0.5 * (a[0] + b[0]), im = (r == c) ? 0.0 : 0.5 * (a[1] - b[1]);
                        a[0] = re; a[1] = im;
                        b[0] = re; b[1] = -im;
                    }
        };
        for (int k = 0; k < K; ++k)
            neg_i_commutator(p->g + (size_t)k * nn * 2, p->h0, ge.data() + (size_t)(K + k) * nn * 2);
        int e = 2 * K;
        for (int k = 0; k < K; ++k)
            for (int l = k + 1; l < K; ++l, ++e)
                neg_i_commutator(p->g + (size_t)k * nn * 2, p->g + (size_t)l * nn * 2,
                                 ge.data() + (size_t)e * nn * 2);
        std::vector<double2> eimg((size_t)Ke * mat);
        for (int pass = 0; pass < 3; ++pass) {
            for (int k = 0; k < Ke; ++k) {
                const double* m = ge.data() + (size_t)k * nn * 2;
                if (pass == 0) c_image(m, n, nb, eimg.data() + (size_t)k * mat);
                else r_image(m, n, np, pass == 2, eimg.data() + (size_t)k * mat);
            }
            DevBuf<double2>& dst = pass == 0 ? ctx->ge_cimg : (pass == 1 ? ctx->ge_rimg : ctx->ge_timg);
            if (dst.upload(eimg, ctx->stream)) return QOCX_ERR_HIP;
        }
        std::vector<qocx::StepInterp> ident(nsteps);
        for (int j = 0; j < nsteps; ++j) ident[j] = qocx::StepInterp{j, j, 1.0, 0.0};
        if (ctx->interp_id.upload(ident, ctx->stream)) return QOCX_ERR_HIP;
        ctx->m4lin_Ke = Ke;
    }

    // initial states, padded
    std::vector<double2> psi((size_t)S * np, make_double2(0, 0));
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < n; ++i)
            psi[(size_t)s * np + i] = make_double2(p->initial_states[2 * ((size_t)s * n + i)],
                                                   p->initial_states[2 * ((size_t)s * n + i) + 1]);
    if (ctx->psi0.upload(psi, ctx->stream)) return QOCX_ERR_HIP;

    // interpolation table at the quadrature times t_j + c_q dt of the Magnus policy
    // (mathmethods.py:54-65; nodes :72, :96-97, :125-127)
    static const double node_c[3][3] = {
        {0.5, 0, 0},
        {0.5 - std::sqrt(3.0) / 6, 0.5 + std::sqrt(3.0) / 6, 0},
        {0.5 - std::sqrt(15.0) / 10, 0.5, 0.5 + std::sqrt(15.0) / 10}};
    std::vector<qocx::StepInterp> interp((size_t)nsteps * nodes);
    std::vector<std::vector<std::pair<int, double>>> rows(K > 0 ? nc : 0);
    if (K > 0) {
        std::vector<double> xs(nc);
        const double stepx = p->evolution_time / (nc - 1);  // numpy.linspace
        for (int i = 0; i < nc; ++i) xs[i] = i * stepx;
        xs[nc - 1] = p->evolution_time;
        for (int j = 0; j < nsteps; ++j)
            for (int q = 0; q < nodes; ++q) {
                const double time = j * ctx->dt;
                const double x = time + ctx->dt * node_c[nodes - 1][q];
                int i1, i2;
                if (x <= xs[0]) {
                    i1 = 0; i2 = 1;
                } else if (x >= xs[nc - 1]) {
                    i1 = nc - 2; i2 = nc - 1;
                } else {
                    int idx = 0;
                    while (!(x <= xs[idx])) ++idx;
                    i1 = idx - 1; i2 = idx;
                }
                qocx::StepInterp& e = interp[(size_t)j * nodes + q];
                e.i1 = i1; e.i2 = i2;
                e.dx = xs[i2] - xs[i1];
                e.off = x - xs[i1];
                const double theta = e.off / e.dx;
                rows[i1].push_back(std::make_pair(j * nodes + q, 1.0 - theta));
                rows[i2].push_back(std::make_pair(j * nodes + q, theta));
            }
    } else {
        for (auto& e : interp) e = qocx::StepInterp{0, 0, 1.0, 0.0};
    }
    if (ctx->interp.upload(interp, ctx->stream)) return QOCX_ERR_HIP;
    std::vector<int> row_ptr(1, 0), col_step;
    std::vector<double> weight;
    for (auto& r : rows) {
        for (auto& e : r) {
            col_step.push_back(e.first);
            weight.push_back(e.second);
        }
        row_ptr.push_back((int)col_step.size());
    }
    if (ctx->row_ptr.upload(row_ptr, ctx->stream)) return QOCX_ERR_HIP;
    if (ctx->col_step.upload(col_step, ctx->stream)) return QOCX_ERR_HIP;
    if (ctx->weight.upload(weight, ctx->stream)) return QOCX_ERR_HIP;

    // costs
    std::vector<qocx::DevCost> dcosts;
    std::vector<double2> pool;
    std::vector<int> counts;
    ctx->has_step_costs = 0;
    for (int ci = 0; ci < p->cost_count; ++ci) {
        const qocx_cost_desc& c = p->costs[ci];
        qocx::DevCost d;
        d.kind = c.kind;
        d.step_cost = c.step_cost ? 1 : 0;
        d.scale = c.scale;
        d.vec_offset = (int)(pool.size() / np);
        d.cnt_offset = (int)counts.size();
        if (!c.vectors) return fail(QOCX_ERR_ARG, "cost vectors missing");
        int nvec = S;
        if (c.kind == QOCX_COST_FORBID) {
            if (!c.counts) return fail(QOCX_ERR_ARG, "forbid counts missing");
            nvec = 0;
            for (int s = 0; s < S; ++s) {
                if (c.counts[s] < 1) return fail(QOCX_ERR_ARG, "forbid count < 1");
                counts.push_back(c.counts[s]);
                nvec += c.counts[s];
            }
        } else if (c.kind != QOCX_COST_TARGET_COHERENT && c.kind != QOCX_COST_TARGET_INCOHERENT) {
            return fail(QOCX_ERR_ARG, "unknown cost kind");
        }
        for (int v = 0; v < nvec; ++v)
            for (int i = 0; i < np; ++i) {
                double2 e = make_double2(0, 0);
                if (i < n) {
                    e.x = c.vectors[2 * ((size_t)v * n + i)];
                    e.y = c.vectors[2 * ((size_t)v * n + i) + 1];
                }
                pool.push_back(e);
            }
        if (d.step_cost) ctx->has_step_costs = 1;
        dcosts.push_back(d);
    }
    ctx->cost_count = (int)dcosts.size();
    // one final-state target cost whose cotangent is ONE scalar times the targets: coherent (any
    // number of states) or a single state
    ctx->unit_ok = dcosts.size() == 1 && !dcosts[0].step_cost &&
                   (dcosts[0].kind == QOCX_DEV_COST_COHERENT ||
                    (dcosts[0].kind == QOCX_DEV_COST_INCOHERENT && S == 1));
    if (ctx->costs.upload(dcosts, ctx->stream)) return QOCX_ERR_HIP;
    if (ctx->cost_vectors.upload(pool, ctx->stream)) return QOCX_ERR_HIP;
    if (ctx->cost_counts.upload(counts, ctx->stream)) return QOCX_ERR_HIP;
    // More states than the wavefront sweep's LDS holds (33 <= n <= 64: more than 13 - a full propagator there
    // has n): the general path, whose sweep keeps its vectors in HBM, takes the problem where it can
    ctx->general_path = nb > 4;
    if (nb <= 4 && qocx::sweep_lds_bytes(nb, S) > 160 * 1024) ctx->general_path = true;
    ctx->has_problem = true;
    ctx->have_results = false;
    ctx->B = 0;
    ctx->inj_count = 0;
    return 0;
}

}  // extern "C"

// 1-norm bound of the step generator from the bound b >= ||dt a(t)|| of its node generators
// (mathmethods.py:96-164: m2 = b; m4 = dt/2 (a1 + a2) + sqrt(3)/12 dt^2 [a2, a1]; m6)
// max over the Pade orders of eps_m(theta) = sum_{j>=1} (b_j / b_0) theta^j (qocx_lu5.h)
static double pade_eps_max(double theta) {
    const double b3[] = {120.0, 60.0, 12.0, 1.0};
    const double b5[] = {30240.0, 15120.0, 3360.0, 420.0, 30.0, 1.0};
    const double b7[] = {17297280.0, 8648640.0, 1995840.0, 277200.0, 25200.0, 1512.0, 56.0, 1.0};
    const double b9[] = {17643225600.0, 8821612800.0, 2075673600.0, 302702400.0, 30270240.0, 2162160.0,
                         110880.0, 3960.0, 90.0, 1.0};
    const double b13[] = {64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
                          129060195264000.0, 10559470521600.0, 670442572800.0, 33522128640.0, 1323241920.0,
                          40840800.0, 960960.0, 16380.0, 182.0, 1.0};
    const double* tabs[] = {b3, b5, b7, b9, b13};
    const int orders[] = {3, 5, 7, 9, 13};
    if (!(theta >= 0.0) || !(theta < 1e300)) return 1e300;
    double worst = 0.0;
    for (int t = 0; t < 5; ++t) {
        double eps = 0.0, tp = 1.0;
        for (int j = 1; j <= orders[t]; ++j) {
            tp *= theta;
            eps += tabs[t][j] / tabs[t][0] * tp;
        }
        worst = std::max(worst, eps);
    }
    return worst;
}

static double magnus_norm_bound(int nodes, double bound) {
    if (nodes == 2) return bound + (std::sqrt(3.0) / 12) * 2 * bound * bound;
    if (nodes == 3) {
        const double b1 = bound, b2 = (std::sqrt(15.0) / 3) * 2 * bound, b3 = (10.0 / 3) * 4 * bound;
        const double c12 = 2 * b1 * b2, x = 20 * b1 + b3 + c12, w = 2 * b3 + c12;
        const double y = b2 + (1.0 / 60) * 2 * b1 * w;
        return b1 + 0.5 * b3 + (1.0 / 240) * 2 * x * y;
    }
    return bound;
}

// Evaluation for Hilbert sizes above 64 (qocx_general.hip): classic order, one stream - factor every step,
// forward sweep, adjoint sweep, K3, scatter - per memory chunk of seeds.
namespace qocx {
size_t general_krylov_scratch(int np, int S);
void launch_general_magnus(const MagnusArgs& a, bool vjp, int blocks, hipStream_t st);
}
static int eval_general(qocx_ctx* ctx, int want_grad) {
    const int B = ctx->B, np = ctx->np, S = ctx->S, K = ctx->K, nsteps = ctx->nsteps;
    const size_t mat = (size_t)np * np;
    const bool explicit_gen = ctx->explicit_mode;
    // M4 with a time-independent system: linear in Ke effective controls with constant matrices (M4LinArgs)
    const bool m4lin = ctx->m4lin_Ke > 0 && ctx->nodes == 2 && !explicit_gen;
    // M6, and M4 on a time-dependent system: generators and reverse rules by qocx_general.hip's magnus_kernel
    const bool magnus = ctx->nodes > 1 && !m4lin && !explicit_gen;
    const int nodes = magnus ? ctx->nodes : 1;
    const int Kk = m4lin ? ctx->m4lin_Ke : K;
    const size_t per_seed = (size_t)nsteps * (mat * 32 + 4) + ctx->slot_cap * S * np * 32 +
                            (size_t)(nsteps + 1) * 4 + (size_t)nsteps * std::max(Kk, 1) * 40;
    // (persistent workgroups with 7 scratch matrices each: as many as 16 GB hold, two per CU at most)
    const int max_blocks = (int)std::max<size_t>(1, std::min<size_t>((size_t)2 * ctx->cu_count, ((size_t)16 << 30) / (7 * mat * 16)));
    int chunk = ctx->chunk_user;
    if (chunk <= 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t have = ctx->q_img.count * 16 + ctx->lu_img.count * 16 + ctx->states.count * 16 +
                            ctx->xs.count * 16 + ctx->magnus_scratch.count * 16;
        const size_t fixed = (size_t)max_blocks * 7 * mat * 16;
        const size_t budget = (size_t)((double)(free_b + have) * 0.6);
        chunk = (int)std::min<size_t>((size_t)B, std::max<size_t>(1, (budget > fixed ? budget - fixed : 0) / per_seed));
    }
    chunk = std::min(chunk, B);
    const size_t cm = (size_t)chunk * nsteps;
    const int blocks = (int)std::min<size_t>(cm, (size_t)max_blocks);
    if (ctx->q_img.ensure(cm * mat) || ctx->lu_img.ensure(cm * mat) || ctx->s_arr.ensure(cm) ||
        ctx->states.ensure((size_t)chunk * ctx->slot_cap * S * np) ||
        ctx->xs.ensure(want_grad ? (size_t)chunk * ctx->slot_cap * S * np : 1) ||
        ctx->offs.ensure((size_t)chunk * (nsteps + 1)) || ctx->gstep.ensure(cm * std::max(Kk, 1)) ||
        (m4lin && (ctx->veff.ensure(cm * Kk) || ctx->gnode.ensure(want_grad ? cm * 2 * K : 1))) ||
        ctx->cost_out.ensure(B) || ctx->grads.ensure((size_t)B * ctx->nc * std::max(K, 1)) ||
        ctx->final_out.ensure((size_t)B * S * np) || ctx->lam_buf.ensure((size_t)chunk * S * np) ||
        ctx->magnus_scratch.ensure((size_t)blocks * 7 * mat))
        return QOCX_ERR_HIP;
    // K3 of many states keeps the chains of every state in scratch: as many workgroups as 8 GB hold
    const size_t k3_elems = qocx::general_krylov_scratch(np, S);
    const int k3_blocks = (int)std::max<size_t>(1, std::min<size_t>((size_t)blocks, ((size_t)8 << 30) / (k3_elems * 16)));
    if (want_grad && ctx->magnus_scratch.ensure((size_t)k3_blocks * k3_elems)) return QOCX_ERR_HIP;
    const int mg_blocks = (int)std::max<size_t>(1, std::min<size_t>((size_t)blocks, ((size_t)8 << 30) / (24 * mat * 16)));
    if (magnus && (ctx->magnus_scratch.ensure((size_t)mg_blocks * 24 * mat) || ctx->m_rm.ensure(cm * mat) ||
                   ctx->mbar_rm.ensure(want_grad ? cm * mat : 1) || ctx->gstep.ensure(cm * nodes * std::max(K, 1))))
        return QOCX_ERR_HIP;
    if (ctx->keep_step_states)
        if (ctx->step_states.ensure((size_t)B * (nsteps + 1) * S * np)) return QOCX_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->status.p, 0, sizeof(int), ctx->stream));
    hipStream_t cs = ctx->stream;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int bc = std::min(chunk, B - b0);
        ctx->last_chunk = bc;
        qocx::GeneralArgs fa;
        fa.np = np; fa.K = K; fa.nc = ctx->nc; fa.nsteps = nsteps; fa.nt = ctx->nt; fa.dt = ctx->dt;
        fa.controls = ctx->controls.p ? ctx->controls.p + (size_t)b0 * ctx->nc * K : nullptr;
        fa.interp = ctx->interp.p;
        fa.h0_rm = ctx->h0_timg.p; fa.g_rm = ctx->g_timg.p;
        fa.gen_rm = explicit_gen ? ctx->gen_rm.p + (size_t)b0 * nsteps * mat : nullptr;
        fa.pade_policy = (int)ctx->knob("pade_order", 0);
        fa.sq_max = std::min(30, ctx->sbound);
        fa.q_img = ctx->q_img.p; fa.pinv_img = ctx->lu_img.p; fa.s_arr = ctx->s_arr.p; fa.status = ctx->status.p;
        fa.scratch = ctx->magnus_scratch.p;
        fa.total = (size_t)bc * nsteps;
        qocx::M4LinArgs m4;
        if (m4lin) {
            m4.controls = fa.controls; m4.interp = ctx->interp.p;
            m4.K = K; m4.Ke = Kk; m4.nc = ctx->nc; m4.nsteps = nsteps; m4.S = S;
            m4.f0dt = (std::sqrt(3.0) / 12) * ctx->dt;
            m4.veff = ctx->veff.p; m4.gstep = ctx->gstep.p; m4.gnode = ctx->gnode.p;
            m4.lam_scale = nullptr;
            m4.total = fa.total;
            qocx::launch_m4lin_controls(m4, cs);
            fa.controls = ctx->veff.p; fa.interp = ctx->interp_id.p; fa.g_rm = ctx->ge_timg.p;
            fa.K = Kk; fa.nc = nsteps;
        }
        const int fblocks = (int)std::min<size_t>(fa.total, (size_t)blocks);
        qocx::MagnusArgs ma;
        if (magnus) {
            ma.controls = fa.controls; ma.interp = ctx->interp.p;
            ma.h0_cimg = ctx->h0_timg.p; ma.g_cimg = ctx->g_timg.p;  // (row-major padded matrices here)
            ma.K = K; ma.nc = ctx->nc; ma.nsteps = nsteps; ma.nt = ctx->nt; ma.nodes = nodes;
            ma.step0 = 0; ma.seg_len = nsteps; ma.skew = 0; ma.dt = ctx->dt;
            ma.m_rm = ctx->m_rm.p; ma.mbar_rm = nullptr; ma.gstep = nullptr;
            ma.scratch = ctx->magnus_scratch.p; ma.total = fa.total; ma.n = np;
            time_begin(ctx, 0, cs);
            qocx::launch_general_magnus(ma, false, std::min(fblocks, mg_blocks), cs);
            time_end(ctx, cs);
            fa.gen_rm = ctx->m_rm.p;  // the factor kernel and K3 take the step generators as they are
        }
        time_begin(ctx, 0, cs);
        if (qocx::launch_general_factor(fa, fblocks, cs)) return fail(QOCX_ERR_HIP, "K1a (general): LDS size refused");
        time_end(ctx, cs);

        qocx::GeneralSweepArgs sa;
        sa.np = np; sa.S = S; sa.nsteps = nsteps; sa.cost_eval_step = ctx->ces;
        sa.has_step_costs = ctx->has_step_costs; sa.phase = want_grad ? 3 : 1;
        sa.q_img = fa.q_img; sa.pinv_img = fa.pinv_img; sa.s_arr = fa.s_arr; sa.psi0 = ctx->psi0.p;
        sa.slot_cap = ctx->slot_cap; sa.states = ctx->states.p; sa.xs = ctx->xs.p; sa.offs = ctx->offs.p;
        sa.lam_buf = ctx->lam_buf.p;
        sa.cost_count = ctx->cost_count; sa.costs = ctx->costs.p; sa.cost_vectors = ctx->cost_vectors.p;
        sa.cost_counts = ctx->cost_counts.p;
        sa.inj_count = ctx->inj_count;
        sa.inj_index = ctx->inj_count > 0 ? ctx->inj_index.p : nullptr;
        sa.inj_bars = ctx->inj_count > 0 ? ctx->inj_bars.p + (size_t)b0 * ctx->inj_count * S * np : nullptr;
        sa.cost_out = ctx->cost_out.p + b0;
        sa.final_out = ctx->final_out.p + (size_t)b0 * S * np;
        sa.step_states = ctx->keep_step_states ? ctx->step_states.p + (size_t)b0 * (nsteps + 1) * S * np : nullptr;
        sa.status = ctx->status.p;
        // Many states, final costs only, few seeds (a full propagator of one control set): the states of a seed
        // in groups of rows on several workgroups (qocx_general.hip, split mode) - knob "general_split" 0: off
        int groups = 1;
        if (S >= 16 && !ctx->has_step_costs && ctx->inj_count == 0 && !ctx->keep_step_states &&
            ctx->knob("general_split", 1) != 0)
            groups = std::min((S + 7) / 8, std::max(1, 2 * ctx->cu_count / bc));
        time_begin(ctx, 1, cs);
        if (groups >= 2) {
            sa.phase = 1 | 8 | (groups << 8);
            qocx::launch_general_sweep(sa, bc, cs);
            sa.phase = 4 | 8 | (want_grad ? 16 : 0);
            qocx::launch_general_sweep(sa, bc, cs);
            if (want_grad) {
                sa.phase = 2 | 8 | (groups << 8);
                qocx::launch_general_sweep(sa, bc, cs);
            }
        } else {
            qocx::launch_general_sweep(sa, bc, cs);
        }
        time_end(ctx, cs);

        if (want_grad) {
            qocx::GeneralKrylovArgs ka;
            ka.np = np; ka.S = S; ka.K = fa.K; ka.nc = fa.nc; ka.nsteps = nsteps; ka.nt = ctx->nt; ka.dt = ctx->dt;
            ka.controls = fa.controls; ka.interp = fa.interp; ka.h0_rm = fa.h0_rm; ka.g_rm = fa.g_rm;
            ka.gen_rm = fa.gen_rm;
            ka.mbar_rm = explicit_gen ? ctx->genbar_rm.p + (size_t)b0 * nsteps * mat : (magnus ? ctx->mbar_rm.p : nullptr);
            ka.s_arr = fa.s_arr; ka.offs = ctx->offs.p; ka.states = ctx->states.p; ka.xs = ctx->xs.p;
            ka.slot_cap = ctx->slot_cap; ka.gstep = ctx->gstep.p; ka.scratch = ctx->magnus_scratch.p;
            ka.total = fa.total;
            // (Magnus generators are skew only to rounding: the general chains there)
            ka.skew = (magnus ? 0 : (explicit_gen ? ctx->explicit_hermitian : ctx->hermitian)) &&
                      ctx->knob("general_skew", 1) != 0;
            time_begin(ctx, 2, cs);
            if (qocx::launch_general_krylov(ka, std::min(fblocks, k3_blocks), cs))
                return fail(QOCX_ERR_HIP, "K3 (general): LDS size refused");
            time_end(ctx, cs);
            if (magnus) {
                ma.m_rm = nullptr; ma.mbar_rm = ctx->mbar_rm.p; ma.gstep = ctx->gstep.p;
                time_begin(ctx, 2, cs);
                qocx::launch_general_magnus(ma, true, std::min(fblocks, mg_blocks), cs);
                time_end(ctx, cs);
            }
            if (!explicit_gen) {
                qocx::ScatterArgs sc;
                sc.gstep = ka.gstep; sc.row_ptr = ctx->row_ptr.p; sc.col_step = ctx->col_step.p;
                sc.weight = ctx->weight.p;
                sc.grads = ctx->grads.p + (size_t)b0 * ctx->nc * K;
                sc.B = bc; sc.nc = ctx->nc; sc.K = K; sc.nsteps = nsteps * ctx->nodes;
                sc.lam_scale = nullptr; sc.S = S;
                time_begin(ctx, 3, cs);
                if (m4lin) {  // effective-control cotangents -> node cotangents
                    qocx::launch_m4lin_chain(m4, cs);
                    sc.gstep = ctx->gnode.p;
                }
                qocx::launch_scatter(sc, cs);
                time_end(ctx, cs);
            }
        }
    }
    HIP_TRY(hipGetLastError());
    int status = 0;
    HIP_TRY(hipMemcpyAsync(&status, ctx->status.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    time_collect(ctx);
    if (status & 2) return fail(QOCX_ERR_ARG, "non-finite generator norm");
    if (status & 1) return fail(QOCX_ERR_SINGULAR, "Singular matrix");
    if (status & 4) return fail(QOCX_ERR_CAPACITY, "squaring sub-step capacity exceeded");
    ctx->have_results = true;
    ctx->have_grads = want_grad != 0;
    ctx->have_step_states = ctx->keep_step_states != 0;
    return 0;
}

extern "C" {

int qocx_upload_controls(qocx_ctx* ctx, int32_t batch, const double* controls) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (!ctx->has_problem) return fail(QOCX_ERR_STATE, "no problem set");
    if (batch < 1) return fail(QOCX_ERR_ARG, "batch must be >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t per = (size_t)ctx->nc * ctx->K;
    double bound = ctx->h0_norm_max;
    if (ctx->K > 0) {
        if (!controls) return fail(QOCX_ERR_ARG, "controls is NULL");
        // One pass over the caller's array: the per-control maxima for the squaring bound, and a
        // copy into a pinned staging buffer the DMA engine reads without a driver-side bounce
        // (4 MB of pageable memory: 0.71 -> 0.27 ms per call at the headline size). The
        // copy is stream-ordered before the kernels of qocx_eval_resident; the staging buffer is
        // only rewritten by the next call, after that evaluation has been synchronised.
        const size_t total = (size_t)batch * per;
        if (ctx->pin_controls_cap < total) {
            if (ctx->pin_controls) (void)hipHostFree(ctx->pin_controls);
            ctx->pin_controls = nullptr;
            ctx->pin_controls_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&ctx->pin_controls, total * sizeof(double), hipHostMallocDefault));
            ctx->pin_controls_cap = total;
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // nothing in flight still reads the staging buffer
        const int K = ctx->K;
        // sum_k |u_k(t)| ||G_k||_1 is largest at a control knot (u is linear between knots, the sum
        // convex): the largest knot sum bounds every step, every Magnus node, and the device's
        // per-step bound (launch_step_table) - tighter than sum_k max_t |u_k(t)| ||G_k||_1
        double smax = 0.0, smid = 0.0, sprev = 0.0;
        double* stage = ctx->pin_controls;
        for (size_t row = 0; row < (size_t)batch * ctx->nc; ++row) {
            const double* src = controls + row * K;
            double* dst = stage + row * K;
            double srow = 0.0;
            for (int k = 0; k < K; ++k) {
                const double v = src[k];
                dst[k] = v;
                srow += fabs(v) * ctx->g_norm_max[k];
            }
            if (!(srow <= smax)) smax = srow;  // also catches NaN
            // midpoint of two knots of the same seed: |u_mid| <= (|u_j| + |u_j+1|) / 2
            if (row % (size_t)ctx->nc != 0) {
                const double m = 0.5 * (sprev + srow);
                if (!(m <= smid)) smid = m;
            }
            sprev = srow;
        }
        if (ctx->nodes == 1 && ctx->nc == ctx->nsteps + 1)
            ctx->norm_bound_mid = (bound + smid) * fabs(ctx->dt) * (1.0 + 1e-12);
        else
            ctx->norm_bound_mid = 1e300;
        bound += smax;
        if (ctx->controls.ensure(total)) return QOCX_ERR_HIP;
        HIP_TRY(hipMemcpyAsync(ctx->controls.p, stage, total * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    }
    bound = magnus_norm_bound(ctx->nodes, bound * fabs(ctx->dt));
    if (!(bound < 1e300)) return fail(QOCX_ERR_ARG, "non-finite controls or Hamiltonian");
    ctx->sbound = pade_scale_count(bound);
    ctx->norm_bound = bound;
    if (ctx->sbound > 10)
        return fail(QOCX_ERR_CAPACITY,
                    "||dt H||_1 bound needs more than 2^10 squaring sub-steps per step; reduce dt");
    ctx->slot_cap = ((size_t)ctx->nsteps << ctx->sbound) + 1;
    ctx->B = batch;
    ctx->have_results = false;
    ctx->explicit_mode = false;
    return 0;
}

int qocx_upload_generators(qocx_ctx* ctx, int32_t batch, const double* generators) {
    if (!ctx || !generators) return fail(QOCX_ERR_ARG, "NULL argument");
    if (!ctx->has_problem) return fail(QOCX_ERR_STATE, "no problem set");
    if (batch < 1) return fail(QOCX_ERR_ARG, "batch must be >= 1");
    if (ctx->K != 0 || ctx->nodes != 1)
        return fail(QOCX_ERR_STATE, "explicit generators need a problem with control_count = 0 and "
                                    "magnus_policy M2");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->n, np = ctx->np, nsteps = ctx->nsteps;
    const size_t count = (size_t)batch * nsteps, mat = (size_t)np * np;
    std::vector<double2> padded(count * mat, make_double2(0, 0));
    double worst = 0;
    bool skew = true;
    for (size_t m = 0; m < count; ++m) {
        const double* g = generators + m * (size_t)n * n * 2;
        double norm1 = 0;
        for (int c = 0; c < n; ++c) {
            double col = 0;
            for (int r = 0; r < n; ++r) {
                const double re = g[2 * ((size_t)r * n + c)], im = g[2 * ((size_t)r * n + c) + 1];
                col += std::hypot(re, im);
                padded[m * mat + (size_t)r * np + c] = make_double2(re, im);
                // skew-Hermitian bit for bit: a[r][c] == -conj(a[c][r])
                if (re != -g[2 * ((size_t)c * n + r)] || im != g[2 * ((size_t)c * n + r) + 1])
                    skew = false;
            }
            if (!(col <= norm1)) norm1 = col;
        }
        if (!(norm1 <= worst)) worst = norm1;
    }
    if (!(worst < 1e300)) return fail(QOCX_ERR_ARG, "non-finite generator");
    ctx->sbound = pade_scale_count(worst);
    ctx->norm_bound = worst;
    ctx->norm_bound_mid = 1e300;
    if (ctx->sbound > 10)
        return fail(QOCX_ERR_CAPACITY,
                    "||dt H||_1 needs more than 2^10 squaring sub-steps per step; reduce dt");
    ctx->slot_cap = ((size_t)nsteps << ctx->sbound) + 1;
    if (ctx->gen_rm.upload(padded, ctx->stream)) return QOCX_ERR_HIP;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->B = batch;
    ctx->have_results = false;
    ctx->explicit_mode = true;
    ctx->explicit_hermitian = skew ? 1 : 0;
    return 0;
}

int qocx_download_generator_cotangents(qocx_ctx* ctx, double* out) {
    if (!ctx || !out) return fail(QOCX_ERR_ARG, "NULL argument");
    if (!ctx->have_results || !ctx->have_grads || !ctx->explicit_mode)
        return fail(QOCX_ERR_STATE, "no generator cotangents (qocx_upload_generators + "
                                    "qocx_eval_resident(want_grad = 1) first)");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->n, np = ctx->np;
    const size_t count = (size_t)ctx->B * ctx->nsteps, mat = (size_t)np * np;
    std::vector<double2> padded(count * mat);
    HIP_TRY(hipMemcpy(padded.data(), ctx->genbar_rm.p, padded.size() * sizeof(double2),
                      hipMemcpyDeviceToHost));
    for (size_t m = 0; m < count; ++m)
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                out[2 * ((m * n + r) * n + c)] = padded[m * mat + (size_t)r * np + c].x;
                out[2 * ((m * n + r) * n + c) + 1] = padded[m * mat + (size_t)r * np + c].y;
            }
    return 0;
}

int qocx_set_state_cotangents(qocx_ctx* ctx, int32_t batch, int32_t count, const int32_t* steps,
                              const double* bars) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (!ctx->has_problem) return fail(QOCX_ERR_STATE, "no problem set");
    if (count <= 0) {
        ctx->inj_count = 0;
        return 0;
    }
    if (batch < 1 || !steps || !bars) return fail(QOCX_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = ctx->n, np = ctx->np, S = ctx->S, nsteps = ctx->nsteps;
    std::vector<int> index(nsteps + 1, -1);
    for (int c = 0; c < count; ++c) {
        if (steps[c] < 1 || steps[c] > nsteps || index[steps[c]] >= 0)
            return fail(QOCX_ERR_ARG, "cotangent steps must be distinct and in 1..N-1");
        index[steps[c]] = c;
    }
    std::vector<double2> padded((size_t)batch * count * S * np, make_double2(0, 0));
    for (size_t v = 0; v < (size_t)batch * count * S; ++v)
        for (int i = 0; i < n; ++i)
            padded[v * np + i] = make_double2(bars[2 * (v * n + i)], bars[2 * (v * n + i) + 1]);
    if (ctx->inj_index.upload(index, ctx->stream) || ctx->inj_bars.upload(padded, ctx->stream))
        return QOCX_ERR_HIP;
    ctx->inj_count = count;
    ctx->inj_batch = batch;
    return 0;
}

int qocx_set_chunk(qocx_ctx* ctx, int32_t seeds_per_chunk) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    ctx->chunk_user = seeds_per_chunk < 0 ? 0 : seeds_per_chunk;
    return 0;
}

int qocx_set_pipeline(qocx_ctx* ctx, int32_t time_segments) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    ctx->pipe_user = time_segments < 0 ? 0 : time_segments;
    return 0;
}

int qocx_set_keep_step_states(qocx_ctx* ctx, int32_t keep) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    ctx->keep_step_states = keep ? 1 : 0;
    return 0;
}

int qocx_eval_resident(qocx_ctx* ctx, int32_t want_grad) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (!ctx->has_problem || ctx->B < 1) return fail(QOCX_ERR_STATE, "no problem / controls");
    HIP_TRY(hipSetDevice(ctx->device));
    const int B = ctx->B, np = ctx->np, mat = np * np, S = ctx->S, K = ctx->K, nsteps = ctx->nsteps;
    const bool explicit_gen = ctx->explicit_mode;
    want_grad = (want_grad && (K > 0 || explicit_gen)) ? 1 : 0;
    if (explicit_gen && want_grad)
        if (ctx->genbar_rm.ensure((size_t)B * nsteps * mat)) return QOCX_ERR_HIP;
    if (ctx->inj_count > 0 && ctx->inj_batch != B)
        return fail(QOCX_ERR_STATE, "state cotangents were set for a different batch size");
    if (ctx->general_path) return eval_general(ctx, want_grad);  // 65 <= n <= 256, or more states than the sweep's LDS (qocx_general.hip)

    // Unit adjoint (qocx_sweep_common.h): a property of the PROBLEM and of the context's knobs, never
    // of the batch size, chunking or segmentation - results stay bit-identical across those.
    // "latency" (the host sets it for the entry points that evaluate ONE control set at a time):
    // where the unit adjoint applies, the two-sided pipeline on four time segments is the lowest
    // latency there is - one seed, n = 32, 1000 steps, forward + gradient: 3.6 ms against 4.15 ms
    // with the blocked sweep and 6.4 ms with one launch of the column-chain sweep; n = 8, 500
    // steps: 1.1 against 2.1 / 1.8 ms (profiles/r03_latency.jsonl) - so it takes precedence
    // over "sweep_impl" = 3 there.
    const bool latency = ctx->knob("latency", 0) != 0;
    // M4 on the M2 kernels (M4LinArgs): the kernels see Ke controls given per step and one node
    const bool m4lin = ctx->m4lin_Ke > 0 && ctx->nodes == 2 && !explicit_gen && ctx->knob("m4_linear", 1);
    const int Kk = m4lin ? ctx->m4lin_Ke : K;  // controls as K1a / K3 see them
    // Dense-state sweep (qocx_sweepd.hip): 8..32 states of a seed as the columns of MFMA GEMMs,
    // with P^-1 in place of the LU factors (K1b's sibling inv_kernel). A property of the problem.
    const bool dense = qocx::sweepd_supports(ctx->nb, S) && ctx->knob("sweep_dense", 1) != 0;
    // Inverse-image sweep (qocx_sweepi.hip): a sub-step is two matrix-vector products with P^-1 from
    // inv_kernel instead of two triangular solves. In latency mode (one control set: the sweep chain
    // is all there is), and always at n <= 16, where Gauss-Jordan on a 16 x 16 matrix costs what its
    // LU costs (0.10 against 0.085 ms per 32 000) and the evaluation is bound by the sweeps: 256 seeds
    // x 1000 steps at n = 8: 3.65 -> 2.79 ms. A property of the problem size, not of the batch.
    const bool inverse_sweep = (latency || (ctx->nb == 1 && ctx->knob("sweep_inverse_small", 1))) &&
                               !dense && qocx::sweepi_supports(ctx->nb, S) &&
                               ctx->knob("sweep_inverse", 1) != 0;
    const bool unit_core = ctx->unit_ok && want_grad && ctx->inj_count == 0 && !explicit_gen && !dense &&
                           (ctx->nodes == 1 || m4lin) && ctx->knob("unit_adjoint", 1);
    // (latency mode, n <= 16: the column-chain sweep is the faster one there - 1.8 against 2.2 us
    // per step - so the two-sided pipeline keeps it; 17 <= n <= 32: two-sided on the blocked sweep)
    const bool sweep3_sel = ctx->knob("sweep_impl", 1) == 3 && ctx->nb <= 2 && !dense && !inverse_sweep &&
                            S <= qocx::sweep3_max_states(ctx->nb) &&
                            !(latency && unit_core && ctx->nb == 1);
    const bool unit = unit_core;  // (both sweeps offer it)
    if (unit)
        if (ctx->lam_scale.ensure((size_t)B * S)) return QOCX_ERR_HIP;
    // chunk size from the memory budget
    const size_t per_seed = (size_t)nsteps * ((size_t)mat * 32 + (size_t)np * 20 + 4) +
                            ctx->slot_cap * S * np * 32 + (size_t)(nsteps + 1) * 4 +
                            (size_t)nsteps * ctx->nodes * std::max(Kk, 1) * 24 +
                            (ctx->nodes > 1 ? (size_t)nsteps * mat * 32 : 0);
    const int nodes = m4lin ? 1 : ctx->nodes;  // nodes of the generator kernels
    // Magnus kernels as four-wave workgroups with every matrix in LDS (qocx_magnus4w.hip)
    const bool magnus4w = nodes > 1 && qocx::magnus4w_supports(ctx->nb, K, ctx->n) && ctx->knob("magnus_4w", 1) != 0;
    int chunk = ctx->chunk_user;
    if (chunk <= 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        size_t have = ctx->q_img.count * 16 + ctx->lu_img.count * 16 + ctx->states.count * 16 +
                      ctx->xs.count * 16;
        size_t budget = (size_t)((double)(free_b + have) * 0.6);
        chunk = (int)std::min<size_t>((size_t)B, std::max<size_t>(1, budget / per_seed));
    }
    chunk = std::min(chunk, B);
    const size_t cm = (size_t)chunk * nsteps;
    if (ctx->q_img.ensure(cm * mat) || ctx->lu_img.ensure(cm * mat) || ctx->dinv.ensure(cm * np) ||
        ctx->perm.ensure(cm * np) || ctx->iperm.ensure(cm * np) || ctx->s_arr.ensure(cm) ||
        ctx->states.ensure((size_t)chunk * ctx->slot_cap * S * np) ||
        ctx->xs.ensure(want_grad ? (size_t)chunk * ctx->slot_cap * S * np : 1) ||
        ctx->offs.ensure((size_t)chunk * (nsteps + 1)) ||
        ctx->gstep.ensure(cm * nodes * std::max(Kk, 1) * (unit ? 2 : 1)) || ctx->cost_out.ensure(B) ||
        (m4lin && (ctx->veff.ensure(cm * Kk) || ctx->gnode.ensure(want_grad ? cm * 2 * K : 1))) ||
        (unit && ctx->offs_x.ensure((size_t)chunk * (nsteps + 1))) ||
        ctx->grads.ensure((size_t)B * ctx->nc * std::max(K, 1)) ||
        ctx->final_out.ensure((size_t)B * S * np))
        return QOCX_ERR_HIP;
    if (ctx->keep_step_states)
        if (ctx->step_states.ensure((size_t)B * (nsteps + 1) * S * np)) return QOCX_ERR_HIP;
    const int magnus_blocks = (int)std::min<size_t>(cm, 1024);
    if (nodes > 1)
        if (ctx->m_rm.ensure(cm * mat) || ctx->mbar_rm.ensure(want_grad ? cm * mat : 1) ||
            ctx->magnus_scratch.ensure(qocx::magnus_scratch_elems(ctx->nb, magnus_blocks)))
            return QOCX_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->status.p, 0, sizeof(int), ctx->stream));

    // Time-segmented pipeline. The serial sweep of a seed is latency bound (one wave, 2(N-1)
    // dependent steps, <= B waves on the whole chip), the other kernels are throughput bound.
    // The steps are therefore cut into `nseg` time segments: the compute stream factors segment
    // after segment (Magnus, K1a, K1b), the high-priority sweep stream follows one segment
    // behind with the forward sweep, then walks back with the adjoint sweep while the compute
    // stream runs K3 on the segments the adjoint sweep has already left.
    const int max_seg = (int)ctx->ev_factored.size();
    if (ctx->lam_buf.ensure((size_t)chunk * S * np)) return QOCX_ERR_HIP;
    hipStream_t cs = ctx->stream;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int bc = std::min(chunk, B - b0);
        ctx->last_chunk = bc;
        int nseg = ctx->pipe_user > 0 ? ctx->pipe_user
                                      : ((size_t)bc * nsteps >= 16384 && nsteps >= 64 ? 8 : 1);
        // (one control set, two-sided: TWO segments - the forward sweep takes the first as soon as it is
        // factored, the adjoint sweep the second, then they swap; every further segment is two more launch
        // latencies on the chain: configs[1] 0.535 -> 0.505 ms, dim 32 x 1000 steps 1.54 -> 1.50)
        if (ctx->pipe_user <= 0 && nseg == 1 && latency && unit && nsteps >= 64) nseg = 2;
        nseg = std::max(1, std::min(std::min(nseg, max_seg), nsteps));
        hipStream_t ss = (nseg == 1) ? cs : ctx->sweep_streams[0];
        // segment boundaries; the last two segments are shorter, because the forward sweep of the
        // last segment and the adjoint sweep of the first one it revisits are exposed
        // Two-sided pipeline (unit adjoint): the adjoint sweep back-propagates the
        // targets from the LAST segment while the forward sweep propagates the states from the
        // FIRST one; the compute stream factors the segments from both ends towards the middle,
        // and K3 follows from the middle outwards once both sweeps have crossed a segment.
        const bool bidir = unit && nseg >= (int)ctx->knob("bidir_min_segments", 2) && ctx->knob("bidir", 1) && (int)ctx->sweep_streams.size() >= 2;
        std::vector<int> lo(nseg + 1);
        {
            std::vector<double> wgt(nseg, 1.0);
            if (bidir) {
                // the two segments in the middle are factored last: what the sweeps still have to
                // do once they exist is exposed, so they are the short ones
                wgt[nseg / 2 - 1] = 0.5; wgt[nseg / 2] = 0.5;
            } else if (nseg >= 4) { wgt[nseg - 2] = 0.6; wgt[nseg - 1] = 0.35; }
            if (const char* env = qocx::diag_getenv("QOCX_SEG_WEIGHTS")) {  // experiments: "w0,w1,..."
                std::vector<double> user;
                for (const char* p = env; *p;) {
                    char* end = nullptr;
                    user.push_back(strtod(p, &end));
                    if (end == p) break;
                    p = (*end == ',') ? end + 1 : end;
                }
                if ((int)user.size() == nseg) wgt = user;
            }
            double tot = 0, run = 0;
            for (double w : wgt) tot += w;
            lo[0] = 0;
            for (int i = 0; i < nseg; ++i) {
                run += wgt[i];
                lo[i + 1] = std::max(lo[i] + 1, (int)llround(nsteps * run / tot));
            }
            lo[nseg] = nsteps;
            for (int i = nseg - 1; i > 0; --i) lo[i] = std::min(lo[i], lo[i + 1] - 1);
        }

        const int dbg_skip_early = (int)ctx->knob("dbg_skip", 0);  // bit 3: K1a stores no Q (timing)
        const bool one_wave_k1a = qocx::diag_getenv("QOCX_PQ1") != nullptr;  // (experiments: the one-wave K1a)
        qocx::FactorArgs fa;
        fa.controls = ctx->controls.p ? ctx->controls.p + (size_t)b0 * ctx->nc * K : nullptr;
        fa.interp = ctx->interp.p;
        fa.h0_cimg = ctx->h0_cimg.p;
        fa.g_cimg = ctx->g_cimg.p;
        fa.K = K; fa.nc = ctx->nc; fa.nsteps = nsteps; fa.nt = ctx->nt; fa.dt = ctx->dt;
        qocx::M4LinArgs m4;
        if (m4lin) {
            m4.controls = fa.controls; m4.interp = ctx->interp.p;
            m4.K = K; m4.Ke = Kk; m4.nc = ctx->nc; m4.nsteps = nsteps; m4.S = S;
            m4.f0dt = (std::sqrt(3.0) / 12) * ctx->dt;
            m4.veff = ctx->veff.p; m4.gstep = ctx->gstep.p; m4.gnode = ctx->gnode.p;
            m4.lam_scale = unit ? ctx->lam_scale.p + (size_t)b0 * S : nullptr;
            m4.total = (size_t)bc * nsteps;
            time_begin(ctx, 0, cs);
            qocx::launch_m4lin_controls(m4, cs);
            time_end(ctx, cs);
            fa.controls = ctx->veff.p; fa.interp = ctx->interp_id.p; fa.g_cimg = ctx->ge_cimg.p;
            fa.K = Kk; fa.nc = nsteps;
        }
        fa.hermitian = explicit_gen ? ctx->explicit_hermitian : ctx->hermitian;
        fa.n = ctx->n;
        fa.skip_q = (dbg_skip_early & 8) ? 1 : 0;
        fa.dbg = (int)ctx->knob("k1a_dbg", 0);  // (diagnostic build: qocx_device.h)
        fa.stamps = nullptr;
        if (qocx::kDiagBuild && ctx->knob("k1a_stamps", 0)) {
            if (ctx->stamps.ensure(1024 * 16)) return QOCX_ERR_HIP;
            HIP_TRY(hipMemsetAsync(ctx->stamps.p, 0, 1024 * 16 * sizeof(unsigned long long), cs));
            HIP_TRY(hipStreamSynchronize(cs));
            fa.stamps = ctx->stamps.p;
        }
        fa.lu_mfma = (int)ctx->knob("lu_mfma", 1);
        fa.lu_dpp = (int)ctx->knob("lu_dpp", 1);
        fa.herm_tiles = (int)ctx->knob("k1a_herm4", 1);
        if (ctx->lu_fallbacks.ensure(1)) return QOCX_ERR_HIP;
        if (b0 == 0) HIP_TRY(hipMemsetAsync(ctx->lu_fallbacks.p, 0, sizeof(int), cs));
        fa.lu_fallbacks = ctx->lu_fallbacks.p;
        fa.pade_policy = (int)ctx->knob("pade_order", 0);  // 0: by norm (qocx_wave.h), 13: always 13
        fa.prefer_low = ctx->norm_bound < 2.539398330063230e-01 ? 2 : ctx->norm_bound < 2.097847961257068 ? 1 : 0;  // theta_5, theta_9
        fa.q_img = ctx->q_img.p; fa.lu_img = ctx->lu_img.p;
        fa.s_arr = ctx->s_arr.p; fa.status = ctx->status.p;
        // K1b fused into the two-wave K1a (17 <= n <= 32; knob "fuse_lu" 0 restores the two kernels)
        const bool fused_lu = ctx->nb == 2 && !one_wave_k1a && !dense && !inverse_sweep &&
                              ctx->knob("fuse_lu", 1) != 0;
        fa.fuse_lu = fused_lu ? 1 : 0;
        fa.dinv = ctx->dinv.p; fa.perm = ctx->perm.p; fa.iperm = ctx->iperm.p;
        // Step table (two-wave K1a, structured M2 problem): one small kernel interpolates the
        // controls of every step and decides its Pade order and squaring count from the bound
        // dt (||H0||_1 + sum |u_k| ||G_k||_1); K1a and K3 then read both instead of interpolating
        // and (K1a) reducing a norm behind a barrier. Knob "step_table" 0 restores the old path.
        const bool step_table = ctx->nb == 2 && !one_wave_k1a && !explicit_gen && nodes == 1 && !m4lin &&
                                !dense && K > 0 && ctx->g_norm_dev.p != nullptr &&
                                ctx->knob("step_table", 1) != 0;
        if (step_table) {
            if (ctx->ustep.ensure((size_t)bc * nsteps * K)) return QOCX_ERR_HIP;
            qocx::StepTableArgs ta;
            ta.controls = fa.controls; ta.interp = fa.interp; ta.K = K; ta.nc = ctx->nc;
            ta.nsteps = nsteps; ta.batch = bc; ta.dt = ctx->dt; ta.h0_norm = ctx->h0_norm_max;
            ta.g_norm = ctx->g_norm_dev.p; ta.pade_policy = fa.pade_policy;
            ta.sq_max = std::min(30, ctx->sbound);
            // only the three-wave K1a (orders 3 and 5) will be launched: no step above order 5
            qocx::FactorArgs probe = fa;
            probe.direct = 1;
            fa.three_wave = (int)ctx->knob("k1a_three", 1);
            // (1: the second halves of the factorisations four to a wave in a kernel of their own behind K1a,
            // 2: on the factor side stream, beside the next segment's K1a. Measured, profiles/r05_k1a_four.txt:
            // K1a 0.650 -> 0.587 ms, the second kernel 0.061 ms - it moves 4 KB per step in and out, 262 MB
            // per segment -, the evaluation 8.05 -> 8.05 (1) / 7.93 ms (2). Off: 1.3 % for a kernel and a
            // stream more and 2 GB more traffic per evaluation.)
            fa.four_steps = (int)ctx->knob("k1a_four", 0);
            fa.gen_share = (int)ctx->knob("k1a_share", 2);
            if (fa.four_steps == 2 && ctx->lu_stream == nullptr) fa.four_steps = 1;
            // (the bound at the step midpoints, where it applies, speaks for the two-wave K1a only: the
            // four-wave kernels and the slot capacity keep the bound over the knots)
            if (fa.three_wave && qocx::pq3_supports(probe) &&
                std::min(ctx->norm_bound, ctx->norm_bound_mid) < 2.539398330063230e-01) {
                fa.prefer_low = 2;
                ta.order_max = 5;
            }
            ta.ustep = ctx->ustep.p; ta.s_arr = fa.s_arr; ta.status = fa.status;
            qocx::launch_step_table(ta, cs);
            fa.controls = ctx->ustep.p; fa.nc = nsteps; fa.direct = 1;
        }
        qocx::LuArgs la;
        la.lu_img = fa.lu_img; la.dinv = ctx->dinv.p; la.perm = ctx->perm.p;
        la.iperm = ctx->iperm.p; la.status = ctx->status.p; la.nsteps = nsteps; la.n = ctx->n;
        la.dbg = ((dbg_skip_early & 16) ? 1 : 0) | ((ctx->knob("k1a_dbg", 0) & 8) ? 2 : 0) |
                 ((ctx->knob("k1a_dbg", 0) & 16) ? 4 : 0) | ((ctx->knob("k1a_dbg", 0) & 32) ? 8 : 0);
        la.inverse = (dense || inverse_sweep) ? 1 : 0;
        // every Pade denominator of the evaluation diagonally dominant by the margin of qocx_lu5.h
        // (eps_m(theta) <= 0.40 for every order m at the host's bound theta of the step norm)
        la.all_dominant = (pade_eps_max(ctx->norm_bound) <= 0.40 && ctx->knob("lu_dpp", 1) != 0) ? 1 : 0;
        // n <= 8: two consecutive steps of a seed as the diagonal blocks of one 16 x 16 tile through K1a and
        // K1b (pade_pq8_kernel, inv16_dpp_kernel<1, true>); the sweeps and K3 see the usual images
        const bool pack8 = ctx->nb == 1 && ctx->n <= 8 && inverse_sweep && !dense && la.all_dominant && !explicit_gen &&
                           nodes == 1 && !m4lin && ctx->knob("pack8", 1) != 0;
        fa.pack8 = la.pack8 = pack8 ? 1 : 0;
        // One control set at a time (latency mode), inverse-image sweep: K1b's sibling umul_kernel leaves the
        // propagator itself in the Q image; the sweeps apply ONE matrix per sub-step, the adjoint sweep hands
        // lambda' to K3, which forms x = P^-H lambda' from the P^-1 image (knob "sweep_umode").
        const bool umode = latency && inverse_sweep && !dense && ctx->nb <= 2 && ctx->knob("sweep_umode", 1) != 0;
        if (umode && ctx->qt_img.ensure((size_t)chunk * nsteps * mat)) return QOCX_ERR_HIP;
        la.redo = nullptr;
        la.fallbacks = ctx->lu_fallbacks.p;
        if (ctx->nb == 4 && ctx->knob("lu_mfma", 1) != 0) {  // qocx_lu4m.hip in front of lu4_kernel
            if (ctx->lu_redo.ensure((size_t)bc * nsteps)) return QOCX_ERR_HIP;
            la.redo = ctx->lu_redo.p;
            // (Measured and not kept: the nine-tile factorisation INSIDE the nine-tile K1a, P through an
            // LDS image as at n <= 32 - 3.81 ms per launch against 2.48 + 0.78 apart: wave 0 factors for
            // 60 000 cycles while the workgroup's 46 KiB of LDS stay allocated.)
        }
        qocx::MagnusArgs ma;
        ma.controls = fa.controls; ma.interp = ctx->interp.p;
        ma.h0_cimg = ctx->h0_cimg.p; ma.g_cimg = ctx->g_cimg.p;
        ma.K = K; ma.nc = ctx->nc; ma.nsteps = nsteps; ma.nt = ctx->nt; ma.nodes = nodes;
        ma.dt = ctx->dt; ma.scratch = ctx->magnus_scratch.p; ma.n = ctx->n;
        ma.skew = (ctx->hermitian && !ctx->knob("magnus_general", 0)) ? 1 : 0;
        qocx::SweepArgs sa;
        sa.q_img = fa.q_img; sa.lu_img = fa.lu_img; sa.dinv = la.dinv;
        sa.perm = la.perm; sa.iperm = la.iperm; sa.s_arr = fa.s_arr;
        sa.psi0 = ctx->psi0.p;
        sa.umode = umode ? 1 : 0;
        sa.qt_img = umode ? ctx->qt_img.p : nullptr;
        sa.S = S; sa.nsteps = nsteps; sa.cost_eval_step = ctx->ces; sa.want_grad = want_grad;
        sa.n = ctx->knob("sweep_nine", 1) ? ctx->n : 0;
        sa.has_step_costs = ctx->has_step_costs; sa.slot_cap = ctx->slot_cap;
        sa.cost_count = ctx->cost_count; sa.costs = ctx->costs.p;
        sa.cost_vectors = ctx->cost_vectors.p; sa.cost_counts = ctx->cost_counts.p;
        sa.states = ctx->states.p;
        sa.xs = ctx->xs.p;
        sa.offs = ctx->offs.p;
        sa.cost_out = ctx->cost_out.p + b0;
        sa.final_out = ctx->final_out.p + (size_t)b0 * S * np;
        sa.step_states = ctx->keep_step_states
                             ? ctx->step_states.p + (size_t)b0 * (nsteps + 1) * S * np : nullptr;
        sa.status = ctx->status.p;
        sa.lam_buf = ctx->lam_buf.p;
        sa.loader = (int)ctx->knob("sweep_loader", 0);
        sa.batch = bc;
        sa.onebuf = (int)ctx->knob("sweep_onebuf", 1);
        sa.one_state = (int)ctx->knob("sweep_one", 1);
        sa.dbg = (int)ctx->knob("sweep3_dbg", 0);  // (bits 8, 9: the column-chain sweep fetches nothing)
        sa.stamps = nullptr;
        if (ctx->knob("sweep3_stamps", 0)) {
            if (ctx->stamps.ensure((size_t)B * 32)) return QOCX_ERR_HIP;
            HIP_TRY(hipMemsetAsync(ctx->stamps.p, 0, (size_t)B * 32 * sizeof(unsigned long long), cs));
            HIP_TRY(hipStreamSynchronize(cs));
            sa.stamps = ctx->stamps.p + (size_t)b0 * 32;
        }
        // blocked-inverse sweep (three wavefronts per seed) unless switched off or the states do
        // not fit beside its LDS ring
        // "sweep_impl": 1 (default) column-chain sweep, 3 blocked sweep. ONE implementation serves
        // every batch size, chunking and segmentation of a context, so that results stay bit
        // identical across them (tests/test_gpu_engine.py::test_chunked_equals_unchunked,
        // test_gpu_fullsize.py). Measured (profiles/r02_sweep_ab.jsonl): the blocked sweep takes
        // 2.1 us per step against 3.2 us when it has the chip to itself - a single-seed evaluation
        // (n = 32, 1000 steps) 4.2 ms against 6.4 ms, 64 seeds 5.8 against 8.1 ms - but inside
        // the segmented pipeline at 256 seeds it loses (14.3 against 13.4 ms): its workgroup owns
        // the CU's LDS, so K1a / K1b / K3 cannot run beside it. The host package selects it for
        // the single-control-set entry points (latency mode), the batched evaluator keeps 1.
        const bool sweep3 = sweep3_sel;
        sa.unit_adjoint = unit ? 1 : 0;
        sa.lam_scale = unit ? ctx->lam_scale.p + (size_t)b0 * S : nullptr;
        sa.offs_x = unit ? ctx->offs_x.p : nullptr;
        // "sweep3_phases": bit 0 forward launches, bit 1 adjoint launches (and combined ones)
        const int s3_phases = (int)ctx->knob("sweep3_phases", 3);
        // "dbg_skip" (timing experiments only, results are garbage): bit 0 no forward sweep,
        // bit 1 no adjoint sweep, bit 2 no K3
        const int dbg_skip = (int)ctx->knob("dbg_skip", 0);
        auto run_sweep = [&](const qocx::SweepArgs& a, int count, hipStream_t st) {
            if ((dbg_skip & 1) && (a.phase & 1)) return;
            if ((dbg_skip & 2) && (a.phase & 2)) return;
            const bool use3 = sweep3 && ((a.phase & 2) ? (s3_phases & 2) : (s3_phases & 1));
            if (dense) qocx::launch_sweepd(a, count, st);
            else if (inverse_sweep) qocx::launch_sweepi(ctx->nb, a, count, st);
            else if (use3) qocx::launch_sweep3(ctx->nb, a, count, st);
            else qocx::launch_sweep(ctx->nb, a, count, st);
        };
        sa.inj_count = ctx->inj_count;
        sa.inj_index = ctx->inj_count > 0 ? ctx->inj_index.p : nullptr;
        sa.inj_bars = ctx->inj_count > 0
                          ? ctx->inj_bars.p + (size_t)b0 * ctx->inj_count * S * np : nullptr;

        // `fs`: the stream of this segment's factor kernels (the compute stream; in the two-sided
        // pipeline every other segment goes to a second one, so that a K1a grid fills the CUs the
        // previous one is draining from - knob "k1a_streams"; measured 8.08-8.15 ms against 8.06-8.07 ms
        // with one stream, profiles/r05_k1a_streams.jsonl, so the default stays at one)
        auto factor_segment = [&](int i, hipStream_t fs) -> int {
            // (Measured and dropped: K1a / K1b of a segment as 2, 4 or 8 pairs of sub-launches, so
            // that K1b might find P in the last-level cache: 13.1 / 13.8 / 15.5 ms against 12.7 -
            // the launch tails cost more than any cache hit returns.)
            const int plo = lo[i], len = lo[i + 1] - lo[i];
            fa.step0 = plo; fa.seg_len = len;
            time_begin(ctx, 0, fs);
            if (explicit_gen) {
                // generators sampled by the host (opaque Hamiltonian): [seed][step] row-major
                qocx::FactorArgs fe = fa;
                qocx::launch_pq_explicit(ctx->nb, ctx->gen_rm.p + (size_t)b0 * nsteps * mat, np, fe,
                                         bc * len, fs);
            } else if (nodes > 1) {
                ma.step0 = plo; ma.seg_len = len; ma.total = (size_t)bc * len;
                ma.m_rm = ctx->m_rm.p; ma.mbar_rm = nullptr; ma.gstep = nullptr;
                if (magnus4w) qocx::launch_magnus4w_fwd(ma, bc, fs);
                else qocx::launch_magnus_fwd(ctx->nb, ma, (int)std::min<size_t>(ma.total, magnus_blocks), fs);
                qocx::launch_pq_explicit(ctx->nb, ma.m_rm, np, fa, bc * len, fs);
            } else {
                qocx::launch_pq(ctx->nb, fa, len, bc, fs);
            }
            time_end(ctx, fs);
            if (!explicit_gen && nodes == 1 && qocx::pq_second_pending(ctx->nb, fa, len)) {
                // the second halves of the segment's factorisations (memory-bound: 4 KB in and out per
                // step) on the side stream, beside the K1a launch of the next segment
                HIP_TRY(hipEventRecord(ctx->ev_pq[i], fs));
                HIP_TRY(hipStreamWaitEvent(ctx->lu_stream, ctx->ev_pq[i], 0));
                time_begin(ctx, 4, ctx->lu_stream);
                qocx::launch_pq3_second(fa, len, bc, ctx->lu_stream);
                time_end(ctx, ctx->lu_stream);
                HIP_TRY(hipEventRecord(ctx->ev_factored[i], ctx->lu_stream));
                if (nseg <= 1) HIP_TRY(hipStreamWaitEvent(fs, ctx->ev_factored[i], 0));
                return 0;
            }
            la.step0 = plo; la.seg_len = len;
            // (K1b on a stream of its own, beside the next segment's K1a: with the v5 kernels no
            // gain; with the two-wave K1a (240 registers) and K1b (160) sharing SIMDs 1 % - K1a
            // then takes 1.08 ms per launch beside K1b instead of 0.80 + 0.32 ms in sequence.
            // Not kept: one more stream and eight more events for 0.14 ms.)
            // n > 32: K1b of this segment on a stream of its own, beside K1a of the next segment - the
            // four-wave K1a is bound by the matrix pipe, the two-wave / one-wave MFMA factorisation by
            // its pivot chains, and both fit a CU (knob "lu_stream"; n = 64: see DESIGN.md section 14).
            // (Measured at n <= 32 with the round-2 kernels: no gain there - comment above.)
            const bool lu_apart = !fused_lu && ctx->nb == 4 && nseg > 1 && ctx->lu_stream != nullptr &&
                                  ctx->knob("lu_stream", 1) != 0;
            if (lu_apart) {
                HIP_TRY(hipEventRecord(ctx->ev_pq[i], fs));
                HIP_TRY(hipStreamWaitEvent(ctx->lu_stream, ctx->ev_pq[i], 0));
                time_begin(ctx, 4, ctx->lu_stream);
                qocx::launch_lu(ctx->nb, la, (size_t)bc * len, ctx->lu_stream);
                time_end(ctx, ctx->lu_stream);
                HIP_TRY(hipEventRecord(ctx->ev_factored[i], ctx->lu_stream));
                return 0;
            }
            if (!fused_lu) {
                time_begin(ctx, 4, cs);
                qocx::launch_lu(ctx->nb, la, la.pack8 ? (size_t)bc * ((len + 1) / 2) : (size_t)bc * len, fs);
                // one control set: the propagator U = P^-1 Q in place of Q, one product per sweep sub-step
                if (umode) qocx::launch_umul(ctx->nb, la, fa.q_img, ctx->qt_img.p, (size_t)bc * len, fs);
                time_end(ctx, cs);
            }
            if (nseg > 1) HIP_TRY(hipEventRecord(ctx->ev_factored[i], fs));
            return 0;
        };
        // forward sweep over segment i on stream st (behind the segment's factorisation)
        auto forward_segment = [&](int i, hipStream_t st) -> int {
            if (nseg > 1) HIP_TRY(hipStreamWaitEvent(st, ctx->ev_factored[i], 0));
            // (While the sweep needed a whole SIMD - 366 registers - the compute stream also
            // waited here until the sweep stream had passed its wait, or the next K1a grid
            // starved the sweep. At 272 registers the sweep fits beside one K1a or K3 wave and
            // the hand-shake only cost time: 14.1 -> 13.95 ms without it.)
            sa.j_begin = lo[i]; sa.j_end = lo[i + 1];
            sa.phase = (nseg == 1) ? (want_grad ? 3 : 1) : 1;
            time_begin(ctx, 1, st);
            run_sweep(sa, bc, st);
            time_end(ctx, st);
            return 0;
        };
        // (tail_ring: bit 0 the forward, bit 1 the adjoint sweeps enqueued behind the LAST factor launch get
        // two operand sets in LDS - nothing but K3 shares the CUs with them then; knob "sweep_tail_ring")
        bool in_tail = false;
        const int tail_ring = (int)ctx->knob("sweep_tail_ring", 0);
        auto forward_range = [&](int jb, int je, hipStream_t st) {  // steps [jb, je), no wait
            sa.j_begin = jb; sa.j_end = je; sa.phase = 1;
            sa.ring2 = (in_tail && (tail_ring & 1)) ? 1 : 0;
            time_begin(ctx, 1, st);
            run_sweep(sa, bc, st);
            time_end(ctx, st);
            sa.ring2 = 0;
        };
        auto adjoint_range = [&](int jb, int je, hipStream_t st) {
            sa.j_begin = jb; sa.j_end = je; sa.phase = 2;
            sa.ring2 = (in_tail && (tail_ring & 2)) ? 1 : 0;
            time_begin(ctx, 1, st);
            run_sweep(sa, bc, st);
            time_end(ctx, st);
            sa.ring2 = 0;
        };
        auto adjoint_segment = [&](int i, hipStream_t st) -> int {
            adjoint_range(lo[i], lo[i + 1], st);
            HIP_TRY(hipEventRecord(ctx->ev_swept[i], st));
            return 0;
        };
        qocx::KrylovArgs ka;
        ka.controls = fa.controls;
        ka.interp = fa.interp;
        ka.h0_rimg = ctx->h0_rimg.p; ka.h0_timg = ctx->h0_timg.p;
        ka.g_rimg = m4lin ? ctx->ge_rimg.p : ctx->g_rimg.p;
        ka.g_timg = m4lin ? ctx->ge_timg.p : ctx->g_timg.p;
        ka.K = fa.K; ka.nc = fa.nc; ka.nsteps = nsteps; ka.nt = ctx->nt; ka.S = S;
        ka.lds_pad = (int)ctx->knob("k3_lds_pad", 0);
        ka.umode = umode ? 1 : 0;
        ka.pinv_img = fa.lu_img;
        ka.direct = fa.direct;
        ka.n = ctx->n;
        ka.dt = ctx->dt; ka.s_arr = ctx->s_arr.p;
        ka.offs = ctx->offs.p;
        ka.offs_x = unit ? ctx->offs_x.p : nullptr;
        ka.states = ctx->states.p;
        ka.xs = ctx->xs.p;
        ka.slot_cap = ctx->slot_cap;
        ka.gstep = ctx->gstep.p;
        ka.m_rm = nodes > 1 ? ctx->m_rm.p : nullptr;
        ka.mbar_rm = nodes > 1 ? ctx->mbar_rm.p : nullptr;
        if (explicit_gen) {
            ka.m_rm = ctx->gen_rm.p + (size_t)b0 * nsteps * mat;
            ka.mbar_rm = want_grad ? ctx->genbar_rm.p + (size_t)b0 * nsteps * mat : nullptr;
        }
        ka.skew = explicit_gen ? ctx->explicit_hermitian : ctx->hermitian;
        auto krylov_range = [&](int jb, int je) -> int {
            const int len = je - jb;
            ka.step0 = jb;
            time_begin(ctx, 2, cs);
            if (dbg_skip & 4) {
            } else if (dense && ctx->knob("krylov_dense", 0)) {
                // (K3 on the matrix cores, qocx_sweepd.hip k3d: correct, and no faster - FP64 MFMA
                // and FP64 VALU peaks are equal on this chip and the vector-unit K3 already runs at
                // 44 TFLOP/s: 21.3 against 19.9 ms per 256 000 steps at S = 32. Off by default.)
                qocx::launch_krylovd(ka, len, bc, cs);
            } else {
                qocx::launch_krylov(ctx->nb, ka, len, bc, cs);
            }
            if (nodes > 1) {
                ma.step0 = jb; ma.seg_len = len; ma.total = (size_t)bc * len;
                ma.m_rm = nullptr; ma.mbar_rm = ka.mbar_rm; ma.gstep = ka.gstep;
                if (magnus4w) qocx::launch_magnus4w_vjp(ma, bc, cs);
                else qocx::launch_magnus_vjp(ctx->nb, ma,
                                             (int)std::min<size_t>(ma.total, magnus_blocks), cs);
            }
            time_end(ctx, cs);
            return 0;
        };
        auto krylov_segment = [&](int i) -> int { return krylov_range(lo[i], lo[i + 1]); };
#define QOCX_STEP(call)            \
    do {                           \
        const int rc_ = (call);    \
        if (rc_ != 0) return rc_;  \
    } while (0)
        if (bidir) {
            hipStream_t sf = ctx->sweep_streams[0], sb = ctx->sweep_streams[1];
            // factor from both ends towards the middle; each sweep takes a segment as soon as it
            // is factored AND the sweep has finished the one before it (stream order)
            // The sweeps and K3 may work in PIECES of a segment (knob "k3_split"), so that what is
            // left of K3 once the sweeps have finished is a piece, not a segment.  Measured at
            // configs[1]: 1 piece 12.86 ms, 2 -> 13.00, 3 -> 13.19, 4 -> 13.55 (every launch
            // refills its pipeline), so the default is whole segments.
            // "k3_split": pieces of every segment; "k3_split_outer": pieces of the FIRST and the LAST
            // segment only - the two whose sweeps finish last: with them in pieces all that is left of
            // K3 once the sweeps have ended is a piece of a segment.
            const int max_pieces = (int)ctx->ev_fwd.size();
            int parts_in = (int)std::max<int64_t>(1, ctx->knob("k3_split", 1));
            // (a small launch - one control set - is a chain of launch latencies: whole segments there,
            // configs[1] 0.57 -> 0.53 ms)
            const bool small_launch = (size_t)bc * nsteps < 16384;
            int parts_out = (int)std::max<int64_t>(parts_in, ctx->knob("k3_split_outer", small_launch ? 1 : 3));
            if ((nseg - 2) * parts_in + 2 * parts_out > max_pieces) parts_in = parts_out = 1;
            struct Piece { int lo, hi; };
            std::vector<Piece> piece;
            std::vector<int> first(nseg + 1, 0);
            for (int i = 0; i < nseg; ++i) {
                const int np_i = (i == 0 || i == nseg - 1) ? parts_out : parts_in;
                first[i] = (int)piece.size();
                for (int part = 0; part < np_i; ++part)
                    piece.push_back({lo[i] + (int)((int64_t)(lo[i + 1] - lo[i]) * part / np_i),
                                     lo[i] + (int)((int64_t)(lo[i + 1] - lo[i]) * (part + 1) / np_i)});
            }
            first[nseg] = (int)piece.size();
            const int P = (int)piece.size();
            std::vector<char> factored(nseg, 0);
            int next_f = 0, next_b = nseg - 1;
            // two factor streams: the second one starts behind everything the compute stream has
            // enqueued so far (the step table, the previous evaluation's tail)
            const bool two_k1a = fused_lu && ctx->lu_stream != nullptr && ctx->knob("k1a_streams", 1) >= 2;
            // (one control set: 0.478 -> 0.456 ms at configs[1], 1.10 -> 1.05 ms at dim 32 x 1000 steps; the
            // 256-seed evaluation, whose factor launches are what it waits for: 7.87 -> 8.00 ms)
            const bool adj_first = ctx->knob("bidir_adj_first", (size_t)bc * nsteps < 16384 ? 1 : 0) != 0;
            if (two_k1a) {
                HIP_TRY(hipEventRecord(ctx->ev_pq[nseg], cs));
                HIP_TRY(hipStreamWaitEvent(ctx->lu_stream, ctx->ev_pq[nseg], 0));
            }
            for (int t = 0; t < nseg; ++t) {
                // (the adjoint sweep is the slower of the two - it gathers its images transposed -: ITS side
                // first, knob "bidir_adj_first")
                const bool back = adj_first ? (t % 2 == 0) : (t % 2 == 1);
                const int i = back ? nseg - 1 - t / 2 : t / 2;
                QOCX_STEP(factor_segment(i, (two_k1a && (t % 2 == 1)) ? ctx->lu_stream : cs));
                factored[i] = 1;
                in_tail = (t == nseg - 1);
                while (next_f < nseg && factored[next_f]) {
                    HIP_TRY(hipStreamWaitEvent(sf, ctx->ev_factored[next_f], 0));
                    for (int p = first[next_f]; p < first[next_f + 1]; ++p) {
                        if (piece[p].hi > piece[p].lo) forward_range(piece[p].lo, piece[p].hi, sf);
                        HIP_TRY(hipEventRecord(ctx->ev_fwd[p], sf));
                    }
                    ++next_f;
                }
                while (next_b >= 0 && factored[next_b]) {
                    HIP_TRY(hipStreamWaitEvent(sb, ctx->ev_factored[next_b], 0));
                    for (int p = first[next_b + 1] - 1; p >= first[next_b]; --p) {
                        if (piece[p].hi > piece[p].lo) adjoint_range(piece[p].lo, piece[p].hi, sb);
                        HIP_TRY(hipEventRecord(ctx->ev_swept[p], sb));
                    }
                    --next_b;
                }
            }
            // K3 from the middle outwards: a piece is complete once the forward sweep (going up)
            // and the adjoint sweep (going down) have both crossed it
            std::vector<int> order;
            const int mid = first[nseg / 2];
            for (int d = 0; d < P; ++d) {
                const int up = mid + d, down = mid - 1 - d;
                if (up < P) order.push_back(up);
                if (down >= 0) order.push_back(down);
            }
            for (int p : order) {
                if (piece[p].hi <= piece[p].lo) continue;
                HIP_TRY(hipStreamWaitEvent(cs, ctx->ev_fwd[p], 0));
                HIP_TRY(hipStreamWaitEvent(cs, ctx->ev_swept[p], 0));
                QOCX_STEP(krylov_range(piece[p].lo, piece[p].hi));
            }
        } else {
            // ---- factor + forward sweep, segment by segment --------------------------------
            for (int i = 0; i < nseg; ++i) {
                QOCX_STEP(factor_segment(i, cs));
                QOCX_STEP(forward_segment(i, ss));
            }
            // ---- adjoint sweep walks back; K3 follows on the compute stream ------------------
            // (Measured and dropped: evaluating a chunk as two seed halves with sweep streams of
            // their own, so that the first half's adjoint sweep runs under the second half's
            // factorisation. Two sweeps then share the chip with half-size grids whose segments
            // take as long as a sweep segment: 16.6 ms against 14.6 ms.)
            if (nseg > 1 && want_grad) {
                for (int i = nseg - 1; i >= 0; --i) QOCX_STEP(adjoint_segment(i, ss));
            } else if (nseg > 1) {
                HIP_TRY(hipEventRecord(ctx->ev_swept[0], ss));
                HIP_TRY(hipStreamWaitEvent(cs, ctx->ev_swept[0], 0));
            }
            if (want_grad)
                for (int i = nseg - 1; i >= 0; --i) {
                    if (nseg > 1) HIP_TRY(hipStreamWaitEvent(cs, ctx->ev_swept[i], 0));
                    QOCX_STEP(krylov_segment(i));
                }
        }
#undef QOCX_STEP
        if (want_grad) {
            qocx::ScatterArgs sc;
            sc.gstep = ka.gstep; sc.row_ptr = ctx->row_ptr.p; sc.col_step = ctx->col_step.p;
            sc.weight = ctx->weight.p;
            sc.grads = ctx->grads.p + (size_t)b0 * ctx->nc * K;
            sc.B = bc; sc.nc = ctx->nc; sc.K = K; sc.nsteps = nsteps * ctx->nodes;
            sc.lam_scale = unit ? ctx->lam_scale.p + (size_t)b0 * S : nullptr;
            sc.S = S;
            time_begin(ctx, 3, cs);
            if (m4lin) {  // effective-control cotangents -> node cotangents (applies the scalar)
                qocx::launch_m4lin_chain(m4, cs);
                sc.gstep = ctx->gnode.p;
                sc.lam_scale = nullptr;
            }
            qocx::launch_scatter(sc, cs);
            time_end(ctx, cs);
        }
    }
    HIP_TRY(hipGetLastError());
    int status = 0;
    HIP_TRY(hipMemcpyAsync(&status, ctx->status.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    time_collect(ctx);
    if (status & 2) return fail(QOCX_ERR_ARG, "non-finite generator norm");
    if (status & 1) return fail(QOCX_ERR_SINGULAR, "Singular matrix");
    if (status & 4) return fail(QOCX_ERR_CAPACITY, "squaring sub-step capacity exceeded");
    ctx->have_results = true;
    ctx->have_grads = want_grad != 0;
    ctx->have_step_states = ctx->keep_step_states != 0;
    return 0;
}

int qocx_download_results(qocx_ctx* ctx, double* cost_out, double* grad_out, double* final_out) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (!ctx->have_results) return fail(QOCX_ERR_STATE, "no evaluation results");
    HIP_TRY(hipSetDevice(ctx->device));
    const int B = ctx->B, np = ctx->np, S = ctx->S, n = ctx->n;
    if (cost_out)
        HIP_TRY(hipMemcpyAsync(cost_out, ctx->cost_out.p, (size_t)B * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
    if (grad_out) {
        if (!ctx->have_grads) return fail(QOCX_ERR_STATE, "gradients were not computed");
        HIP_TRY(hipMemcpyAsync(grad_out, ctx->grads.p, (size_t)B * ctx->nc * ctx->K * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
    }
    std::vector<double2> fin;
    if (final_out) {
        fin.resize((size_t)B * S * np);
        HIP_TRY(hipMemcpyAsync(fin.data(), ctx->final_out.p, fin.size() * sizeof(double2),
                               hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (final_out)
        for (size_t v = 0; v < (size_t)B * S; ++v)
            for (int i = 0; i < n; ++i) {
                final_out[2 * (v * n + i)] = fin[v * np + i].x;
                final_out[2 * (v * n + i) + 1] = fin[v * np + i].y;
            }
    return 0;
}

int qocx_download_step_states(qocx_ctx* ctx, double* states_out) {
    if (!ctx || !states_out) return fail(QOCX_ERR_ARG, "NULL argument");
    if (!ctx->have_results || !ctx->have_step_states)
        return fail(QOCX_ERR_STATE, "step states were not kept (qocx_set_keep_step_states)");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t nvec = (size_t)ctx->B * (ctx->nsteps + 1) * ctx->S;
    std::vector<double2> tmp(nvec * ctx->np);
    HIP_TRY(hipMemcpy(tmp.data(), ctx->step_states.p, tmp.size() * sizeof(double2),
                      hipMemcpyDeviceToHost));
    for (size_t v = 0; v < nvec; ++v)
        for (int i = 0; i < ctx->n; ++i) {
            states_out[2 * (v * ctx->n + i)] = tmp[v * ctx->np + i].x;
            states_out[2 * (v * ctx->n + i) + 1] = tmp[v * ctx->np + i].y;
        }
    return 0;
}

int qocx_eval_schroedinger(qocx_ctx* ctx, int32_t batch, const double* controls, int32_t want_grad,
                           double* cost_out, double* grad_out, double* final_out) {
    int rc = qocx_upload_controls(ctx, batch, controls);
    if (rc) return rc;
    rc = qocx_eval_resident(ctx, want_grad);
    if (rc) return rc;
    return qocx_download_results(ctx, cost_out, (want_grad && ctx->K > 0) ? grad_out : nullptr,
                                 final_out);
}

int qocx_set_timing(qocx_ctx* ctx, int32_t enable) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (enable < 0 || enable > 8) return fail(QOCX_ERR_ARG, "timing mode must be 0, 1 or 2 + kernel index");
    ctx->timing = enable;
    ctx->time_active = false;
    return 0;
}

int qocx_get_timing(qocx_ctx* ctx, int32_t which, int64_t* launches, double* total_ms) {
    if (!ctx || which < 0 || which > 6) return fail(QOCX_ERR_ARG, "bad argument");
    if (launches) *launches = ctx->t_launch[which];
    if (total_ms) *total_ms = ctx->t_ms[which];
    return 0;
}

int qocx_reset_timing(qocx_ctx* ctx) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    for (int i = 0; i < 7; ++i) {
        ctx->t_launch[i] = 0;
        ctx->t_ms[i] = 0;
    }
    return 0;
}

// ---- Lindblad ----------------------------------------------------------------------------

extern "C++" {
namespace {

typedef std::vector<double> cmat;  // row-major n x n complex, interleaved

cmat cm_zero(int n) { return cmat((size_t)2 * n * n, 0.0); }

cmat cm_from(const double* p, int n) { return cmat(p, p + (size_t)2 * n * n); }

// M = M^H to rounding: max |M - M^H| <= 64 eps max |M|
bool cm_is_hermitian(const cmat& a, int n) {
    double big = 0, diff = 0;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c <= r; ++c) {
            const double xr = a[2 * ((size_t)r * n + c)], xi = a[2 * ((size_t)r * n + c) + 1];
            const double yr = a[2 * ((size_t)c * n + r)], yi = a[2 * ((size_t)c * n + r) + 1];
            big = std::max(big, std::max(fabs(xr), fabs(xi)));
            diff = std::max(diff, std::max(fabs(xr - yr), fabs(xi + yi)));
        }
    return diff <= 1.5e-14 * big;
}

cmat cm_adjoint(const cmat& a, int n) {
    cmat o = cm_zero(n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            o[2 * ((size_t)c * n + r)] = a[2 * ((size_t)r * n + c)];
            o[2 * ((size_t)c * n + r) + 1] = -a[2 * ((size_t)r * n + c) + 1];
        }
    return o;
}

cmat cm_transpose(const cmat& a, int n) {
    cmat o = cm_zero(n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            o[2 * ((size_t)c * n + r)] = a[2 * ((size_t)r * n + c)];
            o[2 * ((size_t)c * n + r) + 1] = a[2 * ((size_t)r * n + c) + 1];
        }
    return o;
}

cmat cm_mul(const cmat& a, const cmat& b, int n) {
    cmat o = cm_zero(n);
    for (int r = 0; r < n; ++r)
        for (int k = 0; k < n; ++k) {
            const double ar = a[2 * ((size_t)r * n + k)], ai = a[2 * ((size_t)r * n + k) + 1];
            for (int c = 0; c < n; ++c) {
                const double br = b[2 * ((size_t)k * n + c)], bi = b[2 * ((size_t)k * n + c) + 1];
                o[2 * ((size_t)r * n + c)] += ar * br - ai * bi;
                o[2 * ((size_t)r * n + c) + 1] += ar * bi + ai * br;
            }
        }
    return o;
}

// o = alpha * a (alpha complex)
cmat cm_scale(const cmat& a, double sr, double si) {
    cmat o(a.size());
    for (size_t e = 0; e < a.size(); e += 2) {
        o[e] = sr * a[e] - si * a[e + 1];
        o[e + 1] = sr * a[e + 1] + si * a[e];
    }
    return o;
}

void cm_axpy(cmat& y, double alpha, const cmat& x) {
    for (size_t e = 0; e < y.size(); ++e) y[e] += alpha * x[e];
}

// C-layout dump of an n x n matrix padded to 16 nb: reg r of tile (ti, tj) of lane l <-> element
// (row 16 ti + 4 r + (l >> 4), col 16 tj + (l & 15)), index ((ti nb + tj) 4 + r) 64 + l
int dump_tiles(int n) { return n <= 16 ? 1 : 2; }
int dump_elems(int n) { return 256 * dump_tiles(n) * dump_tiles(n); }

void c_dump(const cmat& m, int n, double2* out) {
    const int nb = dump_tiles(n);
    for (int ti = 0; ti < nb; ++ti)
        for (int tj = 0; tj < nb; ++tj)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = 16 * ti + 4 * r + (lane >> 4), col = 16 * tj + (lane & 15);
                    double2 e = make_double2(0, 0);
                    if (row < n && col < n) {
                        e.x = m[2 * ((size_t)row * n + col)];
                        e.y = m[2 * ((size_t)row * n + col) + 1];
                    }
                    out[((ti * nb + tj) * 4 + r) * 64 + lane] = e;
                }
}

void from_c_dump(const double2* d, int n, double* out) {
    const int nb = dump_tiles(n);
    for (int ti = 0; ti < nb; ++ti)
        for (int tj = 0; tj < nb; ++tj)
            for (int r = 0; r < 4; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int row = 16 * ti + 4 * r + (lane >> 4), col = 16 * tj + (lane & 15);
                    if (row < n && col < n) {
                        const double2 e = d[((ti * nb + tj) * 4 + r) * 64 + lane];
                        out[2 * ((size_t)row * n + col)] = e.x;
                        out[2 * ((size_t)row * n + col) + 1] = e.y;
                    }
                }
}

double cm_norm_inf(const cmat& m, int n) {
    double best = 0;
    for (int r = 0; r < n; ++r) {
        double s = 0;
        for (int c = 0; c < n; ++c) s += hypot(m[2 * ((size_t)r * n + c)], m[2 * ((size_t)r * n + c) + 1]);
        best = std::max(best, s);
    }
    return best;
}

// Spectral norm of the control-free Liouvillian X -> A_L X + X A_R + sum_i gamma_i L_i X L_i^H as
// an operator on C^(n x n) (Frobenius inner product): matrix-free power iteration on its
// adjoint-times-itself, stopped at 1e-4 relative change, + 2 % (the estimate comes from below).
// The sum of the parts' bounds (2 ||H0||_2 + 2 sum gamma ||L||_2^2) over-estimates it 2-3x when
// the dissipators are stiff in a few levels only (a^H a of a 16-level oscillator), and the
// integrator's sub-division count is proportional to this number.
double liouvillian_norm(const cmat& al, const cmat& ar, const std::vector<cmat>& ops,
                        const std::vector<double>& gammas, int n) {
    const cmat alh = cm_adjoint(al, n), arh = cm_adjoint(ar, n);
    std::vector<cmat> opsh;
    for (const auto& o : ops) opsh.push_back(cm_adjoint(o, n));
    auto apply = [&](const cmat& x, bool adjoint) {
        cmat y = cm_mul(adjoint ? alh : al, x, n);
        cm_axpy(y, 1.0, cm_mul(x, adjoint ? arh : ar, n));
        for (size_t i = 0; i < ops.size(); ++i)
            cm_axpy(y, gammas[i], adjoint ? cm_mul(cm_mul(opsh[i], x, n), ops[i], n)
                                          : cm_mul(cm_mul(ops[i], x, n), opsh[i], n));
        return y;
    };
    auto fro = [](const cmat& x) {
        double s = 0;
        for (double e : x) s += e * e;
        return std::sqrt(s);
    };
    cmat x((size_t)2 * n * n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            x[2 * ((size_t)r * n + c)] = 1.0 / (1.0 + r + c) + (r == c ? 1.0 : 0.0);
            x[2 * ((size_t)r * n + c) + 1] = 0.3 * ((3 * r + c) % 4) - 0.4;
        }
    double nx = fro(x);
    for (auto& e : x) e /= nx;
    double sigma = 0, prev = -1;
    for (int it = 0; it < 400; ++it) {
        const cmat y = apply(x, false);
        sigma = fro(y);
        if (!(sigma > 0) || !(sigma < 1e300)) break;
        if (it >= 8 && std::fabs(sigma - prev) <= 1e-4 * sigma) break;
        prev = sigma;
        x = apply(y, true);
        nx = fro(x);
        if (!(nx > 0)) break;
        for (auto& e : x) e /= nx;
    }
    return 1.02 * sigma;
}

int upload_dumps(DevBuf<double2>& dst, const std::vector<cmat>& mats, int n, hipStream_t st) {
    const size_t md = dump_elems(n);
    std::vector<double2> img(mats.size() * md);
    for (size_t i = 0; i < mats.size(); ++i) c_dump(mats[i], n, img.data() + i * md);
    return dst.upload(img, st);
}

}  // namespace
}  // extern "C++"

int qocx_set_lindblad_problem(qocx_ctx* ctx, const qocx_lindblad_problem* p) {
    if (!ctx || !p) return fail(QOCX_ERR_ARG, "NULL argument");
    if (p->struct_size != (int32_t)sizeof(qocx_lindblad_problem))
        return fail(QOCX_ERR_ARG, "qocx_lindblad_problem.struct_size does not match this "
                                  "library's header (stale binding?)");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = p->hilbert_size, S = p->density_count, K = p->control_count;
    const int N = p->system_eval_count, nc = p->control_eval_count, L = p->operator_count;
    if (n < 1 || n > 32)
        return fail(QOCX_ERR_ARG, "hilbert_size must be in 1..32 for the Lindblad engine");
    if (S < 1 || S > 64) return fail(QOCX_ERR_ARG, "density_count must be in 1..64");
    if (K < 0 || K > QOCX_LINDBLAD_MAX_K) return fail(QOCX_ERR_ARG, "control_count must be in 0..8");
    // (1..4 operators: the several-wave / tile-per-wave stage loops; 5..8: the one-wave kernels, whose stage
    // loop walks any number of operators)
    if (L < 0 || L > 8) return fail(QOCX_ERR_ARG, "operator_count must be in 0..8");
    if (N < 2) return fail(QOCX_ERR_ARG, "system_eval_count must be >= 2");
    if (K > 0 && nc < 2) return fail(QOCX_ERR_ARG, "control_eval_count must be >= 2");
    if (p->cost_eval_step < 1) return fail(QOCX_ERR_ARG, "cost_eval_step must be >= 1");
    if (!p->initial_densities || (K > 0 && !p->g) || (L > 0 && (!p->operators || !p->dissipators)))
        return fail(QOCX_ERR_ARG, "missing problem arrays");
    // densities, cotangents and stage derivatives live in LDS when they fit (n <= 16), else in
    // per-seed HBM scratch
    ctx->lb.global_scratch = (n > 16 || qocx::lindblad_lds_size(n, S, L, 0, K) > 160 * 1024) ? 1 : 0;
    if (qocx::lindblad_lds_size(n, S, L, ctx->lb.global_scratch, K) > 160 * 1024)
        return fail(QOCX_ERR_ARG, "too many operators for the kernel's LDS");
    // several waves per seed (generator terms | one per operator | control cotangents) whenever
    // that layout fits LDS: the recursion in time is serial, this shortens every stage
    // (ONE operator - the T1 problem - runs the several-wave launches as two, the second one zero: the
    // four-wave stage loops of section 14 exist for L = 2 only and are 1.5 times faster than the three-wave
    // form of L = 1 although they multiply by that zero: 17.2 -> 11.5 ms on configs[3]'s sizes)
    ctx->lb.pad_op = (L == 1 && n <= 16 && p->op_stages == nullptr && ctx->knob("lindblad_pad_operator", 1) != 0) ? 1 : 0;
    const int Lmw = ctx->lb.pad_op ? 2 : L;
    ctx->lb.multi_wave = (!ctx->lb.global_scratch && L > 0 && L <= 4 &&
                          qocx::lindblad_lds_size(n, S, Lmw, 2, K) <= 160 * 1024 &&
                          !qocx::diag_getenv("QOCX_LINDBLAD_SINGLE_WAVE")) ? 1 : 0;
    ctx->lb.cache_gen = (ctx->lb.multi_wave && p->fixed_subdivision <= 0 &&
                         qocx::lindblad_lds_size(n, S, Lmw, 3, K) <= 160 * 1024) ? 1 : 0;
    auto& lb = ctx->lb;
    lb.has_problem = false;
    lb.n = n; lb.S = S; lb.K = K; lb.nc = nc; lb.N = N; lb.nsteps = N - 1; lb.ces = p->cost_eval_step;
    lb.nops = L; lb.T = p->evolution_time; lb.dt = p->evolution_time / (N - 1);

    const cmat h0 = p->h0 ? cm_from(p->h0, n) : cm_zero(n);
    cmat decay = cm_zero(n);  // sum gamma_i L_i^H L_i
    std::vector<cmat> ops;
    std::vector<double> gammas(L);
    lb.diss_norm = 0;
    for (int i = 0; i < L; ++i) {
        ops.push_back(cm_from(p->operators + (size_t)i * n * n * 2, n));
        gammas[i] = p->dissipators[i];
        cm_axpy(decay, gammas[i], cm_mul(cm_adjoint(ops[i], n), ops[i], n));
        // || rho -> gamma (L rho L^H - {L^H L, rho} / 2) || <= 2 gamma ||L||_2^2 (the factor 2 is
        // applied where the bound is formed)
        const double opn = two_norm(ops[i].data(), n);
        lb.diss_norm += fabs(gammas[i]) * opn * opn;
    }
    if (lb.pad_op) {
        ops.push_back(cm_zero(n));
        gammas.push_back(0.0);
    }
    lb.ops_real = p->op_stages == nullptr;
    for (const cmat& op : ops)
        for (size_t e = 0; e < (size_t)n * n; ++e)
            if (op[2 * e + 1] != 0.0) lb.ops_real = false;
    // A0L = -i H0 - decay/2 ; A0R = +i H0 - decay/2   (mathmethods.py:188, :200-203)
    cmat a0l = cm_scale(h0, 0.0, -1.0), a0r = cm_scale(h0, 0.0, 1.0);
    cm_axpy(a0l, -0.5, decay);
    cm_axpy(a0r, -0.5, decay);
    lb.hermitian = p->h0_stages == nullptr && p->g_stages == nullptr && p->op_stages == nullptr &&
                   cm_is_hermitian(h0, n) && cm_is_hermitian(decay, n);
    lb.h0_norm = two_norm(h0.data(), n);
    // static problem: the control-free Liouvillian as a whole (never above the sum of the parts)
    lb.l0_norm = std::min(liouvillian_norm(a0l, a0r, ops, gammas, n),
                          2 * lb.h0_norm + 2 * lb.diss_norm);
    if (!(lb.l0_norm < 1e300)) lb.l0_norm = 2 * lb.h0_norm + 2 * lb.diss_norm;
    std::vector<cmat> gp, gpd, gpt;
    lb.g_norm.assign(K, 0.0);
    for (int k = 0; k < K; ++k) {
        const cmat gk = cm_from(p->g + (size_t)k * n * n * 2, n);
        lb.g_norm[k] = two_norm(gk.data(), n);
        lb.hermitian = lb.hermitian && cm_is_hermitian(gk, n);
        gp.push_back(cm_scale(gk, 0.0, -1.0));  // Gp = -i G
        gpd.push_back(cm_adjoint(gp.back(), n));
        gpt.push_back(cm_transpose(gp.back(), n));
    }
    if (upload_dumps(lb.a0l, {a0l}, n, ctx->stream) || upload_dumps(lb.a0r, {a0r}, n, ctx->stream) ||
        upload_dumps(lb.a0ld, {cm_adjoint(a0l, n)}, n, ctx->stream) ||
        upload_dumps(lb.a0rd, {cm_adjoint(a0r, n)}, n, ctx->stream) ||
        upload_dumps(lb.gp, gp, n, ctx->stream) || upload_dumps(lb.gpd, gpd, n, ctx->stream) ||
        upload_dumps(lb.gpt, gpt, n, ctx->stream) || upload_dumps(lb.ops, ops, n, ctx->stream) ||
        lb.gammas.upload(gammas, ctx->stream))
        return QOCX_ERR_HIP;
    // Time-dependent Hamiltonian: samples at the stage times of the fixed sub-division
    // (qocx_lindblad_stage_times), turned into per-stage generator dumps.
    lb.fixed_ksub = 0;
    lb.a0_tab.release();
    lb.gp_tab.release();
    lb.op_tab.release();
    lb.gamma_tab.release();
    const bool td_ops = p->op_stages != nullptr && L > 0;
    if ((p->op_stages != nullptr) != (p->diss_stages != nullptr))
        return fail(QOCX_ERR_ARG, "diss_stages and op_stages go together");
    if (td_ops && p->fixed_subdivision <= 0)
        return fail(QOCX_ERR_ARG, "time-dependent lindblad_data needs fixed_subdivision > 0");
    if (p->fixed_subdivision > 0) {
        if (!p->h0_stages) return fail(QOCX_ERR_ARG, "h0_stages missing");
        int64_t count = 0;
        int rc = qocx_lindblad_stage_times(p->evolution_time, N, nc, K, p->fixed_subdivision, nullptr,
                                           0, &count);
        if (rc) return rc;
        const size_t md = dump_elems(n);
        std::vector<double2> tab((size_t)count * 4 * md);
        lb.h0_norm = 0;
        std::vector<double2> otab(td_ops ? (size_t)count * L * md : 0);
        std::vector<double> gtab_d(td_ops ? (size_t)count * L : 0);
        if (td_ops) lb.diss_norm = 0;
        for (int64_t st = 0; st < count; ++st) {
            const cmat h = cm_from(p->h0_stages + (size_t)st * n * n * 2, n);
            // (norms on every fourth stage sample: they vary smoothly in time, the host picked
            // the sub-division with a 25 % margin, and a power iteration per sample is what made
            // this loop slow)
            const bool norm_sample = (st % 4 == 0) || st == count - 1;
            if (norm_sample) lb.h0_norm = std::max(lb.h0_norm, two_norm(h.data(), n));
            cmat l = cm_scale(h, 0.0, -1.0), r = cm_scale(h, 0.0, 1.0);
            cmat decay_st = decay;
            if (td_ops) {  // -1/2 sum_i gamma_i(t) L_i(t)^H L_i(t) of THIS stage time
                decay_st = cm_zero(n);
                double dn = 0;
                for (int i = 0; i < L; ++i) {
                    const cmat li = cm_from(p->op_stages + ((size_t)st * L + i) * n * n * 2, n);
                    const double gi = p->diss_stages[(size_t)st * L + i];
                    cm_axpy(decay_st, gi, cm_mul(cm_adjoint(li, n), li, n));
                    c_dump(li, n, otab.data() + ((size_t)st * L + i) * md);
                    gtab_d[(size_t)st * L + i] = gi;
                    if (norm_sample) {
                        const double opn = two_norm(li.data(), n);
                        dn += fabs(gi) * opn * opn;
                    }
                }
                lb.diss_norm = std::max(lb.diss_norm, dn);
            }
            cm_axpy(l, -0.5, decay_st);
            cm_axpy(r, -0.5, decay_st);
            c_dump(l, n, tab.data() + ((size_t)st * 4 + 0) * md);
            c_dump(r, n, tab.data() + ((size_t)st * 4 + 1) * md);
            c_dump(cm_adjoint(l, n), n, tab.data() + ((size_t)st * 4 + 2) * md);
            c_dump(cm_adjoint(r, n), n, tab.data() + ((size_t)st * 4 + 3) * md);
        }
        if (lb.a0_tab.upload(tab, ctx->stream)) return QOCX_ERR_HIP;
        if (td_ops && (lb.op_tab.upload(otab, ctx->stream) || lb.gamma_tab.upload(gtab_d, ctx->stream)))
            return QOCX_ERR_HIP;
        if (p->g_stages && K > 0) {
            std::vector<double2> gtab((size_t)count * K * 3 * md);
            lb.g_norm.assign(K, 0.0);
            for (int64_t st = 0; st < count; ++st)
                for (int k = 0; k < K; ++k) {
                    const cmat gk = cm_from(p->g_stages + ((size_t)st * K + k) * n * n * 2, n);
                    if (st % 4 == 0 || st == count - 1)
                        lb.g_norm[k] = std::max(lb.g_norm[k], two_norm(gk.data(), n));
                    const cmat gpk = cm_scale(gk, 0.0, -1.0);
                    double2* dst = gtab.data() + (((size_t)st * K + k) * 3) * md;
                    c_dump(gpk, n, dst);
                    c_dump(cm_adjoint(gpk, n), n, dst + md);
                    c_dump(cm_transpose(gpk, n), n, dst + 2 * md);
                }
            if (lb.gp_tab.upload(gtab, ctx->stream)) return QOCX_ERR_HIP;
        }
        lb.fixed_ksub = p->fixed_subdivision;
    }
    std::vector<cmat> rho0;
    for (int s = 0; s < S; ++s) rho0.push_back(cm_from(p->initial_densities + (size_t)s * n * n * 2, n));
    if (upload_dumps(lb.rho0, rho0, n, ctx->stream)) return QOCX_ERR_HIP;
    for (const cmat& r : rho0) lb.hermitian = lb.hermitian && cm_is_hermitian(r, n);

    std::vector<qocx::DevCost> dcosts;
    std::vector<cmat> pool;
    std::vector<int> counts;
    lb.has_step_costs = 0;
    for (int ci = 0; ci < p->cost_count; ++ci) {
        const qocx_cost_desc& c = p->costs[ci];
        qocx::DevCost d;
        d.step_cost = c.step_cost ? 1 : 0;
        d.scale = c.scale;
        d.vec_offset = (int)pool.size();
        d.cnt_offset = (int)counts.size();
        if (!c.vectors) return fail(QOCX_ERR_ARG, "cost matrices missing");
        int nmat = S;
        if (c.kind == QOCX_COST_TARGET_DENSITY) {
            d.kind = QOCX_DEV_COST_TARGET_DENSITY;
        } else if (c.kind == QOCX_COST_FORBID_DENSITY) {
            d.kind = QOCX_DEV_COST_FORBID_DENSITY;
            if (!c.counts) return fail(QOCX_ERR_ARG, "forbid counts missing");
            nmat = 0;
            for (int s = 0; s < S; ++s) {
                if (c.counts[s] < 1) return fail(QOCX_ERR_ARG, "forbid count < 1");
                counts.push_back(c.counts[s]);
                nmat += c.counts[s];
            }
        } else {
            return fail(QOCX_ERR_ARG, "cost kind not valid for the Lindblad path");
        }
        for (int m = 0; m < nmat; ++m) pool.push_back(cm_from(c.vectors + (size_t)m * n * n * 2, n));
        if (d.step_cost) lb.has_step_costs = 1;
        dcosts.push_back(d);
    }
    for (const cmat& m : pool) lb.hermitian = lb.hermitian && cm_is_hermitian(m, n);
    lb.cost_count = (int)dcosts.size();
    lb.unit_ok = dcosts.size() == 1 && !dcosts[0].step_cost &&
                 dcosts[0].kind == QOCX_DEV_COST_TARGET_DENSITY;
    if (lb.costs.upload(dcosts, ctx->stream) || upload_dumps(lb.cost_matrices, pool, n, ctx->stream) ||
        lb.cost_counts.upload(counts, ctx->stream))
        return QOCX_ERR_HIP;
    for (auto& kv : lb.grids) {
        kv.second.substeps.release();
        kv.second.row_ptr.release();
        kv.second.col.release();
        kv.second.weight.release();
    }
    lb.grids.clear();
    lb.has_problem = true;
    lb.have_results = false;
    lb.inj_count = 0;
    return 0;
}

extern "C++" {
namespace {

// End points of the sub-intervals of system step `step`: `ksub` uniform pieces, cut at the
// control knots that fall inside the step.
std::vector<double> lindblad_points(double T, int nsteps, int nc, int K, int ksub, int step) {
    const double dt = T / nsteps;
    const double t0 = step * dt, t1 = (step + 1) * dt;
    std::vector<double> pts;
    for (int q = 0; q < ksub; ++q) pts.push_back(t0 + (t1 - t0) * q / ksub);
    pts.push_back(t1);
    if (K > 0)
        for (int i = 0; i < nc; ++i) {
            const double kn = (i == nc - 1) ? T : i * (T / (nc - 1));
            if (kn > t0 + 1e-12 * dt && kn < t1 - 1e-12 * dt) pts.push_back(kn);
        }
    std::sort(pts.begin(), pts.end());
    pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
    return pts;
}
}  // extern "C++"

// Sub-interval table of one sub-division count: uniform pieces per system step, cut at control
// knots, with the interpolation weights of both ends and the CSR of their transpose.
int build_lindblad_grid(qocx_ctx* ctx, int ksub, qocx_ctx::Lindblad::Grid& gr) {
    auto& lb = ctx->lb;
    const int K = lb.K, nc = lb.nc, nsteps = lb.nsteps;
    std::vector<double> knots(K > 0 ? nc : 0);
    for (int i = 0; i < (int)knots.size(); ++i) knots[i] = i * (lb.T / (nc - 1));
    if (!knots.empty()) knots.back() = lb.T;
    std::vector<qocx::SubStep> subs;
    for (int step = 0; step < nsteps; ++step) {
        const std::vector<double> pts = lindblad_points(lb.T, nsteps, nc, K, ksub, step);
        for (size_t i = 0; i + 1 < pts.size(); ++i) {
            qocx::SubStep ss;
            ss.h = pts[i + 1] - pts[i];
            // both ends interpolate on the knot interval that contains the sub-interval
            int m1 = 0, m2 = 0;
            if (!knots.empty()) {
                const double mid = 0.5 * (pts[i] + pts[i + 1]);
                if (mid <= knots[0]) { m1 = 0; m2 = 1; }
                else if (mid >= knots[nc - 1]) { m1 = nc - 2; m2 = nc - 1; }
                else {
                    int idx = 0;
                    while (!(mid <= knots[idx])) ++idx;
                    m1 = idx - 1; m2 = idx;
                }
            }
            auto end_weights = [&](double x, double& w1, double& w2) {
                if (knots.empty()) { w1 = 1; w2 = 0; return; }
                const double theta = (x - knots[m1]) / (knots[m2] - knots[m1]);
                w1 = 1.0 - theta; w2 = theta;
            };
            ss.ia1 = ss.ib1 = m1; ss.ia2 = ss.ib2 = m2;
            end_weights(pts[i], ss.wa1, ss.wa2);
            end_weights(pts[i + 1], ss.wb1, ss.wb2);
            ss.step = step;
            ss.first_of_step = (i == 0) ? 1 : 0;
            subs.push_back(ss);
        }
    }
    const int nsub = (int)subs.size();
    std::vector<std::vector<std::pair<int, double>>> rows(K > 0 ? nc : 0);
    if (K > 0)
        for (int q = 0; q < nsub; ++q) {
            rows[subs[q].ia1].push_back({2 * q, subs[q].wa1});
            rows[subs[q].ia2].push_back({2 * q, subs[q].wa2});
            rows[subs[q].ib1].push_back({2 * q + 1, subs[q].wb1});
            rows[subs[q].ib2].push_back({2 * q + 1, subs[q].wb2});
        }
    std::vector<int> row_ptr(1, 0), col;
    std::vector<double> weight;
    for (auto& r : rows) {
        for (auto& e : r) { col.push_back(e.first); weight.push_back(e.second); }
        row_ptr.push_back((int)col.size());
    }
    if (gr.substeps.upload(subs, ctx->stream) || gr.row_ptr.upload(row_ptr, ctx->stream) ||
        gr.col.upload(col, ctx->stream) || gr.weight.upload(weight, ctx->stream))
        return QOCX_ERR_HIP;
    gr.nsub = nsub;
    return 0;
}

}  // namespace

int qocx_lindblad_stage_times(double evolution_time, int32_t system_eval_count,
                              int32_t control_eval_count, int32_t control_count,
                              int32_t subdivision, double* times_out, int64_t capacity,
                              int64_t* count_out) {
    if (system_eval_count < 2 || subdivision < 1 || !count_out ||
        (control_count > 0 && control_eval_count < 2))
        return fail(QOCX_ERR_ARG, "bad argument");
    const int nsteps = system_eval_count - 1;
    int64_t count = 0;
    for (int step = 0; step < nsteps; ++step) {
        const std::vector<double> pts = lindblad_points(evolution_time, nsteps, control_eval_count,
                                                        control_count, subdivision, step);
        for (size_t i = 0; i + 1 < pts.size(); ++i)
            for (int st = 0; st < QOCX_RK_STAGES; ++st) {
                if (times_out && count < capacity)
                    times_out[count] = pts[i] + QOCX_RK_C[st] * (pts[i + 1] - pts[i]);
                ++count;
            }
    }
    *count_out = count;
    return 0;
}

int qocx_eval_lindblad(qocx_ctx* ctx, int32_t batch, const double* controls, int32_t want_grad,
                       double* cost_out, double* grad_out, double* final_out) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    auto& lb = ctx->lb;
    if (!lb.has_problem) return fail(QOCX_ERR_STATE, "no Lindblad problem set");
    if (batch < 1) return fail(QOCX_ERR_ARG, "batch must be >= 1");
    HIP_TRY(hipSetDevice(ctx->device));
    const int n = lb.n, S = lb.S, K = lb.K, nc = lb.nc, nsteps = lb.nsteps, B = batch;
    const size_t md = dump_elems(n);
    want_grad = (want_grad && K > 0) ? 1 : 0;
    if (K > 0 && !controls) return fail(QOCX_ERR_ARG, "controls is NULL");
    const bool trace_host = qocx::diag_getenv("QOCX_TRACE_HOST") != nullptr;
    auto now_ms = [] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    };
    const double t_enter = now_ms();
    double t_alloc = 0, t_enq = 0, t_sync = 0;

    // Each seed picks its own sub-division count from ITS controls (|| Liouvillian || * length
    // <= 0.4 per sub-interval), so a seed's result never depends on its batch neighbours. Seeds
    // with equal counts are evaluated together: `order` lists the seeds group by group.
    std::vector<int> ksub_of(B);
    for (int b = 0; b < B; ++b) {
        // || Liouvillian ||_2 <= || control-free part ||_2 + sum_k |u_k| 2 ||G_k||_2; with a
        // time-dependent Hamiltonian / lindblad_data (tables) the control-free part is bounded by
        // the sum of its parts' bounds over the samples
        double ctl = 0;
        for (int k = 0; k < K; ++k) {
            double um = 0;
            for (int i = 0; i < nc; ++i) {
                const double a = fabs(controls[((size_t)b * nc + i) * K + k]);
                if (!(a <= um)) um = a;
            }
            ctl += um * lb.g_norm[k];
        }
        const double base = lb.fixed_ksub > 0 ? 2 * lb.h0_norm + 2 * lb.diss_norm : lb.l0_norm;
        const double bound = base + 2 * ctl;
        if (!(bound < 1e300)) return fail(QOCX_ERR_ARG, "non-finite controls or operators");
        const double pieces = ceil(bound * fabs(lb.dt) / 0.4);
        if (pieces * nsteps > (double)(1 << 24))
            return fail(QOCX_ERR_CAPACITY, "too many sub-intervals");
        ksub_of[b] = std::max(1, (int)pieces);
        if (lb.fixed_ksub > 0) {
            // the time samples of the Hamiltonian exist for one grid only
            if (ksub_of[b] > lb.fixed_ksub)
                return fail(QOCX_ERR_CAPACITY,
                            "controls need a finer sub-division than the Hamiltonian was sampled for");
            ksub_of[b] = lb.fixed_ksub;
        }
    }
    std::map<int, std::vector<int>> groups;
    for (int b = 0; b < B; ++b) groups[ksub_of[b]].push_back(b);
    if (lb.grids.size() > 64) {  // bounded cache of sub-interval tables
        for (auto& kv : lb.grids) {
            kv.second.substeps.release(); kv.second.row_ptr.release();
            kv.second.col.release(); kv.second.weight.release();
        }
        lb.grids.clear();
    }
    size_t ckpt_total = 0, gsub_total = 0;
    lb.order.clear();
    lb.last_subintervals = 0;
    for (auto& kv : groups) {
        auto it = lb.grids.find(kv.first);
        if (it == lb.grids.end()) {
            int rc = build_lindblad_grid(ctx, kv.first, lb.grids[kv.first]);
            if (rc) return rc;
            it = lb.grids.find(kv.first);
        }
        ckpt_total += kv.second.size() * (size_t)it->second.nsub * S * md;
        gsub_total += kv.second.size() * (size_t)it->second.nsub * 2 * std::max(K, 1);
        lb.last_subintervals += (int64_t)kv.second.size() * it->second.nsub;
        for (int b : kv.second) lb.order.push_back(b);
    }
    // The stage values of the forward pass are kept for the adjoint (12 x the checkpoints of the
    // seeds in flight); a group of seeds that does not fit is launched in pieces that do, and
    // only if a piece would fall below 256 seeds does the adjoint recompute the stages instead.
    // Two-sided evaluation (LindbladArgs::phase): where it applies the adjoint's stage cotangents
    // need a buffer like the forward's stage values, and gsub holds complex numbers
    // (n > 16: the tile-per-wave kernel of qocx_lindblad4t.hip in its phases, constant tables only)
    const bool two_sided_small = n <= 16 && lb.nops >= 1 && lb.multi_wave && !lb.global_scratch &&
                                 lb.dbg_wave_mode != 1;
    const bool two_sided_tiles = n > 16 && ctx->knob("lindblad_4t", 1) != 0 && lb.nops <= 4;
    const bool two_sided_ok = want_grad && lb.unit_ok && lb.inj_count == 0 &&
                              (two_sided_small || two_sided_tiles) && lb.fixed_ksub == 0 &&
                              (int)ctx->sweep_streams.size() >= 1 &&
                              ctx->knob("lindblad_two_sided", 1) != 0;
    if (two_sided_ok)
        if (lb.lam_scale.ensure((size_t)B * S)) return QOCX_ERR_HIP;
    size_t stage_budget = 0;  // double2 elements
    if (want_grad) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        stage_budget = (size_t)(0.45 * (double)(free_b + (lb.ystages.count + lb.kbstages.count) *
                                                              sizeof(double2))) /
                       sizeof(double2);
        if (two_sided_ok) stage_budget /= 2;  // kbar_i beside Y_i
        size_t want = 0;
        for (auto& kv : groups) {
            const size_t per_seed = (size_t)lb.grids[kv.first].nsub * S * md * 12;
            const size_t fit = lb.dbg_stage_seeds > 0 ? (size_t)lb.dbg_stage_seeds
                                                      : std::max<size_t>(1, stage_budget / per_seed);
            const size_t piece = std::min<size_t>(kv.second.size(), fit);
            if (piece == kv.second.size() || piece >= (size_t)lb.dbg_min_piece)
                want = std::max(want, piece * per_seed);
        }
        if (want > 0 && lb.ystages.ensure(want)) return QOCX_ERR_HIP;
        if (want > 0 && two_sided_ok && lb.kbstages.ensure(want)) return QOCX_ERR_HIP;
    }
    if (lb.global_scratch &&
        lb.scratch.ensure((size_t)B * qocx::lindblad_scratch_elems(n, S)))
        return QOCX_ERR_HIP;
    const size_t csz = (size_t)nc * K;
    if (lb.controls.ensure((size_t)B * std::max<size_t>(csz, 1)) || lb.cost_out.ensure(B) ||
        lb.grads.ensure((size_t)B * std::max<size_t>(csz, 1)) || lb.gsub.ensure(gsub_total) ||
        lb.checkpoints.ensure(ckpt_total) || lb.final_out.ensure((size_t)B * S * md))
        return QOCX_ERR_HIP;
    if (ctx->keep_step_states)
        if (lb.step_densities.ensure((size_t)B * (nsteps + 1) * S * md)) return QOCX_ERR_HIP;
    if (K > 0) {
        // gathered group by group into the pinned staging buffer (a pageable source of 2 MB costs
        // the copy 10-25 ms of page pinning per call at 256 seeds; from pinned memory it is a DMA)
        const size_t total = (size_t)B * csz;
        if (ctx->pin_controls_cap < total) {
            if (ctx->pin_controls) (void)hipHostFree(ctx->pin_controls);
            ctx->pin_controls = nullptr;
            ctx->pin_controls_cap = 0;
            HIP_TRY(hipHostMalloc((void**)&ctx->pin_controls, total * sizeof(double), hipHostMallocDefault));
            ctx->pin_controls_cap = total;
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // nothing in flight still reads the staging buffer
        for (int pos = 0; pos < B; ++pos)
            memcpy(ctx->pin_controls + (size_t)pos * csz, controls + (size_t)lb.order[pos] * csz,
                   csz * sizeof(double));
        HIP_TRY(hipMemcpyAsync(lb.controls.p, ctx->pin_controls, total * sizeof(double),
                               hipMemcpyHostToDevice, ctx->stream));
    }
    if (lb.inj_count > 0) {
        if (lb.inj_batch != B)
            return fail(QOCX_ERR_STATE, "density cotangents were set for a different batch size");
        std::vector<int> index(nsteps + 1, -1);
        for (int c = 0; c < lb.inj_count; ++c) index[lb.inj_steps[c]] = c;
        const size_t per_seed = (size_t)lb.inj_count * S;
        std::vector<double2> dumps((size_t)B * per_seed * md);
        for (int pos = 0; pos < B; ++pos)
            for (size_t v = 0; v < per_seed; ++v) {
                cmat m(lb.inj_host.begin() + (((size_t)lb.order[pos] * per_seed + v) * n * n * 2),
                       lb.inj_host.begin() + (((size_t)lb.order[pos] * per_seed + v + 1) * n * n * 2));
                c_dump(m, n, dumps.data() + ((size_t)pos * per_seed + v) * md);
            }
        if (lb.inj_index.upload(index, ctx->stream) || lb.inj_bars.upload(dumps, ctx->stream))
            return QOCX_ERR_HIP;
    }
    t_alloc = now_ms();
    size_t pos0 = 0, ckpt_off = 0, gsub_off = 0;
    for (auto& kv : groups) {
        const auto& gr = lb.grids[kv.first];
        const int Bg = (int)kv.second.size(), nsub = gr.nsub;
        const size_t per_seed_stage = (size_t)nsub * S * md * 12;
        int piece = Bg;
        bool keep_stages = false;
        if (want_grad) {
            const size_t fit = lb.dbg_stage_seeds > 0
                                   ? (size_t)lb.dbg_stage_seeds
                                   : std::max<size_t>(1, stage_budget / per_seed_stage);
            if (fit >= (size_t)Bg) { keep_stages = true; }
            else if (fit >= (size_t)lb.dbg_min_piece) { keep_stages = true; piece = (int)fit; }
        }
        // Several waves per seed (one seed per CU) whenever the kernel is built for this problem;
        // batches beyond the CU count go in rounds of one seed per CU. (Round 1 switched to one wave
        // per seed, two seeds per CU, beyond 256 seeds; measured on configs[3] at 300 / 512 / 768 /
        // 1024 seeds: 73.7 / 80.7 / 126.6 / 123.7 ms against 74.4 / 76.6 / 82.7 / 104.8 ms in rounds.)
        bool multi = lb.multi_wave != 0;
        if (lb.dbg_wave_mode == 1) multi = false;
        if (multi && lb.dbg_wave_mode != 2) piece = std::min(piece, ctx->cu_count);
        for (int p0 = 0; p0 < Bg; p0 += piece) {
            const int Bp = std::min(piece, Bg - p0);
            qocx::LindbladArgs la;
            la.controls = lb.controls.p + pos0 * csz; la.substeps = gr.substeps.p;
            la.a0l_cimg = lb.a0l.p; la.a0r_cimg = lb.a0r.p; la.a0ld_cimg = lb.a0ld.p; la.a0rd_cimg = lb.a0rd.p;
            la.gp_cimg = lb.gp.p; la.gpd_cimg = lb.gpd.p; la.gpt_cimg = lb.gpt.p; la.op_cimg = lb.ops.p;
            la.gammas = lb.gammas.p; la.rho0_cimg = lb.rho0.p;
            la.a0_tab = lb.fixed_ksub > 0 ? lb.a0_tab.p : nullptr;
            la.gp_tab = (lb.fixed_ksub > 0 && lb.gp_tab.p) ? lb.gp_tab.p : nullptr;
            la.op_tab = (lb.fixed_ksub > 0 && lb.op_tab.p) ? lb.op_tab.p : nullptr;
            la.gamma_tab = la.op_tab ? lb.gamma_tab.p : nullptr;
            la.n = n; la.S = S; la.K = K; la.nc = nc; la.nops = (multi && lb.pad_op) ? 2 : lb.nops; la.nsub = nsub;
            la.nsteps = nsteps;
            la.cost_eval_step = lb.ces; la.want_grad = want_grad; la.has_step_costs = lb.has_step_costs;
            la.cost_count = lb.cost_count; la.costs = lb.costs.p; la.cost_matrices = lb.cost_matrices.p;
            la.cost_counts = lb.cost_counts.p;
            la.checkpoints = lb.checkpoints.p + ckpt_off; la.gsub = lb.gsub.p + gsub_off;
            la.ystages = keep_stages ? lb.ystages.p : nullptr;  // reused piece after piece
            la.scratch = lb.global_scratch ? lb.scratch.p : nullptr;  // likewise
            // several waves per seed shorten a seed's serial chain by ~1.4x but hold one seed
            // per CU instead of two: worth it while the batch leaves CUs idle
            la.multi_wave = multi ? 1 : 0;
            la.cache_gen = (la.multi_wave && lb.cache_gen) ? 1 : 0;
            la.cost_out = lb.cost_out.p + pos0;
            la.final_out = lb.final_out.p + pos0 * S * md;
            la.step_densities = ctx->keep_step_states
                                    ? lb.step_densities.p + pos0 * (nsteps + 1) * S * md : nullptr;
            la.tile4 = ctx->knob("lindblad_4t", 1) != 0 ? 1 : 0;
            la.hermitian = (lb.hermitian && lb.inj_count == 0 && ctx->knob("lindblad_hermitian", 1) != 0) ? 1 : 0;
            la.stamps = nullptr;
            if (ctx->knob("lindblad_stamps", 0)) {
                // ([B] sets of the forward pass / classic launch, then [B] of the unit adjoint)
                if (ctx->stamps.ensure((size_t)B * 96)) return QOCX_ERR_HIP;
                HIP_TRY(hipMemsetAsync(ctx->stamps.p, 0, (size_t)B * 96 * sizeof(unsigned long long),
                                       ctx->stream));
                la.stamps = ctx->stamps.p + pos0 * 48;
            }
            la.inj_count = lb.inj_count;
            la.inj_index = lb.inj_count > 0 ? lb.inj_index.p : nullptr;
            la.inj_bars = lb.inj_count > 0 ? lb.inj_bars.p + pos0 * lb.inj_count * S * md : nullptr;
            // Two-sided: forward pass and unit adjoint as two launches, then the combine kernel on
            // the whole chip. While both launches find CUs of their own they run on two streams
            // (2 Bp CUs busy instead of Bp); a bigger piece runs them one after the other - the
            // same three kernels, so a seed's result does not depend on the batch it is part of.
            bool two_sided = two_sided_ok && keep_stages && (multi || two_sided_tiles);
            if (two_sided && n > 16) {
                // above one tile only the tile-per-wave kernel knows the phases: ask IT whether it
                // takes these launches (the one-wave form would run the whole evaluation twice)
                qocx::LindbladArgs probe = la;
                probe.phase = 1;
                probe.kbstages = lb.kbstages.p;
                if (!qocx::lindblad4t_supports(probe)) two_sided = false;
            }
            if (two_sided) {
                const int side_limit = (int)ctx->knob("lindblad_side_limit", ctx->cu_count / 2);
                hipStream_t side = Bp <= side_limit ? ctx->sweep_streams[0] : ctx->stream;
                la.kbstages = lb.kbstages.p;
                la.lam_scale = lb.lam_scale.p + pos0 * S;
                // everything enqueued so far (uploads, earlier pieces that reuse the stage buffers)
                if (side != ctx->stream) {
                    HIP_TRY(hipEventRecord(ctx->ev_factored[0], ctx->stream));
                    HIP_TRY(hipStreamWaitEvent(side, ctx->ev_factored[0], 0));
                }
                qocx::LindbladArgs fwd = la, adj = la;
                fwd.phase = 1;
                adj.phase = 2;
                fwd.q2 = adj.q2 = ctx->knob("lindblad_q2", 1) != 0 ? 1 : 0;
                fwd.chain = adj.chain = ctx->knob("lindblad_chain", 1) != 0 ? 1 : 0;
                fwd.ops_real = adj.ops_real = (lb.ops_real && ctx->knob("lindblad_real_ops", 1) != 0) ? 1 : 0;
                if (la.stamps != nullptr) adj.stamps = la.stamps + (size_t)B * 48;
                time_begin(ctx, 5, ctx->stream);
                qocx::launch_lindblad(fwd, Bp, ctx->stream);
                time_end(ctx, ctx->stream);
                time_begin(ctx, 5, side);
                qocx::launch_lindblad(adj, Bp, side);
                time_end(ctx, side);
                if (side != ctx->stream) {
                    HIP_TRY(hipEventRecord(ctx->ev_swept[0], side));
                    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_swept[0], 0));
                }
                time_begin(ctx, 6, ctx->stream);
                qocx::launch_lindblad_combine(la, Bp, ctx->stream);
                time_end(ctx, ctx->stream);
            } else {
                time_begin(ctx, 5, ctx->stream);
                qocx::launch_lindblad(la, Bp, ctx->stream);
                time_end(ctx, ctx->stream);
            }
            if (want_grad) {
                qocx::ScatterArgs sc;
                sc.gstep = la.gsub; sc.row_ptr = gr.row_ptr.p; sc.col_step = gr.col.p;
                sc.weight = gr.weight.p; sc.grads = lb.grads.p + pos0 * csz;
                sc.B = Bp; sc.nc = nc; sc.K = K; sc.nsteps = 2 * nsub;

                time_begin(ctx, 3, ctx->stream);
                qocx::launch_scatter(sc, ctx->stream);
                time_end(ctx, ctx->stream);
            }
            pos0 += Bp;
            ckpt_off += (size_t)Bp * nsub * S * md;
            gsub_off += (size_t)Bp * nsub * 2 * std::max(K, 1);
        }
    }
    HIP_TRY(hipGetLastError());
    t_enq = now_ms();
    std::vector<double2> fin(final_out ? (size_t)B * S * md : 0);
    std::vector<double> cst(B), grd(want_grad && grad_out ? (size_t)B * csz : 0);
    HIP_TRY(hipMemcpyAsync(cst.data(), lb.cost_out.p, (size_t)B * sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    if (!grd.empty())
        HIP_TRY(hipMemcpyAsync(grd.data(), lb.grads.p, grd.size() * sizeof(double),
                               hipMemcpyDeviceToHost, ctx->stream));
    if (final_out)
        HIP_TRY(hipMemcpyAsync(fin.data(), lb.final_out.p, fin.size() * sizeof(double2),
                               hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    t_sync = now_ms();
    time_collect(ctx);
    if (trace_host)
        fprintf(stderr, "qocx_eval_lindblad B=%d: set-up %.2f ms, enqueue %.2f ms, wait %.2f ms\n", B,
                t_alloc - t_enter, t_enq - t_alloc, t_sync - t_enq);
    for (int pos = 0; pos < B; ++pos) {
        const int b = lb.order[pos];
        if (cost_out) cost_out[b] = cst[pos];
        if (!grd.empty())
            memcpy(grad_out + (size_t)b * csz, grd.data() + (size_t)pos * csz, csz * sizeof(double));
        if (final_out)
            for (int s = 0; s < S; ++s)
                from_c_dump(fin.data() + ((size_t)pos * S + s) * md, n,
                            final_out + ((size_t)b * S + s) * n * n * 2);
    }
    lb.B = B;
    lb.have_results = true;
    lb.have_steps = ctx->keep_step_states != 0;
    return 0;
}

// Knobs: the variant switches every build accepts (each of them selects between paths that give
// the same numbers to rounding) and the diagnostic ones that exist in libqocx_diag.so only
// (qocx_diag.h): timing experiments that return garbage and the stamped kernel builds.
static const char* const kVariantKnobs[] = {
    "sweep_loader", "sweep_impl", "sweep3_phases", "magnus_general", "bidir", "bidir_min_segments", "bidir_adj_first", "unit_adjoint",
    "sweep_onebuf", "sweep_one", "latency", "fuse_lu", "lindblad_two_sided", "lindblad_side_limit", "k3_split", "k3_split_outer",
    "m4_linear", "pade_order", "lu_inverse", "sweep_dense", "krylov_dense", "magnus_4w",
    "sweep_inverse", "sweep_inverse_small", "lu_mfma", "lu_dpp", "k1a_three", "pack8", "sweep_umode", "general_split", "general_skew", "k1a_four", "k1a_share", "k1a_streams", "sweep_tail_ring", "k3_lds_pad", "step_table", "sweep_nine", "lindblad_q2", "lindblad_chain", "lindblad_real_ops", "k1a_herm4", "lu_stream", "lindblad_4t", "lindblad_hermitian", "lindblad_pad_operator"};
static const char* const kDiagKnobs[] = {"dbg_skip", "sweep3_dbg", "sweep3_stamps", "lindblad_stamps",
                                         "k1a_stamps", "k1a_dbg", "peak_mode"};

int qocx_knob_kind(const char* name) {
    if (!name) return 0;
    for (const char* k : kVariantKnobs)
        if (strcmp(k, name) == 0) return 1;
    for (const char* k : kDiagKnobs)
        if (strcmp(k, name) == 0) return qocx::kDiagBuild ? 2 : -2;
    return 0;
}

int qocx_build_is_diag(void) { return qocx::kDiagBuild ? 1 : 0; }

int qocx_debug_set_knob(qocx_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return fail(QOCX_ERR_ARG, "NULL argument");
    const int kind = qocx_knob_kind(name);
    if (kind > 0) {
        ctx->knobs[name] = value;
        return 0;
    }
    if (kind == -2)
        return fail(QOCX_ERR_ARG, std::string("diagnostic knob '") + name +
                                      "' exists in libqocx_diag.so only (make diag, -DQOCX_DIAG)");
    return fail(QOCX_ERR_ARG, std::string("unknown knob: ") + name);
}

int qocx_lu_fallbacks(qocx_ctx* ctx, int64_t* count) {
    if (!ctx || !count) return fail(QOCX_ERR_ARG, "NULL argument");
    *count = 0;
    if (ctx->lu_fallbacks.p == nullptr) return 0;
    HIP_TRY(hipSetDevice(ctx->device));
    int v = 0;
    HIP_TRY(hipMemcpy(&v, ctx->lu_fallbacks.p, sizeof(int), hipMemcpyDeviceToHost));
    *count = v;
    return 0;
}

int qocx_lindblad_last_subintervals(qocx_ctx* ctx, int64_t* total) {
    if (!ctx || !total) return fail(QOCX_ERR_ARG, "NULL argument");
    *total = ctx->lb.last_subintervals;
    return 0;
}

int qocx_pade_orders(qocx_ctx* ctx, int64_t* counts) {
    if (!ctx || !counts) return fail(QOCX_ERR_ARG, "NULL argument");
    if (!ctx->have_results) return fail(QOCX_ERR_STATE, "no evaluation results");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t total = (size_t)ctx->last_chunk * ctx->nsteps;
    if (total == 0 || total > ctx->s_arr.count) return fail(QOCX_ERR_STATE, "no step table");
    std::vector<int> entries(total);
    HIP_TRY(hipMemcpy(entries.data(), ctx->s_arr.p, total * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < 5; ++i) counts[i] = 0;
    for (int e : entries) {
        const int o = (e >> 8) & 0xff;  // step_entry (qocx_wave.h): 0 means 13
        counts[o == 3 ? 0 : (o == 5 ? 1 : (o == 7 ? 2 : (o == 9 ? 3 : 4)))] += 1;
    }
    return 0;
}

int qocx_debug_timeline(qocx_ctx* ctx, double* out, int64_t capacity, int64_t* count) {
    if (!ctx || !count) return fail(QOCX_ERR_ARG, "NULL argument");
    const int64_t n = (int64_t)(ctx->timeline.size() / 3);
    *count = n;
    if (out)
        for (int64_t i = 0; i < std::min(n, capacity) * 3; ++i) out[i] = ctx->timeline[(size_t)i];
    return 0;
}

int qocx_debug_read_stamps(qocx_ctx* ctx, uint64_t* out, int64_t count) {
    if (!ctx || !out) return fail(QOCX_ERR_ARG, "NULL argument");
    if ((size_t)count > ctx->stamps.count) return fail(QOCX_ERR_ARG, "more stamps than were collected");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(out, ctx->stamps.p, (size_t)count * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

int qocx_debug_lindblad_knobs(qocx_ctx* ctx, int64_t stage_budget_seeds, int32_t min_piece,
                              int32_t wave_mode) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (stage_budget_seeds < 0 || min_piece < 1 || wave_mode < 0 || wave_mode > 2)
        return fail(QOCX_ERR_ARG, "bad knob value");
    ctx->lb.dbg_stage_seeds = stage_budget_seeds;
    ctx->lb.dbg_min_piece = min_piece;
    ctx->lb.dbg_wave_mode = wave_mode;
    return 0;
}

int qocx_set_density_cotangents(qocx_ctx* ctx, int32_t batch, int32_t count, const int32_t* steps,
                                const double* bars) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    auto& lb = ctx->lb;
    if (!lb.has_problem) return fail(QOCX_ERR_STATE, "no Lindblad problem set");
    if (count <= 0) {
        lb.inj_count = 0;
        return 0;
    }
    if (batch < 1 || !steps || !bars) return fail(QOCX_ERR_ARG, "bad argument");
    std::vector<bool> seen(lb.nsteps + 1, false);
    for (int c = 0; c < count; ++c) {
        if (steps[c] < 1 || steps[c] > lb.nsteps || seen[steps[c]])
            return fail(QOCX_ERR_ARG, "cotangent steps must be distinct and in 1..N-1");
        seen[steps[c]] = true;
    }
    lb.inj_steps.assign(steps, steps + count);
    lb.inj_host.assign(bars, bars + (size_t)batch * count * lb.S * lb.n * lb.n * 2);
    lb.inj_count = count;
    lb.inj_batch = batch;
    return 0;
}

int qocx_download_step_densities(qocx_ctx* ctx, double* densities_out) {
    if (!ctx || !densities_out) return fail(QOCX_ERR_ARG, "NULL argument");
    auto& lb = ctx->lb;
    if (!lb.have_results || !lb.have_steps)
        return fail(QOCX_ERR_STATE, "step densities were not kept (qocx_set_keep_step_states)");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t per_seed = (size_t)(lb.nsteps + 1) * lb.S;
    const size_t md = dump_elems(lb.n);
    std::vector<double2> tmp((size_t)lb.B * per_seed * md);
    HIP_TRY(hipMemcpy(tmp.data(), lb.step_densities.p, tmp.size() * sizeof(double2),
                      hipMemcpyDeviceToHost));
    for (int pos = 0; pos < lb.B; ++pos)
        for (size_t v = 0; v < per_seed; ++v)
            from_c_dump(tmp.data() + ((size_t)pos * per_seed + v) * md, lb.n,
                        densities_out + ((size_t)lb.order[pos] * per_seed + v) * lb.n * lb.n * 2);
    return 0;
}

// ---- RCCL --------------------------------------------------------------------------------

int qocx_comm_unique_id(uint8_t* id128) {
    if (!id128) return fail(QOCX_ERR_ARG, "id128 is NULL");
    qocx_ctx tmp;
    int rc = load_rccl(&tmp);
    if (rc) return rc;
    int e = tmp.rccl.GetUniqueId((void*)id128);
    if (e != 0) return fail(QOCX_ERR_RCCL, "ncclGetUniqueId failed");
    return 0;
}

int qocx_comm_init(qocx_ctx* ctx, const uint8_t* id128, int32_t rank, int32_t world) {
    if (!ctx || !id128) return fail(QOCX_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = load_rccl(ctx);
    if (rc) return rc;
    // ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId id (by value), int rank)
    typedef int (*init_fn)(void**, int, ncclUniqueIdBytes, int);
    init_fn f = (init_fn)dlsym(ctx->rccl.lib, "ncclCommInitRank");
    ncclUniqueIdBytes uid;
    memcpy(uid.internal, id128, 128);
    if (world < 1 || rank < 0 || rank >= world) return fail(QOCX_ERR_ARG, "rank / world out of range");
    if (!f) return fail(QOCX_ERR_RCCL, "librccl has no ncclCommInitRank");
    if (getenv("QOCX_RCCL_DEBUG")) {  // (a log line only: what this rank hands to ncclCommInitRank)
        unsigned long long sum = 1469598103934665603ull;  // FNV-1a of the id: equal on every rank
        for (int i = 0; i < 128; ++i) sum = (sum ^ id128[i]) * 1099511628211ull;
        char bus[64] = "?";
        (void)hipDeviceGetPCIBusId(bus, sizeof(bus), ctx->device);
        int version = 0;
        typedef int (*ver_fn)(int*);
        if (ver_fn v = (ver_fn)dlsym(ctx->rccl.lib, "ncclGetVersion")) (void)v(&version);
        const char* ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
        fprintf(stderr, "[qocx rccl] ncclCommInitRank rank=%d world=%d hip_device=%d pci=%s id_fnv=%016llx "
                        "rccl_version=%d HSA_ENABLE_IPC_MODE_LEGACY=%s pid=%d\n",
                rank, world, ctx->device, bus, sum, version, ipc ? ipc : "(unset)", (int)getpid());
        fflush(stderr);
    }
    int e = f(&ctx->comm, world, uid, rank);
    if (e != 0)
        return fail(QOCX_ERR_RCCL, std::string("ncclCommInitRank: ") +
                                       (ctx->rccl.GetErrorString ? ctx->rccl.GetErrorString(e) : "?"));
    return 0;
}

static int comm_allreduce(qocx_ctx* ctx, double* buf, int64_t count, int op) {
    if (!ctx || !buf || count < 1) return fail(QOCX_ERR_ARG, "bad argument");
    if (!ctx->comm) return fail(QOCX_ERR_STATE, "communicator not initialised");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->comm_buf.ensure((size_t)count)) return QOCX_ERR_HIP;
    HIP_TRY(hipMemcpyAsync(ctx->comm_buf.p, buf, count * sizeof(double), hipMemcpyHostToDevice,
                           ctx->stream));
    // ncclFloat64 = 8 ; ncclSum = 0, ncclMax = 2
    int e = ctx->rccl.AllReduce(ctx->comm_buf.p, ctx->comm_buf.p, (size_t)count, 8, op, ctx->comm,
                                ctx->stream);
    if (e != 0)
        return fail(QOCX_ERR_RCCL, std::string("ncclAllReduce: ") +
                                       (ctx->rccl.GetErrorString ? ctx->rccl.GetErrorString(e) : "?"));
    HIP_TRY(hipMemcpyAsync(buf, ctx->comm_buf.p, count * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int qocx_opt_begin(qocx_ctx* ctx) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (!ctx->has_problem || ctx->B < 1 || ctx->K < 1 || ctx->explicit_mode)
        return fail(QOCX_ERR_STATE, "qocx_opt_begin needs uploaded controls of a structured problem");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t total = (size_t)ctx->B * ctx->nc * ctx->K;
    if (ctx->opt_m.ensure(total) || ctx->opt_v.ensure(total) || ctx->opt_best_controls.ensure(total) ||
        ctx->opt_best_final.ensure((size_t)ctx->B * ctx->S * ctx->np) ||
        ctx->opt_flags.ensure(2 * (size_t)ctx->B) || ctx->opt_max_norms.ensure((size_t)ctx->K))
        return QOCX_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->opt_m.p, 0, total * sizeof(double), ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->opt_v.p, 0, total * sizeof(double), ctx->stream));
    ctx->opt_batch = ctx->B;
    return 0;
}

int qocx_opt_clip(qocx_ctx* ctx, const double* max_norms) {
    if (!ctx || !max_norms) return fail(QOCX_ERR_ARG, "NULL argument");
    if (ctx->opt_batch != ctx->B || ctx->B < 1) return fail(QOCX_ERR_STATE, "qocx_opt_begin has not run for this batch");
    HIP_TRY(hipSetDevice(ctx->device));
    // after the clip |u_k| <= max_norms[k]: the squaring capacity follows from that bound
    double bound = ctx->h0_norm_max;
    for (int k = 0; k < ctx->K; ++k) {
        if (!(max_norms[k] >= 0)) return fail(QOCX_ERR_ARG, "max_norms must be non-negative");
        bound += max_norms[k] * ctx->g_norm_max[k];
    }
    bound = magnus_norm_bound(ctx->nodes, bound * fabs(ctx->dt));
    if (!(bound < 1e300)) return fail(QOCX_ERR_ARG, "non-finite bound");
    const int sb = pade_scale_count(bound);
    if (sb > 10)
        return fail(QOCX_ERR_CAPACITY,
                    "||dt H||_1 bound needs more than 2^10 squaring sub-steps per step; reduce dt");
    ctx->sbound = std::max(ctx->sbound, sb);
    ctx->norm_bound = std::max(ctx->norm_bound, bound);
    ctx->norm_bound_mid = 1e300;  // (the controls move on the device from here on)
    ctx->slot_cap = ((size_t)ctx->nsteps << ctx->sbound) + 1;
    HIP_TRY(hipMemcpyAsync(ctx->opt_max_norms.p, max_norms, ctx->K * sizeof(double),
                           hipMemcpyHostToDevice, ctx->stream));
    if (((size_t)ctx->B * ctx->nc * ctx->K + 255) / 256 > 0x7fffffffu || (size_t)ctx->nc * ctx->K > 65535u * 256u)
        return fail(QOCX_ERR_ARG, "control arrays too large for the optimizer kernels' grids");
    qocx::launch_clip_controls(ctx->controls.p, (size_t)ctx->B * ctx->nc * ctx->K, ctx->K,
                               ctx->opt_max_norms.p, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // max_norms is the caller's memory
    ctx->have_results = false;
    return 0;
}

int qocx_download_costs(qocx_ctx* ctx, double* cost_out) {
    if (!ctx || !cost_out) return fail(QOCX_ERR_ARG, "NULL argument");
    if (!ctx->have_results) return fail(QOCX_ERR_STATE, "no evaluation results");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(cost_out, ctx->cost_out.p, (size_t)ctx->B * sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int qocx_opt_step(qocx_ctx* ctx, int32_t kind, const uint8_t* improved, const uint8_t* update,
                  double learning_rate, double beta_1, double beta_2, double epsilon, double corr_1,
                  double corr_2, int32_t apply_clip_grads, double clip_grads) {
    if (!ctx || !improved || !update) return fail(QOCX_ERR_ARG, "NULL argument");
    if (kind != 0 && kind != 1) return fail(QOCX_ERR_ARG, "kind must be 0 (SGD) or 1 (Adam)");
    if (ctx->opt_batch != ctx->B || ctx->B < 1) return fail(QOCX_ERR_STATE, "qocx_opt_begin has not run for this batch");
    if (!ctx->have_results || !ctx->have_grads) return fail(QOCX_ERR_STATE, "no gradients to step with");
    HIP_TRY(hipSetDevice(ctx->device));
    const int B = ctx->B;
    const size_t per_seed = (size_t)ctx->nc * ctx->K;
    HIP_TRY(hipMemcpyAsync(ctx->opt_flags.p, improved, B, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->opt_flags.p + B, update, B, hipMemcpyHostToDevice, ctx->stream));
    qocx::launch_keep_best(ctx->controls.p, ctx->opt_best_controls.p, per_seed, ctx->final_out.p,
                           ctx->opt_best_final.p, (size_t)ctx->S * ctx->np, ctx->opt_flags.p, B,
                           ctx->stream);
    qocx::OptimArgs a;
    a.kind = kind;
    a.params = ctx->controls.p; a.grads = ctx->grads.p;
    a.moment = ctx->opt_m.p; a.square_moment = ctx->opt_v.p;
    a.update = ctx->opt_flags.p + B;
    a.per_seed = per_seed;
    a.learning_rate = learning_rate; a.beta_1 = beta_1; a.beta_2 = beta_2;
    a.one_m_b1 = 1 - beta_1; a.one_m_b2 = 1 - beta_2;
    a.epsilon = epsilon; a.corr_1 = corr_1; a.corr_2 = corr_2;
    a.clip = clip_grads; a.apply_clip = apply_clip_grads ? 1 : 0;
    qocx::launch_optimizer_update(a, B, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the flag arrays are the caller's memory
    ctx->have_results = false;  // the resident controls are no longer those of the last evaluation
    return 0;
}

int qocx_opt_download_best(qocx_ctx* ctx, double* controls_out, double* final_out) {
    if (!ctx) return fail(QOCX_ERR_ARG, "ctx is NULL");
    if (ctx->opt_batch != ctx->B || ctx->B < 1) return fail(QOCX_ERR_STATE, "qocx_opt_begin has not run for this batch");
    HIP_TRY(hipSetDevice(ctx->device));
    const int B = ctx->B, np = ctx->np, S = ctx->S, n = ctx->n;
    if (controls_out)
        HIP_TRY(hipMemcpyAsync(controls_out, ctx->opt_best_controls.p,
                               (size_t)B * ctx->nc * ctx->K * sizeof(double), hipMemcpyDeviceToHost,
                               ctx->stream));
    std::vector<double2> fin;
    if (final_out) {
        fin.resize((size_t)B * S * np);
        HIP_TRY(hipMemcpyAsync(fin.data(), ctx->opt_best_final.p, fin.size() * sizeof(double2),
                               hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (final_out)
        for (size_t v = 0; v < (size_t)B * S; ++v)
            for (int i = 0; i < n; ++i) {
                final_out[2 * (v * n + i)] = fin[v * np + i].x;
                final_out[2 * (v * n + i) + 1] = fin[v * np + i].y;
            }
    return 0;
}

int qocx_reduce_results(qocx_ctx* ctx, int32_t allreduce, double* out, int64_t count) {
    if (!ctx || !out) return fail(QOCX_ERR_ARG, "NULL argument");
    if (!ctx->have_results) return fail(QOCX_ERR_STATE, "no evaluation results");
    if (allreduce && !ctx->comm) return fail(QOCX_ERR_STATE, "communicator not initialised");
    const int per_seed = count > 1 ? ctx->nc * ctx->K : 0;  // count == 1: the cost only
    if (count != 1 + per_seed || (per_seed > 0 && !ctx->have_grads))
        return fail(QOCX_ERR_ARG, "count must be 1, or 1 + control_eval_count * control_count after "
                                  "an evaluation with gradients");
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->comm_buf.ensure((size_t)count)) return QOCX_ERR_HIP;
    qocx::launch_reduce_results(ctx->cost_out.p, ctx->grads.p, ctx->B, per_seed, ctx->comm_buf.p,
                                ctx->stream);
    if (allreduce) {  // ncclFloat64 = 8 ; ncclSum = 0
        int e = ctx->rccl.AllReduce(ctx->comm_buf.p, ctx->comm_buf.p, (size_t)count, 8, 0, ctx->comm,
                                    ctx->stream);
        if (e != 0)
            return fail(QOCX_ERR_RCCL, std::string("ncclAllReduce: ") +
                                           (ctx->rccl.GetErrorString ? ctx->rccl.GetErrorString(e) : "?"));
    }
    HIP_TRY(hipMemcpyAsync(out, ctx->comm_buf.p, count * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

int qocx_comm_allreduce_sum(qocx_ctx* ctx, double* buf_host, int64_t count) {
    return comm_allreduce(ctx, buf_host, count, 0);
}

int qocx_comm_allreduce_max(qocx_ctx* ctx, double* buf_host, int64_t count) {
    return comm_allreduce(ctx, buf_host, count, 2);
}

int qocx_comm_barrier(qocx_ctx* ctx) {
    double one = 1.0;
    return comm_allreduce(ctx, &one, 1, 0);
}

int qocx_comm_destroy(qocx_ctx* ctx) {
    if (!ctx) return 0;
    if (ctx->comm && ctx->rccl.CommDestroy) ctx->rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    return 0;
}

// ---- debug -------------------------------------------------------------------------------

int qocx_debug_pade_factor(qocx_ctx* ctx, int32_t count, int32_t n, const double* a, double* q_out,
                           double* lu_out, int32_t* perm_out, double* dinv_out, int32_t* s_out) {
    if (!ctx || !a || count < 1) return fail(QOCX_ERR_ARG, "bad argument");
    if (n < 1 || n > 64) return fail(QOCX_ERR_ARG, "n must be in 1..64");
    HIP_TRY(hipSetDevice(ctx->device));
    const int nb = (n <= 16) ? 1 : (n <= 32 ? 2 : 4), np = 16 * nb, mat = np * np;
    DevBuf<double2> a_d, q_d, lu_d, dinv_d;
    DevBuf<int> perm_d, iperm_d, s_d;
    int rc = a_d.ensure((size_t)count * n * n) | q_d.ensure((size_t)count * mat) |
             lu_d.ensure((size_t)count * mat) | dinv_d.ensure((size_t)count * np) |
             perm_d.ensure((size_t)count * np) | iperm_d.ensure((size_t)count * np) |
             s_d.ensure(count);
    if (rc) return QOCX_ERR_HIP;
    HIP_TRY(hipMemcpy(a_d.p, a, (size_t)count * n * n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(ctx->status.p, 0, sizeof(int), ctx->stream));
    qocx::FactorArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.q_img = q_d.p; fa.lu_img = lu_d.p; fa.s_arr = s_d.p; fa.status = ctx->status.p;
    fa.nsteps = count; fa.step0 = 0; fa.seg_len = count; fa.n = n;
    fa.pade_policy = (int)ctx->knob("pade_order", 0);
    const bool inverse = nb <= 2 && ctx->knob("lu_inverse", 0) != 0;  // P^-1 instead of the factors
    const bool fused_lu = nb == 2 && !inverse && qocx::diag_getenv("QOCX_PQ1") == nullptr && ctx->knob("fuse_lu", 1) != 0;
    fa.fuse_lu = fused_lu ? 1 : 0;  // the same kernels the evaluation runs
    fa.lu_mfma = (int)ctx->knob("lu_mfma", 1);
    fa.lu_dpp = (int)ctx->knob("lu_dpp", 1);
    if (ctx->lu_fallbacks.ensure(1)) return QOCX_ERR_HIP;
    HIP_TRY(hipMemsetAsync(ctx->lu_fallbacks.p, 0, sizeof(int), ctx->stream));
    fa.lu_fallbacks = ctx->lu_fallbacks.p;
    fa.dinv = dinv_d.p; fa.perm = perm_d.p; fa.iperm = iperm_d.p;
    qocx::LuArgs la;
    la.lu_img = lu_d.p; la.dinv = dinv_d.p; la.perm = perm_d.p; la.iperm = iperm_d.p;
    la.status = ctx->status.p;
    la.nsteps = count; la.step0 = 0; la.seg_len = count; la.n = n;
    la.inverse = inverse ? 1 : 0;
    {   // the four-to-a-wave inverse of n <= 16 (qocx_lu5.h) where every matrix handed in qualifies
        double theta = 0.0;
        for (int c = 0; c < count; ++c) theta = std::max(theta, one_norm(a + (size_t)c * n * n * 2, n));
        la.all_dominant = (pade_eps_max(theta) <= 0.40 && ctx->knob("lu_dpp", 1) != 0) ? 1 : 0;
    }
    la.fallbacks = ctx->lu_fallbacks.p;
    DevBuf<int> redo_d;
    if (nb == 4 && ctx->knob("lu_mfma", 1) != 0) {
        if (redo_d.ensure((size_t)count)) return QOCX_ERR_HIP;
        la.redo = redo_d.p;
    }
    qocx::launch_pq_explicit(nb, a_d.p, n, fa, count, ctx->stream);
    if (!fused_lu) qocx::launch_lu(nb, la, (size_t)count, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<double2> img((size_t)count * mat), dv((size_t)count * np);
    std::vector<int> pm((size_t)count * np), sv(count);
    HIP_TRY(hipMemcpy(pm.data(), perm_d.p, pm.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(img.data(), q_d.p, img.size() * 16, hipMemcpyDeviceToHost));
    if (q_out)
        for (int m = 0; m < count; ++m)
            from_image(img.data() + (size_t)m * mat, n, np, nullptr, q_out + (size_t)m * n * n * 2);
    HIP_TRY(hipMemcpy(img.data(), lu_d.p, img.size() * 16, hipMemcpyDeviceToHost));
    if (lu_out && inverse) {  // the image is P^-1, column-major
        for (int m = 0; m < count; ++m)
            from_image(img.data() + (size_t)m * mat, n, np, nullptr, lu_out + (size_t)m * n * n * 2);
        if (s_out) {
            HIP_TRY(hipMemcpy(sv.data(), s_d.p, sv.size() * 4, hipMemcpyDeviceToHost));
            memcpy(s_out, sv.data(), count * sizeof(int));
        }
        a_d.release(); q_d.release(); lu_d.release(); dinv_d.release(); perm_d.release(); redo_d.release();
        iperm_d.release(); s_d.release();
        int st_inv = 0;
        HIP_TRY(hipMemcpy(&st_inv, ctx->status.p, sizeof(int), hipMemcpyDeviceToHost));
        if (st_inv & 1) return fail(QOCX_ERR_SINGULAR, "Singular matrix");
        return 0;
    }
    if (lu_out)
        for (int m = 0; m < count; ++m) {
            std::vector<int> rows(pm.begin() + (size_t)m * np, pm.begin() + (size_t)(m + 1) * np);
            for (auto& r : rows) r = std::min(std::max(r, 0), np - 1);
            from_image(img.data() + (size_t)m * mat, n, np, rows.data(), lu_out + (size_t)m * n * n * 2);
        }
    HIP_TRY(hipMemcpy(dv.data(), dinv_d.p, dv.size() * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(sv.data(), s_d.p, sv.size() * 4, hipMemcpyDeviceToHost));
    if (lu_out)  // the device stores U' = D^-1 U above the diagonal: undo the row scaling
        for (int m = 0; m < count; ++m)
            for (int r = 0; r < n; ++r) {
                const double2 d = dv[(size_t)m * np + r];
                const double den = d.x * d.x + d.y * d.y;
                const double ur = d.x / den, ui = -d.y / den;  // U_rr = 1 / dinv_r
                for (int c = r + 1; c < n; ++c) {
                    double* e = lu_out + 2 * (((size_t)m * n + r) * n + c);
                    const double xr = e[0], xi = e[1];
                    e[0] = xr * ur - xi * ui;
                    e[1] = xr * ui + xi * ur;
                }
            }
    for (int m = 0; m < count; ++m)
        for (int i = 0; i < n; ++i) {
            if (perm_out) perm_out[(size_t)m * n + i] = pm[(size_t)m * np + i];
            if (dinv_out) {
                dinv_out[2 * ((size_t)m * n + i)] = dv[(size_t)m * np + i].x;
                dinv_out[2 * ((size_t)m * n + i) + 1] = dv[(size_t)m * np + i].y;
            }
        }
    if (s_out) memcpy(s_out, sv.data(), count * sizeof(int));
    a_d.release(); q_d.release(); lu_d.release(); dinv_d.release(); perm_d.release(); redo_d.release();
    iperm_d.release(); s_d.release();
    int status = 0;
    HIP_TRY(hipMemcpy(&status, ctx->status.p, sizeof(int), hipMemcpyDeviceToHost));
    if (status & 1) return fail(QOCX_ERR_SINGULAR, "Singular matrix");
    return 0;
}

int qocx_debug_mfma_peak(qocx_ctx* ctx, int32_t waves_per_simd, int32_t iters, double* tflops) {
    if (!ctx || !tflops || waves_per_simd < 1 || iters < 1) return fail(QOCX_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    const int blocks = prop.multiProcessorCount * 4 * waves_per_simd;
    DevBuf<double> out;
    if (out.ensure(8)) return QOCX_ERR_HIP;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const int peak_mode = (int)ctx->knob("peak_mode", 0);  // (diagnostic build: pipe_mix_kernel)
    (void)peak_mode;
#ifdef QOCX_DIAG
    if (peak_mode > 0) {
        qocx::launch_pipe_mix(out.p, blocks, 64, peak_mode, ctx->stream);
        HIP_TRY(hipEventRecord(e0, ctx->stream));
        qocx::launch_pipe_mix(out.p, blocks, iters, peak_mode, ctx->stream);
        HIP_TRY(hipEventRecord(e1, ctx->stream));
    } else
#endif
    {
    qocx::launch_mfma_peak(out.p, blocks, 64, ctx->stream);  // warm-up
    HIP_TRY(hipEventRecord(e0, ctx->stream));
    qocx::launch_mfma_peak(out.p, blocks, iters, ctx->stream);
    HIP_TRY(hipEventRecord(e1, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    out.release();
    // one v_mfma_f64_16x16x4_f64 = 16*16*4 multiply-adds = 2048 flop per wave
    *tflops = (double)blocks * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12;
    return 0;
}

int qocx_debug_selftest(qocx_ctx* ctx, int32_t* failures, char* report, int32_t report_len) {
    if (!ctx || !failures) return fail(QOCX_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf<double> out;
    if (out.ensure(512)) return QOCX_ERR_HIP;
    qocx::launch_selftest(out.p, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<double> h(512);
    HIP_TRY(hipMemcpy(h.data(), out.p, 512 * sizeof(double), hipMemcpyDeviceToHost));
    out.release();
    int bad = 0;
    std::string rep;
    double vmax = 0, vsum = 0;
    std::vector<double> v(64);
    for (int l = 0; l < 64; ++l) {
        v[l] = (double)((l * 37) % 64) + 0.25;
        vmax = std::max(vmax, v[l]);
        vsum += v[l];
    }
    for (int l = 0; l < 64; ++l) {
        if (h[l] != vmax) { ++bad; rep += "wave_max lane " + std::to_string(l) + "\n"; }
        if (fabs(h[64 + l] - vsum) > 1e-9) { ++bad; rep += "wave_sum lane " + std::to_string(l) + "\n"; }
        if (h[384 + l] != v[5]) { ++bad; rep += "readlane lane " + std::to_string(l) + "\n"; }
        const int mirror = (l & ~15) | (15 - (l & 15));
        if (h[448 + l] != v[mirror]) { ++bad; rep += "row_mirror lane " + std::to_string(l) + "\n"; }
        for (int r = 0; r < 4; ++r) {
            // C[row][col], row = (lane>>4) + 4 r, col = lane & 15 ; A[i][k] = i + 16k, B[k][j] = 100k + j
            const int row = (l >> 4) + 4 * r, col = l & 15;
            double ref = 0;
            for (int k = 0; k < 4; ++k) ref += (double)(row + 16 * k) * (double)(100 * k + col);
            if (h[128 + l * 4 + r] != ref) {
                ++bad;
                if (rep.size() < 2000)
                    rep += "mfma lane " + std::to_string(l) + " r " + std::to_string(r) + " got " +
                           std::to_string(h[128 + l * 4 + r]) + " want " + std::to_string(ref) + "\n";
            }
        }
    }
    *failures = bad;
    if (report && report_len > 0) {
        strncpy(report, rep.c_str(), report_len - 1);
        report[report_len - 1] = 0;
    }
    return 0;
}

int qocx_host_clip_controls(double* controls, int64_t batch, int64_t nc, int32_t k,
                            const double* max_norms) {
    if (!controls || !max_norms || batch < 0 || nc < 0 || k < 0) return fail(QOCX_ERR_ARG, "bad argument");
    host_parallel_rows(batch, [=](int64_t lo, int64_t hi) {
        for (int64_t b = lo; b < hi; ++b) {
            double* row = controls + (size_t)b * nc * k;
            for (int64_t j = 0; j < nc; ++j)
                for (int32_t c = 0; c < k; ++c) {
                    const double v = row[j * k + c], mod = fabs(v);
                    if (max_norms[c] < mod) row[j * k + c] = (v / mod) * max_norms[c];
                }
        }
    });
    return 0;
}

int qocx_host_optimizer_update(int32_t kind, double* params, const double* grads, double* moment,
                               double* square_moment, int64_t p, const int64_t* rows,
                               int64_t row_count, double learning_rate, double beta_1,
                               double beta_2, double epsilon, double corr_1, double corr_2,
                               int32_t apply_clip_grads, double clip_grads) {
    if (!params || !grads || !rows || p < 0 || row_count < 0) return fail(QOCX_ERR_ARG, "bad argument");
    if (kind != 0 && (!moment || !square_moment)) return fail(QOCX_ERR_ARG, "moments missing");
    const double one_m_b1 = 1 - beta_1, one_m_b2 = 1 - beta_2;
    host_parallel_rows(row_count, [=](int64_t lo, int64_t hi) {
// every product and sum is rounded on its own, as NumPy's array operations are
#pragma clang fp contract(off)
        for (int64_t r = lo; r < hi; ++r) {
            const size_t off = (size_t)rows[r] * (size_t)p;
            double* x = params + off;
            const double* g = grads + off;
            if (kind == 0) {
                for (int64_t i = 0; i < p; ++i) {
                    const double s = learning_rate * g[i];
                    x[i] = x[i] - s;
                }
                continue;
            }
            adam_row(x, g, moment + off, square_moment + off, p, learning_rate, beta_1, beta_2,
                     one_m_b1, one_m_b2, epsilon, corr_1, corr_2, apply_clip_grads, clip_grads);
        }
    });
    return 0;
}

}  // extern "C"
