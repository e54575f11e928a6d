// qocx_sweepi.hip - K2i, the inverse-image sweep for latency mode (one control set, few states,
// n <= 32): a propagator sub-step as TWO matrix-vector products, psi' = P^-1 (Q psi).
//
// The column-chain sweep (qocx_kernels.hip) applies P^-1 by two triangular solves: 2 (NP - 1)
// dependent broadcast -> update stages per sub-step, 3.1 us per step at n = 32 and 1.8 us at n <= 16
// however idle the chip is. With P^-1 itself in place of the factors (K1b's sibling inv_kernel,
// qocx_lu.h) a sub-step is two products of the kind the sweep already does for Q - no chain at
// all - and what bounds a step is getting its 2 NP^2 complex numbers into LDS. The inverse costs
// three times the factorisation, which is why the batched evaluator (throughput: 256 000 matrices
// per evaluation) keeps the factors; an evaluation of ONE control set is a serial chain of sweep
// steps and nothing else.
//
// Workgroup = one seed: wave 0 computes, the other waves (two at NP = 32, one at NP = 16) stream the
// step operands in by LDS-DMA, several steps ahead in a ring of buffer sets (counted vmcnt), one
// raw workgroup barrier per step. The adjoint gathers the transposed images, so x = P^-H lambda'
// and lambda = Q^H x are the same product with conjugated elements. Everything else follows
// sweep_kernel: time segments (phase, j_begin, j_end), squaring sub-steps, every state cost,
// host-supplied cotangents, the unit adjoint (the two-sided pipeline runs the forward and the
// adjoint sweep of the seed side by side), the capacity checks.
#include "qocx_sweep_common.h"

namespace qocx {

namespace sweepi {

template <int NB>
struct Cfg {
    typedef Geo<NB> G;
    static constexpr int NP = G::NP, MAT = G::MAT;
    static constexpr int PIECES = MAT / 64;              // KiB pieces per image
    static constexpr int LOADERS = NB == 2 ? 2 : 1;      // fetch waves (four / two measured: no faster)
    static constexpr int PER_LOADER = 2 * PIECES / LOADERS;  // DMA instructions per loader and step
    // buffer sets in the ring: a fetch takes ~2 us to land, a step ~0.4 us to compute, so the
    // fetches run RING - 1 steps ahead (three sets: 1.1 us per step at either size)
    static constexpr int RING = NB == 2 ? 4 : 8;
    static constexpr int AHEAD = RING - 1;
    static constexpr int Q_OFF = 0;                              // RING x Q image
    static constexpr int PI_OFF = Q_OFF + RING * MAT * 16;       // RING x P^-1 image
    static constexpr int TMP_OFF = PI_OFF + RING * MAT * 16;     // NP complex
    static constexpr int TBL_OFF = TMP_OFF + NP * 16;            // the launch's slice of the step table
    static constexpr int TBL = 2048;                             // (longer launches read it from HBM)
    static constexpr int VEC_OFF = TBL_OFF + TBL * 4;            // [S][NP] states, [S][NP] lambda
    static int bytes(int S) { return VEC_OFF + 2 * S * NP * 16; }
};

__device__ __forceinline__ void raw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// y = M v (CONJ: conj(M) v) with M an R-layout (column-major) image in LDS, v a vector in LDS
template <int NB, bool CONJ>
__device__ __forceinline__ void matvec(const double2* img, const double2* vec, int lane, int h,
                                       double& yre, double& yim) {
    typedef Geo<NB> G;
    constexpr int CPL = G::CPL, H = G::H;
    double ar = 0, ai = 0, br = 0, bi = 0;
#pragma unroll
    for (int cc = 0; cc < CPL; ++cc) {
        const double2 m = img[cc * 64 + lane];
        const double2 v = vec[cc * H + h];
        const double mi = CONJ ? -m.y : m.y;
        if (cc & 1) {
            br = fma(-mi, v.y, fma(m.x, v.x, br));
            bi = fma(mi, v.x, fma(m.x, v.y, bi));
        } else {
            ar = fma(-mi, v.y, fma(m.x, v.x, ar));
            ai = fma(mi, v.x, fma(m.x, v.y, ai));
        }
    }
    yre = sum_groups<NB>(ar + br);
    yim = sum_groups<NB>(ai + bi);
}

template <int NB>
__global__ __launch_bounds__(64 * (1 + Cfg<NB>::LOADERS)) void sweepi_kernel(SweepArgs args) {
    typedef Cfg<NB> C;
    constexpr int NP = C::NP, MAT = C::MAT, H = Geo<NB>::H;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_s_setprio(3);
    double2* qbuf = reinterpret_cast<double2*>(smem + C::Q_OFF);
    double2* pibuf = reinterpret_cast<double2*>(smem + C::PI_OFF);
    double2* tmp = reinterpret_cast<double2*>(smem + C::TMP_OFF);
    double2* vecs = reinterpret_cast<double2*>(smem + C::VEC_OFF);
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0: compute, 1..: loaders
    const int i = lane % NP, h = lane / NP;
    const bool g0 = (h == 0);
    const int b = blockIdx.x, S = args.S, nsteps = args.nsteps;
    double2* lam = vecs + S * NP;
    const size_t cap = args.slot_cap;
    double2* states_b = args.states + (size_t)b * cap * S * NP;
    double2* xs_b = args.xs + (size_t)b * cap * S * NP;
    int* offs_b = args.offs + (size_t)b * (nsteps + 1);
    const size_t m0 = (size_t)b * nsteps;
    const int jb = args.j_begin, je = args.j_end, T = je - jb;
    const bool do_fwd = (args.phase & 1) != 0, do_bwd = (args.phase & 2) != 0;
    const bool unit = args.unit_adjoint != 0;
    int* offs_x = unit ? args.offs_x + (size_t)b * (nsteps + 1) : nullptr;
    // The step table of the launch goes to LDS once: a scalar load per step would sit in front of
    // every barrier and LDS wait of the loop (they share a counter) - 0.3 us of a 0.4 us step.
    int* tbl = reinterpret_cast<int*>(smem + C::TBL_OFF);
    const bool tbl_on = T <= C::TBL;
    {   // an earlier segment (or the other sweep of a two-sided evaluation) overflowed: every wave
        // takes the same decision
        __shared__ int status_seen;
        if (threadIdx.x == 0) status_seen = *(volatile int*)args.status;
        if (tbl_on)
            for (int e = threadIdx.x; e < T; e += blockDim.x) tbl[e] = args.s_arr[m0 + jb + e];
        __syncthreads();
        if ((jb > 0 || !do_fwd) && (status_seen & 4)) return;
    }
    auto entry_of = [&](int step) { return tbl_on ? tbl[step - jb] : args.s_arr[m0 + step]; };

    // ---- the fetch waves --------------------------------------------------------------------
    // Both roles walk the same steps and leave at the same one when the sub-step capacity runs
    // out: the test depends on slot counts only, which every wave keeps.
    auto issue = [&](size_t m, int ring, bool adjoint) {
        // (umode, two fetch waves: the second one's pieces are the P^-1 image, which nobody reads - it keeps
        // walking the barriers)
        if (C::LOADERS == 2 && args.umode && role == 2) return;
#pragma unroll
        for (int jj = 0; jj < C::PER_LOADER; ++jj) {
            const int j = (role - 1) * C::PER_LOADER + jj, jl = j % C::PIECES;
            // (umode, adjoint: the stored U^T image, a straight copy like the forward one)
            const bool stored_t = adjoint && args.umode && j < C::PIECES;
            const double2* img = (stored_t ? args.qt_img : j < C::PIECES ? args.q_img : args.lu_img) + m * MAT;
            double2* dst = (j < C::PIECES ? qbuf : pibuf) + ring * MAT + jl * 64;
            // plain image: piece jl is its KiB jl; transposed (adjoint): LDS element (col H jl + l / NP,
            // row l % NP) is image element (row H jl + l / NP, col l % NP)
            const size_t el = (adjoint && !stored_t) ? (size_t)(lane % NP) * NP + H * jl + lane / NP
                                                     : (size_t)jl * 64 + lane;
            dma16(img + el, dst);
        }
    };
    // one pass of a fetch wave over the T steps of the launch, `first` + d t being step t
    auto loader_pass = [&](bool adjoint, int slot) {
        const int first = adjoint ? je - 1 : jb, d = adjoint ? -1 : 1;
        for (int p = 0; p < C::AHEAD && p < T; ++p) issue(m0 + first + d * p, p, adjoint);
        for (int t = 0; t < T; ++t) {
            const int nsub = 1 << step_squarings(entry_of(first + d * t));
            if (!adjoint && (size_t)slot + nsub >= cap) return false;
            if (adjoint && unit && slot - nsub < 0) return false;
            // step t has landed once at most the fetches of the younger steps are in flight
            const int younger = min(C::AHEAD - 1, T - 1 - t);
            static_assert((C::AHEAD - 1) * C::PER_LOADER < 64, "vmcnt is a six-bit counter");
            switch (younger) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * C::PER_LOADER) : "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::PER_LOADER) : "memory"); break;
                case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((3 * C::PER_LOADER) & 63) : "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((4 * C::PER_LOADER) & 63) : "memory"); break;
                case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((5 * C::PER_LOADER) & 63) : "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((6 * C::PER_LOADER) & 63) : "memory"); break;
            }
            raw_barrier();  // ... / the compute wave has left step t - 1, whose buffers are free
            if (t + C::AHEAD < T)
                issue(m0 + first + d * (t + C::AHEAD), (t + C::AHEAD) % C::RING, adjoint);
            slot += adjoint ? -nsub : nsub;
        }
        raw_barrier();  // the compute wave has left the last step
        return true;
    };
    int slot0 = 0;  // first sub-step slot of the launch (forward) / one past its last (adjoint)
    if (do_fwd) slot0 = (jb == 0) ? 0 : offs_b[jb];
    else if (unit) slot0 = (je == nsteps) ? (int)cap : offs_x[je];
    else slot0 = offs_b[je];
    if (role != 0) {
        if (do_fwd)
            if (!loader_pass(false, slot0)) return;
        if (do_bwd) {
            int s_adj = slot0;
            if (do_fwd) {  // phase 3 (one time segment): where the forward pass ended / the capacity
                s_adj = unit ? (int)cap : 0;
                if (!unit) {
                    s_adj = slot0;
                    for (int t = 0; t < T; ++t) s_adj += 1 << step_squarings(args.s_arr[m0 + jb + t]);
                }
            }
            if (do_fwd) raw_barrier();  // A0: the compute wave has finished its forward epilogue
            (void)loader_pass(true, s_adj);
        }
        return;
    }

    // ---- the compute wave -------------------------------------------------------------------
    double cost = 0;
    int slot = slot0;
    if (do_fwd) {
        if (jb == 0) {
            for (int s = 0; s < S; ++s)
                if (g0) {
                    const double2 p = args.psi0[s * NP + i];
                    vecs[s * NP + i] = p;
                    states_b[(size_t)s * NP + i] = p;
                }
        } else {  // resume: states and partial cost left by the previous segment
            cost = args.cost_out[b];
            for (int s = 0; s < S; ++s)
                if (g0) vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
        }
        wave_sync();
        for (int t = 0; t < T; ++t) {
            const int step = jb + t;
            const int nsub = 1 << step_squarings(entry_of(step));
            if ((size_t)slot + nsub >= cap) {
                atomicOr(args.status, 4);
                return;
            }
            raw_barrier();  // the operands of step t have landed
            if (step != 0 && args.has_step_costs && (step % args.cost_eval_step) == 0)
                cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
            if (g0 && args.step_states != nullptr)
                for (int s = 0; s < S; ++s)
                    args.step_states[(((size_t)b * (nsteps + 1) + step) * S + s) * NP + i] = vecs[s * NP + i];
            if (lane == 0) offs_b[step] = slot;
            const double2* qc = qbuf + (t % C::RING) * MAT;
            const double2* pc = pibuf + (t % C::RING) * MAT;
            for (int sub = 0; sub < nsub; ++sub) {
                for (int s = 0; s < S; ++s) {
                    double zre, zim;
                    matvec<NB, false>(qc, vecs + s * NP, lane, h, zre, zim);  // z = Q psi (umode: psi' = U psi)
                    if (!args.umode) {
                        tmp[i] = make_double2(zre, zim);
                        wave_sync();
                        matvec<NB, false>(pc, tmp, lane, h, zre, zim);        // psi' = P^-1 z
                    }
                    wave_sync();
                    const double2 p = make_double2(zre, zim);
                    vecs[s * NP + i] = p;
                    if (g0) states_b[((size_t)(slot + 1) * S + s) * NP + i] = p;
                    wave_sync();
                }
                ++slot;
            }
        }
        raw_barrier();  // (the fetch waves' closing barrier)
        if (je == nsteps) {
            if (args.has_step_costs && nsteps != 0 && (nsteps % args.cost_eval_step) == 0)
                cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
            if (g0 && args.step_states != nullptr)
                for (int s = 0; s < S; ++s)
                    args.step_states[(((size_t)b * (nsteps + 1) + nsteps) * S + s) * NP + i] = vecs[s * NP + i];
            if (lane == 0) offs_b[nsteps] = slot;
            cost += eval_costs<NB>(args, false, true, vecs, nullptr, h, i);
            if (unit && args.want_grad) unit_adjoint_scales<NB>(args, vecs, b, h, i);
            if (g0)
                for (int s = 0; s < S; ++s) args.final_out[((size_t)b * S + s) * NP + i] = vecs[s * NP + i];
        } else if (lane == 0) {
            offs_b[je] = slot;  // the next segment resumes from here
        }
        if (lane == 0) args.cost_out[b] = cost;
    }
    if (!do_bwd) return;

    // lambda += host-supplied cotangent of the states at system step `step`, if there is one
    auto inject = [&](int step) {
        if (args.inj_index == nullptr) return;
        const int row = args.inj_index[step];
        if (row < 0) return;
        if (g0)
            for (int s = 0; s < S; ++s) {
                const double2 e = args.inj_bars[(((size_t)b * args.inj_count + row) * S + s) * NP + i];
                double2 l = lam[s * NP + i];
                l.x += e.x;
                l.y += e.y;
                lam[s * NP + i] = l;
            }
        wave_sync();
    };
    // ---- adjoint sweep ----------------------------------------------------------------------
    if (je == nsteps && unit) {
        unit_adjoint_seed<NB>(args, lam, 0, 1, h, i);  // lambda = the targets
        slot = (int)cap;
        wave_sync();
    } else if (je == nsteps) {
        if (!do_fwd)
            for (int s = 0; s < S; ++s)
                if (g0) vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
        for (int s = 0; s < S; ++s)
            if (g0) lam[s * NP + i] = make_double2(0, 0);
        wave_sync();
        // cotangent seeds on the final states: non-step costs, and step costs if the final step
        // is a cost step (schroedingerdiscrete.py:412-416 evaluates them before the loop ends)
        (void)eval_costs<NB>(args, (nsteps % args.cost_eval_step) == 0, true, vecs, lam, h, i);
        inject(nsteps);
    } else {  // resume the adjoint sweep below step je
        for (int s = 0; s < S; ++s)
            if (g0) lam[s * NP + i] = args.lam_buf[((size_t)b * S + s) * NP + i];
        wave_sync();
    }
    if (do_fwd) raw_barrier();  // A0
    for (int t = 0; t < T; ++t) {
        const int step = je - 1 - t;
        const int nsub = 1 << step_squarings(entry_of(step));
        if (unit && slot - nsub < 0) {
            atomicOr(args.status, 4);
            return;
        }
        raw_barrier();  // the (transposed) operands of the step have landed
        const double2* qc = qbuf + (t % C::RING) * MAT;   // Q^T: conj gives Q^H
        const double2* pc = pibuf + (t % C::RING) * MAT;  // (P^-1)^T
        for (int sub = nsub - 1; sub >= 0; --sub) {
            --slot;
            for (int s = 0; s < S; ++s) {
                double yre, yim;
                if (args.umode) {
                    // lambda = U^H lambda'; K3 forms x = P^-H lambda' from what goes to `xs` here
                    if (g0) xs_b[((size_t)slot * S + s) * NP + i] = lam[s * NP + i];
                    matvec<NB, true>(qc, lam + s * NP, lane, h, yre, yim);
                } else {
                    double xre, xim;
                    matvec<NB, true>(pc, lam + s * NP, lane, h, xre, xim);  // x = P^-H lambda'
                    const double2 x = make_double2(xre, xim);
                    tmp[i] = x;
                    if (g0) xs_b[((size_t)slot * S + s) * NP + i] = x;
                    wave_sync();
                    matvec<NB, true>(qc, tmp, lane, h, yre, yim);            // lambda = Q^H x
                }
                wave_sync();
                lam[s * NP + i] = make_double2(yre, yim);
                wave_sync();
            }
        }
        if (step != 0 && (step % args.cost_eval_step) == 0 && args.has_step_costs) {
            // step costs were evaluated on the states BEFORE evolving from `step`
            if (g0)
                for (int s = 0; s < S; ++s) vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
            wave_sync();
            (void)eval_costs<NB>(args, true, false, vecs, lam, h, i);
        }
        if (step != 0) inject(step);
        if (unit && lane == 0) offs_x[step] = slot;
    }
    raw_barrier();
    if (jb > 0 && g0)
        for (int s = 0; s < S; ++s) args.lam_buf[((size_t)b * S + s) * NP + i] = lam[s * NP + i];
}

}  // namespace sweepi

// (the states of a seed go through the one compute wave one after the other: measured at n = 16,
// S = 16 that costs 9.6 us per step against 6.6 for the four-wave column chains - up to eight)
bool sweepi_supports(int nb, int S) {
    return (nb == 1 || nb == 2) && S >= 1 && S <= 8;
}

template <int NB>
static void launch_sweepi_t(const SweepArgs& a, int batch, hipStream_t st) {
    typedef sweepi::Cfg<NB> C;
    const int bytes = C::bytes(a.S);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweepi::sweepi_kernel<NB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((sweepi::sweepi_kernel<NB>), dim3(batch), dim3(64 * (1 + C::LOADERS)), bytes, st, a);
}

void launch_sweepi(int nb, const SweepArgs& a, int batch, hipStream_t st) {
    if (nb == 1) launch_sweepi_t<1>(a, batch, st);
    else launch_sweepi_t<2>(a, batch, st);
}

}  // namespace qocx
