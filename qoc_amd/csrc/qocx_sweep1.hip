// qocx_sweep1.hip - K2 for ONE state per seed, n <= 32: the sweep of the batched evaluator's
// headline path (round 5). Same step, same numbers, bit for bit, as sweep_kernel<NB, 1, false, true>
// of qocx_kernels.hip (tests/test_gpu_engine.py::test_one_state_sweep_equals_general_sweep):
//
//   forward  psi' = U'^-1 D^-1 L^-1 Pi Q psi      (2^s times per propagator step)
//   adjoint  x = Pi^T L^-H D^-H U'^-H lambda',  lambda = Q^H x
//   reference: qoc/core/schroedingerdiscrete.py:393-436 (the loop), qoc/standard/functions/expm.py:246-250
//   (solve(P, Q), the squarings), qoc/standard/costs/*.py (qocx_sweep_common.h)
//
// What is different is everything AROUND the arithmetic - the general kernel spent two thirds of a step
// there (profiles/r05_sweep_parts.txt):
//   * the triangular solves broadcast inside the multiply-add (v_fmac_f64_dpp row_newbcast,
//     qocx_sweep_core.h): 29 cycles per stage instead of 62, and no scalar registers in the chain;
//   * the step's operands arrive by LDS-DMA issued from inline assembly in two bursts at the TOP of the
//     step - the LU image of the next step as soon as this step's rows are in registers, the Q image as
//     soon as the matrix-vector product has read the old one - instead of one piece per solve stage
//     behind a branch each (a taken branch per stage when there was nothing to fetch). Issued by the
//     compiler's builtin, every LDS access behind a DMA waited for vmcnt(0): the fetch never overlapped
//     the end of the step. From assembly the kernel counts for itself: one s_waitcnt vmcnt(0) at the top
//     of a step (everything older than a whole step), one counted wait in front of the adjoint's
//     matrix-vector product (the LU pieces of the next step are younger than the Q pieces it needs);
//   * the state / x vector of a sub-step goes to HBM at the top of the NEXT step, behind that wait, so
//     that no store is the youngest operation a wait has to cover;
//   * squaring counts of 64 steps per load (one lane each), not a dependent load per step.
// One operand set in LDS (35 KiB per seed), so that a K1a workgroup still fits beside two sweeps.
#include "qocx_sweep_core.h"

namespace qocx {

namespace sweep1 {

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// lane l's 16 bytes at g + IMM land at lds_dst + 16 l (M0 = destination - IMM: the hardware adds the
// instruction offset to both addresses)
template <int IMM>
__device__ __forceinline__ void dma16(const char* g, unsigned lds_dst) {
    static_assert(IMM >= -4096 && IMM <= 4095, "instruction offset range");
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
                 :
                 : "v"(g), "s"(lds_dst - (unsigned)IMM), "n"(IMM)
                 : "memory");
}
__device__ __forceinline__ void dma4(const int* g, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(g), "s"(lds_dst) : "memory");
}

template <int NB>
struct Pieces {
    typedef Geo<NB> G;
    static constexpr int IMG = G::MAT / 64;                    // 1 KiB pieces of an image
    static constexpr int GROUP = IMG >= 8 ? 8 : IMG;           // pieces reached from one base address
    static constexpr int CENTER = IMG >= 8 ? 4 : 0;
    static constexpr int NGROUP = IMG / GROUP;
    static constexpr int LU_SET = IMG + 2;                     // LU image, 1 / U_kk, perm | iperm
};

// one image: plain (forward: a straight copy) or transposed (adjoint: lane l of piece j reads element
// (l % NP) NP + l / NP + j H, so that the LDS image is the transpose)
template <int NB, bool ADJ>
__device__ __forceinline__ void burst_image(const double2* img, unsigned lds_dst, int lane) {
    typedef Geo<NB> G;
    typedef Pieces<NB> P;
    if constexpr (ADJ) {
        const char* g = reinterpret_cast<const char*>(img + (size_t)(lane % G::NP) * G::NP + lane / G::NP);
        for_each_const(
            [&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                dma16<j * G::H * 16>(g, lds_dst + j * 1024);
            },
            std::make_integer_sequence<int, P::IMG>{});
    } else {
        for_each_const(
            [&](auto Gi) __attribute__((always_inline)) {
                constexpr int gi = decltype(Gi)::value;
                const char* g = reinterpret_cast<const char*>(img + lane + (size_t)(gi * P::GROUP + P::CENTER) * 64);
                for_each_const(
                    [&](auto J) __attribute__((always_inline)) {
                        constexpr int j = gi * P::GROUP + decltype(J)::value;
                        dma16<(j - (gi * P::GROUP + P::CENTER)) * 1024>(g, lds_dst + j * 1024);
                    },
                    std::make_integer_sequence<int, P::GROUP>{});
            },
            std::make_integer_sequence<int, P::NGROUP>{});
    }
}

// LDS of one seed: RING operand sets (Q image, LU image, 1 / U_kk, perm | iperm), then the vectors
template <int NB, int RING>
struct Lds1 {
    typedef SweepLds<NB, 1> L;
    static constexpr int SLOT = L::TMP_OFF;  // one operand set
    static constexpr int SEED_BYTES = RING * SLOT + (L::bytes_static(1) - L::TMP_OFF);
};

// RING = 1: one operand set - the next step's LU image lands while this step's solves run, its Q
// image once the matrix-vector product has read the old one (n = 17..32: 35 KiB per seed, so that K1a
// workgroups fit beside two sweeps). RING > 1 (n <= 16: 9 KiB per set): the sets of the next RING - 1
// steps are in flight - a step of a 16 x 16 problem is shorter than the trip to memory.
template <int NB, int RING>
__global__ __launch_bounds__(128) void sweep1_kernel(SweepArgs args) {
    typedef Geo<NB> G;
    typedef SweepLds<NB, 1> L;
    typedef Pieces<NB> P;
    typedef Lds1<NB, RING> LR;
    constexpr int NP = G::NP, MAT = G::MAT;
    constexpr int PINTS = L::PINTS;
    constexpr int MVB = G::CPL < 4 ? G::CPL : 4;
    constexpr int SET_PIECES = P::LU_SET + P::IMG;  // LDS-DMA instructions of one operand set
    static_assert(RING >= 1 && (RING - 2) * SET_PIECES <= 63, "vmcnt is a 6-bit count");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // wave v of the workgroup is seed (waves per workgroup) * blockIdx.x + v, with LDS of its own
    const int pack_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* smem = smem_raw + pack_wave * LR::SEED_BYTES;
    __builtin_amdgcn_s_setprio(3);  // the serial chain of the evaluation goes first on its SIMD
    double2* qbuf = reinterpret_cast<double2*>(smem + L::Q_OFF);   // (set 0; set k: + k * SLOT bytes)
    double2* lbuf = reinterpret_cast<double2*>(smem + L::L_OFF);
    double2* dbuf = reinterpret_cast<double2*>(smem + L::D_OFF);
    int* pbuf = reinterpret_cast<int*>(smem + L::P_OFF);
    double2* tmp = reinterpret_cast<double2*>(smem + RING * LR::SLOT);
    double2* vecs = reinterpret_cast<double2*>(smem + RING * LR::SLOT + (L::VEC_OFF - L::TMP_OFF));
    double2* lam = vecs + NP;
    const unsigned q_lds = lds_addr(qbuf), l_lds = lds_addr(lbuf), d_lds = lds_addr(dbuf), p_lds = lds_addr(pbuf);
    auto at_set = [&](auto* ptr, int set) __attribute__((always_inline)) {
        return reinterpret_cast<decltype(ptr)>(reinterpret_cast<char*>(ptr) + set * LR::SLOT);
    };
    // everything but the `younger` youngest operand sets has landed
    auto wait_sets = [&](int younger) __attribute__((always_inline)) {
        if constexpr (RING > 2) {
            if (younger >= RING - 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * SET_PIECES) : "memory"); return; }
        }
        if constexpr (RING > 3) {
            if (younger == RING - 3) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 3) * SET_PIECES) : "memory"); return; }
        }
        if constexpr (RING > 4) {
            if (younger == RING - 4) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 4) * SET_PIECES) : "memory"); return; }
        }
        if (younger >= 1 && RING > 2) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SET_PIECES) : "memory"); return; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int b = (int)(blockDim.x >> 6) * blockIdx.x + pack_wave;
    if (b >= args.batch) return;  // odd batch: the last workgroup has one seed
    const int lane = lane_id(), i = lane % NP, h = lane / NP;
    const int nsteps = args.nsteps;
    const size_t cap = args.slot_cap;
    double2* states_b = args.states + (size_t)b * cap * NP;
    double2* xs_b = args.xs + (size_t)b * cap * NP;
    int* offs_b = args.offs + (size_t)b * (nsteps + 1);
    const bool g0 = (h == 0);
    // (timing experiment, diagnostic build: every seed fetches the images of seed 0 - cache hits)
    const size_t m0 = (QOCX_DBG_BITS(args.dbg) & 16384) ? 0 : (size_t)b * nsteps;
    const int jb = args.j_begin, je = args.j_end;
    const bool do_fwd = (args.phase & 1) != 0, do_bwd = (args.phase & 2) != 0;
    if (jb > 0 || !do_fwd)
        if ((*(volatile int*)args.status) & 4) return;  // an earlier segment overflowed

    // the fetches of a step, as two bursts of LDS-DMA
    auto burst_lu = [&](size_t m, auto ADJ, int set = 0) __attribute__((always_inline)) {
        constexpr bool adj = decltype(ADJ)::value;
        const unsigned off = (unsigned)(set * LR::SLOT);
        burst_image<NB, adj>(args.lu_img + m * MAT, l_lds + off, lane);
        dma16<0>(reinterpret_cast<const char*>(args.dinv + m * NP + i), d_lds + off);
        dma4((lane < 32 ? args.perm : args.iperm) + m * NP + (lane & 31) % NP, p_lds + off);
    };
    auto burst_q = [&](size_t m, auto ADJ, int set = 0) __attribute__((always_inline)) {
        burst_image<NB, decltype(ADJ)::value>(args.q_img + m * MAT, q_lds + (unsigned)(set * LR::SLOT), lane);
    };
    auto scalars = [&](bool adjoint, int set = 0) {
        StepScalars sc;
        sc.dv = at_set(dbuf, set)[i];
        sc.pm = min(max(at_set(pbuf, set)[(adjoint ? PINTS / 2 : 0) + i], 0), NP - 1);
        return sc;
    };
    // squaring counts: lane l holds the entry of step sq_base + l
    int sq_base = 0, sq_word = 0;
    bool sq_valid = false;
    auto substeps = [&](int step, bool downwards) {
        if (!sq_valid || step < sq_base || step >= sq_base + 64) {
            sq_base = downwards ? step - 63 : step;
            const int st = sq_base + lane;
            sq_word = (st >= 0 && st < nsteps) ? args.s_arr[m0 + st] : 0;
            sq_valid = true;
        }
        return 1 << step_squarings(__builtin_amdgcn_readlane(sq_word, step - sq_base));
    };
    auto none = [](auto) {};
    std::true_type ADJ_T;
    std::false_type FWD_T;

    double cost = 0;
    int slot = 0;
    bool overflow = false;
    StepRegs<NB> r;
    if (QOCX_DBG_BITS(args.dbg) & 8192) {  // (timing experiment: the rows are never loaded)
#pragma unroll
        for (int c = 0; c < NP; ++c) r.lre[c] = r.lim[c] = 0.0;
    }

    auto before_step = [&](int step) {  // the state of `step` is in vecs
        if (step != 0 && args.has_step_costs && (step % args.cost_eval_step) == 0)
            cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
        if (g0 && args.step_states != nullptr)
            args.step_states[((size_t)b * (nsteps + 1) + step) * NP + i] = vecs[i];
        if (lane == 0) offs_b[step] = slot;
    };

    // ---- forward sweep ----------------------------------------------------------------------
    if (do_fwd) {
        if (jb == 0) {
            if (g0) {
                const double2 p = args.psi0[i];
                vecs[i] = p;
                states_b[i] = p;
            }
        } else {  // resume: state, slot counter and partial cost left by the previous segment
            slot = offs_b[jb];
            cost = args.cost_out[b];
            if (g0) vecs[i] = states_b[(size_t)slot * NP + i];
        }
        wave_sync();
        if (jb < je) {
            (void)substeps(jb, false);
            for (int k = 0; k < (RING > 1 ? RING - 1 : 1); ++k)
                if (jb + k < je) {
                    burst_lu(m0 + jb + k, FWD_T, k);
                    burst_q(m0 + jb + k, FWD_T, k);
                }
        }
        bool pend = false;
        double zre = 0, zim = 0;
        for (int step = jb; step < je; ++step) {
            const int nsub = substeps(step, false);
            const int set = RING > 1 ? (step - jb) % RING : 0;
            wait_sets(RING > 1 ? min(RING - 2, je - 1 - step) : 0);  // this step's operands have landed
            if (pend) states_b[(size_t)slot * NP + i] = make_double2(zre, zim);
            pend = false;
            wave_sync();
            if constexpr (RING > 1) {  // the set of step + RING - 1 into the buffers step - 1 has left
                const int nxt = step + RING - 1;
                if (nxt < je && !(QOCX_DBG_BITS(args.dbg) & 256)) {
                    burst_lu(m0 + nxt, FWD_T, (nxt - jb) % RING);
                    burst_q(m0 + nxt, FWD_T, (nxt - jb) % RING);
                }
            }
            const StepScalars sc = scalars(false, set);
            const double2* qset = at_set(qbuf, set);
            if (!(QOCX_DBG_BITS(args.dbg) & 8192))
                lds_to_regs<NB, false>(qset, at_set(lbuf, set), at_set(pbuf, set), r, sc.pm, lane, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are in registers: the buffer is free
            const bool more = RING == 1 && step + 1 < je && !(QOCX_DBG_BITS(args.dbg) & 256);
            if (more) burst_lu(m0 + step + 1, FWD_T);
            before_step(step);
            for (int sub = 0; sub < nsub; ++sub) {
                if ((size_t)slot + 1 >= cap) {
                    overflow = true;
                    break;
                }
                // z = Pi (Q psi): the lane at position i takes row perm[i] of the Q image
                if (QOCX_DBG_BITS(args.dbg) & 4096) {  // (timing experiment: no matrix-vector product)
                    const double2 e = vecs[i];
                    zre = e.x; zim = e.y;
                } else {
                    lds_matvec<NB, false, MVB>(qset, vecs, h * NP + sc.pm, h, zre, zim);
                }
                if (sub == nsub - 1 && more) {  // the Q image has been read for the last time
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    burst_q(m0 + step + 1, FWD_T);
                }
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) tri_solve<NB, true, false>(r.lre, r.lim, zre, zim, none);
                cscale(zre, zim, sc.dv);
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) tri_solve<NB, false, false>(r.lre, r.lim, zre, zim, none);
                wave_sync();
                vecs[i] = make_double2(zre, zim);  // (every lane group holds the same z: all of them store)
                wave_sync();
                ++slot;
                if (sub < nsub - 1) states_b[(size_t)slot * NP + i] = make_double2(zre, zim);
                else pend = true;  // goes out at the top of the next step, behind its wait
            }
            if (overflow) break;
        }
        if (overflow) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) atomicOr(args.status, 4);
            return;
        }
        if (pend) states_b[(size_t)slot * NP + i] = make_double2(zre, zim);
        wave_sync();
        if (je == nsteps) {
            before_step(nsteps);
            cost += eval_costs<NB>(args, false, true, vecs, nullptr, h, i);
            if (args.unit_adjoint && args.want_grad) unit_adjoint_scales<NB>(args, vecs, b, h, i);
            if (g0) args.final_out[(size_t)b * NP + i] = vecs[i];
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
        if (g0) {
            const double2 e = args.inj_bars[((size_t)b * args.inj_count + row) * NP + i];
            double2 l = lam[i];
            l.x += e.x;
            l.y += e.y;
            lam[i] = l;
        }
        wave_sync();
    };

    // ---- adjoint sweep ------------------------------------------------------------------------
    const bool unit = args.unit_adjoint != 0;
    int* offs_x = unit ? args.offs_x + (size_t)b * (nsteps + 1) : nullptr;
    if (je == nsteps && unit) {
        // lam = the target. The forward sweep may not have numbered the sub-steps yet: the xs slots are
        // counted down from the capacity and recorded per step in offs_x
        unit_adjoint_seed<NB>(args, lam, 0, 1, h, i);
        slot = (int)cap;
        wave_sync();
    } else if (je == nsteps) {
        if (!do_fwd) {  // final state of the forward segments
            slot = offs_b[nsteps];
            if (g0) vecs[i] = states_b[(size_t)slot * NP + i];
        }
        if (g0) lam[i] = make_double2(0, 0);
        wave_sync();
        // cotangent seeds on the final state: non-step costs, and step costs if the final step is a
        // cost step (schroedingerdiscrete.py:412-416 evaluates them before the loop ends)
        (void)eval_costs<NB>(args, (nsteps % args.cost_eval_step) == 0, true, vecs, lam, h, i);
        wave_sync();
        inject(nsteps);
    } else {  // resume the adjoint sweep below step je
        slot = unit ? offs_x[je] : offs_b[je];
        if (g0) lam[i] = args.lam_buf[(size_t)b * NP + i];
        wave_sync();
    }
    {
        sq_valid = false;
        if (jb < je) {
            (void)substeps(je - 1, true);
            if constexpr (RING > 1) {
                for (int k = 0; k < RING - 1; ++k)
                    if (je - 1 - k >= jb) {
                        burst_lu(m0 + je - 1 - k, ADJ_T, k);
                        burst_q(m0 + je - 1 - k, ADJ_T, k);
                    }
            } else {
                burst_lu(m0 + je - 1, ADJ_T);
            }
        }
        bool pend = false;
        double pxre = 0, pxim = 0;
        int pslot = 0, pstep = -1;  // (pstep: the step whose first xs slot is still to be recorded)
        for (int step = je - 1; step >= jb; --step) {
            const int nsub = substeps(step, true);
            const int it = je - 1 - step, set = RING > 1 ? it % RING : 0;
            wait_sets(RING > 1 ? min(RING - 2, step - jb) : 0);  // this step's LU image, 1 / U_kk, permutation (RING > 1: Q too)
            if (pend) xs_b[(size_t)pslot * NP + i] = make_double2(pxre, pxim);
            pend = false;
            if (pstep >= 0 && lane == 0) offs_x[pstep] = slot;
            pstep = -1;
            wave_sync();
            const bool fetch = !(QOCX_DBG_BITS(args.dbg) & 512);
            if constexpr (RING > 1) {
                const int nxt = step - (RING - 1);
                if (nxt >= jb && fetch) {
                    burst_lu(m0 + nxt, ADJ_T, (it + RING - 1) % RING);
                    burst_q(m0 + nxt, ADJ_T, (it + RING - 1) % RING);
                }
            }
            const StepScalars sc = scalars(true, set);
            const double2* qset = at_set(qbuf, set);
            if (!(QOCX_DBG_BITS(args.dbg) & 8192))
                lds_to_regs<NB, true>(qset, at_set(lbuf, set), at_set(pbuf, set), r, sc.pm, lane, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // RING = 1: the step's OWN Q image (read at the end of its first sub-step; the buffer has been
            // free since the previous step's product), then the LU image of the next step
            if (RING == 1 && fetch) burst_q(m0 + step, ADJ_T);
            const bool more = RING == 1 && step - 1 >= jb && fetch;
            if (more) burst_lu(m0 + step - 1, ADJ_T);
            bool first = RING == 1;
            for (int sub = nsub - 1; sub >= 0; --sub) {
                if (slot <= 0) {  // (unit adjoint: nobody has checked the capacity before)
                    overflow = true;
                    break;
                }
                --slot;
                if (pend) xs_b[(size_t)pslot * NP + i] = make_double2(pxre, pxim);
                pend = false;
                const double2 l0 = lam[i];
                double zre = l0.x, zim = l0.y;
                // P^H = U'^H D^H L^H Pi : U'^H a = lambda ; b = a / conj(U_kk) ; L^H v = b
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) tri_solve<NB, true, true>(r.lre, r.lim, zre, zim, none);
                cscale_conj(zre, zim, sc.dv);
                if (!(QOCX_DBG_BITS(args.dbg) & 2048)) tri_solve<NB, false, true>(r.lre, r.lim, zre, zim, none);
                // x = Pi^T v : x_i = v[position of row i]
                const double xre = __shfl(zre, sc.pm), xim = __shfl(zim, sc.pm);
                wave_sync();
                tmp[i] = make_double2(xre, xim);
                wave_sync();
                if (first) {  // younger than the Q pieces: the LU set of the next step, if it went out
                    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P::LU_SET) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    first = false;
                }
                // lambda = Q^H x ; the LDS image is that of Q^T (lane (h,i): Q[cc*H+h][i])
                double yre, yim;
                if (QOCX_DBG_BITS(args.dbg) & 4096) {
                    yre = xre; yim = xim;
                } else {
                    lds_matvec<NB, true, MVB>(qset, tmp, lane, h, yre, yim);
                }
                wave_sync();
                lam[i] = make_double2(yre, yim);
                wave_sync();
                pxre = xre; pxim = xim; pslot = slot;
                pend = true;
            }
            if (overflow) break;
            if (step != 0 && (step % args.cost_eval_step) == 0 && args.has_step_costs) {
                // step costs were evaluated on the state *before* evolving from `step`
                if (g0) vecs[i] = states_b[(size_t)slot * NP + i];
                wave_sync();
                (void)eval_costs<NB>(args, true, false, vecs, lam, h, i);
                wave_sync();
            }
            if (step != 0) inject(step);
            if (unit) pstep = step;  // (recorded at the top of the next step, behind its wait)
        }
        if (overflow) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) atomicOr(args.status, 4);
            return;
        }
        if (pend) xs_b[(size_t)pslot * NP + i] = make_double2(pxre, pxim);
        if (pstep >= 0 && lane == 0) offs_x[pstep] = slot;
    }
    if (jb > 0 && g0) args.lam_buf[(size_t)b * NP + i] = lam[i];
}

}  // namespace sweep1

bool sweep1_supports(int nb, int S) { return nb <= 2 && S == 1; }

// `pack`: seeds (waves) per workgroup, 1 or 2
void launch_sweep1(int nb, const SweepArgs& a, int batch, int pack, hipStream_t st) {
    SweepArgs b = a;
    b.batch = batch;
    pack = pack >= 2 ? 2 : 1;
    if (nb == 1) {
        constexpr int RING = 4;
        const int bytes = sweep1::Lds1<1, RING>::SEED_BYTES * pack;
        static bool attr_set = false;
        if (bytes > 48 * 1024 && !attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep1::sweep1_kernel<1, RING>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            attr_set = true;
        }
        hipLaunchKernelGGL((sweep1::sweep1_kernel<1, RING>), dim3((batch + pack - 1) / pack), dim3(64 * pack), bytes, st, b);
    } else if (a.ring2) {
        const int bytes = sweep1::Lds1<2, 2>::SEED_BYTES * pack;
        static bool attr_set = false;
        if (bytes > 48 * 1024 && !attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep1::sweep1_kernel<2, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            attr_set = true;
        }
        hipLaunchKernelGGL((sweep1::sweep1_kernel<2, 2>), dim3((batch + pack - 1) / pack), dim3(64 * pack), bytes, st, b);
    } else {
        const int bytes = sweep1::Lds1<2, 1>::SEED_BYTES * pack;
        static bool attr_set = false;
        if (bytes > 48 * 1024 && !attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep1::sweep1_kernel<2, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            attr_set = true;
        }
        hipLaunchKernelGGL((sweep1::sweep1_kernel<2, 1>), dim3((batch + pack - 1) / pack), dim3(64 * pack), bytes, st, b);
    }
}

}  // namespace qocx
