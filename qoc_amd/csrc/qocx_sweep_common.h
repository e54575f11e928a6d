// qocx_sweep_common.h - pieces shared by the two sweep kernels (qocx_kernels.hip: the column-chain
// sweep; qocx_sweep3.hip: the blocked-inverse sweep): state-cost evaluation, LDS-DMA helpers,
// lane-group sums.
#ifndef QOCX_SWEEP_COMMON_H
#define QOCX_SWEEP_COMMON_H

#include "qocx_wave.h"

namespace qocx {

// <t|psi> over lane group 0, result wave-uniform.
__device__ __forceinline__ void inner(const double2 t, const double2 p, bool active, double& re,
                                      double& im) {
#pragma clang fp contract(off)  // (the sweep kernels that share this code must round alike)
    double pr = active ? (t.x * p.x + t.y * p.y) : 0.0;  // conj(t) * p
    double pi = active ? (t.x * p.y - t.y * p.x) : 0.0;
    re = wave_sum(pr);
    im = wave_sum(pi);
}

// Evaluate the selected costs on the S states in `vecs` (LDS, [S][NP]). If lam != nullptr also
// adds dC/dRe + i dC/dIm into lam (LDS, [S][NP]). Formulas: qoc/standard/costs/
// targetstateinfidelity.py:52-61, forbidstates.py:64-81; cotangents SURVEY.md Appendix A.
template <int NB>
__device__ __forceinline__ double eval_costs(const SweepArgs& args, bool step_pass,
                                             bool final_pass, const double2* vecs, double2* lam,
                                             int h, int i) {
#pragma clang fp contract(off)
    constexpr int NP = Geo<NB>::NP;
    const int S = args.S;
    const bool act = (h == 0);
    double total = 0;
    for (int ci = 0; ci < args.cost_count; ++ci) {
        const DevCost c = args.costs[ci];
        const bool on = c.step_cost ? step_pass : final_pass;
        if (!on) continue;
        const double2* pool = args.cost_vectors + (size_t)c.vec_offset * NP;
        if (c.kind == QOCX_DEV_COST_COHERENT) {
            double tre = 0, tim = 0;
            for (int s = 0; s < S; ++s) {
                double r, m;
                inner(pool[s * NP + i], vecs[s * NP + i], act, r, m);
                tre += r;
                tim += m;
            }
            total += c.scale * (1.0 - (tre * tre + tim * tim) / ((double)S * S));
            if (lam != nullptr && act) {
                const double f = -2.0 * c.scale / ((double)S * S);
                for (int s = 0; s < S; ++s) {
                    const double2 t = pool[s * NP + i];
                    double2 l = lam[s * NP + i];
                    l.x += f * (tre * t.x - tim * t.y);
                    l.y += f * (tre * t.y + tim * t.x);
                    lam[s * NP + i] = l;
                }
            }
        } else if (c.kind == QOCX_DEV_COST_INCOHERENT) {
            double fid = 0;
            const double f = -2.0 * c.scale / (double)S;
            for (int s = 0; s < S; ++s) {
                double r, m;
                const double2 t = pool[s * NP + i];
                inner(t, vecs[s * NP + i], act, r, m);
                fid += r * r + m * m;
                if (lam != nullptr && act) {
                    double2 l = lam[s * NP + i];
                    l.x += f * (r * t.x - m * t.y);
                    l.y += f * (r * t.y + m * t.x);
                    lam[s * NP + i] = l;
                }
            }
            total += c.scale * (1.0 - fid / (double)S);
        } else {  // QOCX_DEV_COST_FORBID
            int base = 0;
            double acc = 0;
            for (int s = 0; s < S; ++s) {
                const int fs = args.cost_counts[c.cnt_offset + s];
                const double w = 1.0 / (double)fs;
                for (int f = 0; f < fs; ++f) {
                    double r, m;
                    const double2 t = pool[(size_t)(base + f) * NP + i];
                    inner(t, vecs[s * NP + i], act, r, m);
                    acc += w * (r * r + m * m);
                    if (lam != nullptr && act) {
                        const double g = 2.0 * c.scale * w;
                        double2 l = lam[s * NP + i];
                        l.x += g * (r * t.x - m * t.y);
                        l.y += g * (r * t.y + m * t.x);
                        lam[s * NP + i] = l;
                    }
                }
                base += fs;
            }
            total += c.scale * acc;
        }
    }
    if (lam != nullptr) wave_sync();
    return total;
}

// ---- separable final cost ("unit adjoint") ---------------------------------------------------
// When the ONLY cost is one TargetStateInfidelity (coherent or incoherent) on the final states, the
// cotangent of the final states is lam_s = c_s t_s: the target vectors times scalars that depend
// on the final states (eval_costs above: c = f (tre + i tim), resp. c_s = f (r_s + i m_s)). The
// adjoint recursion is linear in lam, so it can run on lam_s = t_s WITHOUT knowing the forward
// result - the classic GRAPE back-propagation of the target - and K3 applies c_s to x = P^-H lam'
// when it forms the gradient. The forward and the adjoint sweep then only meet in K3, and the
// pipeline runs them side by side (qocx_api.hip). args.unit_adjoint selects it; args.lam_scale
// [B][S] carries the scalars from the end of the forward sweep to K3.
template <int NB>
__device__ __forceinline__ void unit_adjoint_scales(const SweepArgs& args, const double2* vecs, int b,
                                                    int h, int i) {
#pragma clang fp contract(off)
    constexpr int NP = Geo<NB>::NP;
    const int S = args.S;
    const DevCost c = args.costs[0];
    const double2* pool = args.cost_vectors + (size_t)c.vec_offset * NP;
    const bool act = (h == 0);
    if (c.kind == QOCX_DEV_COST_COHERENT) {
        double tre = 0, tim = 0;
        for (int s = 0; s < S; ++s) {
            double r, m;
            inner(pool[s * NP + i], vecs[s * NP + i], act, r, m);
            tre += r;
            tim += m;
        }
        const double f = -2.0 * c.scale / ((double)S * S);
        if (lane_id() == 0)
            for (int s = 0; s < S; ++s) args.lam_scale[(size_t)b * S + s] = make_double2(f * tre, f * tim);
    } else {  // QOCX_DEV_COST_INCOHERENT
        const double f = -2.0 * c.scale / (double)S;
        for (int s = 0; s < S; ++s) {
            double r, m;
            inner(pool[s * NP + i], vecs[s * NP + i], act, r, m);
            if (lane_id() == 0) args.lam_scale[(size_t)b * S + s] = make_double2(f * r, f * m);
        }
    }
}

// lam_s = t_s for the states s0, s0 + W, ... (lane group 0 writes)
template <int NB>
__device__ __forceinline__ void unit_adjoint_seed(const SweepArgs& args, double2* lam, int s0, int W,
                                                  int h, int i) {
    constexpr int NP = Geo<NB>::NP;
    const double2* pool = args.cost_vectors + (size_t)args.costs[0].vec_offset * NP;
    if (h == 0)
        for (int s = s0; s < args.S; s += W) lam[s * NP + i] = pool[s * NP + i];
}

// One 16-byte-per-lane LDS-DMA: lane l's 16 bytes at `g` land at lds_base + 16*l.
__device__ __forceinline__ void dma16(const double2* g, double2* lds_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)g,
        (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

__device__ __forceinline__ void dma4(const int* g, int* lds_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)g,
        (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}


// sum of a value over the H lane groups that share a row (lanes i, i + NP, ...)
template <int NB>
__device__ __forceinline__ double sum_groups(double v) {
    if (Geo<NB>::H == 2) {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return make_f64((int)a[0], (int)b[0]) + make_f64((int)a[1], (int)b[1]);
    }
    if (Geo<NB>::H == 4) {
        // rows of 16 lanes: (g0 + g1) + (g2 + g3) by the two swap instructions of gfx950 - no trip
        // through the LDS crossbar (ds_bpermute), which was most of a 16 x 16 matrix-vector product;
        // the same pairs, so the same sums bit for bit
        unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const double s = make_f64((int)a[0], (int)b[0]) + make_f64((int)a[1], (int)b[1]);
        lo = (unsigned)__double2loint(s);
        hi = (unsigned)__double2hiint(s);
        auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return make_f64((int)c[0], (int)d[0]) + make_f64((int)c[1], (int)d[1]);
    }
#pragma unroll
    for (int d = Geo<NB>::NP; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}


}  // namespace qocx

#endif
