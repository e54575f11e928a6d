// qocx_sweep3.hip - K2, blocked form: the serial state sweep with the triangular solves turned
// into dense 16 x 16 block products (round 2; the column-chain form is sweep_kernel in
// qocx_kernels.hip and stays as the fallback and for A/B).
//
// Why. One propagator step of the sweep is psi' = U'^-1 D^-1 L^-1 Pi Q psi (qoc/standard/functions/
// expm.py:246-250 through the LU factors of P). As column substitutions that is 62 DEPENDENT
// stages per step (v_readlane broadcast + exec-masked FMA each): 7 600 cycles per step measured,
// 668 VALU + 348 SALU instructions, and the evaluation waits for it. With
// the inverses of the 16 x 16 diagonal blocks of L and U' at hand,
//
//     L^-1 = [ iL11        0  ]      y1 = iL11 b1,  t = b2 - L21 y1,  y2 = iL22 t
//            [ -iL22 L21 iL11  iL22 ]
//
// a triangular solve is three dense block products, the step seven dependent stages (one 32 x 32
// product with Q, six 16 x 16), no v_readlane, no exec masks.
//
// Who inverts. Not K1b (it is on the critical path of the factor phase): the sweep's workgroup
// has three wavefronts with fixed roles, in lock step, one workgroup barrier per system step:
//
//   wave 2, loader  : LDS-DMA of step t+3's LU image, 1/U_kk and row permutation into a ring of
//                     four LDS slots. The image is gathered row-PERMUTED (lane address from
//                     perm), so everything downstream works in pivot-position order with static
//                     addresses; pieces of 1 KiB land 1 040 B apart, which keeps the transposed
//                     reads of the adjoint at a 2-way bank conflict instead of 32-way.
//   wave 1, inverter: for step t+1, the inverses of the four diagonal blocks (iL11, iL22, iU11,
//                     iU22), one LANE PER COLUMN: X[i][r] = delta_ir - sum_{k<i} T[i][k] X[k][r]
//                     with the scalars T broadcast from the LDS image and X in registers - no
//                     cross-lane traffic at all; the U' blocks run through the same instruction
//                     stream with reversed indices (per-lane address sign).
//   wave 0, compute : the seven stages of step t, operands read from LDS inside the products
//                     (nothing staged in registers), vectors handed from stage to stage through
//                     LDS; Q comes straight from HBM into registers, fetched one step ahead
//                     (forward: row-permuted, coalesced; adjoint: the columns, L2-absorbed).
//
// The adjoint sweep uses the same blocks conjugate-transposed: x = Pi^T L^-H D^-H U'^-H lambda',
// lambda = Q^H x = (Pi Q)^H (Pi x).  Costs, cotangent injection, time segmentation and the layout
// of states / xs / offs in HBM are those of sweep_kernel, so K3 and the pipeline are unchanged.
#include "qocx_sweep_common.h"

namespace qocx {

template <int NB>
struct S3 {
    typedef Geo<NB> G;
    static constexpr int NP = G::NP, H = G::H, CPL = G::CPL, MAT = G::MAT;
    static constexpr int CPP = 64 / NP;        // image columns per 1-KiB DMA piece
    static constexpr int PIECES = MAT / 64;    // pieces per image
    static constexpr int PST = 1024 + 16;      // LDS stride of a piece (bytes)
    static constexpr int IMG = PIECES * PST;
    static constexpr int RING = 4;
    static constexpr int SLOT_D = IMG;           // 64 x 16 B: lane l -> 1/U_kk of position l % NP
    static constexpr int SLOT_P = SLOT_D + 1024; // 64 x 4 B : lane l -> perm[l % NP]
    static constexpr int SLOT = SLOT_P + 256;
    static constexpr int NBLK = 2 * NB;          // iL11 [iL22] iU11 [iU22]
    static constexpr int BLK = 16 * 17 * 16;     // 16 x 16 complex, row-major, pitch 17
    static constexpr int QRING = 3;              // Q images (position-ordered rows, padded pieces)
    static constexpr int Q_OFF = RING * SLOT;
    static constexpr int INV_OFF = Q_OFF + QRING * IMG;
    static constexpr int INV = NBLK * BLK;
    static constexpr int SCR_OFF = INV_OFF + 2 * INV;   // five 16-vectors handed between stages
    static constexpr int SCR = (5 * 16 + NP) * 16;      // + one NP-vector (adjoint: v)
    static constexpr int VEC_OFF = SCR_OFF + SCR;       // [S][NP] states | lambda | b = Pi Q psi
    static constexpr int DMA_OPS = 2 * PIECES + 2;  // LU image + 1/U_kk + perm, Q image
    static int bytes(int S) { return VEC_OFF + 3 * S * NP * 16; }
    // byte offset of element (row, col) of a padded column-major image
    __host__ __device__ static constexpr int img_off(int col, int row) {
        return (col / CPP) * PST + (col % CPP) * (NP * 16) + row * 16;
    }
};

// LDS hand-off between the waves of the workgroup: my LDS writes are done, then the barrier.
// (No __syncthreads(): it would also drain vmcnt, i.e. the loader's fetches still in flight.)
__device__ __forceinline__ void s3_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// y[i16] = sum_c M[i16][c] v[c] (CT: M^H) over a 16 x 16 block (row-major, pitch 17); lane (h, i)
// takes the columns c = j H + h. Every lane group ends with the full sum.
template <int NB, bool CT>
__device__ __forceinline__ void blk_mv(const char* blk, const double2* vec, int h, int i16,
                                       double& yre, double& yim) {
    constexpr int H = Geo<NB>::H, NT = 16 / H;
    double ar = 0, ai = 0;
    double2 m[NT], v[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int c = j * H + h;
        m[j] = *reinterpret_cast<const double2*>(blk + (CT ? (c * 17 + i16) : (i16 * 17 + c)) * 16);
        v[j] = vec[c];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const double mi = CT ? -m[j].y : m[j].y;
        ar = fma(-mi, v[j].y, fma(m[j].x, v[j].x, ar));
        ai = fma(mi, v[j].x, fma(m[j].x, v[j].y, ai));
    }
    yre = sum_groups<NB>(ar);
    yim = sum_groups<NB>(ai);
}

// NB = 2: product with an off-diagonal 16 x 16 quadrant of the (padded, position-ordered) LU
// image. !CT: y[i16] = sum_k E[rowbase + i16][colbase + k] v[k];  CT: y[i16] = sum_p
// conj(E[rowbase + p][colbase + i16]) v[p];  k, p = 2 j + h.
template <bool CT>
__device__ __forceinline__ void img_mv(const char* img, int rowbase, int colbase, const double2* vec,
                                       int h, int i16, double& yre, double& yim) {
    typedef S3<2> L;
    double ar = 0, ai = 0;
    double2 m[8], v[8];
    const int base = CT ? L::img_off(colbase + i16, rowbase + h)
                        : ((colbase / 2) * L::PST + h * 512 + (rowbase + i16) * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        m[j] = *reinterpret_cast<const double2*>(img + base + (CT ? j * 32 : j * L::PST));
        v[j] = vec[2 * j + h];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double mi = CT ? -m[j].y : m[j].y;
        ar = fma(-mi, v[j].y, fma(m[j].x, v[j].x, ar));
        ai = fma(mi, v[j].x, fma(m[j].x, v[j].y, ai));
    }
    yre = sum_groups<2>(ar);
    yim = sum_groups<2>(ai);
}

__device__ __forceinline__ void cmul(double& re, double& im, double2 d, bool conj_d) {
    const double di = conj_d ? -d.y : d.y;
    const double t = re * d.x - im * di;
    im = re * di + im * d.x;
    re = t;
}

// An inverter's step: inverses of the diagonal 16 x 16 blocks of ONE unit-triangular factor in
// `slot` (position order) -> `inv`. UPPER = false: iL11 [iL22] (blocks 0 .. NB-1); true: iU11
// [iU22] (blocks NB ..), through the same recurrence with reversed indices. Lane = 32 g + 2 r + e:
// diagonal block g, column r of the inverse, half e of every row sum:
//     X[i][r] = delta_ir - sum_{k<i} T[i][k] X[k][r],   T = the block (reversed if UPPER),
// lane e adds the terms with k = e (mod 2), the pair meets by one DPP exchange per row. The
// scalars T come from LDS (one address per 32 lanes) at compile-time offsets: a wave that is alone
// on its SIMD issues one v_fma_f64 per 8 cycles, so what counts is the FMAs per lane (240) and
// nothing beside them.
template <int NB, bool UPPER>
__device__ __forceinline__ void invert_blocks(const char* slot, char* inv, int lane) {
    typedef S3<NB> L;
    const int g = (NB == 2) ? (lane >> 5) : 0;  // NB = 1: the upper half of the wave duplicates
    const int r = (lane >> 1) & 15, e = lane & 1;
    const int d0 = 16 * g;
    auto off = [](int i, int k) constexpr {
        return (k / L::CPP) * L::PST + (k % L::CPP) * (L::NP * 16) + i * 16;
    };
    constexpr int MAXOFF = (14 / L::CPP) * L::PST + (14 % L::CPP) * (L::NP * 16) + 15 * 16;
    // !UPPER: T[i][k] at img_off(d0 + k, d0 + i)      = img_off(d0, d0) + off(i, k)
    //  UPPER: T[i][k] at img_off(c0 - k, c0 - i), c0 = d0 + 15:  img_off(c0, c0) - off(i, k)
    //         = (img_off(c0, c0) - MAXOFF) + (MAXOFF - off(i, k))   (both terms >= 0)
    int boff = UPPER ? L::img_off(d0 + 15, d0 + 15) - MAXOFF : L::img_off(d0, d0);
    char* dst = inv + ((UPPER ? NB : 0) + g) * L::BLK +
                (UPPER ? ((15 * 17 + 15) - r) * 16 : r * 16);
    double xre[16], xim[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        xre[k] = (k == r) ? 1.0 : 0.0;
        xim[k] = 0.0;
    }
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        // Row i's scalars may be fetched once row i-2 is done, i.e. while row i-1 is being
        // accumulated, and no earlier: the lane offset passes through an asm that reads x[i-2].
        // Without it hipcc hoists all the LDS reads to the top and spills their results.
        if (i >= 2) asm volatile("" : "+v"(boff) : "v"(xre[i - 2]), "v"(xim[i - 2]));
        const char* src = slot + boff;
        constexpr int KMAX = 8;  // terms per lane in the longest row
        double2 t[KMAX];
        double ar = 0.0, ai = 0.0;
#pragma unroll
        for (int kk = 0; kk < (i + 1) / 2; ++kk) {
            // lane e takes k = 2 kk + e (the last one may be k = i: its T is never used, x[i] = 0
            // there has not been written yet and the product is masked below)
            const int k0 = 2 * kk;
            const int o0 = UPPER ? MAXOFF - off(i, k0) : off(i, k0);
            const int o1 = (k0 + 1 < i) ? (UPPER ? MAXOFF - off(i, k0 + 1) : off(i, k0 + 1)) : o0;
            t[kk] = *reinterpret_cast<const double2*>(src + (e ? o1 : o0));
        }
#pragma unroll
        for (int kk = 0; kk < (i + 1) / 2; ++kk) {
            const int k0 = 2 * kk;
            const bool both = (k0 + 1 < i);  // compile time
            const double xr = both ? (e ? xre[k0 + 1] : xre[k0]) : (e ? 0.0 : xre[k0]);
            const double xi = both ? (e ? xim[k0 + 1] : xim[k0]) : (e ? 0.0 : xim[k0]);
            ar = fma(t[kk].y, xi, fma(-t[kk].x, xr, ar));
            ai = fma(-t[kk].y, xr, fma(-t[kk].x, xi, ai));
        }
        // the pair's sum (quad_perm [1,0,3,2]); a + b == b + a bit for bit, both lanes agree
        ar += dpp_f64<0xB1>(ar);
        ai += dpp_f64<0xB1>(ai);
        xre[i] = ((i == r) ? 1.0 : 0.0) + ar;
        xim[i] = ai;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        *reinterpret_cast<double2*>(dst + (UPPER ? -i : i) * (17 * 16)) = make_double2(xre[i], xim[i]);
}

// What every role needs to walk the steps of a launch in lock step.
struct S3Walk {
    size_t m0;      // first matrix of the seed
    size_t cap;     // sub-step slots per seed
    int jb, je, T;  // steps [jb, je) of this launch
    bool do_fwd, do_bwd;
    int nsteps;
};

template <int NB>
__device__ __forceinline__ char* s3_slot(char* smem, int t) {
    return smem + (t & (S3<NB>::RING - 1)) * S3<NB>::SLOT;
}
template <int NB>
__device__ __forceinline__ char* s3_inv(char* smem, int t) {
    return smem + S3<NB>::INV_OFF + (t & 1) * S3<NB>::INV;
}
// Sub-steps (2^s) of the t-th step of a pass, for every role alike. A scalar load per step would
// put a dependent trip to memory in front of every step (measured: most of 1.3 us per step with
// nothing else to do); instead lane l keeps the squaring count of step t0 + l of the pass and the
// wave refills the 64 of them every 64 steps.
struct S3Subs {
    int sv;
    __device__ __forceinline__ int get(const SweepArgs& args, const S3Walk& w, bool adjoint, int t) {
        if ((t & 63) == 0) {
            const int tt = min(t + lane_id(), w.T - 1);
            sv = args.s_arr[w.m0 + (adjoint ? w.je - 1 - tt : w.jb + tt)];
        }
        const int s = __builtin_amdgcn_readlane(sv, t & 63);
        return 1 << step_squarings(s);
    }
};

// The three roles run the SAME sequence of workgroup barriers:
//   forward  : P0 (slots 0, 1 landed) | P1 (blocks of step 0 inverted, Q of step 0 on its way) |
//              one per step; a launch whose sub-step capacity overflows leaves at the same step
//              in every role (the test depends on slot counts only);
//   adjoint  : A0 (forward pass has left the ring) | P0 | P1 | one per step.
// They are separate functions with loops of their own so that the registers of one role are not
// live in another (one loop nest with role branches inside made hipcc keep the loop invariants of
// all three roles alive at once: 256 VGPRs + 256 AGPRs + scratch).

// ---- wave 2: loader -------------------------------------------------------------------------
template <int NB, bool STAMP>
__device__ __forceinline__ void s3_loader(const SweepArgs& args, const S3Walk& w, char* smem,
                                          int slot) {
    StampClock<STAMP> clk;
    clk.start();
    typedef Geo<NB> G;
    typedef S3<NB> L;
    constexpr int NP = G::NP, MAT = G::MAT;
    const int lane = lane_id(), i = lane % NP;
    int pm_cur = 0;
    auto load_perm = [&](int step) { return args.perm[(w.m0 + step) * NP + i]; };
    // One fetch = [row permutation of the NEXT fetch -> register] + DMA_OPS pieces: the LU image,
    // 1/U_kk and perm of `step` into LU ring slot t, the Q image of `qstep` into Q ring slot tq.
    // Both images are gathered with their rows in pivot-position order; `qstep` runs one step
    // behind `step` (the inverter needs LU one step before the compute wave needs Q), so its
    // permutation is the one the previous fetch used.
    int pm_prev = 0;
    auto fetch = [&](int step, int t, int next_step, int qstep, int tq) {
        const int pm_next = load_perm(next_step);
        char* sl = s3_slot<NB>(smem, t);
        char* ql = smem + L::Q_OFF + (tq % L::QRING) * L::IMG;
        if (!(QOCX_DBG_BITS(args.dbg) & 4)) {
            const double2* src = args.lu_img + (w.m0 + step) * MAT + (size_t)(lane / NP) * NP +
                                 min(max(pm_cur, 0), NP - 1);
#pragma unroll
            for (int j = 0; j < L::PIECES; ++j)
                dma16(src + (size_t)j * L::CPP * NP, reinterpret_cast<double2*>(sl + j * L::PST));
            dma16(args.dinv + (w.m0 + step) * NP + i, reinterpret_cast<double2*>(sl + L::SLOT_D));
            dma4(args.perm + (w.m0 + step) * NP + i, reinterpret_cast<int*>(sl + L::SLOT_P));
        }
        if (!(QOCX_DBG_BITS(args.dbg) & 8) && qstep >= w.jb && qstep < w.je) {
            const double2* src = args.q_img + (w.m0 + qstep) * MAT + (size_t)(lane / NP) * NP +
                                 min(max(pm_prev, 0), NP - 1);
#pragma unroll
            for (int j = 0; j < L::PIECES; ++j)
                dma16(src + (size_t)j * L::CPP * NP, reinterpret_cast<double2*>(ql + j * L::PST));
        }
        pm_prev = pm_cur;
        pm_cur = pm_next;
    };
    auto pass = [&](bool adjoint) {
        // (unit adjoint behind a forward pass in the same launch - one time segment, je = nsteps:
        // the x slots are counted down from the capacity, as in the compute wave)
        if (adjoint && args.unit_adjoint && w.do_fwd) slot = (int)w.cap;
        const int first = adjoint ? w.je - 1 : w.jb, d = adjoint ? -1 : 1;
        auto clampstep = [&](int st) { return min(max(st, w.jb), w.je - 1); };
        // prologue: LU of steps 0, 1, 2 and Q of steps 0, 1 (pass order); all of it lands here
        pm_cur = load_perm(first);
        pm_prev = pm_cur;
        fetch(first, 0, clampstep(first + d), w.jb - 1, 0);
        if (w.T > 1) fetch(first + d, 1, clampstep(first + 2 * d), first, 0);
        else fetch(first, 1, first, first, 0);  // (T = 1: only the Q image matters)
        if (w.T > 2) fetch(first + 2 * d, 2, clampstep(first + 3 * d), first + d, 1);
        else if (w.T > 1) fetch(first + d, 2, first + d, first + d, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s3_barrier();  // P0
        s3_barrier();  // P1
        S3Subs subs;
        for (int t = 0; t < w.T; ++t) {
            const int step = first + d * t;
            const int nsub = subs.get(args, w, adjoint, t);
            if (!adjoint && (size_t)slot + nsub >= w.cap) return false;
            if (adjoint && args.unit_adjoint && slot - nsub < 0) return false;  // (as the compute wave)
            clk.lap(0);
            if (t + 3 < w.T) {
                // LU of step t+3, Q of step t+2: each has two iterations to land
                fetch(step + 3 * d, t + 3, clampstep(step + 4 * d), step + 2 * d, t + 2);
                clk.lap(1);             // issue
                // everything but what this iteration issued has landed
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::DMA_OPS + 1) : "memory");
            } else {
                if (t + 2 < w.T)        // no LU left to fetch (slot t+3 is free), Q of step t+2
                    fetch(step + 2 * d, t + 3, step + 2 * d, step + 2 * d, t + 2);
                clk.lap(1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            clk.lap(2);                 // landing
            slot += adjoint ? -nsub : nsub;
            s3_barrier();
            clk.lap(3);                 // barrier
        }
        return true;
    };
    if (w.do_fwd)
        if (!pass(false)) return;
    if (w.do_bwd) {
        s3_barrier();  // A0
        (void)pass(true);
    }
    clk.finish(args.stamps, 4, 2);
}

// ---- wave 1: inverter -----------------------------------------------------------------------
template <int NB, bool STAMP, bool UPPER>
__device__ __forceinline__ void s3_inverter(const SweepArgs& args, const S3Walk& w, char* smem,
                                            int slot) {
    const int lane = lane_id();
    StampClock<STAMP> clk;
    clk.start();
    auto pass = [&](bool adjoint) {
        // (unit adjoint behind a forward pass in the same launch - one time segment, je = nsteps:
        // the x slots are counted down from the capacity, as in the compute wave)
        if (adjoint && args.unit_adjoint && w.do_fwd) slot = (int)w.cap;
        s3_barrier();  // P0
        if (!(QOCX_DBG_BITS(args.dbg) & 1)) invert_blocks<NB, UPPER>(s3_slot<NB>(smem, 0), s3_inv<NB>(smem, 0), lane);
        s3_barrier();  // P1
        S3Subs subs;
        for (int t = 0; t < w.T; ++t) {
            const int nsub = subs.get(args, w, adjoint, t);
            if (!adjoint && (size_t)slot + nsub >= w.cap) return false;
            if (adjoint && args.unit_adjoint && slot - nsub < 0) return false;
            clk.lap(0);
            if (t + 1 < w.T && !(QOCX_DBG_BITS(args.dbg) & 1))
                invert_blocks<NB, UPPER>(s3_slot<NB>(smem, t + 1), s3_inv<NB>(smem, t + 1), lane);
            clk.lap(1);                 // inversion
            slot += adjoint ? -nsub : nsub;
            s3_barrier();
            clk.lap(2);                 // barrier
        }
        return true;
    };
    if (w.do_fwd)
        if (!pass(false)) return;
    if (w.do_bwd) {
        s3_barrier();  // A0
        (void)pass(true);
    }
    clk.finish(args.stamps, 4, UPPER ? 3 : 1);
}

// ---- wave 0: compute ------------------------------------------------------------------------
template <int NB, bool STAMP>
__device__ __forceinline__ void s3_compute(const SweepArgs& args, const S3Walk& w, char* smem,
                                           int slot) {
    StampClock<STAMP> clk;
    clk.start();
    typedef Geo<NB> G;
    typedef S3<NB> L;
    constexpr int NP = G::NP, H = G::H, CPL = G::CPL;
    const int lane = lane_id(), i = lane % NP, h = lane / NP, i16 = lane & 15;
    const bool g0 = (h == 0);
    const int S = args.S;
    double2* scr = reinterpret_cast<double2*>(smem + L::SCR_OFF);
    double2 *va = scr, *vt = scr + 16, *va2 = scr + 32, *vx2 = scr + 48, *vw = scr + 64;
    double2* vv = scr + 80;  // NP entries
    double2* vecs = reinterpret_cast<double2*>(smem + L::VEC_OFF);
    double2* lam = vecs + S * NP;
    double2* bvec = lam + S * NP;
    const int b = blockIdx.x;
    const int nsteps = w.nsteps, jb = w.jb, je = w.je, T = w.T;
    double2* states_b = args.states + (size_t)b * w.cap * S * NP;
    double2* xs_b = args.xs + (size_t)b * w.cap * S * NP;
    int* offs_b = args.offs + (size_t)b * (nsteps + 1);

    auto q_of = [&](int t) { return smem + L::Q_OFF + (t % L::QRING) * L::IMG; };

    // x = U'^-1 D^-1 L^-1 b (b = bvec[s], position order)
    auto solve_forward = [&](const char* slot_, const char* inv, int s, double& xre, double& xim) {
        const double2* dsl = reinterpret_cast<const double2*>(slot_ + L::SLOT_D);
        const double2* bs = bvec + s * NP;
        if constexpr (NB == 1) {
            double yre, yim;
            blk_mv<1, false>(inv, bs, h, i16, yre, yim);
            cmul(yre, yim, dsl[i16], false);
            wave_sync();
            va[i16] = make_double2(yre, yim);
            wave_sync();
            blk_mv<1, false>(inv + L::BLK, va, h, i16, xre, xim);
        } else {
            double y1r, y1i, tr, ti, y2r, y2i, x2r, x2i, wr, wi, x1r, x1i;
            blk_mv<2, false>(inv, bs, h, i16, y1r, y1i);                  // y1 = iL11 b1
            wave_sync();
            va[i16] = make_double2(y1r, y1i);
            wave_sync();
            img_mv<false>(slot_, 16, 0, va, h, i16, tr, ti);              // L21 y1
            const double2 b2 = bs[16 + i16];
            tr = b2.x - tr;
            ti = b2.y - ti;
            wave_sync();
            vt[i16] = make_double2(tr, ti);
            wave_sync();
            blk_mv<2, false>(inv + L::BLK, vt, h, i16, y2r, y2i);         // y2 = iL22 t
            cmul(y1r, y1i, dsl[i16], false);                              // D^-1
            cmul(y2r, y2i, dsl[16 + i16], false);
            wave_sync();
            va2[i16] = make_double2(y2r, y2i);
            wave_sync();
            blk_mv<2, false>(inv + 3 * L::BLK, va2, h, i16, x2r, x2i);    // x2 = iU22 y2'
            wave_sync();
            vx2[i16] = make_double2(x2r, x2i);
            wave_sync();
            img_mv<false>(slot_, 0, 16, vx2, h, i16, wr, wi);             // U12 x2
            wr = y1r - wr;
            wi = y1i - wi;
            wave_sync();
            vw[i16] = make_double2(wr, wi);
            wave_sync();
            blk_mv<2, false>(inv + 2 * L::BLK, vw, h, i16, x1r, x1i);     // x1 = iU11 w
            xre = (i < 16) ? x1r : x2r;
            xim = (i < 16) ? x1i : x2i;
        }
    };
    // v = L^-H D^-H U'^-H lambda (position order)
    auto solve_adjoint = [&](const char* slot_, const char* inv, int s, double& vre, double& vim) {
        const double2* dsl = reinterpret_cast<const double2*>(slot_ + L::SLOT_D);
        const double2* ls = lam + s * NP;
        if constexpr (NB == 1) {
            double are, aim;
            blk_mv<1, true>(inv + L::BLK, ls, h, i16, are, aim);
            cmul(are, aim, dsl[i16], true);
            wave_sync();
            va[i16] = make_double2(are, aim);
            wave_sync();
            blk_mv<1, true>(inv, va, h, i16, vre, vim);
        } else {
            double a1r, a1i, tr, ti, a2r, a2i, v2r, v2i, wr, wi, v1r, v1i;
            blk_mv<2, true>(inv + 2 * L::BLK, ls, h, i16, a1r, a1i);      // a1 = iU11^H lambda1
            wave_sync();
            va[i16] = make_double2(a1r, a1i);
            wave_sync();
            img_mv<true>(slot_, 0, 16, va, h, i16, tr, ti);               // U12^H a1
            const double2 l2 = ls[16 + i16];
            tr = l2.x - tr;
            ti = l2.y - ti;
            wave_sync();
            vt[i16] = make_double2(tr, ti);
            wave_sync();
            blk_mv<2, true>(inv + 3 * L::BLK, vt, h, i16, a2r, a2i);      // a2 = iU22^H t
            cmul(a1r, a1i, dsl[i16], true);                               // D^-H
            cmul(a2r, a2i, dsl[16 + i16], true);
            wave_sync();
            va2[i16] = make_double2(a2r, a2i);
            wave_sync();
            blk_mv<2, true>(inv + L::BLK, va2, h, i16, v2r, v2i);         // v2 = iL22^H a2'
            wave_sync();
            vx2[i16] = make_double2(v2r, v2i);
            wave_sync();
            img_mv<true>(slot_, 16, 0, vx2, h, i16, wr, wi);              // L21^H v2
            wr = a1r - wr;
            wi = a1i - wi;
            wave_sync();
            vw[i16] = make_double2(wr, wi);
            wave_sync();
            blk_mv<2, true>(inv, vw, h, i16, v1r, v1i);                   // v1 = iL11^H w
            vre = (i < 16) ? v1r : v2r;
            vim = (i < 16) ? v1i : v2i;
        }
    };

    double cost = 0;
    // ============================== forward sweep ==============================================
    if (w.do_fwd) {
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
        s3_barrier();  // P0
        s3_barrier();  // P1
        bool overflow = false;
        S3Subs subs;
        const int ces = args.cost_eval_step;
        int cost_phase = jb % ces;  // step % ces, kept by counting
        for (int t = 0; t < T; ++t) {
            const int step = jb + t;
            const int nsub = subs.get(args, w, false, t);
            const bool cost_step = (cost_phase == 0) && step != 0 && args.has_step_costs;
            cost_phase = (cost_phase + 1 == ces) ? 0 : cost_phase + 1;
            if ((size_t)slot + nsub >= w.cap) {
                overflow = true;
                break;
            }
            // step costs / bookkeeping on the states before evolving from `step`
            if (cost_step) cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
            if (g0 && args.step_states != nullptr)
                for (int s = 0; s < S; ++s)
                    args.step_states[(((size_t)b * (nsteps + 1) + step) * S + s) * NP + i] =
                        vecs[s * NP + i];
            if (lane == 0) offs_b[step] = slot;
            const char* sl = s3_slot<NB>(smem, t);
            const char* iv = s3_inv<NB>(smem, t);
            clk.lap(0);                 // bookkeeping
            for (int sub = 0; sub < nsub; ++sub) {
                for (int s = 0; s < S; ++s) {  // b = (Pi Q) psi
                    const double2* ps = vecs + s * NP;
                    // lane (h, i): row i, columns c = j H + h: the image read piece by piece
                    const char* qrow = q_of(t) + lane * 16;
                    double ar = 0, ai = 0, br = 0, bi = 0;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const double2 q = *reinterpret_cast<const double2*>(
                            qrow + (j / (L::CPP / H)) * L::PST + (j % (L::CPP / H)) * (H * NP * 16));
                        const double2 v = ps[j * H + h];
                        if (j & 1) {
                            br = fma(-q.y, v.y, fma(q.x, v.x, br));
                            bi = fma(q.y, v.x, fma(q.x, v.y, bi));
                        } else {
                            ar = fma(-q.y, v.y, fma(q.x, v.x, ar));
                            ai = fma(q.y, v.x, fma(q.x, v.y, ai));
                        }
                    }
                    ar += br;
                    ai += bi;
                    ar = sum_groups<NB>(ar);
                    ai = sum_groups<NB>(ai);
                    bvec[s * NP + i] = make_double2(ar, ai);
                }
                wave_sync();
                clk.lap(1);             // b = (Pi Q) psi
                for (int s = 0; s < S; ++s) {
                    double xre = bvec[s * NP + i].x, xim = bvec[s * NP + i].y;
                    if (!(QOCX_DBG_BITS(args.dbg) & 2)) solve_forward(sl, iv, s, xre, xim);
                    wave_sync();
                    const double2 p = make_double2(xre, xim);
                    vecs[s * NP + i] = p;
                    states_b[((size_t)(slot + sub + 1) * S + s) * NP + i] = p;
                    wave_sync();
                }
                clk.lap(3);             // solves + stores
            }
            slot += nsub;
            s3_barrier();
            clk.lap(4);                 // barrier
        }
        if (overflow) {
            if (lane == 0) atomicOr(args.status, 4);
            return;
        }
        clk.lap(5);
        if (je == nsteps) {
            if (args.has_step_costs && nsteps != 0 && (nsteps % args.cost_eval_step) == 0)
                cost += eval_costs<NB>(args, true, false, vecs, nullptr, h, i);
            if (g0 && args.step_states != nullptr)
                for (int s = 0; s < S; ++s)
                    args.step_states[(((size_t)b * (nsteps + 1) + nsteps) * S + s) * NP + i] =
                        vecs[s * NP + i];
            if (lane == 0) offs_b[nsteps] = slot;
            cost += eval_costs<NB>(args, false, true, vecs, nullptr, h, i);
            if (args.unit_adjoint && args.want_grad) unit_adjoint_scales<NB>(args, vecs, b, h, i);
            if (g0)
                for (int s = 0; s < S; ++s)
                    args.final_out[((size_t)b * S + s) * NP + i] = vecs[s * NP + i];
        } else if (lane == 0) {
            offs_b[je] = slot;  // the next segment resumes from here
        }
        if (lane == 0) args.cost_out[b] = cost;
    }
    if (!w.do_bwd) {
        clk.finish(args.stamps, 4, 0);
        return;
    }

    // ============================== adjoint sweep ==============================================
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
    // unit adjoint (qocx_sweep_common.h): lambda = the targets, the x slots are counted down from
    // the capacity and recorded per step in offs_x - the sweep does not need the forward sweep
    const bool unit = args.unit_adjoint != 0;
    int* offs_x = unit ? args.offs_x + (size_t)b * (nsteps + 1) : nullptr;
    if (unit && w.do_fwd) slot = (je == nsteps) ? (int)w.cap : offs_x[je];  // (phase 3: the roles agree, see the kernel)
    if (je == nsteps && unit) {
        unit_adjoint_seed<NB>(args, lam, 0, 1, h, i);
        wave_sync();
    } else if (je == nsteps) {
        if (!w.do_fwd)
            for (int s = 0; s < S; ++s)
                if (g0) vecs[s * NP + i] = states_b[((size_t)slot * S + s) * NP + i];
        for (int s = 0; s < S; ++s)
            if (g0) lam[s * NP + i] = make_double2(0, 0);
        wave_sync();
        // cotangent seeds on the final states: non-step costs, and step costs if the final
        // step is a cost step (schroedingerdiscrete.py:412-416)
        (void)eval_costs<NB>(args, (nsteps % args.cost_eval_step) == 0, true, vecs, lam, h, i);
        inject(nsteps);
    } else {  // resume the adjoint sweep below step je
        for (int s = 0; s < S; ++s)
            if (g0) lam[s * NP + i] = args.lam_buf[((size_t)b * S + s) * NP + i];
        wave_sync();
    }
    s3_barrier();  // A0
    s3_barrier();  // P0
    s3_barrier();  // P1
    S3Subs subs_adj;
    const int ces_adj = args.cost_eval_step;
    int cost_phase_adj = (je - 1) % ces_adj;
    for (int t = 0; t < T; ++t) {
        const int step = je - 1 - t;
        const int nsub = subs_adj.get(args, w, true, t);
        const bool cost_step = (cost_phase_adj == 0) && step != 0 && args.has_step_costs;
        cost_phase_adj = (cost_phase_adj == 0) ? ces_adj - 1 : cost_phase_adj - 1;
        if (unit && slot - nsub < 0) {  // (unit adjoint: nobody has checked the capacity before)
            if (lane == 0) atomicOr(args.status, 4);
            return;
        }
        const char* sl = s3_slot<NB>(smem, t);
        const char* iv = s3_inv<NB>(smem, t);
        const int* pslot = reinterpret_cast<const int*>(sl + L::SLOT_P);
        const int prow = min(max(pslot[i], 0), NP - 1);  // row of P at position i
        clk.lap(0);
        for (int sub = nsub - 1; sub >= 0; --sub) {
            const int sub_slot = slot - nsub + sub;
            for (int s = 0; s < S; ++s) {
                double vre = lam[s * NP + i].x, vim = lam[s * NP + i].y;
                if (!(QOCX_DBG_BITS(args.dbg) & 2)) solve_adjoint(sl, iv, s, vre, vim);
                clk.lap(3);             // solves
                // x = Pi^T v goes to K3 in the original row order; lambda = (Pi Q)^H v
                if (g0) xs_b[((size_t)sub_slot * S + s) * NP + prow] = make_double2(vre, vim);
                wave_sync();
                vv[i] = make_double2(vre, vim);
                wave_sync();
                // lambda_i = sum_p conj(Qp[p][i]) v[p], p = j H + h: column i of the image
                const char* qcol = q_of(t) + L::img_off(i, h);
                double ar = 0, ai = 0, br = 0, bi = 0;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const double2 q = *reinterpret_cast<const double2*>(qcol + j * H * 16);
                    const double2 v = vv[j * H + h];
                    if (j & 1) {
                        br = fma(q.y, v.y, fma(q.x, v.x, br));
                        bi = fma(-q.y, v.x, fma(q.x, v.y, bi));
                    } else {
                        ar = fma(q.y, v.y, fma(q.x, v.x, ar));
                        ai = fma(-q.y, v.x, fma(q.x, v.y, ai));
                    }
                }
                ar += br;
                ai += bi;
                ar = sum_groups<NB>(ar);
                ai = sum_groups<NB>(ai);
                wave_sync();
                lam[s * NP + i] = make_double2(ar, ai);
                wave_sync();
                clk.lap(1);             // lambda = (Pi Q)^H v (waits for Q)
            }
        }
        clk.lap(2);
        if (cost_step) {
            // step costs were evaluated on the states *before* evolving from `step`
            if (g0)
                for (int s = 0; s < S; ++s)
                    vecs[s * NP + i] = states_b[((size_t)(slot - nsub) * S + s) * NP + i];
            wave_sync();
            (void)eval_costs<NB>(args, true, false, vecs, lam, h, i);
        }
        if (step != 0) inject(step);
        slot -= nsub;
        if (unit && lane == 0) offs_x[step] = slot;
        clk.lap(0);
        s3_barrier();
        clk.lap(4);
    }
    if (jb > 0 && g0)
        for (int s = 0; s < S; ++s)
            args.lam_buf[((size_t)b * S + s) * NP + i] = lam[s * NP + i];
    clk.finish(args.stamps, 4, 0);
}

template <int NB, bool STAMP>
__global__ __launch_bounds__(256, 2) void sweep3_kernel(SweepArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_s_setprio(3);  // the serial chain of the evaluation goes first on its SIMDs
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0 compute, 1 invert L, 2 load, 3 invert U'
    S3Walk w;
    w.m0 = (size_t)blockIdx.x * args.nsteps;
    w.cap = args.slot_cap;
    w.jb = args.j_begin;
    w.je = args.j_end;
    w.T = w.je - w.jb;
    w.do_fwd = (args.phase & 1) != 0;
    w.do_bwd = (args.phase & 2) != 0;
    w.nsteps = args.nsteps;
    if (w.jb > 0 || !w.do_fwd)
        if ((*(volatile int*)args.status) & 4) return;  // an earlier segment overflowed
    // first sub-step slot of this launch's first step (forward) / one past its last (adjoint only)
    const int* offs_b = args.offs + (size_t)blockIdx.x * (args.nsteps + 1);
    int slot = 0;
    if (w.do_fwd) slot = (w.jb == 0) ? 0 : offs_b[w.jb];
    else if (args.unit_adjoint)  // x slots of their own, counted down from the capacity
        slot = (w.je == args.nsteps) ? (int)w.cap
                                     : args.offs_x[(size_t)blockIdx.x * (args.nsteps + 1) + w.je];
    else slot = offs_b[w.je];
    if (role == 2) s3_loader<NB, STAMP>(args, w, smem, slot);
    else if (role == 1) s3_inverter<NB, STAMP, false>(args, w, smem, slot);
    else if (role == 3) s3_inverter<NB, STAMP, true>(args, w, smem, slot);
    else s3_compute<NB, STAMP>(args, w, smem, slot);
}

template <int NB, bool STAMP>
static void launch_sweep3_t(const SweepArgs& a, int batch, hipStream_t st) {
    const int bytes = S3<NB>::bytes(a.S);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sweep3_kernel<NB, STAMP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL((sweep3_kernel<NB, STAMP>), dim3(batch), dim3(256), bytes, st, a);
}

// largest state count whose vectors fit beside the ring (160 KiB of LDS per workgroup)
int sweep3_max_states(int nb) {
    const int fixed = nb == 1 ? S3<1>::VEC_OFF : S3<2>::VEC_OFF;
    const int np = 16 * nb;
    return (160 * 1024 - fixed) / (3 * np * 16);
}

void launch_sweep3(int nb, const SweepArgs& a, int batch, hipStream_t st) {
    if (nb == 1) launch_sweep3_t<1, false>(a, batch, st);
#ifdef QOCX_DIAG
    else if (a.stamps != nullptr) launch_sweep3_t<2, true>(a, batch, st);  // stamped build (qocx_diag.h)
#endif
    else launch_sweep3_t<2, false>(a, batch, st);
}

}  // namespace qocx
