// qocx_sweep_core.h - what the sweep kernels share (qocx_kernels.hip: the column-chain sweep in its
// general form; qocx_sweep1.hip: the one-state form of the batched evaluator): the unit-diagonal
// triangular solves on a lane-per-row copy of the LU image, the matrix-vector product against an
// image in LDS, the LDS layout of a seed. Moved out of qocx_kernels.hip in round 5.
#ifndef QOCX_SWEEP_CORE_H
#define QOCX_SWEEP_CORE_H

#include <algorithm>
#include <type_traits>
#include <utility>

#include "qocx_wave.h"
#include "qocx_sweep_common.h"

namespace qocx {

// ------------------------------------------------------------------------------------------
// K2: serial state sweep (forward), costs, adjoint sweep (backward)
// ------------------------------------------------------------------------------------------

// Unit-diagonal triangular solves, axpy form, coefficient rows in F-layout (every lane group
// holds the full row i = lane % NP, so all groups run the solve redundantly and z stays
// replicated). Per column k: z_k is broadcast with v_readlane and the rows below (LOWER) or
// above it take one complex FMA; the row predicate is a compile-time EXEC mask, so the update is
// exactly four v_fma_f64. CONJ: use conj of the stored coefficients.
template <int NB, bool LOWER>
__device__ __forceinline__ constexpr unsigned long long row_mask(int k) {
    constexpr int NP = Geo<NB>::NP;
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const int i = l % NP;
        if (LOWER ? (i > k) : (i < k)) m |= (1ull << l);
    }
    return m;
}

// z -= c * zk (or conj(c) * zk) on the lanes of a compile-time EXEC mask (two 32-bit literals:
// no SGPRs are tied up); all 64 lanes are active on entry and on exit.
template <bool CONJ, unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_cfma(double& zre, double& zim, double cre, double cim,
                                            double kre, double kim) {
    if (CONJ) {
        asm volatile(
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_fma_f64 %[zr], -%[cr], %[kr], %[zr]\n\t"
            "v_fma_f64 %[zi], -%[cr], %[ki], %[zi]\n\t"
            "v_fma_f64 %[zr], -%[ci], %[ki], %[zr]\n\t"
            "v_fma_f64 %[zi], %[ci], %[kr], %[zi]\n\t"
            "s_mov_b64 exec, -1"
            : [zr] "+v"(zre), [zi] "+v"(zim)
            : [cr] "v"(cre), [ci] "v"(cim), [kr] "s"(kre), [ki] "s"(kim), [lo] "i"(LO), [hi] "i"(HI)
            : "memory");
    } else {
        asm volatile(
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_fma_f64 %[zr], -%[cr], %[kr], %[zr]\n\t"
            "v_fma_f64 %[zi], -%[cr], %[ki], %[zi]\n\t"
            "v_fma_f64 %[zr], %[ci], %[ki], %[zr]\n\t"
            "v_fma_f64 %[zi], -%[ci], %[kr], %[zi]\n\t"
            "s_mov_b64 exec, -1"
            : [zr] "+v"(zre), [zi] "+v"(zim)
            : [cr] "v"(cre), [ci] "v"(cim), [kr] "s"(kre), [ki] "s"(kim), [lo] "i"(LO), [hi] "i"(HI)
            : "memory");
    }
}

template <int NB, bool LOWER, bool CONJ, int KK, class Hook>
__device__ __forceinline__ void tri_step(const double (&tre)[Geo<NB>::NP],
                                         const double (&tim)[Geo<NB>::NP], double& zre,
                                         double& zim, Hook& hook) {
    constexpr int NP = Geo<NB>::NP;
    constexpr int k = LOWER ? KK : (NP - 1 - KK);
    constexpr unsigned long long mask = row_mask<NB, LOWER>(k);
    const double kre = readlane_f64(zre, k), kim = readlane_f64(zim, k);
    hook(std::integral_constant<int, KK>());  // independent work for the chain's bubbles
    masked_cfma<CONJ, (unsigned)(mask & 0xffffffffull), (unsigned)(mask >> 32)>(
        zre, zim, tre[k], tim[k], kre, kim);
}

template <int NB, bool LOWER, bool CONJ, class Hook, int... KK>
__device__ __forceinline__ void tri_solve_seq(const double (&tre)[Geo<NB>::NP],
                                              const double (&tim)[Geo<NB>::NP], double& zre,
                                              double& zim, Hook& hook,
                                              std::integer_sequence<int, KK...>) {
    (tri_step<NB, LOWER, CONJ, KK>(tre, tim, zre, zim, hook), ...);
}

// ---- the same solves with the broadcast INSIDE the multiply-add (round 5) ------------------------
// v_fmac_f64_dpp row_newbcast:k reads its first factor from lane k of the reader's own row of 16
// lanes, so a stage is four instructions and the value never travels through a scalar register
// (v_readlane -> SGPR -> v_fma was ~110 cycles per stage of the dependent chain, SALU and all; the
// chain is now multiply-add to multiply-add). A DPP operand whose SOURCE lane is switched off
// counts as invalid, so lane k itself stays on during stage k: its coefficient there is the
// diagonal entry of the image, which the sweep has overwritten with zero in LDS before the row went
// to registers (zero_lu_diagonal) - z_k -= 0 z_k. NP = 32: rows 0..15 live in the even rows of 16
// lanes, rows 16..31 in the odd ones, so the solve runs block-wise - the diagonal block of the half
// that goes first stage by stage, then the off-diagonal block as sixteen multiply-adds against a
// copy of the finished half in the other rows (v_permlane16_swap), then the second diagonal block.
// Every entry receives the updates it received before, in the same order: results are bit-identical
// to the v_readlane form (QOCX_SWEEP_DPP 0 builds it).
#ifndef QOCX_SWEEP_DPP
#define QOCX_SWEEP_DPP 1
#endif

// lanes of the rows still in play at stage k, the source row k included
template <int NB, bool LOWER>
__device__ __forceinline__ constexpr unsigned long long row_mask_incl(int k) {
    constexpr int NP = Geo<NB>::NP;
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        const int i = l % NP;
        if (LOWER ? (i >= k) : (i <= k)) m |= (1ull << l);
    }
    return m;
}

// z -= c * z[lane BC of the row] (or conj(c)) on the lanes of the EXEC mask within the rows RM
template <bool CONJ, unsigned LO, unsigned HI, int BC, int RM>
__device__ __forceinline__ void dpp_cfma_diag(double& zre, double& zim, double cre, double cim) {
    if (CONJ) {
        asm volatile(
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_fmac_f64_dpp %[zr], -%[zr], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], -%[zi], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zr], -%[zi], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], %[zr], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "s_mov_b64 exec, -1"
            : [zr] "+v"(zre), [zi] "+v"(zim)
            : [cr] "v"(cre), [ci] "v"(cim), [lo] "i"(LO), [hi] "i"(HI), [bc] "i"(BC), [rm] "i"(RM)
            : "memory");
    } else {
        asm volatile(
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_fmac_f64_dpp %[zr], -%[zr], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], -%[zi], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zr], %[zi], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], -%[zr], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "s_mov_b64 exec, -1"
            : [zr] "+v"(zre), [zi] "+v"(zim)
            : [cr] "v"(cre), [ci] "v"(cim), [lo] "i"(LO), [hi] "i"(HI), [bc] "i"(BC), [rm] "i"(RM)
            : "memory");
    }
}
// z -= c * w[lane BC of the row] within the rows RM, every lane on (w: the finished half, copied
// into these rows). NOPS: wait states in front of the first DPP read of a freshly written w.
template <bool CONJ, int BC, int RM, bool FIRST>
__device__ __forceinline__ void dpp_cfma_off(double& zre, double& zim, double cre, double cim,
                                             double wre, double wim) {
    if (FIRST) asm volatile("s_nop 1" ::: "memory");
    if (CONJ) {
        asm volatile(
            "v_fmac_f64_dpp %[zr], -%[wr], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], -%[wi], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zr], -%[wi], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], %[wr], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf"
            : [zr] "+v"(zre), [zi] "+v"(zim)
            : [cr] "v"(cre), [ci] "v"(cim), [wr] "v"(wre), [wi] "v"(wim), [bc] "i"(BC), [rm] "i"(RM)
            : "memory");
    } else {
        asm volatile(
            "v_fmac_f64_dpp %[zr], -%[wr], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], -%[wi], %[cr] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zr], %[wi], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %[zi], -%[wr], %[ci] row_newbcast:%[bc] row_mask:%[rm] bank_mask:0xf"
            : [zr] "+v"(zre), [zi] "+v"(zim)
            : [cr] "v"(cre), [ci] "v"(cim), [wr] "v"(wre), [wi] "v"(wim), [bc] "i"(BC), [rm] "i"(RM)
            : "memory");
    }
}

// the finished half of z copied into the rows of the other half: even rows of 16 lanes -> the odd
// row behind each (FROM_EVEN), or odd rows -> the even row in front of each
template <bool FROM_EVEN>
__device__ __forceinline__ double half_copy(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return FROM_EVEN ? make_f64((int)a[0], (int)b[0]) : make_f64((int)a[1], (int)b[1]);
}

// stage KK of a diagonal block (NP = 16: of the whole solve)
template <int NB, bool LOWER, bool CONJ, int KK, class Hook>
__device__ __forceinline__ void tri_step_dpp(const double (&tre)[Geo<NB>::NP],
                                             const double (&tim)[Geo<NB>::NP], double& zre,
                                             double& zim, Hook& hook) {
    constexpr int NP = Geo<NB>::NP;
    constexpr int k = LOWER ? KK : (NP - 1 - KK);
    constexpr unsigned long long mask = row_mask_incl<NB, LOWER>(k);
    constexpr int RM = NB == 1 ? 0xf : (k < 16 ? 0x5 : 0xa);
    hook(std::integral_constant<int, KK>());
    if constexpr (NB == 2 && KK == 15) {
        // the off-diagonal block: rows 16..31 take columns 0..15 (LOWER), rows 0..15 columns
        // 31..16 (UPPER), in the order of the stage-by-stage form
        const double wre = half_copy<LOWER>(zre), wim = half_copy<LOWER>(zim);
        for_each_const(
            [&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                constexpr int kk = LOWER ? j : (NP - 1 - j);
                dpp_cfma_off<CONJ, kk % 16, LOWER ? 0xa : 0x5, j == 0>(zre, zim, tre[kk], tim[kk], wre, wim);
            },
            std::make_integer_sequence<int, 16>{});
    } else {
        dpp_cfma_diag<CONJ, (unsigned)(mask & 0xffffffffull), (unsigned)(mask >> 32), k % 16, RM>(
            zre, zim, tre[k], tim[k]);
    }
}
template <int NB, bool LOWER, bool CONJ, class Hook, int... KK>
__device__ __forceinline__ void tri_solve_dpp_seq(const double (&tre)[Geo<NB>::NP],
                                                  const double (&tim)[Geo<NB>::NP], double& zre,
                                                  double& zim, Hook& hook,
                                                  std::integer_sequence<int, KK...>) {
    (tri_step_dpp<NB, LOWER, CONJ, KK>(tre, tim, zre, zim, hook), ...);
}

// hook(kk), kk = 0 .. NP-2, is called once per column (between the broadcast and the update in the
// v_readlane form; in front of the stage in the DPP form).
template <int NB, bool LOWER, bool CONJ, class Hook>
__device__ __forceinline__ void tri_solve(const double (&tre)[Geo<NB>::NP],
                                          const double (&tim)[Geo<NB>::NP], double& zre,
                                          double& zim, Hook& hook) {
#if QOCX_SWEEP_DPP
    if constexpr (NB <= 2) {
        tri_solve_dpp_seq<NB, LOWER, CONJ>(tre, tim, zre, zim, hook,
                                           std::make_integer_sequence<int, Geo<NB>::NP - 1>{});
        return;
    }
#endif
    tri_solve_seq<NB, LOWER, CONJ>(tre, tim, zre, zim, hook,
                                   std::make_integer_sequence<int, Geo<NB>::NP - 1>{});
}

// Two right-hand sides at once (two states of a seed on one wave): the two dependent chains are
// independent of each other, so each fills the other's bubbles - the broadcasts of one pair of
// z_k overlap the updates of the other - and the coefficient row is used twice.
template <bool CONJ, unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_cfma2(double& are, double& aim, double& bre, double& bim,
                                             double cre, double cim, double kar, double kai,
                                             double kbr, double kbi) {
    if (CONJ) {
        asm volatile(
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_fma_f64 %[ar], -%[cr], %[kar], %[ar]\n\t"
            "v_fma_f64 %[ai], -%[cr], %[kai], %[ai]\n\t"
            "v_fma_f64 %[br], -%[cr], %[kbr], %[br]\n\t"
            "v_fma_f64 %[bi], -%[cr], %[kbi], %[bi]\n\t"
            "v_fma_f64 %[ar], -%[ci], %[kai], %[ar]\n\t"
            "v_fma_f64 %[ai], %[ci], %[kar], %[ai]\n\t"
            "v_fma_f64 %[br], -%[ci], %[kbi], %[br]\n\t"
            "v_fma_f64 %[bi], %[ci], %[kbr], %[bi]\n\t"
            "s_mov_b64 exec, -1"
            : [ar] "+v"(are), [ai] "+v"(aim), [br] "+v"(bre), [bi] "+v"(bim)
            : [cr] "v"(cre), [ci] "v"(cim), [kar] "s"(kar), [kai] "s"(kai), [kbr] "s"(kbr),
              [kbi] "s"(kbi), [lo] "i"(LO), [hi] "i"(HI)
            : "memory");
    } else {
        asm volatile(
            "s_mov_b32 exec_lo, %[lo]\n\t"
            "s_mov_b32 exec_hi, %[hi]\n\t"
            "v_fma_f64 %[ar], -%[cr], %[kar], %[ar]\n\t"
            "v_fma_f64 %[ai], -%[cr], %[kai], %[ai]\n\t"
            "v_fma_f64 %[br], -%[cr], %[kbr], %[br]\n\t"
            "v_fma_f64 %[bi], -%[cr], %[kbi], %[bi]\n\t"
            "v_fma_f64 %[ar], %[ci], %[kai], %[ar]\n\t"
            "v_fma_f64 %[ai], -%[ci], %[kar], %[ai]\n\t"
            "v_fma_f64 %[br], %[ci], %[kbi], %[br]\n\t"
            "v_fma_f64 %[bi], -%[ci], %[kbr], %[bi]\n\t"
            "s_mov_b64 exec, -1"
            : [ar] "+v"(are), [ai] "+v"(aim), [br] "+v"(bre), [bi] "+v"(bim)
            : [cr] "v"(cre), [ci] "v"(cim), [kar] "s"(kar), [kai] "s"(kai), [kbr] "s"(kbr),
              [kbi] "s"(kbi), [lo] "i"(LO), [hi] "i"(HI)
            : "memory");
    }
}

template <int NB, bool LOWER, bool CONJ, int KK, class Hook>
__device__ __forceinline__ void tri_step2(const double (&tre)[Geo<NB>::NP],
                                          const double (&tim)[Geo<NB>::NP], double& are, double& aim,
                                          double& bre, double& bim, Hook& hook) {
    constexpr int NP = Geo<NB>::NP;
    constexpr int k = LOWER ? KK : (NP - 1 - KK);
    constexpr unsigned long long mask = row_mask<NB, LOWER>(k);
    const double kar = readlane_f64(are, k), kai = readlane_f64(aim, k);
    const double kbr = readlane_f64(bre, k), kbi = readlane_f64(bim, k);
    hook(std::integral_constant<int, KK>());
    masked_cfma2<CONJ, (unsigned)(mask & 0xffffffffull), (unsigned)(mask >> 32)>(
        are, aim, bre, bim, tre[k], tim[k], kar, kai, kbr, kbi);
}
template <int NB, bool LOWER, bool CONJ, class Hook, int... KK>
__device__ __forceinline__ void tri_solve2_seq(const double (&tre)[Geo<NB>::NP],
                                               const double (&tim)[Geo<NB>::NP], double& are,
                                               double& aim, double& bre, double& bim, Hook& hook,
                                               std::integer_sequence<int, KK...>) {
    (tri_step2<NB, LOWER, CONJ, KK>(tre, tim, are, aim, bre, bim, hook), ...);
}
// DPP form (see tri_solve): the two right-hand sides stage by stage, one behind the other
template <int NB, bool LOWER, bool CONJ, int KK, class Hook>
__device__ __forceinline__ void tri_step2_dpp(const double (&tre)[Geo<NB>::NP],
                                              const double (&tim)[Geo<NB>::NP], double& are, double& aim,
                                              double& bre, double& bim, Hook& hook) {
    auto none = [](auto) {};
    tri_step_dpp<NB, LOWER, CONJ, KK>(tre, tim, are, aim, hook);
    tri_step_dpp<NB, LOWER, CONJ, KK>(tre, tim, bre, bim, none);
}
template <int NB, bool LOWER, bool CONJ, class Hook, int... KK>
__device__ __forceinline__ void tri_solve2_dpp_seq(const double (&tre)[Geo<NB>::NP],
                                                   const double (&tim)[Geo<NB>::NP], double& are,
                                                   double& aim, double& bre, double& bim, Hook& hook,
                                                   std::integer_sequence<int, KK...>) {
    (tri_step2_dpp<NB, LOWER, CONJ, KK>(tre, tim, are, aim, bre, bim, hook), ...);
}
template <int NB, bool LOWER, bool CONJ, class Hook>
__device__ __forceinline__ void tri_solve2(const double (&tre)[Geo<NB>::NP],
                                           const double (&tim)[Geo<NB>::NP], double& are, double& aim,
                                           double& bre, double& bim, Hook& hook) {
#if QOCX_SWEEP_DPP
    if constexpr (NB <= 2) {
        tri_solve2_dpp_seq<NB, LOWER, CONJ>(tre, tim, are, aim, bre, bim, hook,
                                            std::make_integer_sequence<int, Geo<NB>::NP - 1>{});
        return;
    }
#endif
    tri_solve2_seq<NB, LOWER, CONJ>(tre, tim, are, aim, bre, bim, hook,
                                    std::make_integer_sequence<int, Geo<NB>::NP - 1>{});
}

// The same solves with the coefficients fetched from the LDS image stage by stage (NB = 4: a lane's
// row of the LU image would be 256 registers; held there, the sweep wave owns a whole SIMD and only
// one four-wave K3 workgroup fits on the other three). The coefficient of stage KK + 8 is requested
// while stage KK runs (a ring of eight in registers); the asm statements of the chain carry memory
// clobbers, so the requests stay where they are written. Forward: the lane at position i takes row
// pm = perm[i], element (col k, row pm). Adjoint (the image in LDS is the transposed one): column
// perm[k], element (col perm[k], row i), perm[k] by v_readlane from `permv` (lane l holds perm[l]).
template <int NB, bool ADJ>
__device__ __forceinline__ double2 lds_coef(const double2* lb, int pm, int i, int permv, int k) {
    constexpr int NP = Geo<NB>::NP;
    if constexpr (ADJ) {
        const int pc = min(max(__builtin_amdgcn_readlane(permv, k), 0), NP - 1);
        return lb[pc * NP + i];
    } else {
        return lb[k * NP + pm];
    }
}

template <int NB, bool LOWER, bool CONJ, bool ADJ, int KK, int NA, class Hook>
__device__ __forceinline__ void tri_step_lds(const double2* lb, int pm, int i, int permv,
                                             double2 (&ring)[8], double& zre, double& zim,
                                             Hook& hook) {
    constexpr int D = 8;
    constexpr int k = LOWER ? KK : (NA - 1 - KK);
    constexpr unsigned long long mask = row_mask<NB, LOWER>(k);
    const double2 c = ring[KK % D];
    if constexpr (KK + D < NA - 1) {
        constexpr int kn = LOWER ? (KK + D) : (NA - 1 - (KK + D));
        ring[KK % D] = lds_coef<NB, ADJ>(lb, pm, i, permv, kn);
    }
    const double kre = readlane_f64(zre, k), kim = readlane_f64(zim, k);
    hook(std::integral_constant<int, KK>());  // independent work for the chain's bubbles
    masked_cfma<CONJ, (unsigned)(mask & 0xffffffffull), (unsigned)(mask >> 32)>(
        zre, zim, c.x, c.y, kre, kim);
}

template <int NB, bool LOWER, bool CONJ, bool ADJ, int NA, class Hook, int... KK>
__device__ __forceinline__ void tri_solve_lds_seq(const double2* lb, int pm, int i, int permv,
                                                  double2 (&ring)[8], double& zre, double& zim,
                                                  Hook& hook, std::integer_sequence<int, KK...>) {
    (tri_step_lds<NB, LOWER, CONJ, ADJ, KK, NA>(lb, pm, i, permv, ring, zre, zim, hook), ...);
}

// NA < NP (n <= 48 in a 64 x 64 image): rows and columns NA .. NP - 1 are the pad block - unit
// columns, zero multipliers - so the stages of those columns do nothing and are left out.
template <int NB, bool LOWER, bool CONJ, bool ADJ, int NA = Geo<NB>::NP, class Hook>
__device__ __forceinline__ void tri_solve_lds(const double2* lb, int pm, int i, int permv,
                                              double& zre, double& zim, Hook& hook) {
    constexpr int D = 8;
    double2 ring[D];
#pragma unroll
    for (int j = 0; j < D; ++j)
        ring[j] = lds_coef<NB, ADJ>(lb, pm, i, permv, LOWER ? j : (NA - 1 - j));
    tri_solve_lds_seq<NB, LOWER, CONJ, ADJ, NA>(lb, pm, i, permv, ring, zre, zim, hook,
                                                std::make_integer_sequence<int, NA - 1>{});
}

template <int NB>
struct SweepPrefetch {
    static constexpr bool value = NB < 4;
};
// L / U coefficients from the LDS image stage by stage (tri_solve_lds) instead of a register row:
// for sixteen tiles only. Measured at n = 32 and n = 16 (-DQOCX_LDSCOEF_MIN_NB=1, bit-identical
// results): the sweep takes 1.57 instead of 0.525 ms per 125-step segment (0.54 instead of 0.27 at
// n = 16) - the extra LDS round trip sits on the dependent chain there, while at NB = 4 the register
// row costs a whole SIMD and scratch.
#ifndef QOCX_LDSCOEF_MIN_NB
#define QOCX_LDSCOEF_MIN_NB 4
#endif
template <int NB>
struct SweepLdsCoef {
    static constexpr bool value = NB >= QOCX_LDSCOEF_MIN_NB;
};

// NA: columns of an image that are fetched into LDS (NB = 4, n <= 48: the 48 columns that are not
// the pad block - 48 KiB per image instead of 64, so that a K1a workgroup fits on the CU beside the
// sweep's; everywhere else the whole image)
template <int NB, int NBUF = (SweepPrefetch<NB>::value ? 2 : 1), int NA = Geo<NB>::NP>
struct SweepLds {
    typedef Geo<NB> G;
    static constexpr int BUF_BYTES = NA * G::NP * 16;          // one matrix image (NA columns)
    static constexpr int Q_OFF = 0;                            // NBUF x Q image (ring)
    static constexpr int L_OFF = Q_OFF + NBUF * BUF_BYTES;      // NBUF x LU image
    static constexpr int D_OFF = L_OFF + NBUF * BUF_BYTES;      // NBUF x 64 complex: 1/U_kk
    static constexpr int P_OFF = D_OFF + NBUF * 64 * 16;        // NBUF x PINTS int: perm | iperm
    static constexpr int PINTS = G::NP > 32 ? 128 : 64;        // iperm starts at PINTS / 2
    static constexpr int MAX_WAVES = 4;                        // waves per seed (multi-state)
    static constexpr int TMP_OFF = P_OFF + NBUF * PINTS * 4;    // TMPV x NP complex scratch per wave
    static constexpr int TMPV = SweepLdsCoef<NB>::value ? 1 : 2;  // (two: the paired-state form)
    static constexpr int VEC_OFF = TMP_OFF + MAX_WAVES * TMPV * G::NP * 16;  // [S][NP] states, [S][NP] lambda
    static int bytes(int S) { return VEC_OFF + 2 * S * G::NP * 16; }
    __host__ __device__ static constexpr int bytes_static(int S) { return VEC_OFF + 2 * S * G::NP * 16; }
};

// z := z * d and z := z * conj(d) (d = 1 / U_kk between the two solves), as explicit multiply-adds:
// the two sweep kernels must round alike (left to the compiler, the contraction of a product and a
// sum into an FMA depends on the code around it)
__device__ __forceinline__ void cscale(double& zre, double& zim, const double2 d) {
    const double t = fma(zre, d.x, -(zim * d.y));
    zim = fma(zre, d.y, zim * d.x);
    zre = t;
}
__device__ __forceinline__ void cscale_conj(double& zre, double& zim, const double2 d) {
    const double t = fma(zre, d.x, zim * d.y);
    zim = fma(zim, d.x, -(zre * d.y));
    zre = t;
}

// Per-step operands of the sweep, in registers: Q in R-layout (matvec), LU in F-layout (solves).
template <int NB>
struct StepRegs {
    double lre[Geo<NB>::NP], lim[Geo<NB>::NP];
};

struct StepScalars {
    double2 dv;  // this lane's 1/U_ii (position i)
    int pm;      // forward: perm[i] (row at position i); adjoint: iperm[i] (position of row i)
};

// The same with an instruction offset IMM (13 bits, signed): the 16 bytes at g + IMM land at
// lds_dst + 16*l. The hardware adds the offset to the global AND to the LDS address, so M0 gets
// lds_dst - IMM. One per-lane base address then serves many pieces of an image.
template <int IMM>
__device__ __forceinline__ void dma16_imm(const char* g, char* lds_dst) {
    static_assert(IMM >= -4096 && IMM <= 4095, "instruction offset range");
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)g,
        (__attribute__((address_space(3))) void*)(lds_dst - IMM), 16, IMM, 0);
}

// LDS images -> registers. The LU image is stored in original row order, so the row at position
// k is row perm[k]: the forward F-layout gathers row perm[i] per lane; the adjoint (transposed
// image) reads column perm[k] for every k, with perm[k] fetched as an LDS broadcast.
template <int NB, bool ADJOINT>
__device__ __forceinline__ void lds_to_regs(const double2* qb, const double2* lb, const int* pb,
                                            StepRegs<NB>& r, int pm, int lane, int i) {
    typedef Geo<NB> G;
#if QOCX_SWEEP_DPP
    if constexpr (NB <= 2) {
        // the diagonal entry of this lane's row (position i, column i; the pivot, which the solves
        // take from 1/U_kk) becomes zero: stage i of a solve keeps lane i switched on (tri_solve)
        const int d = ADJOINT ? min(max(pb[i], 0), G::NP - 1) * G::NP + i : i * G::NP + pm;
        const_cast<double2*>(lb)[d] = make_double2(0.0, 0.0);
        wave_sync();
    }
#endif
#pragma unroll
    for (int c = 0; c < G::NP; ++c) {
        int src;
        if (ADJOINT) src = min(max(pb[c], 0), G::NP - 1) * G::NP + i;
        else src = c * G::NP + pm;
        const double2 e = lb[src];
        r.lre[c] = e.x;
        r.lim[c] = e.y;
    }
}

// Partial row sums of a matvec whose matrix sits in LDS as an R-layout image (`qlane` = the image
// lane this lane takes: its own, or the one of a permuted row) and whose vector is broadcast from
// LDS; BATCH (matrix, vector) pairs of LDS reads are in flight ahead of their FMAs.
template <int NB, bool CONJ, int BATCH, int NA = Geo<NB>::NP>
__device__ __forceinline__ void lds_matvec(const double2* qb, const double2* vec, int qlane, int h,
                                           double& yre, double& yim) {
    typedef Geo<NB> G;
    constexpr int CPL = NA / G::H, H = G::H;  // (NA < NP: the pad columns carry nothing)
    double ar = 0, ai = 0;
#pragma unroll
    for (int c0 = 0; c0 < CPL; c0 += BATCH) {
        double2 qv[BATCH], xv[BATCH];
#pragma unroll
        for (int cc = 0; cc < BATCH; ++cc) {
            qv[cc] = qb[(c0 + cc) * 64 + qlane];
            xv[cc] = vec[(c0 + cc) * H + h];
        }
#pragma unroll
        for (int cc = 0; cc < BATCH; ++cc) {
            const double qi = CONJ ? -qv[cc].y : qv[cc].y;
            ar = fma(-qi, xv[cc].y, fma(qv[cc].x, xv[cc].x, ar));
            ai = fma(qi, xv[cc].x, fma(qv[cc].x, xv[cc].y, ai));
        }
        asm volatile("" ::: "memory");
    }
    yre = sum_groups<NB>(ar);
    yim = sum_groups<NB>(ai);
}

// Two vectors against the same matrix: every matrix element is read from LDS once.
template <int NB, bool CONJ, int BATCH>
__device__ __forceinline__ void lds_matvec2(const double2* qb, const double2* veca, const double2* vecb,
                                            int qlane, int h, double& yar, double& yai, double& ybr,
                                            double& ybi) {
    typedef Geo<NB> G;
    constexpr int CPL = G::CPL, H = G::H;
    double ar = 0, ai = 0, br = 0, bi = 0;
#pragma unroll
    for (int c0 = 0; c0 < CPL; c0 += BATCH) {
        double2 qv[BATCH], xa[BATCH], xb[BATCH];
#pragma unroll
        for (int cc = 0; cc < BATCH; ++cc) {
            qv[cc] = qb[(c0 + cc) * 64 + qlane];
            xa[cc] = veca[(c0 + cc) * H + h];
            xb[cc] = vecb[(c0 + cc) * H + h];
        }
#pragma unroll
        for (int cc = 0; cc < BATCH; ++cc) {
            const double qi = CONJ ? -qv[cc].y : qv[cc].y;
            ar = fma(-qi, xa[cc].y, fma(qv[cc].x, xa[cc].x, ar));
            ai = fma(qi, xa[cc].x, fma(qv[cc].x, xa[cc].y, ai));
            br = fma(-qi, xb[cc].y, fma(qv[cc].x, xb[cc].x, br));
            bi = fma(qi, xb[cc].x, fma(qv[cc].x, xb[cc].y, bi));
        }
        asm volatile("" ::: "memory");
    }
    yar = sum_groups<NB>(ar);
    yai = sum_groups<NB>(ai);
    ybr = sum_groups<NB>(br);
    ybi = sum_groups<NB>(bi);
}

}  // namespace qocx

#endif
