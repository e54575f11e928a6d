// qocx_wave.h - wave-level primitives, layouts and the LDS-staged complex GEMM on
// v_mfma_f64_16x16x4_f64 shared by the Schroedinger (qocx_kernels.hip) and Lindblad
// (qocx_lindblad.hip) kernels. See qocx_kernels.hip for the layout conventions.
#ifndef QOCX_WAVE_H
#define QOCX_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "qocx_device.h"
#include "qocx_diag.h"

namespace qocx {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __constant__ const double PADE_B[14] = {
    64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
    129060195264000.0,   10559470521600.0,    670442572800.0,     33522128640.0,
    1323241920.0,        40840800.0,          960960.0,           16380.0,
    182.0,               1.0};

#define QOCX_THETA13 5.371920351148152

// Pade order by norm. The reference always evaluates the [13/13] approximant (expm.py:230-233:
// its loop over PADE_ORDERS has no break, so 13 wins whenever the norm is below theta_13). The
// device takes the order from the table the reference cites and carries (Higham 2005, Algorithm
// 2.3; THETA at expm.py:194-209): for ||a||_1 <= theta_m the [m/m] approximant is exp(a + da) with
// ||da|| <= 2^-53 ||a|| - the same matrix as the [13/13] one to rounding, and its derivative the
// same to rounding (tests/test_device_model.py holds both to 1e-14 against the reference's
// formulas) - for 2..5 matrix products instead of 6 in K1a and an adjoint chain of m instead of 13
// terms in K3. Coefficients b_j = (2m - j)! m! / ((2m)! (m - j)! j!) scaled to integers, as in the
// reference's table for m = 13 (expm.py:85-101). K1a decides from an UPPER bound of the norm (sum
// of |re| + |im|), so the order is never lower than the norm allows; it travels to K3 in bits
// 8..15 of the step's entry in s_arr (0: order 13; the squaring count is 0 for any other order).
#define QOCX_THETA3 1.495585217958292e-2
#define QOCX_THETA5 2.539398330063230e-1
#define QOCX_THETA7 9.504178996162932e-1
#define QOCX_THETA9 2.097847961257068
__device__ __constant__ const double PADE_B3[4] = {120.0, 60.0, 12.0, 1.0};
__device__ __constant__ const double PADE_B5[6] = {30240.0, 15120.0, 3360.0, 420.0, 30.0, 1.0};
__device__ __constant__ const double PADE_B7[8] = {17297280.0, 8648640.0, 1995840.0, 277200.0,
                                                   25200.0,    1512.0,    56.0,      1.0};
__device__ __constant__ const double PADE_B9[10] = {17643225600.0, 8821612800.0, 2075673600.0,
                                                    302702400.0,   30270240.0,   2162160.0,
                                                    110880.0,      3960.0,       90.0,
                                                    1.0};
template <int M>
__device__ __forceinline__ double pade_b(int j) {
    if constexpr (M == 3) return PADE_B3[j];
    else if constexpr (M == 5) return PADE_B5[j];
    else if constexpr (M == 7) return PADE_B7[j];
    else if constexpr (M == 9) return PADE_B9[j];
    else return PADE_B[j];
}
__device__ __forceinline__ const double* pade_table(int order) {
    return order == 3 ? PADE_B3 : (order == 5 ? PADE_B5 : (order == 7 ? PADE_B7 : (order == 9 ? PADE_B9 : PADE_B)));
}
// v: an upper bound of ||a||_1; policy 13: always 13 (as the reference executes)
__device__ __forceinline__ int pade_order_for(double v, int policy) {
    if (policy == 13) return 13;
    if (v < QOCX_THETA3) return 3;
    if (v < QOCX_THETA5) return 5;
    if (v < QOCX_THETA7) return 7;
    if (v < QOCX_THETA9) return 9;
    return 13;  // also NaN
}
__device__ __forceinline__ int step_entry(int squarings, int order) {
    return order == 13 ? squarings : (squarings | (order << 8));
}
__device__ __forceinline__ int step_squarings(int entry) { return min(max(entry & 0xff, 0), 30); }
// bit 16: the Pade denominator of the step is diagonally dominant by a margin that makes LAPACK's
// pivots the diagonal ones (qocx_lu5.h, pade_denominator_dominant)
#define QOCX_STEP_DOMINANT 0x10000
__device__ __forceinline__ bool step_dominant(int entry) { return (entry & QOCX_STEP_DOMINANT) != 0; }
__device__ __forceinline__ int step_order(int entry) {
    const int o = (entry >> 8) & 0xff;
    return o == 0 ? 13 : o;
}

// ------------------------------------------------------------------------------------------
// wave-level primitives
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ double make_f64(int lo, int hi) { return __hiloint2double(hi, lo); }

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return make_f64(lo, hi);
}

// lane must be wave-uniform
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return make_f64(lo, hi);
}

// DPP controls (gfx9): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140. Four symmetric exchanges leave every lane of a 16-lane row with the row
// result; the four rows are then combined through SGPRs.
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return fmax(fmax(r0, r1), fmax(r2, r3));
}
__device__ __forceinline__ double wave_sum(double v) {
    v = v + dpp_f64<0xB1>(v);
    v = v + dpp_f64<0x4E>(v);
    v = v + dpp_f64<0x141>(v);
    v = v + dpp_f64<0x140>(v);
    double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16);
    double r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// wave-wide max of a u32: four symmetric DPP exchanges inside each 16-lane row, then
// row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3) carry the row results into lane 63
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Every block is exactly ONE wavefront. The LDS executes a wave's DS instructions in issue
// order, so a later ds_read of the same wave observes an earlier ds_write without a hardware
// barrier; this only has to stop the compiler from reordering the accesses.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 1/x to full double precision: v_rcp_f64 + two Newton steps.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// f(integral_constant<int, P>) for every P of the sequence
template <class F, int... P>
__device__ __forceinline__ void for_each_const(F&& f, std::integer_sequence<int, P...>) {
    (f(std::integral_constant<int, P>()), ...);
}

// ------------------------------------------------------------------------------------------
// layouts
// ------------------------------------------------------------------------------------------

template <int NB>
struct Geo {
    static constexpr int NP = 16 * NB;        // padded Hilbert dimension
    static constexpr int PITCH = NP + 2;      // LDS row pitch (f64) of the planar A-operand slot
    static constexpr int H = 64 / NP;         // lane groups per row in R-layout
    static constexpr int CPL = NP / H;        // columns per lane in R-layout
    static constexpr int MAT = NP * NP;       // complex elements per matrix image
    static constexpr int PLANE = NP * PITCH;  // f64 per LDS plane
};

template <int NB>
struct CMat {  // C-layout complex matrix in registers
    d4 re[NB][NB];
    d4 im[NB][NB];
};

template <int NB>
__device__ __forceinline__ void cmat_zero(CMat<NB>& m) {
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            m.re[ti][tj] = d4{0, 0, 0, 0};
            m.im[ti][tj] = d4{0, 0, 0, 0};
        }
}

template <int NB>
__device__ __forceinline__ void cmat_scale(CMat<NB>& m, double s) {
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            m.re[ti][tj] *= s;
            m.im[ti][tj] *= s;
        }
}

// C-layout registers -> planar LDS slot (row-major, pitch PITCH). 16 consecutive lanes write 16
// consecutive f64: conflict free.
template <int NB>
__device__ __forceinline__ void cmat_to_lds(const CMat<NB>& m, double* lre, double* lim) {
    typedef Geo<NB> G;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int ti = 0; ti < NB; ++ti)
#pragma unroll
        for (int tj = 0; tj < NB; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (16 * ti + 4 * r + q) * G::PITCH + 16 * tj + c;
                lre[off] = m.re[ti][tj][r];
                lim[off] = m.im[ti][tj][r];
            }
}

// planar LDS slot -> column-major HBM image, through R-layout registers.
template <int NB>
__device__ __forceinline__ void lds_to_image(const double* lre, const double* lim, double2* img) {
    typedef Geo<NB> G;
    const int lane = lane_id(), i = lane % G::NP, h = lane / G::NP;
#pragma unroll
    for (int cc = 0; cc < G::CPL; ++cc) {
        const int off = i * G::PITCH + cc * G::H + h;
        img[cc * 64 + lane] = make_double2(lre[off], lim[off]);
    }
}

// acc += A * B. A from the planar LDS slot, B given per (k-step, column tile) by `bf`.
// A fragment of v_mfma_f64_16x16x4_f64: lane (q,c) holds A[16 ti + c][4 kk + q]; with
// PITCH = NP + 2 the 32 lanes of a ds_read_b64 group hit 32 distinct bank pairs.
template <int NB, class BFrag>
__device__ __forceinline__ void zgemm_acc(CMat<NB>& acc, const double* lre, const double* lim,
                                          BFrag bf) {
    typedef Geo<NB> G;
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int kk = 0; kk < 4 * NB; ++kk) {
        double are[NB], aim[NB], nim[NB];
#pragma unroll
        for (int ti = 0; ti < NB; ++ti) {
            const int off = (16 * ti + c) * G::PITCH + 4 * kk + q;
            are[ti] = lre[off];
            aim[ti] = lim[off];
            nim[ti] = -aim[ti];
        }
#pragma unroll
        for (int tj = 0; tj < NB; ++tj) {
            double bre, bim;
            bf(kk, tj, bre, bim);
#pragma unroll
            for (int ti = 0; ti < NB; ++ti) {
                acc.re[ti][tj] = mfma_f64(are[ti], bre, acc.re[ti][tj]);
                acc.re[ti][tj] = mfma_f64(nim[ti], bim, acc.re[ti][tj]);
                acc.im[ti][tj] = mfma_f64(are[ti], bim, acc.im[ti][tj]);
                acc.im[ti][tj] = mfma_f64(aim[ti], bre, acc.im[ti][tj]);
            }
        }
    }
}

// u_k(t_mid): the reference's formula y1 + ((y2 - y1)/(x2 - x1)) * (x3 - x1), mathmethods.py:33.
__device__ __forceinline__ double control_at(const double* ctl_b, const StepInterp& si, int K,
                                              int k) {
    const double y1 = ctl_b[(size_t)si.i1 * K + k];
    const double y2 = ctl_b[(size_t)si.i2 * K + k];
    return y1 + (((y2 - y1) / si.dx) * si.off);
}


// In-kernel stamps (diagnostic builds of a kernel only: STAMP = true, knobs "sweep3_stamps" /
// "lindblad_stamps"; the product kernels execute none). Each role (wave) accumulates shader-clock cycles per phase of its loop
// in scalar registers and stores the sums once at the end: args.stamps[seed][role][8] (+ the
// 100 MHz real-time counter in slot 7, which gives the clock). cdna_hip_programming.md section 7.
template <bool STAMP>
struct StampClock {
    unsigned long long last, acc[8];
    __device__ __forceinline__ void start() {
        if constexpr (STAMP) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0;
            __builtin_amdgcn_sched_barrier(0);
            last = __builtin_amdgcn_s_memtime();
            acc[7] = __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __device__ __forceinline__ void lap(int k) {  // k is a literal at every call site
        if constexpr (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long now;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
            acc[k] += now - last;
            last = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // stamps[workgroup][roles][8] += this wave's sums
    __device__ __forceinline__ void finish(unsigned long long* stamps, int roles, int role) {
        if constexpr (STAMP) {
            acc[7] = __builtin_amdgcn_s_memrealtime() - acc[7];
            if (lane_id() == 0 && stamps != nullptr)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    atomicAdd(stamps + ((size_t)blockIdx.x * roles + role) * 8 + k, acc[k]);
        }
    }
};


}  // namespace qocx

#endif
