// qocx_wave.h - wave-level primitives, layouts and the LDS-staged complex GEMM on
// v_mfma_f64_16x16x4_f64 shared by the Schroedinger (qocx_kernels.hip) and Lindblad
// (qocx_lindblad.hip) kernels. See qocx_kernels.hip for the layout conventions.
#ifndef QOCX_WAVE_H
#define QOCX_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "qocx_device.h"

namespace qocx {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __constant__ const double PADE_B[14] = {
    64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
    129060195264000.0,   10559470521600.0,    670442572800.0,     33522128640.0,
    1323241920.0,        40840800.0,          960960.0,           16380.0,
    182.0,               1.0};

#define QOCX_THETA13 5.371920351148152

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
