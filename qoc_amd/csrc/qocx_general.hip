// qocx_general.hip - the Schroedinger path for Hilbert sizes ABOVE 64 (65 <= n <= 256, M2): the same
// algorithm as the wavefront kernels - Pade numerator / denominator per step, the
// sweep applies P^-1 Q per squaring sub-step, the Krylov-chain adjoint forms the generator cotangent -
// with every matrix in HBM / L2 instead of registers and LDS, one workgroup of four waves per work item.
// The reference is unbounded in n (qoc/core/schroedingerdiscrete.py:356-502, its report has rows to
// n = 1024); this file turns the hard error above n = 64 into a working path. It is NOT tuned like the
// n <= 64 kernels: row-major padded matrices, complex GEMMs on the matrix cores staged through LDS, an explicit
// inverse by Gauss-Jordan with partial pivoting (so a sub-step is two matrix-vector products and no
// serial triangular solve), classic order of the evaluation (factor, forward sweep, adjoint sweep, K3).
//   reference: expm_pade qoc/standard/functions/expm.py:210-252 (orders by norm: the table :194-209),
//   costs qoc/standard/costs/targetstateinfidelity.py:52-61, forbidstates.py:64-81
#include <algorithm>

#include "qocx_wave.h"

namespace qocx {
namespace general {

constexpr int TPB = 256;

__device__ __forceinline__ void cfma(double2& acc, const double2 a, const double2 b) {  // acc += a b
    acc.x = fma(a.x, b.x, fma(-a.y, b.y, acc.x));
    acc.y = fma(a.x, b.y, fma(a.y, b.x, acc.y));
}
__device__ __forceinline__ void cfma_conj(double2& acc, const double2 a, const double2 b) {  // acc += conj(a) b
    acc.x = fma(a.x, b.x, fma(a.y, b.y, acc.x));
    acc.y = fma(a.x, b.y, fma(-a.y, b.x, acc.y));
}
__device__ __forceinline__ double2 cmul(const double2 a, const double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// sums over the workgroup (every thread calls; the result is uniform). red: 16 doubles of LDS
__device__ __forceinline__ double2 block_sum2(double a, double b, double* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[2 * (threadIdx.x >> 6)] = a;
        red[2 * (threadIdx.x >> 6) + 1] = b;
    }
    __syncthreads();
    return make_double2((red[0] + red[2]) + (red[4] + red[6]), (red[1] + red[3]) + (red[5] + red[7]));
}
__device__ __forceinline__ double block_max(double a, double* red) {
    a = wave_max(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// y[r] = sum_c op(M[r][c]) v[c]: a wave per row, lanes along the row (coalesced), one reduction per row.
// v, y in LDS (y != v); ends with a barrier.
template <bool CONJ>
__device__ __forceinline__ void matvec_rows(const double2* __restrict__ M, const double2* v, double2* y, int np) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // four rows of the wave at a time: their loads are in flight together (one row per turn was a trip to memory per
    // row - 31 us per 128 x 128 product), their reductions interleave
    for (int r0 = w; r0 < np; r0 += 16) {
        double2 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = make_double2(0, 0);
        for (int c = lane; c < np; c += 64) {
            const double2 vc = v[c];
            double2 e[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                e[i] = (r0 + 4 * i < np) ? M[(size_t)(r0 + 4 * i) * np + c] : make_double2(0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (CONJ) cfma_conj(acc[i], e[i], vc);
                else cfma(acc[i], e[i], vc);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double sr = wave_sum(acc[i].x), si = wave_sum(acc[i].y);
            if (lane == 0 && r0 + 4 * i < np) y[r0 + 4 * i] = make_double2(sr, si);
        }
    }
    __syncthreads();
}

// y[c] = sum_r op(M[r][c]) v[r]: a lane per column, rows split over the waves that are left; partial sums
// through `part` ([4][np] in LDS). v, y in LDS (y != v); ends with a barrier.
template <bool CONJ>
__device__ __forceinline__ void matvec_cols(const double2* __restrict__ M, const double2* v, double2* y, double2* part,
                                            int np) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = (np + 63) >> 6, groups = chunks >= 4 ? 1 : 4 / chunks;
    // (more than four chunks of 64 columns: a wave takes several, every row itself)
    for (int ch = (chunks >= 4 ? w : w % chunks); ch < chunks; ch += (chunks >= 4 ? 4 : chunks)) {
        const int grp = chunks >= 4 ? 0 : w / chunks, c = ch * 64 + lane;
        if (grp < groups && c < np) {
            double2 acc = make_double2(0, 0);
#pragma unroll 8
            for (int r = grp; r < np; r += groups) {
                if (CONJ) cfma_conj(acc, M[(size_t)r * np + c], v[r]);
                else cfma(acc, M[(size_t)r * np + c], v[r]);
            }
            part[grp * np + c] = acc;
        }
        if (chunks < 4) break;
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < np; cc += TPB) {
        double2 s = part[cc];
        for (int g = 1; g < groups; ++g) {
            s.x += part[g * np + cc].x;
            s.y += part[g * np + cc].y;
        }
        y[cc] = s;
    }
    __syncthreads();
}

// Two of them on one pass over M: y1[c] = sum_r op(M[r][c]) v1[r], y2 likewise from v2 (part: [2][groups][np]).
template <bool CONJ>
__device__ __forceinline__ void matvec_cols2(const double2* __restrict__ M, const double2* v1, double2* y1, const double2* v2,
                                             double2* y2, double2* part, int np) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = (np + 63) >> 6, groups = chunks >= 4 ? 1 : 4 / chunks;
    double2* part2 = part + groups * np;
    for (int ch = (chunks >= 4 ? w : w % chunks); ch < chunks; ch += (chunks >= 4 ? 4 : chunks)) {
        const int grp = chunks >= 4 ? 0 : w / chunks, c = ch * 64 + lane;
        if (grp < groups && c < np) {
            double2 acc1 = make_double2(0, 0), acc2 = make_double2(0, 0);
#pragma unroll 8
            for (int r = grp; r < np; r += groups) {
                const double2 e = M[(size_t)r * np + c];
                if (CONJ) {
                    cfma_conj(acc1, e, v1[r]);
                    cfma_conj(acc2, e, v2[r]);
                } else {
                    cfma(acc1, e, v1[r]);
                    cfma(acc2, e, v2[r]);
                }
            }
            part[grp * np + c] = acc1;
            part2[grp * np + c] = acc2;
        }
        if (chunks < 4) break;
    }
    __syncthreads();
    for (int cc = threadIdx.x; cc < np; cc += TPB) {
        double2 s1 = part[cc], s2 = part2[cc];
        for (int g = 1; g < groups; ++g) {
            s1.x += part[g * np + cc].x;
            s1.y += part[g * np + cc].y;
            s2.x += part2[g * np + cc].x;
            s2.y += part2[g * np + cc].y;
        }
        y1[cc] = s1;
        y2[cc] = s2;
    }
    __syncthreads();
}

// C = A B (row-major np x np in HBM / L2, np a multiple of 16) on the matrix cores: 64 x 64 output tiles, a
// wave owns a 32 x 32 quadrant (2 x 2 tiles of v_mfma_f64_16x16x4_f64, complex = 4 real products per tile and
// k-step), operands staged 16 columns at a time through LDS (34 KiB at smem): lane (q, c) reads its A fragment
// A[16 i + c][4 kk + q] and its B fragment B[4 kk + q][16 j + c] as one 16-byte complex number each.
struct GemmLds {
    double2 as[64][17];
    double2 bs[16][64];
};
// gemm_op<TRANSB, CONJB>: C (mrows x np) = A (mrows x np) op(B), op(B)[k][j] = B[j][k] (TRANSB) or conj(B[k][j])
// (CONJB) - the many-state sweep: the states of a seed as the rows of A (mrows = S, any number).
// TRANSA (K3 of many states): C (np x np) = A^T conj'd per CONJB ... = sum over the kdim rows k of A[k][r] op(B)[k][c]
// (A, B: kdim x np; any kdim); ACC: added to what C holds.
template <bool TRANSA, bool TRANSB, bool CONJB, bool ACC>
__device__ __noinline__ void gemm_op(const double2* __restrict__ A, const double2* __restrict__ B, double2* __restrict__ C,
                                     int mrows, int np, int kdim, char* smem) {
    GemmLds& L = *reinterpret_cast<GemmLds*>(smem);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, q = lane >> 4, c = lane & 15;
    const int wr = (w >> 1) * 32, wc = (w & 1) * 32;
    for (int r0 = 0; r0 < mrows; r0 += 64)
        for (int c0 = 0; c0 < np; c0 += 64) {
            const bool vi[2] = {r0 + wr < mrows, r0 + wr + 16 < mrows}, vj[2] = {c0 + wc < np, c0 + wc + 16 < np};
            d4 re[2][2], im[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) re[i][j] = im[i][j] = d4{0, 0, 0, 0};
            // the next 16 columns travel from memory to registers while this chunk's products run (the factor
            // kernel, two workgroups to a CU, gains nothing from it; the many-state sweep - ONE workgroup per
            // seed, a chain of small products - is a chain of these waits)
            double2 pa[4], pb[4];
            auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int idx = tid + TPB * qq;
                    const int rr = idx >> 4, kk = idx & 15;
                    if (TRANSA) {  // element (r0 + ra, k0 + ka) of A^T = A[k0 + ka][r0 + ra]: runs along the row index
                        const int ka = idx >> 6, ra = idx & 63;
                        pa[qq] = (r0 + ra < mrows && k0 + ka < kdim) ? A[(size_t)(k0 + ka) * np + r0 + ra] : make_double2(0, 0);
                    } else {
                        pa[qq] = (r0 + rr < mrows) ? A[(size_t)(r0 + rr) * np + k0 + kk] : make_double2(0, 0);
                    }
                    if (TRANSB) {  // element (k0 + kk, c0 + rr) of op(B) = B[c0 + rr][k0 + kk]: 256-byte runs along k
                        pb[qq] = (c0 + rr < np) ? B[(size_t)(c0 + rr) * np + k0 + kk] : make_double2(0, 0);
                    } else {
                        const int kb = idx >> 6, cc = idx & 63;
                        pb[qq] = (c0 + cc < np && k0 + kb < kdim) ? B[(size_t)(k0 + kb) * np + c0 + cc] : make_double2(0, 0);
                        if (CONJB) pb[qq].y = -pb[qq].y;
                    }
                }
            };
            fetch(0);
            for (int k0 = 0; k0 < kdim; k0 += 16) {
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const int idx = tid + TPB * qq;
                    if (TRANSA) L.as[idx & 63][idx >> 6] = pa[qq];
                    else L.as[idx >> 4][idx & 15] = pa[qq];
                    if (TRANSB) L.bs[idx & 15][idx >> 4] = pb[qq];
                    else L.bs[idx >> 6][idx & 63] = pb[qq];
                }
                __syncthreads();
                if (k0 + 16 < kdim) fetch(k0 + 16);
                if (vi[0] && vj[0]) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        double2 af[2], bf[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) af[i] = L.as[wr + 16 * i + c][4 * kk + q];
#pragma unroll
                        for (int j = 0; j < 2; ++j) bf[j] = L.bs[4 * kk + q][wc + 16 * j + c];
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (vi[i])
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    if (vj[j]) {
                                        re[i][j] = mfma_f64(af[i].x, bf[j].x, re[i][j]);
                                        re[i][j] = mfma_f64(-af[i].y, bf[j].y, re[i][j]);
                                        im[i][j] = mfma_f64(af[i].x, bf[j].y, im[i][j]);
                                        im[i][j] = mfma_f64(af[i].y, bf[j].x, im[i][j]);
                                    }
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (vi[i] && vj[j])
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (r0 + wr + 16 * i + 4 * r + q < mrows) {
                                double2* dst = C + (size_t)(r0 + wr + 16 * i + 4 * r + q) * np + c0 + wc + 16 * j + c;
                                double2 v = make_double2(re[i][j][r], im[i][j][r]);
                                if (ACC) {
                                    v.x += dst->x;
                                    v.y += dst->y;
                                }
                                *dst = v;
                            }
        }
    __syncthreads();
}
__device__ __forceinline__ void gemm(const double2* __restrict__ A, const double2* __restrict__ B, double2* __restrict__ C,
                                     int np, char* smem) {
    gemm_op<false, false, false, false>(A, B, C, np, np, np, smem);
}

// M := M^-1 in place (row-major np x np in HBM / L2): Gauss-Jordan with partial pivoting (largest
// |re| + |im| of the column, LAPACK's izamax measure; ties to the smaller row), BLOCKED by KB pivots so that the
// matrix crosses the memory system np / KB times instead of np times: the KB columns of a block are
// eliminated in LDS (thread r owns row r), their row interchanges are applied to the other columns, and every
// other column takes the block's KB elimination steps as ONE rank-KB update
//   X <- Y + (G - E)(E^T Y),  Y = the interchanged X,  G = the eliminated panel,  E = the block's unit columns
// (the steps T_j = I + w_j e_kj^T commute past the later interchanges with w_j interchanged along, which is
// what swapping whole panel rows in LDS does). The column interchanges are undone at the end in one pass.
// false (uniform): a zero / non-finite pivot.
// (np > 256: the block's pivot rows R wait in global scratch `rg` instead of LDS, a thread owns several rows of the
// panel; np > 512 - up to 1024 -: blocks of 8 pivots, so that the panel still fits the LDS)
__host__ __device__ constexpr int invert_kb(int np) { return np > 512 ? 8 : 16; }
__host__ __device__ constexpr int invert_lds(int np) {
    return np * (invert_kb(np) + 1) * 16 + (np <= 256 ? invert_kb(np) * np * 16 : 0) + 8 * np + 128;
}
template <int KB>
__device__ __noinline__ bool invert_kb_body(double2* __restrict__ M, int np, char* smem, double2* __restrict__ rg) {
    constexpr int GP = KB + 1, ROWS = KB == 16 ? 2 : 4;  // rows of the panel per thread
    double2* G = reinterpret_cast<double2*>(smem);  // [np][GP]
    double2* R = np <= 256 ? G + (size_t)np * GP : rg;  // [KB][np]
    int* piv = reinterpret_cast<int*>(G + (size_t)np * GP + (np <= 256 ? (size_t)KB * np : 0));
    int* idx = piv + np;
    double* red = reinterpret_cast<double*>(idx + np);
    int* redi = reinterpret_cast<int*>(red + 8);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    bool ok = true, any_swap = false;
    for (int k0 = 0; k0 < np; k0 += KB) {
        for (int e = tid; e < np * KB; e += TPB) {
            const int r = e / KB, j = e - r * KB;
            G[r * GP + j] = M[(size_t)r * np + k0 + j];
        }
        __syncthreads();
        // rows tid, tid + TPB, ... of the panel (np <= ROWS TPB)
        for (int j = 0; j < KB; ++j) {
            const int k = k0 + j;
            double best = -1.0;
            int bi = k;
#pragma unroll
            for (int h = 0; h < ROWS; ++h) {
                const int r = tid + h * TPB;
                if (r < np && r >= k) {
                    const double2 e = G[r * GP + j];
                    double v = fabs(e.x) + fabs(e.y);
                    v = (v == v) ? v : 1e308;
                    if (v > best) {
                        best = v;
                        bi = r;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (lane == 0) {
                red[w] = best;
                redi[w] = bi;
            }
            __syncthreads();
            best = red[0];
            int p = redi[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (red[q] > best || (red[q] == best && redi[q] < p)) {
                    best = red[q];
                    p = redi[q];
                }
            if (!(best > 0.0) || !(best < 1e300)) ok = false;  // singular or not finite: finish without dividing by it
            if (tid == 0) piv[k] = p;
            if (p != k) {
                any_swap = true;
                if (tid < KB) {
                    const double2 t = G[k * GP + tid];
                    G[k * GP + tid] = G[p * GP + tid];
                    G[p * GP + tid] = t;
                }
            }
            __syncthreads();
            double2 inv = make_double2(1.0, 0.0);
            {
                const double2 d = G[k * GP + j];
                const double den = d.x * d.x + d.y * d.y;
                if (ok && den > 0.0) inv = make_double2(d.x / den, -d.y / den);
            }
            double2 f[ROWS];
#pragma unroll
            for (int h = 0; h < ROWS; ++h) {
                const int r = tid + h * TPB;
                f[h] = (r < np && r != k) ? G[r * GP + j] : make_double2(0, 0);
            }
            __syncthreads();
            if (tid < KB) G[k * GP + tid] = (tid == j) ? inv : cmul(G[k * GP + tid], inv);
            __syncthreads();
#pragma unroll
            for (int h = 0; h < ROWS; ++h) {
                const int r = tid + h * TPB;
                if (r < np && r != k) {
                    const double2 nf = make_double2(-f[h].x, -f[h].y);
#pragma unroll
                    for (int jj = 0; jj < KB; ++jj) {
                        const double2 pk = G[k * GP + jj];
                        if (jj == j) {
                            G[r * GP + jj] = cmul(nf, pk);
                        } else {
                            double2 v = G[r * GP + jj];
                            cfma(v, nf, pk);
                            G[r * GP + jj] = v;
                        }
                    }
                }
            }
            __syncthreads();
        }
        // the block's row interchanges on the other columns, in their order
        for (int j = 0; j < KB; ++j) {
            const int k = k0 + j, p = piv[k];
            if (p != k) {
                for (int c = tid; c < np; c += TPB)
                    if (c < k0 || c >= k0 + KB) {
                        const double2 t = M[(size_t)k * np + c];
                        M[(size_t)k * np + c] = M[(size_t)p * np + c];
                        M[(size_t)p * np + c] = t;
                    }
                __syncthreads();
            }
        }
        for (int e = tid; e < KB * np; e += TPB) R[e] = M[(size_t)k0 * np + e];  // rows k0 .. k0 + KB - 1
        __syncthreads();
        {
            // the rank-KB update on the matrix cores, a 16 x 16 tile of the matrix per wave and turn: the tile (zero
            // in the block's own rows, untouched in its own columns) is the accumulator, G[16 ti .., 0..KB-1] the A
            // operand, R the B one
            const int q = lane >> 4, c = lane & 15, nt = np >> 4;
            for (int t = w; t < nt * nt; t += 4) {
                const int ti = t / nt, tj = t - ti * nt;
                if (KB == 16 && 16 * tj == k0) continue;
                const size_t base = (size_t)(16 * ti + q) * np + 16 * tj + c;  // element (16 ti + 4 r + q, 16 tj + c)
                const bool own_col = 16 * tj + c >= k0 && 16 * tj + c < k0 + KB;
                d4 re = d4{0, 0, 0, 0}, im = d4{0, 0, 0, 0};
                if constexpr (KB == 16) {  // (a block is a whole tile row / tile column: decided per tile)
                    if (16 * ti != k0) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const double2 e = M[base + (size_t)4 * rr * np];
                            re[rr] = e.x;
                            im[rr] = e.y;
                        }
                    }
                } else {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int row = 16 * ti + 4 * rr + q;
                        if (!(row >= k0 && row < k0 + KB)) {
                            const double2 e = M[base + (size_t)4 * rr * np];
                            re[rr] = e.x;
                            im[rr] = e.y;
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < KB / 4; ++kk) {
                    const double2 a = G[(16 * ti + c) * GP + 4 * kk + q];
                    const double2 b = R[(4 * kk + q) * np + 16 * tj + c];
                    re = mfma_f64(a.x, b.x, re);
                    re = mfma_f64(-a.y, b.y, re);
                    im = mfma_f64(a.x, b.y, im);
                    im = mfma_f64(a.y, b.x, im);
                }
                if (KB == 16 || !own_col)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) M[base + (size_t)4 * rr * np] = make_double2(re[rr], im[rr]);
            }
            if (KB != 16) __syncthreads();  // (KB = 8: a tile column holds the block's own columns beside eight others - all tiles first)
            for (int e = tid; e < np * KB; e += TPB) {  // the block's own columns: the eliminated panel
                const int rr = e / KB, j = e - rr * KB;
                M[(size_t)rr * np + k0 + j] = G[rr * GP + j];
            }
        }
        __syncthreads();
    }
    if (any_swap) {
        // column c of the inverse is the column that the interchanges, undone from the last to the first, leave there
        if (tid == 0) {
            for (int c = 0; c < np; ++c) idx[c] = c;
            for (int k = np - 1; k >= 0; --k) {
                const int p = piv[k], t = idx[k];
                idx[k] = idx[p];
                idx[p] = t;
            }
        }
        __syncthreads();
        for (int r0 = 0; r0 < np; r0 += KB) {
            for (int e = tid; e < KB * np; e += TPB) R[e] = M[(size_t)r0 * np + e];
            __syncthreads();
            for (int e = tid; e < KB * np; e += TPB) {
                const int j = e / np, c = e - j * np;
                M[(size_t)r0 * np + e] = R[j * np + idx[c]];
            }
            __syncthreads();
        }
    }
    return ok;
}

__device__ __forceinline__ bool invert(double2* __restrict__ M, int np, char* smem, double2* __restrict__ rg) {
    return np > 512 ? invert_kb_body<8>(M, np, smem, rg) : invert_kb_body<16>(M, np, smem, rg);
}

// ---- K1a + K1b ---------------------------------------------------------------------------------------
// One work item = one propagator step of one seed: generator, 1-norm, order / squarings, the Pade
// polynomials, Q to q_img, P^-1 to pinv_img, the step's entry of s_arr. Persistent workgroups (grid-stride)
// with 7 scratch matrices each.
__global__ __launch_bounds__(TPB, 2) void factor_kernel(GeneralArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int np = a.np, mat = np * np, tid = threadIdx.x;
    double* red = reinterpret_cast<double*>(smem + (sizeof(GemmLds) > (size_t)invert_lds(np) ? sizeof(GemmLds) : (size_t)invert_lds(np)));
    double2* sc = a.scratch + (size_t)blockIdx.x * 7 * mat;
    double2 *A = sc, *A2 = sc + mat, *A4 = sc + 2 * (size_t)mat, *A6 = sc + 3 * (size_t)mat, *X = sc + 4 * (size_t)mat,
            *Y = sc + 5 * (size_t)mat, *Z = sc + 6 * (size_t)mat;
    for (size_t wi = blockIdx.x; wi < a.total; wi += gridDim.x) {
        const int b = (int)(wi / a.nsteps), step = (int)(wi % a.nsteps);
        const size_t m = (size_t)b * a.nsteps + step;
        // generator a = -i dt H(u(t_mid), t_mid) (schroedingerdiscrete.py:483-487, mathmethods.py:72), or
        // the host's sample of it (explicit mode)
        if (a.gen_rm != nullptr) {
            const double2* g = a.gen_rm + m * mat;
            for (int e = tid; e < mat; e += TPB) A[e] = g[e];
        } else {
            const size_t tsel = (a.nt == 1) ? 0 : (size_t)step;
            const double2* h0 = a.h0_rm + tsel * mat;
            const double2* gk = a.g_rm + tsel * a.K * mat;
            const double* ctl_b = a.controls + (size_t)b * a.nc * a.K;
            const StepInterp si = a.interp[step];
            for (int e = tid; e < mat; e += TPB) {
                double2 h = h0[e];
                for (int k = 0; k < a.K; ++k) {
                    const double uk = control_at(ctl_b, si, a.K, k);
                    const double2 g = gk[(size_t)k * mat + e];
                    h.x = fma(uk, g.x, h.x);
                    h.y = fma(uk, g.y, h.y);
                }
                A[e] = make_double2(a.dt * h.y, -a.dt * h.x);
            }
        }
        __syncthreads();
        // 1-norm (expm.py:116), order and squarings (expm.py:238-241; order by norm: qocx_wave.h)
        double colmax = 0.0;
        for (int c = tid; c < np; c += TPB) {
            double s = 0.0;
            for (int r = 0; r < np; ++r) {
                const double2 e = A[(size_t)r * np + c];
                s += sqrt(e.x * e.x + e.y * e.y);
            }
            colmax = (s > colmax || !(s == s)) ? ((s == s) ? s : 1e308) : colmax;
        }
        const double norm1 = block_max(colmax, red);
        int sq = 0, order = pade_order_for(norm1, a.pade_policy);
        {
            double th = QOCX_THETA13;
            while (norm1 > th && sq < 30) {
                th *= 2.0;
                ++sq;
            }
            if (!(norm1 <= th)) {
                if (tid == 0) atomicOr(a.status, 2);
                sq = 0;
                order = 13;
            }
            if (sq > a.sq_max) {  // (cannot happen: the host's bound is above every step's norm)
                if (tid == 0) atomicOr(a.status, 4);
                sq = a.sq_max;
            }
        }
        if (sq > 0) {
            const double scale = ldexp(1.0, -sq);
            for (int e = tid; e < mat; e += TPB) {
                double2 v = A[e];
                v.x *= scale;
                v.y *= scale;
                A[e] = v;
            }
            __syncthreads();
        }
        if (tid == 0) a.s_arr[m] = step_entry(sq, order);
        const double* bt = pade_table(order);
        double2* Q = a.q_img + m * mat;
        double2* P = a.pinv_img + m * mat;
        gemm(A, A, A2, np, smem);
        if (order == 13) {
            gemm(A2, A2, A4, np, smem);
            gemm(A2, A4, A6, np, smem);
            for (int e = tid; e < mat; e += TPB) {
                const double2 x2 = A2[e], x4 = A4[e], x6 = A6[e];
                X[e] = make_double2(bt[13] * x6.x + bt[11] * x4.x + bt[9] * x2.x, bt[13] * x6.y + bt[11] * x4.y + bt[9] * x2.y);
            }
            __syncthreads();
            gemm(A6, X, Y, np, smem);
            for (int e = tid; e < mat; e += TPB) {
                const double2 x2 = A2[e], x4 = A4[e], x6 = A6[e];
                double2 y = Y[e];
                y.x += bt[7] * x6.x + bt[5] * x4.x + bt[3] * x2.x;
                y.y += bt[7] * x6.y + bt[5] * x4.y + bt[3] * x2.y;
                if (e / np == e % np) y.x += bt[1];
                Y[e] = y;
            }
            __syncthreads();
            gemm(A, Y, X, np, smem);  // X = U (the odd part)
            for (int e = tid; e < mat; e += TPB) {
                const double2 x2 = A2[e], x4 = A4[e], x6 = A6[e];
                Y[e] = make_double2(bt[12] * x6.x + bt[10] * x4.x + bt[8] * x2.x, bt[12] * x6.y + bt[10] * x4.y + bt[8] * x2.y);
            }
            __syncthreads();
            gemm(A6, Y, Z, np, smem);
            for (int e = tid; e < mat; e += TPB) {
                const double2 x2 = A2[e], x4 = A4[e], x6 = A6[e], u = X[e];
                double2 v = Z[e];
                v.x += bt[6] * x6.x + bt[4] * x4.x + bt[2] * x2.x;
                v.y += bt[6] * x6.y + bt[4] * x4.y + bt[2] * x2.y;
                if (e / np == e % np) v.x += bt[0];
                Q[e] = make_double2(v.x + u.x, v.y + u.y);
                P[e] = make_double2(v.x - u.x, v.y - u.y);
            }
        } else {
            // orders 3, 5, 7, 9: u = a (sum b_{2j+1} a^{2j}), v = sum b_{2j} a^{2j}; A4, A6, A8 (in Z) as needed
            if (order >= 5) gemm(A2, A2, A4, np, smem);
            if (order >= 7) gemm(A2, A4, A6, np, smem);
            if (order >= 9) gemm(A4, A4, Z, np, smem);
            for (int e = tid; e < mat; e += TPB) {
                double2 y = make_double2(0, 0);
                const double2 x2 = A2[e];
                y.x = bt[3] * x2.x;
                y.y = bt[3] * x2.y;
                if (order >= 5) { const double2 x4 = A4[e]; y.x += bt[5] * x4.x; y.y += bt[5] * x4.y; }
                if (order >= 7) { const double2 x6 = A6[e]; y.x += bt[7] * x6.x; y.y += bt[7] * x6.y; }
                if (order >= 9) { const double2 x8 = Z[e]; y.x += bt[9] * x8.x; y.y += bt[9] * x8.y; }
                if (e / np == e % np) y.x += bt[1];
                Y[e] = y;
            }
            __syncthreads();
            gemm(A, Y, X, np, smem);
            for (int e = tid; e < mat; e += TPB) {
                const double2 x2 = A2[e], u = X[e];
                double2 v = make_double2(bt[2] * x2.x, bt[2] * x2.y);
                if (order >= 5) { const double2 x4 = A4[e]; v.x += bt[4] * x4.x; v.y += bt[4] * x4.y; }
                if (order >= 7) { const double2 x6 = A6[e]; v.x += bt[6] * x6.x; v.y += bt[6] * x6.y; }
                if (order >= 9) { const double2 x8 = Z[e]; v.x += bt[8] * x8.x; v.y += bt[8] * x8.y; }
                if (e / np == e % np) v.x += bt[0];
                Q[e] = make_double2(v.x + u.x, v.y + u.y);
                P[e] = make_double2(v.x - u.x, v.y - u.y);
            }
        }
        __syncthreads();
        if (!invert(P, np, smem, Z))  // (Z: free by now - the pivot rows of a block wait there above np = 256)
            if (tid == 0) atomicOr(a.status, 1);
        __syncthreads();
    }
}

// ---- K2 ------------------------------------------------------------------------------------------------
// inner product <t|v> over the workgroup (uniform)
__device__ __forceinline__ double2 inner_g(const double2* t, const double2* v, int np, double* red) {
    double pr = 0, pi = 0;
    for (int i = threadIdx.x; i < np; i += TPB) {
        const double2 a = t[i], p = v[i];
        pr += a.x * p.x + a.y * p.y;  // conj(t) v
        pi += a.x * p.y - a.y * p.x;
    }
    return block_sum2(pr, pi, red);
}

// The selected costs on the S states at `vecs` ([S][np], HBM); lam != nullptr: += dC/dRe + i dC/dIm
// (formulas and cotangents as eval_costs of qocx_sweep_common.h)
__device__ __forceinline__ double eval_costs_g(const GeneralSweepArgs& args, bool step_pass, bool final_pass,
                                               const double2* vecs, double2* lam, double* red) {
    const int np = args.np, S = args.S;
    double total = 0;
    for (int ci = 0; ci < args.cost_count; ++ci) {
        const DevCost c = args.costs[ci];
        const bool on = c.step_cost ? step_pass : final_pass;
        if (!on) continue;
        const double2* pool = args.cost_vectors + (size_t)c.vec_offset * np;
        if (c.kind == QOCX_DEV_COST_COHERENT) {
            double tre = 0, tim = 0;
            for (int s = 0; s < S; ++s) {
                const double2 ip = inner_g(pool + (size_t)s * np, vecs + (size_t)s * np, np, red);
                tre += ip.x;
                tim += ip.y;
            }
            total += c.scale * (1.0 - (tre * tre + tim * tim) / ((double)S * S));
            if (lam != nullptr) {
                const double f = -2.0 * c.scale / ((double)S * S);
                for (int s = 0; s < S; ++s)
                    for (int i = threadIdx.x; i < np; i += TPB) {
                        const double2 t = pool[(size_t)s * np + i];
                        double2 l = lam[(size_t)s * np + i];
                        l.x += f * (tre * t.x - tim * t.y);
                        l.y += f * (tre * t.y + tim * t.x);
                        lam[(size_t)s * np + i] = l;
                    }
            }
        } else if (c.kind == QOCX_DEV_COST_INCOHERENT) {
            double fid = 0;
            const double f = -2.0 * c.scale / (double)S;
            for (int s = 0; s < S; ++s) {
                const double2 ip = inner_g(pool + (size_t)s * np, vecs + (size_t)s * np, np, red);
                fid += ip.x * ip.x + ip.y * ip.y;
                if (lam != nullptr)
                    for (int i = threadIdx.x; i < np; i += TPB) {
                        const double2 t = pool[(size_t)s * np + i];
                        double2 l = lam[(size_t)s * np + i];
                        l.x += f * (ip.x * t.x - ip.y * t.y);
                        l.y += f * (ip.x * t.y + ip.y * t.x);
                        lam[(size_t)s * np + i] = l;
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
                    const double2* t = pool + (size_t)(base + f) * np;
                    const double2 ip = inner_g(t, vecs + (size_t)s * np, np, red);
                    acc += w * (ip.x * ip.x + ip.y * ip.y);
                    if (lam != nullptr) {
                        const double g = 2.0 * c.scale * w;
                        for (int i = threadIdx.x; i < np; i += TPB) {
                            const double2 tv = t[i];
                            double2 l = lam[(size_t)s * np + i];
                            l.x += g * (ip.x * tv.x - ip.y * tv.y);
                            l.y += g * (ip.x * tv.y + ip.y * tv.x);
                            lam[(size_t)s * np + i] = l;
                        }
                    }
                }
                base += fs;
            }
            total += c.scale * acc;
        }
    }
    __syncthreads();
    return total;
}

// One workgroup per seed: forward sweep (phase bit 0) over all steps, adjoint sweep (bit 1) back.
// The loop of schroedingerdiscrete.py:393-436 with psi' = (P^-1 Q)^(2^s) psi per step (expm.py:246-250).
// (two waves per SIMD for every caller of the product functions: their registers are allocated once, for the loosest caller)
__global__ __launch_bounds__(TPB, 2) void sweep_kernel(GeneralSweepArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int np = args.np, S = args.S, nsteps = args.nsteps, tid = threadIdx.x, b = blockIdx.x;
    const int mat = np * np;
    double2* v0 = reinterpret_cast<double2*>(smem);
    double2* v1 = v0 + np;
    double2* v2 = v1 + np;
    double2* part = v2 + np;  // [4][np]
    double* red = reinterpret_cast<double*>(part + 4 * np);
    const size_t cap = args.slot_cap;
    double2* states_b = args.states + (size_t)b * cap * S * np;
    double2* xs_b = args.xs + (size_t)b * cap * S * np;
    int* offs_b = args.offs + (size_t)b * (nsteps + 1);
    double2* lam = args.lam_buf + (size_t)b * S * np;  // (forward sweep, many states: the scratch between its two products)
    // eight states or more: the sweep as products on the matrix cores (a 16-row tile is then at least half full)
    const bool many = S >= 8;
    char* gsm = smem + 7 * np * 16 + 256;
    // Split mode (phase bit 3; many states, final costs only): the states of a seed do not meet between the
    // costs, so gridDim.y workgroups take a group of rows each - a full propagator of ONE control set is
    // otherwise a chain of small products on one CU. The costs and the seed of lambda are a launch of their own
    // (phase bit 2, one workgroup per seed) between the forward launch and the adjoint one.
    const bool split = (args.phase & 8) != 0;
    const int per = split ? (S + (int)gridDim.y - 1) / (int)gridDim.y : S;
    const int s0 = split ? (int)blockIdx.y * per : 0, cnt = min(per, S - s0);
    if (cnt <= 0) return;
    const size_t roff = (size_t)s0 * np;  // the group's first row inside a slot

    if (args.phase & 1) {
        for (int e = tid; e < cnt * np; e += TPB) states_b[roff + e] = args.psi0[roff + e];
        __syncthreads();
        int slot = 0;
        double cost = 0;
        bool overflow = false;
        for (int step = 0; step <= nsteps; ++step) {
            const double2* cur = states_b + (size_t)slot * S * np;
            if (!split && step != 0 && step != nsteps && args.has_step_costs && (step % args.cost_eval_step) == 0)
                cost += eval_costs_g(args, true, false, cur, nullptr, red);
            if (args.step_states != nullptr)
                for (int e = tid; e < cnt * np; e += TPB)
                    args.step_states[((size_t)b * (nsteps + 1) + step) * S * np + roff + e] = cur[roff + e];
            if (tid == 0 && s0 == 0) offs_b[step] = slot;
            if (step == nsteps) break;
            const size_t m = (size_t)b * nsteps + step;
            const int nsub = 1 << step_squarings(args.s_arr[m]);
            const double2* Q = args.q_img + m * mat;
            const double2* Pi = args.pinv_img + m * mat;
            for (int sub = 0; sub < nsub; ++sub) {
                if ((size_t)slot + 1 >= cap) {
                    overflow = true;
                    break;
                }
                if (many) {
                    // the S states of the seed as the rows of a matrix: Psi' = (Psi Q^T) P^-T on the matrix cores
                    gemm_op<false, true, false, false>(states_b + (size_t)slot * S * np + roff, Q, lam + roff, cnt, np, np, gsm);
                    gemm_op<false, true, false, false>(lam + roff, Pi, states_b + (size_t)(slot + 1) * S * np + roff, cnt, np, np, gsm);
                } else {
                    for (int s = 0; s < S; ++s) {
                        for (int i = tid; i < np; i += TPB) v0[i] = states_b[((size_t)slot * S + s) * np + i];
                        __syncthreads();
                        matvec_rows<false>(Q, v0, v1, np);
                        matvec_rows<false>(Pi, v1, v2, np);
                        for (int i = tid; i < np; i += TPB) states_b[((size_t)(slot + 1) * S + s) * np + i] = v2[i];
                        __syncthreads();
                    }
                }
                ++slot;
            }
            if (overflow) break;
        }
        if (overflow) {
            if (tid == 0) atomicOr(args.status, 4);
            return;
        }
        if (!split) {
            // step costs on the final states if the last step is a cost step, then the final costs
            // (schroedingerdiscrete.py:412-416, :428-433)
            const double2* fin = states_b + (size_t)slot * S * np;
            if (args.has_step_costs && (nsteps % args.cost_eval_step) == 0)
                cost += eval_costs_g(args, true, false, fin, nullptr, red);
            cost += eval_costs_g(args, false, true, fin, nullptr, red);
            for (int e = tid; e < S * np; e += TPB) args.final_out[(size_t)b * S * np + e] = fin[e];
            if (tid == 0) args.cost_out[b] = cost;
        }
        __syncthreads();
    }
    if (args.phase & 4) {  // split mode, between the launches: final costs, final states, the seed of lambda
        const double2* fin = states_b + (size_t)offs_b[nsteps] * S * np;
        const double cost = eval_costs_g(args, false, true, fin, nullptr, red);
        for (int e = tid; e < S * np; e += TPB) args.final_out[(size_t)b * S * np + e] = fin[e];
        if (tid == 0) args.cost_out[b] = cost;
        if (args.phase & 16) {
            for (int e = tid; e < S * np; e += TPB) lam[e] = make_double2(0, 0);
            __syncthreads();
            (void)eval_costs_g(args, false, true, fin, lam, red);
        }
        return;
    }
    if (!(args.phase & 2)) return;

    auto inject = [&](int step) {
        if (args.inj_index == nullptr) return;
        const int row = args.inj_index[step];
        if (row < 0) return;
        for (int e = tid; e < S * np; e += TPB) {
            const double2 x = args.inj_bars[((size_t)b * args.inj_count + row) * S * np + e];
            double2 l = lam[e];
            l.x += x.x;
            l.y += x.y;
            lam[e] = l;
        }
        __syncthreads();
    };
    int slot = offs_b[nsteps];
    if (!split) {
        for (int e = tid; e < S * np; e += TPB) lam[e] = make_double2(0, 0);
        __syncthreads();
        (void)eval_costs_g(args, (nsteps % args.cost_eval_step) == 0, true, states_b + (size_t)slot * S * np, lam, red);
        inject(nsteps);
    }
    for (int step = nsteps - 1; step >= 0; --step) {
        const size_t m = (size_t)b * nsteps + step;
        const int nsub = 1 << step_squarings(args.s_arr[m]);
        const double2* Q = args.q_img + m * mat;
        const double2* Pi = args.pinv_img + m * mat;
        for (int sub = nsub - 1; sub >= 0; --sub) {
            --slot;
            if (many) {  // X = Lambda conj(P^-1), Lambda = X conj(Q) (rows: x_s = P^-H lambda_s, lambda_s = Q^H x_s)
                gemm_op<false, false, true, false>(lam + roff, Pi, xs_b + (size_t)slot * S * np + roff, cnt, np, np, gsm);
                gemm_op<false, false, true, false>(xs_b + (size_t)slot * S * np + roff, Q, lam + roff, cnt, np, np, gsm);
                continue;
            }
            for (int s = 0; s < S; ++s) {
                for (int i = tid; i < np; i += TPB) v0[i] = lam[(size_t)s * np + i];
                __syncthreads();
                matvec_cols<true>(Pi, v0, v1, part, np);  // x = P^-H lambda'
                matvec_cols<true>(Q, v1, v2, part, np);   // lambda = Q^H x
                for (int i = tid; i < np; i += TPB) {
                    xs_b[((size_t)slot * S + s) * np + i] = v1[i];
                    lam[(size_t)s * np + i] = v2[i];
                }
                __syncthreads();
            }
        }
        if (split) continue;  // (no step costs, no host cotangents in split mode)
        if (step != 0 && (step % args.cost_eval_step) == 0 && args.has_step_costs)
            (void)eval_costs_g(args, true, false, states_b + (size_t)slot * S * np, lam, red);
        if (step != 0) inject(step);
    }
}

// scratch of one K3 workgroup (complex elements): a, a^T, abar, and for eight states or more the chains of all states
__host__ __device__ constexpr size_t krylov_scratch_elems(int np, int S) {
    return (size_t)3 * np * np + (S >= 8 ? (size_t)28 * S * np : 0) + (np > 256 ? (size_t)26 * np : 0);
}

// ---- K3 ------------------------------------------------------------------------------------------------
// Krylov-chain adjoint of the Pade step (qocx_kernels.hip, krylov_grad_body): per (sub-step, state)
// tau_i = (a^H)^i x, rho_{M-1} = b_M sigma, rho_{i-1} = b_i w_i + a rho_i, abar += sum_i tau_i rho_i^H; then
// g_k = Re <abar, -i dts G_k> (or Mbar = 2^-s abar in explicit mode). One work item = one step of one seed;
// scratch per workgroup: a, a^T, abar.
template <bool MANY>
__device__ __forceinline__ void krylov_body(const GeneralKrylovArgs& a, char* smem) {
    const int np = a.np, mat = np * np, tid = threadIdx.x, S = a.S, K = a.K;
    // (np > 256: the 26 chain vectors wait in global scratch, behind abar; LDS holds sigma, delta and the partial sums)
    const bool big = np > 256;
    double2* lds0 = reinterpret_cast<double2*>(smem);
    double2* gvec = a.scratch + (size_t)blockIdx.x * krylov_scratch_elems(np, S) + (size_t)3 * np * np +
                    (S >= 8 ? (size_t)28 * S * np : 0);
    double2* tau = big ? gvec : lds0;                  // [13][np]
    double2* rho = tau + 13 * np;                      // [13][np]
    double2* sig = big ? lds0 : rho + 13 * np;
    double2* del = sig + np;
    double2* part = del + np;  // [2][4][np]; above np = 256 (one group of rows): [2][np]... kept at [4][np]
    double* red = reinterpret_cast<double*>(part + (big ? 4 : 8) * np);
    // eight states or more: the chains of all states at once, as products on the matrix cores - the tau_j, rho_i of
    // every state in scratch ([13][S][np] each, + sigma, delta), abar += T^T conj(R) over the 13 S rows
    // (a kernel of its own, krylov_many_kernel: the products' registers would cost the vector form its occupancy)
    const size_t sn = (size_t)S * np;
    double2* A = a.scratch + (size_t)blockIdx.x * krylov_scratch_elems(np, S);
    double2* AT = A + mat;
    double2* AB = AT + mat;
    double2* TT = AB + mat;       // [13][S][np]
    double2* RR = TT + 13 * sn;   // [13][S][np]
    double2* SG = RR + 13 * sn;   // [S][np] sigma, then [S][np] delta
    double2* DL = SG + sn;
    for (size_t wi = blockIdx.x; wi < a.total; wi += gridDim.x) {
        const int b = (int)(wi / a.nsteps), step = (int)(wi % a.nsteps);
        const size_t m = (size_t)b * a.nsteps + step;
        const int entry = a.s_arr[m];
        const int sq = step_squarings(entry), M = step_order(entry), nsub = 1 << sq;
        const double dts = a.dt * ldexp(1.0, -sq);
        const double* bt = pade_table(M);
        const size_t cap = a.slot_cap;
        const double2* states_b = a.states + (size_t)b * cap * S * np;
        const double2* xs_b = a.xs + (size_t)b * cap * S * np;
        const int t0 = a.offs[(size_t)b * (a.nsteps + 1) + step];
        if (t0 < 0 || (size_t)t0 + (size_t)nsub >= cap) continue;  // the sweep overflowed (status bit 2)
        const size_t tsel = (a.nt == 1) ? 0 : (size_t)step;
        const double2* gk = a.g_rm ? a.g_rm + tsel * K * mat : nullptr;
        // a skew-Hermitian generator (Hermitian H, host-checked bit for bit), one state at a time: a rho = -a^H rho, so
        // both chains are products with a^H - ONE pass over the matrix per pair of chain steps, no transpose
        const bool skew1 = !MANY && a.skew != 0;
        // the scaled generator and its transpose
        if (a.gen_rm != nullptr) {
            const double2* g = a.gen_rm + m * mat;
            const double scl = ldexp(1.0, -sq);
            for (int e = tid; e < mat; e += TPB) {
                const int r = e / np, c = e - r * np;
                const double2 v = make_double2(scl * g[e].x, scl * g[e].y);
                A[e] = v;
                if (!skew1) AT[(size_t)c * np + r] = v;
            }
        } else {
            const double2* h0 = a.h0_rm + tsel * mat;
            const double* ctl_b = a.controls + (size_t)b * a.nc * K;
            const StepInterp si = a.interp[step];
            for (int e = tid; e < mat; e += TPB) {
                const int r = e / np, c = e - r * np;
                double2 h = h0[e];
                for (int k = 0; k < K; ++k) {
                    const double uk = control_at(ctl_b, si, K, k);
                    const double2 g = gk[(size_t)k * mat + e];
                    h.x = fma(uk, g.x, h.x);
                    h.y = fma(uk, g.y, h.y);
                }
                const double2 v = make_double2(dts * h.y, -dts * h.x);
                A[e] = v;
                if (!skew1) AT[(size_t)c * np + r] = v;
            }
        }
        __syncthreads();
        bool first = true;
        if constexpr (MANY) {
            for (int sub = 0; sub < nsub; ++sub) {
                const size_t t = (size_t)t0 + sub;
                const double2* p0 = states_b + t * sn;
                const double2* p1 = states_b + (t + 1) * sn;
                const double2* xx = xs_b + t * sn;
                for (size_t e = tid; e < sn; e += TPB) {
                    const double2 u = p0[e], v = p1[e];
                    TT[e] = xx[e];
                    SG[e] = make_double2(u.x + v.x, u.y + v.y);
                    DL[e] = make_double2(u.x - v.x, u.y - v.y);
                    RR[(size_t)(M - 1) * sn + e] = make_double2(bt[M] * (u.x + v.x), bt[M] * (u.y + v.y));
                }
                __syncthreads();
                // T_j = T_{j-1} conj(a) (rows: tau_j = a^H tau_{j-1}); R_{i-1} = b_i W_i + R_i a^T (rows: a rho_i)
                for (int jj = 1; jj < M; ++jj)
                    gemm_op<false, false, true, false>(TT + (size_t)(jj - 1) * sn, A, TT + (size_t)jj * sn, S, np, np, smem);
                for (int ii = M - 1; ii >= 1; --ii) {
                    gemm_op<false, true, false, false>(RR + (size_t)ii * sn, A, RR + (size_t)(ii - 1) * sn, S, np, np, smem);
                    const double2* wv = (ii & 1) ? SG : DL;
                    double2* out = RR + (size_t)(ii - 1) * sn;
                    for (size_t e = tid; e < sn; e += TPB) {
                        double2 r = out[e];
                        r.x = fma(bt[ii], wv[e].x, r.x);
                        r.y = fma(bt[ii], wv[e].y, r.y);
                        out[e] = r;
                    }
                    __syncthreads();
                }
                // abar[r][c] (+)= sum over (i, s) of tau_i[s][r] conj(rho_i[s][c])
                if (first) gemm_op<true, false, true, false>(TT, RR, AB, np, np, M * S, smem);
                else gemm_op<true, false, true, true>(TT, RR, AB, np, np, M * S, smem);
                first = false;
            }
        }
        for (int sub = 0; !MANY && sub < nsub; ++sub)  // (up to seven states: state by state, vectors in LDS)
            for (int s = 0; s < S; ++s) {
                const size_t t = (size_t)t0 + sub;
                for (int i = tid; i < np; i += TPB) {
                    const double2 p0 = states_b[(t * S + s) * np + i], p1 = states_b[((t + 1) * S + s) * np + i];
                    tau[i] = xs_b[(t * S + s) * np + i];
                    sig[i] = make_double2(p0.x + p1.x, p0.y + p1.y);
                    del[i] = make_double2(p0.x - p1.x, p0.y - p1.y);
                }
                __syncthreads();
                if (skew1) {
                    for (int i = tid; i < np; i += TPB) rho[(M - 1) * np + i] = make_double2(bt[M] * sig[i].x, bt[M] * sig[i].y);
                    __syncthreads();
                    for (int jj = 1; jj < M; ++jj) {
                        const int ii = M - jj;  // tau_jj = a^H tau_{jj-1};  rho_{ii-1} = b_ii w_ii - a^H rho_ii
                        matvec_cols2<true>(A, tau + (jj - 1) * np, tau + jj * np, rho + ii * np, rho + (ii - 1) * np, part, np);
                        const double2* wv = (ii & 1) ? sig : del;
                        for (int i = tid; i < np; i += TPB) {
                            const double2 r = rho[(ii - 1) * np + i];
                            rho[(ii - 1) * np + i] = make_double2(fma(bt[ii], wv[i].x, -r.x), fma(bt[ii], wv[i].y, -r.y));
                        }
                        __syncthreads();
                    }
                } else {
                    // tau_j = a^H tau_{j-1}: (a^H v)_c = sum_r conj(a[r][c]) v_r
                    for (int jj = 1; jj < M; ++jj) matvec_cols<true>(A, tau + (jj - 1) * np, tau + jj * np, part, np);
                    for (int i = tid; i < np; i += TPB) rho[(M - 1) * np + i] = make_double2(bt[M] * sig[i].x, bt[M] * sig[i].y);
                    __syncthreads();
                    // rho_{i-1} = b_i w_i + a rho_i: (a v)_r = sum_c a^T[c][r] v_c
                    for (int ii = M - 1; ii >= 1; --ii) {
                        matvec_cols<false>(AT, rho + ii * np, rho + (ii - 1) * np, part, np);
                        const double2* wv = (ii & 1) ? sig : del;
                        for (int i = tid; i < np; i += TPB) {
                            double2 r = rho[(ii - 1) * np + i];
                            r.x = fma(bt[ii], wv[i].x, r.x);
                            r.y = fma(bt[ii], wv[i].y, r.y);
                            rho[(ii - 1) * np + i] = r;
                        }
                        __syncthreads();
                    }
                }
                // abar[r][c] += sum_i tau_i[r] conj(rho_i[c])
                for (int e = tid; e < mat; e += TPB) {
                    const int r = e / np, c = e - r * np;
                    double2 acc = first ? make_double2(0, 0) : AB[e];
                    for (int ii = 0; ii < M; ++ii) {
                        const double2 tv = tau[ii * np + r], rv = rho[ii * np + c];
                        acc.x = fma(tv.y, rv.y, fma(tv.x, rv.x, acc.x));
                        acc.y = fma(-tv.x, rv.y, fma(tv.y, rv.x, acc.y));
                    }
                    AB[e] = acc;
                }
                first = false;
                __syncthreads();
            }
        if (a.gen_rm != nullptr) {  // Mbar = 2^-s abar: the host finishes the chain rule
            double2* mb = a.mbar_rm + m * mat;
            const double scl = ldexp(1.0, -sq);
            for (int e = tid; e < mat; e += TPB) mb[e] = make_double2(scl * AB[e].x, scl * AB[e].y);
        } else {
            // g_k = Re sum conj(abar) E_k, E_k = -i dts G_k
            for (int k = 0; k < K; ++k) {
                double acc = 0;
                for (int e = tid; e < mat; e += TPB) {
                    const double2 g = gk[(size_t)k * mat + e], ab = AB[e];
                    acc = fma(ab.y, -dts * g.x, fma(ab.x, dts * g.y, acc));
                }
                const double2 tot = block_sum2(acc, 0.0, red);
                if (tid == 0) a.gstep[m * K + k] = tot.x;
            }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(TPB) void krylov_kernel(GeneralKrylovArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    krylov_body<false>(a, smem);
}
__global__ __launch_bounds__(TPB, 2) void krylov_many_kernel(GeneralKrylovArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    krylov_body<true>(a, smem);
}

// ---- Magnus M4 / M6 generators and their reverse rules (mathmethods.py:96-164) ---------------------------------
// One work item = one step of one seed; every matrix in scratch (24 per workgroup), commutators as two products
// on the matrix cores. MagnusArgs as the wavefront kernels take it, with h0_cimg / g_cimg pointing at the
// row-major padded matrices and n = the padded size. VJP: the node generators, b_i, x, w, y are recomputed, the
// cotangent of M (K3's Mbar) goes back through the commutators (Z = [X, Y]: Xbar = Zbar Y^H - Y^H Zbar,
// Ybar = X^H Zbar - Zbar X^H) and is contracted with -i G_k at every node.
template <bool VJP>
__global__ __launch_bounds__(TPB, 2) void magnus_kernel(MagnusArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int np = a.n, mat = np * np, tid = threadIdx.x, K = a.K, nodes = a.nodes;
    double* red = reinterpret_cast<double*>(smem + sizeof(GemmLds));
    double2* sc = a.scratch + (size_t)blockIdx.x * 24 * mat;
    auto mx = [&](int i) { return sc + (size_t)i * mat; };
    const double F0 = sqrt(15.0) / 3, F1 = 10.0 / 3, F2 = 0.5, F3 = 1.0 / 240, F4 = 1.0 / 60;
    const double M4F0 = sqrt(3.0) / 12;
    const double dt = a.dt;
    // OUT = [X, Y] (T: a temporary)
    auto comm = [&](const double2* X, const double2* Y, double2* OUT, double2* T) {
        gemm(X, Y, OUT, np, smem);
        gemm(Y, X, T, np, smem);
        for (int e = tid; e < mat; e += TPB) {
            double2 v = OUT[e];
            v.x -= T[e].x;
            v.y -= T[e].y;
            OUT[e] = v;
        }
        __syncthreads();
    };
    auto herm = [&](double2* dst, const double2* src) {
        for (int e = tid; e < mat; e += TPB) {
            const int r = e / np, c = e - r * np;
            const double2 v = src[(size_t)c * np + r];
            dst[e] = make_double2(v.x, -v.y);
        }
        __syncthreads();
    };
    // Z = [X, Y], Zbar given: XB = Zbar Y^H - Y^H Zbar, YB = X^H Zbar - Zbar X^H (T, TH: temporaries)
    auto comm_vjp = [&](const double2* X, const double2* Y, const double2* ZB, double2* XB, double2* YB, double2* T,
                        double2* TH) {
        herm(TH, Y);
        gemm(ZB, TH, XB, np, smem);
        gemm(TH, ZB, T, np, smem);
        for (int e = tid; e < mat; e += TPB) {
            double2 v = XB[e];
            v.x -= T[e].x;
            v.y -= T[e].y;
            XB[e] = v;
        }
        __syncthreads();
        herm(TH, X);
        gemm(TH, ZB, YB, np, smem);
        gemm(ZB, TH, T, np, smem);
        for (int e = tid; e < mat; e += TPB) {
            double2 v = YB[e];
            v.x -= T[e].x;
            v.y -= T[e].y;
            YB[e] = v;
        }
        __syncthreads();
    };
    for (size_t wi = blockIdx.x; wi < a.total; wi += gridDim.x) {
        const int b = (int)(wi / a.nsteps), step = (int)(wi % a.nsteps);
        const size_t m = (size_t)b * a.nsteps + step;
        const double* ctl_b = a.controls + (size_t)b * a.nc * K;
        // node generators a_q = -i H(u(t_q), t_q) into matrices 0 .. nodes - 1
        for (int q = 0; q < nodes; ++q) {
            const size_t tsel = (a.nt == 1) ? 0 : (size_t)step * nodes + q;
            const double2* h0 = a.h0_cimg + tsel * mat;
            const double2* gk = a.g_cimg + tsel * K * mat;
            const StepInterp si = a.interp[(size_t)step * nodes + q];
            double2* A = mx(q);
            for (int e = tid; e < mat; e += TPB) {
                double2 h = h0[e];
                for (int k = 0; k < K; ++k) {
                    const double uk = control_at(ctl_b, si, K, k);
                    const double2 g = gk[(size_t)k * mat + e];
                    h.x = fma(uk, g.x, h.x);
                    h.y = fma(uk, g.y, h.y);
                }
                A[e] = make_double2(h.y, -h.x);
            }
        }
        __syncthreads();
        if (nodes == 2) {
            double2 *a1 = mx(0), *a2 = mx(1), *c = mx(2), *t = mx(3);
            if (!VJP) {
                comm(a2, a1, c, t);
                double2* out = a.m_rm + m * mat;
                for (int e = tid; e < mat; e += TPB)
                    out[e] = make_double2((dt / 2) * (a1[e].x + a2[e].x) + M4F0 * dt * dt * c[e].x,
                                          (dt / 2) * (a1[e].y + a2[e].y) + M4F0 * dt * dt * c[e].y);
            } else {
                const double2* mb = a.mbar_rm + m * mat;
                double2 *zb = mx(4), *a2b = mx(5), *a1b = mx(6), *th = mx(7);
                for (int e = tid; e < mat; e += TPB) zb[e] = make_double2(M4F0 * dt * dt * mb[e].x, M4F0 * dt * dt * mb[e].y);
                __syncthreads();
                comm_vjp(a2, a1, zb, a2b, a1b, t, th);
                for (int q = 0; q < 2; ++q) {
                    const double2* ab = q == 0 ? a1b : a2b;
                    const size_t tsel = (a.nt == 1) ? 0 : (size_t)step * 2 + q;
                    const double2* gk = a.g_cimg + tsel * K * mat;
                    for (int k = 0; k < K; ++k) {
                        double acc = 0;
                        for (int e = tid; e < mat; e += TPB) {
                            const double2 g = gk[(size_t)k * mat + e];
                            const double bx = (dt / 2) * mb[e].x + ab[e].x, by = (dt / 2) * mb[e].y + ab[e].y;
                            acc = fma(bx, g.y, fma(-by, g.x, acc));
                        }
                        const double2 tot = block_sum2(acc, 0.0, red);
                        if (tid == 0) a.gstep[(m * 2 + q) * K + k] = tot.x;
                    }
                }
            }
        } else {  // M6
            double2 *a1 = mx(0), *a2 = mx(1), *a3 = mx(2), *b1 = mx(3), *b2 = mx(4), *b3 = mx(5), *c12 = mx(6), *x = mx(7),
                    *w = mx(8), *y = mx(9), *t = mx(10), *th = mx(11);
            for (int e = tid; e < mat; e += TPB) {
                const double2 u1 = a1[e], u2 = a2[e], u3 = a3[e];
                b1[e] = make_double2(dt * u2.x, dt * u2.y);
                b2[e] = make_double2(F0 * dt * (u3.x - u1.x), F0 * dt * (u3.y - u1.y));
                b3[e] = make_double2(F1 * dt * (u3.x - 2 * u2.x + u1.x), F1 * dt * (u3.y - 2 * u2.y + u1.y));
            }
            __syncthreads();
            comm(b1, b2, c12, t);
            for (int e = tid; e < mat; e += TPB) {
                x[e] = make_double2(-20 * b1[e].x - b3[e].x + c12[e].x, -20 * b1[e].y - b3[e].y + c12[e].y);
                w[e] = make_double2(2 * b3[e].x + c12[e].x, 2 * b3[e].y + c12[e].y);
            }
            __syncthreads();
            comm(b1, w, y, t);  // y := [b1, w] for now
            for (int e = tid; e < mat; e += TPB) y[e] = make_double2(b2[e].x - F4 * y[e].x, b2[e].y - F4 * y[e].y);
            __syncthreads();
            if (!VJP) {
                double2* cxy = mx(12);
                comm(x, y, cxy, t);
                double2* out = a.m_rm + m * mat;
                for (int e = tid; e < mat; e += TPB)
                    out[e] = make_double2(b1[e].x + F2 * b3[e].x + F3 * cxy[e].x, b1[e].y + F2 * b3[e].y + F3 * cxy[e].y);
            } else {
                const double2* mb = a.mbar_rm + m * mat;
                double2 *zb = mx(12), *xb = mx(13), *yb = mx(14), *wb = mx(15), *d1 = mx(16), *d2 = mx(17), *b1b = mx(18),
                        *b2b = mx(19), *b3b = mx(20), *c12b = mx(21);
                for (int e = tid; e < mat; e += TPB) zb[e] = make_double2(F3 * mb[e].x, F3 * mb[e].y);
                __syncthreads();
                comm_vjp(x, y, zb, xb, yb, t, th);
                for (int e = tid; e < mat; e += TPB) {
                    b1b[e] = make_double2(mb[e].x - 20 * xb[e].x, mb[e].y - 20 * xb[e].y);
                    b3b[e] = make_double2(F2 * mb[e].x - xb[e].x, F2 * mb[e].y - xb[e].y);
                    c12b[e] = xb[e];
                    b2b[e] = yb[e];
                    zb[e] = make_double2(-F4 * yb[e].x, -F4 * yb[e].y);
                }
                __syncthreads();
                comm_vjp(b1, w, zb, d1, wb, t, th);
                for (int e = tid; e < mat; e += TPB) {
                    b1b[e] = make_double2(b1b[e].x + d1[e].x, b1b[e].y + d1[e].y);
                    b3b[e] = make_double2(b3b[e].x + 2 * wb[e].x, b3b[e].y + 2 * wb[e].y);
                    c12b[e] = make_double2(c12b[e].x + wb[e].x, c12b[e].y + wb[e].y);
                }
                __syncthreads();
                comm_vjp(b1, b2, c12b, d1, d2, t, th);
                // a1bar = -F0 dt b2bar + F1 dt b3bar, a2bar = dt b1bar - 2 F1 dt b3bar, a3bar = F0 dt b2bar + F1 dt b3bar
                for (int q = 0; q < 3; ++q) {
                    const size_t tsel = (a.nt == 1) ? 0 : (size_t)step * 3 + q;
                    const double2* gk = a.g_cimg + tsel * K * mat;
                    for (int k = 0; k < K; ++k) {
                        double acc = 0;
                        for (int e = tid; e < mat; e += TPB) {
                            const double2 g = gk[(size_t)k * mat + e];
                            const double p1x = b1b[e].x + d1[e].x, p1y = b1b[e].y + d1[e].y;
                            const double p2x = b2b[e].x + d2[e].x, p2y = b2b[e].y + d2[e].y;
                            const double p3x = b3b[e].x, p3y = b3b[e].y;
                            double bx, by;
                            if (q == 0) {
                                bx = -F0 * dt * p2x + F1 * dt * p3x;
                                by = -F0 * dt * p2y + F1 * dt * p3y;
                            } else if (q == 1) {
                                bx = dt * p1x - 2 * F1 * dt * p3x;
                                by = dt * p1y - 2 * F1 * dt * p3y;
                            } else {
                                bx = F0 * dt * p2x + F1 * dt * p3x;
                                by = F0 * dt * p2y + F1 * dt * p3y;
                            }
                            acc = fma(bx, g.y, fma(-by, g.x, acc));
                        }
                        const double2 tot = block_sum2(acc, 0.0, red);
                        if (tid == 0) a.gstep[(m * 3 + q) * K + k] = tot.x;
                    }
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace general

void launch_general_magnus(const MagnusArgs& a, bool vjp, int blocks, hipStream_t st) {
    const int bytes = (int)sizeof(general::GemmLds) + 256;
    if (vjp) hipLaunchKernelGGL(general::magnus_kernel<true>, dim3(blocks), dim3(general::TPB), bytes, st, a);
    else hipLaunchKernelGGL(general::magnus_kernel<false>, dim3(blocks), dim3(general::TPB), bytes, st, a);
}

int general_factor_lds(int np) { return std::max((int)sizeof(general::GemmLds), general::invert_lds(np)) + 256; }
int general_sweep_lds(int np) { return 7 * np * 16 + 256 + (int)sizeof(general::GemmLds); }
int general_krylov_lds(int np) {
    return std::max((np > 256 ? 2 + 4 : 13 + 13 + 2 + 8) * np * 16, (int)sizeof(general::GemmLds)) + 256;
}
size_t general_krylov_scratch(int np, int S) { return general::krylov_scratch_elems(np, S); }

int launch_general_factor(const GeneralArgs& a, int blocks, hipStream_t st) {
    const int bytes = general_factor_lds(a.np);
    static int attr_bytes = 0;
    if (bytes > 48 * 1024 && bytes > attr_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(general::factor_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return 1;
        attr_bytes = bytes;
    }
    hipLaunchKernelGGL(general::factor_kernel, dim3(blocks), dim3(general::TPB), bytes, st, a);
    return 0;
}

// a.phase: bit 0 forward, bit 1 adjoint; split mode (bit 3; bits 8.. = workgroups per seed): bit 2 the cost launch
// (bit 4: it also seeds lambda)
void launch_general_sweep(const GeneralSweepArgs& a, int batch, hipStream_t st) {
    const int groups = (a.phase & 8) ? std::max(1, a.phase >> 8) : 1;
    const int bytes = general_sweep_lds(a.np);
    static int attr_bytes = 0;
    if (bytes > 48 * 1024 && bytes > attr_bytes) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(general::sweep_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        attr_bytes = bytes;
    }
    hipLaunchKernelGGL(general::sweep_kernel, dim3(batch, (a.phase & 4) ? 1 : groups), dim3(general::TPB), bytes, st, a);
}

int launch_general_krylov(const GeneralKrylovArgs& a, int blocks, hipStream_t st) {
    const int bytes = general_krylov_lds(a.np);
    static int attr_bytes = 0;
    if (bytes > 48 * 1024 && bytes > attr_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(general::krylov_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(general::krylov_many_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return 1;
        attr_bytes = bytes;
    }
    if (a.S >= 8) hipLaunchKernelGGL(general::krylov_many_kernel, dim3(blocks), dim3(general::TPB), bytes, st, a);
    else hipLaunchKernelGGL(general::krylov_kernel, dim3(blocks), dim3(general::TPB), bytes, st, a);
    return 0;
}

}  // namespace qocx
