// qocx_pade3.hip - K1a for 17 <= n <= 32, Hermitian generators, Pade orders 3 and 5: THREE waves per
// propagator step, one 16 x 16 tile of every product each (round 5).
//
// The two-wave kernel (qocx_pade2.hip) gives wave 0 one tile of the three a Hermitian product needs
// and wave 1 two: wave 0 waits a fifth of its life at barriers, and both waves carry two tiles of
// every live matrix - 242 registers, two waves per SIMD, and the stamps say the kernel is bound by
// that occupancy, not by either pipe (the matrix cores are busy ~45 % of a launch, the vector unit
// ~35 %). Here wave 0 owns tile (0,0), wave 1 tile (0,1), wave 2 tile (1,1): equal work between
// barriers, one tile of every matrix per wave, and registers for three or four waves per SIMD.
//
// What makes one tile per wave possible without exchanging B operands: at orders 3 and 5 (Higham 2005,
// (10.33); the reference's pade3 / pade5, qoc/standard/functions/expm.py:119-135) the products are
//     x2 = a a          A = a  (LDS slot)   B = a  (the wave's own column block, from the generator)
//     x4 = x2 x2        A = x2 (LDS slot)   B = x2 (the SAME slot)
//     u  = w a + b1 a   A = w  (LDS slot)   B = a  (registers again),  w = b5 x4 + b3 x2
// so every B operand is either in the slot already or never left the wave. Higher orders need a B
// operand that is neither (x6 = x2 x4) and stay on the two-wave kernel: both kernels are launched
// over the same grid, each workgroup reads its step's order from the step table and the one the step
// does not belong to leaves at once (the two-wave launch is skipped when the host's bound of the step
// norm is below theta_5 for the whole evaluation, FactorArgs::prefer_low == 2).
// The denominators of such steps are diagonally dominant (eps <= 0.135, qocx_lu5.h): wave 0 factors P
// from the LDS image with the vector-unit factorisation, no check, no fall-back; waves 1 and 2 leave.
//
//   reference: expm.py:116-135 (pade3, pade5), :246 (solve(P, Q)); schroedingerdiscrete.py:483-489
#include "qocx_wave.h"
#include "qocx_lu.h"
#include "qocx_lu5.h"

namespace qocx {

namespace pade3 {

constexpr int PITCH = Geo<2>::PITCH, PLANE = Geo<2>::PLANE, MAT = Geo<2>::MAT;
constexpr int SLOT_F64 = 2 * PLANE;  // re | im planes
constexpr int LP = 33;               // pitch (complex) of the column-major P image the LU reads
constexpr int LDS_BYTES = SLOT_F64 * 8;
static_assert(32 * LP * 16 <= SLOT_F64 * 8, "the P image reuses the A-operand slot");

struct Tile {  // one 16 x 16 complex tile, accumulator layout: lane (q, c), register r = (4 r + q, c)
    d4 re, im;
};

__device__ __forceinline__ void stage_tile(double* slot, int ti, int tj, const d4& re, const d4& im) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int off = (16 * ti + 4 * r + q) * PITCH + 16 * tj + c;
        slot[off] = re[r];
        slot[PLANE + off] = im[r];
    }
}
// tile (1, 0) of the slot := conj(src)^T, src being tile (0, 1) of a Hermitian matrix
__device__ __forceinline__ void stage_mirror10(double* slot, const d4& re, const d4& im) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int off = (16 + c) * PITCH + 4 * r + q;
        slot[off] = re[r];
        slot[PLANE + off] = -im[r];
    }
}

// acc += A(ti, :) B(:, tj), 3M scheme (three real products per complex one); A from the slot, the B
// fragment of k-step kk from `bf`
template <class BFrag>
__device__ __forceinline__ void gemm3(d4& t1, d4& t2, d4& t3, const double* slot, int ti, BFrag bf) {
    const int q = lane_id() >> 4, c = lane_id() & 15;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int off = (16 * ti + c) * PITCH + 4 * kk + q;
        const double are = slot[off], aim = slot[PLANE + off];
        double bre, bim;
        bf(kk, bre, bim);
        t1 = mfma_f64(are, bre, t1);
        t2 = mfma_f64(aim, bim, t2);
        t3 = mfma_f64(are + aim, bre + bim, t3);
    }
}

// PARK: the second half of the factorisation is left to lu5_second_kernel. The Schur complement S of a
// step is 16 x 16: lu5 eliminates it a row per lane in every row of 16 lanes at once, and with ONE matrix
// in the wave the four rows of lanes hold four copies of it (800 of the factorisation's 2 760 vector
// instructions at a quarter of the lanes). Wave 0 parks S in the (1, 1) block of the step's LU image
// instead (4 KB through L2), and lu5_second_kernel gives every row of lanes the S of a DIFFERENT step:
// the same instructions, four matrices. (Four steps per workgroup, the second halves at its end, was
// tried first and is 50 % slower: the waves that wait for wave 0 keep their slots, whereas here - as
// before - they leave and the next workgroup's products fill the matrix pipe beside the factorisation.)
template <bool STAMP, bool PARK>
__global__ __launch_bounds__(192, 3) void pade_pq3_kernel(FactorArgs args) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* sl = reinterpret_cast<double*>(smem_raw);
    StampClock<STAMP> clk;
    clk.start();
    const int step = args.step0 + blockIdx.x, b = blockIdx.y;
    const int lane = lane_id(), q = lane >> 4, c = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ti = (w == 2) ? 1 : 0, tj = (w >= 1) ? 1 : 0;  // this wave's tile
    const size_t m = (size_t)b * args.nsteps + step;
    const int entry = args.s_arr[m];
    const int order = step_order(entry);
    if (order > 5) return;  // (workgroup-uniform: a step of a higher order belongs to the two-wave kernel)

    // ---- generator: the wave's column block tj of a = dt (-i H), H = h0 + sum_k u_k g_k
    // (schroedingerdiscrete.py:485-486, mathmethods.py:90-93); C-image index ((ti * 2 + tj) * 4 + r) * 64 + lane
    // Only three of the six tiles the waves hold are formed from the images in L2: wave 2 (tile (1,1)) has
    // wave 1's column block and takes it from the slot behind the first barrier; tile (1,0) of wave 0's
    // column block is the mirror of wave 1's tile (0,1) - a^H = -a, bit for bit with Hermitian images -,
    // which wave 1 puts into the slot beside its own tiles.
    d4 are[2], aim[2];
    const int share = args.gen_share;  // 0: every wave its own column block, 1: wave 2 from the slot, 2: and wave 0's tile (1,0)
    if (w < 2 || share == 0) {
        const int tcount = (w == 0 && share == 2) ? 1 : 2;
        const size_t tsel = (args.nt == 1) ? 0 : (size_t)step;
        const double2* h0 = args.h0_cimg + tsel * MAT;
        const double2* g = args.g_cimg + tsel * args.K * MAT;
        const double* ctl = args.controls + ((size_t)b * args.nc + step) * args.K;  // step table: [B][nsteps][K]
        d4 hre[2], him[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (t < tcount) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double2 e = h0[((t * 2 + tj) * 4 + r) * 64 + lane];
                    hre[t][r] = e.x;
                    him[t][r] = e.y;
                }
            }
        for (int k = 0; k < args.K; ++k) {
            const double uk = ctl[k];
            const double2* gk = g + (size_t)k * MAT;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (t < tcount) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double2 e = gk[((t * 2 + tj) * 4 + r) * 64 + lane];
                        hre[t][r] += uk * e.x;
                        him[t][r] += uk * e.y;
                    }
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (t < tcount) {
                are[t] = args.dt * him[t];
                aim[t] = -args.dt * hre[t];
            }
    }
    // the whole of `a` into the slot: wave 0 tile (0,0), wave 1 column block 1 and the mirror of its tile (0,1)
    if (w == 0) {
        stage_tile(sl, 0, 0, are[0], aim[0]);
        if (share != 2) stage_tile(sl, 1, 0, are[1], aim[1]);
    }
    if (w == 1) {
        stage_tile(sl, 0, 1, are[0], aim[0]);
        stage_tile(sl, 1, 1, are[1], aim[1]);
        if (share == 2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // element (16 + c, 4 r + q) = -conj(a[4 r + q][16 + c])
            const int off = (16 + c) * PITCH + 4 * r + q;
            sl[off] = -are[0][r];
            sl[PLANE + off] = aim[0][r];
        }
    }
    clk.lap(0);
    __syncthreads();  // 1
    clk.lap(2);
    if (w == 2 && share != 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int off = (16 * t + 4 * r + q) * PITCH + 16 * tj + c;
                are[t][r] = sl[off];
                aim[t][r] = sl[PLANE + off];
            }
    }
    if (w == 0 && share == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // tile (1, 0)
            const int off = (16 + 4 * r + q) * PITCH + c;
            are[1][r] = sl[off];
            aim[1][r] = sl[PLANE + off];
        }
    }

    const double* bt = pade_table(order);
    auto b_own = [&](int kk, double& bre, double& bim) {  // B = a, the wave's column block
        bre = are[kk >> 2][kk & 3];
        bim = aim[kk >> 2][kk & 3];
    };
    auto b_slot = [&](int kk, double& bre, double& bim) {  // B = the matrix in the slot
        const int off = (4 * kk + q) * PITCH + 16 * tj + c;
        bre = sl[off];
        bim = sl[PLANE + off];
    };
    const d4 zero = {0, 0, 0, 0};
    // ---- x2 = a a
    Tile x, wt, v;
    {
        d4 t1 = zero, t2 = zero, t3 = zero;
        gemm3(t1, t2, t3, sl, ti, b_own);
        x.re = t1 - t2;
        x.im = t3 - t1 - t2;
    }
    wt.re = bt[3] * x.re;
    wt.im = bt[3] * x.im;
    v.re = bt[2] * x.re;
    v.im = bt[2] * x.im;
    if (ti == tj) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * r + q == c) v.re[r] += bt[0];
    }
    clk.lap(1);
    __syncthreads();  // 2: every read of a is done
    clk.lap(2);
    if (order >= 5) {
        // ---- x4 = x2 x2, both operands from the slot
        stage_tile(sl, ti, tj, x.re, x.im);
        if (w == 1) stage_mirror10(sl, x.re, x.im);
        clk.lap(1);
        __syncthreads();  // 3
        clk.lap(2);
        {
            d4 t1 = zero, t2 = zero, t3 = zero;
            gemm3(t1, t2, t3, sl, ti, b_slot);
            x.re = t1 - t2;
            x.im = t3 - t1 - t2;
        }
        wt.re += bt[5] * x.re;
        wt.im += bt[5] * x.im;
        v.re += bt[4] * x.re;
        v.im += bt[4] * x.im;
        clk.lap(1);
        __syncthreads();  // 4: every read of x2 is done
        clk.lap(2);
    }
    // ---- u = w a + b1 a (w: the odd part's polynomial in a^2; they commute: expm.py:126, :134)
    stage_tile(sl, ti, tj, wt.re, wt.im);
    if (w == 1) stage_mirror10(sl, wt.re, wt.im);
    Tile u;
    {
        d4 t1 = bt[1] * are[ti], t2 = zero, t3 = bt[1] * (are[ti] + aim[ti]);
        clk.lap(1);
        __syncthreads();  // 5
        clk.lap(2);
        gemm3(t1, t2, t3, sl, ti, b_own);
        u.re = t1 - t2;
        u.im = t3 - t1 - t2;
    }
    clk.lap(1);
    __syncthreads();  // 6: every read of the slot is done - it becomes the image of P
    clk.lap(2);
    // ---- Q = v + u to HBM, P = v - u to the LDS image, straight from the accumulator layout (for a
    // fixed r the four q-lanes of a column hold rows 4r..4r+3: one 64-byte run of the column-major
    // image). Q = P^H: wave 1 also writes tile (1,0) of each image as the mirror of ITS tile of the other.
    double2* q_img = args.q_img + m * MAT;
    double2* simg = reinterpret_cast<double2*>(sl);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int col = 16 * tj + c, row = 16 * ti + 4 * r + q;
        q_img[col * 32 + row] = make_double2(v.re[r] + u.re[r], v.im[r] + u.im[r]);
        simg[col * LP + row] = make_double2(v.re[r] - u.re[r], v.im[r] - u.im[r]);
    }
    if (w == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // element (16 + c, 4 r + q)
            q_img[(4 * r + q) * 32 + 16 + c] = make_double2(v.re[r] - u.re[r], -(v.im[r] - u.im[r]));
            simg[(4 * r + q) * LP + 16 + c] = make_double2(v.re[r] + u.re[r], -(v.im[r] + u.im[r]));
        }
    }
    clk.lap(3);
    __syncthreads();  // 7: the image is complete
    clk.lap(2);
    if (w == 0) {
        LuArgs lu;
        lu.lu_img = args.lu_img; lu.dinv = args.dinv; lu.perm = args.perm; lu.iperm = args.iperm;
        lu.status = args.status; lu.nsteps = args.nsteps; lu.step0 = args.step0; lu.seg_len = args.seg_len;
        lu.n = args.n; lu.dbg = 0; lu.fallbacks = nullptr;
        if constexpr (PARK) {
            d4 sre, sim;
            lu5::lu_dpp_first(lu, m, simg, LP, sre, sim);
            double2* img = args.lu_img + m * 1024;  // S[4 r + q][c] -> column 16 + c, row 16 + 4 r + q
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 + c) * 32 + 16 + 4 * r + q] = make_double2(sre[r], sim[r]);
        } else {
            if (!(QOCX_DBG_BITS(args.dbg) & 4)) lu5::lu_dpp_body(lu, m, simg, LP);
        }
        clk.lap(4);
    }
    if constexpr (STAMP) {
        clk.acc[7] = __builtin_amdgcn_s_memrealtime() - clk.acc[7];
        if (lane == 0 && args.stamps != nullptr && w < 2) {  // (the tool reads two roles: waves 0 and 1)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                atomicAdd(args.stamps + ((blockIdx.x + 7 * blockIdx.y) & 1023) * 16 + w * 8 + k, clk.acc[k]);
        }
    }
}

// the second halves of the factorisations pade_pq3_kernel<., true> parked: a wave per FOUR steps of the
// launch (work item i -> seed i / seg_len, step step0 + i % seg_len), a row of 16 lanes each
__global__ __launch_bounds__(64) void lu5_second_kernel(FactorArgs args, int total) {
    const int lane = lane_id(), dr = lane >> 4, j = lane & 15;
    const int item = 4 * (int)blockIdx.x + dr;
    const int it = item < total ? item : total - 1;
    const size_t m = (size_t)(it / args.seg_len) * args.nsteps + args.step0 + it % args.seg_len;
    // (a step above order 5 was factored whole by the two-wave kernel)
    const bool mine = item < total && step_order(args.s_arr[m]) <= 5;
    const double2* img = args.lu_img + m * 1024;
    lu5::Block x;
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const double2 e = img[(16 + cc) * 32 + 16 + j];
        x.re[cc] = mine ? e.x : (cc == j ? 1.0 : 0.0);  // (no step in this row of lanes: the identity)
        x.im[cc] = mine ? e.y : 0.0;
    }
    LuArgs lu;
    lu.lu_img = args.lu_img; lu.dinv = args.dinv; lu.perm = args.perm; lu.iperm = args.iperm;
    lu.status = args.status; lu.nsteps = args.nsteps; lu.step0 = args.step0; lu.seg_len = args.seg_len;
    lu.n = args.n; lu.dbg = 0; lu.fallbacks = nullptr;
    lu5::lu_dpp_second(lu, m, x, mine);
}

}  // namespace pade3

// structured M2 problem through the step table, Hermitian generators: the steps at order 3 or 5
bool pq3_supports(const FactorArgs& a) {
    return a.hermitian && a.direct && a.fuse_lu && a.lu_dpp && a.pade_policy != 13;
}
void launch_pq3(const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
#ifdef QOCX_DIAG
    if (a.stamps != nullptr) {
        hipLaunchKernelGGL((pade3::pade_pq3_kernel<true, false>), dim3(nsteps, batch), dim3(192), pade3::LDS_BYTES, st, a);
        return;
    }
#endif
    if (pq3_parks(a, nsteps)) {
        hipLaunchKernelGGL((pade3::pade_pq3_kernel<false, true>), dim3(nsteps, batch), dim3(192), pade3::LDS_BYTES, st, a);
        if (a.four_steps == 1) launch_pq3_second(a, nsteps, batch, st);  // (2: the caller does, on a stream of its choice)
        return;
    }
    hipLaunchKernelGGL((pade3::pade_pq3_kernel<false, false>), dim3(nsteps, batch), dim3(192), pade3::LDS_BYTES, st, a);
}
// FactorArgs::four_steps: the launch parks the Schur complements and leaves the second halves of the
// factorisations to launch_pq3_second
bool pq3_parks(const FactorArgs& a, int nsteps) {
    return a.four_steps != 0 && a.seg_len == nsteps && a.stamps == nullptr && !(QOCX_DBG_BITS(a.dbg) & 4);
}
void launch_pq3_second(const FactorArgs& a, int nsteps, int batch, hipStream_t st) {
    const int total = nsteps * batch;
    hipLaunchKernelGGL(pade3::lu5_second_kernel, dim3((total + 3) / 4), dim3(64), 0, st, a, total);
}

}  // namespace qocx
