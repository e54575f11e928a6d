// dpp_chain.hip - GPU-BOX TOOLING (micro-benchmark, not part of the product): cycles per
// instruction pattern of the sweep's triangular-solve stage on one wave, s_memtime around an
// unrolled run of 32 stages, repeated.  hipcc --offload-arch=gfx950 -O3 -o dpp_chain dpp_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, int iters) {
    double zr = 1.0 + threadIdx.x * 1e-3, zi = 0.5, cr = 1e-3, ci = 2e-3, wr = 0.25, wi = 0.125;
    double t0 = 0, t1 = 0;
    unsigned long long a, b;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // dependent plain FMA chain (one accumulator)
            asm volatile(REP32("v_fma_f64 %0, %1, %2, %0\n\t") : "+v"(zr) : "v"(cr), "v"(ci));
        } else if (MODE == 1) {  // dependent DPP FMAC chain reading its own accumulator through DPP
            asm volatile(REP32("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(zr) : "v"(cr));
        } else if (MODE == 2) {  // the stage: exec mask + 4 DPP FMACs + exec restore
            asm volatile(REP32(
                "s_mov_b32 exec_lo, 0xfffffff0\n\ts_mov_b32 exec_hi, 0xfffffff0\n\t"
                "v_fmac_f64_dpp %0, -%0, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%1, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %1, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%0, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci));
        } else if (MODE == 3) {  // the stage without the exec writes
            asm volatile(REP32(
                "v_fmac_f64_dpp %0, -%0, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%1, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %1, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%0, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci));
        } else if (MODE == 4) {  // round 4's stage: v_readlane x4, exec mask, 4 plain FMAs
            asm volatile("v_mov_b64 v[10:11], %0\n\tv_mov_b64 v[12:13], %1\n\t" REP32(
                "v_readlane_b32 s20, v10, 5\n\tv_readlane_b32 s21, v11, 5\n\t"
                "v_readlane_b32 s22, v12, 5\n\tv_readlane_b32 s23, v13, 5\n\t"
                "s_mov_b32 exec_lo, 0xfffffff0\n\ts_mov_b32 exec_hi, 0xfffffff0\n\t"
                "v_fma_f64 v[10:11], -%2, s[20:21], v[10:11]\n\t"
                "v_fma_f64 v[12:13], -%2, s[22:23], v[12:13]\n\t"
                "v_fma_f64 v[10:11], %3, s[22:23], v[10:11]\n\t"
                "v_fma_f64 v[12:13], -%3, s[20:21], v[12:13]\n\t"
                "s_mov_b64 exec, -1\n\t") "v_mov_b64 %0, v[10:11]\n\tv_mov_b64 %1, v[12:13]\n\t"
                : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci) : "s20", "s21", "s22", "s23", "v10", "v11", "v12", "v13");
        } else if (MODE == 5) {  // broadcast by v_mov_b64_dpp into temporaries, plain masked FMAs
            asm volatile(REP32(
                "v_mov_b64_dpp %4, %0 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b64_dpp %5, %1 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "s_mov_b32 exec_lo, 0xfffffff0\n\ts_mov_b32 exec_hi, 0xfffffff0\n\t"
                "v_fma_f64 %0, -%2, %4, %0\n\t"
                "v_fma_f64 %1, -%2, %5, %1\n\t"
                "v_fma_f64 %0, %3, %5, %0\n\t"
                "v_fma_f64 %1, -%3, %4, %1\n\t"
                "s_mov_b64 exec, -1\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci), "v"(t0), "v"(t1));
        } else if (MODE == 6) {  // off-diagonal block: independent of z, 4 FMACs per column from w
            asm volatile(REP32(
                "v_fmac_f64_dpp %0, -%4, %2 row_newbcast:4 row_mask:0xa bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%5, %2 row_newbcast:4 row_mask:0xa bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %5, %3 row_newbcast:4 row_mask:0xa bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%4, %3 row_newbcast:4 row_mask:0xa bank_mask:0xf\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci), "v"(wr), "v"(wi));
        } else if (MODE == 7) {  // independent plain FMAs (issue rate)
            asm volatile(REP32("v_fma_f64 %0, %2, %3, %0\n\tv_fma_f64 %1, %2, %3, %1\n\tv_fma_f64 %4, %2, %3, %4\n\tv_fma_f64 %5, %2, %3, %5\n\t")
                         : "+v"(zr), "+v"(zi), "+v"(t0), "+v"(t1) : "v"(cr), "v"(ci));
        } else if (MODE == 8) {  // the stage with v_mul (no accumulate dependency through DPP): DPP read of a fresh register
            asm volatile(REP32(
                "v_fmac_f64_dpp %0, -%4, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b64 %4, %0\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci), "v"(wr), "v"(wi));
        } else if (MODE == 9) {  // stage with s_nop padding instead of exec writes
            asm volatile(REP32(
                "s_nop 0\n\ts_nop 0\n\t"
                "v_fmac_f64_dpp %0, -%0, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%1, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %1, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%0, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "s_nop 0\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci));
        } else if (MODE == 10) {  // dependent v_fmac_f64 (non-DPP, VOP2) chain
            asm volatile(REP32("v_fmac_f64 %0, %1, %2\n\t") : "+v"(zr) : "v"(cr), "v"(ci));
        } else if (MODE == 11) {  // two-accumulator alternation plain FMA (distance 2)
            asm volatile(REP32("v_fma_f64 %0, %2, %3, %0\n\tv_fma_f64 %1, %2, %3, %1\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci));
        } else if (MODE == 12) {  // stage: DPP FMACs with exec set by ONE s_mov_b64 from an SGPR pair
            asm volatile(REP32(
                "s_mov_b64 exec, %4\n\t"
                "v_fmac_f64_dpp %0, -%0, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%1, %2 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %0, %1, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %1, -%0, %3 row_newbcast:4 row_mask:0x5 bank_mask:0xf\n\t") : "+v"(zr), "+v"(zi) : "v"(cr), "v"(ci), "s"(0xfffffff0fffffff0ull));
            asm volatile("s_mov_b64 exec, -1");
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b)::"memory");
    out[blockIdx.x * 64 + threadIdx.x] = zr + zi + t0 + t1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = b - a;
}

template <int MODE>
void run(const char* name, int per_rep) {
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 8 * 4); hipMalloc(&cyc, 8 * 4);
    const int iters = 200;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("{\"mode\": %d, \"pattern\": \"%s\", \"cycles_per_repeat\": %.2f, \"instructions_per_repeat\": %d}\n", MODE, name,
           (double)c / (iters * 32.0), per_rep);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("dependent v_fma_f64", 1);
    run<10>("dependent v_fmac_f64 (VOP2)", 1);
    run<1>("dependent v_fmac_f64_dpp on its own accumulator", 1);
    run<11>("two alternating FMA accumulators", 2);
    run<7>("four independent FMA accumulators", 4);
    run<2>("stage: 2 s_mov exec + 4 fmac_dpp + s_mov exec", 7);
    run<3>("stage: 4 fmac_dpp, no exec writes", 4);
    run<9>("stage: 4 fmac_dpp, s_nop in place of exec writes", 7);
    run<12>("stage: s_mov_b64 exec from SGPR + 4 fmac_dpp", 5);
    run<4>("round-4 stage: 4 v_readlane + exec + 4 v_fma", 11);
    run<5>("stage: 2 v_mov_b64_dpp + exec + 4 v_fma", 9);
    run<6>("off-diagonal column: 4 fmac_dpp from w", 4);
    run<8>("fmac_dpp + v_mov_b64 feeding the next DPP read", 2);
    return 0;
}
