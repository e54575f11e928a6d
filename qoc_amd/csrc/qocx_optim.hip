// qocx_optim.hip - the multi-start GRAPE driver's per-iteration arithmetic on the device: control
// clipping (qoc/core/common.py:8-30), best-so-far bookkeeping and the Adam / SGD update
// (qoc/standard/optimizers/adam.py:110-165, sgd.py) for B control sets resident in HBM. Plain
// elementwise kernels, HBM bound (seven 4 MB arrays per update at the headline size: ~10 us).
//
// Every product and sum is rounded on its own (no contraction into fused multiply-adds) and
// division / square root are the IEEE ones, so a seed walks the trajectory of the reference's
// NumPy arithmetic bit for bit (tests/test_gpu_api.py: B = 8 equals eight single-seed runs).
#include "qocx_device.h"

namespace qocx {

#pragma clang fp contract(off)

__global__ void clip_controls_kernel(double* controls, size_t total, int k, const double* max_norms) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const double v = controls[idx], mod = fabs(v), mx = max_norms[idx % k];
    if (mx < mod) controls[idx] = (v / mod) * mx;
}

// best_controls[b] = controls[b], best_final[b] = final[b] for the seeds flagged in `improved`
__global__ void keep_best_kernel(const double* controls, double* best_controls, size_t per_seed,
                                 const double2* final_states, double2* best_final, size_t final_per_seed,
                                 const unsigned char* improved) {
    const size_t b = blockIdx.x;  // (the seed index on grid.x: no 65535 limit on the batch)
    if (!improved[b]) return;
    const size_t idx = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (idx < per_seed) best_controls[b * per_seed + idx] = controls[b * per_seed + idx];
    if (idx < final_per_seed) best_final[b * final_per_seed + idx] = final_states[b * final_per_seed + idx];
}

__global__ void optimizer_update_kernel(OptimArgs a) {
    const size_t b = blockIdx.x;
    if (!a.update[b]) return;
    const size_t idx = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
    if (idx >= a.per_seed) return;
    const size_t e = b * a.per_seed + idx;
    double g = a.grads[e];
    if (a.kind == 0) {  // SGD: params - learning_rate * grads
        const double s = a.learning_rate * g;
        a.params[e] = a.params[e] - s;
        return;
    }
    if (a.apply_clip) g = g < -a.clip ? -a.clip : (g > a.clip ? a.clip : g);
    const double m0 = a.beta_1 * a.moment[e], m1 = a.one_m_b1 * g;
    const double m = m0 + m1;
    const double sq = g * g;
    const double v0 = a.beta_2 * a.square_moment[e], v1 = a.one_m_b2 * sq;
    const double v = v0 + v1;
    a.moment[e] = m;
    a.square_moment[e] = v;
    const double mh = m / a.corr_1, vh = v / a.corr_2;
    const double den = sqrt(vh) + a.epsilon;
    const double q = mh / den;
    const double s = a.learning_rate * q;
    a.params[e] = a.params[e] - s;
}

void launch_clip_controls(double* controls, size_t total, int k, const double* max_norms, hipStream_t st) {
    if (total == 0 || k <= 0) return;
    hipLaunchKernelGGL(clip_controls_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       controls, total, k, max_norms);
}
void launch_keep_best(const double* controls, double* best_controls, size_t per_seed,
                      const double2* final_states, double2* best_final, size_t final_per_seed,
                      const unsigned char* improved, int batch, hipStream_t st) {
    const size_t widest = per_seed > final_per_seed ? per_seed : final_per_seed;
    if (batch <= 0 || widest == 0) return;
    hipLaunchKernelGGL(keep_best_kernel, dim3(batch, (unsigned)((widest + 255) / 256)), dim3(256), 0, st,
                       controls, best_controls, per_seed, final_states, best_final, final_per_seed,
                       improved);
}
void launch_optimizer_update(const OptimArgs& a, int batch, hipStream_t st) {
    if (batch <= 0 || a.per_seed == 0) return;
    hipLaunchKernelGGL(optimizer_update_kernel, dim3(batch, (unsigned)((a.per_seed + 255) / 256)),
                       dim3(256), 0, st, a);
}

}  // namespace qocx
