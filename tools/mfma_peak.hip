// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 on gfx950, alone and next to FP64 VALU work.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: every wave runs MFMAs; mode 1: waves 0-3 MFMA, waves 4-7 v_fma_f64 (wave w and w+4 share a SIMD;
// the VALU waves run 10x the iterations so both kinds stay busy for the whole launch); mode 2: every wave v_fma_f64
__global__ void __launch_bounds__(512) peak_kernel(int iters, int mode, double *out) {
    const int wave = threadIdx.x >> 6;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    const bool do_mfma = (mode == 0) || (mode == 1 && wave < 4);
    if (mode == 1 && !do_mfma) iters *= 10;
    if (do_mfma) {
        d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
        double x[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int k = 0; k < 8; k++) x[k] = __builtin_fma(x[k], 1.0000001, 1e-9);
        }
        double s = 0; for (int k = 0; k < 8; k++) s += x[k];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

int main() {
    double *out; hipMalloc(&out, 8 * 1024 * 1024 * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpb : {4, 8}) for (int mode : {0, 1, 2}) {
        if (wpb == 4 && mode == 1) continue;
        const int blocks = 256 * (wpb == 4 ? 1 : 1);
        peak_kernel<<<blocks, wpb * 64>>>(100, mode, out);
        hipDeviceSynchronize();
        hipEventRecord(e0); peak_kernel<<<blocks, wpb * 64>>>(iters, mode, out); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waves = blocks * wpb;
        const double mf_waves = mode == 0 ? waves : (mode == 1 ? waves / 2 : 0), va_waves = waves - mf_waves;
        const double mfma_flop = mf_waves * iters * 32.0 * 2048.0, valu_flop = va_waves * iters * (mode == 1 ? 10.0 : 1.0) * 32.0 * 64 * 2;
        printf("waves/CU=%d mode=%d: %.2f ms  MFMA %.2f TF  VALU-f64 %.2f TF  (cycles/MFMA/SIMD at 2.4GHz: %.1f)\n", wpb, mode, ms,
               mfma_flop / ms / 1e9, valu_flop / ms / 1e9, mf_waves ? ms * 1e-3 * 2.4e9 / (iters * 32.0 * (mf_waves / 1024.0)) : 0.0);
    }
    return 0;
}
