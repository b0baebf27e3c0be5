// Micro-benchmark: rate of v_mfma_f64_16x16x4_f64 against the number of independent accumulator chains per wavefront and
// the number of wavefronts per SIMD (what a kernel needs in flight to keep the FP64 matrix pipe busy).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_chains.hip -o /tmp/mfma_chains && /tmp/mfma_chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void __launch_bounds__(1024) chains_kernel(int iters, double *out) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    d4 acc[CH];
#pragma unroll
    for (int t = 0; t < CH; t++) acc[t] = d4{0, 0, 0, 0};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32 / CH; u++)
#pragma unroll
            for (int t = 0; t < CH; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int t = 0; t < CH; t++) s += acc[t][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH> void run(int wpb, double *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256;
    chains_kernel<CH><<<blocks, wpb * 64>>>(10, out);
    hipDeviceSynchronize();
    hipEventRecord(e0); chains_kernel<CH><<<blocks, wpb * 64>>>(iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)iters * 32.0 * wpb / 4.0;
    printf("chains/wave=%d waves/SIMD=%d: %.1f TF, %.1f cycles per MFMA per SIMD (2.4 GHz)\n", CH, wpb / 4,
           256.0 * wpb * iters * 32.0 * 2048.0 / ms / 1e9, ms * 1e-3 * 2.4e9 / mfmas_per_simd);
}

int main() {
    double *out; hipMalloc(&out, 1024 * 1024 * sizeof(double));
    for (int wpb : {4, 8, 16}) { run<1>(wpb, out); run<2>(wpb, out); run<4>(wpb, out); run<8>(wpb, out); }
    return 0;
}
