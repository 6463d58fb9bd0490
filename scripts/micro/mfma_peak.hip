// Microbenchmark: achievable v_mfma_f32_32x32x2_f32 rate on this MI355X (calibrates the roofline peak).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(out, iters, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(out, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d blocks=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}
static void sustained() {
    float* out; (void)hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 40; ++i) k<4><<<2048, 256>>>(out, 2000, 0.5f, 0.25f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("sustained rep %d: %.1f ms  %.1f TFLOP/s\n", rep, ms, 40.0 * 2048 * 4 * 2000 * 8 * 4 * 4096.0 / ms / 1e9);
    }
}
int main(int argc, char** argv) {
    if (argc > 1) { sustained(); return 0; }
    run<1>(256, 20000); run<1>(512, 20000); run<1>(1024, 10000); run<4>(256, 5000); run<4>(512, 5000); run<4>(2048, 2000);
    return 0;
}
