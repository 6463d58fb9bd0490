// Can the vector ALUs add fp32 throughput BESIDE the matrix cores?  gfx950's data sheet gives the same 157.3 TFLOP/s for
// vector fp32 (packed FMA) and matrix fp32.  Register-only loops: waves that issue v_mfma_f32_32x32x2_f32, waves that issue
// v_pk_fma_f32, and both kinds on every SIMD at once.  (An idea for the fp32 ceiling, DESIGN 7 "Why fp32 caps ..."; lab only.)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_mfma_overlap.hip -o scripts/micro/_bin/valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// mode bit 0: waves with (wave & 1) == 0 run MFMA; bit 1: waves with (wave & 1) == 1 run packed FMA.
// WAVES per workgroup = 8 -> two waves per SIMD: one of each kind when mode == 3.
__global__ void __launch_bounds__(512) k_burn(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (wave & 1) == 0 ? (mode & 1) : false;
    const bool do_valu = (wave & 1) == 1 ? (mode & 2) != 0 : false;
    float s = 0.0f;
    if (do_mfma) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const float a = 0.5f + threadIdx.x * 1e-3f, b = 0.25f - threadIdx.x * 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    }
    if (do_valu) {
        f32x2 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x2{(float)i, (float)threadIdx.x};
        const f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-7f, -1e-7f};
        // 32 MFMAs of 64 cycles = 2048 cycles per outer iteration on the other wave; 512 packed FMAs of 4 cycles match it
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(acc[i], a, b);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    }
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 4096));
    const int blocks = 256 * 2, iters = 2000;
    const char* names[4] = {"", "MFMA waves only (1 per SIMD)", "packed-FMA waves only (1 per SIMD)", "one MFMA wave + one packed-FMA wave per SIMD"};
    for (int mode = 1; mode <= 3; ++mode) {
        k_burn<<<blocks, 512>>>(out, 50, mode);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        k_burn<<<blocks, 512>>>(out, iters, mode);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double mfma = (mode & 1) ? (double)blocks * 4 * iters * 32 * 4096.0 : 0.0;          // 4 MFMA waves per workgroup
        const double valu = (mode & 2) ? (double)blocks * 4 * iters * 512 * 256.0 : 0.0;          // 4 FMA waves, 256 FLOP per instruction
        printf("%-48s %8.3f ms   matrix %6.1f TFLOP/s   vector %6.1f TFLOP/s   sum %6.1f\n", names[mode], ms, mfma / ms / 1e9, valu / ms / 1e9, (mfma + valu) / ms / 1e9);
    }
    return 0;
}
