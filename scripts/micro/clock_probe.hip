// Shader-clock probe: one wave spins for `spin_us` and reports core-clock ticks (s_memtime) against
// the constant 100 MHz reference (s_memrealtime).  Launched on a second stream while another kernel
// loads the chip, it tells which clock the MFMA pipes really run at under that load (DVFS).
//   hipcc --offload-arch=gfx950 -shared -fPIC -O3 clock_probe.hip -o clock_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_clock_probe(unsigned long long* out, unsigned long long spin_ref_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ref_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 2] = c1 - c0;
    out[blockIdx.x * 2 + 1] = r1 - r0;
}

extern "C" int clock_probe(unsigned long long* out_dev, int blocks, double spin_us, void* stream) {
    k_clock_probe<<<blocks, 64, 0, (hipStream_t)stream>>>(out_dev, (unsigned long long)(spin_us * 100.0));
    return (int)hipGetLastError();
}

// Register-only MFMA load (scripts/micro/mfma_peak.hip's kernel) to run the probe beside: does the clock hold when
// NOTHING but the matrix pipe is busy?
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) k_mfma_burn(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const float a = 0.5f + threadIdx.x * 1e-3f, b = 0.25f - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int mfma_burn(float* out_dev, int blocks, int iters, void* stream) {
    k_mfma_burn<<<blocks, 256, 0, (hipStream_t)stream>>>(out_dev, iters);
    return (int)hipGetLastError();
}
