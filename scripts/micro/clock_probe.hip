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
