// Memory-latency probe: ONE lane chases a random cyclic permutation through a buffer (dependent loads, one in flight)
// and reports the 100 MHz wall time per hop.  Launched on a side stream while the pipeline's hipGraphs replay on the
// others: how much longer does a dependent load take when the chip is busy?  (scripts/latency_under_load.py)
//   hipcc --offload-arch=gfx950 -shared -fPIC -O3 lat_probe.hip -o lat_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_lat_probe(const unsigned* next, int hops, unsigned long long* out) {
    if (threadIdx.x != 0) return;
    unsigned i = blockIdx.x * 977u % 1024u;                       // different start per block
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int h = 0; h < hops; ++h) i = __builtin_nontemporal_load(next + (size_t)i * 32);      // one 128-byte line per element
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 2] = r1 - r0;
    out[blockIdx.x * 2 + 1] = i;
}

extern "C" int lat_probe(const unsigned* next_dev, int hops, unsigned long long* out_dev, int blocks, void* stream) {
    k_lat_probe<<<blocks, 64, 0, (hipStream_t)stream>>>(next_dev, hops, out_dev);
    return (int)hipGetLastError();
}
