// Microbenchmark ladder: what costs the fp32 MFMA pipe its rate in a conv-shaped main loop?
// One "chunk" = 64 MFMAs (v_mfma_f32_32x32x2_f32, 4 accumulators) per wave, as in the 128x128 conv tile.
//   V0 register-only MFMAs
//   V1 + 16 ds_read_b128 per chunk feeding the operands
//   V2 + 8 ds_write_b128 per chunk into the other buffer + one barrier per chunk (double buffering)
//   V3 + 8 global buffer-style 16-byte loads per chunk (L2 resident) feeding those writes
// 512 workgroups x 256 threads, 73.7 KB of LDS each (two per CU), like the conv launch.
//   hipcc --offload-arch=gfx950 -O3 mfma_ladder.hip -o mfma_ladder && ./mfma_ladder
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 256, STRIDE = 36;          // floats: 128 A rows + 128 B rows, 144-byte rows

template <int V>
__global__ void __launch_bounds__(256) k(const f32x4* __restrict__ g, float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [2][ROWS][STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * ROWS * STRIDE; i += 256) lds[i] = 1e-3f * (i & 63);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f32x4 stage[8];
    const f32x4* gp = g + (size_t)(blockIdx.x & 63) * 2048 + tid;
    if (V >= 3) for (int q = 0; q < 8; ++q) stage[q] = gp[q * 256];
    else for (int q = 0; q < 8; ++q) stage[q] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(tid + q);
    float ca = 0.5f + tid * 1e-3f, cb = 0.25f - tid * 1e-3f;
    for (int c = 0; c < chunks; ++c) {
        const int buf = c & 1;
        if (V >= 2) {
            float* w = lds + (buf ^ 1) * ROWS * STRIDE;
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(w + ((tid >> 3) + 32 * q) * STRIDE + (tid & 7) * 4) = stage[q];
        }
        if (V >= 3) {
#pragma unroll
            for (int q = 0; q < 8; ++q) stage[q] = gp[((c + 1) & 7) * 64 + q * 256];
        }
        const float* a = lds + buf * ROWS * STRIDE + (wm * 64 + li) * STRIDE + lh * 4;
        const float* b = lds + buf * ROWS * STRIDE + (128 + wn * 64 + li) * STRIDE + lh * 4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 fa[2], fb[2];
            if (V >= 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * STRIDE + kk * 8); fb[i] = *reinterpret_cast<const f32x4*>(b + i * 32 * STRIDE + kk * 8); }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) { fa[i] = f32x4{ca, cb, ca, cb}; fb[i] = f32x4{cb, ca, cb, ca}; }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (V >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    for (int q = 0; q < 8; ++q) s += stage[q][0];
    out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const f32x4* g, float* out, int blocks, int chunks) {
    const size_t lds = 2 * ROWS * STRIDE * sizeof(float);
    (void)hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k<V><<<blocks, 256, lds>>>(g, out, chunks);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) k<V><<<blocks, 256, lds>>>(g, out, chunks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 10.0 * blocks * 4 * chunks * 64 * 4096.0;
    printf("V%d blocks=%d: %.3f ms per launch  %.1f TFLOP/s\n", V, blocks, ms / 10, flops / ms / 1e9);
}

int main() {
    f32x4* g; float* out;
    (void)hipMalloc(&g, 64 * 2048 * 16 + 4096 * 16); (void)hipMemset(g, 0, 64 * 2048 * 16 + 4096 * 16);
    (void)hipMalloc(&out, 1024 * 256 * 4);
    for (int blocks : {512, 460}) {
        run<0>(g, out, blocks, 144); run<1>(g, out, blocks, 144); run<2>(g, out, blocks, 144); run<3>(g, out, blocks, 144);
    }
    return 0;
}
