// What does writing a (rows x 2048) bf16 tensor cost by itself?  Persistent workgroups, 128x128-tile order, 16-byte pieces of whole
// rows (the bf16 conv epilogue's store pattern): stores only; stores + the residual read; 8-byte pieces (the swapped-operand form).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/store_probe.hip -o scripts/micro/_bin/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: 16-byte row pieces, stores only; 1: + residual read; 2: 8-byte pieces, 32 rows x 16 B per wave instruction; 3: mode 2 + residual
__global__ void __launch_bounds__(512) k_store(const char* res, char* y, int rows, int cout, int tiles_n, int ntiles) {
    const int tid = threadIdx.x;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        if (MODE < 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = tm * 128 + q * 32 + (tid >> 4);
                if (m < rows) {
                    const size_t o = ((size_t)m * cout + tn * 128 + (tid & 15) * 8) * 2;
                    i32x4 v = {tid, q, tile, 1};
                    if (MODE == 1) { const i32x4 r = *reinterpret_cast<const i32x4*>(res + o); v += r; }
                    *reinterpret_cast<i32x4*>(y + o) = v;
                }
            }
        } else {
            const int wave = tid >> 6, lane = tid & 63, wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = tm * 128 + wm * 64 + i * 32 + li;
                    if (m < rows) {
                        const size_t o = ((size_t)m * cout + tn * 128 + wn * 32 + 8 * g + 4 * lh) * 2;
                        i32x2 v = {tid, tile};
                        if (MODE == 3) { const i32x2 r = *reinterpret_cast<const i32x2*>(res + o); v += r; }
                        *reinterpret_cast<i32x2*>(y + o) = v;
                    }
                }
        }
    }
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 117600, cout = argc > 2 ? atoi(argv[2]) : 2048;
    const size_t bytes = (size_t)rows * cout * 2;
    char *res, *y; CK(hipMalloc(&res, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMemset(res, 1, bytes));
    const int tiles_n = cout / 128, ntiles = ((rows + 127) / 128) * tiles_n;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 4; ++mode)
        for (int grid : {512, 1024, ntiles}) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 5; ++i) {
                    if (mode == 0) k_store<0><<<grid, 512>>>(res, y, rows, cout, tiles_n, ntiles);
                    if (mode == 1) k_store<1><<<grid, 512>>>(res, y, rows, cout, tiles_n, ntiles);
                    if (mode == 2) k_store<2><<<grid, 512>>>(res, y, rows, cout, tiles_n, ntiles);
                    if (mode == 3) k_store<3><<<grid, 512>>>(res, y, rows, cout, tiles_n, ntiles);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms / 5 < best ? ms / 5 : best;
            }
            printf("mode %d (%s%s) grid %5d: %.1f us  %.2f TB/s written%s\n", mode, mode < 2 ? "16-byte row pieces" : "8-byte pieces", (mode & 1) ? " + residual read" : "",
                   grid, best * 1e3, bytes / (best * 1e-3) / 1e12, (mode & 1) ? " (+ as much read)" : "");
        }
    return 0;
}
