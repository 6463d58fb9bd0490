// Lab harness (GPU box): a 256x256-tile bf16 GEMM body with direct global->LDS staging, to find out what the detector
// head's GEMMs could reach BEFORE anything of it goes into the product kernel (DESIGN 9 item 1, 11 "bf16 analysis").
//   C[M][N] (bf16) = A[M][K] (bf16, K contiguous) x B[N][K]^T (bf16, K contiguous), f32 accumulate
// i.e. a 1x1 convolution on NHWC rows with the packed [cout][k] filter.  Build (CPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/bf16_lab.hip -o scripts/micro/_bin/bf16_lab
// Run (GPU box):  scripts/micro/_bin/bf16_lab            -> checks each variant against a plain kernel, then times the shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct GemmArgs { const void* A; const void* B; void* C; int M, N, K; int tiles_m, tiles_n; int skip_epilogue; int persistent; };

constexpr int BK = 64;                    // bf16 elements per stage = 128 B per row = eight 16-byte granules
constexpr unsigned OOB = 0x80000000u;

// LDS image of one operand stage: [rows][8 granules], granule g of row r stored in slot g ^ ((r >> 1) & 7): a 16-lane
// group of a ds_read_b128 (16 consecutive rows, same k) then touches 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz(int row, int g) { return g ^ ((row >> 1) & 7); }

// VARIANT 0: stage T+1 requested at the top of stage T, one barrier per stage (vmcnt(0) before it).
// VARIANT 1: barrier after k-step 2 of 4: k-step 3 and the next stage's first fragments run behind it; stage T+2 is
//            requested right after the barrier into the buffer every wave has just finished reading.
template <int VARIANT, int TMW>                               // TMW: 32-row tiles per wave: 4 -> 256-row tile, 2 -> 128-row tile
__global__ void __launch_bounds__(512) k_gemm256(const GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)          // (the host pass drops an instantiation whose body casts to an LDS pointer in dependent code)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int BM = 64 * TMW, TILE = 256, STAGE_BYTES = TILE * BK * 2, A_STAGE = BM * BK * 2;
    char* As = lds;                                                 // [2][BM][128 B]
    char* Bs = lds + 2 * A_STAGE;                                   // [2][256][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                        // 2 x 4 waves, 128 x 64 outputs each
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware tile map: consecutive workgroup ids go to different XCDs; give each XCD a contiguous run of tiles
    const int nwg = p.tiles_m * p.tiles_n;
    // persistent form: gridDim.x workgroups (one per CU) walk the tiles round by round; a tile's output stores are fire
    // and forget, so they drain while the next tile's operands stream in and multiply
    const int nround = p.persistent ? (int)gridDim.x : nwg;
    int bid0 = blockIdx.x;
    {
        const int q = nround / 8, r = nround % 8, xcd = bid0 % 8;
        bid0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid0 / 8;
    }
  for (int bid = bid0; bid < nwg; bid += nround) {
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n;           // column tiles of one row tile adjacent: A rows shared in L2
    const int m0 = tm * BM, n0 = tn * TILE;

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)((size_t)p.M * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)((size_t)p.N * p.K * 2), 0x00020000);

    // staging: one wave instruction = 64 lanes x 16 B = 8 rows x 128 B, LDS destination lane-linear; the wave's 4 + 4
    // instructions per stage cover rows wave*32 + i*8 + (lane >> 3) of each operand; lane's slot = lane & 7, it fetches
    // the granule that the swizzle stores there
    constexpr int PA = BM / 64;
    unsigned a_off[PA], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 8 + i * 64 + (lane >> 3);
        const int g = swz(row, lane & 7);
        if (i < PA) a_off[i] = m0 + row < p.M ? (unsigned)(((size_t)(m0 + row) * p.K + g * 8) * 2) : OOB;
        b_off[i] = n0 + row < p.N ? (unsigned)(((size_t)(n0 + row) * p.K + g * 8) * 2) : OOB;
    }
    auto stage = [&](int kt, int buf) {
        const int koff = kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < PA) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(As + buf * A_STAGE + (wave * 8 + i * 64) * 128), 16, a_off[i], koff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)(Bs + buf * STAGE_BYTES + (wave * 8 + i * 64) * 128), 16, b_off[i], koff, 0, 0);
        }
    };
    // fragment addresses: row = tile * 32 + li; k-step s of 4: granule 2 s + lh
    auto frag = [&](const char* base, int row, int s) -> bf16x8 {
        return *reinterpret_cast<const bf16x8*>(base + row * 128 + swz(row, 2 * s + lh) * 16);
    };

    f32x16 acc[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / BK;
    const int arow = wm * (32 * TMW) + li, brow = wn * 64 + li;

    if constexpr (VARIANT == 0) {
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
            const char* a = As + buf * A_STAGE;
            const char* b = Bs + buf * STAGE_BYTES;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 fa[TMW], fb[2];
#pragma unroll
                for (int i = 0; i < TMW; ++i) fa[i] = frag(a, arow + i * 32, s);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = frag(b, brow + j * 32, s);
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        bf16x8 na[TMW], nb[2];                                        // k-step 0 of the NEXT stage
        stage(0, 0);
        if (nk > 1) stage(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < TMW; ++i) na[i] = frag(As, arow + i * 32, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) nb[j] = frag(Bs, brow + j * 32, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            const char* a = As + buf * A_STAGE;
            const char* b = Bs + buf * STAGE_BYTES;
            const char* an = As + (buf ^ 1) * A_STAGE;
            const char* bn = Bs + (buf ^ 1) * STAGE_BYTES;
            bf16x8 fa[4][TMW], fb[4][2];
#pragma unroll
            for (int i = 0; i < TMW; ++i) fa[0][i] = na[i];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[0][j] = nb[j];
#pragma unroll
            for (int s = 1; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TMW; ++i) fa[s][i] = frag(a, arow + i * 32, s);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[s][j] = frag(b, brow + j * 32, s);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
            // every fragment of this stage is in registers and this wave's share of stage kt+1 has landed
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage(kt + 2, buf);
            if (kt + 1 < nk) {
#pragma unroll
                for (int i = 0; i < TMW; ++i) na[i] = frag(an, arow + i * 32, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) nb[j] = frag(bn, brow + j * 32, 0);
            }
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[3][i], fb[3][j], acc[i][j], 0, 0, 0);
        }
    }

    // plain epilogue (lab): lane owns column li of each 32-wide tile, rows 8 (e >> 2) + 4 lh + (e & 3)
    __bf16* C = reinterpret_cast<__bf16*>(p.C);
    if (p.skip_epilogue) { if (acc[0][0][0] == 123.456f) C[0] = (__bf16)1.0f; continue; }      // (main loop + prologue only)
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (32 * TMW) + i * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
                if (m < p.M && n < p.N) C[(size_t)m * p.N + n] = (__bf16)acc[i][j][e];
            }
        }
  }
#endif
}

__global__ void k_ref(const __bf16* A, const __bf16* B, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)B[(size_t)n * K + k];
    C[(size_t)m * N + n] = s;
}

__global__ void k_fill(__bf16* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (__bf16)(((int)(h & 0xffff) - 32768) / 32768.0f);
    }
}


template <int VARIANT, int TMW>
static void launch(GemmArgs a, hipStream_t s) {
    static bool once = false;
    const int ldsb = 2 * (64 * TMW + 256) * BK * 2;
    a.tiles_m = (a.M + 64 * TMW - 1) / (64 * TMW);
    if (!once) { CHECK(hipFuncSetAttribute((const void*)k_gemm256<VARIANT, TMW>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb)); once = true; }
    const int nt = a.tiles_m * a.tiles_n;
    k_gemm256<VARIANT, TMW><<<a.persistent && nt > 256 ? 256 : nt, 512, ldsb, s>>>(a);
}

int main(int argc, char** argv) {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    struct Shape { const char* name; int M, N, K; };
    std::vector<Shape> shapes = {
        {"check", 1000, 520, 256}, {"s5a_2a x1", 14700, 512, 1024}, {"s5x_2a x1", 14700, 512, 2048}, {"s5_2b-as-1x1 x1", 14700, 512, 4608}, {"s5_2c x1", 14700, 2048, 512}, {"s5a_1 x1", 14700, 2048, 1024}, {"s5x_2a x4", 58800, 512, 2048}, {"s5_2c x4", 58800, 2048, 512},
        {"s5a_1 x4", 58800, 2048, 1024}, {"s5_2b-as-1x1 x4", 58800, 512, 4608}, {"4096^3", 4096, 4096, 4096}, {"8192^3", 8192, 8192, 8192}};
    for (const Shape& sh : shapes) {
        __bf16 *A, *B, *C;
        CHECK(hipMalloc(&A, (size_t)sh.M * sh.K * 2)); CHECK(hipMalloc(&B, (size_t)sh.N * sh.K * 2)); CHECK(hipMalloc(&C, (size_t)sh.M * sh.N * 2));
        k_fill<<<1024, 256, 0, s>>>(A, (size_t)sh.M * sh.K, 1u);
        k_fill<<<1024, 256, 0, s>>>(B, (size_t)sh.N * sh.K, 2u);
        GemmArgs a{A, B, C, sh.M, sh.N, sh.K, (sh.M + 255) / 256, (sh.N + 255) / 256, argc > 1 && !strcmp(argv[1], "noepi") && strcmp(sh.name, "check") ? 1 : 0, 0};
        const bool try_persistent = argc > 1 && !strcmp(argv[1], "persistent");
        for (int variant = 0; variant < (try_persistent ? 6 : 4); ++variant) {           // 0, 1: 256-row tiles; 2, 3: 128-row tiles; 4, 5: 0 and 2 persistent
            a.persistent = variant >= 4;
            auto run = [&]() { if (variant == 0 || variant == 4) launch<0, 4>(a, s); else if (variant == 1) launch<1, 4>(a, s); else if (variant == 2 || variant == 5) launch<0, 2>(a, s); else launch<1, 2>(a, s); };
            if (!strcmp(sh.name, "check")) {
                float* R;
                CHECK(hipMalloc(&R, (size_t)sh.M * sh.N * 4));
                k_ref<<<dim3((sh.N + 255) / 256, sh.M), 256, 0, s>>>(A, B, R, sh.M, sh.N, sh.K);
                CHECK(hipMemsetAsync(C, 0xff, (size_t)sh.M * sh.N * 2, s));
                run();
                CHECK(hipStreamSynchronize(s));
                std::vector<float> r((size_t)sh.M * sh.N);
                std::vector<unsigned short> c((size_t)sh.M * sh.N);
                CHECK(hipMemcpy(r.data(), R, r.size() * 4, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(c.data(), C, c.size() * 2, hipMemcpyDeviceToHost));
                double worst = 0;
                for (size_t i = 0; i < r.size(); ++i) {
                    unsigned u = (unsigned)c[i] << 16; float f; memcpy(&f, &u, 4);
                    const double d = fabs((double)f - r[i]) / fmax(1.0, fabs((double)r[i]));
                    if (!(d <= worst)) worst = d;
                }
                printf("variant %d check: worst relative error %.3g (bf16 output: <= 4e-3 expected)\n", variant, worst);
                CHECK(hipFree(R));
                continue;
            }
            for (int i = 0; i < 3; ++i) run();
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, s));
                for (int i = 0; i < 10; ++i) run();
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / 10 < best) best = ms / 10;
            }
            printf("%-18s M=%6d N=%5d K=%5d  variant %d  %9.1f us  %7.1f TFLOP/s  (%d tiles)\n", sh.name, sh.M, sh.N, sh.K, variant,
                   best * 1e3, 2.0 * sh.M * sh.N * sh.K / (best * 1e-3) / 1e12, ((sh.M + ((variant < 2 || variant == 4) ? 255 : 127)) / ((variant < 2 || variant == 4) ? 256 : 128)) * a.tiles_n);
        }
        CHECK(hipFree(A)); CHECK(hipFree(B)); CHECK(hipFree(C));
    }
    return 0;
}
