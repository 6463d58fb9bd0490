// Lab harness (GPU box): fp32 GEMM on the bf16 matrix cores by EXACT operand splitting ("bf16x6").
//   C[M][N] (f32) = A[M][K] (f32, K contiguous) x B[N][K]^T (f32, K contiguous)
// Every f32 operand is the exact sum of three bf16 values (8 + 8 + 8 mantissa bits):  a = a1 + a2 + a3,  b = b1 + b2 + b3.
// Of the nine partial products the six with i + j <= 4 are kept (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1); each is EXACT in
// f32 (8 x 8 bits) and the three dropped ones are below 2^-24 |ab|, i.e. under one f32 rounding of the product.  The sum
// over k accumulates in f32 inside v_mfma_f32_32x32x16_bf16, as the native v_mfma_f32_32x32x2_f32 does.
// gfx950: bf16 MFMA runs at 16x the f32 MFMA rate, so six of them per product are 2.67x the native fp32 matrix peak on
// paper (416 vs 157 TFLOP/s).  This harness measures what a conv-shaped main loop gets, and its error against fp64.
//   A is split in the loader (global f32 -> registers -> three bf16 planes in LDS); B (the filter) is split once, on the
//   host, into three bf16 planes [3][N][K].
// Build (CPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/x6_lab.hip -o scripts/micro/_bin/x6_lab
// Run (GPU box):    scripts/micro/_bin/x6_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { const float* A; const __bf16* Bp; float* C; int M, N, K; int tiles_m, tiles_n; int terms; };

constexpr int BK = 32;                // f32 elements of k per chunk = two k-steps of the 32x32x16 MFMA
#ifndef ROWB_DEF
#define ROWB_DEF 64
#endif
constexpr int ROWB = ROWB_DEF;        // LDS bytes per row per plane: 80 = 64 B + 16 B pad; 64 = unpadded rows, 16-byte slot s of row r stored at slot s ^ ((r >> 2) & 3)
__device__ __forceinline__ int swz(int row) { return ROWB == 64 ? (row >> 2) & 3 : 0; }
constexpr unsigned OOB = 0x80000000u;

__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 a1 = (__bf16)v[e];                       // round to nearest even (v_cvt_pk_bf16_f32)
        const float r1 = v[e] - (float)a1;                    // exact
        const __bf16 a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;                      // exact, <= 8 significant bits
        h[e] = a1; m[e] = a2; l[e] = (__bf16)r2;
    }
}

// BM x BN tile on WM x WN waves, each wave TM x TN tiles of 32x32.  One LDS buffer (six planes), the next chunk waits in
// registers: [compute chunk t from LDS | loads of chunk t+1 in flight] -> barrier -> split + store chunk t+1 -> barrier.
template <int TM, int TN, int WM, int WN, int TERMS, int DEPTH = 1, int SKIP = 0>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_x6(const Args p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    constexpr int PA = BM * 8 / NT;                       // 16-byte f32 pieces of A per thread per chunk (8 per row)
    constexpr int PB = BN * 4 / NT;                       // 16-byte bf16 pieces of one B plane per thread per chunk (4 per row)
    static_assert(PA >= 1 && PB >= 1, "tile too small for the thread count");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* As = lds;                                       // [3][BM][ROWB]
    char* Bs = lds + 3 * BM * ROWB;                       // [3][BN][ROWB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    {   // XCD-aware: consecutive ids go to different XCDs; give each XCD a contiguous run of tiles (column tiles of a row tile adjacent)
        const int nwg = p.tiles_m * p.tiles_n, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n, m0 = tm * BM, n0 = tn * BN;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const size_t plane = (size_t)p.N * p.K * 2;
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Bp), 0, (int)(3 * plane), 0x00020000);

    unsigned a_off[PA], b_off[PB];
    int a_lds[PA], b_lds[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pc = tid + NT * i, row = pc >> 3, g = pc & 7;
        a_off[i] = m0 + row < p.M ? (unsigned)(((size_t)(m0 + row) * p.K + g * 4) * 4) : OOB;
        a_lds[i] = row * ROWB + 16 * ((g >> 1) ^ swz(row)) + 8 * (g & 1);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int pc = tid + NT * i, row = pc >> 2, g = pc & 3;
        b_off[i] = n0 + row < p.N ? (unsigned)(((size_t)(n0 + row) * p.K + g * 8) * 2) : OOB;
        b_lds[i] = row * ROWB + 16 * (g ^ swz(row));
    }
    // staging register sets: chunk c waits in set c & 1 (DEPTH 2: requested TWO chunks ahead, a whole chunk period to land)
    f32x4 ra[DEPTH == 2 ? 2 : 1][PA];
    f32x4 rb[DEPTH == 2 ? 2 : 1][3][PB];
    auto load = [&](int kt, auto setc) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_off[i], kt * (BK * 4), 0));
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                rb[S][pl][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, b_off[i] == OOB ? OOB : b_off[i] + (unsigned)(pl * plane), kt * (BK * 2), 0));
    };
    constexpr int BUFB = 3 * (BM + BN) * ROWB;            // bytes of one LDS buffer (six planes)
    auto store = [&](auto setc, int buf = 0) {
        constexpr int S = decltype(setc)::value;
        if constexpr (SKIP & 2) { asm volatile("" :: "v"(ra[S][0]), "v"(rb[S][0][0]), "v"(rb[S][2][PB - 1]), "v"(ra[S][PA - 1])); return; }     // ladder: no LDS stores
        char* A0 = As + buf * BUFB; char* B0 = Bs + buf * BUFB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bf16x4 h, m, l;
            if constexpr (SKIP & 1) {                        // ladder: no split arithmetic (the same bytes stored)
                h = __builtin_bit_cast(bf16x4, __builtin_shufflevector(ra[S][i], ra[S][i], 0, 1));
                m = __builtin_bit_cast(bf16x4, __builtin_shufflevector(ra[S][i], ra[S][i], 2, 3)); l = h;
            } else
            split3(ra[S][i], h, m, l);
            *reinterpret_cast<bf16x4*>(A0 + 0 * BM * ROWB + a_lds[i]) = h;
            *reinterpret_cast<bf16x4*>(A0 + 1 * BM * ROWB + a_lds[i]) = m;
            *reinterpret_cast<bf16x4*>(A0 + 2 * BM * ROWB + a_lds[i]) = l;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(B0 + pl * BN * ROWB + b_lds[i]) = rb[S][pl][i];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, DEPTH == 2 ? 1 : 0>;

    f32x16 acc[TM][TN];
    f32x16 acc_lo[TM][TN];                                // SKIP & 32: the five small partial products accumulate apart from a1*b1
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc_lo[i][j][e] = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / BK;
    const char* abase = As + (wm * TM * 32 + li) * ROWB;
    const char* bbase = Bs + (wn * TN * 32 + li) * ROWB;
    const int koff[2] = {16 * (lh ^ swz(li)), 16 * ((2 + lh) ^ swz(li))};      // k-step s: logical slot 2 s + lh
    auto compute = [&](int buf = 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(abase + buf * BUFB + pl * BM * ROWB + i * 32 * ROWB + koff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(bbase + buf * BUFB + pl * BN * ROWB + j * 32 * ROWB + koff[s]);
            }
            // smallest terms first; (a index, b index) pairs with i + j <= 4 (0-based: ia + ib <= 2)
            constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 6 - TERMS; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if ((SKIP & 32) && t < 5) acc_lo[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[IA[t]][i], fb[IB[t]][j], acc_lo[i][j], 0, 0, 0);
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[IA[t]][i], fb[IB[t]][j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (DEPTH == 1) {
        load(0, I0{});
        store(I0{});
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load(kt + 1, I0{});
            compute();
            __syncthreads();                              // every wave is done reading chunk kt
            if (kt + 1 < nk) {
                store(I0{});
                __syncthreads();
            }
        }
    } else if constexpr (DEPTH == 3) {
        // TWO LDS buffers, one register set, ONE barrier per chunk: chunk kt multiplies from buffer kt & 1 while chunk kt+1
        // (in registers since the previous iteration) is split and stored into the other buffer and chunk kt+2 is requested
        load(0, I0{});
        store(I0{}, 0);
        if (nk > 1) load(1, I0{});
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                store(I0{}, (kt + 1) & 1);
                load(kt + 2 < nk ? kt + 2 : 0, I0{});
            }
            compute(kt & 1);
            __syncthreads();
        }
    } else {
        // chunk c in set c & 1.  Iteration kt: MFMAs of chunk kt from LDS | barrier | store chunk kt+1 (its loads were
        // issued during iteration kt-1) and request chunk kt+3 into the registers just freed | barrier
        auto sync = [&]() { if constexpr (!(SKIP & 8)) __syncthreads(); };
        if constexpr (SKIP & 16) {                           // stagger: the second workgroup of a CU starts half a chunk period late
            if ((int)blockIdx.x >= (int)gridDim.x / 2) { for (int z = 0; z < 12; ++z) __builtin_amdgcn_s_sleep(127); }
        }
        auto loadx = [&](int k, auto setc) { if constexpr (!(SKIP & 4)) load(k, setc); };
        load(0, I0{});
        load(nk > 1 ? 1 : 0, I1{});
        store(I0{});
        load(nk > 2 ? 2 : 0, I0{});
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            compute();
            sync();
            store(I1{});                                  // chunk kt+1
            loadx(kt + 3 < nk ? kt + 3 : 0, I1{});
            sync();
            compute();
            sync();
            if (kt + 2 < nk) {
                store(I0{});                              // chunk kt+2
                loadx(kt + 4 < nk ? kt + 4 : 0, I0{});
                sync();
            }
        }
        if (kt < nk) compute();
    }
    // plain epilogue: lane owns column li of each 32x32 tile, rows (e & 3) + 8 (e >> 2) + 4 lh
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * TM * 32 + i * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < p.M) p.C[(size_t)m * p.N + n] = (SKIP & 32) ? acc[i][j][e] + acc_lo[i][j][e] : acc[i][j][e];
            }
        }
    }
}

// ---- the same loop on v_mfma_f32_16x16x32_bf16 (one MFMA = the whole 32-deep chunk of a 16x16 tile): under the clock the
// chip holds in MFMA-dense loops this shape delivers 1.12-1.15x the FLOP/s of 32x32x16 (MI355X_MICROARCH.md, DVFS give-back 7).
// Lane l supplies A[row l & 15][k = 8 (l >> 4) .. +7]: one ds_read_b128 at logical slot l >> 4 of row l & 15.
// Swizzle for THIS read pattern: slot s of row r at s ^ F[(r >> 2) & 3], F = {0, 2, 3, 1} (each b128 lane group then meets 16 distinct slots).
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz16(int row) { const int q = (row >> 2) & 3; return (0x1320 >> (4 * q)) & 3; }      // F = {0, 2, 3, 1}

template <int TM, int TN, int WM, int WN>           // TM x TN tiles of 16x16 per wave
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_x6_16(const Args p) {
    constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN, NT = 64 * WM * WN;
    constexpr int PA = BM * 8 / NT, PB = BN * 4 / NT;
    static_assert(PA >= 1 && PB >= 1, "tile too small for the thread count");
    constexpr int RB = 64;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* As = lds;
    char* Bs = lds + 3 * BM * RB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, lr = lane & 15, ls = lane >> 4;
    int bid = blockIdx.x;
    {
        const int nwg = p.tiles_m * p.tiles_n, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n, m0 = tm * BM, n0 = tn * BN;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const size_t plane = (size_t)p.N * p.K * 2;
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Bp), 0, (int)(3 * plane), 0x00020000);
    unsigned a_off[PA], b_off[PB];
    int a_lds[PA], b_lds[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pc = tid + NT * i, row = pc >> 3, g = pc & 7;
        a_off[i] = m0 + row < p.M ? (unsigned)(((size_t)(m0 + row) * p.K + g * 4) * 4) : OOB;
        a_lds[i] = row * RB + 16 * ((g >> 1) ^ swz16(row)) + 8 * (g & 1);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int pc = tid + NT * i, row = pc >> 2, g = pc & 3;
        b_off[i] = n0 + row < p.N ? (unsigned)(((size_t)(n0 + row) * p.K + g * 8) * 2) : OOB;
        b_lds[i] = row * RB + 16 * (g ^ swz16(row));
    }
    f32x4 ra[2][PA];
    f32x4 rb[2][3][PB];
    auto load = [&](int kt, auto setc) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_off[i], kt * (BK * 4), 0));
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                rb[S][pl][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, b_off[i] == OOB ? OOB : b_off[i] + (unsigned)(pl * plane), kt * (BK * 2), 0));
    };
    auto store = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bf16x4 h, m, l;
            split3(ra[S][i], h, m, l);
            *reinterpret_cast<bf16x4*>(As + 0 * BM * RB + a_lds[i]) = h;
            *reinterpret_cast<bf16x4*>(As + 1 * BM * RB + a_lds[i]) = m;
            *reinterpret_cast<bf16x4*>(As + 2 * BM * RB + a_lds[i]) = l;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(Bs + pl * BN * RB + b_lds[i]) = rb[S][pl][i];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    f32x4v acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / BK;
    const char* abase = As + (wm * TM * 16 + lr) * RB + 16 * (ls ^ swz16(lr));      // tile bases are multiples of 16 rows: (row >> 2) & 3 is the lane's own
    const char* bbase = Bs + (wn * TN * 16 + lr) * RB + 16 * (ls ^ swz16(lr));
    auto compute = [&]() {
        bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(abase + pl * BM * RB + i * 16 * RB);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(bbase + pl * BN * RB + j * 16 * RB);
        }
        constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[IA[t]][i], fb[IB[t]][j], acc[i][j], 0, 0, 0);
    };
    load(0, I0{});
    load(nk > 1 ? 1 : 0, I1{});
    store(I0{});
    load(nk > 2 ? 2 : 0, I0{});
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        compute();
        __syncthreads();
        store(I1{});
        load(kt + 3 < nk ? kt + 3 : 0, I1{});
        __syncthreads();
        compute();
        __syncthreads();
        if (kt + 2 < nk) {
            store(I0{});
            load(kt + 4 < nk ? kt + 4 : 0, I0{});
            __syncthreads();
        }
    }
    if (kt < nk) compute();
    // C/D of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 16 + j * 16 + lr;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * TM * 16 + i * 16 + ls * 4 + e;
                if (m < p.M) p.C[(size_t)m * p.N + n] = acc[i][j][e];
            }
    }
}

template <int TM, int TN, int WM, int WN>
static float run16(const Args& a0, int reps, const char* name) {
    constexpr int BM = 16 * TM * WM, BN = 16 * TN * WN;
    Args a = a0;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)3 * (BM + BN) * 64;
    CHECK(hipFuncSetAttribute((const void*)k_gemm_x6_16<TM, TN, WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_gemm_x6_16<TM, TN, WM, WN><<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    if (reps <= 0) return 0.0f;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_gemm_x6_16<TM, TN, WM, WN><<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    printf("  %-34s M=%-6d N=%-5d K=%-5d  %8.1f us  %7.1f TFLOP/s fp32-equivalent  (%d workgroups, %zu B LDS)\n", name, a.M, a.N, a.K, us, 2.0 * a.M * a.N * a.K / us / 1e6, a.tiles_m * a.tiles_n, lds);
    return (float)us;
}

// ---- ONE workgroup per CU, 16 waves (4 per SIMD), 256x128 tile, TWO LDS buffers (147 KB), ONE barrier per chunk: chunk kt multiplies
// from buffer kt & 1 while chunk kt+1 (in registers since the previous iteration) is split and stored into the other buffer and
// chunk kt+2 is requested.  MODE 1: the split / store / load of the next chunk sits between the two k-steps of the current one.
template <int MODE>
__global__ void __launch_bounds__(1024) k_gemm_x6_db16(const Args p) {
    constexpr int TM = 2, TN = 1, WM = 4, WN = 4, BM = 256, BN = 128, NT = 1024, RB = 64;
    constexpr int PA = BM * 8 / NT;                       // 2
    constexpr int NBP = 3 * BN * 4;                       // 16-byte pieces of the three B planes per chunk: 1536
    constexpr int PBT = (NBP + NT - 1) / NT;              // 2 (the second one for tid < 512 only)
    constexpr int BUFB = 3 * (BM + BN) * RB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = p.tiles_m * p.tiles_n, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n, m0 = tm * BM, n0 = tn * BN;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const size_t plane = (size_t)p.N * p.K * 2;
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.Bp), 0, (int)(3 * plane), 0x00020000);
    unsigned a_off[PA], b_off[PBT];
    int a_lds[PA], b_lds[PBT];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pc = tid + NT * i, row = pc >> 3, g = pc & 7;
        a_off[i] = m0 + row < p.M ? (unsigned)(((size_t)(m0 + row) * p.K + g * 4) * 4) : OOB;
        a_lds[i] = row * RB + 16 * ((g >> 1) ^ swz(row)) + 8 * (g & 1);
    }
#pragma unroll
    for (int i = 0; i < PBT; ++i) {
        const int q = tid + NT * i, pl = q / (BN * 4), r = q % (BN * 4), row = r >> 2, g = r & 3;
        const bool ok = q < NBP && n0 + row < p.N;
        b_off[i] = ok ? (unsigned)(((size_t)(n0 + row) * p.K + g * 8) * 2 + pl * plane) : OOB;
        b_lds[i] = q < NBP ? 3 * BM * RB + pl * BN * RB + row * RB + 16 * (g ^ swz(row)) : -1;
    }
    f32x4 ra[PA], rb[PBT];
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_off[i], kt * (BK * 4), 0));
#pragma unroll
        for (int i = 0; i < PBT; ++i) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, b_off[i], kt * (BK * 2), 0));
    };
    auto store = [&](int buf) {
        char* base = lds + buf * BUFB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bf16x4 h, m, l;
            split3(ra[i], h, m, l);
            *reinterpret_cast<bf16x4*>(base + 0 * BM * RB + a_lds[i]) = h;
            *reinterpret_cast<bf16x4*>(base + 1 * BM * RB + a_lds[i]) = m;
            *reinterpret_cast<bf16x4*>(base + 2 * BM * RB + a_lds[i]) = l;
        }
#pragma unroll
        for (int i = 0; i < PBT; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<f32x4*>(base + b_lds[i]) = rb[i];
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const int nk = p.K / BK;
    const int aoff = (wm * TM * 32 + li) * RB, boff = 3 * BM * RB + (wn * TN * 32 + li) * RB;
    const int koff[2] = {16 * (lh ^ swz(li)), 16 * ((2 + lh) ^ swz(li))};
    auto kstep = [&](int buf, int s) {
        const char* base = lds + buf * BUFB;
        bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(base + aoff + pl * BM * RB + i * 32 * RB + koff[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(base + boff + pl * BN * RB + j * 32 * RB + koff[s]);
        }
        constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[IA[t]][i], fb[IB[t]][j], acc[i][j], 0, 0, 0);
    };
    load(0);
    store(0);
    if (nk > 1) load(1);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if constexpr (MODE == 0) {
            if (kt + 1 < nk) { store(buf ^ 1); load(kt + 2 < nk ? kt + 2 : 0); }
            kstep(buf, 0);
            kstep(buf, 1);
        } else {
            kstep(buf, 0);
            if (kt + 1 < nk) { store(buf ^ 1); load(kt + 2 < nk ? kt + 2 : 0); }
            kstep(buf, 1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + li;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * TM * 32 + i * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < p.M) p.C[(size_t)m * p.N + n] = acc[i][j][e];
            }
        }
    }
}

template <int MODE>
static float run_db16(const Args& a0, int reps, const char* name) {
    Args a = a0;
    a.tiles_m = (a.M + 255) / 256; a.tiles_n = (a.N + 127) / 128;
    const size_t lds = (size_t)2 * 3 * (256 + 128) * 64;
    CHECK(hipFuncSetAttribute((const void*)k_gemm_x6_db16<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_gemm_x6_db16<MODE><<<a.tiles_m * a.tiles_n, 1024, lds>>>(a);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    if (reps <= 0) return 0.0f;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_gemm_x6_db16<MODE><<<a.tiles_m * a.tiles_n, 1024, lds>>>(a);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    printf("  %-34s M=%-6d N=%-5d K=%-5d  %8.1f us  %7.1f TFLOP/s fp32-equivalent  (%d workgroups, %zu B LDS)\n", name, a.M, a.N, a.K, us, 2.0 * a.M * a.N * a.K / us / 1e6, a.tiles_m * a.tiles_n, lds);
    return (float)us;
}

// reference: the native fp32 matrix instruction, same tile walk, no pipelining (for the error comparison only)
__global__ void __launch_bounds__(256) k_gemm_f32_ref(const Args p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x % p.tiles_n;
    const int m = tm * 64 + (wave >> 1) * 32 + li, n = tn * 64 + (wave & 1) * 32 + li;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const float* B = reinterpret_cast<const float*>(p.Bp);
    for (int k = 0; k < p.K; k += 2) {
        const float a = m < p.M ? p.A[(size_t)m * p.K + k + lh] : 0.0f;
        const float b = n < p.N ? B[(size_t)n * p.K + k + lh] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const int mb = tm * 64 + (wave >> 1) * 32 + 4 * lh;
    for (int e = 0; e < 16; ++e) {
        const int mm = mb + (e & 3) + 8 * (e >> 2);
        if (mm < p.M && n < p.N) p.C[(size_t)mm * p.N + n] = acc[e];
    }
}

static inline unsigned short f2bf(float f) {                     // round to nearest even
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int TM, int TN, int WM, int WN, int TERMS, int DEPTH = 1, int SKIP = 0>
static float run(const Args& a0, int reps, const char* name, bool quiet = false) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    Args a = a0;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)3 * (BM + BN) * ROWB * (DEPTH == 3 ? 2 : 1);
    CHECK(hipFuncSetAttribute((const void*)k_gemm_x6<TM, TN, WM, WN, TERMS, DEPTH, SKIP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_gemm_x6<TM, TN, WM, WN, TERMS, DEPTH, SKIP><<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    if (reps <= 0) return 0.0f;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_gemm_x6<TM, TN, WM, WN, TERMS, DEPTH, SKIP><<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, tf = 2.0 * a.M * a.N * a.K / us / 1e6;
    if (!quiet) printf("  %-34s M=%-6d N=%-5d K=%-5d  %8.1f us  %7.1f TFLOP/s fp32-equivalent  (%d workgroups, %zu B LDS)\n", name, a.M, a.N, a.K, us, tf, a.tiles_m * a.tiles_n, lds);
    return (float)us;
}

int main(int argc, char** argv) {
    // ---- 1. numerics: error against fp64 of (a) the native f32 MFMA, (b) x6, (c) x3 (a1b1 + a1b2 + a2b1: "bf16x3")
    {
        const int M = 256, N = 256;
        for (int K : {512, 4608, -4608}) {                     // (negative: the same K with ALL-POSITIVE operands -- nothing cancels)
            const bool positive = K < 0;
            K = K < 0 ? -K : K;
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            srand(7);
            for (auto& v : A) v = positive ? 0.5f + 0.5f * (float)rand() / RAND_MAX : (float)rand() / RAND_MAX * 2.0f - 1.0f;                 // activations: O(1), mixed sign
            for (auto& v : B) v = positive ? 0.025f + 0.025f * (float)rand() / RAND_MAX : ((float)rand() / RAND_MAX * 2.0f - 1.0f) * 0.05f;        // filters
            if (positive) printf("  -- all-positive operands\n");
            std::vector<unsigned short> Bp((size_t)3 * N * K);
            for (size_t i = 0; i < B.size(); ++i) {
                const unsigned short h = f2bf(B[i]); const float r1 = B[i] - bf2f(h);
                const unsigned short m = f2bf(r1); const float r2 = r1 - bf2f(m);
                Bp[i] = h; Bp[B.size() + i] = m; Bp[2 * B.size() + i] = f2bf(r2);
                if (bf2f(h) + bf2f(m) + bf2f(f2bf(r2)) != B[i]) { printf("split is not exact at %zu\n", i); return 1; }
            }
            float *dA, *dB, *dC; __bf16* dBp;
            CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4)); CHECK(hipMalloc(&dBp, Bp.size() * 2));
            CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
            std::vector<double> ref((size_t)M * N), mag((size_t)M * N);
            for (int m = 0; m < M; ++m)
                for (int n = 0; n < N; ++n) {
                    double s = 0, sa = 0;
                    for (int k = 0; k < K; ++k) { const double t = (double)A[(size_t)m * K + k] * B[(size_t)n * K + k]; s += t; sa += fabs(t); }
                    ref[(size_t)m * N + n] = s; mag[(size_t)m * N + n] = sa;
                }
            std::vector<float> C((size_t)M * N);
            auto report = [&](const char* what) {
                CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
                double worst = 0, mean = 0, worst_abs = 0;
                for (size_t i = 0; i < C.size(); ++i) { const double e = fabs(C[i] - ref[i]); worst = fmax(worst, e / mag[i]); mean += e / mag[i]; worst_abs = fmax(worst_abs, e / fmax(1.0, fabs(ref[i]))); }
                printf("  K=%-5d %-44s max |err| / sum|a b| = %.3g   mean %.3g   max |err| / max(1,|c|) = %.3g\n", K, what, worst, mean / C.size(), worst_abs);
            };
            Args a = {dA, reinterpret_cast<const __bf16*>(dB), dC, M, N, K, M / 64, N / 64, 6};
            k_gemm_f32_ref<<<a.tiles_m * a.tiles_n, 256>>>(a);
            CHECK(hipDeviceSynchronize());
            report("native v_mfma_f32_32x32x2_f32");
            a.Bp = dBp;
            run<2, 2, 2, 2, 6>(a, 0, ""); report("bf16x6 (six exact partial products)");
            run<2, 2, 2, 2, 6, 2>(a, 0, ""); report("bf16x6, loads two chunks ahead");
            run<2, 2, 2, 2, 6, 2, 32>(a, 0, ""); report("bf16x6, a1*b1 and the five small terms in separate accumulators");
            run<2, 1, 2, 4, 6, 2>(a, 0, ""); report("bf16x6, 8 waves, loads two chunks ahead");
            run<2, 1, 2, 4, 6, 3>(a, 0, ""); report("bf16x6, 8 waves, LDS double buffer");
            run16<4, 2, 2, 4>(a, 0, ""); report("bf16x6 on v_mfma_f32_16x16x32_bf16, 128x128 8 waves");
            run_db16<1>(a, 0, ""); report("bf16x6, 16 waves 256x128, LDS double buffer");
            run<2, 2, 2, 2, 3>(a, 0, ""); report("bf16x3 (a1b1 + a1b2 + a2b1)");
            run<2, 2, 2, 2, 1>(a, 0, ""); report("plain bf16 (a1b1)");
            CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(dBp));
        }
    }
    // ---- 2. rate on the detector head's GEMM shapes (configs[1]: 300 RoIs x 49 positions = 14 700 rows)
    struct Shape { int M, N, K; const char* what; } shapes[] = {
        {14700, 512, 4608, "head 3x3 512->512 as a GEMM"}, {14700, 2048, 512, "head 512->2048"}, {14700, 512, 2048, "head 2048->512"},
        {2394, 512, 9216, "rpn_conv1 3x3 1024->512"}, {2394, 1024, 256, "res4 256->1024"}, {2394, 256, 2304, "res4 3x3 256->256"},
        {117600, 512, 4608, "head 3x3, 8 images"}, {4096, 4096, 4096, "4096^3"}};
    for (const Shape& s : shapes) {
        float *dA, *dC; __bf16* dBp;
        CHECK(hipMalloc(&dA, (size_t)s.M * s.K * 4)); CHECK(hipMalloc(&dC, (size_t)s.M * s.N * 4)); CHECK(hipMalloc(&dBp, (size_t)3 * s.N * s.K * 2));
        std::vector<float> A((size_t)s.M * s.K);
        for (size_t i = 0; i < A.size(); ++i) A[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
        CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        std::vector<unsigned short> Bp((size_t)3 * s.N * s.K);
        for (size_t i = 0; i < Bp.size(); ++i) Bp[i] = f2bf(((float)((i * 40503u) & 0xFFFF) / 65536.0f - 0.5f) * 0.05f);
        CHECK(hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
        Args a = {dA, dBp, dC, s.M, s.N, s.K, 0, 0, 6};
        printf("%s\n", s.what);
        run<2, 2, 2, 2, 6>(a, 10, "128x128, 4 waves (64x64 per wave)");
        run<2, 2, 2, 2, 6, 2>(a, 10, "128x128, 4 waves, loads 2 ahead");
        run<2, 1, 2, 4, 6>(a, 10, "128x128, 8 waves (64x32 per wave)");
        run<2, 1, 2, 4, 6, 2>(a, 10, "128x128, 8 waves, loads 2 ahead");
        run<2, 2, 4, 2, 6, 2>(a, 10, "256x128, 8 waves, loads 2 ahead");
        run<2, 1, 2, 4, 6, 2, 16>(a, 10, "128x128, 8 waves, 2 ahead, staggered");
        run_db16<0>(a, 10, "256x128, 16 waves, LDS double buffer");
        run_db16<1>(a, 10, "256x128, 16 waves, DB, store mid-chunk");
        run16<4, 2, 2, 4>(a, 10, "128x128, 8 waves, 16x16x32 MFMA");
        run16<4, 4, 2, 2>(a, 10, "128x128, 4 waves, 16x16x32 MFMA");
        run16<4, 4, 4, 2>(a, 10, "256x128, 8 waves, 16x16x32 MFMA");
        run<2, 1, 2, 4, 6, 2, 1>(a, 10, "  ladder: no split arithmetic");
        run<2, 1, 2, 4, 6, 2, 2>(a, 10, "  ladder: no LDS stores");
        run<2, 1, 2, 4, 6, 2, 4>(a, 10, "  ladder: no global loads");
        run<2, 1, 2, 4, 6, 2, 6>(a, 10, "  ladder: no loads, no stores");
        run<2, 1, 2, 4, 6, 2, 14>(a, 10, "  ladder: LDS reads + MFMAs only (no barriers)");
        run<2, 1, 2, 4, 1, 2, 14>(a, 10, "  ladder: LDS reads + ONE product only");
        run<2, 1, 2, 4, 6, 3>(a, 10, "128x128, 8 waves, LDS double buffer");
        run<2, 2, 2, 2, 6, 3>(a, 10, "128x128, 4 waves, LDS double buffer");
        run<2, 2, 4, 2, 6, 3>(a, 10, "256x128, 8 waves, LDS double buffer");
        run<2, 2, 2, 4, 6>(a, 10, "128x256, 8 waves (64x64 per wave)");
        run<2, 2, 4, 2, 6>(a, 10, "256x128, 8 waves (64x64 per wave)");
        run<1, 1, 2, 2, 6>(a, 10, "64x64, 4 waves");
        run<2, 2, 2, 2, 1>(a, 10, "(128x128 4 waves, ONE product: loop cost)");
        CHECK(hipFree(dA)); CHECK(hipFree(dC)); CHECK(hipFree(dBp));
    }
    return 0;
}
