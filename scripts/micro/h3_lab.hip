// Lab harness (GPU box): fp32 GEMM on the fp16 matrix cores by a TWO-way operand split with a scaled low part ("f16x3").
//   C[M][N] (f32) = A[M][K] (f32, K contiguous) x B[N][K]^T (f32, K contiguous)
// Every operand tensor is brought into fp16's range by ONE power of two (exact):  a' = a * 2^eA  with  max|a'| in [2^14, 2^15),
// then      ah = f16(a')                 (11 significant bits, round to nearest even)
//           al = f16((a' - ah) * 2^11)   (the exact residual, scaled up so that it has ah's exponent range: 11 more bits)
// so a' = ah + al * 2^-11 + eps with |eps| <= 2^-23 |a'| (the 13-bit residual loses its last bit when its top bit is set: half of the time), same for b.
//   acc0 += ah * bh                      (v_mfma_f32_32x32x16_f16: products exact, f32 accumulate)
//   acc1 += ah * bl + al * bh            (its own accumulator: it carries the 2^-11 weight)
//   [TERMS == 4:  acc2 += al * bl        (2^-22)]
//   c = (acc0 + acc1 * 2^-11 [+ acc2 * 2^-22]) * 2^-(eA + eB)
// THREE MFMAs per 32x32x16 block of products instead of the six of the three-way bf16 split (scripts/micro/x6_lab.hip): the
// fp32-equivalent ceiling of the matrix pipe doubles (2.5 PFLOP/s / 3 = 833 TFLOP/s on paper).  What is given up: the split is
// no longer exact -- operands carry 23-24 bits and (TERMS == 3) the al * bl term, <= 2^-22 |ab| and typically 2^-25 |ab| with a
// random sign, is dropped.  This harness measures the error against fp64 next to the native f32 MFMA and the rate of the
// conv-shaped main loops.
// Build (CPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/h3_lab.hip -o scripts/micro/_bin/h3_lab
// Run (GPU box):    scripts/micro/_bin/h3_lab [acc|rate|all]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// A: f32 [M][K] (ASRC 0) or two fp16 planes [2][M][K] already scaled (ASRC 1); Bp: two fp16 planes [2][N][K] already scaled
struct Args { const float* A; const _Float16* Ap; const _Float16* Bp; float* C; int M, N, K; int tiles_m, tiles_n; float sA, inv_sA, inv_sB; };

constexpr int BK = 32;                // f32 elements of k per chunk = two k-steps of the 32x32x16 MFMA
constexpr int ROWB = 64;              // LDS bytes per row per plane; 16-byte slot s of row r stored at slot s ^ ((r >> 2) & 3)
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
constexpr unsigned OOB = 0x80000000u;

// The same split on the mixed-precision FMA instructions (VOP3P v_fma_mix*: each source f32 or f16, result f32 or one f16 half):
//   ah = f16(a * s + 0)  [mixlo / mixhi: the product is exact, ONE rounding],  r = a * s - ah  [mix_f32, exact],  al = f16(r * 2048 + 0)
// three full-rate instructions per element, no packed-f32 multiply (half rate beside MFMAs) and no separate conversions.
__device__ __forceinline__ void split2_mix(const f32x4 v, float s, f16x4& h, f16x4& l) {
    const float k2048 = 2048.0f;
    unsigned hh[2], ll[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float r0, r1;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh[q]) : "v"(v[2 * q]), "s"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh[q]) : "v"(v[2 * q + 1]), "s"(s));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v[2 * q]), "s"(s), "v"(hh[q]));
        asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v[2 * q + 1]), "s"(s), "v"(hh[q]));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(ll[q]) : "v"(r0), "s"(k2048));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(ll[q]) : "v"(r1), "s"(k2048));
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    h = __builtin_bit_cast(f16x4, (u32x2){hh[0], hh[1]});
    l = __builtin_bit_cast(f16x4, (u32x2){ll[0], ll[1]});
}

__device__ __forceinline__ void split2(const f32x4 v, float s, f16x4& h, f16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * s;                             // exact (power of two)
        const _Float16 a1 = (_Float16)x;                      // round to nearest even
        const float r = x - (float)a1;                        // exact
        h[e] = a1; l[e] = (_Float16)(r * 2048.0f);
    }
}

// MODE 0: ONE LDS buffer (four planes), two staging register sets, loads requested two chunks ahead, two barriers per chunk
//         (the product's k_conv_igemm_x6 loop).  MODE 1: TWO LDS buffers, one register set, ONE barrier per chunk (k_conv_igemm_x6_db).
// MODE 2: MODE 1's buffers with the fragment reads software-pipelined ACROSS the barrier: two fragment register sets; the barrier sits
//         between the chunk's two k-steps, so a wave leaves it with twelve MFMAs in hand (k-step 1, read before the barrier) while the
//         next chunk's k-step 0 fragments come in -- no wave ever waits on an LDS read with an empty matrix pipe.
template <int TM, int TN, int WM, int WN, int TERMS, int MODE, int ASRC, int SKIP = 0>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_h3(const Args p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    constexpr int NPA = ASRC ? 2 * BM * 4 : BM * 8;       // 16-byte pieces of A per chunk (f32: 8 per row; planes: 4 per row per plane)
    constexpr int PA = (NPA + NT - 1) / NT;
    constexpr int NPB = 2 * BN * 4;
    constexpr int PB = (NPB + NT - 1) / NT;
    constexpr int BUFB = 2 * (BM + BN) * ROWB;            // one LDS buffer: A hi, A lo, B hi, B lo
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nwg = p.tiles_m * p.tiles_n, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n, m0 = tm * BM, n0 = tn * BN;
    const __amdgpu_buffer_rsrc_t arsrc = ASRC
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Ap), 0, (int)((size_t)2 * p.M * p.K * 2), 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)((size_t)p.M * p.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Bp), 0, (int)((size_t)2 * p.N * p.K * 2), 0x00020000);

    unsigned a_off[PA], b_off[PB];
    int a_lds[PA], b_lds[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int q = tid + NT * i;
        if constexpr (ASRC) {
            const int pl = q / (BM * 4), r = q % (BM * 4), row = r >> 2, g = r & 3;
            const bool ok = q < NPA && m0 + row < p.M;
            a_off[i] = ok ? (unsigned)(((size_t)pl * p.M * p.K + (size_t)(m0 + row) * p.K + g * 8) * 2) : OOB;
            a_lds[i] = q < NPA ? pl * BM * ROWB + row * ROWB + 16 * (g ^ swz(row)) : -1;
        } else {
            const int row = q >> 3, g = q & 7;
            a_off[i] = (q < NPA && m0 + row < p.M) ? (unsigned)(((size_t)(m0 + row) * p.K + g * 4) * 4) : OOB;
            a_lds[i] = q < NPA ? row * ROWB + 16 * ((g >> 1) ^ swz(row)) + 8 * (g & 1) : -1;
        }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int q = tid + NT * i, pl = q / (BN * 4), r = q % (BN * 4), row = r >> 2, g = r & 3;
        const bool ok = q < NPB && n0 + row < p.N;
        b_off[i] = ok ? (unsigned)(((size_t)pl * p.N * p.K + (size_t)(n0 + row) * p.K + g * 8) * 2) : OOB;
        b_lds[i] = q < NPB ? 2 * BM * ROWB + pl * BN * ROWB + row * ROWB + 16 * (g ^ swz(row)) : -1;
    }
    constexpr int NS = MODE == 0 ? 2 : 1;
    f32x4 ra[NS][PA], rb[NS][PB];
    auto load = [&](int kt, auto setc) {
        constexpr int S = decltype(setc)::value;
        if constexpr (SKIP & 4) return;
#pragma unroll
        for (int i = 0; i < PA; ++i) ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, a_off[i], kt * (ASRC ? BK * 2 : BK * 4), 0));
#pragma unroll
        for (int i = 0; i < PB; ++i) rb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, b_off[i], kt * (BK * 2), 0));
    };
    const float sA = p.sA;
    auto store = [&](auto setc, int buf) {
        constexpr int S = decltype(setc)::value;
        if constexpr (SKIP & 2) { asm volatile("" :: "v"(ra[S][0]), "v"(rb[S][0]), "v"(rb[S][PB - 1]), "v"(ra[S][PA - 1])); return; }
        char* base = lds + buf * BUFB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if (a_lds[i] < 0) continue;
            if constexpr (ASRC) {
                *reinterpret_cast<f32x4*>(base + a_lds[i]) = ra[S][i];
            } else {
                f16x4 h, l;
                if constexpr (SKIP & 1) {                    // ladder: no split arithmetic (the same bytes stored)
                    h = __builtin_bit_cast(f16x4, __builtin_shufflevector(ra[S][i], ra[S][i], 0, 1));
                    l = __builtin_bit_cast(f16x4, __builtin_shufflevector(ra[S][i], ra[S][i], 2, 3));
                } else if constexpr (SKIP & 64) split2_mix(ra[S][i], sA, h, l);
                else split2(ra[S][i], sA, h, l);
                *reinterpret_cast<f16x4*>(base + a_lds[i]) = h;
                *reinterpret_cast<f16x4*>(base + BM * ROWB + a_lds[i]) = l;
            }
        }
#pragma unroll
        for (int i = 0; i < PB; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<f32x4*>(base + b_lds[i]) = rb[S][i];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, NS - 1>;

    f32x16 acc0[TM][TN], acc1[TM][TN], acc2[TERMS == 4 ? TM : 1][TERMS == 4 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.0f; acc1[i][j][e] = 0.0f; if constexpr (TERMS == 4) acc2[i][j][e] = 0.0f; }

    const int nk = p.K / BK;
    const int aoff = (wm * TM * 32 + li) * ROWB, boff = 2 * BM * ROWB + (wn * TN * 32 + li) * ROWB;
    const int koff[2] = {16 * (lh ^ swz(li)), 16 * ((2 + lh) ^ swz(li))};      // k-step s: logical slot 2 s + lh
    auto compute = [&](int buf) {
        const char* base = lds + buf * BUFB;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const f16x8*>(base + aoff + pl * BM * ROWB + i * 32 * ROWB + koff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const f16x8*>(base + boff + pl * BN * ROWB + j * 32 * ROWB + koff[s]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (TERMS == 4) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[1][j], acc2[i][j], 0, 0, 0);
                    if constexpr (TERMS >= 3) {
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0][j], acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1][j], acc1[i][j], 0, 0, 0);
                    }
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], acc0[i][j], 0, 0, 0);
                }
        }
    };
    auto sync = [&]() { if constexpr (!(SKIP & 8)) __syncthreads(); };
    if constexpr (MODE == 2) {
        f16x8 fa[2][2][TM], fb[2][2][TN];
        auto readF = [&](auto setc, int buf, int s) {
            constexpr int S = decltype(setc)::value;
            const char* base = lds + buf * BUFB;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[S][pl][i] = *reinterpret_cast<const f16x8*>(base + aoff + pl * BM * ROWB + i * 32 * ROWB + koff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[S][pl][j] = *reinterpret_cast<const f16x8*>(base + boff + pl * BN * ROWB + j * 32 * ROWB + koff[s]);
            }
        };
        auto mfma = [&](auto setc) {
            constexpr int S = decltype(setc)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][1][i], fb[S][0][j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][0][i], fb[S][1][j], acc1[i][j], 0, 0, 0);
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][0][i], fb[S][0][j], acc0[i][j], 0, 0, 0);
                }
        };
        using J0 = std::integral_constant<int, 0>;
        using J1 = std::integral_constant<int, 1>;
        load(0, I0{});
        store(I0{}, 0);
        if (nk > 1) load(1, I0{});
        __syncthreads();
        readF(J0{}, 0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            readF(J1{}, buf, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma(J0{});
            if (kt + 1 < nk) {
                store(I0{}, buf ^ 1);
                load(kt + 2 < nk ? kt + 2 : 0, I0{});
            }
            __builtin_amdgcn_sched_barrier(0);
            sync();
            if (kt + 1 < nk) readF(J0{}, buf ^ 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma(J1{});
        }
    } else if constexpr (MODE == 1) {
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
            sync();
        }
    } else {
        load(0, I0{});
        load(nk > 1 ? 1 : 0, I1{});
        store(I0{}, 0);
        load(nk > 2 ? 2 : 0, I0{});
        __syncthreads();
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            compute(0);
            sync();
            store(I1{}, 0);                               // chunk kt+1
            load(kt + 3 < nk ? kt + 3 : 0, I1{});
            sync();
            compute(0);
            sync();
            if (kt + 2 < nk) {
                store(I0{}, 0);                           // chunk kt+2
                load(kt + 4 < nk ? kt + 4 : 0, I0{});
                sync();
            }
        }
        if (kt < nk) compute(0);
    }
    // plain epilogue: lane owns column li of each 32x32 tile, rows (e & 3) + 8 (e >> 2) + 4 lh
    const float w1 = 1.0f / 2048.0f, w2 = w1 * w1;
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
                float v = acc1[i][j][e];
                if constexpr (TERMS == 4) v += acc2[i][j][e] * w1;
                v = acc0[i][j][e] + v * w1;
                if (m < p.M) p.C[(size_t)m * p.N + n] = (v * p.inv_sA) * p.inv_sB;
            }
        }
    }
}

// reference: the native fp32 matrix instruction (for the error comparison only)
__global__ void __launch_bounds__(256) k_gemm_f32_ref(const float* A, const float* B, float* C, int M, int N, int K, int tiles_n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m = tm * 64 + (wave >> 1) * 32 + li, n = tn * 64 + (wave & 1) * 32 + li;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    for (int k = 0; k < K; k += 2) {
        const float a = m < M ? A[(size_t)m * K + k + lh] : 0.0f;
        const float b = n < N ? B[(size_t)n * K + k + lh] : 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const int mb = tm * 64 + (wave >> 1) * 32 + 4 * lh;
    for (int e = 0; e < 16; ++e) {
        const int mm = mb + (e & 3) + 8 * (e >> 2);
        if (mm < M && n < N) C[(size_t)mm * N + n] = acc[e];
    }
}

// ---- host helpers
static float pow2f(int e) { return ldexpf(1.0f, e); }
static int scale_exp(const std::vector<float>& v) {              // e with max|v| * 2^e in [2^14, 2^15)
    float mx = 0; for (float x : v) mx = fmaxf(mx, fabsf(x));
    if (mx == 0) return 0;
    int ex; frexpf(mx, &ex);                                      // mx = f * 2^ex, f in [0.5, 1)
    int e = 15 - ex;
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
static void split_host(const std::vector<float>& v, int e, std::vector<_Float16>& planes) {
    const size_t n = v.size();
    planes.resize(2 * n);
    const float s = pow2f(e);
    for (size_t i = 0; i < n; ++i) {
        const float x = v[i] * s;
        const _Float16 h = (_Float16)x;
        planes[i] = h; planes[n + i] = (_Float16)((x - (float)h) * 2048.0f);
    }
}

template <int TM, int TN, int WM, int WN, int TERMS, int MODE, int ASRC, int SKIP = 0>
static float run(const Args& a0, int reps, const char* name) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    Args a = a0;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * ROWB * (MODE >= 1 ? 2 : 1);
    auto kern = k_gemm_h3<TM, TN, WM, WN, TERMS, MODE, ASRC, SKIP>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    if (reps <= 0) return 0.0f;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) kern<<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, tf = 2.0 * a.M * a.N * a.K / us / 1e6;
    printf("  %-46s %8.1f us  %7.1f TFLOP/s fp32-eq  (%d workgroups, %zu B LDS)\n", name, us, tf, a.tiles_m * a.tiles_n, lds);
    fflush(stdout);
    return (float)us;
}


// ---- RING: both operands as fp16 planes, brought in by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write), a ring
// of STAGES chunk buffers (chunk kt + STAGES - 1 requested while chunk kt multiplies, ONE barrier per chunk), and -- S16 -- the
// 16x16x32 MFMA shape on the same wave tile (same LDS bytes, same products; the chip holds a higher clock on it under load).
// A wave-instruction fills 16 consecutive 64-byte rows (1 KB): lane l -> row l >> 2, physical slot l & 3 = logical slot ^ f(row >> 2 & 3).
__device__ __forceinline__ int swz16(int g) { return (0x1320 >> (4 * (g & 3))) & 3; }      // f = [0, 2, 3, 1]: conflict-free for both MFMA shapes' reads
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int TM, int TN, int WM, int WN, int S16, int STAGES, int SKIP = 0>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_h3_ring(const Args p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN, NW = WM * WN;
    constexpr int STAGEB = 2 * (BM + BN) * ROWB;          // A hi, A lo, B hi, B lo
    constexpr int NA = 2 * BM / 16, NI = 2 * (BM + BN) / 16;      // 1 KB pieces per stage: the first NA belong to A
    constexpr int PI = (NI + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int bid = blockIdx.x;
    {
        const int nwg = p.tiles_m * p.tiles_n, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int tm = bid / p.tiles_n, tn = bid % p.tiles_n, m0 = tm * BM, n0 = tn * BN;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Ap), 0, (int)((size_t)2 * p.M * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.Bp), 0, (int)((size_t)2 * p.N * p.K * 2), 0x00020000);
    // this wave's pieces: j = wave + NW * i
    unsigned g_off[PI];
    const int lrow = lane >> 2, lp = (lane & 3) ^ (S16 ? swz16(lane >> 4) : ((lane >> 4) & 3));
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int j = wave + NW * i;
        if (j < NA) {
            const int pl = j / (BM / 16), row = (j % (BM / 16)) * 16 + lrow;
            g_off[i] = (m0 + row < p.M) ? (unsigned)(((size_t)pl * p.M * p.K + (size_t)(m0 + row) * p.K + lp * 8) * 2) : OOB;
        } else {
            const int jb = j - NA, pl = jb / (BN / 16), row = (jb % (BN / 16)) * 16 + lrow;
            g_off[i] = (j < NI && n0 + row < p.N) ? (unsigned)(((size_t)pl * p.N * p.K + (size_t)(n0 + row) * p.K + lp * 8) * 2) : OOB;
        }
    }
    auto issue = [&](int kt, int stage) {
        if constexpr (SKIP & 4) return;
        char* base = lds + stage * STAGEB;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int j = wave + NW * i;
            if (j < NA) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(base + j * 1024), 16, g_off[i], kt * (BK * 2), 0, 0);
            else if (j < NI) __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lds_ptr_t)(base + j * 1024), 16, g_off[i], kt * (BK * 2), 0, 0);
        }
#endif
    };
    const int nk = p.K / BK;
    constexpr int SM = S16 ? 2 * TM : TM, SN = S16 ? 2 * TN : TN;       // sub-tiles per wave
    typedef float accv __attribute__((ext_vector_type(S16 ? 4 : 16)));
    accv acc0[SM][SN], acc1[SM][SN];
#pragma unroll
    for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int j = 0; j < SN; ++j)
#pragma unroll
            for (int e = 0; e < (S16 ? 4 : 16); ++e) { acc0[i][j][e] = 0.0f; acc1[i][j][e] = 0.0f; }
    const int li = lane & 31, lh = lane >> 5, l16 = lane & 15, lq = lane >> 4;
    auto compute = [&](int stage) {
        const char* base = lds + stage * STAGEB;
        if constexpr (S16) {
            const int aoff = (wm * TM * 32 + l16) * ROWB + 16 * (lq ^ swz16(l16 >> 2));
            const int boff = 2 * BM * ROWB + (wn * TN * 32 + l16) * ROWB + 16 * (lq ^ swz16(l16 >> 2));
            f16x8 fa[2][SM], fb[2][SN];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int j = 0; j < SN; ++j) fb[pl][j] = *reinterpret_cast<const f16x8*>(base + boff + pl * BN * ROWB + j * 16 * ROWB);
#pragma unroll
                for (int i = 0; i < SM; ++i) fa[pl][i] = *reinterpret_cast<const f16x8*>(base + aoff + pl * BM * ROWB + i * 16 * ROWB);
            }
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int j = 0; j < SN; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[1][i], fb[0][j], acc1[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[1][j], acc1[i][j], 0, 0, 0);
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[0][i], fb[0][j], acc0[i][j], 0, 0, 0);
                }
        } else {
            const int aoff = (wm * TM * 32 + li) * ROWB, boff = 2 * BM * ROWB + (wn * TN * 32 + li) * ROWB;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ko = 16 * ((2 * s + lh) ^ swz(li));
                f16x8 fa[2][TM], fb[2][TN];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const f16x8*>(base + aoff + pl * BM * ROWB + i * 32 * ROWB + ko);
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const f16x8*>(base + boff + pl * BN * ROWB + j * 32 * ROWB + ko);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0][j], acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1][j], acc1[i][j], 0, 0, 0);
                        acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], acc0[i][j], 0, 0, 0);
                    }
            }
        }
    };
    // prologue: STAGES - 1 chunks in flight
#pragma unroll
    for (int c = 0; c < STAGES - 1; ++c) if (c < nk) issue(c, c);
    // chunk 0 landed = all but the newest (STAGES - 2) chunks' pieces of this wave (short loops: wait for everything)
    if (nk >= STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((STAGES - 2) * PI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0, st_in = STAGES - 1;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + STAGES - 1 < nk) issue(kt + STAGES - 1, st_in);       // into the stage chunk kt - 1 was read from: everyone passed the last barrier
        compute(st);
        // chunk kt + 1 landed: at most the (STAGES - 2) newer chunks of this wave may stay in flight (fewer near the end: wait for all)
        if (kt + STAGES - 1 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((STAGES - 2) * PI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if constexpr (!(SKIP & 8)) __builtin_amdgcn_s_barrier();
        st = st + 1 == STAGES ? 0 : st + 1;
        st_in = st_in + 1 == STAGES ? 0 : st_in + 1;
    }
    const float w1 = 1.0f / 2048.0f;
    if constexpr (S16) {
#pragma unroll
        for (int j = 0; j < SN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 16 + l16;
            if (n >= p.N) continue;
#pragma unroll
            for (int i = 0; i < SM; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + wm * TM * 32 + i * 16 + 4 * lq + e;
                    if (m < p.M) p.C[(size_t)m * p.N + n] = ((acc0[i][j][e] + acc1[i][j][e] * w1) * p.inv_sA) * p.inv_sB;
                }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + li;
            if (n >= p.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * TM * 32 + i * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
                    if (m < p.M) p.C[(size_t)m * p.N + n] = ((acc0[i][j][e] + acc1[i][j][e] * w1) * p.inv_sA) * p.inv_sB;
                }
        }
    }
}

template <int TM, int TN, int WM, int WN, int S16, int STAGES, int SKIP = 0>
static float run_ring(const Args& a0, int reps, const char* name) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    Args a = a0;
    a.tiles_m = (a.M + BM - 1) / BM; a.tiles_n = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * ROWB * STAGES;
    auto kern = k_gemm_h3_ring<TM, TN, WM, WN, S16, STAGES, SKIP>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    if (reps <= 0) return 0.0f;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) kern<<<a.tiles_m * a.tiles_n, 64 * WM * WN, lds>>>(a);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, tf = 2.0 * a.M * a.N * a.K / us / 1e6;
    printf("  %-46s %8.1f us  %7.1f TFLOP/s fp32-eq  (%d workgroups, %zu B LDS)\n", name, us, tf, a.tiles_m * a.tiles_n, lds);
    fflush(stdout);
    return (float)us;
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { double u = urand(), v = urand(); if (u < 1e-12) u = 1e-12; return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }

static void accuracy() {
    const int M = 256, N = 256;
    const char* kinds[] = {"mixed sign, O(1) x O(0.05)", "all positive", "wide exponents 2^-12..2^12", "integers -60..60", "integers -3000..3000",
                           "adversarial: every residual +max", "tiny 1e-30 x 1e-6", "post-ReLU (half zeros) with one 1000x outlier"};
    for (int kind = 0; kind < 8; ++kind)
        for (int K : {512, 4608}) {
            std::vector<float> A((size_t)M * K), B((size_t)N * K);
            srand(7 + kind);
            for (auto& v : A) {
                switch (kind) {
                    case 0: v = (float)(urand() * 2 - 1); break;
                    case 1: v = (float)(0.5 + 0.5 * urand()); break;
                    case 2: v = (float)(nrand() * ldexp(1.0, rand() % 25 - 12)); break;
                    case 3: v = (float)(rand() % 121 - 60); break;
                    case 4: v = (float)(rand() % 6001 - 3000); break;
                    case 5: v = 1.0f + ldexpf(1.0f, -11) * (float)(0.9 + 0.09 * urand()); break;
                    case 6: v = (float)(nrand() * 1e-30); break;
                    default: v = (float)fmax(0.0, nrand()); break;
                }
            }
            if (kind == 7) A[12345 % A.size()] = 1000.0f;
            for (auto& v : B) {
                switch (kind) {
                    case 0: v = (float)((urand() * 2 - 1) * 0.05); break;
                    case 1: v = (float)(0.025 + 0.025 * urand()); break;
                    case 2: v = (float)(nrand() * ldexp(1.0, rand() % 25 - 12)); break;
                    case 3: v = (float)(rand() % 121 - 60); break;
                    case 4: v = (float)(rand() % 6001 - 3000); break;
                    case 5: v = 1.0f + ldexpf(1.0f, -11) * (float)(0.9 + 0.09 * urand()); break;
                    case 6: v = (float)(nrand() * 1e-6); break;
                    default: v = (float)(nrand() * 0.03); break;
                }
            }
            const int eA = scale_exp(A), eB = scale_exp(B);
            std::vector<_Float16> Ap, Bp;
            split_host(A, eA, Ap); split_host(B, eB, Bp);
            float *dA, *dB, *dC; _Float16 *dAp, *dBp;
            CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
            CHECK(hipMalloc(&dAp, Ap.size() * 2)); CHECK(hipMalloc(&dBp, Bp.size() * 2));
            CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
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
                double worst = 0, mean = 0; size_t exact = 0;
                for (size_t i = 0; i < C.size(); ++i) { const double e = fabs(C[i] - ref[i]); worst = fmax(worst, e / mag[i]); mean += e / mag[i]; exact += (double)C[i] == ref[i]; }
                printf("  %-46s K=%-5d %-30s max |err|/sum|ab| = %.3g   mean %.3g   exact %zu/%zu\n", kinds[kind], K, what, worst, mean / C.size(), exact, C.size());
            };
            k_gemm_f32_ref<<<(M / 64) * (N / 64), 256>>>(dA, dB, dC, M, N, K, N / 64);
            CHECK(hipDeviceSynchronize());
            report("native v_mfma_f32_32x32x2_f32");
            Args a = {dA, dAp, dBp, dC, M, N, K, 0, 0, pow2f(eA), pow2f(-eA), pow2f(-eB)};
            run<2, 1, 2, 4, 3, 0, 0>(a, 0, ""); report("f16x3 (3 MFMAs)");
            run<2, 1, 2, 4, 4, 0, 0>(a, 0, ""); report("f16x4 (+ al*bl)");
            run<2, 1, 4, 4, 3, 1, 0>(a, 0, ""); report("f16x3, 16 waves double buffer");
            run<2, 1, 2, 4, 3, 0, 1>(a, 0, ""); report("f16x3, A planes from memory");
            run<2, 1, 4, 4, 3, 1, 0, 64>(a, 0, ""); report("f16x3, split on v_fma_mix*");
            CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(dAp)); CHECK(hipFree(dBp));
        }
}

static void rate() {
    struct Shape { int M, N, K; const char* what; } shapes[] = {
        {14700, 512, 4608, "head 3x3 512->512 as a GEMM"}, {14700, 2048, 512, "head 512->2048"}, {14700, 512, 2048, "head 2048->512"},
        {2394, 512, 9216, "rpn_conv1 3x3 1024->512"}, {37101, 64, 576, "stage 2 3x3 64->64"}, {37101, 256, 64, "stage 2 64->256"},
        {9375, 512, 128, "stage 3 128->512"}, {9375, 128, 1152, "stage 3 3x3 128->128"}, {2394, 1024, 256, "res4 256->1024"},
        {117600, 512, 4608, "head 3x3, 8 images"}};
    for (const Shape& s : shapes) {
        float *dA, *dC; _Float16 *dAp, *dBp;
        CHECK(hipMalloc(&dA, (size_t)s.M * s.K * 4)); CHECK(hipMalloc(&dC, (size_t)s.M * s.N * 4));
        CHECK(hipMalloc(&dAp, (size_t)2 * s.M * s.K * 2)); CHECK(hipMalloc(&dBp, (size_t)2 * s.N * s.K * 2));
        std::vector<float> A((size_t)s.M * s.K), B((size_t)s.N * s.K);
        for (size_t i = 0; i < A.size(); ++i) A[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
        for (size_t i = 0; i < B.size(); ++i) B[i] = ((float)((i * 40503u) & 0xFFFF) / 65536.0f - 0.5f) * 0.05f;
        const int eA = scale_exp(A), eB = scale_exp(B);
        std::vector<_Float16> Ap, Bp;
        split_host(A, eA, Ap); split_host(B, eB, Bp);
        CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
        Args a = {dA, dAp, dBp, dC, s.M, s.N, s.K, 0, 0, pow2f(eA), pow2f(-eA), pow2f(-eB)};
        printf("%s  M=%d N=%d K=%d\n", s.what, s.M, s.N, s.K);
        const int R = 10;
        run<2, 1, 2, 4, 3, 0, 0>(a, R, "128x128 8w (64x32/wave), 1 buf, loads 2 ahead");
        run<2, 1, 4, 4, 3, 1, 0>(a, R, "256x128 16w, LDS double buffer");
        run<2, 1, 4, 4, 3, 1, 0, 64>(a, R, "256x128 16w, LDS double buffer, fma_mix split");
        run<2, 2, 4, 2, 3, 1, 0, 64>(a, R, "256x128 8w (64x64/wave) dbuf, fma_mix split");
        run<1, 1, 2, 2, 3, 0, 0, 64>(a, R, "64x64 4w, 1 buf, fma_mix split");
        run<2, 2, 2, 2, 3, 0, 0>(a, R, "128x128 4w (64x64/wave), 1 buf");
        run<2, 2, 2, 2, 3, 1, 0>(a, R, "128x128 4w (64x64/wave), double buffer");
        run<2, 2, 4, 2, 3, 1, 0>(a, R, "256x128 8w (64x64/wave), double buffer");
        run<2, 2, 2, 4, 3, 1, 0>(a, R, "128x256 8w (64x64/wave), double buffer");
        run<1, 1, 2, 2, 3, 0, 0>(a, R, "64x64 4w, 1 buf");
        run<1, 1, 2, 2, 3, 1, 0>(a, R, "64x64 4w, double buffer");
        run<2, 1, 2, 2, 3, 0, 0>(a, R, "128x64 4w (64x32/wave), 1 buf");
        run<2, 1, 2, 4, 4, 0, 0>(a, R, "128x128 8w, FOUR terms");
        run<2, 1, 4, 4, 4, 1, 0>(a, R, "256x128 16w double buffer, FOUR terms");
        run<2, 1, 2, 4, 3, 0, 1>(a, R, "128x128 8w, A as fp16 planes from memory");
        run<2, 1, 4, 4, 3, 1, 1>(a, R, "256x128 16w dbuf, A as fp16 planes");
        run<2, 2, 4, 2, 3, 1, 1>(a, R, "256x128 8w (64x64/wave) dbuf, A as fp16 planes");
        std::vector<float> c1((size_t)s.M * s.N), c2((size_t)s.M * s.N);
        CHECK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        run<2, 2, 4, 2, 3, 2, 1>(a, R, "PIPE 256x128 8w (64x64/wave), A planes");
        CHECK(hipMemcpy(c2.data(), dC, c2.size() * 4, hipMemcpyDeviceToHost));
        printf("    pipelined == double-buffer result bitwise: %s\n", memcmp(c1.data(), c2.data(), c1.size() * 4) ? "NO" : "yes");
        run<2, 2, 2, 4, 3, 2, 1>(a, R, "PIPE 128x256 8w (64x64/wave), A planes");
        run<2, 2, 4, 2, 3, 2, 0>(a, R, "PIPE 256x128 8w (64x64/wave), f32 A");
        run<2, 2, 2, 4, 3, 2, 0>(a, R, "PIPE 128x256 8w (64x64/wave), f32 A");
        run<2, 1, 4, 4, 3, 2, 1>(a, R, "PIPE 256x128 16w (64x32/wave), A planes");
        run<2, 1, 4, 4, 3, 2, 0>(a, R, "PIPE 256x128 16w (64x32/wave), f32 A");
        run<2, 2, 2, 2, 3, 2, 1>(a, R, "PIPE 128x128 4w (64x64/wave), A planes");
        run<2, 2, 4, 2, 3, 2, 1, 14>(a, R, "  ladder PIPE 256x128 8w: LDS reads + MFMAs only");
        run<2, 2, 4, 2, 3, 2, 1, 6>(a, R, "  ladder PIPE 256x128 8w: no loads, no stores");
        run<2, 2, 4, 2, 3, 1, 1, 14>(a, R, "  ladder dbuf 256x128 8w: LDS reads + MFMAs only");
        run<2, 1, 4, 4, 3, 1, 0, 1>(a, R, "  ladder 256x128 16w: no split arithmetic");
        run<2, 1, 4, 4, 3, 1, 0, 2>(a, R, "  ladder 256x128 16w: no LDS stores");
        run<2, 1, 4, 4, 3, 1, 0, 6>(a, R, "  ladder 256x128 16w: no loads, no stores");
        run<2, 1, 4, 4, 3, 1, 0, 14>(a, R, "  ladder 256x128 16w: LDS reads + MFMAs only");
        CHECK(hipFree(dA)); CHECK(hipFree(dC)); CHECK(hipFree(dAp)); CHECK(hipFree(dBp));
    }
}


static void ring() {
    struct Shape { int M, N, K; const char* what; } shapes[] = {
        {14700, 512, 4608, "head 3x3 512->512 as a GEMM"}, {14700, 2048, 512, "head 512->2048"}, {14700, 512, 2048, "head 2048->512"},
        {37101, 256, 64, "stage 2 64->256"}, {9375, 512, 128, "stage 3 128->512"}, {2394, 2560, 1024, "hoisted pair 1024->2560"}};
    for (const Shape& s : shapes) {
        float *dA, *dC; _Float16 *dAp, *dBp;
        CHECK(hipMalloc(&dA, (size_t)s.M * s.K * 4)); CHECK(hipMalloc(&dC, (size_t)s.M * s.N * 4));
        CHECK(hipMalloc(&dAp, (size_t)2 * s.M * s.K * 2)); CHECK(hipMalloc(&dBp, (size_t)2 * s.N * s.K * 2));
        std::vector<float> A((size_t)s.M * s.K), B((size_t)s.N * s.K);
        srand(3);
        for (size_t i = 0; i < A.size(); ++i) A[i] = (float)nrand();
        for (size_t i = 0; i < B.size(); ++i) B[i] = (float)(nrand() * 0.05);
        const int eA = scale_exp(A), eB = scale_exp(B);
        std::vector<_Float16> Ap, Bp;
        split_host(A, eA, Ap); split_host(B, eB, Bp);
        CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dAp, Ap.data(), Ap.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dBp, Bp.data(), Bp.size() * 2, hipMemcpyHostToDevice));
        Args a = {dA, dAp, dBp, dC, s.M, s.N, s.K, 0, 0, pow2f(eA), pow2f(-eA), pow2f(-eB)};
        printf("%s  M=%d N=%d K=%d  (random normal operands)\n", s.what, s.M, s.N, s.K);
        const int R = 20;
        std::vector<float> c1((size_t)s.M * s.N), c2((size_t)s.M * s.N);
        auto same = [&](const char* what) {
            CHECK(hipMemcpy(c2.data(), dC, c2.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0; size_t diff = 0;
            for (size_t i = 0; i < c1.size(); ++i) { diff += c1[i] != c2[i]; worst = fmax(worst, fabs((double)c1[i] - c2[i]) / fmax(1e-6, fabs((double)c1[i]))); }
            printf("    %s vs double-buffer result: %zu / %zu elements differ, worst relative %.3g\n", what, diff, c1.size(), worst);
        };
        run<2, 1, 4, 4, 3, 1, 1>(a, R, "256x128 16w dbuf, A as fp16 planes (product)");
        CHECK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        run<2, 2, 4, 2, 3, 1, 1>(a, R, "256x128 8w (64x64/wave) dbuf, A planes");
        run_ring<2, 1, 4, 4, 0, 2>(a, R, "RING 256x128 16w 32x32x16, 2 stages"); same("ring 32x32x16");
        run_ring<2, 1, 4, 4, 0, 3>(a, R, "RING 256x128 16w 32x32x16, 3 stages");
        run_ring<2, 1, 4, 4, 1, 3>(a, R, "RING 256x128 16w 16x16x32, 3 stages"); same("ring 16x16x32");
        run_ring<2, 2, 4, 2, 0, 3>(a, R, "RING 256x128 8w (64x64) 32x32x16, 3 stages");
        run_ring<2, 2, 4, 2, 1, 3>(a, R, "RING 256x128 8w (64x64) 16x16x32, 3 stages"); same("ring 8w 16x16x32");
        run_ring<2, 2, 2, 4, 1, 3>(a, R, "RING 128x256 8w (64x64) 16x16x32, 3 stages");
        run_ring<2, 1, 2, 4, 1, 4>(a, R, "RING 128x128 8w (64x32) 16x16x32, 4 stages");
        run_ring<2, 1, 2, 4, 1, 2>(a, R, "RING 128x128 8w (64x32) 16x16x32, 2 stages (2 wg/CU)");
        run_ring<2, 1, 4, 4, 0, 3, 12>(a, R, "  ladder RING 16w 32x32x16: LDS reads + MFMAs only");
        run_ring<2, 1, 4, 4, 1, 3, 12>(a, R, "  ladder RING 16w 16x16x32: LDS reads + MFMAs only");
        run_ring<2, 2, 4, 2, 1, 3, 12>(a, R, "  ladder RING 8w 16x16x32: LDS reads + MFMAs only");
        run_ring<2, 1, 4, 4, 1, 3, 4>(a, R, "  ladder RING 16w 16x16x32: no DMA (barriers kept)");
        CHECK(hipFree(dA)); CHECK(hipFree(dC)); CHECK(hipFree(dAp)); CHECK(hipFree(dBp));
    }
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    if (!strcmp(what, "acc") || !strcmp(what, "all")) accuracy();
    if (!strcmp(what, "rate") || !strcmp(what, "all")) rate();
    if (!strcmp(what, "ring")) ring();
    return 0;
}
