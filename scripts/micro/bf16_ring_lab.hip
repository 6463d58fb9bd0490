// bf16 conv-engine lab, round 4: RING forms of the 1x1 stride-1 layers (persistent workgroups, operands streamed through LDS slots
// across tile boundaries) -- four versions, all bit-identical to the product's tiled kernel, none faster:
//   v1  k_gemm_ring_bf16<RES, BK, NSLOT>      symmetric waves, row-piece epilogue through a 16 KB LDS strip
//   v2  k_gemm_ring2_bf16<RES, NSLOT, AUX>    MFMA operands swapped: a lane owns one output row, 8-byte stores straight from registers
//   v3  k_gemm_ring3_bf16<RES, NSLOT, NLOAD>  dedicated loader waves (the compute waves' vmcnt sees no operand request)
//   v4  k_gemm_ring4_bf16<RES, NSTORE>        dedicated store waves (bf16 half tiles handed over through LDS)
//   v5  k_gemm_ring5_bf16<RES>                sixteen waves, the epilogue of tile j spread over the chunk steps of tile j + 1
// Lab switches: -DFRCNN_RING_NOMUL / -DFRCNN_RING_NOEPI (v1: the request stream alone), -DFRCNN_RING_FAKEEPI (v2: accumulators
// consumed, nothing stored).  Dev tool, GPU box only.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude scripts/micro/bf16_ring_lab.hip -o scripts/micro/_bin/bf16_ring_lab
//   bf16_ring_lab <rows> <cin> <cout> <residual 0/1> <variant>     variant: v1_32_4 v1_64_3 v1_64_4 v2_2 v2_3 v2_4 v2_2nt v3_2_1 v3_2_2 v3_4_2 v4_2 v4_4 v5
#include "../../faster_rcnn_amd/csrc/conv_bf16.hip"
#include <string>
#include <vector>
#include <string.h>

namespace frcnn {
static char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }

// ------------------------------------------------------------------------------------------------------------------
// RING form for the 1x1 stride-1 layers of a batched pass (round 4): persistent workgroups, operands streamed through a
// ring of four LDS slots ACROSS tile boundaries.
//
// Per-workgroup phase stamps of the tiled kernel on 512 -> 2048 over 117 600 rows (scripts/micro/bf16_stamps.hip): 3.2 us
// from entry until the first chunk has landed, 8.6 us for eight chunks (1.1 us each: one chunk requested at a time, the
// matrix pipe ~20 % busy), 1.3 us accumulators -> LDS, 3.8 us row pieces + stores, 1.3 us until the next workgroup enters
// the CU: every phase waits on latency, and a CU's throughput is the operand bytes it has in flight (32 KB per workgroup
// during 44 % of its life).  The strip kernel above (16 KB in flight per CU) measured exactly that: 18 GB/s per CU.
// Here a workgroup walks a SEQUENCE of 128x128 tiles and never stops requesting: k-chunks are 32 deep (16 KB of A + B
// rows, 64-byte LDS rows, XOR-swizzled 16-byte slots), slot (t & 3) takes chunk t of the workgroup's whole sequence,
// chunk t + 3 is requested as chunk t is multiplied -- so three chunks (48 KB per workgroup, two workgroups per CU) are in
// flight during the main loop AND during a tile's epilogue, whose first chunks of the NEXT tile land meanwhile.  The
// epilogue turns the tile through a separate 32-row f32 strip (16 KB) with the row-piece form of k_conv_igemm_bf16; its
// residual pieces are requested at the tile's first chunk, its stores are branch-free (pieces outside the tensor ride
// on the buffer descriptor) so that `s_waitcnt vmcnt(n)` can count past them: requests complete in order.
// 1x1, stride 1, no padding (Y[m] = X[m] . W^T in either row layout), bf16 output, cout % 128 == 0, cin % 32 == 0,
// cin >= 128.  Same k order and epilogue arithmetic as the tiled forms: bit-identical.
template <bool RES, int BK, int NSLOT>
__global__ void __launch_bounds__(512) k_gemm_ring_bf16(const ConvArgsBf16 p) {
    constexpr int BM = 128, BN = 128, ROWB = BK * 2, SLOT = (BM + BN) * ROWB, EP = 4;
    constexpr int RPW = 1024 / ROWB;                       // rows a wave instruction lands (16 x 64 B or 8 x 128 B)
    constexpr int NP = BM / (8 * RPW);                     // wave instructions per operand per chunk and wave
    constexpr int NL = 2 * NP;
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    char* ring = smem_r;                                                   // [4][A 128 rows | B 128 rows][64 B]
    float* stg = reinterpret_cast<float*>(smem_r + NSLOT * SLOT);          // [32][128] f32
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
    const int nk = p.Kpad / BK, tiles_n = p.Cout / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    // the workgroups resident together hold a contiguous run of (row tile, column tile) ids; an XCD (blockIdx & 7) holds a
    // contiguous eighth of it: the column tiles of a row tile share their A rows in that XCD's L2
    const int G = gridDim.x;
    const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (first >= ntiles) return;
    const int T = ((ntiles - first + G - 1) / G) * nk;                     // chunks of this workgroup's whole sequence

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, (int)((size_t)p.M * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(RES ? p.residual : p.x), 0, RES ? (int)((size_t)p.M * p.Cout * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(p.y), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);

    // ---- request side: a wave instruction lands 16 rows x 64 B (lane-linear); slot s of row r holds granule s ^ ((r >> 2) & 3)
    const int drow = wave * RPW + lane / (ROWB / 16);                      // (+ 8 RPW per further pass)
    const int dgran = BK == 32 ? ((lane & 3) ^ ((drow >> 2) & 3)) * 16 : ((lane & 7) ^ ((drow >> 1) & 7)) * 16;
    int i_tile = first, i_c = 0;
    unsigned ia_off[NP], ib_off[NP];
    auto aim = [&](int L) {
        const int tm = L / tiles_n, tn = L - tm * tiles_n;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int m = tm * BM + drow + i * 8 * RPW, n = tn * BN + drow + i * 8 * RPW;
            ia_off[i] = (L < ntiles && m < p.M) ? (unsigned)((size_t)m * p.Cin * 2 + dgran) : OOB_OFFSET_B;
            ib_off[i] = L < ntiles ? (unsigned)((size_t)n * p.Kpad * 2 + dgran) : OOB_OFFSET_B;
        }
    };
    auto request = [&](int t) {
        char* dst = ring + (t % NSLOT) * SLOT + wave * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + i * 8192), 16, ia_off[i], i_c * ROWB, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + BM * ROWB + i * 8192), 16, ib_off[i], i_c * ROWB, 0, 0);
        }
#endif
        if (++i_c == nk) { i_c = 0; i_tile += G; aim(i_tile); }
    };
    aim(first);
#pragma unroll
    for (int t = 0; t < NSLOT - 1; ++t) request(t);                        // nk >= NSLOT - 1

    // ---- compute side
    const int arow = wm * 64 + li, brow = wn * 32 + li, sw = BK == 32 ? (li >> 2) & 3 : (li >> 1) & 7;
    const int prow = tid >> 4, pcol = (tid & 15) * 8, swap = (tid & 8) ? 4 : 0;
    i32x4 rpre[EP];
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    int tile = first, c = 0;
    bool later_tile = false;
    for (int t = 0; t < T; ++t) {
        // chunk t must have landed.  Behind it in the queue: chunks t + 1, t + 2 (two wave instructions each), the residual
        // pieces requested at a c == 0 step and the stores of a c == nk - 1 step among the last three steps
        if (t >= T - (NSLOT - 2)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else {
            const int later = ((RES && c >= 1 && c <= NSLOT - 1) ? 1 : 0) + ((later_tile && c <= NSLOT - 2) ? 1 : 0);
            if (later == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NSLOT - 2) * NL) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NSLOT - 2) * NL + EP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NSLOT - 2) * NL + 2 * EP) : "memory");
        }
        __builtin_amdgcn_s_barrier();                                      // ... for every wave; and slot (t - 1) & 3 is free
        if (t + NSLOT - 1 < T) request(t + NSLOT - 1);
#ifndef FRCNN_RING_NOMUL                                                   // (lab builds: the request stream alone)
        {
            const char* a = ring + (t % NSLOT) * SLOT + arow * ROWB;
            const char* b = ring + (t % NSLOT) * SLOT + BM * ROWB + brow * ROWB;
#pragma unroll
            for (int st = 0; st < BK / 16; ++st) {
                const int slot = ((2 * st + lh) ^ sw) * 16;
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b + slot);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a + i * 32 * ROWB + slot);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
                }
            }
        }
#endif
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
        if (RES && c == 0) {
#pragma unroll
            for (int q = 0; q < EP; ++q) {
                const int m = m0 + q * 32 + prow;
                rpre[q] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, m < p.M ? (unsigned)(((size_t)m * p.Cout + n0 + pcol) * 2) : OOB_OFFSET_B, 0, 0);
            }
        }
#ifdef FRCNN_RING_NOEPI
        if (c == nk - 1) { tile += G; c = 0; later_tile = false; } else ++c;
        if (false) {
#else
        if (c == nk - 1) {
#endif
            // ---- epilogue of this tile: 32-row strips through the f32 LDS strip, row pieces as in k_conv_igemm_bf16
            const int ncl = wn * 32 + li;
            const float sc = p.scale ? p.scale[n0 + ncl] : 1.0f;
            const float sh = p.shift ? p.shift[n0 + ncl] : 0.0f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if (wm == (s4 >> 1)) {
                    float* dst = stg + (4 * lh) * BN + ncl;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[s4 & 1][e] * sc + sh;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const int m = m0 + s4 * 32 + prow;
                const float* src = stg + prow * BN + pcol;
                const f32x4 va = *reinterpret_cast<const f32x4*>(src + swap);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(src + (swap ^ 4));
                const f32x4 v0 = swap ? vb : va, v1 = swap ? va : vb;
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                if (RES) {
                    const bf16x8 r = __builtin_bit_cast(bf16x8, rpre[s4]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += (float)r[k];
                }
                bf16x8 ob;
#pragma unroll
                for (int k = 0; k < 8; ++k) ob[k] = (__bf16)activate_b(v[k], p.act);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ob), yrsrc,
                                                       m < p.M ? (unsigned)(((size_t)m * p.Cout + n0 + pcol) * 2) : OOB_OFFSET_B, 0, 0);
                if (s4 < 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                          // the next strip overwrites what was just read
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
            tile += G; c = 0; later_tile = true;
        } else {
#ifndef FRCNN_RING_NOEPI
            ++c;
#endif
        }
    }
}

// Ring form, second version: the MFMA operands SWAPPED (weights as the A operand, activations as B), so that a lane of the
// 32x32 accumulator owns ONE output row and four runs of four consecutive channels -- the epilogue needs no LDS and no
// barrier: every wave scales, adds its residual pieces (8-byte loads requested at the tile's first chunk), activates,
// rounds and stores 8 bytes per run straight from its registers, while the other waves and the request stream run on.
// (Same products summed over k in the same order: bit-identical to the other forms; tests compare.)
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <bool RES, int NSLOT, int AUX>
__global__ void __launch_bounds__(512) k_gemm_ring2_bf16(const ConvArgsBf16 p) {
    constexpr int BM = 128, BN = 128, BK = 64, ROWB = BK * 2, SLOT = (BM + BN) * ROWB, EP = 8, NL = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    char* ring = smem_q;                                                   // [NSLOT][A 128 rows | B 128 rows][128 B]
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
    const int nk = p.Kpad / BK, tiles_n = p.Cout / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (first >= ntiles) return;
    const int T = ((ntiles - first + G - 1) / G) * nk;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, (int)((size_t)p.M * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(RES ? p.residual : p.x), 0, RES ? (int)((size_t)p.M * p.Cout * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(p.y), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);

    const int drow = wave * 8 + (lane >> 3);                               // (+ 64 for the second pass)
    const int dgran = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;
    int i_tile = first, i_c = 0;
    unsigned ia_off[2], ib_off[2];
    auto aim = [&](int L) {
        const int tm = L / tiles_n, tn = L - tm * tiles_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tm * BM + drow + i * 64, n = tn * BN + drow + i * 64;
            ia_off[i] = (L < ntiles && m < p.M) ? (unsigned)((size_t)m * p.Cin * 2 + dgran) : OOB_OFFSET_B;
            ib_off[i] = L < ntiles ? (unsigned)((size_t)n * p.Kpad * 2 + dgran) : OOB_OFFSET_B;
        }
    };
    auto request = [&](int t) {
        char* dst = ring + (t % NSLOT) * SLOT + wave * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + i * 8192), 16, ia_off[i], i_c * ROWB, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + BM * ROWB + i * 8192), 16, ib_off[i], i_c * ROWB, 0, 0);
        }
#endif
        if (++i_c == nk) { i_c = 0; i_tile += G; aim(i_tile); }
    };
    aim(first);
#pragma unroll
    for (int t = 0; t < NSLOT - 1; ++t) request(t);                        // nk >= NSLOT - 1

    const int arow = wm * 64 + li, brow = wn * 32 + li, sw = (li >> 1) & 7;
    i32x2 rpre[EP];
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    int tile = first, c = 0;
    bool later_tile = false;
    for (int t = 0; t < T; ++t) {
        if (t >= T - (NSLOT - 2)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else {
            const int later = ((RES && c >= 1 && c <= NSLOT - 1) ? 1 : 0) + ((later_tile && c <= NSLOT - 2) ? 1 : 0);
            if (later == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NSLOT - 2) * NL) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NSLOT - 2) * NL + EP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NSLOT - 2) * NL + 2 * EP) : "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + NSLOT - 1 < T) request(t + NSLOT - 1);
        {
            const char* a = ring + (t % NSLOT) * SLOT + arow * ROWB;
            const char* b = ring + (t % NSLOT) * SLOT + BM * ROWB + brow * ROWB;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int slot = ((2 * st + lh) ^ sw) * 16;
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b + slot);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a + i * 32 * ROWB + slot);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc[i], 0, 0, 0);      // D[n][m]
                }
            }
        }
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * BM + wm * 64 + li, n0 = tn * BN + wn * 32 + 4 * lh;         // this lane's first row / first channel
        if (RES && c == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m0 + i * 32;
                    rpre[i * 4 + g] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rrsrc, m < p.M ? (unsigned)(((size_t)m * p.Cout + n0 + 8 * g) * 2) : OOB_OFFSET_B, 0, AUX));
                }
        }
#ifdef FRCNN_RING_FAKEEPI                                                  // (lab builds: the accumulators are consumed, nothing is stored)
        if (c == nk - 1) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) { sum += acc[i][e]; acc[i][e] = 0.0f; }
            if (sum == 12345.678f) reinterpret_cast<float*>(p.y)[tid] = sum;
            tile += G; c = 0; later_tile = false;
        } else ++c;
        if (false) {
#else
        if (c == nk - 1) {
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + 8 * g;
                const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + i * 32;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[i][4 * g + k] * sc[k] + sh[k];
                    if (RES) {
                        const bf16x4 r = __builtin_bit_cast(bf16x4, rpre[i * 4 + g]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += (float)r[k];
                    }
                    bf16x4 ob;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ob[k] = (__bf16)activate_b(v[k], p.act);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, ob), yrsrc,
                                                          m < p.M ? (unsigned)(((size_t)m * p.Cout + n) * 2) : OOB_OFFSET_B, 0, AUX);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
            tile += G; c = 0; later_tile = true;
        } else {
#ifndef FRCNN_RING_FAKEEPI
            ++c;
#endif
        }
    }
}

template <int NSLOT, int AUX>
static int launch_ring2_bf16(const ConvArgsBf16& a, hipStream_t s) {
    const size_t lds = (size_t)NSLOT * (128 + 128) * 128;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_gemm_ring2_bf16<true, NSLOT, AUX>, lds, "conv2d_bf16")) return e;
    static std::atomic<uint64_t> lds_seen2{0};
    if (int e = raise_lds_once(lds_seen2, (const void*)k_gemm_ring2_bf16<false, NSLOT, AUX>, lds, "conv2d_bf16")) return e;
    const int per_cu = lds <= 81920 ? 2 : 1;
    const int ntiles = ((a.M + 127) / 128) * (a.Cout / 128);
    const int grid = ntiles >= 256 * per_cu ? 256 * per_cu : ((ntiles + 7) / 8) * 8;
    if (a.residual) k_gemm_ring2_bf16<true, NSLOT, AUX><<<grid, 512, lds, s>>>(a);
    else k_gemm_ring2_bf16<false, NSLOT, AUX><<<grid, 512, lds, s>>>(a);
    return check_launch("conv2d_fwd_bf16 (ring)");
}

// Ring form, third version: LOADER waves.  `s_waitcnt vmcnt` retires a wave's requests in order, stores included: in the
// forms above the wait for the first chunk requested AFTER a tile's stores also waits for those stores to be acknowledged
// (1.5-3.7 us behind 32 KB per workgroup), once per tile -- 200-300 us of a 500 us launch (lab: the same loop with the
// accumulators consumed but nothing stored takes 260 us).  Here NLOAD extra waves do nothing but request chunks and wait for
// them (they never store), and the eight compute waves never wait for a chunk (they learn of its arrival from the
// workgroup barrier behind the loaders' wait); their own vmcnt only sees residual pieces and stores.
template <bool RES, int NSLOT, int NLOAD>
__global__ void __launch_bounds__(512 + 64 * NLOAD) k_gemm_ring3_bf16(const ConvArgsBf16 p) {
    constexpr int BM = 128, BN = 128, BK = 64, ROWB = BK * 2, SLOT = (BM + BN) * ROWB, EP = 8;
    constexpr int NLW = 32 / NLOAD;                                        // wave instructions (8 rows x 128 B) per chunk and loader wave
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    char* ring = smem_t;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = p.Kpad / BK, tiles_n = p.Cout / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (first >= ntiles) return;
    const int T = ((ntiles - first + G - 1) / G) * nk;

    if (wave >= 8) {
        // ---- loader wave
        const int lw = wave - 8;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, (int)((size_t)p.M * p.Cin * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000);
        int i_tile = first, i_c = 0;
        unsigned off[NLW];                                                 // instruction j of a chunk: j < 16 rows 8j.. of A, else rows 8 (j - 16).. of B
        auto aim = [&](int L) {
            const int tm = L / tiles_n, tn = L - tm * tiles_n;
#pragma unroll
            for (int q = 0; q < NLW; ++q) {
                const int j = lw * NLW + q;
                const int row = (j & 15) * 8 + (lane >> 3);
                const int gran = ((lane & 7) ^ ((row >> 1) & 7)) * 16;
                if (j < 16) { const int m = tm * BM + row; off[q] = (L < ntiles && m < p.M) ? (unsigned)((size_t)m * p.Cin * 2 + gran) : OOB_OFFSET_B; }
                else { const int n = tn * BN + row; off[q] = L < ntiles ? (unsigned)((size_t)n * p.Kpad * 2 + gran) : OOB_OFFSET_B; }
            }
        };
        auto request = [&](int t) {
            char* dst = ring + (t % NSLOT) * SLOT + lw * NLW * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int q = 0; q < NLW; ++q) {
                const int j = lw * NLW + q;
                if (j < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + q * 1024), 16, off[q], i_c * ROWB, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + q * 1024), 16, off[q], i_c * ROWB, 0, 0);
            }
#endif
            if (++i_c == nk) { i_c = 0; i_tile += G; aim(i_tile); }
        };
        aim(first);
#pragma unroll
        for (int t = 0; t < NSLOT - 1; ++t) request(t);
        for (int t = 0; t < T; ++t) {
            if (t >= T - (NSLOT - 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSLOT - 2) * NLW) : "memory");
            __builtin_amdgcn_s_barrier();                                  // chunk t is in LDS; slot (t - 1) % NSLOT is free
            if (t + NSLOT - 1 < T) request(t + NSLOT - 1);
        }
        return;
    }

    // ---- compute waves
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(RES ? p.residual : p.x), 0, RES ? (int)((size_t)p.M * p.Cout * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(p.y), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);
    const int arow = wm * 64 + li, brow = wn * 32 + li, sw = (li >> 1) & 7;
    i32x2 rpre[EP];
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    int tile = first, c = 0;
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const char* a = ring + (t % NSLOT) * SLOT + arow * ROWB;
            const char* b = ring + (t % NSLOT) * SLOT + BM * ROWB + brow * ROWB;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int slot = ((2 * st + lh) ^ sw) * 16;
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b + slot);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a + i * 32 * ROWB + slot);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc[i], 0, 0, 0);      // D[n][m]
                }
            }
        }
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        const int m0 = tm * BM + wm * 64 + li, n0 = tn * BN + wn * 32 + 4 * lh;
        if (RES && c == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m0 + i * 32;
                    rpre[i * 4 + g] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rrsrc, m < p.M ? (unsigned)(((size_t)m * p.Cout + n0 + 8 * g) * 2) : OOB_OFFSET_B, 0, 0));
                }
        }
        if (c == nk - 1) {
            if (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's residual pieces (and the previous tile's stores, long gone)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + 8 * g;
                const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int m = m0 + i * 32;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[i][4 * g + k] * sc[k] + sh[k];
                    if (RES) {
                        const bf16x4 r = __builtin_bit_cast(bf16x4, rpre[i * 4 + g]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += (float)r[k];
                    }
                    bf16x4 ob;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ob[k] = (__bf16)activate_b(v[k], p.act);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, ob), yrsrc,
                                                          m < p.M ? (unsigned)(((size_t)m * p.Cout + n) * 2) : OOB_OFFSET_B, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
            tile += G; c = 0;
        } else {
            ++c;
        }
    }
}

template <int NSLOT, int NLOAD>
static int launch_ring3_bf16(const ConvArgsBf16& a, hipStream_t s) {
    const size_t lds = (size_t)NSLOT * (128 + 128) * 128;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_gemm_ring3_bf16<true, NSLOT, NLOAD>, lds, "conv2d_bf16")) return e;
    static std::atomic<uint64_t> lds_seen2{0};
    if (int e = raise_lds_once(lds_seen2, (const void*)k_gemm_ring3_bf16<false, NSLOT, NLOAD>, lds, "conv2d_bf16")) return e;
    const int per_cu = lds <= 81920 ? 2 : 1;
    const int ntiles = ((a.M + 127) / 128) * (a.Cout / 128);
    const int grid = ntiles >= 256 * per_cu ? 256 * per_cu : ((ntiles + 7) / 8) * 8;
    if (a.residual) k_gemm_ring3_bf16<true, NSLOT, NLOAD><<<grid, 512 + 64 * NLOAD, lds, s>>>(a);
    else k_gemm_ring3_bf16<false, NSLOT, NLOAD><<<grid, 512 + 64 * NLOAD, lds, s>>>(a);
    return check_launch("conv2d_fwd_bf16 (ring)");
}

// Ring form, fourth version: STORE waves.  The eight compute waves request, multiply and finish their outputs in registers
// (swapped operands, 8-byte residual pieces), then hand the rounded bf16 tile to NSTORE extra waves through a 16 KB LDS
// half-tile (64 rows x 256 B, 8-byte slots XOR-swizzled by the row); only those waves issue global stores, as whole
// 256-byte row segments, and they never wait for a load: the compute waves' vmcnt queues hold loads only.
template <bool RES, int NSTORE>
__global__ void __launch_bounds__(512 + 64 * NSTORE) k_gemm_ring4_bf16(const ConvArgsBf16 p) {
    constexpr int BM = 128, BN = 128, BK = 64, ROWB = BK * 2, SLOT = (BM + BN) * ROWB, EP = 8, NSLOT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem_u[];
    char* ring = smem_u;                                                   // [2][A 128 rows | B 128 rows][128 B]
    char* stg = smem_u + NSLOT * SLOT;                                     // [64][32 slots of 8 B], slot s of row r at s ^ (r & 31)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nk = p.Kpad / BK, tiles_n = p.Cout / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (first >= ntiles) return;
    const int T = ((ntiles - first + G - 1) / G) * nk;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(p.y), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);

    if (wave >= 8) {
        // ---- store wave: 16-byte pieces of whole rows, 64 rows x 16 pieces per half tile
        const int st = tid - 512;
        constexpr int PER = 64 * 16 / (64 * NSTORE);                       // pieces per thread and half tile
        int tile = first, c = 0;
        for (int t = 0; t < T; ++t) {
            __builtin_amdgcn_s_barrier();
            if (c == nk - 1) {
                const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    __builtin_amdgcn_s_barrier();                          // this half is in LDS
                    i32x4 v[PER];
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int id = q * (64 * NSTORE) + st, row = id >> 4, pc = id & 15;
                        const int s0 = (2 * pc) ^ (row & 31);              // physical slot of the piece's first half (its partner is s0 ^ 1)
                        v[q] = *reinterpret_cast<const i32x4*>(stg + row * 256 + (s0 & ~1) * 8);
                        if (row & 1) v[q] = i32x4{v[q][2], v[q][3], v[q][0], v[q][1]};
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (half == 0) __builtin_amdgcn_s_barrier();           // the half tile may be overwritten
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int id = q * (64 * NSTORE) + st, row = id >> 4, pc = id & 15;
                        const int m = tm * BM + half * 64 + row;
                        __builtin_amdgcn_raw_buffer_store_b128(v[q], yrsrc, m < p.M ? (unsigned)(((size_t)m * p.Cout + tn * BN + pc * 8) * 2) : OOB_OFFSET_B, 0, 0);
                    }
                }
                tile += G; c = 0;
            } else {
                ++c;
            }
        }
        return;
    }

    // ---- compute waves (they also request the operand chunks)
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, (int)((size_t)p.M * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(RES ? p.residual : p.x), 0, RES ? (int)((size_t)p.M * p.Cout * 2) : 0, 0x00020000);
    const int drow = wave * 8 + (lane >> 3);
    const int dgran = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;
    int i_tile = first, i_c = 0;
    unsigned ia_off[2], ib_off[2];
    auto aim = [&](int L) {
        const int tm = L / tiles_n, tn = L - tm * tiles_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tm * BM + drow + i * 64, n = tn * BN + drow + i * 64;
            ia_off[i] = (L < ntiles && m < p.M) ? (unsigned)((size_t)m * p.Cin * 2 + dgran) : OOB_OFFSET_B;
            ib_off[i] = L < ntiles ? (unsigned)((size_t)n * p.Kpad * 2 + dgran) : OOB_OFFSET_B;
        }
    };
    auto request = [&](int t) {
        char* dst = ring + (t % NSLOT) * SLOT + wave * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(dst + i * 8192), 16, ia_off[i], i_c * ROWB, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + BM * ROWB + i * 8192), 16, ib_off[i], i_c * ROWB, 0, 0);
        }
#endif
        if (++i_c == nk) { i_c = 0; i_tile += G; aim(i_tile); }
    };
    aim(first);
    request(0);

    const int arow = wm * 64 + li, brow = wn * 32 + li, sw = (li >> 1) & 7;
    i32x2 rpre[EP];
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    int tile = first, c = 0;
    for (int t = 0; t < T; ++t) {
        // chunk t has landed; the residual pieces requested at a c == 0 step may still be in flight at c == 1
        if (RES && c == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(EP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < T) request(t + 1);
        {
            const char* a = ring + (t % NSLOT) * SLOT + arow * ROWB;
            const char* b = ring + (t % NSLOT) * SLOT + BM * ROWB + brow * ROWB;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int slot = ((2 * st + lh) ^ sw) * 16;
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b + slot);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a + i * 32 * ROWB + slot);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc[i], 0, 0, 0);      // D[n][m]
                }
            }
        }
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        if (RES && c == 0) {
            const int m0 = tm * BM + wm * 64 + li, n0 = tn * BN + wn * 32 + 4 * lh;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = m0 + i * 32;
                    rpre[i * 4 + g] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rrsrc, m < p.M ? (unsigned)(((size_t)m * p.Cout + n0 + 8 * g) * 2) : OOB_OFFSET_B, 0, 0));
                }
        }
        if (c == nk - 1) {
            // outputs finished in registers, rounded, handed over half a tile at a time (wm = 0 rows 0..63, then wm = 1)
            i32x2 ob[8];
            const int nl = wn * 32 + 4 * lh;                               // this lane's first channel inside the tile
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = tn * BN + nl + 8 * g;
                const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + n) : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4*>(p.shift + n) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[i][4 * g + k] * sc[k] + sh[k];
                    if (RES) {
                        const bf16x4 r = __builtin_bit_cast(bf16x4, rpre[i * 4 + g]);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] += (float)r[k];
                    }
                    bf16x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (__bf16)activate_b(v[k], p.act);
                    ob[i * 4 + g] = __builtin_bit_cast(i32x2, o);
                }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (wm == half) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int row = i * 32 + li, slot = (nl >> 2) + 2 * g;
                            *reinterpret_cast<i32x2*>(stg + row * 256 + ((slot ^ (row & 31)) * 8)) = ob[i * 4 + g];
                        }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();                              // half tile visible to the store waves
                if (half == 0) __builtin_amdgcn_s_barrier();               // ... and read by them
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
            tile += G; c = 0;
        } else {
            ++c;
        }
    }
}

template <int NSTORE>
static int launch_ring4_bf16(const ConvArgsBf16& a, hipStream_t s) {
    const size_t lds = (size_t)2 * (128 + 128) * 128 + 64 * 256;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_gemm_ring4_bf16<true, NSTORE>, lds, "conv2d_bf16")) return e;
    static std::atomic<uint64_t> lds_seen2{0};
    if (int e = raise_lds_once(lds_seen2, (const void*)k_gemm_ring4_bf16<false, NSTORE>, lds, "conv2d_bf16")) return e;
    const int ntiles = ((a.M + 127) / 128) * (a.Cout / 128);
    const int grid = ntiles >= 512 ? 512 : ((ntiles + 7) / 8) * 8;
    if (a.residual) k_gemm_ring4_bf16<true, NSTORE><<<grid, 512 + 64 * NSTORE, lds, s>>>(a);
    else k_gemm_ring4_bf16<false, NSTORE><<<grid, 512 + 64 * NSTORE, lds, s>>>(a);
    return check_launch("conv2d_fwd_bf16 (ring)");
}

// the ring form (k_gemm_ring_bf16): eligibility of a descriptor
static bool ring_ok_bf16(const frcnn_conv_desc* d, bool has_mask, int y_is_f32) {
    if (has_mask || y_is_f32 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad_top || d->pad_left) return false;
    if ((d->cin % 32) || d->cin < 128 || (d->cout % 128) || d->ho != d->h || d->wo != d->w) return false;
    return true;
}

template <int BK, int NSLOT>
static int launch_ring_bf16(const ConvArgsBf16& a, hipStream_t s) {
    const size_t lds = (size_t)NSLOT * (128 + 128) * BK * 2 + 32 * 128 * 4;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_gemm_ring_bf16<true, BK, NSLOT>, lds, "conv2d_bf16")) return e;
    static std::atomic<uint64_t> lds_seen2{0};
    if (int e = raise_lds_once(lds_seen2, (const void*)k_gemm_ring_bf16<false, BK, NSLOT>, lds, "conv2d_bf16")) return e;
    const int per_cu = lds <= 81920 ? 2 : 1;
    const int ntiles = ((a.M + 127) / 128) * (a.Cout / 128);
    const int grid = ntiles >= 256 * per_cu ? 256 * per_cu : ((ntiles + 7) / 8) * 8;
    if (a.residual) k_gemm_ring_bf16<true, BK, NSLOT><<<grid, 512, lds, s>>>(a);
    else k_gemm_ring_bf16<false, BK, NSLOT><<<grid, 512, lds, s>>>(a);
    return check_launch("conv2d_fwd_bf16 (ring)");
}


// Ring form, fifth version: the epilogue DEFERRED into the next tile's chunk steps.  store_probe.hip: the 481 MB of a 512 -> 2048
// output leave in 82-88 us by themselves (5.5-5.9 TB/s) and in 188-191 us with the residual read beside them (5.0 TB/s of
// copy traffic) -- and the tiled kernel's 458 us are its 260 us operand / multiply phase PLUS that: all workgroups reach their
// epilogue together, HBM saturates for 6.5 us, then sits idle for 10 (a convoy).  Here the memory traffic of tile j is
// spread over the steps of tile j + 1: at the tile boundary the accumulators move to a second register set, and each of the
// next four steps turns ONE 32-row strip through one of two LDS strips (written by the four waves that own it after the step's
// barrier, read as whole-row pieces by everybody after the next one), adds the residual pieces requested a tile ago and stores.
// Sixteen waves (4 x 4, 32x32 each: the second accumulator set costs 16 registers), one workgroup per CU, three operand slots.
__device__ __forceinline__ void wait_vm_upto8(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
    }
}

template <bool RES>
__global__ void __launch_bounds__(1024) k_gemm_ring5_bf16(const ConvArgsBf16 p) {
    constexpr int BM = 128, BN = 128, BK = 64, ROWB = BK * 2, SLOT = (BM + BN) * ROWB, NSLOT = 3;
    extern __shared__ __attribute__((aligned(16))) char smem_v[];
    char* ring = smem_v;                                                   // [3][A 128 rows | B 128 rows][128 B]
    float* stg = reinterpret_cast<float*>(smem_v + NSLOT * SLOT);          // [2][32][128] f32
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
    const int nk = p.Kpad / BK, tiles_n = p.Cout / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int first = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (first >= ntiles) return;
    const int T = ((ntiles - first + G - 1) / G) * nk;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, (int)((size_t)p.M * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(RES ? p.residual : p.x), 0, RES ? (int)((size_t)p.M * p.Cout * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(p.y), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);

    // ---- request side: wave w lands rows 8w .. 8w + 7 of A and of B (one wave instruction each)
    const int drow = wave * 8 + (lane >> 3);
    const int dgran = ((lane & 7) ^ ((drow >> 1) & 7)) * 16;
    int i_tile = first, i_c = 0;
    unsigned ia_off, ib_off;
    auto aim = [&](int L) {
        const int tm = L / tiles_n, tn = L - tm * tiles_n;
        const int m = tm * BM + drow, n = tn * BN + drow;
        ia_off = (L < ntiles && m < p.M) ? (unsigned)((size_t)m * p.Cin * 2 + dgran) : OOB_OFFSET_B;
        ib_off = L < ntiles ? (unsigned)((size_t)n * p.Kpad * 2 + dgran) : OOB_OFFSET_B;
    };
    int issued = 0, mark_a = 0, mark_b = 0;                                // vmcnt bookkeeping: instructions issued so far / right after the
    auto request = [&](int t) {                                            // requests of the two chunks in flight
        char* dst = ring + (t % NSLOT) * SLOT + wave * 1024;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)dst, 16, ia_off, i_c * ROWB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(dst + BM * ROWB), 16, ib_off, i_c * ROWB, 0, 0);
#endif
        issued += 2;
        if (++i_c == nk) { i_c = 0; i_tile += G; aim(i_tile); }
    };
    aim(first);
    request(0); mark_a = issued;
    request(1); mark_b = issued;

    // ---- compute side
    const int arow = wm * 32 + li, brow = wn * 32 + li, sw = (li >> 1) & 7;
    const int prow = tid >> 5, pcol = (tid & 31) * 4;                      // this thread's 8-byte piece of a 32-row strip
    i32x2 rpre[4] = {}, rprev[4] = {}, r_res = {0, 0};
    f32x16 acc, accp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.0f; accp[e] = 0.0f; }
    int tile = first, c = 0;
    int e_step = 99, ptile = 0;                                            // steps since the previous tile ended (1..4: its strips are written)
    bool read_pending = false; int r_strip = 0, r_tile = 0;
    for (int t = 0; t < T + 5; ++t) {
        if (t < T) wait_vm_upto8(issued - mark_a);                         // chunk t has landed (what was issued behind its request may fly on)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t < T) {
            mark_a = mark_b;
            if (t + 2 < T) request(t + 2);
            mark_b = issued;
            const char* a = ring + (t % NSLOT) * SLOT + arow * ROWB;
            const char* b = ring + (t % NSLOT) * SLOT + BM * ROWB + brow * ROWB;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int slot = ((2 * st + lh) ^ sw) * 16;
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a + slot);
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b + slot);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
            }
            if (RES && c == 0) {
                const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = tm * BM + q * 32 + prow;
                    rpre[q] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rrsrc, m < p.M ? (unsigned)(((size_t)m * p.Cout + tn * BN + pcol) * 2) : OOB_OFFSET_B, 0, 0));
                }
                issued += 4;
            }
        }
        // ---- the previous tile's epilogue, one strip per step
        if (read_pending) {
            const int tm = r_tile / tiles_n, tn = r_tile - tm * tiles_n;
            const int m = tm * BM + r_strip * 32 + prow;
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + (r_strip & 1) * (32 * BN) + prow * BN + pcol);
            float v[4] = {v4[0], v4[1], v4[2], v4[3]};
            if (RES) {
                const bf16x4 r = __builtin_bit_cast(bf16x4, r_res);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += (float)r[k];
            }
            bf16x4 ob;
#pragma unroll
            for (int k = 0; k < 4; ++k) ob[k] = (__bf16)activate_b(v[k], p.act);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, ob), yrsrc,
                                                  m < p.M ? (unsigned)(((size_t)m * p.Cout + tn * BN + pcol) * 2) : OOB_OFFSET_B, 0, 0);
            issued += 1;
            read_pending = false;
        }
        if (e_step >= 1 && e_step <= 4) {
            const int s4 = e_step - 1;
            if (wm == s4) {
                const int tm = ptile / tiles_n, tn = ptile - tm * tiles_n;
                const int ncl = wn * 32 + li;
                const float sc = p.scale ? p.scale[tn * BN + ncl] : 1.0f;
                const float sh = p.shift ? p.shift[tn * BN + ncl] : 0.0f;
                float* dst = stg + (s4 & 1) * (32 * BN) + (4 * lh) * BN + ncl;
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = accp[e] * sc + sh;
            }
            read_pending = true; r_strip = s4; r_tile = ptile; r_res = rprev[0];
            rprev[0] = rprev[1]; rprev[1] = rprev[2]; rprev[2] = rprev[3];
        }
        ++e_step;
        if (t < T) {
            if (c == nk - 1) {
                accp = acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) rprev[q] = rpre[q];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
                ptile = tile; e_step = 1;
                tile += G; c = 0;
            } else {
                ++c;
            }
        }
    }
}

static int launch_ring5_bf16(const ConvArgsBf16& a, hipStream_t s) {
    const size_t lds = (size_t)3 * (128 + 128) * 128 + 2 * 32 * 128 * 4;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_gemm_ring5_bf16<true>, lds, "conv2d_bf16")) return e;
    static std::atomic<uint64_t> lds_seen2{0};
    if (int e = raise_lds_once(lds_seen2, (const void*)k_gemm_ring5_bf16<false>, lds, "conv2d_bf16")) return e;
    const int ntiles = ((a.M + 127) / 128) * (a.Cout / 128);
    const int grid = ntiles >= 256 ? 256 : ((ntiles + 7) / 8) * 8;
    if (a.residual) k_gemm_ring5_bf16<true><<<grid, 1024, lds, s>>>(a);
    else k_gemm_ring5_bf16<false><<<grid, 1024, lds, s>>>(a);
    return check_launch("conv2d_fwd_bf16 (ring)");
}


}  // namespace frcnn

using namespace frcnn;
extern "C" const char* frcnn_last_error(void) { return frcnn::g_err; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill_b(__bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (__bf16)(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
    }
}
static __bf16* dev_rand(size_t n, unsigned seed, float scale) {
    __bf16* p; CK(hipMalloc(&p, n * 2));
    k_fill_b<<<1024, 256>>>(p, n, seed, scale);
    return p;
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: bf16_ring_lab rows cin cout residual variant\n"); return 2; }
    const int rows = atoi(argv[1]), cin = atoi(argv[2]), cout = atoi(argv[3]), res = atoi(argv[4]);
    const std::string v = argv[5];
    frcnn_conv_desc d; memset(&d, 0, sizeof d);
    d.n = 1; d.h = rows; d.w = 1; d.cin = cin; d.cout = cout; d.kh = d.kw = 1; d.stride = 1; d.ho = rows; d.wo = 1; d.act = 1; d.tile = 0;
    if (!ring_ok_bf16(&d, false, 0) || cin % 64 || cin < 256) { fprintf(stderr, "shape not taken by the ring forms\n"); return 2; }
    __bf16* x = dev_rand((size_t)rows * cin, 1, 1.0f);
    __bf16* w = dev_rand((size_t)cout * cin, 2, 0.05f);
    __bf16* r = res ? dev_rand((size_t)rows * cout, 3, 1.0f) : nullptr;
    __bf16 *y0, *y1; CK(hipMalloc(&y0, (size_t)rows * cout * 2)); CK(hipMalloc(&y1, (size_t)rows * cout * 2));
    hipStream_t s; CK(hipStreamCreate(&s));
    ConvArgsBf16 a; memset(&a, 0, sizeof a);
    a.x = x; a.w = w; a.residual = r; a.y = y1; a.n_img = 1; a.H = rows; a.W = 1; a.Cin = cin; a.Cout = cout; a.R = a.S = 1; a.stride = 1;
    a.Ho = rows; a.Wo = 1; a.M = rows; a.Kpad = cin; a.act = 1; a.splits = 1; a.pix_stride = cin; a.img_stride = rows * cin; a.inv_S = 65536;
    auto ring = [&]() -> int {
        if (v == "v1_32_4") return launch_ring_bf16<32, 4>(a, s);
        if (v == "v1_64_3") return launch_ring_bf16<64, 3>(a, s);
        if (v == "v1_64_4") return launch_ring_bf16<64, 4>(a, s);
        if (v == "v2_2") return launch_ring2_bf16<2, 0>(a, s);
        if (v == "v2_3") return launch_ring2_bf16<3, 0>(a, s);
        if (v == "v2_4") return launch_ring2_bf16<4, 0>(a, s);
        if (v == "v2_2nt") return launch_ring2_bf16<2, 2>(a, s);
        if (v == "v3_2_1") return launch_ring3_bf16<2, 1>(a, s);
        if (v == "v3_2_2") return launch_ring3_bf16<2, 2>(a, s);
        if (v == "v3_4_2") return launch_ring3_bf16<4, 2>(a, s);
        if (v == "v4_2") return launch_ring4_bf16<2>(a, s);
        if (v == "v4_4") return launch_ring4_bf16<4>(a, s);
        if (v == "v5") return launch_ring5_bf16(a, s);
        fprintf(stderr, "unknown variant\n"); exit(2);
    };
    auto tiled = [&]() { if (int e = frcnn_conv2d_fwd_bf16(&d, x, w, nullptr, nullptr, r, y0, 0, s)) { fprintf(stderr, "tiled: %d %s\n", e, frcnn_last_error()); exit(1); } };
    auto timed = [&](auto&& f) {
        for (int i = 0; i < 3; ++i) f();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) f(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        return best * 100.0f;
    };
    const float t0 = timed(tiled);
    const float t1 = timed([&]() { if (int e = ring()) { fprintf(stderr, "ring: %d %s\n", e, frcnn_last_error()); exit(1); } });
    std::vector<unsigned short> h0((size_t)rows * cout), h1((size_t)rows * cout);
    CK(hipMemcpy(h0.data(), y0, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, h1.size() * 2, hipMemcpyDeviceToHost));
    size_t diff = 0; for (size_t i = 0; i < h0.size(); ++i) diff += h0[i] != h1[i];
    const double fl = 2.0 * rows * cin * cout;
    printf("rows %d %d -> %d res %d: tiled %.1f us (%.0f TFLOP/s), %s %.1f us (%.0f TFLOP/s), %zu of %zu outputs differ\n",
           rows, cin, cout, res, t0, fl / t0 / 1e6, v.c_str(), t1, fl / t1 / 1e6, diff, h0.size());
    return diff != 0;
}
