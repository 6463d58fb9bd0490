// bf16 convolution path (BASELINE configs[3]: ResNet-101, 600x1500, "bf16 conv + fp32 NMS") for gfx950.
//
// Same implicit-GEMM structure as conv_igemm.hip's v2 kernel -- NHWC activations, filters packed
// [Cout][Kpad] with k = [channel chunk][tap][channel], SRD buffer loads with out-of-range -> 0 for the
// halo and the tile tails, two-deep global->register->LDS operand pipeline, XCD-aware tile map, fused
// scale/shift/residual/activation epilogue -- with bf16 operands on v_mfma_f32_32x32x16_bf16 (f32
// accumulate, 16x the f32-input MFMA rate):
//   * a k-chunk is 64 channels = 128 B per row, so the 16-byte staging pattern (8 lanes per row, 32 rows
//     per pass) and the 144-byte padded LDS rows are IDENTICAL to the f32 kernel;
//   * lane (i, h) feeds MFMA k-step s with the 8 bf16 at byte offset 32*s + 16*h of row i
//     (A[i][8h+j], B[8h+j][i], cdna guide s3) -- one ds_read_b128 per operand tile per MFMA;
//   * activations stay bf16 in HBM between layers (half the bytes); the epilogue rounds once (RNE,
//     v_cvt_pk_bf16_f32) after the f32 scale/shift/residual/activation.
// At these shapes the kernel is operand-bandwidth bound, not MFMA bound: a 128x128x64 step needs
// 32 KB of operands for 16 MFMAs (512 cycles) per wave.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgsBf16 {
    const __bf16* x; const __bf16* w; const float* scale; const float* shift; const __bf16* residual; void* y;
    const __bf16* mask;     // optional [M][Cout]: output zeroed where mask <= 0 (ReLU backward fused into the input-gradient conv)
    int n_img, H, W, Cin, Cout, R, S, stride, pad_top, pad_left, Ho, Wo;
    int M, Kpad, act, out_f32;
    int tiles_m, tiles_n;
    int layout, pix_stride, img_stride, inv_S;      // position-major layout + tap walk, as in conv_igemm.hip's ConvArgs
    int splits;             // split-K (see conv_igemm.hip): K-slices per tile, f32 partial slabs, arrival tickets
    float* slabs;
    unsigned* tickets;
    int res_pre;            // 1: the kernel may request its residual pieces BEFORE the main loop (16-byte addressable, tensor under 2 GiB)
    int epi_rows;           // 1: the workgroup-wide "row pieces" epilogue (cout % 8 == 0); 0: the per-wave patches (dev knob / fallback)
};

constexpr int BKH = 64;                 // channels per k-chunk (128 B)
constexpr int LDS_STRIDE_B = 144;       // bytes per LDS row
constexpr unsigned OOB_OFFSET_B = 0x80000000u;

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
// Lab builds (scripts/micro/bf16_stamps.hip) compile this file with FRCNN_LAB_STAMPS: a workgroup of the direct-to-LDS forms then
// records the 100 MHz wall clock at its phase boundaries.  The product library never defines it.
#ifdef FRCNN_LAB_STAMPS
__device__ unsigned long long* g_lab_stamps_b = nullptr;
#define LAB_STAMP_B(i) do { if (g_lab_stamps_b && threadIdx.x == 0) g_lab_stamps_b[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LAB_STAMP_B(i) do { } while (0)
#endif
constexpr int SG_VALU = 0x2, SG_MFMA = 0x8, SG_VMEM_RD = 0x20, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

__device__ __forceinline__ int xcd_remap_b(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ float activate_b(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return 1.0f / (1.0f + __expf(-v));
    return v;
}

// WM x WN waves, each owning TM x TN 32x32 tiles.  2x2 waves is the base shape.  2x4 waves (512 threads, TM=2,
// TN=1) put FOUR waves on a SIMD with two workgroups per CU: a bf16 chunk is only 8-16 MFMAs (256-512 cycles) per
// wave, far less than the global->LDS->fragment latency of the next chunk, so the MFMA pipe needs more waves to
// draw from than the f32 kernel (whose chunk is 4096 cycles of MFMA) does.
// MASKED: the epilogue also applies the ReLU-backward mask (training's input-gradient pass).  A template parameter,
// not a runtime test: with the test in, the inference instantiations ran 14 % slower (596 -> 514 img/s on configs[3]).
template <int TM, int TN, bool SPLITK = false, int WM = 2, int WN = 2, bool MASKED = false, int VARIANT = 0>
__global__ void __launch_bounds__(64 * WM * WN) k_conv_igemm_bf16(const ConvArgsBf16 p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int RPP = NT / 8;                          // tile rows staged per pass (8 lanes x 16 B per row)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    static_assert(VARIANT < 3 || !SPLITK, "the direct-to-LDS forms take whole-K tiles");
    constexpr int LSTR = VARIANT >= 3 ? 128 : LDS_STRIDE_B;   // VARIANTs 3 / 4: unpadded rows, XOR-swizzled 16-byte slots
    constexpr int RING = VARIANT == 4 ? 4 : 2;                 // LDS operand buffers (VARIANT 4: a ring of four, three chunks in flight)
    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    char* As = smem_b;                                   // [RING][BM][LSTR]
    char* Bs = smem_b + RING * BM * LSTR;                // [RING][BN][LSTR]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    LAB_STAMP_B(0);

    const int splits = SPLITK ? p.splits : 1;
    const int nwg = p.tiles_m * p.tiles_n * splits;
    const int logical = xcd_remap_b(blockIdx.x, nwg);
    const int tile = SPLITK ? logical / splits : logical;
    const int slice = SPLITK ? logical - tile * splits : 0;
    // VARIANT 3 (256-wide column tiles, a handful per row tile): the column tiles of ONE row tile are adjacent, so an XCD's
    // resident workgroups share their A rows in L2 and the whole filter (<= 4 MB on the head) stays there; the other forms
    // walk row tiles first (their column tile's filter slice is what the neighbours share)
    const int tile_n = VARIANT >= 3 ? tile % p.tiles_n : tile / p.tiles_m;
    const int tile_m = VARIANT >= 3 ? tile / p.tiles_n : tile - tile_n * p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000);

    // byte column inside the 128-B chunk row this thread stages.  VARIANT 3: the LDS slot is the lane's position in the
    // wave instruction (tid & 7); slot g' of row r holds granule g' ^ ((r >> 1) & 7), so THAT is what the lane fetches
    // (rows advance by 64 per pass: the XOR term is the same for all of a thread's rows)
    const int lrow = tid >> 3, lcolb = VARIANT >= 3 ? (((tid & 7) ^ ((lrow >> 1) & 7)) * 16) : (tid & 7) * 16;
    int a_h[PA], a_w[PA], a_off[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + lrow + RPP * i;
        if (m < p.M) {
            int wo, ho, img;
            if (p.layout) { img = m % p.n_img; const int pos = m / p.n_img; ho = pos / p.Wo; wo = pos - ho * p.Wo; }
            else { wo = m % p.Wo; const int t = m / p.Wo; ho = t % p.Ho; img = t / p.Ho; }
            a_h[i] = ho * p.stride - p.pad_top;
            a_w[i] = wo * p.stride - p.pad_left;
            a_off[i] = (img * p.img_stride + (a_h[i] * p.W + a_w[i]) * p.pix_stride) * 2 + lcolb;
        } else {
            a_h[i] = -(1 << 28); a_w[i] = 0; a_off[i] = 0;
        }
    }
    unsigned b_off[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + lrow + RPP * i;
        b_off[i] = n < p.Cout ? (unsigned)(n * p.Kpad * 2 + lcolb) : OOB_OFFSET_B;
    }

    // taps this tile needs (position-major rows: taps that only meet zero padding are skipped; conv_igemm.hip)
    const int RS = p.R * p.S;
    const unsigned all_taps = RS >= 32 ? 0xffffffffu : (1u << RS) - 1u;
    unsigned tap_mask = all_taps;
    if (p.layout) {
        const int pos_lo = m0 / p.n_img, pos_hi = (min(m0 + BM, p.M) - 1) / p.n_img;
        if (pos_hi - pos_lo < 8) {
            unsigned mk = 0;
            for (int pos = pos_lo; pos <= pos_hi; ++pos) {
                const int ho = pos / p.Wo, wo = pos - ho * p.Wo;
                const int h0 = ho * p.stride - p.pad_top, w0 = wo * p.stride - p.pad_left;
                for (int r = 0; r < p.R; ++r)
                    for (int sx = 0; sx < p.S; ++sx)
                        if ((unsigned)(h0 + r) < (unsigned)p.H && (unsigned)(w0 + sx) < (unsigned)p.W) mk |= 1u << (r * p.S + sx);
            }
            if (mk) tap_mask = mk;
        }
    }
    const int n_taps = __popc(tap_mask);
    const int nk_all = (p.Kpad / (BKH * RS)) * n_taps;
    const int kb = SPLITK ? (int)((long long)slice * nk_all / splits) : 0;
    const int ke = SPLITK ? (int)((long long)(slice + 1) * nk_all / splits) : nk_all;

    unsigned rem = tap_mask;
    int c0 = 0, w_grp = 0;
    if (SPLITK) {
        const int grp = kb / n_taps;
        c0 = grp * BKH; w_grp = grp * RS * (BKH * 2);
        for (int t = kb - grp * n_taps; t > 0; --t) rem &= rem - 1;
    }
    // the NEXT chunk of this workgroup's sequence -> a set of staging registers (calls past the end of the range fetch
    // in-bounds or zero data that is never multiplied)
    auto load_into = [&](i32x4 (&xa)[PA], i32x4 (&xb)[PB]) {
        const int tap = __builtin_ctz(rem);
        const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
        const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 2;
        const int w_off = w_grp + tap * (BKH * 2);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            xb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i], w_off, 0);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            xa[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET_B, 0, 0);
        }
        rem &= rem - 1;
        const int wrap = (rem == 0);
        rem |= wrap ? tap_mask : 0u;
        c0 += wrap * BKH;
        w_grp += wrap * (RS * BKH * 2);
    };
    auto store_from = [&](const i32x4 (&xa)[PA], const i32x4 (&xb)[PB], int buf) {
        char* a = As + buf * BM * LDS_STRIDE_B;
        char* b = Bs + buf * BN * LDS_STRIDE_B;
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<i32x4*>(a + (lrow + RPP * i) * LDS_STRIDE_B + lcolb) = xa[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<i32x4*>(b + (lrow + RPP * i) * LDS_STRIDE_B + lcolb) = xb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // Round 3: the residual pieces this lane adds in the epilogue are requested HERE, before the main loop (the f32 kernel's
    // EPI_PRE).  The short-k, wide-output layers -- every block's 2c: 256 -> 1024 over 28 576 rows of a batch of eight
    // (4 chunks), 512 -> 2048 over 117 600 RoI rows (8 chunks) -- moved their bytes at 2.2-3.7 TB/s: the main loop is over in
    // 1-2 us, and only then did each half tile issue its 16-byte residual load and wait for it, four round trips in a row per
    // wave.  One or two 32x32 tiles per wave: 2 or 4 pieces of 16 bytes per lane (8-16 registers; the 128x128 eight-wave
    // tile has 69 of its 128 in use).  Same values added in the same order: bit-identical.
    // Round 3, epilogue through ONE workgroup-wide LDS tile ("row pieces"): the per-wave patch above hands every lane 8
    // channels of a 32-column tile, so a wave's store instruction covers sixteen 64-BYTE row segments -- half cache lines,
    // the other half written by the neighbouring wave some time later.  With the whole BM x BN f32 tile in LDS (it fits in
    // the dead operand buffers of every tile up to 128x128) thread t owns 16-byte pieces of WHOLE rows: piece q is row
    // q * RPPE + t / PPR, channels 8 (t % PPR) .., a wave instruction writes four full 256-byte row segments.  Same
    // arithmetic per element in the same order as the per-wave form: bit-identical (tests compare the two).
    constexpr int PPR = BN / 8, RPPE = NT / PPR, EP = BM / RPPE;
    constexpr bool EPI_ROWS = (size_t)BM * BN * 4 <= (size_t)RING * (BM + BN) * LSTR && NT % PPR == 0 && BM % RPPE == 0;
    // The residual pieces this lane adds in the epilogue are requested HERE, before the main loop (the f32 kernel's EPI_PRE).
    // The short-k, wide-output layers -- every block's 2c: 256 -> 1024 over 28 576 rows of a batch of eight (4 chunks),
    // 512 -> 2048 over 117 600 RoI rows (8 chunks) -- have a main loop of 1-2 us, after which each piece issued its 16-byte
    // residual load and waited for it.  2-4 pieces of 16 bytes per lane (8-16 registers; the 128x128 eight-wave direct-to-LDS
    // tile has 69 of its 128 in use, the register-staged one 118: no room there).  Same values, same order: bit-identical.
    constexpr bool RES_PRE = !SPLITK && !MASKED && EPI_ROWS && (TM * TN == 1 || (VARIANT >= 3 && TM * TN == 2));
    i32x4 rpre[RES_PRE ? EP : 1];
    bool pre = false;
    if constexpr (RES_PRE) {
        pre = p.res_pre && p.residual != nullptr && p.epi_rows;
        if (pre) {
            const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<__bf16*>(p.residual), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);
#pragma unroll
            for (int q = 0; q < EP; ++q) {
                const int m = m0 + q * RPPE + tid / PPR, n = n0 + (tid % PPR) * 8;
                const unsigned off = (m < p.M && n < p.Cout) ? (unsigned)(((size_t)m * p.Cout + n) * 2) : OOB_OFFSET_B;
                rpre[q] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0);
            }
        }
    }

    constexpr int MF = TM * TN;            // MFMAs per k-step (k = 16)
    constexpr int NL = PA + PB;
    constexpr int NF = TM + TN;
    if constexpr (VARIANT >= 3) {
        // Direct global -> LDS staging (buffer_load ... lds) for the 256-wide tiles of the detector head (round 2).
        // The 128x128 tile reads 1.5 fragments per MFMA and moves 32 KB of operands per 512 CU-cycles of matrix work; a
        // 128x64 (256x256 tile) or 64x64 (128x256) patch per wave reads 0.75 / 1.0 and halves the operand traffic.  What
        // made that tile lose in round 1 was the staging: at one workgroup per CU nothing hides the VGPR round trip.
        // Here a wave instruction lands 8 rows x 128 B straight in LDS (lane-linear destination: rows are unpadded and
        // the bank-conflict swizzle is applied on the SOURCE granule and again on the fragment read), chunk T+1 is
        // requested at the top of chunk T into the buffer everybody finished reading before the last barrier, and one
        // raw s_barrier per chunk follows `s_waitcnt vmcnt(0) lgkmcnt(0)`: this wave's share of T+1 has landed and its
        // reads of T are done.  Halo / tail rows ride on the buffer descriptor (zeros land in LDS).  Same k order as the
        // other loops.  Lab: scripts/micro/bf16_lab.hip (512->512 3x3 as a GEMM 995 TFLOP/s, 1024->2048 962).
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        // prep(): the offsets of the NEXT chunk of this workgroup's sequence (tap, channel block, halo test) into
        // registers; issue(buf): its eight-or-so wave instructions.  prep() for chunk T+2 runs behind the MFMAs of chunk T,
        // so that at the top of chunk T+1 the requests leave at once: they have one chunk (1-2 k cycles) to land, and in
        // the first version the ~300 cycles of address arithmetic in front of them showed up as barrier time
        // (512->2048 over 14 700 rows, main loop only: 33 us against 27 in the lab harness).
        unsigned a_v[PA];
        int w_v;
        auto prep = [&]() {
            const int tap = __builtin_ctz(rem);
            const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
            const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 2;
            w_v = w_grp + tap * (BKH * 2);
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                a_v[i] = ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET_B;
            }
            rem &= rem - 1;
            const int wrap = (rem == 0);
            rem |= wrap ? tap_mask : 0u;
            c0 += wrap * BKH;
            w_grp += wrap * (RS * BKH * 2);
        };
        auto issue = [&](int buf) {
            char* a = As + buf * BM * LSTR + wave * 8 * LSTR;
            char* b = Bs + buf * BN * LSTR + wave * 8 * LSTR;
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only: the host pass silently DROPS an instantiation whose body casts to an LDS pointer in dependent code (undefined kernel stub at load time)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(b + i * RPP * LSTR), 16, b_off[i], w_v, 0, 0);
#pragma unroll
            for (int i = 0; i < PA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(a + i * RPP * LSTR), 16, a_v[i], 0, 0, 0);
#endif
        };
        const int arow = wm * TM * 32 + li, brow = wn * TN * 32 + li;
        const int sw = (li >> 1) & 7;                             // (row >> 1) & 7 of every fragment row of this lane (tiles are 32 rows apart)
        auto multiply = [&](int buf) {
            const char* a = As + buf * BM * LSTR + arow * LSTR;
            const char* b = Bs + buf * BN * LSTR + brow * LSTR;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int slot = ((2 * st + lh) ^ sw) * 16;
                bf16x8 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * LSTR + slot);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * LSTR + slot);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        };
        if constexpr (VARIANT == 4) {
            // Round 3: a RING of four operand buffers, three chunks in flight.  On the short-k, wide-output layers of a batched
            // pass (512 -> 2048 over 117 600 RoI rows: 8 chunks; 256 -> 1024 over 28 576: 4) a chunk is 256 MFMA cycles per
            // wave, while its operands take 1-2 us to arrive (the first column tile of a row tile misses the L2): with ONE
            // chunk of lead and two workgroups per CU the matrix pipe idled 78 % of the time (553 TFLOP/s, 2.4 TB/s).  Here
            // chunk t + 3 is requested as chunk t starts; `s_waitcnt vmcnt(n)` waits for the OLDEST chunk only (a chunk is NL
            // wave instructions per thread, requests complete in order), then one barrier makes it visible to all waves and
            // frees the buffer chunk t - 1 was read from.  128 KB of LDS: one workgroup per CU.  Same k order: bit-identical.
            const int nk = ke - kb;
            prep(); issue(0);
            prep(); if (nk > 1) issue(1);
            prep(); if (nk > 2) issue(2);
            prep();                                               // chunk kb + 3
            for (int t = 0; t < nk; ++t) {
                const int later = min(2, nk - 1 - t);             // chunks requested behind chunk t (wave-uniform)
                if (later == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * NL) : "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (t + 3 < nk) issue((t + 3) & 3);               // into the buffer of chunk t - 1: every wave is past its reads
                multiply(t & 3);
                prep();                                           // chunk t + 4, behind this chunk's MFMAs
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // the epilogue reuses the buffers
        } else {
        prep();
        issue(0);
        prep();                                                   // chunk kb + 1 (never issued if the range has one chunk)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        LAB_STAMP_B(1);
        for (int kc = kb; kc < ke; ++kc) {
            const int buf = (kc - kb) & 1;
            if (kc + 1 < ke) issue(buf ^ 1);
            multiply(buf);
            prep();                                               // chunk kc + 2, behind this chunk's MFMAs
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        }
    } else if constexpr (VARIANT == 2) {
        // The mid-chunk-barrier schedule of conv_igemm.hip's VARIANT 2.  A bf16 chunk is only 4 x MF MFMAs of 32 cycles
        // per wave (128 cycles on the 64x64 tile), far less than the [LDS stores -> barrier -> fragment reads] chain that
        // ended every chunk of the loop below; here k-steps 0,1 multiply fragments read during the previous chunk, the
        // barrier sits between them and k-steps 2,3, and the stores of chunk T+2, the reads of chunk T+1's step 0 and the
        // global loads of chunk T+4 (two staging register sets) ride behind it.  Same k order: bit-identical results.
        i32x4 sa0[PA], sb0[PB], sa1[PA], sb1[PB];
        bf16x8 na[TM], nb[TN];
        load_into(sa0, sb0);
        load_into(sa1, sb1);
        store_from(sa0, sb0, 0);
        load_into(sa0, sb0);
        store_from(sa1, sb1, 1);
        load_into(sa1, sb1);
        __syncthreads();
        {
            const char* a = As + (wm * TM * 32 + li) * LDS_STRIDE_B + lh * 16;
            const char* b = Bs + (wn * TN * 32 + li) * LDS_STRIDE_B + lh * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) na[i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * LDS_STRIDE_B);
#pragma unroll
            for (int j = 0; j < TN; ++j) nb[j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * LDS_STRIDE_B);
        }
        auto chunk = [&](auto parity, auto& xa, auto& xb) {
            constexpr int P = decltype(parity)::value;
            const char* a = As + P * BM * LDS_STRIDE_B + (wm * TM * 32 + li) * LDS_STRIDE_B + lh * 16;
            const char* b = Bs + P * BN * LDS_STRIDE_B + (wn * TN * 32 + li) * LDS_STRIDE_B + lh * 16;
            const char* an = As + (P ^ 1) * BM * LDS_STRIDE_B + (wm * TM * 32 + li) * LDS_STRIDE_B + lh * 16;
            const char* bn = Bs + (P ^ 1) * BN * LDS_STRIDE_B + (wn * TN * 32 + li) * LDS_STRIDE_B + lh * 16;
            bf16x8 fa[4][TM], fb[4][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = na[i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = nb[j];
#pragma unroll
            for (int st = 1; st < 4; ++st) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[st][i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * LDS_STRIDE_B + st * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[st][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * LDS_STRIDE_B + st * 32);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][i], fb[st][j], acc[i][j], 0, 0, 0);
            SGB(SG_DS_RD, NF);                                       // step-1 fragments first
#pragma unroll
            for (int q = 0; q < MF; ++q) { SGB(SG_MFMA, 1); SGB(SG_DS_RD, (2 * NF + MF - 1) / MF); }      // k-step 0: steps 2,3 fragments
#pragma unroll
            for (int q = 0; q < MF; ++q) SGB(SG_MFMA, 1);            // k-step 1
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            store_from(xa, xb, P);                                   // chunk T+2 -> the buffer every wave has just finished reading
            load_into(xa, xb);                                       // chunk T+4
#pragma unroll
            for (int i = 0; i < TM; ++i) na[i] = *reinterpret_cast<const bf16x8*>(an + i * 32 * LDS_STRIDE_B);
#pragma unroll
            for (int j = 0; j < TN; ++j) nb[j] = *reinterpret_cast<const bf16x8*>(bn + j * 32 * LDS_STRIDE_B);
#pragma unroll
            for (int st = 2; st < 4; ++st)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][i], fb[st][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < MF; ++q) { SGB(SG_MFMA, 1); SGB(SG_DS_WR, (NL + MF - 1) / MF); }             // k-step 2: LDS stores
            SGB(SG_DS_RD, NF);
#pragma unroll
            for (int q = 0; q < MF; ++q) { SGB(SG_MFMA, 1); SGB(SG_VALU, 6); SGB(SG_VMEM_RD, (NL + MF - 1) / MF); }   // k-step 3: global loads
        };
        int kc = kb;
        for (; kc + 1 < ke; kc += 2) {
            chunk(std::integral_constant<int, 0>{}, sa0, sb0);
            chunk(std::integral_constant<int, 1>{}, sa1, sb1);
        }
        if (kc < ke) chunk(std::integral_constant<int, 0>{}, sa0, sb0);
        __syncthreads();
    } else {
    i32x4 ra[PA], rb[PB];
    auto load_chunk = [&](int) { load_into(ra, rb); };
    auto store_chunk = [&](int buf) { store_from(ra, rb, buf); };
    load_chunk(kb);
    store_chunk(0);
    load_chunk(kb + 1 < ke ? kb + 1 : kb);
    __syncthreads();

    for (int kc = kb; kc < ke; ++kc) {
        const int buf = (kc - kb) & 1;
        store_chunk(buf ^ 1);
        load_chunk(kc + 2 < ke ? kc + 2 : ke - 1);
        const char* a = As + buf * BM * LDS_STRIDE_B + (wm * TM * 32 + li) * LDS_STRIDE_B + lh * 16;
        const char* b = Bs + buf * BN * LDS_STRIDE_B + (wn * TN * 32 + li) * LDS_STRIDE_B + lh * 16;
        bf16x8 fa[4][TM], fb[4][TN];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[s][i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * LDS_STRIDE_B + s * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[s][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * LDS_STRIDE_B + s * 32);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
        // interleave: fragments for step s+1 and the staging traffic ride behind step s's MFMAs
        SGB(SG_DS_RD, NF);
#pragma unroll
        for (int q = 0; q < MF; ++q) { SGB(SG_MFMA, 1); SGB(SG_DS_WR, (NL + MF - 1) / MF); if (q < NF) SGB(SG_DS_RD, 1); }
#pragma unroll
        for (int q = 0; q < MF; ++q) { SGB(SG_MFMA, 1); SGB(SG_VALU, 6); SGB(SG_VMEM_RD, (NL + MF - 1) / MF); if (q < NF) SGB(SG_DS_RD, 1); }
#pragma unroll
        for (int q = 0; q < MF; ++q) { SGB(SG_MFMA, 1); if (q < NF) SGB(SG_DS_RD, 1); }
#pragma unroll
        for (int q = 0; q < MF; ++q) SGB(SG_MFMA, 1);
        __syncthreads();
    }
    }

    if constexpr (SPLITK) {
        // write-through partial tile -> ticket -> the last arriver sums the slices in order (conv_igemm.hip)
        const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
            p.slabs, 0, (int)((size_t)p.tiles_m * p.tiles_n * splits * (BM * BN) * 4), 0x00020000);
        const unsigned slab_off = (unsigned)((tile * splits + slice) * (BM * BN) * 4 + tid * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), srsrc,
                                                           slab_off + ((i * TN + j) * 4 + q) * (NT * 16), 0, 16 /* sc1 */);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* last = reinterpret_cast<int*>(smem_b);
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(&p.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int is_last = (t == (unsigned)(splits - 1));
            if (is_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&p.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *last = is_last;
        }
        __syncthreads();
        const int is_last = *last;
        __syncthreads();                                     // the epilogue reuses this LDS word
        if (!is_last) return;
        const float4* base = reinterpret_cast<const float4*>(p.slabs + (size_t)tile * splits * (BM * BN));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        for (int sl = 0; sl < splits; ++sl) {
            const float4* sp = base + (size_t)sl * (BM * BN / 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = sp[((i * TN + j) * 4 + q) * NT + tid];
                        acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                    }
        }
    }

    // epilogue: C/D map col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    LAB_STAMP_B(2);
    if constexpr (EPI_ROWS) {
        if (p.epi_rows) {
            float* stg = reinterpret_cast<float*>(smem_b);               // [BM][BN] f32, unpadded (see the read order below)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ncl = wn * TN * 32 + j * 32 + li, nc = n0 + ncl;
                const float sc = (p.scale && nc < p.Cout) ? p.scale[nc] : 1.0f;
                const float sh = (p.shift && nc < p.Cout) ? p.shift[nc] : 0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float* dst = stg + (wm * TM * 32 + i * 32 + 4 * lh) * BN + ncl;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[i][j][e] * sc + sh;
                }
            }
            __syncthreads();
            LAB_STAMP_B(3);
            const int prow = tid / PPR, pcol = (tid % PPR) * 8;
            // a row is BN floats, unpadded: lanes c and c + 8 of a row would meet on the same banks, so the upper eight read
            // their second 16 bytes first
            const int swap = (tid % PPR) & 8 ? 4 : 0;
#pragma unroll
            for (int q = 0; q < EP; ++q) {
                const int row = q * RPPE + prow, m = m0 + row, n = n0 + pcol;
                const float* src = stg + row * BN + pcol;
                const f32x4 va = *reinterpret_cast<const f32x4*>(src + swap);
                const f32x4 vb = *reinterpret_cast<const f32x4*>(src + (swap ^ 4));
                const f32x4 v0 = swap ? vb : va, v1 = swap ? va : vb;
                if (m < p.M && n < p.Cout) {
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    const size_t o = (size_t)m * p.Cout + n;
                    if (p.residual) {
                        bf16x8 r;
                        if (RES_PRE && pre) r = __builtin_bit_cast(bf16x8, rpre[RES_PRE ? q : 0]);
                        else r = *reinterpret_cast<const bf16x8*>(p.residual + o);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += (float)r[k];
                    }
                    if constexpr (MASKED) {
                        const bf16x8 mk = *reinterpret_cast<const bf16x8*>(p.mask + o);
#pragma unroll
                        for (int k = 0; k < 8; ++k) if (!((float)mk[k] > 0.0f)) v[k] = 0.0f;
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = activate_b(v[k], p.act);
                    if (p.out_f32) {
                        float* y = reinterpret_cast<float*>(p.y) + o;
                        *reinterpret_cast<f32x4*>(y) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(y + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        bf16x8 ob;
#pragma unroll
                        for (int k = 0; k < 8; ++k) ob[k] = (__bf16)v[k];
                        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + o) = ob;
                    }
                }
            }
#ifdef FRCNN_LAB_STAMPS
            LAB_STAMP_B(4);                                      // stores issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LAB_STAMP_B(5);                                      // stores done
            if (g_lab_stamps_b && threadIdx.x == 0) {
                unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                g_lab_stamps_b[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
            }
#endif
            return;
        }
    }
    if ((p.Cout & 7) == 0) {
        // Vectorised form.  In the accumulator layout a lane owns ONE column, so storing from it means 2-byte stores
        // (and 2-byte residual / mask loads) -- 16 memory instructions per 32x32 tile, each moving 128 B.  Instead
        // every wave turns its tiles through a private 32 x 36-float LDS patch (the operand buffers are dead: the
        // main loop ended on a barrier): written column-per-lane, read back row-major, eight consecutive channels
        // per lane -> one 16-byte residual load, one 16-byte mask load and one 16-byte store per lane per half tile.
        float* stg = reinterpret_cast<float*>(smem_b) + wave * (32 * 36);
        const int r_row = lane >> 2, r_col = (lane & 3) * 8;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nt = n0 + wn * TN * 32 + j * 32;
            const int nc = nt + li;
            const float sc = (p.scale && nc < p.Cout) ? p.scale[nc] : 1.0f;
            const float sh = (p.shift && nc < p.Cout) ? p.shift[nc] : 0.0f;
            const int n = nt + r_col;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mt = m0 + wm * TM * 32 + i * 32;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    stg[((e & 3) + 8 * (e >> 2) + 4 * lh) * 36 + li] = acc[i][j][e] * sc + sh;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int row = half * 16 + r_row, m = mt + row;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + row * 36 + r_col);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + row * 36 + r_col + 4);
                    if (m < p.M && n < p.Cout) {
                        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        const size_t o = (size_t)m * p.Cout + n;
                        if (p.residual) {
                            const bf16x8 r = *reinterpret_cast<const bf16x8*>(p.residual + o);
#pragma unroll
                            for (int q = 0; q < 8; ++q) v[q] += (float)r[q];
                        }
                        if constexpr (MASKED) {
                            const bf16x8 mk = *reinterpret_cast<const bf16x8*>(p.mask + o);
#pragma unroll
                            for (int q = 0; q < 8; ++q) if (!((float)mk[q] > 0.0f)) v[q] = 0.0f;
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = activate_b(v[q], p.act);
                        if (p.out_f32) {
                            float* y = reinterpret_cast<float*>(p.y) + o;
                            *reinterpret_cast<f32x4*>(y) = f32x4{v[0], v[1], v[2], v[3]};
                            *reinterpret_cast<f32x4*>(y + 4) = f32x4{v[4], v[5], v[6], v[7]};
                        } else {
                            bf16x8 ob;
#pragma unroll
                            for (int q = 0; q < 8; ++q) ob[q] = (__bf16)v[q];
                            *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + o) = ob;
                        }
                    }
                }
            }
        }
#ifdef FRCNN_LAB_STAMPS
        LAB_STAMP_B(3); LAB_STAMP_B(4);                          // (per-wave patches: no workgroup-wide staging phase)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LAB_STAMP_B(5);
        if (g_lab_stamps_b && threadIdx.x == 0) {
            unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_lab_stamps_b[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
        }
#endif
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + li;
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] : 1.0f;
        const float sh = p.shift ? p.shift[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * TM * 32 + i * 32 + 4 * lh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                if (m < p.M) {
                    float v = acc[i][j][e] * sc + sh;
                    if (p.residual) v += (float)p.residual[(size_t)m * p.Cout + n];
                    if constexpr (MASKED) { if (!((float)p.mask[(size_t)m * p.Cout + n] > 0.0f)) v = 0.0f; }
                    v = activate_b(v, p.act);
                    if (p.out_f32) reinterpret_cast<float*>(p.y)[(size_t)m * p.Cout + n] = v;
                    else reinterpret_cast<__bf16*>(p.y)[(size_t)m * p.Cout + n] = (__bf16)v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Row-STRIP form for the short-k, wide-output 1x1 layers of a batched pass (round 3): every ResNet block's branch2c --
// 512 -> 2048 over 117 600 RoI rows, 256 -> 1024 over 28 576 map rows -- and the stride-1 shortcuts.
//
// PMC on the tiled kernel (scripts/pmc_bf16_one.sh, 512 -> 2048): matrix pipe 25 % busy, waves waiting 54 % of their cycles,
// 5.3 GB through the L2 for 1.1 GB of algorithmic bytes, at ~47 GB/s per CU -- the per-CU vector-memory path, not HBM
// (3.3 TB/s) and not the MFMAs, sets the time: a 128x128 tile pulls 32 KB of operands per 64-deep chunk and its A rows are
// pulled again by each of the 16 column tiles.  Here a workgroup owns a STRIP of BM rows and walks ALL column tiles itself:
// the strip's A operand (BM x K bf16 <= 64 KB) is brought into LDS ONCE and stays, only the filter streams (two 16 KB ring
// buffers, chunk t + 1 requested while chunk t multiplies -- across column-tile boundaries, so it also flies during a
// column tile's epilogue), the residual pieces of a column tile are requested when its first chunk starts, and the stores of
// tile j drain under the chunks of tile j + 1.  Operand bytes through the CU: K x (BM + N) instead of K x (BM + 128) x N / 128.
// 1x1, stride 1, no padding (a GEMM: Y[m] = X[m] . W^T in either row layout), bf16 output, cin % 64 == 0, BM x K x 2 <= 64 KB.
// Same k order, same epilogue arithmetic as the tiled forms: bit-identical.
template <int BM>
__global__ void __launch_bounds__(512) k_gemm_strip_bf16(const ConvArgsBf16 p) {
    constexpr int NT = 512, BN = 128, RPP = 64, LSTR = 128;
    constexpr int TM = BM / 64;                            // 32-row tiles per wave (2 x 4 waves: wm owns BM / 2 rows, wn 32 columns)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int PPR = BN / 8, RPPE = NT / PPR, EP = BM / RPPE;
    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    char* Al = smem_s;                                     // [nk][BM][128 B], the strip's whole A operand (<= 64 KB)
    char* Wl = smem_s + 65536;                             // [2][128][128 B] filter ring
    float* stg = reinterpret_cast<float*>(smem_s + 65536 + 2 * BN * LSTR);       // [BM][128] f32 epilogue tile
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM;
    const int nk = p.Kpad / BKH, tiles_n = (p.Cout + BN - 1) / BN, T = nk * tiles_n;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, (int)((size_t)p.M * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(p.residual ? p.residual : p.x), 0, p.residual ? (int)((size_t)p.M * p.Cout * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<__bf16*>(p.y), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);
    const int lrow = tid >> 3, lcolb = ((tid & 7) ^ ((lrow >> 1) & 7)) * 16;       // source granule of this lane's LDS slot (the XOR swizzle)
    unsigned a_off[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + lrow + RPP * i;
        a_off[i] = m < p.M ? (unsigned)((size_t)m * p.Cin * 2 + lcolb) : OOB_OFFSET_B;
    }
    auto issue_w = [&](int t) {                            // chunk t of the strip's (column tile, k chunk) sequence -> ring buffer t & 1
        const int j = t / nk, c = t - j * nk;
        char* b = Wl + (t & 1) * BN * LSTR + wave * 8 * LSTR;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int n = j * BN + lrow + RPP * i;
            const unsigned off = n < p.Cout ? (unsigned)((size_t)n * p.Kpad * 2 + lcolb) : OOB_OFFSET_B;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(b + i * RPP * LSTR), 16, off, c * (BKH * 2), 0, 0);
        }
#endif
    };
    const bool pre = p.residual != nullptr;
    i32x4 rpre[EP];
    auto issue_res = [&](int j) {
#pragma unroll
        for (int q = 0; q < EP; ++q) {
            const int m = m0 + q * RPPE + tid / PPR, n = j * BN + (tid % PPR) * 8;
            const unsigned off = (m < p.M && n < p.Cout) ? (unsigned)(((size_t)m * p.Cout + n) * 2) : OOB_OFFSET_B;
            rpre[q] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off, 0, 0);
        }
    };
    // ---- the strip's A operand, once
#if defined(__HIP_DEVICE_COMPILE__)
    for (int c = 0; c < nk; ++c) {
        char* a = Al + c * BM * LSTR + wave * 8 * LSTR;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(a + i * RPP * LSTR), 16, a_off[i], c * (BKH * 2), 0, 0);
    }
#endif
    issue_w(0);
    if (pre) issue_res(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int arow = wm * TM * 32 + li, brow = wn * 32 + li, sw = (li >> 1) & 7;
    const int prow = tid / PPR, pcol = (tid % PPR) * 8, swap = (tid % PPR) & 8 ? 4 : 0;
    int t = 0;
    for (int j = 0; j < tiles_n; ++j) {
        f32x16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
        for (int c = 0; c < nk; ++c, ++t) {
            if (t + 1 < T) issue_w(t + 1);                 // into the buffer chunk t - 1 was read from: everyone passed the last barrier
            if (c == 0 && j > 0 && pre) issue_res(j);
            const char* a = Al + c * BM * LSTR + arow * LSTR;
            const char* b = Wl + (t & 1) * BN * LSTR + brow * LSTR;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int slot = ((2 * st + lh) ^ sw) * 16;
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b + slot);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a + i * 32 * LSTR + slot);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
                }
            }
            const bool last = c == nk - 1;
            if (last) {
                // ---- epilogue of column tile j (the row-piece form of k_conv_igemm_bf16): scale / shift per column in the
                // accumulator layout -> f32 tile in LDS -> 16-byte pieces of whole rows + residual + activation -> bf16
                const int n0 = j * BN, ncl = wn * 32 + li, nc = n0 + ncl;
                const float sc = (p.scale && nc < p.Cout) ? p.scale[nc] : 1.0f;
                const float sh = (p.shift && nc < p.Cout) ? p.shift[nc] : 0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float* dst = stg + (wm * TM * 32 + i * 32 + 4 * lh) * BN + ncl;
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * BN] = acc[i][e] * sc + sh;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int q = 0; q < EP; ++q) {
                    const int row = q * RPPE + prow, m = m0 + row, n = n0 + pcol;
                    const float* src = stg + row * BN + pcol;
                    const f32x4 va = *reinterpret_cast<const f32x4*>(src + swap);
                    const f32x4 vb = *reinterpret_cast<const f32x4*>(src + (swap ^ 4));
                    const f32x4 v0 = swap ? vb : va, v1 = swap ? va : vb;
                    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if (pre) {
                        const bf16x8 r = __builtin_bit_cast(bf16x8, rpre[q]);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += (float)r[k];
                    }
                    bf16x8 ob;
#pragma unroll
                    for (int k = 0; k < 8; ++k) ob[k] = (__bf16)activate_b(v[k], p.act);
                    // branch-free: EVERY wave issues exactly EP store instructions per column tile (pieces outside the tensor
                    // ride on the buffer descriptor) -- the vmcnt arithmetic at the end of the iteration counts on it
                    const unsigned yoff = (m < p.M && n < p.Cout) ? (unsigned)(((size_t)m * p.Cout + n) * 2) : OOB_OFFSET_B;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, ob), yrsrc, yoff, 0, 0);
                }
            }
            // chunk t + 1 must have landed before the next iteration reads it; what this iteration issued BEHIND it may stay in
            // flight: the residual pieces of tile j (c == 0), the stores of tile j (last chunk) -- requests complete in order
            const int later = ((c == 0 && j > 0 && pre) ? EP : 0) + (last ? EP : 0);
            if (later == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (later == EP) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(EP) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * EP) : "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// f32 HWIO [R][S][Cin][Cout] -> bf16 packed [Cout][Kpad], k = ((c/64)*R*S + tap)*64 + c%64 (Cin % 64 == 0)
__global__ void k_pack_hwio_bf16(const float* w, int RS, int Cin, int Cout, int Kpad, __bf16* out) {
    const size_t total = (size_t)Cout * Kpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kpad), n = (int)(i / Kpad);
        const int j = k % BKH, kc = k / BKH, tap = kc % RS, cc = kc / RS;
        out[i] = (__bf16)w[((size_t)tap * Cin + cc * BKH + j) * Cout + n];
    }
}

// Post-optimiser-step refresh of the bf16 device forms of MANY trainable layers in one launch (the bf16 twin of
// conv_igemm.hip's k_refresh_packed): forward pack, input-gradient pack (transposed, flipped, scale folded) and the
// folded epilogue shift.  packed / packed_dgrad of frcnn_pack_job point at bf16 storage here.
constexpr int REFRESH_JOBS_B = 32;
struct RefreshTableB { frcnn_pack_job job[REFRESH_JOBS_B]; int first_block[REFRESH_JOBS_B + 1]; int n; };   // workgroups in proportion to job size (conv_igemm.hip)
static int refresh_blocks_b(const frcnn_pack_job& j) {
    const long long elems = (long long)j.kh * j.kw * j.cin * j.cout;
    long long g = (elems + 8191) / 8192;
    return (int)(g < 4 ? 4 : (g > 2048 ? 2048 : g));
}
__global__ void __launch_bounds__(256) k_refresh_packed_bf16(const RefreshTableB t) {
    int ji = 0;
    while (ji + 1 < t.n && (int)blockIdx.x >= t.first_block[ji + 1]) ++ji;
    const frcnn_pack_job& j = t.job[ji];
    const int bx = (int)blockIdx.x - t.first_block[ji], gsz = t.first_block[ji + 1] - t.first_block[ji];
    const int RS = j.kh * j.kw;
    const size_t stride = (size_t)gsz * blockDim.x, first = (size_t)bx * blockDim.x + threadIdx.x;
    if (j.packed) {
        // 64 x 64 (channel x cout) tiles through LDS: cout-contiguous reads, channel-contiguous 128-B writes
        __shared__ float tile[BKH][65];
        __bf16* out = reinterpret_cast<__bf16*>(j.packed);
        const int Kpad = RS * j.cin, nblk = (j.cout + 63) / 64, ntiles = RS * (j.cin / BKH) * nblk;
        const int lane = threadIdx.x & 63, jr = threadIdx.x >> 6;
        for (int tl = bx; tl < ntiles; tl += gsz) {
            const int nb = tl % nblk, kc = tl / nblk, tap = kc % RS, cc = kc / RS, n0 = nb * 64;
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) {
                const int c = jr + 4 * pp;
                tile[c][lane] = n0 + lane < j.cout ? j.w_hwio[((size_t)tap * j.cin + cc * BKH + c) * j.cout + n0 + lane] : 0.0f;
            }
            __syncthreads();
            const int c2 = threadIdx.x & 31, nr = threadIdx.x >> 5;      // thread -> (channel pair, row): 4-byte stores, 128 B per row
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const int n = nr + 8 * pp;
                if (n0 + n < j.cout) {
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    bf16x2 v = {(__bf16)tile[2 * c2][n], (__bf16)tile[2 * c2 + 1][n]};
                    *reinterpret_cast<bf16x2*>(out + (size_t)(n0 + n) * Kpad + kc * BKH + 2 * c2) = v;
                }
            }
            __syncthreads();
        }
    }
    if (j.packed_dgrad) {
        __bf16* out = reinterpret_cast<__bf16*>(j.packed_dgrad);
        const int Kpad = RS * j.cout;                          // rows = cin, k over (cout chunk of 64, tap', cout % 64)
        const size_t total2 = (size_t)j.cin * Kpad / 2;        // two consecutive cout per thread: 8-byte reads, 4-byte stores
        for (size_t i2 = first; i2 < total2; i2 += stride) {
            const size_t i = 2 * i2;
            const int k = (int)(i % Kpad), ci = (int)(i / Kpad);
            const int jj = k % BKH, kc = k / BKH, tap = kc % RS, co = (kc / RS) * BKH + jj;
            const int r = j.kh - 1 - tap / j.kw, sx = j.kw - 1 - tap % j.kw;
            const float2 wv = *reinterpret_cast<const float2*>(j.w_hwio + ((size_t)(r * j.kw + sx) * j.cin + ci) * j.cout + co);
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 v = {(__bf16)(wv.x * (j.scale ? j.scale[co] : 1.0f)), (__bf16)(wv.y * (j.scale ? j.scale[co + 1] : 1.0f))};
            *reinterpret_cast<bf16x2*>(out + i) = v;
        }
    }
    if (j.shift)
        for (size_t i = first; i < (size_t)j.cout; i += stride)
            j.shift[i] = (j.bias ? j.bias[i] : 0.0f) * (j.scale ? j.scale[i] : 1.0f) + (j.shift_const ? j.shift_const[i] : 0.0f);
}

__global__ void k_cast_bf16_f32(const __bf16* x, size_t n, float* y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = (float)x[i];
}

// ReLU backward on bf16 tensors: g *= (y > 0)
__global__ void k_relu_bwd_bf16(__bf16* g, const __bf16* y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (!((float)y[i] > 0.0f)) g[i] = (__bf16)0.0f;
}

// AveragePooling2D(k x k -> 1) backward fused with the ReLU in front of it: gx[n][q][c] = (y[n][q][c] > 0) * g[n][c] / hw
__global__ void k_avgpool_bwd_masked_bf16(const float* g_pooled, const __bf16* y, int hw, int C, size_t total, float inv, __bf16* gx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t img = i / ((size_t)hw * C);
        gx[i] = (float)y[i] > 0.0f ? (__bf16)(g_pooled[img * C + c] * inv) : (__bf16)0.0f;
    }
}

__global__ void k_cast_f32_bf16(const float4* x, size_t n4, __bf16* y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        __bf16* o = y + 4 * i;
        o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    }
}

// AveragePooling2D(k) over a k x k bf16 map -> f32 [n][C] (f32 accumulate)
__global__ void k_avgpool_bf16_f32(const __bf16* x, int n, int hw, int C, int pos_major, float* y) {
    const size_t total = (size_t)n * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const size_t img = i / C;
        float acc = 0.0f;
        if (pos_major) for (int q = 0; q < hw; ++q) acc += (float)x[(size_t)q * total + i];           // x[pos][img][c]
        else for (int q = 0; q < hw; ++q) acc += (float)x[(img * hw + q) * C + c];                    // x[img][pos][c]
        y[i] = acc / (float)hw;
    }
}

// The same, eight channels (16 bytes) per lane: the scalar form above moves 2 bytes per lane per load (1.85 TB/s over the
// 482 MB of a batch of eight's res5c output); per channel the additions are the same, in the same order: bit-identical.
__global__ void k_avgpool_bf16_f32_v8(const __bf16* x, int n, int hw, int C, int pos_major, float* y) {
    const size_t total8 = (size_t)n * C / 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i * 8;                               // first of this lane's eight (img, c) outputs: same img (C % 8 == 0)
        const int c = (int)(e % C);
        const size_t img = e / C;
        float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int q = 0;
        for (; q + 7 <= hw; q += 7) {                          // seven loads in flight (a 7x7 window: seven rounds), summed in position order
            bf16x8 v[7];
#pragma unroll
            for (int j = 0; j < 7; ++j)
                v[j] = *reinterpret_cast<const bf16x8*>(pos_major ? x + (size_t)(q + j) * n * C + e : x + (img * hw + q + j) * C + c);
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += (float)v[j][k];
        }
        for (; q < hw; ++q) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(pos_major ? x + (size_t)q * n * C + e : x + (img * hw + q) * C + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
        }
        *reinterpret_cast<f32x4*>(y + e) = f32x4{acc[0] / (float)hw, acc[1] / (float)hw, acc[2] / (float)hw, acc[3] / (float)hw};
        *reinterpret_cast<f32x4*>(y + e + 4) = f32x4{acc[4] / (float)hw, acc[5] / (float)hw, acc[6] / (float)hw, acc[7] / (float)hw};
    }
}

// RoiResizeConv on a bf16 feature map (custom_layers.py:35-56): f32 lerp, bf16 output
__global__ void __launch_bounds__(256) k_roi_fwd_bf16(const __bf16* feat, int rows, int cols, int C, const float4* rois, int pool,
                                                      const float* fill, int relu, int pos_major, __bf16* out) {
    const int pix = blockIdx.x;
    const int px = pix % pool, py = (pix / pool) % pool, r = pix / (pool * pool);
    const size_t orow = pos_major ? (size_t)(py * pool + px) * (gridDim.x / (pool * pool)) + r : (size_t)pix;
    const float4 roi = rois[r];
    const int x1 = (int)roi.x, y1 = (int)roi.y, x2 = (int)roi.z, y2 = (int)roi.w;
    const int h = y2 - y1, w = x2 - x1;
    __bf16* o = out + orow * C;
    const bool ok = h > 0 && w > 0 && x1 >= 0 && y1 >= 0 && x2 <= cols && y2 <= rows;
    if (!ok) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) { const float v = fill ? fill[c] : 0.0f; o[c] = (__bf16)(relu ? fmaxf(v, 0.0f) : v); }
        return;
    }
    const float sy = (float)h / (float)pool, sx = (float)w / (float)pool;
    const float fy = (float)py * sy, fx = (float)px * sx;
    const int ly = (int)fy, lx = (int)fx;
    const float ty = fy - (float)ly, tx = fx - (float)lx;
    const int ylo = y1 + ly, yhi = y1 + min(ly + 1, h - 1), xlo = x1 + lx, xhi = x1 + min(lx + 1, w - 1);
    const __bf16* tl = feat + ((size_t)ylo * cols + xlo) * C;
    const __bf16* tr = feat + ((size_t)ylo * cols + xhi) * C;
    const __bf16* bl = feat + ((size_t)yhi * cols + xlo) * C;
    const __bf16* br = feat + ((size_t)yhi * cols + xhi) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float a = (float)tl[c], b = (float)tr[c], d = (float)bl[c], e = (float)br[c];
        const float top = a + (b - a) * tx, bot = d + (e - d) * tx;
        const float v = top + (bot - top) * ty;
        o[c] = (__bf16)(relu ? fmaxf(v, 0.0f) : v);
    }
}

// The same for the RoIs of SEVERAL images in one launch (batched inference): RoI r belongs to image r / n_per_img, whose map
// starts at feat + image * rows * cols * C; the output is ONE tensor over all RoIs ([7][7][all RoIs][C] position-major, or
// [all RoIs][7][7][C]), so the detector head runs one GEMM pass over every image's RoIs.  Eight channels (16 bytes) per lane;
// per element the arithmetic is k_roi_fwd_bf16's (f32 lerp in the same order, one rounding): bit-identical per RoI.
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__global__ void __launch_bounds__(256) k_roi_fwd_bf16_batch(const __bf16* feat, int rows, int cols, int C, const float4* rois, int n_per_img, int pool,
                                                            const float* fill, int relu, int pos_major, __bf16* out) {
    const int pix = blockIdx.x;
    const int px = pix % pool, py = (pix / pool) % pool, r = pix / (pool * pool);
    const size_t orow = pos_major ? (size_t)(py * pool + px) * (gridDim.x / (pool * pool)) + r : (size_t)pix;
    const float4 roi = rois[r];
    const int x1 = (int)roi.x, y1 = (int)roi.y, x2 = (int)roi.z, y2 = (int)roi.w;
    const int h = y2 - y1, w = x2 - x1;
    __bf16* o = out + orow * C;
    const bool ok = h > 0 && w > 0 && x1 >= 0 && y1 >= 0 && x2 <= cols && y2 <= rows;
    if (!ok) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) { const float v = fill ? fill[c] : 0.0f; o[c] = (__bf16)(relu ? fmaxf(v, 0.0f) : v); }
        return;
    }
    const __bf16* f = feat + (size_t)(r / n_per_img) * rows * cols * C;
    const float sy = (float)h / (float)pool, sx = (float)w / (float)pool;
    const float fy = (float)py * sy, fx = (float)px * sx;
    const int ly = (int)fy, lx = (int)fx;
    const float ty = fy - (float)ly, tx = fx - (float)lx;
    const int ylo = y1 + ly, yhi = y1 + min(ly + 1, h - 1), xlo = x1 + lx, xhi = x1 + min(lx + 1, w - 1);
    const u16x8* tl = reinterpret_cast<const u16x8*>(f + ((size_t)ylo * cols + xlo) * C);
    const u16x8* tr = reinterpret_cast<const u16x8*>(f + ((size_t)ylo * cols + xhi) * C);
    const u16x8* bl = reinterpret_cast<const u16x8*>(f + ((size_t)yhi * cols + xlo) * C);
    const u16x8* br = reinterpret_cast<const u16x8*>(f + ((size_t)yhi * cols + xhi) * C);
    u16x8* o8 = reinterpret_cast<u16x8*>(o);
    for (int c = threadIdx.x; c < C / 8; c += blockDim.x) {
        const u16x8 va = tl[c], vb = tr[c], vd = bl[c], ve = br[c];
        u16x8 res;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float a = bf16_bits_to_f32(va[k]), b = bf16_bits_to_f32(vb[k]), d = bf16_bits_to_f32(vd[k]), e = bf16_bits_to_f32(ve[k]);
            const float top = a + (b - a) * tx, bot = d + (e - d) * tx;
            const float v = top + (bot - top) * ty;
            res[k] = __builtin_bit_cast(unsigned short, (__bf16)(relu ? fmaxf(v, 0.0f) : v));
        }
        o8[c] = res;
    }
}

static int g_bf16_variant = getenv("FRCNN_BF16_VARIANT") ? atoi(getenv("FRCNN_BF16_VARIANT")) : 2;   // dev knob: 0 = the round-1 loop

template <int TM, int TN, int WM, int WN, bool MASKED, int VARIANT>
static int launch_bf16_v(const ConvArgsBf16& a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    ConvArgsBf16 p = a;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const size_t lds = VARIANT == 4 ? (size_t)4 * (BM + BN) * 128 : (size_t)2 * (BM + BN) * (VARIANT == 3 ? 128 : LDS_STRIDE_B);
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_bf16<TM, TN, false, WM, WN, MASKED, VARIANT>, lds, "conv2d_bf16")) return e;
    k_conv_igemm_bf16<TM, TN, false, WM, WN, MASKED, VARIANT><<<p.tiles_m * p.tiles_n, 64 * WM * WN, lds, s>>>(p);
    return check_launch("conv2d_fwd_bf16");
}

template <int TM, int TN, int WM, int WN, bool MASKED>
static int launch_bf16_m(const ConvArgsBf16& a, hipStream_t s) {
    return g_bf16_variant == 2 ? launch_bf16_v<TM, TN, WM, WN, MASKED, 2>(a, s) : launch_bf16_v<TM, TN, WM, WN, MASKED, 0>(a, s);
}

template <int TM, int TN, int WM = 2, int WN = 2>
static int launch_bf16(const ConvArgsBf16& a, hipStream_t s) {
    return a.mask ? launch_bf16_m<TM, TN, WM, WN, true>(a, s) : launch_bf16_m<TM, TN, WM, WN, false>(a, s);
}

constexpr size_t SPLITK_TICKET_BYTES_B = 16384;     // same workspace layout as frcnn_conv2d_fwd_ws

static int launch_bf16_splitk(const ConvArgsBf16& a, hipStream_t s) {
    ConvArgsBf16 p = a;
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = (p.Cout + 63) / 64;
    const size_t lds = (size_t)2 * (64 + 64) * LDS_STRIDE_B;
    if (g_bf16_variant == 2) {
        if (p.mask) k_conv_igemm_bf16<1, 1, true, 2, 2, true, 2><<<p.tiles_m * p.tiles_n * p.splits, 256, lds, s>>>(p);
        else k_conv_igemm_bf16<1, 1, true, 2, 2, false, 2><<<p.tiles_m * p.tiles_n * p.splits, 256, lds, s>>>(p);
    } else {
        if (p.mask) k_conv_igemm_bf16<1, 1, true, 2, 2, true><<<p.tiles_m * p.tiles_n * p.splits, 256, lds, s>>>(p);
        else k_conv_igemm_bf16<1, 1, true><<<p.tiles_m * p.tiles_n * p.splits, 256, lds, s>>>(p);
    }
    return check_launch("conv2d_fwd_bf16 (split-K)");
}

// K-slices per tile for the 64x64 bf16 kernel: same policy as the f32 engine (conv_igemm.hip choose_splits)
// the row-strip GEMM form (k_gemm_strip_bf16): eligibility of a descriptor, and the strip height it takes
static int strip_rows_bf16(const frcnn_conv_desc* d, bool has_mask, int y_is_f32) {
    if (has_mask || y_is_f32 || d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad_top || d->pad_left) return 0;
    if ((d->cin % BKH) || (d->cout & 7) || d->ho != d->h || d->wo != d->w) return 0;
    const int K = d->cin;
    if (128 * K * 2 <= 65536) return 128;
    if (64 * K * 2 <= 65536) return 64;
    return 0;
}

template <int BM>
static int launch_strip_bf16(const ConvArgsBf16& a, hipStream_t s) {
    const size_t lds = 65536 + 2 * 128 * 128 + (size_t)BM * 128 * 4;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_gemm_strip_bf16<BM>, lds, "conv2d_bf16")) return e;
    k_gemm_strip_bf16<BM><<<(a.M + BM - 1) / BM, 512, lds, s>>>(a);
    return check_launch("conv2d_fwd_bf16 (row strips)");
}

static int choose_splits_bf16(const frcnn_conv_desc* d, int cfg) {
    if (cfg != 2 && cfg != 12) return 1;
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long tiles = ((M + 63) / 64) * ((d->cout + 63) / 64);
    const int nk = d->kh * d->kw * d->cin / BKH;
    if (tiles * sizeof(unsigned) > SPLITK_TICKET_BYTES_B) return 1;
    int s = d->tile / 100;
    if (s <= 0) {
        if (tiles >= 384 || nk < 8) return 1;
        s = tiles >= 100 ? 3 : (int)((456 + tiles - 1) / tiles);
        if (s > nk / 4) s = nk / 4;
        if (s > 16) s = 16;
    }
    if (s > nk) s = nk;
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

static int choose_config_bf16(const frcnn_conv_desc* d) {
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long t128 = ((M + 127) / 128) * ((d->cout + 127) / 128);
    int cfg = d->tile % 100;
    // measured on MI355X (scripts/conv_shapes.py --bf16): the 8-wave 128x128 tile (2x4 waves) beats the 4-wave one on
    // every shape with >= 1 tile per CU (+8 % head 3x3, +38 % on 512 -> 2048); thin outputs take its 128x64 sibling.
    // Alone on the chip a launch wants >= 1 big tile per CU, else 64x64 tiles (C4 one image in flight: 312 img/s at
    // a threshold of 256 tiles, 261 at 32); beside other images' launches (tile code 50) the big tiles' better
    // MFMA efficiency wins down to a handful of tiles (C4 four in flight: 596 img/s at 16, 556 at 256).
    if (cfg == 0 || cfg == 50) {
        // the 256-wide direct-to-LDS tiles (round 2; one workgroup per CU): 256x256 with at least one tile per CU,
        // else 128x256 with >= 200 tiles (the head's 512-wide layers over 300 RoIs: 230) -- lab table in DESIGN 7
        static const bool wide = getenv("FRCNN_BF16_WIDE") && atoi(getenv("FRCNN_BF16_WIDE")) != 0;   // dev knob (off: code 47 ties or beats them)
        const long long nt256 = (d->cout + 255) / 256;
        const long long t256 = ((M + 255) / 256) * nt256, t128x256 = ((M + 127) / 128) * nt256;
        if (wide && (d->cout % 256) == 0 && t256 >= 256) return 45;
        if (wide && (d->cout % 256) == 0 && t128x256 >= 200) return 46;
        // round 3, batched passes (M = 8 x 14 700 RoI rows / 8 x 3 572 map rows; bench.py --config c4 --conv-table): with
        // several ROUNDS of 256x256 tiles the wide form wins on the long-k layers -- 2048->512 311 -> 264 us, the head 3x3
        // 465 -> 440, rpn_conv1 281 -> 264 -- and loses on the short-k, wide-output ones (512->2048 446 -> 505 us, 256->1024
        // 36.8 -> 46.4: those are bound by their output / residual bytes, and the 128x128 tile requests its residual
        // pieces before the main loop), so: long k, two or more rounds of whole tiles
        if ((d->cout % 256) == 0 && t256 >= 512 && d->kh * d->kw * d->cin >= 1024) return 45;
        // round 4 (scripts/conv_shapes_c4.py): a very long k loop pays for ONE round of 256x256 tiles too -- rpn_conv1 over eight maps
        // (28 576 rows, k 9 216, 224 tiles) 225 us against 282 on 128x128 tiles
        if ((d->cout % 256) == 0 && t256 >= 192 && d->kh * d->kw * d->cin >= 8192) return 45;
        static const bool dma64 = getenv("FRCNN_BF16_DMA64") && atoi(getenv("FRCNN_BF16_DMA64")) != 0;   // dev knob
        static const int big = getenv("FRCNN_BF16_BIG") ? atoi(getenv("FRCNN_BF16_BIG")) : 47;            // dev knob: 42 = register-staged
        // 47 (direct-to-LDS 128x128) on long row ranges -- the detector head over 300 RoIs: 1024->2048 103 -> 81 us, 3x3
        // 83 -> 77 (901 TFLOP/s), 512->2048 69 -> 51 -- and the register-staged 42 on the training steps' 2-3 k rows
        // (measured equal to 1-2 % slower there, beside the weight-gradient stream)
        // 64-column layers on long row ranges (stage 2 of a batched pass, 445 808 rows): the direct-to-LDS 64x64 tile, five workgroups
        // per CU -- 29.6 / 66.4 / 57.3 us against 32.3 / 69.5 / 62.5 on the register-staged 128x64 one
        cfg = t128 >= (cfg == 50 ? 16 : 256) ? (d->cout >= 128 ? (M >= 8192 ? big : 42) : (M >= 65536 ? 48 : 43)) : (dma64 ? 48 : 2);
    }
    return cfg;
}

static inline int ew_grid_b(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace frcnn

using namespace frcnn;

extern "C" {

int frcnn_conv_packed_k_bf16(int kh, int kw, int cin) { return kh * kw * cin; }

int frcnn_pack_conv_weights_bf16(const float* w_hwio, int kh, int kw, int cin, int cout, void* packed_bf16, void* stream) {
    if (!w_hwio || !packed_bf16 || kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return fail(FRCNN_E_ARG, "pack_conv_weights_bf16: bad argument");
    if (cin % BKH) return fail(FRCNN_E_UNSUPPORTED, "pack_conv_weights_bf16: cin must be a multiple of 64");
    const size_t total = (size_t)cout * kh * kw * cin;
    k_pack_hwio_bf16<<<ew_grid_b(total), 256, 0, as_stream(stream)>>>(w_hwio, kh * kw, cin, cout, kh * kw * cin, (__bf16*)packed_bf16);
    return check_launch("pack_conv_weights_bf16");
}

int frcnn_conv2d_fwd_bf16(const frcnn_conv_desc* d, const void* x_bf16, const void* w_packed_bf16,
                          const float* scale, const float* shift, const void* residual_bf16, void* y, int y_is_f32, void* stream) {
    return frcnn_conv2d_fwd_bf16_ws(d, x_bf16, w_packed_bf16, scale, shift, residual_bf16, y, y_is_f32, nullptr, 0, stream);
}

size_t frcnn_conv2d_workspace_bytes_bf16(const frcnn_conv_desc* d) {
    if (!d || d->cin <= 0 || (d->cin % BKH) != 0) return 0;
    const int splits = choose_splits_bf16(d, choose_config_bf16(d));
    if (splits <= 1) return 0;
    const long long M = (long long)d->n * d->ho * d->wo;
    const size_t tiles = (size_t)((M + 63) / 64) * ((d->cout + 63) / 64);
    return SPLITK_TICKET_BYTES_B + tiles * splits * 64 * 64 * sizeof(float);
}

int frcnn_conv2d_fwd_bf16_ws(const frcnn_conv_desc* d, const void* x_bf16, const void* w_packed_bf16,
                             const float* scale, const float* shift, const void* residual_bf16, void* y, int y_is_f32,
                             void* workspace, size_t workspace_bytes, void* stream) {
    return frcnn_conv2d_fwd_bf16_masked(d, x_bf16, w_packed_bf16, scale, shift, residual_bf16, nullptr, y, y_is_f32, workspace, workspace_bytes, stream);
}

int frcnn_conv2d_fwd_bf16_masked(const frcnn_conv_desc* d, const void* x_bf16, const void* w_packed_bf16,
                                 const float* scale, const float* shift, const void* residual_bf16, const void* mask_bf16,
                                 void* y, int y_is_f32, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !x_bf16 || !w_packed_bf16 || !y) return fail(FRCNN_E_ARG, "conv2d_fwd_bf16: null pointer");
    if (d->cin % BKH) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_bf16: cin must be a multiple of 64");
    if ((size_t)d->n * d->h * d->w * d->cin * 2 >= 0x7fffffffull || (size_t)d->cout * d->kh * d->kw * d->cin * 2 >= 0x7fffffffull)
        return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_bf16: tensor exceeds the 2 GiB buffer-descriptor range");
    ConvArgsBf16 a;
    a.x = (const __bf16*)x_bf16; a.w = (const __bf16*)w_packed_bf16; a.scale = scale; a.shift = shift;
    a.residual = (const __bf16*)residual_bf16; a.y = y; a.mask = (const __bf16*)mask_bf16;
    a.n_img = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.R = d->kh; a.S = d->kw;
    a.stride = d->stride; a.pad_top = d->pad_top; a.pad_left = d->pad_left; a.Ho = d->ho; a.Wo = d->wo;
    a.M = d->n * d->ho * d->wo; a.Kpad = d->kh * d->kw * d->cin; a.act = d->act; a.out_f32 = y_is_f32;
    a.tiles_m = a.tiles_n = 0;
    a.splits = 1; a.slabs = nullptr; a.tickets = nullptr;
    static const bool no_pre = getenv("FRCNN_BF16_NO_RES_PRE") != nullptr;               // dev knob (A/B)
    a.res_pre = !no_pre && residual_bf16 && (d->cout & 7) == 0 && (reinterpret_cast<uintptr_t>(residual_bf16) & 15) == 0
             && (size_t)a.M * d->cout * 2 < 0x7fffffffull;
    static const bool wave_epi = getenv("FRCNN_BF16_WAVE_EPILOGUE") != nullptr;          // dev knob (A/B, bit-identity tests)
    a.epi_rows = !wave_epi && (d->cout & 7) == 0;
    a.layout = d->layout ? 1 : 0;
    a.pix_stride = a.layout ? d->n * d->cin : d->cin;
    a.img_stride = a.layout ? d->cin : d->h * d->w * d->cin;
    a.inv_S = (65536 + d->kw - 1) / d->kw;
    if (d->kh * d->kw > 32) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_bf16: at most 32 filter taps");
    hipStream_t s = as_stream(stream);
    const int cfg = choose_config_bf16(d);
    static std::atomic<uint64_t> lds_seen{0};
    if (workspace) {
        const size_t need = frcnn_conv2d_workspace_bytes_bf16(d);
        if (need) {
            if (workspace_bytes < need) return fail(FRCNN_E_WORKSPACE, "conv2d_fwd_bf16: workspace needs %zu bytes", need);
            if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_bf16<1, 1, true>, (size_t)2 * 128 * LDS_STRIDE_B, "conv2d_bf16")) return e;
            a.splits = choose_splits_bf16(d, cfg);
            a.tickets = (unsigned*)workspace;
            a.slabs = (float*)((char*)workspace + SPLITK_TICKET_BYTES_B);
            return launch_bf16_splitk(a, s);
        }
    }
    if (cfg == 60) {                                                 // row strips (explicit code; auto: choose_config_bf16)
        const int bm = strip_rows_bf16(d, mask_bf16 != nullptr, y_is_f32);
        if (bm && (!residual_bf16 || (reinterpret_cast<uintptr_t>(residual_bf16) & 15) == 0) && (reinterpret_cast<uintptr_t>(y) & 15) == 0
            && (size_t)a.M * d->cout * 2 < 0x7fffffffull && (size_t)a.M * d->cin * 2 < 0x7fffffffull)
            return bm == 128 ? launch_strip_bf16<128>(a, s) : launch_strip_bf16<64>(a, s);
    }
    switch (cfg == 60 ? 47 : cfg) {                                  // (60 on a shape that is not a strip shape: the 128x128 direct-to-LDS tile)
        case 1: case 11: return launch_bf16<2, 2>(a, s);
        case 2: case 12: return launch_bf16<1, 1>(a, s);
        case 3: case 13: return launch_bf16<2, 1>(a, s);
        case 41: return launch_bf16<1, 2, 4, 2>(a, s);           // 128x128, 8 waves (4x2)
        case 42: return launch_bf16<2, 1, 2, 4>(a, s);           // 128x128, 8 waves (2x4)
        case 43: return launch_bf16<1, 1, 4, 2>(a, s);           // 128x64, 8 waves
        case 44: return launch_bf16<1, 1, 4, 4>(a, s);           // 128x128, 16 waves
        case 45: if (!a.mask) return launch_bf16_v<4, 2, 2, 4, false, 3>(a, s);     // 256x256, 8 waves, direct-to-LDS staging
                 return launch_bf16<2, 1, 2, 4>(a, s);
        case 46: if (!a.mask) return launch_bf16_v<2, 2, 2, 4, false, 3>(a, s);     // 128x256, 8 waves, direct-to-LDS staging
                 return launch_bf16<2, 1, 2, 4>(a, s);
        // (the masked launches -- training's input gradients, 2-3 k rows -- stay on the register-staged forms)
        case 47: if (!a.mask) return launch_bf16_v<2, 1, 2, 4, false, 3>(a, s);     // 128x128, 8 waves, direct-to-LDS staging (two workgroups per CU)
                 return launch_bf16<2, 1, 2, 4>(a, s);
        case 49: if (!a.mask) return launch_bf16_v<2, 1, 2, 4, false, 4>(a, s);     // 128x128, 8 waves, direct-to-LDS RING of four buffers (one workgroup per CU)
                 return launch_bf16<2, 1, 2, 4>(a, s);
        case 48: if (!a.mask) return launch_bf16_v<1, 1, 2, 2, false, 3>(a, s);     // 64x64, 4 waves, direct-to-LDS staging
                 return launch_bf16<1, 1>(a, s);
        default: return fail(FRCNN_E_ARG, "conv2d_fwd_bf16: unknown tile config %d", cfg);
    }
}

int frcnn_refresh_packed_bf16(const frcnn_pack_job* jobs, int n_jobs, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(FRCNN_E_ARG, "refresh_packed_bf16: bad argument");
    for (int i = 0; i < n_jobs; ++i) {
        const frcnn_pack_job& j = jobs[i];
        if (!j.w_hwio || j.kh <= 0 || j.kw <= 0 || j.cin <= 0 || j.cout <= 0) return fail(FRCNN_E_ARG, "refresh_packed_bf16: job %d is malformed", i);
        if (j.packed && (j.cin % BKH)) return fail(FRCNN_E_UNSUPPORTED, "refresh_packed_bf16: job %d: cin must be a multiple of 64", i);
        if (j.packed_dgrad && (j.cout % BKH)) return fail(FRCNN_E_UNSUPPORTED, "refresh_packed_bf16: job %d: cout must be a multiple of 64 for the input-gradient form", i);
    }
    for (int b = 0; b < n_jobs; b += REFRESH_JOBS_B) {
        RefreshTableB t;
        const int n = n_jobs - b < REFRESH_JOBS_B ? n_jobs - b : REFRESH_JOBS_B;
        int blocks = 0;
        for (int i = 0; i < n; ++i) { t.job[i] = jobs[b + i]; t.first_block[i] = blocks; blocks += refresh_blocks_b(jobs[b + i]); }
        for (int i = n; i < REFRESH_JOBS_B; ++i) { t.job[i] = jobs[b]; t.first_block[i] = blocks; }
        t.first_block[REFRESH_JOBS_B] = blocks;
        t.n = n;
        k_refresh_packed_bf16<<<blocks, 256, 0, as_stream(stream)>>>(t);
        if (int e = check_launch("refresh_packed_bf16")) return e;
    }
    return FRCNN_OK;
}

int frcnn_cast_bf16_to_f32(const void* x_bf16, size_t n, float* y, void* stream) {
    if (!x_bf16 || !y) return fail(FRCNN_E_ARG, "cast_bf16_to_f32: null pointer");
    if (n == 0) return FRCNN_OK;
    k_cast_bf16_f32<<<ew_grid_b(n), 256, 0, as_stream(stream)>>>((const __bf16*)x_bf16, n, y);
    return check_launch("cast_bf16_to_f32");
}

int frcnn_relu_bwd_inplace_bf16(void* g_bf16, const void* y_bf16, size_t n, void* stream) {
    if (!g_bf16 || !y_bf16) return fail(FRCNN_E_ARG, "relu_bwd_inplace_bf16: null pointer");
    if (n == 0) return FRCNN_OK;
    k_relu_bwd_bf16<<<ew_grid_b(n), 256, 0, as_stream(stream)>>>((__bf16*)g_bf16, (const __bf16*)y_bf16, n);
    return check_launch("relu_bwd_inplace_bf16");
}

int frcnn_avgpool_bwd_masked_bf16(const float* g_pooled, const void* y_bf16, int n, int k, int c, void* gx_bf16, void* stream) {
    if (!g_pooled || !y_bf16 || !gx_bf16 || n <= 0 || k <= 0 || c <= 0) return fail(FRCNN_E_ARG, "avgpool_bwd_masked_bf16: bad argument");
    const size_t total = (size_t)n * k * k * c;
    k_avgpool_bwd_masked_bf16<<<ew_grid_b(total), 256, 0, as_stream(stream)>>>(g_pooled, (const __bf16*)y_bf16, k * k, c, total, 1.0f / (float)(k * k), (__bf16*)gx_bf16);
    return check_launch("avgpool_bwd_masked_bf16");
}

int frcnn_cast_f32_to_bf16(const float* x, size_t n, void* y_bf16, void* stream) {
    if (!x || !y_bf16 || (n & 3)) return fail(FRCNN_E_ARG, "cast_f32_to_bf16: bad argument (n must be a multiple of 4)");
    if (n == 0) return FRCNN_OK;
    k_cast_f32_bf16<<<ew_grid_b(n / 4), 256, 0, as_stream(stream)>>>((const float4*)x, n / 4, (__bf16*)y_bf16);
    return check_launch("cast_f32_to_bf16");
}

int frcnn_avgpool_bf16_to_f32(const void* x_bf16, int n, int k, int c, float* y, void* stream) {
    return frcnn_avgpool_bf16_to_f32_ex(x_bf16, n, k, c, 0, y, stream);
}

int frcnn_avgpool_bf16_to_f32_ex(const void* x_bf16, int n, int k, int c, int layout, float* y, void* stream) {
    if (!x_bf16 || !y || n <= 0 || k <= 0 || c <= 0) return fail(FRCNN_E_ARG, "avgpool_bf16_to_f32: bad argument");
    if ((c & 7) == 0 && ((reinterpret_cast<uintptr_t>(x_bf16) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
        k_avgpool_bf16_f32_v8<<<ew_grid_b((size_t)n * c / 8), 256, 0, as_stream(stream)>>>((const __bf16*)x_bf16, n, k * k, c, layout, y);
    else
        k_avgpool_bf16_f32<<<ew_grid_b((size_t)n * c), 256, 0, as_stream(stream)>>>((const __bf16*)x_bf16, n, k * k, c, layout, y);
    return check_launch("avgpool_bf16_to_f32");
}

int frcnn_roi_crop_resize_fwd_bf16(const void* feat_bf16, int rows, int cols, int C, const float* rois, int n, int pool, void* out_bf16, void* stream) {
    return frcnn_roi_crop_resize_fwd_bf16_ex(feat_bf16, rows, cols, C, rois, n, pool, nullptr, 0, 0, out_bf16, stream);
}

int frcnn_roi_crop_resize_fwd_bf16_ex(const void* feat_bf16, int rows, int cols, int C, const float* rois, int n, int pool,
                                      const float* fill, int relu, int layout, void* out_bf16, void* stream) {
    if (n < 0 || rows <= 0 || cols <= 0 || C <= 0 || pool <= 0) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_bf16: bad shape");
    if (n == 0) return FRCNN_OK;
    if (!feat_bf16 || !rois || !out_bf16) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_bf16: null pointer");
    k_roi_fwd_bf16<<<n * pool * pool, 256, 0, as_stream(stream)>>>((const __bf16*)feat_bf16, rows, cols, C, (const float4*)rois, pool,
                                                                   fill, relu, layout, (__bf16*)out_bf16);
    return check_launch("roi_crop_resize_fwd_bf16");
}

int frcnn_roi_crop_resize_fwd_bf16_batch(const void* feat_bf16, int n_img, int rows, int cols, int C, const float* rois, int n_per_img, int pool,
                                         const float* fill, int relu, int layout, void* out_bf16, void* stream) {
    if (n_img <= 0 || n_per_img <= 0 || rows <= 0 || cols <= 0 || C <= 0 || pool <= 0) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_bf16_batch: bad shape");
    if (C % 8) return fail(FRCNN_E_UNSUPPORTED, "roi_crop_resize_fwd_bf16_batch: C must be a multiple of 8");
    if (!feat_bf16 || !rois || !out_bf16) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_bf16_batch: null pointer");
    if ((reinterpret_cast<uintptr_t>(feat_bf16) | reinterpret_cast<uintptr_t>(out_bf16)) & 15) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_bf16_batch: 16-byte aligned tensors");
    const long long blocks = (long long)n_img * n_per_img * pool * pool;
    if (blocks > 0x7fffffffLL) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_bf16_batch: too many RoIs");
    // one workgroup per output pixel, one lane per 8 channels: 64 lanes for the 512-channel map, 256 for the 2048-channel one
    const int threads = C / 8 >= 256 ? 256 : (C / 8 > 64 ? 128 : 64);
    k_roi_fwd_bf16_batch<<<(unsigned)blocks, threads, 0, as_stream(stream)>>>((const __bf16*)feat_bf16, rows, cols, C, (const float4*)rois, n_per_img, pool,
                                                                          fill, relu, layout, (__bf16*)out_bf16);
    return check_launch("roi_crop_resize_fwd_bf16_batch");
}

}  // extern "C"
