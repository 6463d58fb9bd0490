// fp32 implicit-GEMM convolution on the bf16 matrix cores by EXACT three-way operand splitting ("bf16x6"), gfx950.
//
// Same contract, operands and layouts as k_conv_igemm_f32_v2 (conv_igemm.hip: f32 NHWC / position-major activations in,
// f32 out, fused scale / shift / residual / mask / activation epilogue, filter taps walked per 32-channel chunk, padding
// halo and ragged edges on the buffer descriptors) -- only the multiply differs:
//
//   every f32 value is the exact sum of three bf16 values,  a = a1 + a2 + a3  (8 + 8 + 8 significand bits: a1 = bf16(a),
//   a2 = bf16(a - a1), a3 = a - a1 - a2, each subtraction exact), likewise b.  Of the nine partial products the six with
//   i + j <= 4 are kept -- a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 -- each EXACT in f32 (8 x 8 bits); the three dropped ones lie
//   below 2^-24 |ab|, under one f32 rounding of the product.  The sum over k accumulates in f32 inside
//   v_mfma_f32_32x32x16_bf16 exactly as it does inside v_mfma_f32_32x32x2_f32.
//
// Measured against fp64 (scripts/micro/x6_lab.hip, K = 512 / 4608, mixed-sign operands): max |err| / sum|ab| 2.19e-7 /
// 1.83e-7 here against 2.69e-7 / 2.15e-7 for the native f32 MFMA -- the result is an fp32 GEMM, not a reduced-precision
// one (the two-term "bf16x3" form measures 1.2e-6, plain bf16 5.6e-4).
// Why: the bf16 MFMA issues 16x the FLOP of the f32 MFMA per cycle (MI355X_MICROARCH.md, matrix cores), so six of them per
// product are 2.67x the native fp32 matrix rate on paper (416 vs 157 TFLOP/s), with HALF the LDS fragment reads per MFMA of
// the plain bf16 kernel (nine 16-byte reads feed twelve MFMAs).
//
// Operands: the activation tile is split in the loader (global f32 -> registers -> three bf16 planes in LDS: ~7 VALU
// operations per element, once per element per 128 output columns); the filter is split once at pack time into three
// planes [3][Cout][Kpad] bf16 in the f32 kernel's k order (frcnn_pack_conv_weights_x6).
// LDS: six planes of unpadded 64-byte rows; the 16-byte slot s of row r is stored at slot s ^ ((r >> 2) & 3), which puts the
// sixteen rows of every ds_read_b128 lane group on sixteen distinct four-bank slots and keeps the two rows of a store group
// on different halves of the 32 store banks (PMC: SQ_LDS_BANK_CONFLICT 28 % of the LDS cycles with padded 80-byte rows).
// ONE buffer (48 KB for a 128x128 tile); the next two chunks wait in registers (requested two chunks ahead, so a load has a
// whole chunk period to land); two workgroups per CU cover each other's split / store / barrier phase.  The 16-byte
// epilogue turns the tile through the same LDS in wave-row passes (64 rows at a time), so the workgroup never holds more
// than the operand planes: what is left of the CU's 160 KB stays available to other streams' workgroups.
#include "conv_f32_common.h"

namespace frcnn {

// The matrix instruction (conv_h3.hip H3_S16's note): v_mfma_f32_16x16x32_bf16 -- one instruction spans the 32-deep chunk of a 16x16 block, a
// wave's 32x32 block is four of them; the chip holds a higher clock on this shape under load.  Every launch form of the engine uses it.
constexpr bool X6_S16 = true;
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
// one chunk of a wave's TM x TN blocks from the six LDS planes; the six products of a block in the engine's order (smallest first)
template <int TM, int TN>
__device__ __forceinline__ void x6_chunk(const char* a0, int a_plane, const char* b0, int b_plane, int lane, f32x4 (&s)[TM][TN][4]) {
    const int r = lane & 15, off = r * X6_ROWB + 16 * ((lane >> 4) ^ x6_swz(r));
    bf16x8s fb[3][TN][2];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) fb[pl][j][ci] = *reinterpret_cast<const bf16x8s*>(b0 + pl * b_plane + (j * 32 + ci * 16) * X6_ROWB + off);
    constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            bf16x8s fa[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const bf16x8s*>(a0 + pl * a_plane + (i * 32 + ri * 16) * X6_ROWB + off);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ci = 0; ci < 2; ++ci)
                        s[i][j][ri * 2 + ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[IA[t]], fb[IB[t]][j][ci], s[i][j][ri * 2 + ci], 0, 0, 0);
        }
}
template <int TM, int TN>
__device__ __forceinline__ void x6_gather(const f32x4 (&s)[TM][TN][4], f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = s[i][j][e >> 2][e & 3];
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void x6_split(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 a1 = (__bf16)v[e];                       // round to nearest even (v_cvt_pk_bf16_f32)
        const float r1 = v[e] - (float)a1;                    // exact
        const __bf16 a2 = (__bf16)r1;
        const float r2 = r1 - (float)a2;                      // exact; at most 8 significant bits are left
        h[e] = a1; m[e] = a2; l[e] = (__bf16)r2;
    }
}

// SPLITK (64x64 tiles): small grids with a long k loop (rpn_conv1, stage 4's 3x3 at 2 394 rows) cut the chunk sequence into
// `splits` slices, one workgroup each; the f32 kernel's protocol (conv_igemm.hip): write-through partial tiles, a ticket per
// tile, the last arriver sums the slabs in slice order (bitwise reproducible) and runs the epilogue.
template <int TM, int TN, int WM, int WN, bool SPLITK = false>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu((TM * TN == 1 || (TM * TN == 2 && WM * WN == 8)) ? 4 : 1)))      // (64x64 and the two-workgroup 128x128 tile: four waves per SIMD, as with the 32x32x16 form)
k_conv_igemm_x6(const ConvArgs p) {
    using T = X6Tile<TM, TN, WM, WN>;
    constexpr int NT = T::NT, BM = T::BM, BN = T::BN;
    constexpr int RPP = NT / 8;                           // tile rows staged per pass of A (8 lanes x 16 B of f32 per row)
    constexpr int PA = BM / RPP;
    constexpr int RPB = NT / 4;                           // rows per pass of one B plane (4 lanes x 16 B of bf16 per row)
    constexpr int PB = BN / RPB;
    static_assert(BM % RPP == 0 && BN % RPB == 0 && PA >= 1 && PB >= 1, "tile rows must be a multiple of the staging pass");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* As = lds;                                       // [3][BM][X6_ROWB]
    char* Bs = lds + 3 * BM * X6_ROWB;                    // [3][BN][X6_ROWB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int splits = SPLITK ? p.splits : 1;
    const int nwg = p.tiles_m * p.tiles_n * splits;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tile = SPLITK ? logical / splits : logical;         // a tile's slices are neighbours on one XCD
    const int slice = SPLITK ? logical - tile * splits : 0;
    int tile_n = tile / p.tiles_m, tile_m = tile - tile_n * p.tiles_m;
    if (p.group_m > 0) {                                  // grouped order (see k_conv_igemm_f32_v2): g row tiles x all column tiles
        const int per = p.group_m * p.tiles_n, g = tile / per, m_base = g * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - m_base), r = tile - g * per;
        tile_n = r / gm;
        tile_m = m_base + r - tile_n * gm;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const size_t plane_bytes = (size_t)p.Cout * p.Kpad * 2;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)(3 * plane_bytes), 0x00020000);         // p.w: the three bf16 planes

    const int lrow = tid >> 3, lcol = (tid & 7) * 4;      // A: row within a pass, first of this lane's four channels
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
            a_off[i] = (img * p.img_stride + (a_h[i] * p.W + a_w[i]) * p.pix_stride + lcol) * 4;     // may be "negative" in the halo
        } else {
            a_h[i] = -(1 << 28); a_w[i] = 0; a_off[i] = 0;
        }
    }
    const int brow = tid >> 2, bcol = (tid & 3) * 8;      // B: row within a pass, first of this lane's eight k
    unsigned b_off[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + brow + RPB * i;
        b_off[i] = n < p.Cout ? (unsigned)((n * p.Kpad + bcol) * 2) : OOB_OFFSET;
    }

    // filter taps this tile needs (position-major rows: taps that meet only zero padding for ALL rows of the tile are skipped)
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
    const int nk_all = (p.Kpad / (BK * RS)) * n_taps;     // chunks of this tile: (channel group, needed tap) pairs
    const int kb = SPLITK ? (int)((long long)slice * nk_all / splits) : 0;
    const int nk = (SPLITK ? (int)((long long)(slice + 1) * nk_all / splits) : nk_all) - kb;      // this workgroup's chunks

    unsigned rem = tap_mask;                              // taps of the current channel group still to load
    int c0 = 0, w_grp = 0;                                // channel offset / bf16 byte offset of the group's filter chunks
    if (SPLITK) {
        const int grp = kb / n_taps;
        c0 = grp * BK; w_grp = grp * RS * (BK * 2);
        for (int t = kb - grp * n_taps; t > 0; --t) rem &= rem - 1;
    }
    // the NEXT chunk of the sequence -> staging set S (calls walk the sequence in order; calls past its end fetch in-bounds
    // or zero data that is never multiplied)
    f32x4 ra[2][PA];
    f32x4 rb[2][3][PB];
    auto load_next = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
        const int tap = __builtin_ctz(rem);               // wave-uniform
        const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
        const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 4;
        const int w_off = w_grp + tap * (BK * 2);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                rb[S][pl][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    wrsrc, b_off[i] == OOB_OFFSET ? OOB_OFFSET : b_off[i] + (unsigned)(pl * plane_bytes), w_off, 0));
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            ra[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET, 0, 0));
        }
        rem &= rem - 1;                                   // branch-free walk to the next needed tap
        const int wrap = (rem == 0);
        rem |= wrap ? tap_mask : 0u;
        c0 += wrap * BK;
        w_grp += wrap * (RS * BK * 2);
    };
    auto store = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bf16x4 h, m, l;
            x6_split(ra[S][i], h, m, l);
            const int row = lrow + RPP * i, g = tid & 7;
            char* dst = As + row * X6_ROWB + 16 * ((g >> 1) ^ x6_swz(row)) + 8 * (g & 1);
            *reinterpret_cast<bf16x4*>(dst) = h;
            *reinterpret_cast<bf16x4*>(dst + BM * X6_ROWB) = m;
            *reinterpret_cast<bf16x4*>(dst + 2 * BM * X6_ROWB) = l;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                *reinterpret_cast<f32x4*>(Bs + pl * BN * X6_ROWB + (brow + RPB * i) * X6_ROWB + 16 * ((tid & 3) ^ x6_swz(brow + RPB * i))) = rb[S][pl][i];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    f32x4 s16[TM][TN][4];                                 // (X6_S16) the same accumulators as sixteen-row blocks
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) s16[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const char* abase = As + (wm * TM * 32 + li) * X6_ROWB;
    const char* bbase = Bs + (wn * TN * 32 + li) * X6_ROWB;
    const int koff[2] = {16 * (lh ^ x6_swz(li)), 16 * ((2 + lh) ^ x6_swz(li))};       // k-step s reads logical slot 2 s + lh (tile bases are multiples of 32 rows)
    auto compute = [&]() {
        if constexpr (X6_S16) {
            x6_chunk<TM, TN>(As + wm * TM * 32 * X6_ROWB, BM * X6_ROWB, Bs + wn * TN * 32 * X6_ROWB, BN * X6_ROWB, lane, s16);
            return;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {                     // two k-steps of 16 per 32-channel chunk
            bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(abase + pl * BM * X6_ROWB + i * 32 * X6_ROWB + koff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(bbase + pl * BN * X6_ROWB + j * 32 * X6_ROWB + koff[s]);
            }
            // smallest terms first: (a3,b1) (a1,b3) (a2,b2) (a2,b1) (a1,b2) (a1,b1)
            constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[IA[t]][i], fb[IB[t]][j], acc[i][j], 0, 0, 0);
        }
    };
    // chunk c waits in set c & 1.  Iteration: MFMAs of chunk kt from LDS | barrier | split + store chunk kt+1 (requested one
    // iteration earlier) and request chunk kt+3 into the registers just freed | barrier
    load_next(I0{});                                      // chunk 0
    load_next(I1{});                                      // chunk 1
    store(I0{});
    load_next(I0{});                                      // chunk 2
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        compute();
        __syncthreads();
        store(I1{});                                      // chunk kt+1
        load_next(I1{});                                  // chunk kt+3
        __syncthreads();
        compute();
        __syncthreads();
        if (kt + 2 < nk) {
            store(I0{});                                  // chunk kt+2
            load_next(I0{});                              // chunk kt+4
            __syncthreads();
        }
    }
    if (kt < nk) {
        compute();
        __syncthreads();                                  // the epilogue reuses the buffer
    }
    if constexpr (X6_S16) x6_gather<TM, TN>(s16, acc);
    if constexpr (SPLITK) {
        // publish this slice's partial tile WRITE-THROUGH (sc1 stores need no release fence), thread-major 16-byte rows
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
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), srsrc, slab_off + ((i * TN + j) * 4 + q) * (NT * 16), 0, 16 /* sc1 */);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains ...
        __syncthreads();                                       // ... before ONE lane draws the ticket
        int* last = reinterpret_cast<int*>(lds);
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(&p.tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int is_last = (t == (unsigned)(splits - 1));
            if (is_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                 // drop this CU's stale L1 lines
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&p.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            *last = is_last;
        }
        __syncthreads();
        if (!*last) return;
        const float4* base = reinterpret_cast<const float4*>(p.slabs + (size_t)tile * splits * (BM * BN));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        for (int sl = 0; sl < splits; ++sl) {                  // slice order: two runs are bitwise equal
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
        __syncthreads();                                       // the flag word is read; the epilogue reuses the buffer
    }
    if (p.vec_epi) x6_epilogue_vec<TM, TN, WM, WN, false, X6_S16>(acc, p, m0, n0, tid, wm, wn, li, lh, reinterpret_cast<float*>(lds));
    else epilogue<TM, TN, X6_S16>(acc, p, m0, n0, wm, wn, li, lh);
}

// ---- the same engine as ONE 16-wave workgroup per CU (tile code 76): 256x128 tile, four waves per SIMD from one workgroup, TWO LDS
// buffers (147 KB) and ONE barrier per chunk -- chunk kt multiplies from buffer kt & 1 while chunk kt+1 (in registers since the
// previous iteration) is split and stored into the other buffer and chunk kt+2 is requested.  No phase in which the CU's matrix
// pipe has nothing queued because both of its workgroups are staging at once: alone on the chip 341 / 170 / 161 us on the head's
// 3x3 / 512->2048 / 2048->512 GEMMs against 382 / 196 / 189 for the two-workgroup form (scripts/micro/x6_lab.hip); it takes the
// whole CU's LDS, so nothing of another stream co-resides with it.
__global__ void __launch_bounds__(1024) k_conv_igemm_x6_db(const ConvArgs p) {
    constexpr int TM = 2, TN = 1, WM = 4, WN = 4, BM = 256, BN = 128, NT = 1024;
    constexpr int RPP = NT / 8, PA = BM / RPP;            // A: 128 rows per pass, 2 passes
    constexpr int NBP = 3 * BN * 4;                       // 16-byte pieces of the three filter planes per chunk: 1536
    constexpr int PBT = (NBP + NT - 1) / NT;              // 2 (the second one for tid < 512 only)
    constexpr int BUFB = 3 * (BM + BN) * X6_ROWB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    int tile_n = tile / p.tiles_m, tile_m = tile - tile_n * p.tiles_m;
    if (p.group_m > 0) {
        const int per = p.group_m * p.tiles_n, g = tile / per, m_base = g * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - m_base), r = tile - g * per;
        tile_n = r / gm;
        tile_m = m_base + r - tile_n * gm;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const size_t plane_bytes = (size_t)p.Cout * p.Kpad * 2;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)(3 * plane_bytes), 0x00020000);

    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    int a_h[PA], a_w[PA], a_off[PA], a_lds[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = lrow + RPP * i, m = m0 + row, g = tid & 7;
        a_lds[i] = row * X6_ROWB + 16 * ((g >> 1) ^ x6_swz(row)) + 8 * (g & 1);
        if (m < p.M) {
            int wo, ho, img;
            if (p.layout) { img = m % p.n_img; const int pos = m / p.n_img; ho = pos / p.Wo; wo = pos - ho * p.Wo; }
            else { wo = m % p.Wo; const int t = m / p.Wo; ho = t % p.Ho; img = t / p.Ho; }
            a_h[i] = ho * p.stride - p.pad_top;
            a_w[i] = wo * p.stride - p.pad_left;
            a_off[i] = (img * p.img_stride + (a_h[i] * p.W + a_w[i]) * p.pix_stride + lcol) * 4;
        } else {
            a_h[i] = -(1 << 28); a_w[i] = 0; a_off[i] = 0;
        }
    }
    unsigned b_off[PBT];
    int b_lds[PBT];
#pragma unroll
    for (int i = 0; i < PBT; ++i) {
        const int q = tid + NT * i, pl = q / (BN * 4), r = q % (BN * 4), row = r >> 2, g = r & 3;
        const bool ok = q < NBP && n0 + row < p.Cout;
        b_off[i] = ok ? (unsigned)(((size_t)(n0 + row) * p.Kpad + g * 8) * 2 + pl * plane_bytes) : OOB_OFFSET;
        b_lds[i] = q < NBP ? 3 * BM * X6_ROWB + pl * BN * X6_ROWB + row * X6_ROWB + 16 * (g ^ x6_swz(row)) : -1;
    }

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
    const int nk = (p.Kpad / (BK * RS)) * n_taps;
    unsigned rem = tap_mask;
    int c0 = 0, w_grp = 0;
    f32x4 ra[PA], rb[PBT];
    auto load_next = [&]() {
        const int tap = __builtin_ctz(rem);
        const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
        const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 4;
        const int w_off = w_grp + tap * (BK * 2);
#pragma unroll
        for (int i = 0; i < PBT; ++i) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i], w_off, 0));
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET, 0, 0));
        }
        rem &= rem - 1;
        const int wrap = (rem == 0);
        rem |= wrap ? tap_mask : 0u;
        c0 += wrap * BK;
        w_grp += wrap * (RS * BK * 2);
    };
    auto store = [&](int buf) {
        char* base = lds + buf * BUFB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            bf16x4 h, m, l;
            x6_split(ra[i], h, m, l);
            *reinterpret_cast<bf16x4*>(base + a_lds[i]) = h;
            *reinterpret_cast<bf16x4*>(base + BM * X6_ROWB + a_lds[i]) = m;
            *reinterpret_cast<bf16x4*>(base + 2 * BM * X6_ROWB + a_lds[i]) = l;
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
    f32x4 s16[TM][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) s16[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int aoff = (wm * TM * 32 + li) * X6_ROWB, boff = 3 * BM * X6_ROWB + (wn * TN * 32 + li) * X6_ROWB;
    const int koff[2] = {16 * (lh ^ x6_swz(li)), 16 * ((2 + lh) ^ x6_swz(li))};
    auto kstep = [&](int buf, int s) {
        const char* base = lds + buf * BUFB;
        bf16x8 fa[3][TM], fb[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const bf16x8*>(base + aoff + pl * BM * X6_ROWB + i * 32 * X6_ROWB + koff[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const bf16x8*>(base + boff + pl * BN * X6_ROWB + j * 32 * X6_ROWB + koff[s]);
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
    load_next();                                          // chunk 0
    store(0);
    load_next();                                          // chunk 1 (past-the-end fetches are never multiplied)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) { store(buf ^ 1); load_next(); }
        if constexpr (X6_S16) {
            const char* base = lds + buf * BUFB;
            x6_chunk<TM, TN>(base + wm * TM * 32 * X6_ROWB, BM * X6_ROWB, base + 3 * BM * X6_ROWB + wn * TN * 32 * X6_ROWB, BN * X6_ROWB, lane, s16);
        } else {
            kstep(buf, 0);
            kstep(buf, 1);
        }
        __syncthreads();
    }
    if constexpr (X6_S16) x6_gather<TM, TN>(s16, acc);
    if (p.vec_epi) x6_epilogue_vec<TM, TN, WM, WN, false, X6_S16>(acc, p, m0, n0, tid, wm, wn, li, lh, reinterpret_cast<float*>(lds));
    else epilogue<TM, TN, X6_S16>(acc, p, m0, n0, wm, wn, li, lh);
}

static int launch_x6_db(const ConvArgs& a, hipStream_t s) {
    ConvArgs p = a;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.Cout + 127) / 128;
    constexpr size_t lds = (size_t)2 * 3 * (256 + 128) * X6_ROWB;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_x6_db, lds, "conv2d_x6")) return e;
    k_conv_igemm_x6_db<<<p.tiles_m * p.tiles_n, 1024, lds, s>>>(p);
    return check_launch("conv2d_fwd_x6");
}

// f32 packed filter [Cout][Kpad] (frcnn_pack_conv_weights' k order) -> three bf16 planes [3][Cout][Kpad]
__global__ void k_pack_x6(const float* w, size_t n, __bf16* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = w[i];
        const __bf16 a1 = (__bf16)v;
        const float r1 = v - (float)a1;
        const __bf16 a2 = (__bf16)r1;
        out[i] = a1; out[n + i] = a2; out[2 * n + i] = (__bf16)(r1 - (float)a2);
    }
}

// the same for MANY filters in one launch (a training step re-derives the planes of every trainable layer's forward and
// input-gradient filter after the optimiser has rewritten the f32 forms): a workgroup finds its job by the block prefix
constexpr int X6_REFRESH_JOBS = 48;
struct X6RefreshTable { const float* w[X6_REFRESH_JOBS]; __bf16* out[X6_REFRESH_JOBS]; unsigned long long n[X6_REFRESH_JOBS]; int first_block[X6_REFRESH_JOBS + 1]; int jobs; };
__global__ void __launch_bounds__(256) k_pack_x6_batch(const X6RefreshTable t) {
    int j = 0;
    while (j + 1 < t.jobs && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const size_t n = t.n[j];
    const int nb = t.first_block[j + 1] - t.first_block[j], b = (int)blockIdx.x - t.first_block[j];
    const float* w = t.w[j];
    __bf16* out = t.out[j];
    for (size_t i = ((size_t)b * 256 + threadIdx.x) * 4; i < n; i += (size_t)nb * 256 * 4) {       // n % 4 == 0 (packed k is a multiple of 32)
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + i);
        typedef __bf16 bf16x4p __attribute__((ext_vector_type(4)));
        bf16x4p a1, a2, a3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a1[e] = (__bf16)v[e];
            const float r1 = v[e] - (float)a1[e];
            a2[e] = (__bf16)r1;
            a3[e] = (__bf16)(r1 - (float)a2[e]);
        }
        *reinterpret_cast<bf16x4p*>(out + i) = a1; *reinterpret_cast<bf16x4p*>(out + n + i) = a2; *reinterpret_cast<bf16x4p*>(out + 2 * n + i) = a3;
    }
}

template <int TM, int TN, int WM, int WN>
static int launch_x6(const ConvArgs& a, hipStream_t s) {
    using T = X6Tile<TM, TN, WM, WN>;
    ConvArgs p = a;
    p.tiles_m = (p.M + T::BM - 1) / T::BM;
    p.tiles_n = (p.Cout + T::BN - 1) / T::BN;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_x6<TM, TN, WM, WN>, T::lds, "conv2d_x6")) return e;
    k_conv_igemm_x6<TM, TN, WM, WN><<<p.tiles_m * p.tiles_n, T::NT, T::lds, s>>>(p);
    return check_launch("conv2d_fwd_x6");
}

// tile codes of the split-bf16 engine (frcnn_conv_desc.tile; 0 / 50 = auto)
int launch_conv_x6(const ConvArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case 71: return launch_x6<2, 1, 2, 4>(a, s);      // 128x128, 8 waves (64x32 per wave), two workgroups per CU
        case 72: return launch_x6<2, 2, 4, 2>(a, s);      // 256x128, 8 waves (64x64 per wave), one workgroup per CU
        case 73: return launch_x6<2, 2, 2, 2>(a, s);      // 128x128, 4 waves (64x64 per wave)
        case 74: return launch_x6<1, 1, 2, 2>(a, s);      // 64x64, 4 waves
        case 75: return launch_x6<1, 1, 2, 4>(a, s);      // 64x128, 8 waves
        case 76: return launch_x6_db(a, s);               // 256x128, 16 waves, two LDS buffers, one workgroup per CU
        case 77: return launch_x6<2, 1, 2, 2>(a, s);      // 128x64, 4 waves (64x32 per wave): the 64-column layers
        case 174: {                                       // 64x64 with split-K (a.splits / a.slabs / a.tickets set by the caller)
            using T = X6Tile<1, 1, 2, 2>;
            ConvArgs p = a;
            p.tiles_m = (p.M + 63) / 64;
            p.tiles_n = (p.Cout + 63) / 64;
            k_conv_igemm_x6<1, 1, 2, 2, true><<<p.tiles_m * p.tiles_n * p.splits, T::NT, T::lds, s>>>(p);
            return check_launch("conv2d_fwd_x6 (split-K)");
        }
        case 171: {                                       // 128x128 on eight waves with split-K: the taller small grids (>= 64 such tiles)
            using T = X6Tile<2, 1, 2, 4>;
            ConvArgs p = a;
            p.tiles_m = (p.M + 127) / 128;
            p.tiles_n = (p.Cout + 127) / 128;
            static std::atomic<uint64_t> lds_seen{0};
            if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_x6<2, 1, 2, 4, true>, T::lds, "conv2d_x6")) return e;
            k_conv_igemm_x6<2, 1, 2, 4, true><<<p.tiles_m * p.tiles_n * p.splits, T::NT, T::lds, s>>>(p);
            return check_launch("conv2d_fwd_x6 (split-K)");
        }
        default: return fail(FRCNN_E_ARG, "conv2d_fwd_x6: unknown tile config %d", cfg);
    }
}

int x6_tile_width(int cfg) { return (cfg == 74 || cfg == 77) ? 64 : 128; }

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_pack_conv_weights_x6(const float* w_packed, int cout, int kpad, void* planes_bf16, void* stream) {
    if (!w_packed || !planes_bf16 || cout <= 0 || kpad <= 0 || (kpad % 32)) return fail(FRCNN_E_ARG, "pack_conv_weights_x6: bad argument");
    const size_t n = (size_t)cout * kpad;
    if (3 * n * 2 >= 0x7fffffffull) return fail(FRCNN_E_UNSUPPORTED, "pack_conv_weights_x6: filter planes over 2 GiB");
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    k_pack_x6<<<grid, 256, 0, as_stream(stream)>>>(w_packed, n, (__bf16*)planes_bf16);
    return check_launch("pack_conv_weights_x6");
}

extern "C" int frcnn_refresh_x6_planes(const frcnn_x6_job* jobs, int n_jobs, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(FRCNN_E_ARG, "refresh_x6_planes: bad argument");
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs[i].w_packed || !jobs[i].planes_bf16 || jobs[i].rows <= 0 || jobs[i].kpad <= 0 || (jobs[i].kpad % 32)
            || (reinterpret_cast<uintptr_t>(jobs[i].w_packed) & 15) || (reinterpret_cast<uintptr_t>(jobs[i].planes_bf16) & 7))
            return fail(FRCNN_E_ARG, "refresh_x6_planes: job %d is malformed (packed k must be a multiple of 32, pointers 16 / 8-byte aligned)", i);
    for (int b = 0; b < n_jobs; b += X6_REFRESH_JOBS) {
        X6RefreshTable t;
        const int n = n_jobs - b < X6_REFRESH_JOBS ? n_jobs - b : X6_REFRESH_JOBS;
        int blocks = 0;
        for (int i = 0; i < X6_REFRESH_JOBS; ++i) {
            const frcnn_x6_job& j = jobs[b + (i < n ? i : 0)];
            t.w[i] = j.w_packed; t.out[i] = (__bf16*)j.planes_bf16; t.n[i] = (unsigned long long)j.rows * j.kpad;
            t.first_block[i] = blocks;
            if (i < n) { const unsigned long long g = (t.n[i] + 4095) / 4096; blocks += (int)(g < 1 ? 1 : (g > 512 ? 512 : g)); }      // 16 elements per thread
        }
        t.first_block[X6_REFRESH_JOBS] = blocks;
        t.jobs = n;
        k_pack_x6_batch<<<blocks, 256, 0, as_stream(stream)>>>(t);
        if (int e = check_launch("refresh_x6_planes")) return e;
    }
    return FRCNN_OK;
}
