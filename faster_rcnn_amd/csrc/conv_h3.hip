// fp32 implicit-GEMM convolution on the fp16 matrix cores by a TWO-way operand split with a scaled low part ("f16x3"), gfx950.
//
// Same contract, operands and layouts as k_conv_igemm_f32_v2 / k_conv_igemm_x6 (f32 NHWC / position-major activations in, f32 out,
// fused scale / shift / residual / mask / activation epilogue, filter taps walked per 32-channel chunk, halo and ragged edges on the
// buffer descriptors) -- only the multiply differs:
//
//   each operand TENSOR is first brought into fp16's range by one power of two (exact):  a' = a * 2^eA,  max|a'| in [2^14, 2^15)
//   (eA from an upper bound of max|a| that the producing launch left in a magnitude record, ConvArgs.x_amax; the filter's from the
//   maximum stored in front of its planes), then
//       ah = f16(a')                  11 significant bits, round to nearest even
//       al = f16((a' - ah) * 2^11)    the EXACT residual (13 bits), scaled up into ah's exponent range, rounded to 11 bits
//   so a' = ah + al * 2^-11 + eps,  |eps| <= 2^-23 |a'|: the residual is a 13-bit multiple of a's last place; when its top bit is set
//   (half of the cases) fp16 keeps it to two such units, i.e. the operand is off by at most ONE unit in its last place; same for b.
//       acc0 += ah * bh               v_mfma_f32_32x32x16_f16: every product exact in f32, f32 accumulate
//       acc1 += ah * bl + al * bh     an accumulator of its own: it carries the weight 2^-11
//       c = (acc0 + acc1 * 2^-11) * 2^-eA * 2^-eB
//   The al * bl term (<= 2^-22 |ab|, typically 2^-25 |ab| with a random sign) is dropped.
//
// THREE MFMAs per block of products where the three-way bf16 split (conv_x6.hip) needs six: the fp32-equivalent ceiling of the matrix
// pipe doubles (2.5 PFLOP/s / 3), and a chunk moves four LDS planes instead of six.  What is given up against conv_x6.hip: the split
// is not exact -- an operand carries 23-24 bits instead of all 24.  Measured against fp64 (scripts/micro/h3_lab.hip, K = 512 / 4608):
// max |err| / sum|ab| 7.2e-8 / 8.4e-8 on mixed-sign operands (native v_mfma_f32_32x32x2_f32: 2.8e-7 / 2.3e-7; the six-product bf16
// split: 2.2e-7 / 1.8e-7), 5.0e-7 / 1.7e-6 when nothing cancels (native 1.4e-6 / 4.0e-6), exact on integers up to 2048, and 2.9e-7
// on the adversarial input whose every residual has the same sign and the largest size (native 1.6e-6): at or under the native f32
// matrix instruction everywhere, because one 16-deep MFMA block rounds once where the 2-deep f32 instruction rounds eight times.
// Range: fp16 keeps full precision over 2^-12 .. 2^15 after scaling, so values down to 2^-27 of the tensor's bound keep all their
// bits and smaller ones lose them gradually (absolute error <= 2^-50 of the bound); a bound up to 2^8 too large costs nothing.
//
// Loop structure, LDS image (unpadded 64-byte rows, XOR-swizzled 16-byte slots) and epilogue are those of conv_x6.hip; operands are
// split in the loader (global f32 -> registers -> two fp16 planes in LDS: one multiply, two conversions and one fused multiply-add
// per element pair, half of the three-way split's arithmetic); the filter is split once at pack time (frcnn_pack_conv_weights_h3).
#include "conv_f32_common.h"

namespace frcnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int H3_HEADER_BYTES = 16;       // in front of a filter's two planes: word 0 = max|w| of the packed filter (f32)

// The matrix instruction.  S16: v_mfma_f32_16x16x32_f16 -- one instruction spans the whole 32-deep chunk of a 16x16 block; a wave's 32x32
// block is four of them.  Same products, same LDS bytes and the same cycles per product as two k-steps of v_mfma_f32_32x32x16_f16, but
// under load the chip holds a higher clock on this shape (MI355X_MICROARCH.md 'DVFS give-back' item 7; lab, scripts/micro/h3_lab.hip
// `ring`: the head's 3x3 GEMM's LDS-read + MFMA loop 166 -> 149 us on random operands).  Every launch form uses the SAME shape, so a
// layer's result does not depend on the tile that computed it (the bitwise tests across tiles, layouts and split-K hold).
constexpr bool H3_S16 = true;
// (LDS slot swizzle: conv_f32_common.h x6_swz, conflict-free for this fragment)
__device__ __forceinline__ int h3_swz(int row) { return x6_swz(row); }

// One 32-deep chunk of a wave's TM x TN blocks of 32x32 from the four LDS planes (a_hi / b_hi: this wave's first row of the high planes;
// the low planes lie a_lo / b_lo bytes behind).  acc0 += ah bh; acc1 += al bh + ah bl.  Accumulator element e of a block -- S16: 16x16
// block e >> 2 (2 * row half + column half), col = lane & 15, row = 4 * (lane >> 4) + (e & 3); else the 32x32 map (epilogue()).
template <int TM, int TN>
__device__ __forceinline__ void h3_chunk(const char* a_hi, int a_lo, const char* b_hi, int b_lo, int lane, f32x4 (&s0)[TM][TN][4], f32x4 (&s1)[TM][TN][4]) {
    static_assert(H3_S16, "the 16x16x32 form");
    const int r = lane & 15, off = r * X6_ROWB + 16 * ((lane >> 4) ^ h3_swz(r));      // (block bases are multiples of 16 rows: the swizzle is the lane's)
    f16x8 fb[2][TN][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) fb[pl][j][ci] = *reinterpret_cast<const f16x8*>(b_hi + pl * b_lo + (j * 32 + ci * 16) * X6_ROWB + off);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(a_hi + (i * 32 + ri * 16) * X6_ROWB + off);
            const f16x8 al = *reinterpret_cast<const f16x8*>(a_hi + a_lo + (i * 32 + ri * 16) * X6_ROWB + off);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    f32x4& c1 = s1[i][j][ri * 2 + ci];
                    f32x4& c0 = s0[i][j][ri * 2 + ci];
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, fb[0][j][ci], c1, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, fb[1][j][ci], c1, 0, 0, 0);
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, fb[0][j][ci], c0, 0, 0, 0);
                }
        }
}
template <int TM, int TN>
__device__ __forceinline__ void h3_gather(const f32x4 (&s)[TM][TN][4], f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = s[i][j][e >> 2][e & 3];
}

// e with amax * 2^e in [2^14, 2^15); clamped so that 2^e, 2^(e + 11) and 2^-e are normal f32 numbers whatever amax is
__host__ __device__ __forceinline__ int h3_exponent(float amax) {
    unsigned b;
    __builtin_memcpy(&b, &amax, 4);
    const int e = 141 - (int)((b >> 23) & 0xffu);         // amax = 1.m * 2^(be - 127)  ->  14 - (be - 127)
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__host__ __device__ __forceinline__ float h3_pow2(int e) {
    const unsigned b = (unsigned)(e + 127) << 23;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// `seen`: the running packed maximum of |h| (conv_f32_common.h h3_see / h3_report: a high part at or above 2^15 means the record the scale
// came from was not an upper bound).  With MODE.FP16_OVFL set (h3_fp16_saturate) a conversion that overflows clamps to +-65504 instead of
// turning into an infinity whose residual is a NaN.
__device__ __forceinline__ void h3_split(const f32x4 v, float s, f16x4& h, f16x4& l, unsigned& seen) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * s;                             // exact: s is a power of two
        const _Float16 a1 = (_Float16)x;                      // round to nearest even (v_cvt_pk_f16_f32)
        const float r = x - (float)a1;                        // exact
        h[e] = a1; l[e] = (_Float16)(r * 2048.0f);
    }
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2 hp = __builtin_bit_cast(i32x2, h);
    h3_see(seen, (unsigned)hp[0]);
    h3_see(seen, (unsigned)hp[1]);
}

// the two scaled accumulators -> the f32 sum of products the epilogues expect
template <int TM, int TN>
__device__ __forceinline__ void h3_combine(f32x16 (&acc0)[TM][TN], const f32x16 (&acc1)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc0[i][j][e] = acc0[i][j][e] + acc1[i][j][e] * (1.0f / 2048.0f);
}
template <int TM, int TN>
__device__ __forceinline__ void h3_unscale(f32x16 (&acc)[TM][TN], float inv_a, float inv_b) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = (acc[i][j][e] * inv_a) * inv_b;     // two exact steps: 2^-(eA + eB) itself may leave f32's range
}

// tile walk shared by the two kernels: which (row tile, column tile) a workgroup owns
__device__ __forceinline__ void h3_tile_of(const ConvArgs& p, int tile, int& tile_m, int& tile_n) {
    tile_n = tile / p.tiles_m; tile_m = tile - tile_n * p.tiles_m;
    if (p.group_m > 0) {                                  // grouped order (see k_conv_igemm_f32_v2): g row tiles x all column tiles
        const int per = p.group_m * p.tiles_n, g = tile / per, m_base = g * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - m_base), r = tile - g * per;
        tile_n = r / gm;
        tile_m = m_base + r - tile_n * gm;
    }
}

// filter taps a tile needs (position-major rows: taps that meet only zero padding for ALL rows of the tile are skipped)
__device__ __forceinline__ unsigned h3_tap_mask(const ConvArgs& p, int m0, int BM) {
    const int RS = p.R * p.S;
    const unsigned all_taps = RS >= 32 ? 0xffffffffu : (1u << RS) - 1u;
    if (!p.layout) return all_taps;
    const int pos_lo = m0 / p.n_img, pos_hi = (min(m0 + BM, p.M) - 1) / p.n_img;
    if (pos_hi - pos_lo >= 8) return all_taps;
    unsigned mk = 0;
    for (int pos = pos_lo; pos <= pos_hi; ++pos) {
        const int ho = pos / p.Wo, wo = pos - ho * p.Wo;
        const int h0 = ho * p.stride - p.pad_top, w0 = wo * p.stride - p.pad_left;
        for (int r = 0; r < p.R; ++r)
            for (int sx = 0; sx < p.S; ++sx)
                if ((unsigned)(h0 + r) < (unsigned)p.H && (unsigned)(w0 + sx) < (unsigned)p.W) mk |= 1u << (r * p.S + sx);
    }
    return mk ? mk : all_taps;
}

// ONE LDS buffer (four planes), the next two chunks waiting in registers, two barriers per chunk: k_conv_igemm_x6's loop.
// SPLITK: the f32 kernel's protocol (write-through partial tiles, a ticket per tile, the last arriver sums the slabs in slice order).
template <int TM, int TN, int WM, int WN, bool SPLITK = false>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(TM * TN == 1 ? 4 : 1)))      // (the 64x64 tile: four waves per SIMD, as with the 32x32x16 form)
k_conv_igemm_h3(const ConvArgs p) {
    using T = X6Tile<TM, TN, WM, WN, 2>;
    constexpr int NT = T::NT, BM = T::BM, BN = T::BN;
    constexpr int RPP = NT / 8;                           // tile rows staged per pass of A (8 lanes x 16 B of f32 per row)
    constexpr int PA = BM / RPP;
    constexpr int RPB = NT / 4;                           // rows per pass of one B plane (4 lanes x 16 B of fp16 per row)
    constexpr int PB = BN / RPB;
    static_assert(BM % RPP == 0 && BN % RPB == 0 && PA >= 1 && PB >= 1, "tile rows must be a multiple of the staging pass");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* As = lds;                                       // [2][BM][X6_ROWB]
    char* Bs = lds + 2 * BM * X6_ROWB;                    // [2][BN][X6_ROWB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int splits = SPLITK ? p.splits : 1;
    const int nwg = p.tiles_m * p.tiles_n * splits;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tile = SPLITK ? logical / splits : logical;         // a tile's slices are neighbours on one XCD
    const int slice = SPLITK ? logical - tile * splits : 0;
    int tile_m, tile_n;
    h3_tile_of(p, tile, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const size_t plane_bytes = (size_t)p.Cout * p.Kpad * 2;
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wbase + H3_HEADER_BYTES), 0, (int)(2 * plane_bytes), 0x00020000);

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

    const int RS = p.R * p.S;
    const unsigned tap_mask = h3_tap_mask(p, m0, BM);
    const int n_taps = __popc(tap_mask);
    const int nk_all = (p.Kpad / (BK * RS)) * n_taps;     // chunks of this tile: (channel group, needed tap) pairs
    const int kb = SPLITK ? (int)((long long)slice * nk_all / splits) : 0;
    const int nk = (SPLITK ? (int)((long long)(slice + 1) * nk_all / splits) : nk_all) - kb;      // this workgroup's chunks

    unsigned rem = tap_mask;                              // taps of the current channel group still to load
    int c0 = 0, w_grp = 0;                                // channel offset / fp16 byte offset of the group's filter chunks
    if (SPLITK) {
        const int grp = kb / n_taps;
        c0 = grp * BK; w_grp = grp * RS * (BK * 2);
        for (int t = kb - grp * n_taps; t > 0; --t) rem &= rem - 1;
    }
    f32x4 ra[2][PA];
    f32x4 rb[2][2][PB];
    auto load_next = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
        const int tap = __builtin_ctz(rem);               // wave-uniform
        const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
        const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 4;
        const int w_off = w_grp + tap * (BK * 2);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
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
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    load_next(I0{});                                      // chunk 0
    load_next(I1{});                                      // chunk 1: both in flight while the two scales are fetched

    // the two power-of-two scales (every wave derives the same numbers from the same words)
    h3_fp16_saturate();
    const float x_max = amax_read(p.x_amax);
    const int eA = h3_exponent(x_max);
    const int eB = h3_exponent(*reinterpret_cast<const float*>(wbase));
    const float sA = h3_pow2(eA);
    h3_check_record(p.x_amax, x_max);
    unsigned seen = 0u;

    auto store = [&](auto setc) {
        constexpr int S = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            f16x4 h, l;
            h3_split(ra[S][i], sA, h, l, seen);
            const int row = lrow + RPP * i, g = tid & 7;
            char* dst = As + row * X6_ROWB + 16 * ((g >> 1) ^ h3_swz(row)) + 8 * (g & 1);
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + BM * X6_ROWB) = l;
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int i = 0; i < PB; ++i)
                *reinterpret_cast<f32x4*>(Bs + pl * BN * X6_ROWB + (brow + RPB * i) * X6_ROWB + 16 * ((tid & 3) ^ h3_swz(brow + RPB * i))) = rb[S][pl][i];
    };

    f32x16 acc0[TM][TN], acc1[TM][TN];
    f32x4 s0[TM][TN][4], s1[TM][TN][4];                   // (S16) the same accumulators as sixteen-row blocks
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.0f; acc1[i][j][e] = 0.0f; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { s0[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; s1[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
        }

    const char* abase = As + (wm * TM * 32 + li) * X6_ROWB;
    const char* bbase = Bs + (wn * TN * 32 + li) * X6_ROWB;
    const int koff[2] = {16 * (lh ^ h3_swz(li)), 16 * ((2 + lh) ^ h3_swz(li))};       // k-step s reads logical slot 2 s + lh (tile bases are multiples of 32 rows)
    auto compute = [&]() {
        if constexpr (H3_S16) {
            h3_chunk<TM, TN>(As + wm * TM * 32 * X6_ROWB, BM * X6_ROWB, Bs + wn * TN * 32 * X6_ROWB, BN * X6_ROWB, lane, s0, s1);
            return;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {                     // two k-steps of 16 per 32-channel chunk
            f16x8 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const f16x8*>(abase + pl * BM * X6_ROWB + i * 32 * X6_ROWB + koff[s]);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const f16x8*>(bbase + pl * BN * X6_ROWB + j * 32 * X6_ROWB + koff[s]);
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
    };
    // chunk c waits in set c & 1.  Iteration: MFMAs of chunk kt from LDS | barrier | split + store chunk kt+1 (requested one
    // iteration earlier) and request chunk kt+3 into the registers just freed | barrier
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
    h3_report(p.x_amax, seen);
    if constexpr (H3_S16) { h3_gather<TM, TN>(s0, acc0); h3_gather<TM, TN>(s1, acc1); }
    h3_combine<TM, TN>(acc0, acc1);
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
                    f32x4 v = {acc0[i][j][4 * q], acc0[i][j][4 * q + 1], acc0[i][j][4 * q + 2], acc0[i][j][4 * q + 3]};
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
                for (int e = 0; e < 16; ++e) acc0[i][j][e] = 0.0f;
        for (int sl = 0; sl < splits; ++sl) {                  // slice order: two runs are bitwise equal
            const float4* sp = base + (size_t)sl * (BM * BN / 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = sp[((i * TN + j) * 4 + q) * NT + tid];
                        acc0[i][j][4 * q] += v.x; acc0[i][j][4 * q + 1] += v.y; acc0[i][j][4 * q + 2] += v.z; acc0[i][j][4 * q + 3] += v.w;
                    }
        }
        __syncthreads();                                       // the flag word is read; the epilogue reuses the buffer
    }
    h3_unscale<TM, TN>(acc0, h3_pow2(-eA), h3_pow2(-eB));
    if (p.vec_epi) x6_epilogue_vec<TM, TN, WM, WN, false, H3_S16>(acc0, p, m0, n0, tid, wm, wn, li, lh, reinterpret_cast<float*>(lds));
    else epilogue<TM, TN, H3_S16>(acc0, p, m0, n0, wm, wn, li, lh);
}

// Plane input, three operand stages, nothing staged through registers (round 6): the activations' and the filter's fp16 planes go from
// memory to LDS unchanged, so a wave instruction lands 16 rows x 64 B straight in LDS (`buffer_load ... lds`: lane-linear destination;
// the XOR swizzle of the 16-byte slots is applied to the SOURCE piece a lane asks for, and again by the fragment reads).  A workgroup
// keeps a RING of three stages: chunk t + 2 is requested as chunk t starts, `s_waitcnt vmcnt(n)` waits for the oldest chunk only (a
// chunk is PA + PB wave instructions per thread; requests complete in order), one barrier per chunk makes it visible and frees the
// stage chunk t - 1 was read from.  Against k_conv_igemm_h3_db<.., true>: no ds_write of the operands (a third of the loop's LDS
// traffic), no staging registers, two chunks of lead instead of one.  Same k order, same products: bit-identical results.  It is the
// loop of k_conv_igemm_h3_db<.., APLANES = true, RING = true>, the plane-reading kernel's second instantiation: long reductions take it
// (the head's 3x3: 144 chunks, 748 -> 712 us per four-image launch), the sixteen chunks of 512 -> 2048 run 3-5 % slower on it -- a fixed
// ~2.7 us per tile that neither the LDS request nor the place of the scale reads explains (scripts/dev/r6_ring_shortk.sh) -- and keep
// the double buffer.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void h3_ring_tile(const ConvArgs& p, char* lds) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN;
    constexpr int GA = 2 * BM / 16, GB = 2 * BN / 16;     // 16-row groups (one wave instruction each) of the two planes of A / of B
    constexpr int PA = GA / NW, PB = (GB + NW - 1) / NW, NL = PA + PB;
    constexpr int STAGE = 2 * (BM + BN) * X6_ROWB, NSTAGE = 3;
    static_assert(GA % NW == 0 && GB % NW == 0, "every wave issues the same number of requests per chunk (vmcnt counts them)");
    static_assert(H3_S16, "the 16x16x32 form");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    int tile_m, tile_n;
    h3_tile_of(p, xcd_remap(blockIdx.x, nwg), tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const size_t plane_bytes = (size_t)p.Cout * p.Kpad * 2;
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const size_t x_elems = (size_t)p.n_img * p.H * p.W * p.Cin;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x_planes), 0, (int)(x_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase + H3_HEADER_BYTES), 0, (int)(2 * plane_bytes), 0x00020000);

    h3_fp16_saturate();

    // lane -> (row of its 16-row group, LDS slot); the slot holds piece slot ^ swz(row) of the row's 64 bytes
    const int grow = lane >> 2, slot = lane & 3;
    int a_h[PA], a_w[PA], a_off[PA], a_dst[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int g = wave + NW * i, pl = g / (BM / 16), r0 = 16 * (g % (BM / 16));
        const int row = r0 + grow, m = m0 + row, piece = slot ^ h3_swz(row);
        a_dst[i] = pl * BM * X6_ROWB + r0 * X6_ROWB;
        if (m < p.M) {
            int wo, ho, img;
            if (p.layout) { img = m % p.n_img; const int pos = m / p.n_img; ho = pos / p.Wo; wo = pos - ho * p.Wo; }
            else { wo = m % p.Wo; const int t = m / p.Wo; ho = t % p.Ho; img = t / p.Ho; }
            a_h[i] = ho * p.stride - p.pad_top;
            a_w[i] = wo * p.stride - p.pad_left;
            a_off[i] = (img * p.img_stride + (a_h[i] * p.W + a_w[i]) * p.pix_stride + piece * 8) * 2 + (int)(pl * x_elems * 2);
        } else {
            a_h[i] = -(1 << 28); a_w[i] = 0; a_off[i] = 0;
        }
    }
    unsigned b_off[PB];
    int b_dst[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int g = wave + NW * i, pl = g / (BN / 16), r0 = 16 * (g % (BN / 16));
        const int row = r0 + grow, piece = slot ^ h3_swz(row);
        b_dst[i] = 2 * BM * X6_ROWB + pl * BN * X6_ROWB + r0 * X6_ROWB;
        b_off[i] = n0 + row < p.Cout ? (unsigned)(((size_t)(n0 + row) * p.Kpad + piece * 8) * 2 + pl * plane_bytes) : OOB_OFFSET;
    }

    const int RS = p.R * p.S;
    const unsigned tap_mask = h3_tap_mask(p, m0, BM);
    const int nk = (p.Kpad / (BK * RS)) * __popc(tap_mask);
    unsigned rem = tap_mask;
    int c0 = 0, w_grp = 0;
    unsigned a_v[PA];
    int w_v;
    auto prep = [&]() {                                   // the offsets of the NEXT chunk of this tile's sequence (tap, channel block, halo test)
        const int tap = __builtin_ctz(rem);
        const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
        const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 2;
        w_v = w_grp + tap * (BK * 2);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            a_v[i] = ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET;
        }
        rem &= rem - 1;
        const int wrap = (rem == 0);
        rem |= wrap ? tap_mask : 0u;
        c0 += wrap * BK;
        w_grp += wrap * (RS * BK * 2);
    };
    auto issue = [&](int stage) {
        char* base = lds + stage * STAGE;
#if defined(__HIP_DEVICE_COMPILE__)   // device pass only (conv_bf16.hip: the host pass drops an instantiation whose body casts to an LDS pointer in dependent code)
#pragma unroll
        for (int i = 0; i < PB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(base + b_dst[i]), 16, b_off[i], w_v, 0, 0);
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lds_ptr_t)(base + a_dst[i]), 16, a_v[i], 0, 0, 0);
#else
        (void)base; (void)xrsrc; (void)wrsrc; (void)w_v; (void)sizeof(lds_ptr_t);
#endif
    };
    f32x16 acc0[TM][TN], acc1[TM][TN];
    f32x4 s0[TM][TN][4], s1[TM][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) { s0[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; s1[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }

    prep(); issue(0);
    prep(); if (nk > 1) issue(1);
    // the scales, behind the first two chunks' requests (a short reduction -- 512 -> 2048 is sixteen chunks -- cannot afford their
    // latency in front); pinned here so that no load of theirs drifts into the loop, whose s_waitcnt counts the ring's requests
    const float x_max = amax_read(p.x_amax);
    const int eA = *p.x_pexp;
    const int eB = h3_exponent(*reinterpret_cast<const float*>(wbase));
    int eY = 0;
    if (p.y_planes) {
        eY = h3_exponent(p.bound_c * x_max + p.bound_d + (p.res_amax ? amax_read(p.res_amax) : 0.0f));
        if (blockIdx.x == 0 && tid == 0) *p.y_pexp = eY;
    }
    asm volatile("" :: "v"(eA), "v"(eB), "v"(eY) : "memory");
    prep();                                               // chunk 2
    int stage = 0, free_stage = 2;
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NL) : "memory");      // chunk t has landed (chunk t + 1 may be in flight)
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nk) issue(free_stage);                // into the stage chunk t - 1 was read from: every wave is past its reads
        const char* base = lds + stage * STAGE;
        h3_chunk<TM, TN>(base + wm * TM * 32 * X6_ROWB, BM * X6_ROWB, base + 2 * BM * X6_ROWB + wn * TN * 32 * X6_ROWB, BN * X6_ROWB, lane, s0, s1);
        prep();                                           // chunk t + 3, behind this chunk's MFMAs
        free_stage = stage;
        stage = stage == NSTAGE - 1 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // the epilogue reuses the stages
    h3_gather<TM, TN>(s0, acc0); h3_gather<TM, TN>(s1, acc1);
    h3_combine<TM, TN>(acc0, acc1);
    h3_unscale<TM, TN>(acc0, h3_pow2(-eA), h3_pow2(-eB));
    if (p.y_planes) x6_epilogue_vec<TM, TN, WM, WN, true, H3_S16>(acc0, p, m0, n0, tid, wm, wn, li, lh, reinterpret_cast<float*>(lds), h3_pow2(eY));
    else if (p.vec_epi) x6_epilogue_vec<TM, TN, WM, WN, false, H3_S16>(acc0, p, m0, n0, tid, wm, wn, li, lh, reinterpret_cast<float*>(lds));
    else epilogue<TM, TN, H3_S16>(acc0, p, m0, n0, wm, wn, li, lh);
}

// ---- ONE workgroup per CU, TWO LDS buffers (96 KB for a 256x128 tile) and ONE barrier per chunk: chunk kt multiplies from buffer
// kt & 1 while chunk kt+1 (in registers since the previous iteration) is split and stored into the other buffer and chunk kt+2 is
// requested (k_conv_igemm_x6_db's loop).  <2,1,4,4>: sixteen waves of 64x32; <2,2,4,2>: eight waves of 64x64 (eight fragment reads
// per twelve MFMAs instead of six per six).  Lab (scripts/micro/h3_lab.hip, the head's 3x3 / 512->2048 / 2048->512 GEMMs alone on
// the chip): 246 / 127 / 104 us and 242 / 118 / 107 us against 345 / 172 / 160 for k_conv_igemm_x6_db on the same box.
// APLANES: the activations arrive ALREADY split -- two fp16 planes [2][rows][Cin] a producing launch's epilogue wrote under the scale
// 2^*x_pexp (ConvArgs.x_planes): 16-byte pieces go from memory to LDS unchanged, no conversion and no arithmetic in the loader (lab:
// the head's 3x3 / 512 -> 2048 / 2048 -> 512 GEMMs 195 / 103 / 89 us against 246 / 127 / 104 with the split in the loader).
template <int TM, int TN, int WM, int WN, bool APLANES = false, int RING = 0>
__global__ void __launch_bounds__(64 * WM * WN) k_conv_igemm_h3_db(const ConvArgs p) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
    constexpr int RPP = APLANES ? NT / 4 : NT / 8, PA = APLANES ? (2 * BM) / RPP : BM / RPP;      // A: rows staged per pass (planes: 4 pieces of 16 B per row per plane)
    constexpr int NBP = 2 * BN * 4;                       // 16-byte pieces of the two filter planes per chunk
    constexpr int PBT = (NBP + NT - 1) / NT;
    constexpr int BUFB = 2 * (BM + BN) * X6_ROWB;
    static_assert(BM % RPP == 0 && PA >= 1, "tile rows must be a multiple of the staging pass");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if constexpr (RING) {                                 // (an instantiation of its own: as a run-time branch beside the other loop the ring keeps a
        static_assert(APLANES, "the ring reads planes");  //  quarter of its gain -- 729 -> 717-723 instead of 748 -> 703-713 us -- to the shared register file)
        h3_ring_tile<TM, TN, WM, WN>(p, lds);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    int tile_m, tile_n;
    h3_tile_of(p, xcd_remap(blockIdx.x, nwg), tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const size_t plane_bytes = (size_t)p.Cout * p.Kpad * 2;
    const char* wbase = reinterpret_cast<const char*>(p.w);
    const size_t x_elems = (size_t)p.n_img * p.H * p.W * p.Cin;
    const __amdgpu_buffer_rsrc_t xrsrc = APLANES
        ? __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x_planes), 0, (int)(x_elems * 4), 0x00020000)      // two planes of 2 bytes
        : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)(x_elems * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase + H3_HEADER_BYTES), 0, (int)(2 * plane_bytes), 0x00020000);

    // f32 activations: a lane stages 4 channels (16 bytes of f32) of one row, 8 lanes per row; planes: 8 channels (16 bytes of fp16) of
    // one row of one plane, 4 lanes per row, the second half of the passes takes the low plane
    constexpr int ESZ = APLANES ? 2 : 4;
    const int lrow = APLANES ? tid >> 2 : tid >> 3, lcol = APLANES ? (tid & 3) * 8 : (tid & 7) * 4;
    int a_h[PA], a_w[PA], a_off[PA], a_lds[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pl = APLANES ? (lrow + RPP * i) / BM : 0;
        const int row = lrow + RPP * i - pl * BM, m = m0 + row;
        if constexpr (APLANES) a_lds[i] = pl * BM * X6_ROWB + row * X6_ROWB + 16 * ((tid & 3) ^ h3_swz(row));
        else { const int g = tid & 7; a_lds[i] = row * X6_ROWB + 16 * ((g >> 1) ^ h3_swz(row)) + 8 * (g & 1); }
        if (m < p.M) {
            int wo, ho, img;
            if (p.layout) { img = m % p.n_img; const int pos = m / p.n_img; ho = pos / p.Wo; wo = pos - ho * p.Wo; }
            else { wo = m % p.Wo; const int t = m / p.Wo; ho = t % p.Ho; img = t / p.Ho; }
            a_h[i] = ho * p.stride - p.pad_top;
            a_w[i] = wo * p.stride - p.pad_left;
            a_off[i] = (img * p.img_stride + (a_h[i] * p.W + a_w[i]) * p.pix_stride + lcol) * ESZ + (int)(pl * x_elems * 2);
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
        b_lds[i] = q < NBP ? 2 * BM * X6_ROWB + pl * BN * X6_ROWB + row * X6_ROWB + 16 * (g ^ h3_swz(row)) : -1;
    }

    const int RS = p.R * p.S;
    const unsigned tap_mask = h3_tap_mask(p, m0, BM);
    const int n_taps = __popc(tap_mask);
    const int nk = (p.Kpad / (BK * RS)) * n_taps;
    unsigned rem = tap_mask;
    int c0 = 0, w_grp = 0;
    f32x4 ra[PA], rb[PBT];
    auto load_into = [&](f32x4 (&ra)[PA], f32x4 (&rb)[PBT]) {
        const int tap = __builtin_ctz(rem);
        const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
        const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * ESZ;
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
    auto load_next = [&]() { load_into(ra, rb); };
    load_next();                                          // chunk 0: in flight while the two scales are fetched
    h3_fp16_saturate();
    const float x_max = amax_read(p.x_amax);
    const int eA = APLANES ? *p.x_pexp : h3_exponent(x_max);
    const int eB = h3_exponent(*reinterpret_cast<const float*>(wbase));
    const float sA = h3_pow2(eA);
    if constexpr (!APLANES) h3_check_record(p.x_amax, x_max);
    unsigned seen = 0u;
    // the output's planes: |y| <= bound_c * max|x| + bound_d (+ max|residual|) whatever the data, so that bound's exponent cannot overflow
    int eY = 0;
    if (p.y_planes) {
        eY = h3_exponent(p.bound_c * x_max + p.bound_d + (p.res_amax ? amax_read(p.res_amax) : 0.0f));
        if (blockIdx.x == 0 && tid == 0) *p.y_pexp = eY;
    }
    auto store_from = [&](int buf, f32x4 (&ra)[PA], f32x4 (&rb)[PBT]) {
        char* base = lds + buf * BUFB;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            if constexpr (APLANES) {
                *reinterpret_cast<f32x4*>(base + a_lds[i]) = ra[i];
            } else {
                f16x4 h, l;
                h3_split(ra[i], sA, h, l, seen);
                *reinterpret_cast<f16x4*>(base + a_lds[i]) = h;
                *reinterpret_cast<f16x4*>(base + BM * X6_ROWB + a_lds[i]) = l;
            }
        }
#pragma unroll
        for (int i = 0; i < PBT; ++i)
            if (b_lds[i] >= 0) *reinterpret_cast<f32x4*>(base + b_lds[i]) = rb[i];
    };
    auto store = [&](int buf) { store_from(buf, ra, rb); };
    f32x16 acc0[TM][TN], acc1[TM][TN];
    f32x4 s0[TM][TN][4], s1[TM][TN][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc0[i][j][e] = 0.0f; acc1[i][j][e] = 0.0f; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { s0[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; s1[i][j][q] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
        }
    const int aoff = (wm * TM * 32 + li) * X6_ROWB, boff = 2 * BM * X6_ROWB + (wn * TN * 32 + li) * X6_ROWB;
    const int koff[2] = {16 * (lh ^ h3_swz(li)), 16 * ((2 + lh) ^ h3_swz(li))};
    auto kstep = [&](int buf, int s) {
        const char* base = lds + buf * BUFB;
        f16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const f16x8*>(base + aoff + pl * BM * X6_ROWB + i * 32 * X6_ROWB + koff[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const f16x8*>(base + boff + pl * BN * X6_ROWB + j * 32 * X6_ROWB + koff[s]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i], fb[0][j], acc1[i][j], 0, 0, 0);
                acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[1][j], acc1[i][j], 0, 0, 0);
                acc0[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i], fb[0][j], acc0[i][j], 0, 0, 0);
            }
    };
    auto compute = [&](int buf) {
        if constexpr (H3_S16) {
            const char* base = lds + buf * BUFB;
            h3_chunk<TM, TN>(base + wm * TM * 32 * X6_ROWB, BM * X6_ROWB, base + 2 * BM * X6_ROWB + wn * TN * 32 * X6_ROWB, BN * X6_ROWB, lane, s0, s1);
        } else {
            kstep(buf, 0);
            kstep(buf, 1);
        }
    };
    store(0);
    load_next();                                          // chunk 1 (past-the-end fetches are never multiplied)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) { store(buf ^ 1); load_next(); }
        compute(buf);
        __syncthreads();
    }
    if constexpr (!APLANES) h3_report(p.x_amax, seen);
    if constexpr (H3_S16) { h3_gather<TM, TN>(s0, acc0); h3_gather<TM, TN>(s1, acc1); }
    h3_combine<TM, TN>(acc0, acc1);
    h3_unscale<TM, TN>(acc0, h3_pow2(-eA), h3_pow2(-eB));
    if (p.y_planes) x6_epilogue_vec<TM, TN, WM, WN, true, H3_S16>(acc0, p, m0, n0, tid, wm, wn, li, lh, reinterpret_cast<float*>(lds), h3_pow2(eY));
    else if (p.vec_epi) x6_epilogue_vec<TM, TN, WM, WN, false, H3_S16>(acc0, p, m0, n0, tid, wm, wn, li, lh, reinterpret_cast<float*>(lds));
    else epilogue<TM, TN, H3_S16>(acc0, p, m0, n0, wm, wn, li, lh);
}

// dev knob: FRCNN_H3_RING=0 keeps plane-input launches on the register-staged double buffer (k_conv_igemm_h3_db<.., true>)
static const bool g_h3_ring = !(getenv("FRCNN_H3_RING") && atoi(getenv("FRCNN_H3_RING")) == 0);
static const int g_h3_ring_min_chunks = getenv("FRCNN_H3_RING_MIN_CHUNKS") ? atoi(getenv("FRCNN_H3_RING_MIN_CHUNKS")) : 32;

template <int TM, int TN, int WM, int WN>
static int launch_h3_db(const ConvArgs& a, hipStream_t s, bool ring_ok = true) {
    ConvArgs p = a;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    constexpr size_t lds = (size_t)2 * 2 * (BM + BN) * X6_ROWB;
    static_assert(lds >= X6Tile<TM, TN, WM, WN, 2>::epi, "the epilogue's wave-row must fit in the operand buffers");
    static std::atomic<uint64_t> lds_seen{0}, lds_seen_planes{0};
    if (p.x_planes) {
        if constexpr (WM * WN == 16) {
            // long reductions only (the head's 3x3: 144 chunks); FRCNN_H3_RING=0 / FRCNN_H3_RING_MIN_CHUNKS are dev knobs
            if (ring_ok && g_h3_ring && p.Kpad / BK >= g_h3_ring_min_chunks) {
                constexpr size_t ring_lds = (size_t)3 * 2 * (BM + BN) * X6_ROWB;
                static std::atomic<uint64_t> lds_seen_ring{0};
                if (int e = raise_lds_once(lds_seen_ring, (const void*)k_conv_igemm_h3_db<TM, TN, WM, WN, true, 1>, ring_lds, "conv2d_h3")) return e;
                k_conv_igemm_h3_db<TM, TN, WM, WN, true, 1><<<p.tiles_m * p.tiles_n, 64 * WM * WN, ring_lds, s>>>(p);
                return check_launch("conv2d_fwd_h3 (planes in, ring)");
            }
        }
        if (int e = raise_lds_once(lds_seen_planes, (const void*)k_conv_igemm_h3_db<TM, TN, WM, WN, true>, lds, "conv2d_h3")) return e;
        k_conv_igemm_h3_db<TM, TN, WM, WN, true><<<p.tiles_m * p.tiles_n, 64 * WM * WN, lds, s>>>(p);
        return check_launch("conv2d_fwd_h3 (planes in)");
    }
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_h3_db<TM, TN, WM, WN>, lds, "conv2d_h3")) return e;
    k_conv_igemm_h3_db<TM, TN, WM, WN><<<p.tiles_m * p.tiles_n, 64 * WM * WN, lds, s>>>(p);
    return check_launch("conv2d_fwd_h3");
}

template <int TM, int TN, int WM, int WN, bool SPLITK = false>
static int launch_h3(const ConvArgs& a, hipStream_t s) {
    using T = X6Tile<TM, TN, WM, WN, 2>;
    ConvArgs p = a;
    p.tiles_m = (p.M + T::BM - 1) / T::BM;
    p.tiles_n = (p.Cout + T::BN - 1) / T::BN;
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_h3<TM, TN, WM, WN, SPLITK>, T::lds, "conv2d_h3")) return e;
    k_conv_igemm_h3<TM, TN, WM, WN, SPLITK><<<p.tiles_m * p.tiles_n * (SPLITK ? p.splits : 1), T::NT, T::lds, s>>>(p);
    return check_launch(SPLITK ? "conv2d_fwd_h3 (split-K)" : "conv2d_fwd_h3");
}

// tile codes of the f16x3 engine (frcnn_conv_desc.tile; 0 / 50 = auto)
int launch_conv_h3(const ConvArgs& a, int cfg, hipStream_t s) {
    switch (cfg) {
        case 81: return launch_h3<2, 1, 2, 4>(a, s);          // 128x128, 8 waves (64x32 per wave)
        case 82: return launch_h3_db<2, 2, 4, 2>(a, s);       // 256x128, 8 waves (64x64 per wave), two LDS buffers, one workgroup per CU
        case 83: return launch_h3<2, 2, 2, 2>(a, s);          // 128x128, 4 waves (64x64 per wave)
        case 84: return launch_h3<1, 1, 2, 2>(a, s);          // 64x64, 4 waves
        case 85: return launch_h3_db<2, 1, 4, 4>(a, s, false);  // 86 with plane input kept on the two register-staged buffers whatever the reduction's length
        case 86: return launch_h3_db<2, 1, 4, 4>(a, s);       // 256x128, 16 waves, two LDS buffers (plane input with a long reduction: the three-stage ring), one workgroup per CU
        case 87: return launch_h3<2, 1, 2, 2>(a, s);          // 128x64, 4 waves (64x32 per wave): the 64-column layers
        case 184: return launch_h3<1, 1, 2, 2, true>(a, s);   // 64x64 with split-K (a.splits / a.slabs / a.tickets set by the caller)
        case 181: return launch_h3<2, 1, 2, 4, true>(a, s);   // 128x128 on eight waves with split-K: the taller small grids
        default: return fail(FRCNN_E_ARG, "conv2d_fwd_h3: unknown tile config %d", cfg);
    }
}

int h3_tile_width(int cfg) { return (cfg == 84 || cfg == 87) ? 64 : 128; }

// ---- packing: max|w| of the packed filter into the header word, then the two planes under that scale
__global__ void __launch_bounds__(256) k_h3_wmax(const float* w, size_t n, unsigned* header) {
    float v = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v = fmaxf(v, fabsf(w[i]));
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0 && v > 0.0f) atomicMax(header, __float_as_uint(v));
}

__global__ void __launch_bounds__(256) k_pack_h3(const float* w, size_t n, const float* header, _Float16* out) {
    const float s = h3_pow2(h3_exponent(*header));
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * blockDim.x * 4) {       // n % 4 == 0
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + i);
        f16x4 h, l;
        unsigned unused = 0u;                                 // (the scale comes from the maximum of these very values)
        h3_split(v, s, h, l, unused);
        *reinterpret_cast<f16x4*>(out + i) = h;
        *reinterpret_cast<f16x4*>(out + n + i) = l;
    }
}

// the same for MANY filters (a training step re-derives the planes of every trainable layer's forward and input-gradient filter after the
// optimiser has rewritten the f32 forms; conv_x6.hip k_pack_x6_batch): a workgroup finds its job by the block prefix.  Three launches
// per table: headers to zero, max|w| per filter, the two planes under that maximum's scale.
constexpr int H3_REFRESH_JOBS = 48;
struct H3RefreshTable { const float* w[H3_REFRESH_JOBS]; char* out[H3_REFRESH_JOBS]; unsigned long long n[H3_REFRESH_JOBS]; int first_block[H3_REFRESH_JOBS + 1]; int jobs; };
__global__ void __launch_bounds__(256) k_h3_refresh_clear(const H3RefreshTable t) {
    const int j = threadIdx.x >> 2;
    if (j < t.jobs) reinterpret_cast<float*>(t.out[j])[threadIdx.x & 3] = 0.0f;
}
__global__ void __launch_bounds__(256) k_h3_refresh_wmax(const H3RefreshTable t) {
    int j = 0;
    while (j + 1 < t.jobs && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const size_t n = t.n[j];
    const int nb = t.first_block[j + 1] - t.first_block[j], b = (int)blockIdx.x - t.first_block[j];
    const float* w = t.w[j];
    float v = 0.0f;
    for (size_t i = ((size_t)b * 256 + threadIdx.x) * 4; i < n; i += (size_t)nb * 256 * 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(w + i);
        v = fmaxf(fmaxf(v, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3])));
    }
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0 && v > 0.0f) atomicMax(reinterpret_cast<unsigned*>(t.out[j]), __float_as_uint(v));
}
__global__ void __launch_bounds__(256) k_h3_refresh_pack(const H3RefreshTable t) {
    int j = 0;
    while (j + 1 < t.jobs && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const size_t n = t.n[j];
    const int nb = t.first_block[j + 1] - t.first_block[j], b = (int)blockIdx.x - t.first_block[j];
    const float* w = t.w[j];
    const float s = h3_pow2(h3_exponent(*reinterpret_cast<const float*>(t.out[j])));
    _Float16* out = reinterpret_cast<_Float16*>(t.out[j] + H3_HEADER_BYTES);
    for (size_t i = ((size_t)b * 256 + threadIdx.x) * 4; i < n; i += (size_t)nb * 256 * 4) {       // n % 4 == 0 (packed k is a multiple of 32)
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + i);
        f16x4 h, l;
        unsigned unused = 0u;                                 // (the scale comes from the maximum of these very values)
        h3_split(v, s, h, l, unused);
        *reinterpret_cast<f16x4*>(out + i) = h;
        *reinterpret_cast<f16x4*>(out + n + i) = l;
    }
}

// ---- magnitude records
__global__ void __launch_bounds__(256) k_amax_clear(float* rec, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) rec[i] = 0.0f;
}

__global__ void __launch_bounds__(256) k_amax_f32(const float* x, size_t n, float* rec) {
    float v = 0.0f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(x + 4 * i);
        v = fmaxf(fmaxf(v, fmaxf(fabsf(q[0]), fabsf(q[1]))), fmaxf(fabsf(q[2]), fabsf(q[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) v = fmaxf(v, fabsf(x[4 * n4 + threadIdx.x]));
    amax_publish(rec, v);
}

// dst := max(dst, max over src's slots, floor): the bound of a tensor made from `src`'s tensor by a map that cannot exceed
// max(|input|, floor) -- max-pooling, the bilinear RoI resampling with a fill vector, ReLU
__global__ void __launch_bounds__(64) k_amax_merge(float* dst, const float* src, float floor_value, int* exponent_out) {
    const float v = fmaxf(fmaxf(src ? amax_read(src) : 0.0f, floor_value), amax_read(dst));
    if (threadIdx.x == 0) {
        if (v > 0.0f) atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(v));
        if (exponent_out) *exponent_out = h3_exponent(v);       // the scale of planes written under this bound (frcnn_roi_crop_resize_fwd_planes)
    }
}

// OR of the status words (word 1) of `n` records -> *out (one workgroup: a pass has a few hundred records)
__global__ void __launch_bounds__(256) k_amax_status(const float* rec, int n, int* out) {
    __shared__ unsigned part[4];
    unsigned bits = 0;
    for (int r = threadIdx.x; r < n; r += 256) bits |= reinterpret_cast<const unsigned*>(rec + (size_t)r * AMAX_SLOTS * AMAX_STRIDE)[1];
#pragma unroll
    for (int o = 32; o; o >>= 1) bits |= __shfl_xor(bits, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) *out = (int)(part[0] | part[1] | part[2] | part[3]);
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_amax_status(const float* records, int n_records, int32_t* out, void* stream) {
    if (!records || n_records <= 0 || !out) return fail(FRCNN_E_ARG, "amax_status: bad argument");
    k_amax_status<<<1, 256, 0, as_stream(stream)>>>(records, n_records, out);
    return check_launch("amax_status");
}

extern "C" size_t frcnn_conv_h3_planes_bytes(int cout, int packed_k) {
    if (cout <= 0 || packed_k <= 0) return 0;
    return (size_t)H3_HEADER_BYTES + (size_t)4 * cout * packed_k;
}

extern "C" int frcnn_pack_conv_weights_h3(const float* w_packed, int cout, int kpad, void* planes_f16, void* stream) {
    if (!w_packed || !planes_f16 || cout <= 0 || kpad <= 0 || (kpad % 32)) return fail(FRCNN_E_ARG, "pack_conv_weights_h3: bad argument");
    if ((reinterpret_cast<uintptr_t>(w_packed) & 15) || (reinterpret_cast<uintptr_t>(planes_f16) & 15))
        return fail(FRCNN_E_ARG, "pack_conv_weights_h3: 16-byte aligned buffers required");
    const size_t n = (size_t)cout * kpad;
    if (2 * n * 2 >= 0x7fffffffull) return fail(FRCNN_E_UNSUPPORTED, "pack_conv_weights_h3: filter planes over 2 GiB");
    hipStream_t s = as_stream(stream);
    int grid = (int)((n + 1023) / 1024);
    if (grid > 2048) grid = 2048;
    k_amax_clear<<<1, 64, 0, s>>>(reinterpret_cast<float*>(planes_f16), H3_HEADER_BYTES / 4);
    k_h3_wmax<<<grid, 256, 0, s>>>(w_packed, n, reinterpret_cast<unsigned*>(planes_f16));
    k_pack_h3<<<grid, 256, 0, s>>>(w_packed, n, reinterpret_cast<const float*>(planes_f16),
                                   reinterpret_cast<_Float16*>(reinterpret_cast<char*>(planes_f16) + H3_HEADER_BYTES));
    return check_launch("pack_conv_weights_h3");
}

extern "C" int frcnn_refresh_h3_planes(const frcnn_x6_job* jobs, int n_jobs, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(FRCNN_E_ARG, "refresh_h3_planes: bad argument");
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs[i].w_packed || !jobs[i].planes_bf16 || jobs[i].rows <= 0 || jobs[i].kpad <= 0 || (jobs[i].kpad % 32)
            || (reinterpret_cast<uintptr_t>(jobs[i].w_packed) & 15) || (reinterpret_cast<uintptr_t>(jobs[i].planes_bf16) & 15))
            return fail(FRCNN_E_ARG, "refresh_h3_planes: job %d is malformed (packed k must be a multiple of 32, 16-byte aligned buffers)", i);
    hipStream_t s = as_stream(stream);
    for (int b = 0; b < n_jobs; b += H3_REFRESH_JOBS) {
        H3RefreshTable t;
        const int n = n_jobs - b < H3_REFRESH_JOBS ? n_jobs - b : H3_REFRESH_JOBS;
        int blocks = 0;
        for (int i = 0; i < H3_REFRESH_JOBS; ++i) {
            const frcnn_x6_job& j = jobs[b + (i < n ? i : 0)];
            t.w[i] = j.w_packed; t.out[i] = reinterpret_cast<char*>(j.planes_bf16); t.n[i] = (unsigned long long)j.rows * j.kpad;
            t.first_block[i] = blocks;
            if (i < n) { const unsigned long long g = (t.n[i] + 4095) / 4096; blocks += (int)(g < 1 ? 1 : (g > 512 ? 512 : g)); }      // 16 elements per thread
        }
        t.first_block[H3_REFRESH_JOBS] = blocks;
        t.jobs = n;
        k_h3_refresh_clear<<<1, 256, 0, s>>>(t);
        k_h3_refresh_wmax<<<blocks, 256, 0, s>>>(t);
        k_h3_refresh_pack<<<blocks, 256, 0, s>>>(t);
        if (int e = check_launch("refresh_h3_planes")) return e;
    }
    return FRCNN_OK;
}

extern "C" int frcnn_amax_record_floats(void) { return AMAX_SLOTS * AMAX_STRIDE; }

extern "C" int frcnn_amax_clear(float* records, int n_records, void* stream) {
    if (!records || n_records <= 0) return fail(FRCNN_E_ARG, "amax_clear: bad argument");
    const size_t n = (size_t)n_records * AMAX_SLOTS * AMAX_STRIDE;
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    k_amax_clear<<<grid, 256, 0, as_stream(stream)>>>(records, n);
    return check_launch("amax_clear");
}

extern "C" int frcnn_amax_f32(const float* x, size_t n, float* record, void* stream) {
    if (!x || !record || n == 0) return fail(FRCNN_E_ARG, "amax_f32: bad argument");
    if (reinterpret_cast<uintptr_t>(x) & 15) return fail(FRCNN_E_ARG, "amax_f32: 16-byte aligned tensor required");
    int grid = (int)((n / 4 + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    k_amax_f32<<<grid, 256, 0, as_stream(stream)>>>(x, n, record);
    return check_launch("amax_f32");
}

extern "C" int frcnn_amax_merge(float* dst_record, const float* src_record, float floor_value, int32_t* exponent_out, void* stream) {
    if (!dst_record || !(floor_value >= 0.0f)) return fail(FRCNN_E_ARG, "amax_merge: bad argument");
    k_amax_merge<<<1, 64, 0, as_stream(stream)>>>(dst_record, src_record, floor_value, exponent_out);
    return check_launch("amax_merge");
}
