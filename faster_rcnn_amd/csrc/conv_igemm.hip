// Implicit-GEMM NHWC convolution on the CDNA4 matrix cores (gfx950), f32 in / f32 accumulate.
//
// Replaces Keras Conv2D (+bias) / BatchNormalization(training=False) / Scale / Activation /
// add in resnet.py:150-176, 218-247, 408-412, 464-474, 508-533 and vgg.py:96-137, 172-185,
// 233-247 (Dense = 1x1 conv on a 1x1 map).
//
// GEMM view:  Y[m][n] = sum_k A[m][k] * Wt[n][k]
//   m = (img, ho, wo) output pixel, n = output channel, k = (channel chunk, filter tap (r,s), channel in chunk).
//   A is never materialised (no im2col): for one 32-wide k-chunk inside a single filter tap
//   (r,s) the A row of pixel m is the 128 contiguous bytes x[img][ho*st+r-pt][wo*st+s-pl][c0..c0+32)
//   of the NHWC input, or zeros in the padding halo.
//   Wt is the filter pre-packed to [Cout][Kpad] (k contiguous), so both operands are
//   "row-major, k contiguous" and share one LDS image + one fragment-read pattern.
//
// Workgroup = 256 threads = 4 wave64 in a 2x2 arrangement; each wave owns TM x TN tiles of
// 32x32 outputs (v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
// 16 accumulator VGPRs per tile).  One ds_read_b128 per operand tile yields FOUR mfma k-steps
// (lane half h reads k = 8*kk + 4h .. +3; step j multiplies k=8kk+j (h=0) and 8kk+4+j (h=1)),
// so a 32-deep chunk costs (TM+TN)*4 LDS reads for TM*TN*16 MFMAs.  LDS rows are padded to 36
// floats (144 B): the 16-lane ds_read_b128 groups then hit 16 distinct 4-bank slots.
// Global->LDS staging is register double-buffered: loads for chunk t+1 are issued before
// the MFMAs of chunk t and written to the other LDS buffer after them; one barrier per chunk.
// The blockIdx -> tile map is XCD-aware (8 XCDs round-robin on blockIdx): each XCD walks a
// contiguous run of m-tiles of ONE n-tile so that n-tile's filter slice stays in its 4 MB L2.
//
// Epilogue (fused, in registers): v = acc*scale[n] + shift[n] (folded bias+BN(+Scale)),
// + residual[m][n], activation (none / relu / sigmoid), store NHWC.
#include "conv_f32_common.h"

namespace frcnn {

template <int TM, int TN, bool GENERIC_A>
__global__ void __launch_bounds__(256) k_conv_igemm_f32(const ConvArgs p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int PA = BM / 32, PB = BN / 32;          // float4 rows staged per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [2][BM][LDS_STRIDE]
    float* Bs = smem + 2 * BM * LDS_STRIDE;             // [2][BN][LDS_STRIDE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;

    const int nwg = p.tiles_m * p.tiles_n;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tile_n = logical / p.tiles_m, tile_m = logical - tile_n * p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread staging coordinates
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    int a_h[PA], a_w[PA];
    size_t a_img[PA];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = m0 + lrow + 32 * i;
        if (m < p.M) {
            const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, img = t / p.Ho;
            a_h[i] = ho * p.stride - p.pad_top;
            a_w[i] = wo * p.stride - p.pad_left;
            a_img[i] = (size_t)img * p.H * p.W * p.Cin;
        } else {
            a_h[i] = -(1 << 28); a_w[i] = 0; a_img[i] = 0;
        }
    }
    const float* b_ptr[PB];
    bool b_ok[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + lrow + 32 * i;
        b_ok[i] = n < p.Cout;
        b_ptr[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.Kpad + lcol;
    }

    f32x4 ra[PA], rb[PB];
    int r_tap = 0, s_tap = 0, c0 = 0;                   // filter tap / channel offset of the NEXT chunk to load

    auto load_chunk = [&](int kc) {
#pragma unroll
        for (int i = 0; i < PB; ++i)
            rb[i] = b_ok[i] ? *reinterpret_cast<const f32x4*>(b_ptr[i] + kc * BK) : f32x4{0, 0, 0, 0};
        if constexpr (!GENERIC_A) {
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const float* src = p.x + a_img[i] + ((size_t)hi * p.W + wi) * p.Cin + c0 + lcol;
                ra[i] = ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0, 0, 0, 0};
            }
            // k order = [channel chunk][tap][32 channels]: consecutive chunks re-read the SAME 128-byte
            // channel slice of neighbouring pixels, so the 9 taps of a 3x3 hit L1/L2 instead of streaming
            // the whole input tile 9 times (measured: 46 % L2 hit rate and ~1 GB fetched per launch before)
            if (++s_tap == p.S) { s_tap = 0; if (++r_tap == p.R) { r_tap = 0; c0 += BK; } }
        } else {
            // small-Cin path (stem, Cin=3): decode every k separately
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                f32x4 v = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kc * BK + lcol + e;
                    if (k < p.K) {
                        const int c = k % p.Cin, rs = k / p.Cin, s = rs % p.S, r = rs / p.S;
                        const int hi = a_h[i] + r, wi = a_w[i] + s;
                        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                            v[e] = p.x[a_img[i] + ((size_t)hi * p.W + wi) * p.Cin + c];
                    }
                }
                ra[i] = v;
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* a = As + buf * BM * LDS_STRIDE;
        float* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<f32x4*>(a + (lrow + 32 * i) * LDS_STRIDE + lcol) = ra[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<f32x4*>(b + (lrow + 32 * i) * LDS_STRIDE + lcol) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.Kpad / BK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) load_chunk(kc + 1);
        const float* a = As + buf * BM * LDS_STRIDE + (wm * TM * 32 + li) * LDS_STRIDE + lh * 4;
        const float* b = Bs + buf * BN * LDS_STRIDE + (wn * TN * 32 + li) * LDS_STRIDE + lh * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    epilogue<TM, TN>(acc, p, m0, n0, wm, wn, li, lh);
}

// ------------------------------------------------------------------------------------
// v2 main loop: branch-free and software-pipelined inside ONE wave.
//   * operands come through buffer loads (SRD + 32-bit offsets): padding halo, m >= M and
//     n >= Cout rows simply carry an out-of-range offset and read back zeros, so the loop
//     body is a single basic block the scheduler may interleave freely;
//   * sched_group_barrier pins the interleave: the next chunk's 8 global loads ride behind
//     the first MFMAs, each kk-step's fragment reads behind the previous step's MFMAs and
//     the LDS stores behind the last MFMAs.  A 32x32x2 f32 MFMA occupies the matrix pipe
//     for 64 cycles but the wave's issue port only briefly, so those VALU/VMEM/DS
//     instructions issue in the shadow of the wave's own MFMAs instead of in a separate
//     phase (v1: matrix pipe 79 % busy on a 2-wave SIMD, both waves stalling in lockstep).
// Lab builds (scripts/micro/conv_lab.hip) compile this file with FRCNN_LAB_STAMPS: every workgroup then records
// the 100 MHz wall clock at its phase boundaries.  The product library never defines it.
#ifdef FRCNN_LAB_STAMPS
__device__ unsigned long long* g_lab_stamps = nullptr;
#define LAB_STAMP(i) do { if (g_lab_stamps && threadIdx.x == 0) g_lab_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LAB_STAMP(i) do { } while (0)
#endif
//
// SPLITK: small grids (stage 4, the RPN heads, the dense layers: <= 152 tiles for 256 CUs and a long k loop)
// cut K into `splits` slices, one workgroup each.  Every slice writes its f32 partial tile to a slab, publishes
// it with ONE agent-scope release and draws a ticket; the workgroup that draws the last ticket acquires, sums
// the slabs in slice order (fixed order: two runs are bitwise equal) and runs the fused epilogue.  One launch,
// no atomics on data (cdna guide s5 "in-launch split-K reduction").
//
// CIN3: the 3-channel stems (ResNet conv1 7x7, VGG block1_conv1 3x3).  The filter is packed as if the image had FOUR
// channels (k = tap*4 + c, c == 3 a zero column, see packed_k), so a 32-wide chunk is eight taps and each of the eight
// lanes that stage a row fetches ITS tap of its pixel with one 12-byte buffer load (per-lane r, s instead of the
// wave-uniform tap walk) and appends a zero.  Taps beyond R*S load nothing.  Replaces the per-element gather of the
// v1 kernel for these layers.
template <int TM, int TN, int VARIANT = 0, int WM = 2, int WN = 2, bool SPLITK = false, bool CIN3 = false>
__global__ void __launch_bounds__(64 * WM * WN) k_conv_igemm_f32_v2(const ConvArgs p) {
    // WM x WN waves, each owning TM x TN 32x32 tiles.  2x2 waves (256 threads) is the base shape; 4x2 waves
    // (512 threads) on the same 128x128 tile halves the registers per wave so FOUR waves share a SIMD
    // instead of two and a wave's barrier / LDS-latency gaps are covered by three partners.
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int RPP = NT / 8;                          // tile rows staged per pass (8 lanes x 16 B per row)
    constexpr int PA = BM / RPP, PB = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    LAB_STAMP(0);

    const int splits = SPLITK ? p.splits : 1;
    const int nwg = p.tiles_m * p.tiles_n * splits;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int tile = SPLITK ? logical / splits : logical;          // a tile's slices are neighbours on one XCD
    const int slice = SPLITK ? logical - tile * splits : 0;
    int tile_n = tile / p.tiles_m, tile_m = tile - tile_n * p.tiles_m;
    if (p.group_m > 0) {
        // Grouped order.  An XCD's workgroups hold a contiguous run of ids (xcd_remap), ~128 tiles at a time.  In the plain
        // order those are 128 ROW tiles of one column tile: the filter slice is shared, every A tile is its own
        // (1x1 512->2048 on the head's 14 700 rows: 128 x 128 KB = 16 MB live per 4 MB L2, A streamed from beyond L2 once
        // per column tile, ~0.9 GB per launch).  Grouped, the run is g row tiles x 128/g column tiles: (g + 128/g) tiles of
        // operands live, each fetched once per group.
        const int per = p.group_m * p.tiles_n, g = tile / per, m_base = g * p.group_m;
        const int gm = min(p.group_m, p.tiles_m - m_base), r = tile - g * per;
        tile_n = r / gm;
        tile_m = m_base + r - tile_n * gm;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 4), 0x00020000);

    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
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
            a_off[i] = (img * p.img_stride + (a_h[i] * p.W + a_w[i]) * p.pix_stride + (CIN3 ? 0 : lcol)) * 4;     // may be "negative" in the halo
        } else {
            a_h[i] = -(1 << 28); a_w[i] = 0; a_off[i] = 0;
        }
    }
    unsigned b_off[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = n0 + lrow + RPP * i;
        b_off[i] = n < p.Cout ? (unsigned)((n * p.Kpad + lcol) * 4) : OOB_OFFSET;
    }

    // Filter taps this tile needs, one bit per tap.  With position-major rows (layout 1: the detector head's
    // [7][7][roi][c] tensors) a 128-row tile spans one or two output positions, so the taps that fall into the
    // zero padding for ALL of them -- 14 % of the chunks of a 3x3 SAME conv on 7x7 maps -- are skipped outright;
    // they would only add exact zeros, so the result is bit-identical.
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

    // this workgroup's chunk range [kb, ke) of the (channel group, needed tap) sequence
    const int nk_all = CIN3 ? p.Kpad / BK : (p.Kpad / (BK * RS)) * n_taps;
    const int kb = SPLITK ? (int)((long long)slice * nk_all / splits) : 0;
    const int ke = SPLITK ? (int)((long long)(slice + 1) * nk_all / splits) : nk_all;

    unsigned rem = tap_mask;                                 // taps of the current channel group still to load
    int c0 = 0, w_grp = 0;                                   // channel offset / byte offset of the group's filter chunks
    if (SPLITK) {
        const int grp = kb / n_taps;
        c0 = grp * BK; w_grp = grp * RS * (BK * 4);
        for (int t = kb - grp * n_taps; t > 0; --t) rem &= rem - 1;
    }
    int kc3 = kb;                                            // CIN3: next chunk (of eight taps) to load
    // the NEXT chunk of this workgroup's sequence -> a set of staging registers (calls walk the sequence in order; calls
    // past the end of the range fetch in-bounds or zero data that is never multiplied)
    auto load_into = [&](i32x4 (&xa)[PA], i32x4 (&xb)[PB]) {
        if constexpr (CIN3) {
            const int tap = kc3 * 8 + (tid & 7);             // per lane: this lane's tap of the chunk
            const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
            const int tap_off = (r_tap * p.W + s_tap) * 12;  // 3 channels x 4 bytes per pixel
#pragma unroll
            for (int i = 0; i < PB; ++i)
                xb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i], kc3 * (BK * 4), 0);
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
                const bool ok = tap < RS && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const i32x3 v = __builtin_amdgcn_raw_buffer_load_b96(xrsrc, ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET, 0, 0);
                xa[i] = i32x4{v[0], v[1], v[2], 0};
            }
            ++kc3;
            return;
        }
        const int tap = __builtin_ctz(rem);                  // wave-uniform
        const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
        const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 4;
        const int w_off = w_grp + tap * (BK * 4);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            xb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i], w_off, 0);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int hi = a_h[i] + r_tap, wi = a_w[i] + s_tap;
            const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            xa[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET, 0, 0);
        }
        // branch-free walk to the next needed tap (a scalar branch here would split the loop body into two
        // scheduling regions and undo the interleave below)
        rem &= rem - 1;
        const int wrap = (rem == 0);
        rem |= wrap ? tap_mask : 0u;
        c0 += wrap * BK;
        w_grp += wrap * (RS * BK * 4);
    };
    auto store_from = [&](const i32x4 (&xa)[PA], const i32x4 (&xb)[PB], int buf) {
        float* a = As + buf * BM * LDS_STRIDE;
        float* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int i = 0; i < PA; ++i) *reinterpret_cast<i32x4*>(a + (lrow + RPP * i) * LDS_STRIDE + lcol) = xa[i];
#pragma unroll
        for (int i = 0; i < PB; ++i) *reinterpret_cast<i32x4*>(b + (lrow + RPP * i) * LDS_STRIDE + lcol) = xb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // the residual pieces this thread will add in the vector epilogue, requested now: a 64x64 tile's main loop is a few
    // microseconds, about as long as the fetch
    using EV = EpiVec<TM, TN, WM, WN>;
    constexpr bool EPI_PRE = EV::fits && !SPLITK && TM * TN == 1;
    f32x4 rpre[EPI_PRE ? EV::PASSES : 1];
    if constexpr (EPI_PRE) {
        if (p.vec_epi) epi_prefetch_residual<TM, TN, WM, WN>(p, m0, n0, tid, rpre);
    }

    constexpr int MF = TM * TN * 4;        // MFMAs per kk-step
    constexpr int NL = PA + PB;            // global loads == LDS stores per chunk per thread
    constexpr int NF = TM + TN;            // fragment reads per kk-step
    if constexpr (VARIANT == 2) {
        // ---- VARIANT 2: the barrier sits in the MIDDLE of a chunk and nothing waits behind it.
        // In VARIANTs 0/1 every chunk ends [LDS stores -> barrier -> first fragment reads -> first MFMA]: a wave alone
        // on its SIMD (stage-4 grids: one or two workgroups per CU) idles the matrix pipe for that whole chain -- timed
        // at 1 700 cycles per 1 024-cycle chunk (scripts/micro/conv_lab.hip, stamps).  Here chunk T is multiplied as
        //   first half : MFMAs of k-steps 0,1 (step-0 fragments were read during chunk T-1); fragment reads of steps 1..3
        //   barrier    : all waves have finished READING buffer T%2 and their stores of chunk T+1 (made during the
        //                second half of chunk T-1) are visible
        //   second half: MFMAs of k-steps 2,3; LDS stores of chunk T+2 into buffer T%2; fragment reads of chunk T+1's
        //                step 0; global loads of chunk T+4
        // so the operands of the MFMAs that follow the barrier are already in registers, the stores and the next reads
        // ride in the shadow of k-steps 2,3, and a chunk's global loads have TWO chunks of MFMA time to land (two
        // staging register sets; same two LDS buffers as before).
        // staging register sets: two for the 64-wide tiles (a chunk's loads get two chunks of MFMA time), one for the
        // big tiles (a 128x128 chunk is 4 096 MFMA cycles per wave: one chunk of lead is plenty, and the registers are needed)
        constexpr int SETS = TM * TN == 1 ? 2 : 1;
        i32x4 sa0[PA], sb0[PB], sa1[SETS == 2 ? PA : 1], sb1[SETS == 2 ? PB : 1];
        f32x4 na[TM], nb[TN];                                  // step-0 fragments of the chunk about to start
        load_into(sa0, sb0);                                   // chunk 0
        if constexpr (SETS == 2) {
            load_into(sa1, sb1);                               // chunk 1
            store_from(sa0, sb0, 0);
            load_into(sa0, sb0);                               // chunk 2
            store_from(sa1, sb1, 1);
            load_into(sa1, sb1);                               // chunk 3
        } else {
            store_from(sa0, sb0, 0);
            load_into(sa0, sb0);                               // chunk 1
            store_from(sa0, sb0, 1);
            load_into(sa0, sb0);                               // chunk 2
        }
        __syncthreads();
        LAB_STAMP(1);
        {
            const float* a = As + (wm * TM * 32 + li) * LDS_STRIDE + lh * 4;
            const float* b = Bs + (wn * TN * 32 + li) * LDS_STRIDE + lh * 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) na[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_STRIDE);
#pragma unroll
            for (int j = 0; j < TN; ++j) nb[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_STRIDE);
        }
        auto chunk = [&](auto parity, auto& xa, auto& xb) {
            constexpr int P = decltype(parity)::value;
            const float* a = As + P * BM * LDS_STRIDE + (wm * TM * 32 + li) * LDS_STRIDE + lh * 4;
            const float* b = Bs + P * BN * LDS_STRIDE + (wn * TN * 32 + li) * LDS_STRIDE + lh * 4;
            const float* an = As + (P ^ 1) * BM * LDS_STRIDE + (wm * TM * 32 + li) * LDS_STRIDE + lh * 4;
            const float* bn = Bs + (P ^ 1) * BN * LDS_STRIDE + (wn * TN * 32 + li) * LDS_STRIDE + lh * 4;
            f32x4 fa[BK / 8][TM], fb[BK / 8][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = na[i];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = nb[j];
#pragma unroll
            for (int kk = 1; kk < BK / 8; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[kk][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[kk][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_STRIDE + kk * 8);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i][e], fb[kk][j][e], acc[i][j], 0, 0, 0);
            SGB(SG_DS_RD, NF);                                       // step-1 fragments first
#pragma unroll
            for (int q = 0; q < MF; ++q) {                           // k-step 0: the step-2,3 fragment reads behind its MFMAs
                SGB(SG_MFMA, 1);
                if (q < NF) SGB(SG_DS_RD, 2);
            }
#pragma unroll
            for (int q = 0; q < MF; ++q) SGB(SG_MFMA, 1);            // k-step 1
            __builtin_amdgcn_sched_barrier(0);                       // k-steps 0,1 stay in FRONT of the barrier: its wait then falls behind 8 queued MFMAs
            __syncthreads();
            store_from(xa, xb, P);                                   // chunk T+2 -> the buffer every wave has just finished reading
            load_into(xa, xb);                                       // chunk T+4
#pragma unroll
            for (int i = 0; i < TM; ++i) na[i] = *reinterpret_cast<const f32x4*>(an + i * 32 * LDS_STRIDE);
#pragma unroll
            for (int j = 0; j < TN; ++j) nb[j] = *reinterpret_cast<const f32x4*>(bn + j * 32 * LDS_STRIDE);
#pragma unroll
            for (int kk = 2; kk < BK / 8; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i][e], fb[kk][j][e], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < MF; ++q) {                           // k-step 2: LDS stores, then the next chunk's step-0 reads
                SGB(SG_MFMA, 1);
                if (q < NL) SGB(SG_DS_WR, 1);
            }
            if (NL > MF) SGB(SG_DS_WR, NL - MF);
            SGB(SG_DS_RD, NF);
#pragma unroll
            for (int q = 0; q < MF; ++q) {                           // k-step 3: global loads
                SGB(SG_MFMA, 1);
                if (q < NL) { SGB(SG_VALU, 4); SGB(SG_VMEM_RD, 1); }
            }
        };
        int kc = kb;
        for (; kc + 1 < ke; kc += 2) {
            chunk(std::integral_constant<int, 0>{}, sa0, sb0);
            if constexpr (SETS == 2) chunk(std::integral_constant<int, 1>{}, sa1, sb1);
            else chunk(std::integral_constant<int, 1>{}, sa0, sb0);
        }
        if (kc < ke) chunk(std::integral_constant<int, 0>{}, sa0, sb0);
        __syncthreads();                                             // the epilogue reuses the buffers
    } else {
    i32x4 ra[PA], rb[PB];
    auto load_chunk = [&](int) { load_into(ra, rb); };
    auto store_chunk = [&](int buf) { store_from(ra, rb, buf); };
    // Two-deep operand pipeline: chunk t+2 travels global->registers while chunk t+1 travels
    // registers->LDS and chunk t feeds the MFMAs.  The loads issued in iteration t are consumed
    // (ds_write) at the top of iteration t+1, so they have a whole chunk of MFMA time to land and
    // the compiler cannot sink them next to their use.
    load_chunk(kb);
    store_chunk(0);
    load_chunk(kb + 1 < ke ? kb + 1 : kb);
    __syncthreads();
    LAB_STAMP(1);

    for (int kc = kb; kc < ke; ++kc) {
        const int buf = (kc - kb) & 1;
        if constexpr (VARIANT == 0) {
            store_chunk(buf ^ 1);                              // chunk kc+1 (harmless duplicate at the tail)
            load_chunk(kc + 2 < ke ? kc + 2 : ke - 1);         // always in range: keeps the body branch-free
        }
        const float* a = As + buf * BM * LDS_STRIDE + (wm * TM * 32 + li) * LDS_STRIDE + lh * 4;
        const float* b = Bs + buf * BN * LDS_STRIDE + (wn * TN * 32 + li) * LDS_STRIDE + lh * 4;
        f32x4 fa[BK / 8][TM], fb[BK / 8][TN];
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[kk][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[kk][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_STRIDE + kk * 8);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i][e], fb[kk][j][e], acc[i][j], 0, 0, 0);

        if constexpr (VARIANT == 0) {
            // ---- pinned interleave (per wave, per chunk): everything that is not an MFMA issues in
            // the shadow of the wave's own MFMAs
            SGB(SG_DS_RD, NF);                                       // kk = 0 fragments
#pragma unroll
            for (int q = 0; q < MF; ++q) {                           // kk-step 0: LDS stores, then kk=1 fragments
                SGB(SG_MFMA, 1);
                if (q < NL) SGB(SG_DS_WR, 1);
                else if (q - NL < NF) SGB(SG_DS_RD, 1);
            }
#pragma unroll
            for (int q = 0; q < MF; ++q) {                           // kk-step 1: global loads, then kk=2 fragments
                SGB(SG_MFMA, 1);
                if (q < NL) { SGB(SG_VALU, 4); SGB(SG_VMEM_RD, 1); }
                else if (q - NL < NF) SGB(SG_DS_RD, 1);
            }
#pragma unroll
            for (int q = 0; q < MF; ++q) {                           // kk-step 2: kk=3 fragments
                SGB(SG_MFMA, 1);
                if (q < NF) SGB(SG_DS_RD, 1);
            }
#pragma unroll
            for (int q = 0; q < MF; ++q) SGB(SG_MFMA, 1);            // kk-step 3
        } else {
            // VARIANT 1: the LDS stores come AFTER all fragment reads in program order (the compiler keeps
            // may-alias LDS accesses ordered), so they can ride behind the last MFMAs instead of sitting in
            // front of the first one; the next loads follow the stores (register reuse) at the tail.
            store_chunk(buf ^ 1);
            load_chunk(kc + 2 < ke ? kc + 2 : ke - 1);
            SGB(SG_DS_RD, NF);
#pragma unroll
            for (int kk = 0; kk < BK / 8 - 1; ++kk) {
#pragma unroll
                for (int q = 0; q < MF; ++q) {
                    SGB(SG_MFMA, 1);
                    if (q < NF) SGB(SG_DS_RD, 1);
                    else if (kk == BK / 8 - 2 && q - NF < NL) SGB(SG_DS_WR, 1);
                }
            }
#pragma unroll
            for (int q = 0; q < MF; ++q) {
                SGB(SG_MFMA, 1);
                if (MF - NF < NL && q < NL - (MF - NF)) SGB(SG_DS_WR, 1);
                else if (q < NL + (MF - NF < NL ? NL - (MF - NF) : 0)) { SGB(SG_VALU, 4); SGB(SG_VMEM_RD, 1); }
            }
        }
        __syncthreads();
    }
    }
    LAB_STAMP(2);
    if constexpr (SPLITK) {
        // publish this slice's partial tile WRITE-THROUGH (sc1 stores need no release fence: cdna guide G16 R1);
        // thread-major 16-B rows, so the stores and the reducer's loads coalesce
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains ...
        __syncthreads();                                       // ... before ONE lane draws the ticket
        int* last = reinterpret_cast<int*>(smem);              // the main loop is done with the (one) LDS array
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
        LAB_STAMP(3);
        if (!*last) return;
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
        LAB_STAMP(4);
    }
    if constexpr (EV::fits) {
        if (p.vec_epi) epilogue_vec<TM, TN, WM, WN, EPI_PRE>(acc, p, m0, n0, tid, wm, wn, li, lh, smem, rpre);
        else epilogue<TM, TN>(acc, p, m0, n0, wm, wn, li, lh);
    } else {
        epilogue<TM, TN>(acc, p, m0, n0, wm, wn, li, lh);
    }
#ifdef FRCNN_LAB_STAMPS
    LAB_STAMP(5);                                            // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LAB_STAMP(6);                                            // stores done
    if (g_lab_stamps && threadIdx.x == 0) {
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_lab_stamps[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

// ------------------------------------------------------------------------------------
// Balanced ("stream-K") launch form of the v2 kernel, late-store variant, 2x2 waves.
//
// A launch of T tiles on S concurrent workgroup slots runs ceil(T/S) rounds whatever T is: the detector head's
// 460 128x128 tiles on 512 slots leave 52 CUs with one workgroup while 204 carry two, and the launch lasts as long
// as the loaded ones (0.90 of the slots' work; a bare MFMA + LDS loop of this shape measures 127 vs 141 TFLOP/s,
// scripts/micro/mfma_ladder.hip).  Here the launch has G = rounds x S workgroups and the UNIT of work is a k-chunk:
// the tiles' chunks, tile after tile (a tile's count depends on its row range only: position-major rows skip
// padding-only taps), form one sequence of U units and workgroup w takes units [w*U/G, (w+1)*U/G).  A range covers
// the tail of one tile, possibly whole tiles, and the head of another.  A tile covered by ONE workgroup goes straight
// to the epilogue; otherwise each contributor publishes its f32 partial tile in slot (w - first contributor) with
// write-through stores and adds its chunk count to the tile's ticket -- the one that completes the count sums the
// slots in slot order (deterministic) and runs the epilogue, exactly the split-K protocol with unequal slices.
constexpr int SK_SLOTS = 4;             // partial-tile slots per output tile (the host keeps ranges long enough)

template <int TM, int TN>
__global__ void __launch_bounds__(256) k_conv_igemm_f32_sk(const ConvArgs p) {
    constexpr int WM = 2, WN = 2, NT = 256;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int RPP = NT / 8;
    constexpr int PA = BM / RPP, PB = BN / RPP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDS_STRIDE;
    int* pref = reinterpret_cast<int*>(smem + 2 * (BM + BN) * LDS_STRIDE);       // [tiles_m + 1] chunk-count prefix over row tiles

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int lrow = tid >> 3, lcol = (tid & 7) * 4;
    const int RS = p.R * p.S;
    const unsigned all_taps = RS >= 32 ? 0xffffffffu : (1u << RS) - 1u;
    const int groups = p.Kpad / (BK * RS);

    auto taps_of = [&](int tile_m) -> unsigned {
        if (!p.layout) return all_taps;
        const int m0 = tile_m * BM;
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
    };
    for (int m = tid; m < p.tiles_m; m += NT) pref[m + 1] = groups * __popc(taps_of(m));
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        pref[0] = 0;
        for (int m = 1; m <= p.tiles_m; ++m) { run += pref[m]; pref[m] = run; }
    }
    __syncthreads();
    const int Pm = pref[p.tiles_m];
    const long long U = (long long)Pm * p.tiles_n;
    const int G = gridDim.x;
    const int w = xcd_remap(blockIdx.x, G);
    long long u = (long long)w * U / G;
    const long long u_end = (long long)(w + 1) * U / G;

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)((size_t)p.Cout * p.Kpad * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.slabs, 0, (int)((size_t)p.tiles_m * p.tiles_n * SK_SLOTS * (BM * BN) * 4), 0x00020000);

    while (u < u_end) {
        __syncthreads();                                     // the previous segment is done with the LDS (operands, flag)
        const int tile_n = (int)(u / Pm);
        const int r_u = (int)(u - (long long)tile_n * Pm);
        int lo = 0, hi = p.tiles_m;                          // largest tile_m with pref[tile_m] <= r_u
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pref[mid] <= r_u) lo = mid; else hi = mid; }
        const int tile_m = lo;
        const int nk_t = pref[tile_m + 1] - pref[tile_m];
        const int kb = r_u - pref[tile_m];
        const int ke = (int)min((long long)nk_t, kb + (u_end - u));
        const int tile = tile_n * p.tiles_m + tile_m;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        const unsigned tap_mask = taps_of(tile_m);
        const int n_taps = __popc(tap_mask);

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
                a_off[i] = (img * p.img_stride + (a_h[i] * p.W + a_w[i]) * p.pix_stride + lcol) * 4;
            } else {
                a_h[i] = -(1 << 28); a_w[i] = 0; a_off[i] = 0;
            }
        }
        unsigned b_off[PB];
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int n = n0 + lrow + RPP * i;
            b_off[i] = n < p.Cout ? (unsigned)((n * p.Kpad + lcol) * 4) : OOB_OFFSET;
        }

        i32x4 ra[PA], rb[PB];
        unsigned rem = tap_mask;
        const int grp0 = kb / n_taps;
        int c0 = grp0 * BK, w_grp = grp0 * RS * (BK * 4);
        for (int t = kb - grp0 * n_taps; t > 0; --t) rem &= rem - 1;
        auto load_chunk = [&]() {
            const int tap = __builtin_ctz(rem);
            const int r_tap = (tap * p.inv_S) >> 16, s_tap = tap - r_tap * p.S;
            const int tap_off = ((r_tap * p.W + s_tap) * p.pix_stride + c0) * 4;
            const int w_off = w_grp + tap * (BK * 4);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, b_off[i], w_off, 0);
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int hi2 = a_h[i] + r_tap, wi = a_w[i] + s_tap;
                const bool ok = (unsigned)hi2 < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? (unsigned)(a_off[i] + tap_off) : OOB_OFFSET, 0, 0);
            }
            rem &= rem - 1;
            const int wrap = (rem == 0);
            rem |= wrap ? tap_mask : 0u;
            c0 += wrap * BK;
            w_grp += wrap * (RS * BK * 4);
        };
        auto store_chunk = [&](int buf) {
            float* a = As + buf * BM * LDS_STRIDE;
            float* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
            for (int i = 0; i < PA; ++i) *reinterpret_cast<i32x4*>(a + (lrow + RPP * i) * LDS_STRIDE + lcol) = ra[i];
#pragma unroll
            for (int i = 0; i < PB; ++i) *reinterpret_cast<i32x4*>(b + (lrow + RPP * i) * LDS_STRIDE + lcol) = rb[i];
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

        load_chunk();
        store_chunk(0);
        load_chunk();                                        // past the range's end at most: loaded, never multiplied
        __syncthreads();

        constexpr int MF = TM * TN * 4, NL = PA + PB, NF = TM + TN;
        for (int kc = kb; kc < ke; ++kc) {
            const int buf = (kc - kb) & 1;
            const float* a = As + buf * BM * LDS_STRIDE + (wm * TM * 32 + li) * LDS_STRIDE + lh * 4;
            const float* b = Bs + buf * BN * LDS_STRIDE + (wn * TN * 32 + li) * LDS_STRIDE + lh * 4;
            f32x4 fa[BK / 8][TM], fb[BK / 8][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[kk][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDS_STRIDE + kk * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[kk][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDS_STRIDE + kk * 8);
            }
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk][i][e], fb[kk][j][e], acc[i][j], 0, 0, 0);
            store_chunk(buf ^ 1);
            load_chunk();
            SGB(SG_DS_RD, NF);
#pragma unroll
            for (int kk = 0; kk < BK / 8 - 1; ++kk) {
#pragma unroll
                for (int q = 0; q < MF; ++q) {
                    SGB(SG_MFMA, 1);
                    if (q < NF) SGB(SG_DS_RD, 1);
                    else if (kk == BK / 8 - 2 && q - NF < NL) SGB(SG_DS_WR, 1);
                }
            }
#pragma unroll
            for (int q = 0; q < MF; ++q) {
                SGB(SG_MFMA, 1);
                if (MF - NF < NL && q < NL - (MF - NF)) SGB(SG_DS_WR, 1);
                else if (q < NL + (MF - NF < NL ? NL - (MF - NF) : 0)) { SGB(SG_VALU, 4); SGB(SG_VMEM_RD, 1); }
            }
            __syncthreads();
        }
        u += ke - kb;

        if (kb != 0 || ke != nk_t) {
            // partial tile: which contributors does this tile have?  first = the workgroup whose range holds the tile's
            // first unit, last = the one holding its last (w*U/G <= x  <=>  w <= ((x+1)*G - 1) / U)
            const long long ts = (long long)tile_n * Pm + pref[tile_m];
            const int w_first = (int)(((ts + 1) * G - 1) / U);
            const int w_last = (int)(((ts + nk_t) * G - 1) / U);
            const unsigned slab_off = (unsigned)(((size_t)tile * SK_SLOTS + (w - w_first)) * (BM * BN) * 4 + tid * 16);
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
            int* last = reinterpret_cast<int*>(smem);
            if (tid == 0) {
                const unsigned mine = (unsigned)(ke - kb);
                const unsigned t = __hip_atomic_fetch_add(&p.tickets[tile], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int is_last = (t + mine == (unsigned)nk_t);
                if (is_last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&p.tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                *last = is_last;
            }
            __syncthreads();
            if (!*last) continue;                             // somebody else finishes this tile
            const float4* base = reinterpret_cast<const float4*>(p.slabs + (size_t)tile * SK_SLOTS * (BM * BN));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
            for (int sl = 0; sl <= w_last - w_first; ++sl) {
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
        if (p.vec_epi) epilogue_vec<TM, TN, WM, WN, false>(acc, p, m0, n0, tid, wm, wn, li, lh, smem, nullptr);
        else epilogue<TM, TN>(acc, p, m0, n0, wm, wn, li, lh);
    }
}

// ------------------------------------------------------------------------------------
// filter packing: Keras HWIO [R][S][Cin][Cout] -> [Cout][Kpad].
//   Cin % 32 == 0 : packed k = ((c/32)*R*S + tap)*32 + c%32   (channel chunk outer, tap inner)
//   Cin == 3      : packed k = tap*4 + c, c == 3 and taps >= R*S zero (the stem kernel stages eight taps per chunk)
//   otherwise     : packed k = tap*Cin + c, zero padded to Kpad  (small-Cin path decodes k itself)
__host__ __device__ __forceinline__ int packed_k(int RS, int Cin) {
    return ((Cin == 3 ? RS * 4 : RS * Cin) + BK - 1) / BK * BK;
}
__device__ __forceinline__ float pack_hwio_elem(const float* w, int RS, int Cin, int Cout, int Kpad, size_t i) {
    const int k = (int)(i % Kpad), n = (int)(i / Kpad);
    if ((Cin % BK) == 0) {
        const int j = k % BK, kc = k / BK, tap = kc % RS, cc = kc / RS;
        return w[((size_t)tap * Cin + cc * BK + j) * Cout + n];
    }
    if (Cin == 3) {
        const int tap = k >> 2, c = k & 3;
        return (c < 3 && tap < RS) ? w[((size_t)tap * 3 + c) * Cout + n] : 0.0f;
    }
    return k < RS * Cin ? w[(size_t)k * Cout + n] : 0.0f;
}
__global__ void k_pack_hwio(const float* w, int RS, int Cin, int Cout, int Kpad, float* out) {
    const size_t total = (size_t)Cout * Kpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        out[i] = pack_hwio_elem(w, RS, Cin, Cout, Kpad, i);
}

// ------------------------------------------------------------------------------------
// Backward of a stride-1 convolution w.r.t. its input = a forward convolution of the output
// gradient with the filter transposed (Cin <-> Cout) and flipped in both taps.  The per-channel
// epilogue scale s[co] of the forward layer (folded BatchNorm) multiplies the incoming gradient,
// which is the same as scaling the transposed filter's INPUT channel co, so it is folded here and
// the dgrad launch is an ordinary frcnn_conv2d_fwd on these weights.
//   w'[r'][s'][co][ci] = w[R-1-r'][S-1-s'][ci][co] * s[co]      (conv' has Cin' = Cout, Cout' = Cin)
__device__ __forceinline__ float pack_dgrad_elem(const float* w, const float* scale, int R, int S, int Cin, int Cout, int Kpad, size_t i) {
    const int RS = R * S;                                   // rows = Cout' = Cin, k over (co chunk, tap', co)
    const int k = (int)(i % Kpad), ci = (int)(i / Kpad);
    int tap, co;
    if ((Cout % BK) == 0) { const int j = k % BK, kc = k / BK; tap = kc % RS; co = (kc / RS) * BK + j; }
    else if (Cout == 3) { tap = k >> 2; co = k & 3; if (co == 3 || tap >= RS) return 0.0f; }
    else { if (k >= RS * Cout) return 0.0f; tap = k / Cout; co = k % Cout; }
    const int r = R - 1 - tap / S, sx = S - 1 - tap % S;
    return w[((size_t)(r * S + sx) * Cin + ci) * Cout + co] * (scale ? scale[co] : 1.0f);
}
__global__ void k_pack_dgrad(const float* w, const float* scale, int R, int S, int Cin, int Cout, int Kpad, float* out) {
    const size_t total = (size_t)Cin * Kpad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        out[i] = pack_dgrad_elem(w, scale, R, S, Cin, Cout, Kpad, i);
}

// One launch re-derives EVERY trainable layer's device-side forms from the fp32 master weights after an
// optimiser step: forward pack, input-gradient pack and the folded epilogue shift.  The job table rides in
// the kernel arguments (no table upload); blockIdx.y = job.
constexpr int REFRESH_JOBS = 32;
// Workgroups are dealt out in proportion to each job's size (first_block: prefix over the jobs): with a fixed 96 per job
// the 4.7 M-element RPN filter kept 96 workgroups busy long after the 1x1 layers' had left (159 us per fp32 RPN step).
struct RefreshTable { frcnn_pack_job job[REFRESH_JOBS]; int first_block[REFRESH_JOBS + 1]; int n; };
static int refresh_blocks(const frcnn_pack_job& j) {
    const long long elems = (long long)j.kh * j.kw * j.cin * j.cout;
    long long g = (elems + 2047) / 2048;              // (round 6: 8192 per workgroup left the dense layer's element-wise input-gradient pack on 26 workgroups)
    // (at least 16 workgroups: the small f32 layers of a mixed-precision step -- rpn_out_cls / rpn_out_bbreg, 512 -> 9 / 36 -- are a
    //  launch of their own whose 16 transposing tiles went through 4 workgroups one after the other: 24-33 us of a 1.2 ms step)
    return (int)(g < 16 ? 16 : (g > 2048 ? 2048 : g));
}
__global__ void __launch_bounds__(256) k_refresh_packed(const RefreshTable t) {
    int ji = 0;
    while (ji + 1 < t.n && (int)blockIdx.x >= t.first_block[ji + 1]) ++ji;
    const frcnn_pack_job& j = t.job[ji];
    const int bx = (int)blockIdx.x - t.first_block[ji], gsz = t.first_block[ji + 1] - t.first_block[ji];
    const int RS = j.kh * j.kw;
    const size_t stride = (size_t)gsz * blockDim.x, first = (size_t)bx * blockDim.x + threadIdx.x;
    if (j.packed && (j.cin % BK) == 0) {
        // HWIO has cout fastest, the packed rows have the 32 channels of a chunk fastest: transpose 32 x 64
        // (channel x cout) tiles through LDS so both the reads (256 B) and the writes (128 B) are whole segments
        __shared__ float tile[BK][65];
        const int Kpad = RS * j.cin, nblk = (j.cout + 63) / 64, ntiles = RS * (j.cin / BK) * nblk;
        const int lane = threadIdx.x & 63, jr = threadIdx.x >> 6, wn = threadIdx.x >> 3, j4 = (threadIdx.x & 7) * 4;
        for (int tl = bx; tl < ntiles; tl += gsz) {
            const int nb = tl % nblk, kc = tl / nblk, tap = kc % RS, cc = kc / RS, n0 = nb * 64;
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const int c = jr + 4 * pp;
                tile[c][lane] = n0 + lane < j.cout ? j.w_hwio[((size_t)tap * j.cin + cc * BK + c) * j.cout + n0 + lane] : 0.0f;
            }
            __syncthreads();
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int n = wn + 32 * pass;
                if (n0 + n < j.cout)
                    *reinterpret_cast<float4*>(j.packed + (size_t)(n0 + n) * Kpad + kc * BK + j4) =
                        make_float4(tile[j4][n], tile[j4 + 1][n], tile[j4 + 2][n], tile[j4 + 3][n]);
            }
            __syncthreads();
        }
    } else if (j.packed) {
        const int Kpad = packed_k(RS, j.cin);
        const size_t total = (size_t)j.cout * Kpad;
        for (size_t i = first; i < total; i += stride) j.packed[i] = pack_hwio_elem(j.w_hwio, RS, j.cin, j.cout, Kpad, i);
    }
    if (j.packed_dgrad) {
        const int Kpad = packed_k(RS, j.cout);
        const size_t total = (size_t)j.cin * Kpad;
        for (size_t i = first; i < total; i += stride) j.packed_dgrad[i] = pack_dgrad_elem(j.w_hwio, j.scale, j.kh, j.kw, j.cin, j.cout, Kpad, i);
    }
    if (j.shift)
        for (size_t i = first; i < (size_t)j.cout; i += stride)
            j.shift[i] = (j.bias ? j.bias[i] : 0.0f) * (j.scale ? j.scale[i] : 1.0f) + (j.shift_const ? j.shift_const[i] : 0.0f);
}

// Bias gradients of many layers in ONE launch: out[co] = scale[co] * sum_m g[m][co].  A 1024-thread workgroup
// owns 64 columns of one job; its 16 waves stride over the rows (256-B coalesced reads) and are summed in a
// fixed order, so the result is reproducible.  blockIdx.x walks the (job, column group) pairs.
constexpr int COLSUM_JOBS = 64;
struct ColsumTable { frcnn_colsum_job job[COLSUM_JOBS]; int first_block[COLSUM_JOBS + 1]; int n; };
__global__ void __launch_bounds__(1024) k_colsum_batch(const ColsumTable t) {
    __shared__ float part[16][64];
    int ji = 0;
    while (ji + 1 < t.n && (int)blockIdx.x >= t.first_block[ji + 1]) ++ji;
    const frcnn_colsum_job& j = t.job[ji];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = ((int)blockIdx.x - t.first_block[ji]) * 64 + lane;
    float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
    if (co < j.cout && !j.g_is_bf16) {
        const float* g = reinterpret_cast<const float*>(j.g) + co;
        int m = wave;
        for (; m + 48 < j.m; m += 64) {
            v0 += g[(size_t)m * j.cout]; v1 += g[(size_t)(m + 16) * j.cout];
            v2 += g[(size_t)(m + 32) * j.cout]; v3 += g[(size_t)(m + 48) * j.cout];
        }
        for (; m < j.m; m += 16) v0 += g[(size_t)m * j.cout];
    } else if (co < j.cout) {
        const __bf16* g = reinterpret_cast<const __bf16*>(j.g) + co;
        int m = wave;
        for (; m + 48 < j.m; m += 64) {
            v0 += (float)g[(size_t)m * j.cout]; v1 += (float)g[(size_t)(m + 16) * j.cout];
            v2 += (float)g[(size_t)(m + 32) * j.cout]; v3 += (float)g[(size_t)(m + 48) * j.cout];
        }
        for (; m < j.m; m += 16) v0 += (float)g[(size_t)m * j.cout];
    }
    part[wave][lane] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (wave == 0 && co < j.cout) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += part[w][lane];
        j.out[co] = j.scale ? s * j.scale[co] : s;
    }
}

// ------------------------------------------------------------------------------------
// Weight gradient on the matrix cores:  dW[tap][ci][co] = s[co] * sum_m A[m][(tap,ci)] * G[m][co]
// (A = implicit im2col of the layer input x, G = gradient w.r.t. the layer's pre-activation
// output, m = output pixel).  The reduction index is the PIXEL, so both operands are staged
// [pixel][channel] exactly as they lie in HBM (NHWC) and the 32x32x2 MFMA reads them with
// conflict-free ds_read_b32 (lane = channel).  Workgroup = 4 waves = 64 (ci) x 64 (co) outputs of
// one filter tap; grid.z splits the pixel range, each slice writes its own partial slab and a
// second kernel reduces the slabs in a fixed order (bitwise reproducible; no float atomics).
struct WgradArgs {
    const void* x; const void* g; float* partial;            // x / g: f32, or bf16 for the IN_BF16 instantiation
    int n_img, H, W, Cin, Cout, R, S, stride, pad_top, pad_left, Ho, Wo, M;
    int m_per_slice;
};

constexpr int WG_MC = 32;                 // pixels per staged chunk
constexpr int WG_LD = 64 + 4;             // LDS row stride in floats (272 B keeps 16-B alignment for the b128 stores)

// IN_BF16: activations and gradients arrive in bf16 (mixed-precision training); they are widened while being
// staged, the products accumulate in f32 on the same f32-input MFMA, dW leaves in f32 for the master weights.
__device__ __forceinline__ f32x4 load4_bf16(const __bf16* p) {
    const uint2 raw = *reinterpret_cast<const uint2*>(p);                 // 4 x bf16 = 8 bytes
    f32x4 v;
    v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
    v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    return v;
}

template <bool IN_BF16>
__device__ __forceinline__ void wgrad_body_f32(const WgradArgs& p, int bx, int by, int bz) {
    __shared__ __attribute__((aligned(16))) float Xs[2][WG_MC][WG_LD];
    __shared__ __attribute__((aligned(16))) float Gs[2][WG_MC][WG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int ci_tiles = (p.Cin + 63) / 64;
    const int tap = bx / ci_tiles, ci0 = (bx % ci_tiles) * 64;
    const int r_tap = tap / p.S, s_tap = tap % p.S;
    const int co0 = by * 64;
    const int m_begin = bz * p.m_per_slice, m_end = min(p.M, m_begin + p.m_per_slice);

    // staging: 256 threads move 32 pixels x 64 channels (16 float4 per pixel) per operand per chunk
    const int srow = tid >> 4, scol = (tid & 15) * 4;      // rows srow and srow+16
    f32x4 rx[2], rg[2];
    auto load = [&](int mc) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = mc + srow + 16 * q;
            f32x4 vx = {0, 0, 0, 0}, vg = {0, 0, 0, 0};
            if (m < m_end) {
                const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, img = t / p.Ho;
                const int hi = ho * p.stride - p.pad_top + r_tap, wi = wo * p.stride - p.pad_left + s_tap;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                    const size_t off = (((size_t)img * p.H + hi) * p.W + wi) * p.Cin + ci0 + scol;
                    if constexpr (IN_BF16) {
                        const __bf16* src = reinterpret_cast<const __bf16*>(p.x) + off;
                        if (ci0 + scol + 3 < p.Cin) vx = load4_bf16(src);
                        else for (int e = 0; e < 4; ++e) if (ci0 + scol + e < p.Cin) vx[e] = (float)src[e];
                    } else {
                        const float* src = reinterpret_cast<const float*>(p.x) + off;
                        if (ci0 + scol + 3 < p.Cin) vx = *reinterpret_cast<const f32x4*>(src);
                        else for (int e = 0; e < 4; ++e) if (ci0 + scol + e < p.Cin) vx[e] = src[e];
                    }
                }
                const size_t goff = (size_t)m * p.Cout + co0 + scol;
                if constexpr (IN_BF16) {
                    const __bf16* gs = reinterpret_cast<const __bf16*>(p.g) + goff;
                    if (co0 + scol + 3 < p.Cout && (p.Cout & 3) == 0) vg = load4_bf16(gs);
                    else for (int e = 0; e < 4; ++e) if (co0 + scol + e < p.Cout) vg[e] = (float)gs[e];
                } else {
                    const float* gs = reinterpret_cast<const float*>(p.g) + goff;
                    if (co0 + scol + 3 < p.Cout && (p.Cout & 3) == 0) vg = *reinterpret_cast<const f32x4*>(gs);
                    else for (int e = 0; e < 4; ++e) if (co0 + scol + e < p.Cout) vg[e] = gs[e];
                }
            }
            rx[q] = vx; rg[q] = vg;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<f32x4*>(&Xs[buf][srow + 16 * q][scol]) = rx[q];
            *reinterpret_cast<f32x4*>(&Gs[buf][srow + 16 * q][scol]) = rg[q];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    const int n_chunks = (m_end - m_begin + WG_MC - 1) / WG_MC;
    if (n_chunks > 0) {
        load(m_begin);
        store(0);
        __syncthreads();
        for (int c = 0; c < n_chunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < n_chunks) load(m_begin + (c + 1) * WG_MC);
#pragma unroll
            for (int st = 0; st < WG_MC / 2; ++st) {
                const float a = Xs[buf][2 * st + lh][wk * 32 + li];
                const float b = Gs[buf][2 * st + lh][wn * 32 + li];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (c + 1 < n_chunks) store(buf ^ 1);
            __syncthreads();
        }
    }
    // partial slab layout = HWIO: [slice][tap][ci][co]
    const int co = co0 + wn * 32 + li;
    if (co < p.Cout) {
        float* dst = p.partial + ((size_t)bz * p.R * p.S + tap) * p.Cin * p.Cout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ci0 + wk * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
            if (ci < p.Cin) dst[(size_t)ci * p.Cout + co] = acc[e];
        }
    }
}

template <bool IN_BF16>
__global__ void __launch_bounds__(256) k_conv_wgrad_f32(const WgradArgs p) {
    wgrad_body_f32<IN_BF16>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The 128 (ci) x 128 (co) form of the f32 weight gradient, for layers with cin, cout >= 128 (every trainable layer of the
// ResNet stages 3-5, the RPN and VGG from block 2 on).  The 64x64 body above reads one A and one B value per MFMA
// (ds_read_b32, and its 68-float rows put the two k-rows of a read 4 banks apart: 2-way conflicts) and moves 16 KB of
// operands per 262 kFLOP -- ~70 TFLOP/s on the training steps' layers (profiles/round2_lab/train_trace_by_grid_f32_*).
// Here each wave owns 64 x 64: its two 32-wide row tiles are the EVEN and the ODD channels of its 64 (the output-row
// permutation is free, the epilogue undoes it), so ONE ds_read_b64 per operand feeds four MFMAs; a b64 read is served
// half-wave by half-wave, each half one unpadded 128-float row segment = every bank once.  Operand traffic per FLOP
// halves, the chunk (32 pixels) is 64 MFMAs = 4096 cycles per wave against 8 + 8 staging copies per thread, two
// workgroups (64 KB of LDS each) share a CU.  Pixel coordinates advance incrementally (no division in the loop);
// halo / tail / channel edges ride on the buffer descriptors.  Slab layout and the fixed-order slice reduction are
// unchanged; the pixel order inside a slice is the 64x64 body's, only the slice boundaries move with the tile count.
constexpr int WGB_LD = 128;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wgrad_body_f32_big(const WgradArgs& p, int bx, int by, int bz) {
    __shared__ __attribute__((aligned(16))) float Xs[2][WG_MC][WGB_LD];
    __shared__ __attribute__((aligned(16))) float Gs[2][WG_MC][WGB_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int ci_tiles = (p.Cin + 127) / 128;
    const int tap = bx / ci_tiles, ci0 = (bx % ci_tiles) * 128;
    const int r_tap = tap / p.S, s_tap = tap % p.S;
    const int co0 = by * 128;
    const int m_begin = bz * p.m_per_slice, m_end = min(p.M, m_begin + p.m_per_slice);

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.g), 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);

    // staging: 256 threads move 32 pixels x 128 channels per operand per chunk: 8 rows per pass, 4 passes
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const bool ci_ok = ci0 + scol < p.Cin, co_ok = co0 + scol < p.Cout;
    const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
    // n / d for 0 <= n < 2^23 (the host keeps M below that): the float product is within one of the quotient
    auto divmod = [](int n, int d, float inv, int& q, int& r) {
        q = (int)((float)n * inv); r = n - q * d;
        if (r < 0) { r += d; --q; }
        if (r >= d) { r -= d; ++q; }
    };
    int mc = m_begin;
    i32x4 rx[4], rg[4];
    auto load = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mc + srow + 8 * q;
            int wo, t, ho, img;
            divmod(m, p.Wo, inv_wo, t, wo);
            divmod(t, p.Ho, inv_ho, img, ho);
            const int hi = ho * p.stride - p.pad_top + r_tap, wi = wo * p.stride - p.pad_left + s_tap;
            const bool in = m < m_end && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned xoff = (unsigned)(((img * p.H + hi) * p.W + wi) * p.Cin + ci0 + scol) * 4u;
            rx[q] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, in && ci_ok ? xoff : OOB_OFFSET, 0, 0);
            const unsigned goff = (unsigned)(m * p.Cout + co0 + scol) * 4u;
            rg[q] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, m < m_end && co_ok ? goff : OOB_OFFSET, 0, 0);
        }
        mc += WG_MC;
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<i32x4*>(&Xs[buf][srow + 8 * q][scol]) = rx[q];
            *reinterpret_cast<i32x4*>(&Gs[buf][srow + 8 * q][scol]) = rg[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // chunk c multiplies from LDS buffer c & 1 while chunk c+1 (in registers since the previous iteration) moves into the
    // other buffer and chunk c+2 is requested: every global load has a whole chunk (64 MFMAs = 4096 cycles) to land.
    // Loads past the slice carry m >= m_end: zeros.
    const int n_chunks = (m_end - m_begin + WG_MC - 1) / WG_MC;
    if (n_chunks > 0) {
        load();
        store(0);
        load();
        __syncthreads();
        for (int c = 0; c < n_chunks; ++c) {
            const int buf = c & 1;
            store(buf ^ 1);
            load();
            const float* xa = &Xs[buf][lh][wk * 64 + 2 * li];
            const float* gb = &Gs[buf][lh][wn * 64 + 2 * li];
            f32x2 fa[WG_MC / 2], fb[WG_MC / 2];
#pragma unroll
            for (int st = 0; st < WG_MC / 2; ++st) {
                fa[st] = *reinterpret_cast<const f32x2*>(xa + 2 * st * WGB_LD);
                fb[st] = *reinterpret_cast<const f32x2*>(gb + 2 * st * WGB_LD);
            }
#pragma unroll
            for (int st = 0; st < WG_MC / 2; ++st) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][0], fb[st][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][0], fb[st][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][1], fb[st][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[st][1], fb[st][1], acc[1][1], 0, 0, 0);
            }
            // issue order: fragment reads run ahead of the MFMAs that need them, the LDS stores ride behind the first
            // MFMAs, the address arithmetic and the eight global loads behind the middle ones
            SGB(SG_DS_RD, 4);
#pragma unroll
            for (int q = 0; q < 8; ++q) { SGB(SG_MFMA, 1); SGB(SG_DS_WR, 1); SGB(SG_DS_RD, 1); }
#pragma unroll
            for (int q = 8; q < 28; ++q) { SGB(SG_MFMA, 1); SGB(SG_DS_RD, 1); }
#pragma unroll
            for (int q = 28; q < 36; ++q) { SGB(SG_MFMA, 1); SGB(SG_VALU, 16); SGB(SG_VMEM_RD, 1); }
#pragma unroll
            for (int q = 36; q < 64; ++q) SGB(SG_MFMA, 1);
            __syncthreads();
        }
    }
    // partial slab layout = HWIO: [slice][tap][ci][co]; tile (i, j) of this wave = channels 2*row + i, 2*col + j
    float* dst = p.partial + ((size_t)bz * p.R * p.S + tap) * p.Cin * p.Cout;
    const int co = co0 + wn * 64 + 2 * li;
    if (co < p.Cout) {                               // cout is a multiple of 4: the pair is inside together
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + wk * 64 + 2 * (4 * lh + (e & 3) + 8 * (e >> 2)) + i;
                if (ci < p.Cin) {
                    f32x2 v; v[0] = acc[i][0][e]; v[1] = acc[i][1][e];
                    *reinterpret_cast<f32x2*>(dst + (size_t)ci * p.Cout + co) = v;
                }
            }
    }
}

// Weight gradient on the bf16 matrix cores (mixed-precision training): x and g arrive in bf16, [pixel][channel] as
// they lie in NHWC.  The reduction index is the PIXEL, i.e. both MFMA operands are k-strided in memory; they are
// staged untransposed (coalesced 16-byte copies, 192-byte LDS rows) and read back with ds_read_b64_tr_b16, the
// gfx950 transposing LDS read: a 16-lane group fetches 4 pixel rows x 16 channels and each lane receives one
// channel's 4 pixels, so two reads form the 8-pixel operand of v_mfma_f32_32x32x16_bf16.  The 192-byte row stride
// puts the 4 rows of a half-wave's block on disjoint 16-bank groups (conflict-free).  Same grid, slabs and
// fixed-order reduction as the f32 kernel; 16x its MFMA rate.
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
constexpr int WB_MC = 64;                 // pixels per staged chunk
constexpr int WB_ROW = 192;               // LDS row stride in bytes (128 B of channels + 64 B)

__device__ __forceinline__ void wgrad_body_bf16(const WgradArgs& p, int bx, int by, int bz) {
    __shared__ __attribute__((aligned(16))) char Xs[2][WB_MC][WB_ROW];
    __shared__ __attribute__((aligned(16))) char Gs[2][WB_MC][WB_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int ci_tiles = (p.Cin + 63) / 64;
    const int tap = bx / ci_tiles, ci0 = (bx % ci_tiles) * 64;
    const int r_tap = tap / p.S, s_tap = tap % p.S;
    const int co0 = by * 64;
    const int m_begin = bz * p.m_per_slice, m_end = min(p.M, m_begin + p.m_per_slice);
    const __bf16* xg = reinterpret_cast<const __bf16*>(p.x);
    const __bf16* gg = reinterpret_cast<const __bf16*>(p.g);

    // staging: 256 threads move 64 pixels x 64 channels (8 x 16 B per pixel) per operand per chunk, two passes of 32 rows
    const int srow = tid >> 3, scol = (tid & 7) * 8;       // channel offset inside the 64-wide tile
    i32x4 rx[2], rg[2];
    auto load = [&](int mc) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = mc + srow + 32 * q;
            i32x4 vx = {0, 0, 0, 0}, vg = {0, 0, 0, 0};
            if (m < m_end) {
                const int wo = m % p.Wo, t = m / p.Wo, ho = t % p.Ho, img = t / p.Ho;
                const int hi = ho * p.stride - p.pad_top + r_tap, wi = wo * p.stride - p.pad_left + s_tap;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && ci0 + scol < p.Cin)
                    vx = *reinterpret_cast<const i32x4*>(xg + (((size_t)img * p.H + hi) * p.W + wi) * p.Cin + ci0 + scol);
                if (co0 + scol < p.Cout) vg = *reinterpret_cast<const i32x4*>(gg + (size_t)m * p.Cout + co0 + scol);
            }
            rx[q] = vx; rg[q] = vg;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<i32x4*>(&Xs[buf][srow + 32 * q][scol * 2]) = rx[q];
            *reinterpret_cast<i32x4*>(&Gs[buf][srow + 32 * q][scol * 2]) = rg[q];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;

    // transposed-read addressing: lane 4q+p of a 16-lane group supplies (row q, channels 4p..4p+3) of its block;
    // lanes 0-15 / 16-31 take channels 0-15 / 16-31 of the wave's 32, lanes 32-63 the next 8 pixels (lh)
    const int g16 = lane & 15, tq = g16 >> 2, tp = g16 & 3, cblk = ((lane >> 4) & 1) * 16;
    const int a_byte = (8 * lh + tq) * WB_ROW + (wk * 32 + cblk + 4 * tp) * 2;
    const int b_byte = (8 * lh + tq) * WB_ROW + (wn * 32 + cblk + 4 * tp) * 2;
    typedef i16x4 __attribute__((address_space(3))) * lds_i16x4;

    const int n_chunks = (m_end - m_begin + WB_MC - 1) / WB_MC;
    if (n_chunks > 0) {
        load(m_begin);
        store(0);
        __syncthreads();
        for (int c = 0; c < n_chunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < n_chunks) load(m_begin + (c + 1) * WB_MC);
            const char* xa = &Xs[buf][0][0] + a_byte;
            const char* gb = &Gs[buf][0][0] + b_byte;
#pragma unroll
            for (int st = 0; st < WB_MC / 16; ++st) {       // 16 pixels per MFMA
                const i16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)(xa + (16 * st) * WB_ROW));
                const i16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)(xa + (16 * st + 4) * WB_ROW));
                const i16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)(gb + (16 * st) * WB_ROW));
                const i16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)(gb + (16 * st + 4) * WB_ROW));
                const i16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                const i16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8w, av), __builtin_bit_cast(bf16x8w, bv), acc, 0, 0, 0);
            }
            if (c + 1 < n_chunks) store(buf ^ 1);
            __syncthreads();
        }
    }
    const int co = co0 + wn * 32 + li;
    if (co < p.Cout) {
        float* dst = p.partial + ((size_t)bz * p.R * p.S + tap) * p.Cin * p.Cout;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int ci = ci0 + wk * 32 + 4 * lh + (e & 3) + 8 * (e >> 2);
            if (ci < p.Cin) dst[(size_t)ci * p.Cout + co] = acc[e];
        }
    }
}

__global__ void __launch_bounds__(256) k_conv_wgrad_bf16(const WgradArgs p) {
    wgrad_body_bf16(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Weight gradient of an f32 layer on the bf16 matrix cores by exact three-way operand splitting (the forward engine of
// conv_x6.hip, for dW = X^T . G): BOTH operands arrive in f32 ([pixel][channel] as they lie in NHWC) and are split as they move
// into LDS -- x = x1 + x2 + x3 with bf16 pieces, each subtraction exact -- into three planes of 32 pixels x 128 channels
// (320-byte rows: the four pixel rows of a transposing read land on disjoint 16-bank groups).  The reduction index is the pixel,
// so fragments come back through ds_read_b64_tr_b16 as in the bf16 body above; the six partial products with i + j <= 4 go
// through v_mfma_f32_32x32x16_bf16, smallest first, into the same f32 accumulators.  Tile, grid, slabs, slice boundaries and
// the fixed-order reduction are those of the 128x128 f32 body (kind 3); ONE LDS buffer of 60 KB (two workgroups per CU), the
// next chunk's operands wait in registers.  Error against fp64: the native f32 kernel's level (tests/test_conv_bwd_gpu.py).
constexpr int WX_ROW = 320;               // LDS row stride in bytes: 128 channels x 2 B + 64
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wg_split3(const f32x4 v, i32x2& h, i32x2& m, i32x2& l) {
    typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
    bf16x4s hh, mm, ll;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hh[e] = (__bf16)v[e];
        const float r1 = v[e] - (float)hh[e];
        mm[e] = (__bf16)r1;
        ll[e] = (__bf16)(r1 - (float)mm[e]);
    }
    h = __builtin_bit_cast(i32x2, hh); m = __builtin_bit_cast(i32x2, mm); l = __builtin_bit_cast(i32x2, ll);
}

__device__ __forceinline__ void wgrad_body_x6_big(const WgradArgs& p, int bx, int by, int bz) {
    __shared__ __attribute__((aligned(16))) char Xp[3][WG_MC][WX_ROW];
    __shared__ __attribute__((aligned(16))) char Gp[3][WG_MC][WX_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int ci_tiles = (p.Cin + 127) / 128;
    const int tap = bx / ci_tiles, ci0 = (bx % ci_tiles) * 128;
    const int r_tap = tap / p.S, s_tap = tap % p.S;
    const int co0 = by * 128;
    const int m_begin = bz * p.m_per_slice, m_end = min(p.M, m_begin + p.m_per_slice);

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.g), 0, (int)((size_t)p.M * p.Cout * 4), 0x00020000);

    // staging: 256 threads move 32 pixels x 128 channels per operand per chunk: 8 rows per pass, 4 passes (the f32 body's walk)
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    const bool ci_ok = ci0 + scol < p.Cin, co_ok = co0 + scol < p.Cout;
    const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
    auto divmod = [](int n, int d, float inv, int& q, int& r) {
        q = (int)((float)n * inv); r = n - q * d;
        if (r < 0) { r += d; --q; }
        if (r >= d) { r -= d; ++q; }
    };
    int mc = m_begin;
    i32x4 rx[4], rg[4];
    auto load = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mc + srow + 8 * q;
            int wo, t, ho, img;
            divmod(m, p.Wo, inv_wo, t, wo);
            divmod(t, p.Ho, inv_ho, img, ho);
            const int hi = ho * p.stride - p.pad_top + r_tap, wi = wo * p.stride - p.pad_left + s_tap;
            const bool in = m < m_end && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned xoff = (unsigned)(((img * p.H + hi) * p.W + wi) * p.Cin + ci0 + scol) * 4u;
            rx[q] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, in && ci_ok ? xoff : OOB_OFFSET, 0, 0);
            const unsigned goff = (unsigned)(m * p.Cout + co0 + scol) * 4u;
            rg[q] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, m < m_end && co_ok ? goff : OOB_OFFSET, 0, 0);
        }
        mc += WG_MC;
    };
    auto store = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            i32x2 h, m, l;
            wg_split3(__builtin_bit_cast(f32x4, rx[q]), h, m, l);
            *reinterpret_cast<i32x2*>(&Xp[0][srow + 8 * q][scol * 2]) = h;
            *reinterpret_cast<i32x2*>(&Xp[1][srow + 8 * q][scol * 2]) = m;
            *reinterpret_cast<i32x2*>(&Xp[2][srow + 8 * q][scol * 2]) = l;
            wg_split3(__builtin_bit_cast(f32x4, rg[q]), h, m, l);
            *reinterpret_cast<i32x2*>(&Gp[0][srow + 8 * q][scol * 2]) = h;
            *reinterpret_cast<i32x2*>(&Gp[1][srow + 8 * q][scol * 2]) = m;
            *reinterpret_cast<i32x2*>(&Gp[2][srow + 8 * q][scol * 2]) = l;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // transposed-read addressing (wgrad_body_bf16): lane 4q+p of a 16-lane group supplies (pixel row q, channels 4p..4p+3) of its
    // block; lanes 0-15 / 16-31 take channels 0-15 / 16-31 of a 32-channel tile, lanes 32-63 the next 8 pixels (lh)
    const int g16 = lane & 15, tq = g16 >> 2, tp = g16 & 3, cblk = ((lane >> 4) & 1) * 16;
    const int a_byte = (8 * lh + tq) * WX_ROW + (wk * 64 + cblk + 4 * tp) * 2;
    const int b_byte = (8 * lh + tq) * WX_ROW + (wn * 64 + cblk + 4 * tp) * 2;
    typedef i16x4 __attribute__((address_space(3))) * lds_i16x4;
    auto frag = [&](const char* base) {
        const i16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)base);
        const i16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)(base + 4 * WX_ROW));
        return __builtin_bit_cast(bf16x8w, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    const int n_chunks = (m_end - m_begin + WG_MC - 1) / WG_MC;
    if (n_chunks > 0) {
        load();
        store();
        load();                                              // chunk 1 (zeros past the slice)
        __syncthreads();
        for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
            for (int st = 0; st < WG_MC / 16; ++st) {        // 16 pixels per MFMA
                bf16x8w fa[3][2], fb[3][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        fa[pl][i] = frag(&Xp[pl][0][0] + a_byte + (16 * st) * WX_ROW + i * 64);
                        fb[pl][i] = frag(&Gp[pl][0][0] + b_byte + (16 * st) * WX_ROW + i * 64);
                    }
                constexpr int IA[6] = {2, 0, 1, 1, 0, 0}, IB[6] = {0, 2, 1, 0, 1, 0};      // smallest terms first
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[IA[t]][i], fb[IB[t]][j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();                                 // everybody is done reading chunk c
            if (c + 1 < n_chunks) {
                store();                                     // chunk c + 1 (in registers since the previous iteration)
                load();                                      // chunk c + 2
            }
            __syncthreads();
        }
    }
    // partial slab layout = HWIO: [slice][tap][ci][co]; tile (i, j) of this wave = channels 32 i + row, 32 j + column
    float* dst = p.partial + ((size_t)bz * p.R * p.S + tap) * p.Cin * p.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + wn * 64 + 32 * j + li;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + wk * 64 + 32 * i + 4 * lh + (e & 3) + 8 * (e >> 2);
                if (ci < p.Cin) dst[(size_t)ci * p.Cout + co] = acc[i][j][e];
            }
    }
}

// The 128 (ci) x 128 (co) form of the bf16 weight gradient (mixed-precision steps; layers with cin, cout >= 128): the tile and
// wave layout of the split-engine body above with ONE plane per operand -- each wave owns 64 x 64, eight transposing reads feed four
// MFMAs per 16 pixels (the 64x64 body: four reads per MFMA) -- 64 pixels per chunk in ONE 40 KB LDS buffer, the next chunk's
// operands waiting in registers.  Slabs, slice boundaries and the fixed-order reduction as everywhere.
constexpr int WB2_MC = 64;
__device__ __forceinline__ void wgrad_body_bf16_big(const WgradArgs& p, int bx, int by, int bz) {
    __shared__ __attribute__((aligned(16))) char Xb[WB2_MC][WX_ROW];
    __shared__ __attribute__((aligned(16))) char Gb[WB2_MC][WX_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    const int ci_tiles = (p.Cin + 127) / 128;
    const int tap = bx / ci_tiles, ci0 = (bx % ci_tiles) * 128;
    const int r_tap = tap / p.S, s_tap = tap % p.S;
    const int co0 = by * 128;
    const int m_begin = bz * p.m_per_slice, m_end = min(p.M, m_begin + p.m_per_slice);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, (int)((size_t)p.n_img * p.H * p.W * p.Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.g), 0, (int)((size_t)p.M * p.Cout * 2), 0x00020000);

    // staging: 256 threads move 64 pixels x 128 channels (16 x 16 B per pixel) per operand per chunk: 16 rows per pass, 4 passes
    const int srow = tid >> 4, scol = (tid & 15) * 8;
    const bool ci_ok = ci0 + scol < p.Cin, co_ok = co0 + scol < p.Cout;
    const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
    auto divmod = [](int n, int d, float inv, int& q, int& r) {
        q = (int)((float)n * inv); r = n - q * d;
        if (r < 0) { r += d; --q; }
        if (r >= d) { r -= d; ++q; }
    };
    int mc = m_begin;
    i32x4 rx[4], rg[4];
    auto load = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = mc + srow + 16 * q;
            int wo, t, ho, img;
            divmod(m, p.Wo, inv_wo, t, wo);
            divmod(t, p.Ho, inv_ho, img, ho);
            const int hi = ho * p.stride - p.pad_top + r_tap, wi = wo * p.stride - p.pad_left + s_tap;
            const bool in = m < m_end && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const unsigned xoff = (unsigned)(((img * p.H + hi) * p.W + wi) * p.Cin + ci0 + scol) * 2u;
            rx[q] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, in && ci_ok ? xoff : OOB_OFFSET, 0, 0);
            const unsigned goff = (unsigned)(m * p.Cout + co0 + scol) * 2u;
            rg[q] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, m < m_end && co_ok ? goff : OOB_OFFSET, 0, 0);
        }
        mc += WB2_MC;
    };
    auto store = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<i32x4*>(&Xb[srow + 16 * q][scol * 2]) = rx[q];
            *reinterpret_cast<i32x4*>(&Gb[srow + 16 * q][scol * 2]) = rg[q];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int g16 = lane & 15, tq = g16 >> 2, tp = g16 & 3, cblk = ((lane >> 4) & 1) * 16;
    const int a_byte = (8 * lh + tq) * WX_ROW + (wk * 64 + cblk + 4 * tp) * 2;
    const int b_byte = (8 * lh + tq) * WX_ROW + (wn * 64 + cblk + 4 * tp) * 2;
    typedef i16x4 __attribute__((address_space(3))) * lds_i16x4;
    auto frag = [&](const char* base) {
        const i16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)base);
        const i16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4)(base + 4 * WX_ROW));
        return __builtin_bit_cast(bf16x8w, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    const int n_chunks = (m_end - m_begin + WB2_MC - 1) / WB2_MC;
    if (n_chunks > 0) {
        load();
        store();
        load();
        __syncthreads();
        for (int c = 0; c < n_chunks; ++c) {
#pragma unroll
            for (int st = 0; st < WB2_MC / 16; ++st) {
                bf16x8w fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[i] = frag(&Xb[0][0] + a_byte + (16 * st) * WX_ROW + i * 64);
                    fb[i] = frag(&Gb[0][0] + b_byte + (16 * st) * WX_ROW + i * 64);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (c + 1 < n_chunks) {
                store();
                load();
            }
            __syncthreads();
        }
    }
    float* dst = p.partial + ((size_t)bz * p.R * p.S + tap) * p.Cin * p.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + wn * 64 + 32 * j + li;
        if (co >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ci = ci0 + wk * 64 + 32 * i + 4 * lh + (e & 3) + 8 * (e >> 2);
                if (ci < p.Cin) dst[(size_t)ci * p.Cout + co] = acc[i][j][e];
            }
    }
}

__global__ void __launch_bounds__(256) k_conv_wgrad_bf16_big(const WgradArgs p) {
    wgrad_body_bf16_big(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// dW = s[co] * sum over slices (fixed order); dbias[co] handled by k_colsum
__global__ void k_wgrad_reduce(const float* partial, int slices, size_t elems, int Cout, const float* scale, float* dw) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.0f;
        for (int sidx = 0; sidx < slices; ++sidx) v += partial[(size_t)sidx * elems + i];
        dw[i] = scale ? v * scale[i % Cout] : v;
    }
}

// ------------------------------------------------------------------------------------
// The weight gradients of ALL trainable layers of a training step in one launch per operand kind, plus one launch for
// their slice reductions.  A step of the RPN / detector model has 22 / 32 trainable convolutions whose weight
// gradients are tiny GEMMs (stage 4: 1.25 - 2.8 GFLOP over 2 394 pixels): launched one by one, each paid its own
// ramp and tail (2 launches per layer, 28 us + 19 us on average for the 1x1 layers, the chip half empty) -- 1.4 ms of a
// 3.9 ms step.  Nothing reads a weight gradient before the optimiser, so the host queues them during the backward
// pass and issues them together: ~30 000 workgroups of mixed shapes keep every CU's four slots turning over.  The
// job table rides in the kernel arguments; a workgroup finds its job by scanning the block prefix.  Arithmetic, slab
// layout and the fixed slice order are those of the single-layer kernels: results are bit-identical.
constexpr int WGRAD_BATCH = 32;
struct WgradBatch { WgradArgs job[WGRAD_BATCH]; int first_block[WGRAD_BATCH + 1]; int gx[WGRAD_BATCH]; int gy[WGRAD_BATCH]; int n; };
struct WgradReduceJob { const float* partial; const float* scale; float* dw; unsigned long long elems; int slices, cout; };
struct WgradReduceBatch { WgradReduceJob job[2 * WGRAD_BATCH]; int first_block[2 * WGRAD_BATCH + 1]; int n; };   // workgroups in proportion to job size

template <int KIND>          // 0: f32 operands; 1: bf16 operands on the bf16 MFMA; 2: bf16 operands widened onto the f32 MFMA; 3: f32, 128x128 tiles; 4: f32 operands split onto the bf16 MFMA, 128x128 tiles; 5: bf16 operands, 128x128 tiles
__global__ void __launch_bounds__(256) k_conv_wgrad_batch(const WgradBatch t) {
    int j = 0;
    while (j + 1 < t.n && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const int local = (int)blockIdx.x - t.first_block[j];
    const int bx = local % t.gx[j], r = local / t.gx[j], by = r % t.gy[j], bz = r / t.gy[j];
    if constexpr (KIND == 1) wgrad_body_bf16(t.job[j], bx, by, bz);
    else if constexpr (KIND == 3) wgrad_body_f32_big(t.job[j], bx, by, bz);
    else if constexpr (KIND == 4) wgrad_body_x6_big(t.job[j], bx, by, bz);
    else if constexpr (KIND == 5) wgrad_body_bf16_big(t.job[j], bx, by, bz);
    else wgrad_body_f32<KIND == 2>(t.job[j], bx, by, bz);
}

__global__ void __launch_bounds__(256) k_conv_wgrad_f32_big(const WgradArgs p) {
    wgrad_body_f32_big(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

__global__ void __launch_bounds__(256) k_conv_wgrad_x6_big(const WgradArgs p) {
    wgrad_body_x6_big(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

__global__ void __launch_bounds__(256) k_wgrad_reduce_batch(const WgradReduceBatch t) {
    int ji = 0;
    while (ji + 1 < t.n && (int)blockIdx.x >= t.first_block[ji + 1]) ++ji;
    const WgradReduceJob& j = t.job[ji];
    const size_t bx = (size_t)((int)blockIdx.x - t.first_block[ji]), gsz = (size_t)(t.first_block[ji + 1] - t.first_block[ji]);
    for (size_t i = bx * blockDim.x + threadIdx.x; i < j.elems; i += gsz * blockDim.x) {
        float v = 0.0f;
        for (int sidx = 0; sidx < j.slices; ++sidx) v += j.partial[(size_t)sidx * j.elems + i];
        j.dw[i] = j.scale ? v * j.scale[i % j.cout] : v;
    }
}

// dbias[co] = s[co] * sum_m G[m][co], two stages: (64 columns x 1 row slice) per workgroup into a
// partial table, then a fixed-order sum over the slices (reproducible).
constexpr int COLSUM_SLICES = 64;
__global__ void __launch_bounds__(256) k_colsum_partial(const float* g, int M, int Cout, int rows_per_slice, float* partial) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = blockIdx.x * 64 + lane;
    const int m0 = blockIdx.y * rows_per_slice, m1 = min(M, m0 + rows_per_slice);
    float v = 0.0f;
    if (co < Cout) for (int m = m0 + wave; m < m1; m += 4) v += g[(size_t)m * Cout + co];
    part[wave][lane] = v;
    __syncthreads();
    if (wave == 0 && co < Cout) partial[(size_t)blockIdx.y * Cout + co] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
__global__ void k_colsum_final(const float* partial, int slices, int Cout, const float* scale, float* out) {
    const int co = blockIdx.x * blockDim.x + threadIdx.x;
    if (co >= Cout) return;
    float t = 0.0f;
    for (int sidx = 0; sidx < slices; ++sidx) t += partial[(size_t)sidx * Cout + co];
    out[co] = scale ? t * scale[co] : t;
}

// ------------------------------------------------------------------------------------
// pooling (NHWC, VALID): MaxPooling2D (resnet.py:412, vgg.py:100-128) / AveragePooling2D (resnet.py:515)
template <bool IS_MAX>
__global__ void k_pool(const float4* x, int n_img, int H, int W, int C4, int k, int stride, int Ho, int Wo, float4* y) {
    const size_t total = (size_t)n_img * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int img = (int)(t / Ho);
        const float4* base = x + (((size_t)img * H + ho * stride) * W + wo * stride) * C4 + c;
        float4 acc = IS_MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0, 0, 0, 0);
        for (int r = 0; r < k; ++r)
            for (int s = 0; s < k; ++s) {
                const float4 v = base[((size_t)r * W + s) * C4];
                if (IS_MAX) { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
                else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            }
        if (!IS_MAX) { const float inv = (float)(k * k); acc.x /= inv; acc.y /= inv; acc.z /= inv; acc.w /= inv; }
        y[i] = acc;
    }
}

// k_pool writing its result as the f16x3 engine's two fp16 planes (hi, lo under the scale 2^*pexp) for the convolution behind it: a window's
// maximum / mean cannot exceed the largest |input|, so the scale comes from the INPUT's magnitude record before the launch
// (frcnn_amax_merge), as for the RoI resampling (roi.hip k_roi_fwd_planes).  VGG's block<n>_conv1 layers then stage their input unchanged.
template <bool IS_MAX>
__global__ void __launch_bounds__(256) k_pool_planes(const float4* x, int n_img, int H, int W, int C4, int k, int stride, int Ho, int Wo,
                                                     const int* pexp, unsigned* status, _Float16* planes, size_t plane_elems) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1u);         // MODE.FP16_OVFL (conv_f32_common.h, the engine's fences)
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const size_t total = (size_t)n_img * Ho * Wo * C4;
    const unsigned sb = (unsigned)(*pexp + 127) << 23;
    float sc;
    __builtin_memcpy(&sc, &sb, 4);
    unsigned seen = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t t = i / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int img = (int)(t / Ho);
        const float4* base = x + (((size_t)img * H + ho * stride) * W + wo * stride) * C4 + c;
        float4 acc = IS_MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0, 0, 0, 0);
        for (int r = 0; r < k; ++r)
            for (int s = 0; s < k; ++s) {
                const float4 v = base[((size_t)r * W + s) * C4];
                if (IS_MAX) { acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w); }
                else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            }
        if (!IS_MAX) { const float inv = (float)(k * k); acc.x /= inv; acc.y /= inv; acc.z /= inv; acc.w /= inv; }
        const float xs[4] = {acc.x * sc, acc.y * sc, acc.z * sc, acc.w * sc};
        f16x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const _Float16 a1 = (_Float16)xs[q];
            h[q] = a1; l[q] = (_Float16)((xs[q] - (float)a1) * 2048.0f);
            unsigned short hb;
            __builtin_memcpy(&hb, &a1, 2);
            seen = seen > (unsigned)(hb & 0x7fffu) ? seen : (unsigned)(hb & 0x7fffu);
        }
        reinterpret_cast<f16x4*>(planes)[i] = h;
        reinterpret_cast<f16x4*>(planes + plane_elems)[i] = l;
    }
    if (status) {
#pragma unroll
        for (int o = 32; o; o >>= 1) { const unsigned t = __shfl_xor(seen, o); seen = seen > t ? seen : t; }
        const unsigned bits = seen > 0x7bffu ? 7u : seen == 0x7bffu ? 3u : seen >= 0x7800u ? 1u : 0u;
        if ((threadIdx.x & 63) == 0 && bits) atomicOr(status, bits);
    }
}

// AveragePooling2D over ALL positions of position-major tensors x[pos][img][c] (frcnn_conv_desc.layout == 1):
// y[img][c] = (sum over pos, in raster order) / npos -- the same additions and the same division as k_pool<false>
// performs on the NHWC tensor, so the result is bit-identical.
__global__ void k_avgpool_pos_major(const float4* x, int npos, int n_img, int C4, float4* y) {
    const size_t total = (size_t)n_img * C4, plane = total;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0, 0, 0, 0);
        for (int q = 0; q < npos; ++q) { const float4 v = x[(size_t)q * plane + i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        const float inv = (float)npos;
        acc.x /= inv; acc.y /= inv; acc.z /= inv; acc.w /= inv;
        y[i] = acc;
    }
}

// row softmax over the first `cols` entries of each row (Dense(..., activation='softmax'), resnet.py:522)
__global__ void k_softmax_rows(const float* x, int rows, int cols, int ldx, float* y, int ldy) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * ldx;
    float mx = -INFINITY;
    for (int c = 0; c < cols; ++c) mx = fmaxf(mx, xr[c]);
    float sum = 0.0f;
    for (int c = 0; c < cols; ++c) sum += expf(xr[c] - mx);
    for (int c = 0; c < cols; ++c) y[(size_t)r * ldy + c] = expf(xr[c] - mx) / sum;
}

// The two dense heads of the detector run as ONE GEMM (kernels concatenated along the output axis): this splits its
// rows back into dense_class_C (softmax over the first `cols` entries, exactly k_softmax_rows) and dense_reg_C (the
// remaining `tail` entries, copied) -- resnet.py:522-533, vgg.py:241-247.
// Round 6: 32 lanes per row (one thread per row walked its 21 + 80 columns alone: 32 us for the 64 rows of a training step, on the
// step's critical path).  The arithmetic is the one-thread loop's, bit for bit: the maximum is order-independent, every lane adds
// e_0, e_1, ... in column order (the other lanes' values arrive by shuffle), the quotients and the copy are per column.
__global__ void __launch_bounds__(256) k_dense_heads_split(const float* x, int rows, int cols, int tail, int ldx, float* cls, float* reg) {
    const int lane = threadIdx.x & 31, r = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (r >= rows) return;                                   // (a whole 32-lane group leaves together)
    const float* xr = x + (size_t)r * ldx;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 32) mx = fmaxf(mx, xr[c]);
#pragma unroll
    for (int o = 16; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
    float sum = 0.0f;
    for (int c0 = 0; c0 < cols; c0 += 32) {
        const float e = c0 + lane < cols ? expf(xr[c0 + lane] - mx) : 0.0f;
        const int cnt = cols - c0 < 32 ? cols - c0 : 32;
        for (int k = 0; k < cnt; ++k) sum += __shfl(e, k, 32);
    }
    for (int c = lane; c < cols; c += 32) cls[(size_t)r * cols + c] = expf(xr[c] - mx) / sum;
    for (int c = lane; c < tail; c += 32) reg[(size_t)r * tail + c] = xr[cols + c];
}

template <int TM, int TN, bool G>
static int launch_conv(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    ConvArgs p = a;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_f32<TM, TN, G>, lds, "conv2d")) return e;
    k_conv_igemm_f32<TM, TN, G><<<p.tiles_m * p.tiles_n, 256, lds, s>>>(p);
    return check_launch("conv2d_fwd");
}

template <int TM, int TN, int VARIANT = 0, int WM = 2, int WN = 2>
static int launch_conv_v2(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    ConvArgs p = a;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_f32_v2<TM, TN, VARIANT, WM, WN>, lds, "conv2d")) return e;
    k_conv_igemm_f32_v2<TM, TN, VARIANT, WM, WN><<<p.tiles_m * p.tiles_n, 64 * WM * WN, lds, s>>>(p);
    return check_launch("conv2d_fwd");
}

static int launch_conv_cin3(const ConvArgs& a, hipStream_t s, bool variant2) {
    ConvArgs p = a;
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = (p.Cout + 63) / 64;
    const size_t lds = (size_t)2 * (64 + 64) * LDS_STRIDE * sizeof(float);
    if (variant2) k_conv_igemm_f32_v2<1, 1, 2, 2, 2, false, true><<<p.tiles_m * p.tiles_n, 256, lds, s>>>(p);
    else k_conv_igemm_f32_v2<1, 1, 1, 2, 2, false, true><<<p.tiles_m * p.tiles_n, 256, lds, s>>>(p);
    return check_launch("conv2d_fwd (3-channel stem)");
}

constexpr size_t SPLITK_TICKET_BYTES = 16384;      // head of the workspace: one u32 per output tile (<= 4096 tiles)

template <int TM, int TN, int VARIANT = 0, int WM = 2, int WN = 2>
static int launch_conv_v2_splitk(const ConvArgs& a, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    ConvArgs p = a;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float);
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_conv_igemm_f32_v2<TM, TN, VARIANT, WM, WN, true>, lds, "conv2d")) return e;
    k_conv_igemm_f32_v2<TM, TN, VARIANT, WM, WN, true><<<p.tiles_m * p.tiles_n * p.splits, 64 * WM * WN, lds, s>>>(p);
    return check_launch("conv2d_fwd (split-K)");
}

}  // namespace frcnn

using namespace frcnn;

static bool g_scalar_epilogue = getenv("FRCNN_SCALAR_EPILOGUE") != nullptr;      // dev knob: the 4-byte epilogue everywhere
static int g_group_m = getenv("FRCNN_GROUP_M") ? atoi(getenv("FRCNN_GROUP_M")) : -1;   // dev knob: tile-order group size (-1 = auto)

// tile / main-loop selection shared by frcnn_conv2d_fwd and frcnn_conv2d_config
static int choose_streamk(const frcnn_conv_desc* d, int cfg);

// what the two-layer launch (frcnn_conv2d_fwd_dual) makes of a single-layer tile choice
static int dual_config(int cfg) {
    if (cfg >= 61) cfg -= 40;                               // no balanced form for the two-layer launch
    if (cfg < 11 || (cfg >= 41 && cfg <= 43)) cfg = 23;     // v2 main loops with the 2x2-wave tiles only
    return cfg;
}

static int choose_config(const frcnn_conv_desc* d) {
    const long long M = (long long)d->n * d->ho * d->wo;
    const int Kpad = (d->kh * d->kw * d->cin + BK - 1) / BK * BK;
    const bool generic = (d->cin % BK) != 0;
    const long long t128 = ((M + 127) / 128) * ((d->cout + 127) / 128);
    int cfg = d->tile % 100;    // 0 = auto; the hundreds digit(s) force the split-K factor (choose_splits)
    static const int forced = getenv("FRCNN_FORCE_TILE") ? atoi(getenv("FRCNN_FORCE_TILE")) : 0;   // dev knob
    if (cfg == 0 && forced && !generic) cfg = forced;
    const bool shared_chip = cfg == 50;                     // "auto, other launches run beside this one" (several images in flight)
    if (cfg == 50) cfg = 0;
    if (cfg == 0) {
        // measured on MI355X over every conv shape of the C2 pipeline (scripts/conv_shapes.py):
        // the 64x64 v2 kernel wins wherever the grid is small or k is short; 128x128 v2 only
        // pays once there are >= 1.5 tiles per CU slot AND a long k loop to amortise its prologue
        // position-major multi-tap layers skip padding-only taps per tile: tiles then differ in length, and only
        // a grid with several tiles per CU slot (64x64: 1840 tiles for the head 3x3) turns that into a shorter
        // launch (500 vs 570 us); the 460 128x128 tiles all start at once and the full-length ones set the time
        // (with several images in flight the neighbours fill the freed slots: pipelines then ask for tile 21)
        // round 2 (scripts/micro/conv_lab.hip): the mid-chunk-barrier main loop (23 / 26) beats the late-store loop on
        // every 64x64 launch (trunk + RPN head 1 355 -> 1 269 us per image, bit-identical) and on the 1x1 big-tile
        // launches (2048->512: 253 -> 246 us); the 3x3 big-tile launches keep the late-store loop (528 vs 539 us)
        if (generic) cfg = 2;
        else if (d->layout && d->kh * d->kw > 1 && !shared_chip) cfg = choose_streamk(d, 21) ? 21 : 23;   // balanced 128x128 beats both
        else if (t128 >= 384 && Kpad >= 1024) cfg = d->kh * d->kw == 1 ? 26 : 21;
        else cfg = 23;
    }
    if (d->layout && cfg >= 1 && cfg <= 4) cfg += 10;       // only the v2 main loop knows the position-major layout
    const bool fits_srd = (size_t)d->n * d->h * d->w * d->cin * 4 < 0x7fffffffull && (size_t)d->cout * Kpad * 4 < 0x7fffffffull;
    // the v2 main loops walk the filter taps through a 32-bit mask: larger filters (6x6 and up) stay on the v1 kernels
    const bool v1_only = !fits_srd || generic || d->kh * d->kw > 32;
    if (cfg >= 61 && v1_only) cfg -= 60;
    if (cfg >= 41 && v1_only) cfg = (cfg == 43) ? 3 : 1;
    if (cfg >= 23 && cfg <= 26 && v1_only) cfg = cfg >= 25 ? 1 : 2;
    if (cfg >= 21 && v1_only) cfg -= 20;
    if (cfg >= 11 && v1_only) cfg -= 10;
    if (generic) cfg = (cfg == 2) ? 2 : 3;
    return cfg;
}

// K-slices per output tile for the 64x64 kernel (1 = plain launch).  desc.tile / 100 forces a value (dev knob).
static int choose_splits(const frcnn_conv_desc* d, int cfg) {
    if (cfg != 22 && cfg != 23) return 1;
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long tiles = ((M + 63) / 64) * ((d->cout + 63) / 64);
    const int nk = (d->kh * d->kw * d->cin + BK - 1) / BK;
    if (tiles * sizeof(unsigned) > SPLITK_TICKET_BYTES) return 1;
    int s = d->tile / 100;
    if (s <= 0) {
        // measured on MI355X (scripts/conv_shapes.py, C2 shapes): grids under 1.5 tiles per CU with >= 16 chunks
        // gain from 3 slices (stage 3/4 3x3 and 1x1-reduce, rpn_conv1: -10..-37 %); tiny grids (RPN heads, dense)
        // take enough slices for ~2 workgroups per CU, at least 4 chunks each; shorter k loops lose to the combine
        // 384..639 tiles with a long k loop (the detector head at 64 training RoIs: 392 tiles, 144 / 64 chunks) fill
        // 38-60 % of the 1024 slots: four slices take 174 -> 143 us (3x3) and 85 -> 76 us (2048 -> 512)
        if (tiles >= 384 && tiles < 640 && nk >= 64) return 4;
        if (tiles >= 384 || nk < 16) return 1;
        // (round 2, mid-chunk-barrier loop: 100..383 tiles with a LONG k loop take five slices -- stage-4 3x3 31.9 -> 30.9 us,
        // rpn_conv1 190.5 -> 178.8 us: 5 x 304 workgroups sit 6-deep on the 256 CUs where 3 x 304 sit 4-deep on some and 3 on others)
        s = tiles >= 100 ? (nk >= 64 ? 5 : 3) : (int)((456 + tiles - 1) / tiles);
        if (s > nk / 4) s = nk / 4;
        if (s > 16) s = 16;
    }
    if (s > nk) s = nk;
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

// Balanced (stream-K) launch: G workgroups for this descriptor, or 0 when the plain / split-K forms are better.
// Auto picks it for the two late-store tiles when the grid wastes >= 6 % of its last round of CU slots, the k loop is
// long enough for the partial-tile traffic not to matter (>= 32 chunks) and no tile can meet more than SK_SLOTS
// ranges.  desc.tile 61 / 62 force it (tests), a hundreds digit (forced split-K factor / "never split") disables it.
static int streamk_tile_taps(const frcnn_conv_desc* d, int tile_m, int BM) {
    const int RS = d->kh * d->kw;
    if (!d->layout) return RS;
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long m0 = (long long)tile_m * BM;
    const long long m1 = (m0 + BM < M ? m0 + BM : M) - 1;
    const int pos_lo = (int)(m0 / d->n), pos_hi = (int)(m1 / d->n);
    if (pos_hi - pos_lo >= 8 || RS > 32) return RS;
    unsigned mk = 0;
    for (int pos = pos_lo; pos <= pos_hi; ++pos) {
        const int ho = pos / d->wo, wo = pos - ho * d->wo;
        const int h0 = ho * d->stride - d->pad_top, w0 = wo * d->stride - d->pad_left;
        for (int r = 0; r < d->kh; ++r)
            for (int sx = 0; sx < d->kw; ++sx)
                if ((unsigned)(h0 + r) < (unsigned)d->h && (unsigned)(w0 + sx) < (unsigned)d->w) mk |= 1u << (r * d->kw + sx);
    }
    return mk ? __builtin_popcount(mk) : RS;
}

static int choose_streamk(const frcnn_conv_desc* d, int cfg) {
    const bool forced = (cfg == 61 || cfg == 62);
    if (!forced && (cfg != 21 && cfg != 22 && cfg != 26)) return 0;
    if (d->tile / 100 != 0 || (d->cin % BK) != 0) return 0;
    const int big = (cfg == 21 || cfg == 26 || cfg == 61);
    const int BM = big ? 128 : 64;
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long tiles_m = (M + BM - 1) / BM, tiles_n = (d->cout + BM - 1) / BM;
    const long long tiles = tiles_m * tiles_n;
    const int slots = 256 * (big ? 2 : 4);                   // workgroups the chip holds at once (LDS: 2 x 74 KB / 4 x 37 KB per CU)
    const int RS = d->kh * d->kw;
    const int groups = d->cin / BK, nk_max = groups * RS;
    if (tiles_m > 1024 || tiles * sizeof(unsigned) > SPLITK_TICKET_BYTES) return 0;
    if ((size_t)tiles * SK_SLOTS * BM * BM * 4 >= 0x7fffffffull) return 0;
    const long long rounds = (tiles + slots - 1) / slots;
    const long long G = rounds * slots;
    if (!forced) {
        // measured (scripts/layout_compare.py, 300 RoIs): on the 128x128 tile the balanced form wins wherever it is
        // eligible (3x3 575 -> 483 us, 2048->512 276 -> 263, 1024->512 154 -> 149); on the 64x64 tile the partial-tile
        // traffic eats the gain (506 -> 520, 280 -> 294), and beside other images' launches (tile 50) the idle slots
        // are already taken: four images in flight run 3 % slower with it
        static const bool sk_shared = getenv("FRCNN_SK_SHARED") != nullptr;       // dev knob: the balanced form beside other images' launches too
        if (!big || (d->tile % 100 == 50 && !sk_shared)) return 0;
        if (nk_max < 32 || tiles * 100 > G * 94 || tiles * 2 < G) return 0;
    }
    long long U = 0;
    for (int m = 0; m < tiles_m; ++m) U += (long long)groups * streamk_tile_taps(d, m, BM);
    U *= tiles_n;
    if (U < G || U / G < nk_max / 2 + 1) return 0;           // a tile would meet more than SK_SLOTS ranges
    return (int)G;
}

template <int TM, int TN>
static int launch_conv_sk(const ConvArgs& a, int G, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    ConvArgs p = a;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const size_t lds = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(float) + (size_t)(p.tiles_m + 1) * sizeof(int);
    static std::atomic<size_t> attr_lds[64];                 // per device ordinal (the limit is a per-device function attribute); grows with tiles_m
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(FRCNN_E_HIP, "conv2d: no current HIP device");
    if (lds > attr_lds[dev & 63].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute((const void*)k_conv_igemm_f32_sk<TM, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return fail(FRCNN_E_HIP, "conv2d: cannot raise dynamic LDS to %zu", lds);
        attr_lds[dev & 63].store(lds, std::memory_order_release);
    }
    k_conv_igemm_f32_sk<TM, TN><<<G, 256, lds, s>>>(p);
    return check_launch("conv2d_fwd (balanced)");
}

extern "C" {

int frcnn_conv_packed_k(int kh, int kw, int cin) { return packed_k(kh * kw, cin); }

int frcnn_pack_conv_weights(const float* w_hwio, int kh, int kw, int cin, int cout, float* packed, void* stream) {
    if (!w_hwio || !packed || kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return fail(FRCNN_E_ARG, "pack_conv_weights: bad argument");
    const int Kpad = frcnn_conv_packed_k(kh, kw, cin);
    const size_t total = (size_t)cout * Kpad;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    k_pack_hwio<<<grid, 256, 0, as_stream(stream)>>>(w_hwio, kh * kw, cin, cout, Kpad, packed);
    return check_launch("pack_conv_weights");
}

int frcnn_conv2d_fwd(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                     const float* scale, const float* shift, const float* residual, float* y, void* stream) {
    return frcnn_conv2d_fwd_masked(d, x, w_packed, scale, shift, residual, nullptr, y, stream);
}

int frcnn_conv2d_fwd_masked(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                            const float* scale, const float* shift, const float* residual, const float* mask, float* y, void* stream) {
    return frcnn_conv2d_fwd_ws(d, x, w_packed, scale, shift, residual, mask, y, nullptr, 0, stream);
}

size_t frcnn_conv2d_workspace_bytes(const frcnn_conv_desc* d) {
    if (!d || d->cin <= 0 || (d->cin % BK) != 0) return 0;
    const int cfg0 = choose_config(d);
    if (choose_streamk(d, cfg0)) {
        const int BM = (cfg0 == 21 || cfg0 == 26 || cfg0 == 61) ? 128 : 64;
        const long long M = (long long)d->n * d->ho * d->wo;
        const size_t tiles = (size_t)((M + BM - 1) / BM) * ((d->cout + BM - 1) / BM);
        return SPLITK_TICKET_BYTES + tiles * SK_SLOTS * BM * BM * sizeof(float);
    }
    const int splits = choose_splits(d, cfg0);
    if (splits <= 1) return 0;
    const long long M = (long long)d->n * d->ho * d->wo;
    const size_t tiles = (size_t)((M + 63) / 64) * ((d->cout + 63) / 64);
    return SPLITK_TICKET_BYTES + tiles * splits * 64 * 64 * sizeof(float);
}

struct DualOut { int n1; int act1; float* y2; int act2; };      // frcnn_conv2d_fwd_dual: the launch's second layer

// which matrix path a forward launch takes, and the magnitude records that ride along (all NULL: nothing is tracked)
enum { ENGINE_NATIVE = 0, ENGINE_X6 = 1, ENGINE_H3 = 2 };
struct ConvRange {
    const float* x_amax; float* y_amax; float* y2_amax;
    // f16x3 engine, activations as fp16 planes (frcnn_conv2d_fwd_h3_planes); all null / 0 otherwise
    const void* x_planes = nullptr; const int* x_pexp = nullptr; void* y_planes = nullptr; int* y_pexp = nullptr; const float* res_amax = nullptr;
    float bound_c = 0.0f, bound_d = 0.0f;
    const void* res_planes = nullptr; const int* res_pexp = nullptr;      // the residual as planes (frcnn_conv2d_fwd_h3_planes_res)
};

static int conv_fwd_impl(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                         const float* scale, const float* shift, const float* residual, const float* mask, float* y,
                         const DualOut* dual, void* workspace, size_t workspace_bytes, void* stream, int engine = ENGINE_NATIVE,
                         const ConvRange* range = nullptr);

// split-K factor of the split-bf16 engine's 64x64 form: grids under ~1.5 tiles per CU slot with a LONG k loop only
// (k >= 2048: rpn_conv1, stage 4's 3x3, the 64-RoI training head); everything else runs unsplit or stays native
// The split engine's tile for a descriptor (n1 > 0: a two-layer launch whose first layer has n1 columns).
// auto: 128x128 tiles on eight waves (two workgroups per CU); long k on a grid of >= 200 256x128 tiles: the 16-wave double-buffered
// form (head 3x3 337 vs 350 us, 2048 -> 512 157 vs 169; 512 -> 2048 ties and stays); under one 128x128 tile per CU, 64x64.
// 64-column layers (a 128-wide tile would be half empty: stage 2's 3x3 48.8 us against 30.9 on 64x64 tiles, 36.3 native): 64x64,
// or 128x64 on four waves once there are >= 1024 of them (VGG16 conv1_2, 600 000 rows: 361 us against 389 / 458 native).
static int x6_config(const frcnn_conv_desc* d, int n1) {
    const int t = d->tile % 100;
    if (t >= 71 && t <= 77) return t;
    const long long M = (long long)d->n * d->ho * d->wo;
    const int K = d->kh * d->kw * d->cin;
    int cfg;
    if (d->cout <= 64) cfg = ((M + 127) / 128) >= 1024 ? 77 : 74;
    else {
        const long long t128 = ((M + 127) / 128) * ((d->cout + 127) / 128), t256 = ((M + 255) / 256) * ((d->cout + 127) / 128);
        cfg = t128 >= 256 ? ((K >= 1024 && t256 >= 200) ? 76 : 71) : 74;
    }
    if (n1 > 0 && (n1 % 128) != 0 && cfg != 77) cfg = 74;       // the layer boundary of a paired launch must be a tile boundary (16-byte epilogue)
    return cfg;
}

// Split-K on the split engine: tile edge (64 or 128) and slices for a descriptor.  64x64 tiles (four waves) fill the chip from the
// smallest grids; from ~64 tiles of 128x128 on, the eight-wave 128x128 tile (nine fragment reads per twelve MFMAs instead of six per
// six) is the better workgroup -- the detector head's 3x3 over 64 RoIs (3 136 rows, k 4 608), rpn_conv1.  tile % 100: 74 / 78 force
// the 64 / 128 form, tile / 100 the slice count (dev).
static int x6_sk_tile(const frcnn_conv_desc* d) {
    const int t = d->tile % 100;
    if (t == 78) return 128;
    if (t == 74) return 64;
    static const long long min128 = getenv("FRCNN_X6_SK128_MIN") ? atoll(getenv("FRCNN_X6_SK128_MIN")) : 64;      // dev knob
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long t128 = ((M + 127) / 128) * ((d->cout + 127) / 128);
    return t128 >= min128 ? 128 : 64;
}

static int choose_splits_x6(const frcnn_conv_desc* d) {
    if (d->cin % BK) return 1;
    const int t = d->tile % 100;
    if (t != 0 && t != 50 && t != 74 && t != 78) return 1;
    const int edge = x6_sk_tile(d);
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long tiles = ((M + edge - 1) / edge) * ((d->cout + edge - 1) / edge);
    const long long tiles64 = ((M + 63) / 64) * ((d->cout + 63) / 64);
    const int nk = (d->kh * d->kw * d->cin) / BK;
    if (tiles * sizeof(unsigned) > SPLITK_TICKET_BYTES) return 1;
    int s = d->tile / 100;
    if (s <= 0) {
        if (tiles64 >= 640 || nk < 64) return 1;
        if (edge == 128) s = (int)(512 / tiles);                     // ONE round of two workgroups per CU: 3 136 x 512 (100 tiles) 94 us at 5 slices, 110 at 4 or 6; rpn_conv1 (76 tiles) 161 at 6, 169 / 175 at 5 / 3
        else s = tiles >= 100 ? 3 : (int)((768 + tiles - 1) / tiles);      // sweep (MI355X): rpn_conv1 (304 tiles) 209 / 192 / 204 / 189 us at 2 / 3 / 4 / 5 slices, stage 4 3x3 (152) 32.6 / 34.2 / 33.4 at 3 / 4 / 6
        if (s > nk / 8) s = nk / 8;
        if (s > 16) s = 16;
    }
    if (s > nk) s = nk;
    return s < 1 ? 1 : s;
}

size_t frcnn_conv2d_x6_workspace_bytes(const frcnn_conv_desc* d) {
    if (!d || d->cin <= 0) return 0;
    const int splits = choose_splits_x6(d);
    if (splits <= 1) return 0;
    const int edge = x6_sk_tile(d);
    const long long M = (long long)d->n * d->ho * d->wo;
    const size_t tiles = (size_t)((M + edge - 1) / edge) * ((d->cout + edge - 1) / edge);
    return SPLITK_TICKET_BYTES + tiles * splits * edge * edge * sizeof(float);
}

// ---- which matrix path a forward launch of this descriptor should take: the measured policy, for hosts in any language.
// prefer: FRCNN_ENGINE_X6 / FRCNN_ENGINE_H3 = the split engine the caller has filter planes for (FRCNN_ENGINE_NATIVE: always native).
// An explicit tile code picks its engine (71..78: bf16x6, 81..88: f16x3).  Otherwise a split engine takes launches with cin % 32 == 0,
// at most 32 taps, >= 64 output columns and >= 256 output tiles of 64x64 (MI355X, configs[1] shapes, each launch alone on the chip:
// scripts/conv_shapes.py -- the head's 14 700-row GEMMs 236 / 148 / 115 us on f16x3, 352 / 208 / 166 on bf16x6, 531 / 282 / 267 native;
// almost every trunk layer wins by 5-15 %); smaller grids stay on the native split-K launches unless the engine's own split-K form
// applies (>= 128 columns, a workspace at hand: rpn_conv1, stage 4's 3x3).
enum { ENGINE_MIN_TILES = 256, ENGINE_MIN_COUT = 64 };

int frcnn_conv2d_engine(const frcnn_conv_desc* d, int prefer, int workspace_present) {
    if (!d) return fail(FRCNN_E_ARG, "conv2d_engine: null descriptor");
    if (prefer != ENGINE_NATIVE && prefer != ENGINE_X6 && prefer != ENGINE_H3) return fail(FRCNN_E_ARG, "conv2d_engine: unknown engine %d", prefer);
    const int t = d->tile % 100;
    const bool splittable = d->cin > 0 && (d->cin % BK) == 0;
    if (t >= 71 && t <= 78) return splittable ? ENGINE_X6 : ENGINE_NATIVE;
    if (t >= 81 && t <= 88) return splittable ? ENGINE_H3 : ENGINE_NATIVE;
    if (prefer == ENGINE_NATIVE || (t != 0 && t != 50)) return ENGINE_NATIVE;
    if (!splittable || d->cout < ENGINE_MIN_COUT || d->kh * d->kw > 32) return ENGINE_NATIVE;
    const long long M = (long long)d->n * d->ho * d->wo;
    if (((M + 63) / 64) * ((d->cout + 63) / 64) >= ENGINE_MIN_TILES) return prefer;
    const size_t need = prefer == ENGINE_X6 ? frcnn_conv2d_x6_workspace_bytes(d) : frcnn_conv2d_h3_workspace_bytes(d);
    return (d->cout >= 128 && workspace_present && need > 0) ? prefer : ENGINE_NATIVE;
}

int frcnn_conv2d_fwd_x6(const frcnn_conv_desc* d, const float* x, const void* w_planes_bf16,
                        const float* scale, const float* shift, const float* residual, const float* mask, float* y,
                        void* workspace, size_t workspace_bytes, void* stream) {
    return conv_fwd_impl(d, x, reinterpret_cast<const float*>(w_planes_bf16), scale, shift, residual, mask, y, nullptr, workspace, workspace_bytes, stream, ENGINE_X6);
}

int frcnn_conv2d_fwd_dual_x6(const frcnn_conv_desc* d, const float* x, const void* w_planes_bf16, const float* scale, const float* shift,
                             float* y1, int n1, int act1, float* y2, int act2, void* stream) {
    if (!d || !y2 || n1 <= 0 || n1 >= d->cout) return fail(FRCNN_E_ARG, "conv2d_fwd_dual_x6: need 0 < n1 < cout and two outputs");
    if (d->ldy > 0 || d->ldres > 0) return fail(FRCNN_E_ARG, "conv2d_fwd_dual_x6: dense outputs only (ldy = ldres = 0)");
    const DualOut dual = {n1, act1, y2, act2};
    return conv_fwd_impl(d, x, reinterpret_cast<const float*>(w_planes_bf16), scale, shift, nullptr, nullptr, y1, &dual, nullptr, 0, stream, ENGINE_X6);
}

// ---- the f16x3 engine (conv_h3.hip).  Tile for a descriptor: the double-buffered 256x128 forms wherever a launch has >= 256 tiles of
// 128x128 (lab: the head's 3x3 / 512 -> 2048 / 2048 -> 512 GEMMs 246 / 127 / 104 us on sixteen waves against 359 / 154 / 138 on the
// two-workgroup 128x128 tile and 331 / 147 / 148 on 64x64 tiles); everything smaller, and every 64-column layer, on 64x64 tiles (128x64 on
// four waves once there are >= 1024 row tiles of 64 columns).
static int h3_config(const frcnn_conv_desc* d, int n1) {
    const int t = d->tile % 100;
    if (t >= 81 && t <= 87) return t;
    const long long M = (long long)d->n * d->ho * d->wo;
    int cfg;
    if (d->cout <= 64) cfg = ((M + 127) / 128) >= 1024 ? 87 : 84;
    else {
        const long long t128 = ((M + 127) / 128) * ((d->cout + 127) / 128);
        // beside other passes' launches the big tile pays from half as many tiles on (scripts/dev/r6_shared_big_min.sh: from 128 / 256 /
        // 512 / 1024 tiles 555.9 / 554.5 / 544.5 / 540.5 img/s): what it leaves idle, other passes fill
        static const int big_min_shared = getenv("FRCNN_H3_BIG_MIN_TILES_SHARED") ? atoi(getenv("FRCNN_H3_BIG_MIN_TILES_SHARED")) : 128;
        cfg = t128 >= (t == 50 ? big_min_shared : 256) ? 86 : 84;
    }
    // Beside other passes' launches (tile code 50: the chip is saturated -- sixteen images per 29 ms against 1.9 ms of isolated conv time
    // per image -- and idle CUs are the other passes' to fill) a launch too small for the 256x128 form does its FLOPs cheaper on 128x128
    // tiles (eight waves, code 81) than on 64x64: stage 4's 256-column layers of a four-image pass, 544.2 -> 549.7 img/s, backbone in
    // flight 0.469 -> 0.458 ms per image (scripts/dev/r6_shared_small.sh; four waves of 64x64, code 83: 544.2).  Alone on the chip the
    // 64x64 tiles stay (150 workgroups of 128x128 leave 106 CUs idle).  Same chunk order: the same bits.  FRCNN_H3_SHARED_SMALL=0: off.
    static const int shared_small = getenv("FRCNN_H3_SHARED_SMALL") ? atoi(getenv("FRCNN_H3_SHARED_SMALL")) : 81;
    static const int shared_small_rows = getenv("FRCNN_H3_SHARED_SMALL_ROWS") ? atoi(getenv("FRCNN_H3_SHARED_SMALL_ROWS")) : 4096;
    if (t == 50 && cfg == 84 && shared_small && d->cout >= 128 && M >= shared_small_rows) cfg = shared_small;
    if (n1 > 0 && (n1 % 128) != 0 && cfg != 87) cfg = 84;       // the layer boundary of a paired launch must be a tile boundary (16-byte epilogue)
    return cfg;
}

// split-K of the f16x3 engine: the split-bf16 engine's rules (tile edge, slices) -- the chunk count per tile is the same
static int h3_sk_tile(const frcnn_conv_desc* d) {
    const int t = d->tile % 100;
    if (t == 88) return 128;
    if (t == 84) return 64;
    // (the eight-wave 128x128 tile needs 154 registers with its two accumulator sets: one workgroup per CU, so the rule that sends
    // the split-bf16 engine's taller small grids there -- ONE round of two workgroups per CU -- does not carry over)
    return 64;
}

static int choose_splits_h3(const frcnn_conv_desc* d) {
    if (d->cin % BK) return 1;
    const int t = d->tile % 100;
    if (t != 0 && t != 50 && t != 84 && t != 88) return 1;
    const int edge = h3_sk_tile(d);
    const long long M = (long long)d->n * d->ho * d->wo;
    const long long tiles = ((M + edge - 1) / edge) * ((d->cout + edge - 1) / edge);
    const long long tiles64 = ((M + 63) / 64) * ((d->cout + 63) / 64);
    const int nk = (d->kh * d->kw * d->cin) / BK;
    if (tiles * sizeof(unsigned) > SPLITK_TICKET_BYTES) return 1;
    int s = d->tile / 100;
    if (s <= 0) {
        if (tiles64 >= 640 || nk < 32 || (nk < 64 && tiles64 >= 256)) return 1;       // (short reductions: only grids that leave most CUs idle -- stage 4's 1x1 1024 -> 256 at 152 tiles: 21.9 us native split-K, 19.0 here)
        if (edge == 128) s = (int)(512 / tiles);
        else s = tiles >= 100 ? 3 : (int)((768 + tiles - 1) / tiles);
        if (s > nk / 8) s = nk / 8;
        if (s > 16) s = 16;
    }
    if (s > nk) s = nk;
    return s < 1 ? 1 : s;
}

size_t frcnn_conv2d_h3_workspace_bytes(const frcnn_conv_desc* d) {
    if (!d || d->cin <= 0) return 0;
    const int splits = choose_splits_h3(d);
    if (splits <= 1) return 0;
    const int edge = h3_sk_tile(d);
    const long long M = (long long)d->n * d->ho * d->wo;
    const size_t tiles = (size_t)((M + edge - 1) / edge) * ((d->cout + edge - 1) / edge);
    return SPLITK_TICKET_BYTES + tiles * splits * edge * edge * sizeof(float);
}

int frcnn_conv2d_h3_config(const frcnn_conv_desc* d, int n1) {
    if (!d) return fail(FRCNN_E_ARG, "conv2d_h3_config: null descriptor");
    return h3_config(d, n1);
}

int frcnn_conv2d_fwd_h3(const frcnn_conv_desc* d, const float* x, const float* x_amax, const void* w_planes_f16,
                        const float* scale, const float* shift, const float* residual, const float* mask, float* y, float* y_amax,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!x_amax) return fail(FRCNN_E_ARG, "conv2d_fwd_h3: the input's magnitude record is required (frcnn_amax_f32 makes one)");
    const ConvRange rg = {x_amax, y_amax, nullptr};
    return conv_fwd_impl(d, x, reinterpret_cast<const float*>(w_planes_f16), scale, shift, residual, mask, y, nullptr, workspace, workspace_bytes, stream, ENGINE_H3, &rg);
}

int frcnn_conv2d_fwd_dual_h3(const frcnn_conv_desc* d, const float* x, const float* x_amax, const void* w_planes_f16, const float* scale, const float* shift,
                             float* y1, int n1, int act1, float* y1_amax, float* y2, int act2, float* y2_amax, void* stream) {
    if (!d || !y2 || n1 <= 0 || n1 >= d->cout) return fail(FRCNN_E_ARG, "conv2d_fwd_dual_h3: need 0 < n1 < cout and two outputs");
    if (d->ldy > 0 || d->ldres > 0) return fail(FRCNN_E_ARG, "conv2d_fwd_dual_h3: dense outputs only (ldy = ldres = 0)");
    if (!x_amax) return fail(FRCNN_E_ARG, "conv2d_fwd_dual_h3: the input's magnitude record is required");
    const DualOut dual = {n1, act1, y2, act2};
    const ConvRange rg = {x_amax, y1_amax, y2_amax};
    return conv_fwd_impl(d, x, reinterpret_cast<const float*>(w_planes_f16), scale, shift, nullptr, nullptr, y1, &dual, nullptr, 0, stream, ENGINE_H3, &rg);
}

int frcnn_conv2d_fwd_h3_planes(const frcnn_conv_desc* d, const float* x, const frcnn_h3_planes* x_planes, const float* x_amax, const void* w_planes_f16,
                               const float* scale, const float* shift, const float* residual, const float* residual_amax,
                               float* y, float* y_amax, const frcnn_h3_planes* y_planes, float bound_c, float bound_d, void* stream) {
    return frcnn_conv2d_fwd_h3_planes_res(d, x, x_planes, x_amax, w_planes_f16, scale, shift, residual, nullptr, residual_amax, y, y_amax, y_planes, bound_c, bound_d, stream);
}

int frcnn_conv2d_fwd_h3_planes_res(const frcnn_conv_desc* d, const float* x, const frcnn_h3_planes* x_planes, const float* x_amax, const void* w_planes_f16,
                                   const float* scale, const float* shift, const float* residual, const frcnn_h3_planes* residual_planes,
                                   const float* residual_amax, float* y, float* y_amax, const frcnn_h3_planes* y_planes, float bound_c, float bound_d,
                                   void* stream) {
    if (!x_amax) return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: the input's magnitude record is required");
    if (residual && residual_planes) return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: the residual as an f32 tensor OR as planes");
    if (residual_planes && (!residual_planes->planes || !residual_planes->exponent || (reinterpret_cast<uintptr_t>(residual_planes->planes) & 15)))
        return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: incomplete or misaligned residual planes");
    if ((x != nullptr) == (x_planes != nullptr)) return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: exactly one of x / x_planes");
    if (!y && !y_planes) return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: no output");
    if (x_planes && (!x_planes->planes || !x_planes->exponent)) return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: incomplete input planes");
    if (y_planes && (!y_planes->planes || !y_planes->exponent || !(bound_c >= 0.0f) || !(bound_d >= 0.0f) || ((residual || residual_planes) && !residual_amax)))
        return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: output planes need their buffers, the filter's bound constants and, with a residual, its magnitude record");
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if ((x_planes && !al16(x_planes->planes)) || (y_planes && !al16(y_planes->planes))) return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: 16-byte aligned planes required");
    ConvRange rg = {x_amax, y_amax, nullptr};
    if (x_planes) { rg.x_planes = x_planes->planes; rg.x_pexp = x_planes->exponent; }
    if (y_planes) { rg.y_planes = y_planes->planes; rg.y_pexp = y_planes->exponent; rg.res_amax = (residual || residual_planes) ? residual_amax : nullptr; rg.bound_c = bound_c; rg.bound_d = bound_d; }
    if (residual_planes) { rg.res_planes = residual_planes->planes; rg.res_pexp = residual_planes->exponent; }
    return conv_fwd_impl(d, x, reinterpret_cast<const float*>(w_planes_f16), scale, shift, residual, nullptr, y, nullptr, nullptr, 0, stream, ENGINE_H3, &rg);
}

// the native launches of frcnn_conv2d_fwd_ws / frcnn_conv2d_fwd_dual that also leave max|y| in a magnitude record: what feeds an
// f16x3 launch from a layer that stays on the native kernels (the 3-channel stem, stage 4's 256-column 1x1 layers)
int frcnn_conv2d_fwd_ws_amax(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                             const float* scale, const float* shift, const float* residual, const float* mask, float* y, float* y_amax,
                             void* workspace, size_t workspace_bytes, void* stream) {
    const ConvRange rg = {nullptr, y_amax, nullptr};
    return conv_fwd_impl(d, x, w_packed, scale, shift, residual, mask, y, nullptr, workspace, workspace_bytes, stream, ENGINE_NATIVE, &rg);
}

int frcnn_conv2d_fwd_ws(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                        const float* scale, const float* shift, const float* residual, const float* mask, float* y,
                        void* workspace, size_t workspace_bytes, void* stream) {
    return conv_fwd_impl(d, x, w_packed, scale, shift, residual, mask, y, nullptr, workspace, workspace_bytes, stream);
}

// Split-K workspace of the two-layer launch: the balanced (stream-K) form is not used there.
size_t frcnn_conv2d_dual_workspace_bytes(const frcnn_conv_desc* d) {
    if (!d || d->cin <= 0 || (d->cin % BK) != 0) return 0;
    const int splits = choose_splits(d, choose_config(d));
    if (splits <= 1) return 0;
    const long long M = (long long)d->n * d->ho * d->wo;
    const size_t tiles = (size_t)((M + 63) / 64) * ((d->cout + 63) / 64);
    return SPLITK_TICKET_BYTES + tiles * splits * 64 * 64 * sizeof(float);
}

int frcnn_conv2d_fwd_dual(const frcnn_conv_desc* d, const float* x, const float* w_packed, const float* scale, const float* shift,
                          float* y1, int n1, int act1, float* y2, int act2,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !y2 || n1 <= 0 || n1 >= d->cout) return fail(FRCNN_E_ARG, "conv2d_fwd_dual: need 0 < n1 < cout and two outputs");
    if (d->ldy > 0 || d->ldres > 0) return fail(FRCNN_E_ARG, "conv2d_fwd_dual: dense outputs only (ldy = ldres = 0)");
    if ((d->cin % BK) != 0 || d->kh * d->kw > 32) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_dual: cin %% 32 == 0 and at most 32 taps");
    const DualOut dual = {n1, act1, y2, act2};
    return conv_fwd_impl(d, x, w_packed, scale, shift, nullptr, nullptr, y1, &dual, workspace, workspace_bytes, stream);
}

static int conv_fwd_impl(const frcnn_conv_desc* d, const float* x, const float* w_packed,
                         const float* scale, const float* shift, const float* residual, const float* mask, float* y,
                         const DualOut* dual, void* workspace, size_t workspace_bytes, void* stream, int engine, const ConvRange* range) {
    const bool planes_io = range && (range->x_planes || range->y_planes);
    if (!d || !w_packed || (!planes_io && (!x || !y))) return fail(FRCNN_E_ARG, "conv2d_fwd: null pointer");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cin <= 0 || d->cout <= 0 || d->kh <= 0 || d->kw <= 0 || d->stride <= 0 || d->ho <= 0 || d->wo <= 0)
        return fail(FRCNN_E_ARG, "conv2d_fwd: bad shape");
    ConvArgs a;
    a.x = x; a.w = w_packed; a.scale = scale; a.shift = shift; a.residual = residual; a.y = y; a.mask = mask;
    a.n_img = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.R = d->kh; a.S = d->kw;
    a.stride = d->stride; a.pad_top = d->pad_top; a.pad_left = d->pad_left; a.Ho = d->ho; a.Wo = d->wo;
    const long long M = (long long)d->n * d->ho * d->wo;
    if (M > 0x7fffffffLL) return fail(FRCNN_E_ARG, "conv2d_fwd: too many output pixels");
    a.M = (int)M; a.K = d->kh * d->kw * d->cin; a.Kpad = frcnn_conv_packed_k(d->kh, d->kw, d->cin);
    a.act = d->act; a.ldy = d->ldy > 0 ? d->ldy : d->cout; a.ldres = d->ldres > 0 ? d->ldres : d->cout;
    a.n_split = 0; a.y2 = nullptr; a.ldy2 = 0; a.act2 = 0;
    a.x_amax = range ? range->x_amax : nullptr; a.y_amax = range ? range->y_amax : nullptr; a.y2_amax = range ? range->y2_amax : nullptr;
    a.x_planes = range ? range->x_planes : nullptr; a.x_pexp = range ? range->x_pexp : nullptr;
    a.y_planes = range ? range->y_planes : nullptr; a.y_pexp = range ? range->y_pexp : nullptr;
    a.res_amax = range ? range->res_amax : nullptr; a.bound_c = range ? range->bound_c : 0.0f; a.bound_d = range ? range->bound_d : 0.0f;
    a.res_planes = range ? range->res_planes : nullptr; a.res_pexp = range ? range->res_pexp : nullptr;
    if (dual) { a.n_split = dual->n1; a.act = dual->act1; a.ldy = dual->n1; a.y2 = dual->y2; a.ldy2 = d->cout - dual->n1; a.act2 = dual->act2; }
    a.tiles_m = a.tiles_n = 0;
    a.splits = 1; a.slabs = nullptr; a.tickets = nullptr;
    a.layout = d->layout ? 1 : 0;
    a.pix_stride = a.layout ? d->n * d->cin : d->cin;
    a.img_stride = a.layout ? d->cin : d->h * d->w * d->cin;
    a.inv_S = (65536 + d->kw - 1) / d->kw;
    // the vector epilogue addresses rows in 16-byte pieces through 32-bit buffer offsets
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    a.vec_epi = !g_scalar_epilogue && (d->cout & 3) == 0 && (a.ldy & 3) == 0 && al16(y) && (size_t)M * a.ldy * 4 < 0x7fffffffull
             && (!residual || ((a.ldres & 3) == 0 && al16(residual) && (size_t)M * a.ldres * 4 < 0x7fffffffull))
             && (!mask || (al16(mask) && (size_t)M * d->cout * 4 < 0x7fffffffull)) && (!scale || al16(scale)) && (!shift || al16(shift));
    hipStream_t s = as_stream(stream);
    const bool generic = (d->cin % BK) != 0;
    if (engine == ENGINE_H3) {
        // the f16x3 engine (conv_h3.hip): w_packed points at the header + two fp16 filter planes
        if (generic || d->kh * d->kw > 32) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_h3: cin %% 32 == 0 and at most 32 taps");
        if ((size_t)2 * d->cout * a.Kpad * 2 >= 0x7fffffffull) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_h3: filter planes over 2 GiB");
        if (reinterpret_cast<uintptr_t>(w_packed) & 15) return fail(FRCNN_E_ARG, "conv2d_fwd_h3: 16-byte aligned filter planes required");
        const int hcfg = h3_config(d, dual ? dual->n1 : 0);
        if (planes_io) {
            // activations as fp16 planes: the double-buffered 256x128 forms only, 16-byte epilogue, one layer, no mask
            if ((hcfg != 86 && hcfg != 85 && hcfg != 82) || dual || mask || (!a.vec_epi && y) || d->ldy > 0)
                return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_h3_planes: needs the 256x128 tile (>= 256 output tiles of 128x128; frcnn_conv2d_h3_config 86 / 82), a dense single-layer launch without a mask");
            if ((d->cout & 3) || (d->cin & 7)) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_h3_planes: cin %% 8 == 0 and cout %% 4 == 0");
            if (a.res_planes && (residual || (y && !a.vec_epi))) return fail(FRCNN_E_ARG, "conv2d_fwd_h3_planes: residual planes exclude an f32 residual and need 16-byte addressable output rows");
            if ((size_t)M * d->cout * 4 >= 0x7fffffffull || (size_t)d->n * d->h * d->w * d->cin * 4 >= 0x7fffffffull)
                return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_h3_planes: tensor planes over 2 GiB");
            if (!y) a.vec_epi = 1;
            const int bn = h3_tile_width(hcfg);
            a.group_m = g_group_m >= 0 ? g_group_m : (d->cout > bn ? 1 : 0);
            return launch_conv_h3(a, hcfg, s);
        }
        if (workspace && !dual) {
            const size_t need = frcnn_conv2d_h3_workspace_bytes(d);
            if (need) {
                if (workspace_bytes < need) return fail(FRCNN_E_WORKSPACE, "conv2d_fwd_h3: workspace needs %zu bytes", need);
                a.splits = choose_splits_h3(d);
                a.tickets = (unsigned*)workspace;
                a.slabs = (float*)((char*)workspace + SPLITK_TICKET_BYTES);
                a.group_m = 0;
                return launch_conv_h3(a, h3_sk_tile(d) == 128 ? 181 : 184, s);
            }
        }
        const int bn = h3_tile_width(hcfg);
        if (dual) a.vec_epi = a.vec_epi && dual->n1 % bn == 0 && (a.ldy2 & 3) == 0 && al16(dual->y2) && (size_t)M * a.ldy2 * 4 < 0x7fffffffull;
        a.group_m = g_group_m >= 0 ? g_group_m : (d->cout > bn ? 1 : 0);      // column tiles of a row tile adjacent on one XCD
        return launch_conv_h3(a, hcfg, s);
    }
    if (engine == ENGINE_X6) {
        // the split-bf16 engine (conv_x6.hip): w_packed points at the three bf16 filter planes
        if (generic || d->kh * d->kw > 32) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_x6: cin %% 32 == 0 and at most 32 taps");
        if ((size_t)3 * d->cout * a.Kpad * 2 >= 0x7fffffffull) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd_x6: filter planes over 2 GiB");
        const int xcfg = x6_config(d, dual ? dual->n1 : 0);
        if (workspace && !dual) {
            const size_t need = frcnn_conv2d_x6_workspace_bytes(d);
            if (need) {
                if (workspace_bytes < need) return fail(FRCNN_E_WORKSPACE, "conv2d_fwd_x6: workspace needs %zu bytes", need);
                a.splits = choose_splits_x6(d);
                a.tickets = (unsigned*)workspace;
                a.slabs = (float*)((char*)workspace + SPLITK_TICKET_BYTES);
                a.group_m = 0;
                return launch_conv_x6(a, x6_sk_tile(d) == 128 ? 171 : 174, s);
            }
        }
        const int bn = x6_tile_width(xcfg);
        if (dual) a.vec_epi = a.vec_epi && dual->n1 % bn == 0 && (a.ldy2 & 3) == 0 && al16(dual->y2) && (size_t)M * a.ldy2 * 4 < 0x7fffffffull;
        a.group_m = g_group_m >= 0 ? g_group_m : (d->cout > bn ? 1 : 0);      // column tiles of a row tile adjacent on one XCD
        return launch_conv_x6(a, xcfg, s);
    }
    int cfg = choose_config(d);
    if (dual) {
        cfg = dual_config(cfg);
        // the 16-byte epilogue picks the output per TILE: the boundary between the layers must be a tile boundary and
        // both outputs 16-byte addressable; otherwise every lane picks per column (the 4-byte epilogue)
        const int bn = (cfg == 21 || cfg == 26 || cfg == 11 || cfg == 24 || cfg == 14) ? 128 : 64;
        a.vec_epi = a.vec_epi && dual->n1 % bn == 0 && (a.ldy2 & 3) == 0 && al16(dual->y2) && (size_t)M * a.ldy2 * 4 < 0x7fffffffull;
    }
    // tile order (ConvArgs.group_m): multi-round 64x64 launches with many column tiles walk groups of four row tiles
    // (1x1 512->2048 on 14 700 rows: 276 -> 261 us, scripts/micro/conv_lab.hip under FRCNN_GROUP_M); single-round
    // grids and the big tiles measured no difference and keep the plain order
    {
        const long long t64 = ((M + 63) / 64) * ((d->cout + 63) / 64);
        a.group_m = g_group_m >= 0 ? g_group_m : ((cfg == 22 || cfg == 23) && t64 > 1024 && d->cout >= 512 ? 4 : 0);
        // round 3 (scripts/group_m_sweep.sh, rocprofv3 --pmc FETCH_SIZE): the plain 128x128 launches gave the column tiles
        // of one row tile to DIFFERENT XCDs (an XCD's run of ids was ~57 row tiles of one column tile), so every A row
        // tile crossed the fabric once per column tile.  With groups of ONE row tile the column tiles that share A rows are
        // neighbours on one XCD: 2048->512 on 14 700 rows fetches 105 instead of 259 MB (raw counter) at 287 vs 291 us, the
        // 3x3 181 instead of 203 MB (its nine taps reach the neighbouring positions' rows, which live on other XCDs).
        if (g_group_m < 0 && (cfg == 21 || cfg == 26) && d->cout > 128) a.group_m = 1;
    }
    if (cfg == 21 && !workspace && d->tile % 100 == 0 && a.layout && d->kh * d->kw > 1)
        cfg = 23;                                               // the 128x128 choice there counted on the balanced form
    if (a.layout && (generic || cfg < 11 || d->kh * d->kw > 32))
        return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd: position-major layout needs cin %% 32 == 0, a tensor under 2 GiB and at most 32 taps");
    if (d->cin == 3) {                                          // the stems: filter packed 4 wide, eight taps per chunk
        if (a.layout) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd: position-major layout needs cin %% 32 == 0");
        if ((size_t)d->n * d->h * d->w * 12 >= 0x7fffffffull) return fail(FRCNN_E_UNSUPPORTED, "conv2d_fwd: 3-channel input over 2 GiB");
        return launch_conv_cin3(a, s, d->tile % 100 != 32);      // the mid-chunk-barrier loop (conv1 50.6 -> 49.4 us); 32: dev code, the late-store loop
    }
    if (!generic && workspace) {
        const size_t need = dual ? frcnn_conv2d_dual_workspace_bytes(d) : frcnn_conv2d_workspace_bytes(d);
        if (need) {
            if (workspace_bytes < need) return fail(FRCNN_E_WORKSPACE, "conv2d_fwd: workspace needs %zu bytes", need);
            if (const int G = dual ? 0 : choose_streamk(d, cfg)) {
                a.tickets = (unsigned*)workspace;
                a.slabs = (float*)((char*)workspace + SPLITK_TICKET_BYTES);
                return (cfg == 21 || cfg == 26 || cfg == 61) ? launch_conv_sk<2, 2>(a, G, s) : launch_conv_sk<1, 1>(a, G, s);
            }
            a.splits = choose_splits(d, cfg);
            a.tickets = (unsigned*)workspace;
            a.slabs = (float*)((char*)workspace + SPLITK_TICKET_BYTES);
            return cfg == 23 ? launch_conv_v2_splitk<1, 1, 2>(a, s) : launch_conv_v2_splitk<1, 1, 1>(a, s);
        }
    }
    if (generic) {
        if (cfg == 2) return launch_conv<1, 1, true>(a, s);
        return launch_conv<2, 1, true>(a, s);
    }
    switch (cfg) {
        case 41: return launch_conv_v2<1, 2, 1, 4, 2>(a, s);     // 128x128, 8 waves
        case 42: return launch_conv_v2<2, 1, 1, 2, 4>(a, s);     // 128x128, 8 waves (2x4)
        case 43: return launch_conv_v2<1, 1, 1, 4, 2>(a, s);     // 128x64, 8 waves
        case 61: case 21: return launch_conv_v2<2, 2, 1>(a, s);     // (61 / 62 without a workspace: the plain launch)
        case 62: case 22: return launch_conv_v2<1, 1, 1>(a, s);
        case 23: return launch_conv_v2<1, 1, 2>(a, s);           // 64x64, mid-chunk barrier main loop
        case 24: return launch_conv_v2<1, 2, 2>(a, s);           // 64x128
        case 25: return launch_conv_v2<2, 1, 2>(a, s);           // 128x64
        case 26: return launch_conv_v2<2, 2, 2>(a, s);           // 128x128
        case 11: return launch_conv_v2<2, 2>(a, s);
        case 12: return launch_conv_v2<1, 1>(a, s);
        case 13: return launch_conv_v2<2, 1>(a, s);
        case 14: return launch_conv_v2<4, 2>(a, s);
        case 1: return launch_conv<2, 2, false>(a, s);
        case 2: return launch_conv<1, 1, false>(a, s);
        case 3: return launch_conv<2, 1, false>(a, s);
        case 4: return launch_conv<4, 2, false>(a, s);
        default: return fail(FRCNN_E_ARG, "conv2d_fwd: unknown tile config %d", cfg);
    }
}

int frcnn_pack_conv_weights_dgrad(const float* w_hwio, const float* scale, int kh, int kw, int cin, int cout, float* packed, void* stream) {
    if (!w_hwio || !packed || kh <= 0 || kw <= 0 || cin <= 0 || cout <= 0) return fail(FRCNN_E_ARG, "pack_conv_weights_dgrad: bad argument");
    const int Kpad = frcnn_conv_packed_k(kh, kw, cout);
    const size_t total = (size_t)cin * Kpad;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    k_pack_dgrad<<<grid, 256, 0, as_stream(stream)>>>(w_hwio, scale, kh, kw, cin, cout, Kpad, packed);
    return check_launch("pack_conv_weights_dgrad");
}

// f32 operands, cin and cout >= 128: the 128x128-tile kernel (dev knob FRCNN_WGRAD_BIG=0: the 64x64 kernel everywhere)
// frcnn_conv_desc.tile 71..77 on an f32 job asks for the split-bf16 engine (kind 4) where the 128x128 form applies
static bool wgrad_wants_x6(const frcnn_conv_desc* d) { const int t = d->tile % 100; return t >= 71 && t <= 77; }

static bool wgrad_big(const frcnn_conv_desc* d, bool in_bf16) {
    static const bool on = !(getenv("FRCNN_WGRAD_BIG") && atoi(getenv("FRCNN_WGRAD_BIG")) == 0);
    if (!on || in_bf16 || d->cin < 128 || d->cout < 128 || (d->cin & 3) || (d->cout & 3)) return false;
    const size_t xb = (size_t)d->n * d->h * d->w * d->cin * 4, gb = (size_t)d->n * d->ho * d->wo * d->cout * 4;
    return xb < 0x80000000ull && gb < 0x80000000ull && (long long)d->n * d->ho * d->wo < (1 << 23);         // 32-bit buffer offsets, float-exact pixel index
}

// the 128x128 bf16 form (kind 5): both channel counts >= 128 and multiples of 8, 32-bit byte offsets, float-exact pixel index
static bool wgrad_big_bf16(const frcnn_conv_desc* d) {
    static const bool on = !(getenv("FRCNN_WGRAD_BIG_BF16") && atoi(getenv("FRCNN_WGRAD_BIG_BF16")) == 0);
    if (!on || d->cin < 128 || d->cout < 128 || (d->cin & 7) || (d->cout & 7)) return false;
    const size_t xb = (size_t)d->n * d->h * d->w * d->cin * 2, gb = (size_t)d->n * d->ho * d->wo * d->cout * 2;
    return xb < 0x80000000ull && gb < 0x80000000ull && (long long)d->n * d->ho * d->wo < (1 << 23);
}

static int wgrad_slices(const frcnn_conv_desc* d, bool big, bool fast = false) {
    const long long M = (long long)d->n * d->ho * d->wo;
    const int tw = big ? 128 : 64;
    const long long tiles = (long long)d->kh * d->kw * ((d->cin + tw - 1) / tw) * ((d->cout + tw - 1) / tw);
    // ~256 workgroups per layer: the layers of a step are launched together (frcnn_conv2d_wgrad_batch), so the chip is
    // filled by the batch, not by one layer, and fewer slices mean fewer partial slabs to write and re-read (measured,
    // scripts/micro/train_ab2.py, target 2048 -> 256: mixed RPN step 2.36 -> 2.08 ms, detector step 3.30 -> 2.85 ms)
    static const long long target = getenv("FRCNN_WGRAD_TARGET") ? atoll(getenv("FRCNN_WGRAD_TARGET")) : 256;   // dev knob
    static const long long target_big = getenv("FRCNN_WGRAD_TARGET_BIG") ? atoll(getenv("FRCNN_WGRAD_TARGET_BIG")) : 256;
    // the split-bf16 form's workgroups finish sooner: fewer, longer slices (bench_train.py: 2.18 / 4.02 ms at 256, 2.12 / 3.96 at 96-128,
    // 2.10 / 3.93 at 48, 2.19 / 4.04 at 32)
    static const long long target_x6 = getenv("FRCNN_WGRAD_TARGET_X6") ? atoll(getenv("FRCNN_WGRAD_TARGET_X6")) : 96;
    const long long tg = big ? ((fast || wgrad_wants_x6(d)) ? target_x6 : target_big) : target;
    long long s = (tg + tiles - 1) / tiles;
    const long long max_s = (M + 4 * WG_MC - 1) / (4 * WG_MC); // at least 4 chunks per slice
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
}
static int wgrad_slices_max(const frcnn_conv_desc* d) {
    const int a = wgrad_slices(d, false), b = wgrad_big(d, false) ? wgrad_slices(d, true) : 0;
    const int c = (wgrad_big(d, false) || wgrad_big_bf16(d)) ? wgrad_slices(d, true, true) : 0;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
}

size_t frcnn_conv2d_wgrad_workspace_bytes(const frcnn_conv_desc* d) {
    if (!d) return 0;
    const size_t dw = (size_t)wgrad_slices_max(d) * d->kh * d->kw * d->cin * d->cout * sizeof(float);
    const size_t db = (size_t)COLSUM_SLICES * d->cout * sizeof(float);
    return align_up(dw > db ? dw : db, 256);
}

static int wgrad_impl(const frcnn_conv_desc* d, const void* x, const void* g, bool in_bf16, const float* scale,
                      float* dw_hwio, float* dbias, void* workspace, size_t workspace_bytes, void* stream);

int frcnn_conv2d_wgrad(const frcnn_conv_desc* d, const float* x, const float* g, const float* scale,
                       float* dw_hwio, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_impl(d, x, g, false, scale, dw_hwio, dbias, workspace, workspace_bytes, stream);
}

int frcnn_conv2d_wgrad_bf16(const frcnn_conv_desc* d, const void* x_bf16, const void* g_bf16, const float* scale,
                            float* dw_hwio, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    return wgrad_impl(d, x_bf16, g_bf16, true, scale, dw_hwio, dbias, workspace, workspace_bytes, stream);
}

static int wgrad_impl(const frcnn_conv_desc* d, const void* x, const void* g, bool in_bf16, const float* scale,
                      float* dw_hwio, float* dbias, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d || !x || !g || !dw_hwio) return fail(FRCNN_E_ARG, "conv2d_wgrad: null pointer");
    if ((d->cin & 3) && d->cin >= 4) return fail(FRCNN_E_UNSUPPORTED, "conv2d_wgrad: cin must be a multiple of 4 (or < 4)");
    if (in_bf16 && ((d->cin & 3) || (d->cout & 3))) return fail(FRCNN_E_UNSUPPORTED, "conv2d_wgrad_bf16: cin and cout must be multiples of 4");
    if (!workspace || workspace_bytes < frcnn_conv2d_wgrad_workspace_bytes(d))
        return fail(FRCNN_E_WORKSPACE, "conv2d_wgrad: workspace needs %zu bytes", frcnn_conv2d_wgrad_workspace_bytes(d));
    WgradArgs a;
    a.x = x; a.g = g; a.partial = (float*)workspace;
    a.n_img = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.R = d->kh; a.S = d->kw;
    a.stride = d->stride; a.pad_top = d->pad_top; a.pad_left = d->pad_left; a.Ho = d->ho; a.Wo = d->wo;
    a.M = d->n * d->ho * d->wo;
    const bool big16 = in_bf16 && wgrad_big_bf16(d);
    const bool big = wgrad_big(d, in_bf16) || big16;
    const int slices = wgrad_slices(d, big, big16);
    a.m_per_slice = ((a.M + slices - 1) / slices + WG_MC - 1) / WG_MC * WG_MC;
    hipStream_t s = as_stream(stream);
    const int tw = big ? 128 : 64;
    dim3 grid(d->kh * d->kw * ((d->cin + tw - 1) / tw), (d->cout + tw - 1) / tw, slices);
    if (big16) k_conv_wgrad_bf16_big<<<grid, 256, 0, s>>>(a);
    else if (big && wgrad_wants_x6(d)) k_conv_wgrad_x6_big<<<grid, 256, 0, s>>>(a);
    else if (big) k_conv_wgrad_f32_big<<<grid, 256, 0, s>>>(a);
    else if (in_bf16 && (d->cin & 7) == 0 && (d->cout & 7) == 0) k_conv_wgrad_bf16<<<grid, 256, 0, s>>>(a);      // bf16 MFMA
    else if (in_bf16) k_conv_wgrad_f32<true><<<grid, 256, 0, s>>>(a);                                          // widened, f32 MFMA
    else k_conv_wgrad_f32<false><<<grid, 256, 0, s>>>(a);
    if (int e = check_launch("conv2d_wgrad")) return e;
    const size_t elems = (size_t)d->kh * d->kw * d->cin * d->cout;
    int rgrid = (int)((elems + 255) / 256);
    if (rgrid > 4096) rgrid = 4096;
    k_wgrad_reduce<<<rgrid, 256, 0, s>>>((const float*)workspace, slices, elems, d->cout, scale, dw_hwio);
    if (int e = check_launch("conv2d_wgrad reduce")) return e;
    if (dbias && in_bf16) {
        frcnn_colsum_job job;
        job.g = g; job.scale = scale; job.out = dbias; job.m = a.M; job.cout = d->cout; job.g_is_bf16 = 1; job.reserved = 0;
        return frcnn_colsum_batch(&job, 1, stream);
    }
    if (dbias) {                                   // the slab workspace is free again after the reduce (stream order)
        int cs = (a.M + 63) / 64;
        if (cs > COLSUM_SLICES) cs = COLSUM_SLICES;
        if (cs < 1) cs = 1;
        const int rows_per_slice = (a.M + cs - 1) / cs;
        k_colsum_partial<<<dim3((d->cout + 63) / 64, cs), 256, 0, s>>>((const float*)g, a.M, d->cout, rows_per_slice, (float*)workspace);
        if (int e = check_launch("conv2d_wgrad bias")) return e;
        k_colsum_final<<<(d->cout + 255) / 256, 256, 0, s>>>((const float*)workspace, cs, d->cout, scale, dbias);
        if (int e = check_launch("conv2d_wgrad bias")) return e;
    }
    return FRCNN_OK;
}

static int wgrad_kind(const frcnn_wgrad_job& j) {
    if (!j.in_bf16) return wgrad_big(&j.d, false) ? (wgrad_wants_x6(&j.d) ? 4 : 3) : 0;
    if (wgrad_big_bf16(&j.d)) return 5;
    return ((j.d.cin & 7) == 0 && (j.d.cout & 7) == 0) ? 1 : 2;
}

static size_t wgrad_slab_bytes(const frcnn_conv_desc* d) {
    return align_up((size_t)wgrad_slices_max(d) * d->kh * d->kw * d->cin * d->cout * sizeof(float), 256);
}

size_t frcnn_conv2d_wgrad_batch_workspace_bytes(const frcnn_wgrad_job* jobs, int n_jobs) {
    size_t tot = 0;
    for (int i = 0; jobs && i < n_jobs; ++i) tot += wgrad_slab_bytes(&jobs[i].d);
    return tot;
}

int frcnn_conv2d_wgrad_batch(const frcnn_wgrad_job* jobs, int n_jobs, void* workspace, size_t workspace_bytes, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(FRCNN_E_ARG, "conv2d_wgrad_batch: bad argument");
    if (n_jobs == 0) return FRCNN_OK;
    if (!workspace || workspace_bytes < frcnn_conv2d_wgrad_batch_workspace_bytes(jobs, n_jobs))
        return fail(FRCNN_E_WORKSPACE, "conv2d_wgrad_batch: workspace needs %zu bytes", frcnn_conv2d_wgrad_batch_workspace_bytes(jobs, n_jobs));
    hipStream_t s = as_stream(stream);
    // every job's slab region, in job order
    size_t off = 0;
    static thread_local float* slab[1024];
    if (n_jobs > 1024) return fail(FRCNN_E_ARG, "conv2d_wgrad_batch: more than 1024 jobs");
    for (int i = 0; i < n_jobs; ++i) {
        const frcnn_wgrad_job& j = jobs[i];
        if (!j.x || !j.g || !j.dw) return fail(FRCNN_E_ARG, "conv2d_wgrad_batch: job %d has a null pointer", i);
        if ((j.d.cin & 3) && j.d.cin >= 4) return fail(FRCNN_E_UNSUPPORTED, "conv2d_wgrad_batch: job %d: cin must be a multiple of 4 (or < 4)", i);
        if (j.in_bf16 && ((j.d.cin & 3) || (j.d.cout & 3))) return fail(FRCNN_E_UNSUPPORTED, "conv2d_wgrad_batch: job %d: bf16 operands need cin, cout multiples of 4", i);
        slab[i] = (float*)((char*)workspace + off);
        off += wgrad_slab_bytes(&j.d);
    }
    for (int kind = 0; kind < 6; ++kind) {
        WgradBatch t;
        t.n = 0;
        int blocks = 0;
        auto flush = [&]() -> int {
            if (t.n == 0) return FRCNN_OK;
            t.first_block[t.n] = blocks;
            for (int q = t.n; q < WGRAD_BATCH; ++q) { t.job[q] = t.job[0]; t.gx[q] = t.gy[q] = 1; t.first_block[q + 1] = blocks; }
            if (kind == 0) k_conv_wgrad_batch<0><<<blocks, 256, 0, s>>>(t);
            else if (kind == 1) k_conv_wgrad_batch<1><<<blocks, 256, 0, s>>>(t);
            else if (kind == 2) k_conv_wgrad_batch<2><<<blocks, 256, 0, s>>>(t);
            else if (kind == 3) k_conv_wgrad_batch<3><<<blocks, 256, 0, s>>>(t);
            else if (kind == 4) k_conv_wgrad_batch<4><<<blocks, 256, 0, s>>>(t);
            else k_conv_wgrad_batch<5><<<blocks, 256, 0, s>>>(t);
            t.n = 0; blocks = 0;
            return check_launch("conv2d_wgrad_batch");
        };
        for (int i = 0; i < n_jobs; ++i) {
            const frcnn_wgrad_job& j = jobs[i];
            if (wgrad_kind(j) != kind) continue;
            const frcnn_conv_desc* d = &j.d;
            WgradArgs& a = t.job[t.n];
            a.x = j.x; a.g = j.g; a.partial = slab[i];
            a.n_img = d->n; a.H = d->h; a.W = d->w; a.Cin = d->cin; a.Cout = d->cout; a.R = d->kh; a.S = d->kw;
            a.stride = d->stride; a.pad_top = d->pad_top; a.pad_left = d->pad_left; a.Ho = d->ho; a.Wo = d->wo;
            a.M = d->n * d->ho * d->wo;
            const int slices = wgrad_slices(d, kind >= 3, kind == 5);
            a.m_per_slice = ((a.M + slices - 1) / slices + WG_MC - 1) / WG_MC * WG_MC;
            const int tw = kind >= 3 ? 128 : 64;
            t.gx[t.n] = d->kh * d->kw * ((d->cin + tw - 1) / tw);
            t.gy[t.n] = (d->cout + tw - 1) / tw;
            t.first_block[t.n] = blocks;
            blocks += t.gx[t.n] * t.gy[t.n] * slices;
            if (++t.n == WGRAD_BATCH) if (int e = flush()) return e;
        }
        if (int e = flush()) return e;
    }
    // the slice reductions of all jobs: grid.y = job
    for (int b = 0; b < n_jobs; b += 2 * WGRAD_BATCH) {
        WgradReduceBatch r;
        r.n = n_jobs - b < 2 * WGRAD_BATCH ? n_jobs - b : 2 * WGRAD_BATCH;
        for (int i = 0; i < 2 * WGRAD_BATCH; ++i) {
            const int q = b + (i < r.n ? i : 0);
            const frcnn_conv_desc* d = &jobs[q].d;
            r.job[i].partial = slab[q]; r.job[i].scale = jobs[q].scale; r.job[i].dw = jobs[q].dw;
            r.job[i].elems = (unsigned long long)d->kh * d->kw * d->cin * d->cout;
            r.job[i].slices = wgrad_slices(d, wgrad_kind(jobs[q]) >= 3, wgrad_kind(jobs[q]) == 5); r.job[i].cout = d->cout;
        }
        int rblocks = 0;
        for (int i = 0; i <= 2 * WGRAD_BATCH; ++i) {
            r.first_block[i] = rblocks;
            if (i < r.n) {
                const unsigned long long g = (r.job[i].elems + 2047) / 2048;       // 8 elements (x slices) per thread
                rblocks += (int)(g < 4 ? 4 : (g > 2048 ? 2048 : g));
            }
        }
        k_wgrad_reduce_batch<<<rblocks, 256, 0, s>>>(r);
        if (int e = check_launch("conv2d_wgrad_batch reduce")) return e;
    }
    return FRCNN_OK;
}

int frcnn_refresh_packed(const frcnn_pack_job* jobs, int n_jobs, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(FRCNN_E_ARG, "refresh_packed: bad argument");
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs[i].w_hwio || jobs[i].kh <= 0 || jobs[i].kw <= 0 || jobs[i].cin <= 0 || jobs[i].cout <= 0)
            return fail(FRCNN_E_ARG, "refresh_packed: job %d is malformed", i);
    for (int b = 0; b < n_jobs; b += REFRESH_JOBS) {
        RefreshTable t;
        const int n = n_jobs - b < REFRESH_JOBS ? n_jobs - b : REFRESH_JOBS;
        int blocks = 0;
        for (int i = 0; i < n; ++i) { t.job[i] = jobs[b + i]; t.first_block[i] = blocks; blocks += refresh_blocks(jobs[b + i]); }
        for (int i = n; i < REFRESH_JOBS; ++i) { t.job[i] = jobs[b]; t.first_block[i] = blocks; }        // never indexed
        t.first_block[REFRESH_JOBS] = blocks;
        t.n = n;
        k_refresh_packed<<<blocks, 256, 0, as_stream(stream)>>>(t);
        if (int e = check_launch("refresh_packed")) return e;
    }
    return FRCNN_OK;
}

int frcnn_colsum_batch(const frcnn_colsum_job* jobs, int n_jobs, void* stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(FRCNN_E_ARG, "colsum_batch: bad argument");
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs[i].g || !jobs[i].out || jobs[i].m <= 0 || jobs[i].cout <= 0) return fail(FRCNN_E_ARG, "colsum_batch: job %d is malformed", i);
    for (int b = 0; b < n_jobs; b += COLSUM_JOBS) {
        ColsumTable t;
        t.n = n_jobs - b < COLSUM_JOBS ? n_jobs - b : COLSUM_JOBS;
        int blocks = 0;
        for (int i = 0; i < t.n; ++i) { t.job[i] = jobs[b + i]; t.first_block[i] = blocks; blocks += (jobs[b + i].cout + 63) / 64; }
        for (int i = t.n; i < COLSUM_JOBS; ++i) { t.job[i] = jobs[b]; t.first_block[i] = blocks; }
        t.first_block[COLSUM_JOBS] = blocks;
        k_colsum_batch<<<blocks, 1024, 0, as_stream(stream)>>>(t);
        if (int e = check_launch("colsum_batch")) return e;
    }
    return FRCNN_OK;
}

int frcnn_conv2d_config(const frcnn_conv_desc* d) {
    if (!d) return fail(FRCNN_E_ARG, "conv2d_config: null descriptor");
    if (d->cin == 3) return 30;                                 // the 3-channel stem kernel, whatever tile was asked for
    const int cfg = choose_config(d);
    if (choose_streamk(d, cfg)) return (cfg == 21 || cfg == 26 || cfg == 61) ? 61 : 62;     // what a launch WITH a workspace runs
    return (cfg == 61 || cfg == 62) ? cfg - 40 : cfg;                           // asked for, but the shape is not eligible
}

int frcnn_conv2d_x6_config(const frcnn_conv_desc* d, int n1) {
    if (!d) return fail(FRCNN_E_ARG, "conv2d_x6_config: null descriptor");
    return x6_config(d, n1);
}

int frcnn_conv2d_dual_config(const frcnn_conv_desc* d, int has_workspace) {
    if (!d) return fail(FRCNN_E_ARG, "conv2d_dual_config: null descriptor");
    int cfg = dual_config(choose_config(d));
    if (cfg == 21 && !has_workspace && d->tile % 100 == 0 && d->layout && d->kh * d->kw > 1) cfg = 23;      // as conv_fwd_impl
    return cfg;
}

int frcnn_pool2d_fwd(const float* x, int n, int h, int w, int c, int k, int stride, int is_max, float* y, void* stream) {
    if (!x || !y || n <= 0 || h < k || w < k || c <= 0 || (c & 3) || k <= 0 || stride <= 0) return fail(FRCNN_E_ARG, "pool2d_fwd: bad argument (C must be a multiple of 4)");
    const int Ho = (h - k) / stride + 1, Wo = (w - k) / stride + 1;
    const size_t total = (size_t)n * Ho * Wo * (c / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (is_max) k_pool<true><<<grid, 256, 0, as_stream(stream)>>>((const float4*)x, n, h, w, c / 4, k, stride, Ho, Wo, (float4*)y);
    else k_pool<false><<<grid, 256, 0, as_stream(stream)>>>((const float4*)x, n, h, w, c / 4, k, stride, Ho, Wo, (float4*)y);
    return check_launch("pool2d_fwd");
}

int frcnn_pool2d_fwd_planes(const float* x, int n, int h, int w, int c, int k, int stride, int is_max, const frcnn_h3_planes* out, void* stream) {
    if (!x || !out || !out->planes || !out->exponent || n <= 0 || h < k || w < k || c <= 0 || (c & 3) || k <= 0 || stride <= 0 || (reinterpret_cast<uintptr_t>(out->planes) & 15))
        return fail(FRCNN_E_ARG, "pool2d_fwd_planes: bad argument (C must be a multiple of 4, planes 16-byte aligned, exponent set before the launch)");
    const int Ho = (h - k) / stride + 1, Wo = (w - k) / stride + 1;
    const size_t total = (size_t)n * Ho * Wo * (c / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (is_max) k_pool_planes<true><<<grid, 256, 0, as_stream(stream)>>>((const float4*)x, n, h, w, c / 4, k, stride, Ho, Wo, out->exponent, (unsigned*)out->status, (_Float16*)out->planes, total * 4);
    else k_pool_planes<false><<<grid, 256, 0, as_stream(stream)>>>((const float4*)x, n, h, w, c / 4, k, stride, Ho, Wo, out->exponent, (unsigned*)out->status, (_Float16*)out->planes, total * 4);
    return check_launch("pool2d_fwd_planes");
}

int frcnn_avgpool_pos_major(const float* x, int npos, int n, int c, float* y, void* stream) {
    if (!x || !y || npos <= 0 || n <= 0 || c <= 0 || (c & 3)) return fail(FRCNN_E_ARG, "avgpool_pos_major: bad argument (C must be a multiple of 4)");
    const size_t total = (size_t)n * (c / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    k_avgpool_pos_major<<<grid, 256, 0, as_stream(stream)>>>((const float4*)x, npos, n, c / 4, (float4*)y);
    return check_launch("avgpool_pos_major");
}

int frcnn_softmax_rows(const float* x, int rows, int cols, int ldx, float* y, int ldy, void* stream) {
    if (rows < 0 || cols <= 0 || ldx < cols || ldy < cols) return fail(FRCNN_E_ARG, "softmax_rows: bad argument");
    if (rows == 0) return FRCNN_OK;
    if (!x || !y) return fail(FRCNN_E_ARG, "softmax_rows: null pointer");
    k_softmax_rows<<<(rows + 63) / 64, 64, 0, as_stream(stream)>>>(x, rows, cols, ldx, y, ldy);
    return check_launch("softmax_rows");
}

int frcnn_dense_heads_split(const float* x, int rows, int cols, int tail, int ldx, float* cls, float* reg, void* stream) {
    if (rows < 0 || cols <= 0 || tail < 0 || ldx < cols + tail) return fail(FRCNN_E_ARG, "dense_heads_split: bad argument");
    if (rows == 0) return FRCNN_OK;
    if (!x || !cls || (tail && !reg)) return fail(FRCNN_E_ARG, "dense_heads_split: null pointer");
    k_dense_heads_split<<<(rows + 7) / 8, 256, 0, as_stream(stream)>>>(x, rows, cols, tail, ldx, cls, reg);
    return check_launch("dense_heads_split");
}

}  // extern "C"
