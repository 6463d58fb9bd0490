// Device-side pieces shared by the f32 convolution engines of libfrcnn_hip.so (gfx950): the launch argument block, the
// XCD-aware tile map, the fused epilogues (4-byte "lane owns a column" form and the 16-byte row-piece form) and the
// scheduling-group constants.  Included by conv_igemm.hip (v_mfma_f32_32x32x2_f32 main loops) and conv_x6.hip (the
// same GEMM on the bf16 matrix cores by exact three-way operand splitting).
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* x; const float* w; const float* scale; const float* shift; const float* residual; float* y;
    const float* mask;      // optional [M][Cout]: output is zeroed where mask <= 0 (ReLU backward fused into dgrad)
    int n_img, H, W, Cin, Cout, R, S, stride, pad_top, pad_left, Ho, Wo;
    int M, K, Kpad, act, ldy, ldres;
    int tiles_m, tiles_n;
    int layout;             // 0: x[img][h][w][c], rows m = (img, ho, wo); 1 (v2 only): x[h][w][img][c], m = (ho, wo, img)
    int pix_stride;         // elements between w-neighbours of one image (Cin, or n_img*Cin when layout == 1)
    int img_stride;         // elements between images (H*W*Cin, or Cin when layout == 1)
    int inv_S;              // ceil(65536 / S): tap / S without an integer division
    int splits;             // split-K: K-slices per output tile (1 = none)
    float* slabs;           // [tile][slice][BM*BN] f32 partial tiles
    unsigned* tickets;      // [tile] arrival counters: zero on entry, left zero on exit
    int group_m;            // tile order inside an XCD's run of ids: 0 = all row tiles of one column tile, then the next column;
                            // g > 0 = groups of g row tiles, every column tile of a group before the next group (L2 working set)
    int vec_epi;            // 1: y / residual / mask rows are 16-byte addressable (set by frcnn_conv2d_fwd_ws): the v2 / balanced kernels use epilogue_vec
    // Two layers in one launch (frcnn_conv2d_fwd_dual): columns [0, n_split) are layer 1 -> y (ldy, act), columns
    // [n_split, Cout) are layer 2 -> y2 (ldy2, act2).  n_split == 0: one layer.  No residual / mask in this form.
    int n_split; float* y2; int ldy2, act2;
    // Magnitude records (frcnn_amax_*; AMAX_SLOTS x AMAX_STRIDE floats each, NULL = not tracked): every epilogue folds max|y| of what
    // it stores into y_amax (y2_amax: the second layer of a paired launch); the f16x3 engine (conv_h3.hip) reads x_amax -- an UPPER
    // BOUND of |x| -- to bring the activations into fp16's range by one power of two.
    const float* x_amax; float* y_amax; float* y2_amax;
    // Activations as two fp16 planes (conv_h3.hip, the 256x128 forms): x_planes [2][rows][Cin] read INSTEAD of x, scaled by 2^*x_pexp;
    // y_planes [2][M][Cout] written beside / instead of y under the scale 2^eY, eY from the bound bound_c * max|x| + bound_d
    // (+ max|residual|, res_amax) that every workgroup derives from the same device words; *y_pexp = eY.
    const void* x_planes; const int* x_pexp; void* y_planes; int* y_pexp; const float* res_amax; float bound_c, bound_d;
    // the residual as two fp16 planes [2][M][Cout] under the scale 2^*res_pexp (a block's output handed on as planes only: the next
    // block's shortcut reads them back, residual == NULL then); the vector epilogues of the f16x3 256x128 forms only
    const void* res_planes = nullptr; const int* res_pexp = nullptr;
};

// ---- magnitude records.  A record is AMAX_SLOTS floats, one per 128-byte line: a launch has hundreds to thousands of waves and a
// single word takes ~90 atomics per microsecond (MI355X_MICROARCH.md, dequeue), so each wave folds its maximum into slot
// (global wave id) % AMAX_SLOTS with ONE no-return atomic max on the value's bits (non-negative floats order like unsigned
// integers); a reader takes the maximum over the slots.  Records are cleared by frcnn_amax_clear (a kernel: a memset node inside
// a captured graph is not ordered against its neighbours on this runtime).
constexpr int AMAX_SLOTS = 32, AMAX_STRIDE = 32;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__device__ __forceinline__ void amax_publish(float* rec, float v) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0 && v > 0.0f) {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        atomicMax(reinterpret_cast<unsigned*>(rec + (w & (AMAX_SLOTS - 1)) * AMAX_STRIDE), __float_as_uint(v));
    }
}

__device__ __forceinline__ float amax_read(const float* rec) {
    const int lane = threadIdx.x & 63;
    return wave_max(lane < AMAX_SLOTS ? rec[lane * AMAX_STRIDE] : 0.0f);
}

// ---- the f16x3 engine's fences (include/frcnn_hip.h "Status word of a magnitude record").  Word 1 of a record is a STICKY status word
// (cleared with the record by frcnn_amax_clear): a launch that scales a tensor into fp16's range under that record ORs in
//   H3_UNDER      a value at or above 2^15 after scaling: the record was not an upper bound (stale, or a producer's bug),
//   H3_SATURATED  ... and at fp16's largest finite value: the conversion was CLAMPED to +-65504 (MODE.FP16_OVFL), the result is finite but
//                 wrong there,
//   H3_NONFINITE  an infinity or a NaN went by (their fp16 patterns lie above 65504's), or the record itself is not finite.
// The kernels watch the HIGH fp16 plane they produce anyway: a running packed maximum of |h| (one v_and + one v_pk_max_u16 per PAIR of
// values), folded over the wave at the end of the launch.  frcnn_amax_status gathers the words of a pass's records into one word that
// travels with the outputs.
constexpr unsigned H3_UNDER = 1u, H3_SATURATED = 2u, H3_NONFINITE = 4u;
__device__ __forceinline__ void h3_fp16_saturate() {
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1u);         // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL: an overflowing f16 result clamps to +-MAX, true infinities stay
}
__device__ __forceinline__ void h3_see(unsigned& seen, unsigned hpair) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 m = __builtin_elementwise_max(__builtin_bit_cast(u16x2, seen), __builtin_bit_cast(u16x2, hpair & 0x7fff7fffu));
    seen = __builtin_bit_cast(unsigned, m);
}
__device__ __forceinline__ void h3_see(unsigned& seen, _Float16 h) {
    unsigned short b;
    __builtin_memcpy(&b, &h, 2);
    h3_see(seen, (unsigned)b);
}
__device__ __forceinline__ unsigned h3_status_bits(unsigned m16) {      // m16: the largest |h| bit pattern seen
    return m16 > 0x7bffu ? (H3_UNDER | H3_SATURATED | H3_NONFINITE) : m16 == 0x7bffu ? (H3_UNDER | H3_SATURATED) : m16 >= 0x7800u ? H3_UNDER : 0u;
}
__device__ __forceinline__ void h3_report(const float* rec, unsigned seen) {
    unsigned m = seen & 0xffffu;
    m = m > (seen >> 16) ? m : (seen >> 16);
#pragma unroll
    for (int o = 32; o; o >>= 1) { const unsigned t = __shfl_xor(m, o); m = m > t ? m : t; }
    const unsigned bits = h3_status_bits(m);
    if ((threadIdx.x & 63) == 0 && bits) atomicOr(reinterpret_cast<unsigned*>(const_cast<float*>(rec)) + 1, bits);
}
// a record that is not finite (an Inf in the tensor it was measured on): flagged; the launch goes on under a meaningless scale
__device__ __forceinline__ void h3_check_record(const float* rec, float bound) {
    if (!(bound < __builtin_inff()) && blockIdx.x == 0 && threadIdx.x == 0)
        atomicOr(reinterpret_cast<unsigned*>(const_cast<float*>(rec)) + 1, H3_NONFINITE);
}

constexpr int BK = 32;
constexpr int LDS_STRIDE = BK + 4;     // floats per LDS row (144 B)

__device__ __forceinline__ float activate(float v, int act) {
    if (act == 1) return fmaxf(v, 0.0f);
    if (act == 2) return 1.0f / (1.0f + __expf(-v));
    return v;
}

// Bijective XCD remap (cdna guide T1): consecutive logical ids land on the same XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// fused epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
// S16 (conv_h3.hip): the 32x32 block was accumulated as FOUR 16x16 blocks of v_mfma_f32_16x16x32 (element e: block e >> 2 = 2 * row half +
// column half; inside it col = lane & 15, row = 4 * (lane >> 4) + (e & 3)): a lane owns two columns (16 apart) and eight rows.
template <int TM, int TN, bool S16 = false>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[TM][TN], const ConvArgs& p, int m0, int n0, int wm, int wn, int li, int lh) {
    float vmax1 = 0.0f, vmax2 = 0.0f;                               // max |y| this lane stored, per layer of a paired launch
    const int lane = li + 32 * lh;
#pragma unroll
    for (int jc = 0; jc < (S16 ? 2 * TN : TN); ++jc) {
        const int j = S16 ? jc >> 1 : jc, ci = S16 ? jc & 1 : 0;
        const int n = n0 + wn * TN * 32 + j * 32 + (S16 ? 16 * ci + (lane & 15) : li);
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] : 1.0f;
        const float sh = p.shift ? p.shift[n] : 0.0f;
        const bool second = p.n_split && n >= p.n_split;        // this lane's column belongs to the launch's second layer
        float* const yb = second ? p.y2 : p.y;
        const int ld = second ? p.ldy2 : p.ldy, act = second ? p.act2 : p.act, nn = second ? n - p.n_split : n;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * TM * 32 + i * 32 + (S16 ? 4 * (lane >> 4) : 4 * lh);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (S16 && ((e >> 2) & 1) != ci) continue;              // the other column of this lane
                const int m = S16 ? mb + 16 * (e >> 3) + (e & 3) : mb + (e & 3) + 8 * (e >> 2);
                if (m < p.M) {
                    float v = acc[i][j][e] * sc + sh;
                    if (p.residual) v += p.residual[(size_t)m * p.ldres + n];
                    if (p.mask && !(p.mask[(size_t)m * p.Cout + n] > 0.0f)) v = 0.0f;
                    v = activate(v, act);
                    yb[(size_t)m * ld + nn] = v;
                    if (second) vmax2 = fmaxf(vmax2, fabsf(v)); else vmax1 = fmaxf(vmax1, fabsf(v));
                }
            }
        }
    }
    if (p.y_amax) amax_publish(p.y_amax, vmax1);
    if (p.y2_amax) amax_publish(p.y2_amax, vmax2);
}


typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x3 __attribute__((ext_vector_type(3)));
constexpr unsigned OOB_OFFSET = 0x80000000u;     // >= num_records of any tensor we accept (< 2 GiB)

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

constexpr int SG_VALU = 0x2, SG_MFMA = 0x8, SG_VMEM_RD = 0x20, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

// ------------------------------------------------------------------------------------
// Vectorised epilogue.  In the accumulator layout a lane owns ONE output column, so the plain epilogue above moves
// 4 bytes per lane (sixteen residual loads and sixteen stores per 32x32 tile, each touching two 128-byte row pieces)
// and, issued after the main loop, leaves their latency exposed: on the trunk's 64x64-tile launches it lasted as
// long as the main loop itself (in-kernel timestamps, scripts/micro/conv_lab.hip: res3 2c 10.0 us against 9.0,
// res2 2c 9.4 against 4.2, the head's 512->2048 layers 21.7 against 21.0).  Here the workgroup turns its BM x BN
// tile through LDS (the operand buffers are dead by then) and every thread owns 16-byte pieces of whole rows:
// scale / shift / residual / mask / y all move as b128, one wave instruction covers 1 KB of full row segments, and
// out-of-range rows / columns ride on the buffer descriptors (no branches).  The residual pieces of a 64x64 tile
// are fetched BEFORE the main loop (four registers' worth per thread) when the kernel has no split-K reducer.
// Per element the arithmetic and its order are those of epilogue(): results are bit-identical.
template <int TM, int TN, int WM, int WN>
struct EpiVec {
    static constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
    static constexpr int LD = BN + 4;                    // floats per staged row (16-byte aligned rows)
    static constexpr int C4 = BN / 4;                    // 16-byte pieces per row
    static constexpr int RPP = NT / C4;                  // rows per pass
    static constexpr int PASSES = BM / RPP;
    static constexpr bool fits = (size_t)BM * LD <= (size_t)2 * (BM + BN) * LDS_STRIDE && BM % RPP == 0 && NT % C4 == 0;
};

// byte offsets of this thread's pieces in y (stride ldy) / residual (ldres) / mask (Cout); OOB_OFFSET outside the tensor
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ unsigned epi_piece_off(const ConvArgs& p, int m0, int n0, int tid, int q, int ld) {
    using E = EpiVec<TM, TN, WM, WN>;
    const int m = m0 + q * E::RPP + tid / E::C4, n = n0 + (tid % E::C4) * 4;
    return (m < p.M && n < p.Cout) ? (unsigned)(((size_t)m * ld + n) * 4) : OOB_OFFSET;
}

template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void epi_prefetch_residual(const ConvArgs& p, int m0, int n0, int tid, f32x4 (&rpre)[EpiVec<TM, TN, WM, WN>::PASSES]) {
    using E = EpiVec<TM, TN, WM, WN>;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual : p.x), 0, p.residual ? (int)((size_t)p.M * p.ldres * 4) : 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < E::PASSES; ++q)
        rpre[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, epi_piece_off<TM, TN, WM, WN>(p, m0, n0, tid, q, p.ldres), 0, 0));
}

template <int TM, int TN, int WM, int WN, bool HAVE_PRE>
__device__ __forceinline__ void epilogue_vec(f32x16 (&acc)[TM][TN], const ConvArgs& p, int m0, int n0, int tid, int wm, int wn, int li, int lh,
                                             float* smem, const f32x4* rpre) {
    using E = EpiVec<TM, TN, WM, WN>;
    // two layers in one launch: n_split is a multiple of the tile width here (the host falls back to the scalar epilogue
    // otherwise), so the whole tile belongs to ONE of them -- workgroup-uniform choice of output, stride, activation
    const bool second = p.n_split && n0 >= p.n_split;
    float* const yb = second ? p.y2 : p.y;
    const int y_ld = second ? p.ldy2 : p.ldy, y_act = second ? p.act2 : p.act, y_n0 = second ? p.n_split : 0;
    const int y_cols = p.n_split ? (second ? p.Cout - p.n_split : p.n_split) : p.Cout;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((size_t)p.M * y_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual : p.x), 0, p.residual ? (int)((size_t)p.M * p.ldres * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.mask ? p.mask : p.x), 0, p.mask ? (int)((size_t)p.M * p.Cout * 4) : 0, 0x00020000);
    // big tiles walk their passes in groups of four so the pieces in flight stay within ~32 registers: the 128x128
    // kernels must keep (VGPR + AGPR) <= 256 for two waves per SIMD
    constexpr int GP = E::PASSES < 4 ? E::PASSES : 4;
    static_assert(E::PASSES % GP == 0, "passes must split into whole groups");
    f32x4 rres[GP], rmask[GP];
    auto fetch = [&](int g) {                               // the global reads of group g
        if constexpr (!HAVE_PRE) {
            if (p.residual) {
#pragma unroll
                for (int q = 0; q < GP; ++q)
                    rres[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, epi_piece_off<TM, TN, WM, WN>(p, m0, n0, tid, g * GP + q, p.ldres), 0, 0));
            }
        }
        if (p.mask) {
#pragma unroll
            for (int q = 0; q < GP; ++q)
                rmask[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, epi_piece_off<TM, TN, WM, WN>(p, m0, n0, tid, g * GP + q, p.Cout), 0, 0));
        }
    };
    fetch(0);                                               // in flight while the tile goes through LDS
    const int n = n0 + (tid % E::C4) * 4;
    f32x4 sc = {1.0f, 1.0f, 1.0f, 1.0f}, sh = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n < p.Cout) {
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
    }
    __syncthreads();                                        // every wave is done with the operand buffers (and the split-K flag word)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float* dst = smem + (wm * TM * 32 + i * 32 + 4 * lh) * E::LD + wn * TN * 32 + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * E::LD] = acc[i][j][e];
        }
    __syncthreads();
    const float* src = smem + (tid / E::C4) * E::LD + (tid % E::C4) * 4;
    float vmax = 0.0f;
#pragma unroll 1
    for (int g = 0; g < E::PASSES / GP; ++g) {
        if (g) fetch(g);
#pragma unroll
        for (int q = 0; q < GP; ++q) {
            const int pass = g * GP + q;
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + pass * E::RPP * E::LD);
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = a[c] * sc[c] + sh[c];
                if (p.residual) t += HAVE_PRE ? rpre[q][c] : rres[q][c];
                if (p.mask && !(rmask[q][c] > 0.0f)) t = 0.0f;
                v[c] = activate(t, y_act);
            }
            const int ym = m0 + pass * E::RPP + tid / E::C4, yn = n0 - y_n0 + (tid % E::C4) * 4;
            const bool in = ym < p.M && yn < y_cols;
            const unsigned yoff = in ? (unsigned)(((size_t)ym * y_ld + yn) * 4) : OOB_OFFSET;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yrsrc, yoff, 0, 0);
            if (in) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
    if (float* rec = second ? p.y2_amax : p.y_amax) amax_publish(rec, vmax);
}


// ------------------------------------------------------------------------------------
// Shared by the two split engines (conv_x6.hip: three bf16 planes per operand; conv_h3.hip: two fp16 planes): the LDS image of
// an operand plane -- unpadded 64-byte rows (32 k of 2 bytes), the 16-byte slot s of row r stored at slot s ^ ((r >> 2) & 3) -- the
// tile's LDS budget, and the 16-byte epilogue that turns the tile through that LDS one wave-row at a time.
constexpr int X6_ROWB = 64;           // LDS bytes per row per plane
// 16-byte slot s of LDS row r is stored at slot s ^ x6_swz(r): f = [0, 2, 3, 1] over the row's group of four (r >> 2 & 3).  Under f the
// fragment reads of BOTH matrix-instruction shapes are conflict-free (ds_read_b128 serves four groups of 16 lanes; with the 16x16x32
// fragment -- lane l: row l & 15, slot l >> 4 -- the identity map would put rows 0-3 / slot 0 and rows 4-7 / slot 1 of one group on the
// same banks); writers cover whole rows, so any per-group bijection costs them nothing.
__device__ __forceinline__ int x6_swz(int row) { return (0x1320 >> (4 * ((row >> 2) & 3))) & 3; }


template <int TM, int TN, int WM, int WN, int PLANES = 3>
struct X6Tile {
    static constexpr int NT = 64 * WM * WN, BM = 32 * TM * WM, BN = 32 * TN * WN;
    static constexpr size_t planes = (size_t)PLANES * (BM + BN) * X6_ROWB;
    static constexpr size_t epi = (size_t)(32 * TM) * (BN + 4) * 4;     // the 16-byte epilogue: one wave-row of the tile at a time
    static constexpr size_t lds = planes > epi ? planes : epi;
};

// The 16-byte epilogue of conv_f32_common.h (epilogue_vec: same arithmetic per element, same order) with the tile turned through
// LDS one WAVE-ROW at a time: pass h stages the 32 TM rows owned by the waves with wm == h, every thread then owns 16-byte pieces
// of whole rows (scale / shift / residual / mask / y as b128, out-of-range pieces on the buffer descriptors).
// PLANES (conv_h3.hip): the tile also leaves as two fp16 planes under the scale y_scale (a power of two): ah = f16(v * s), al = f16((v * s - ah) * 2^11),
// 8-byte pieces at [plane][row][column]; the f32 store is skipped when the launch has no f32 output.
template <int TM, int TN, int WM, int WN, bool PLANES = false, bool S16 = false>
__device__ __forceinline__ void x6_epilogue_vec(f32x16 (&acc)[TM][TN], const ConvArgs& p, int m0, int n0, int tid, int wm, int wn, int li, int lh, float* smem,
                                                float y_scale = 1.0f) {
    constexpr int NT = 64 * WM * WN, BN = 32 * TN * WN, HB = 32 * TM, LD = BN + 4, C4 = BN / 4, RPP = NT / C4, PASSES = HB / RPP;
    static_assert(HB % RPP == 0 && NT % C4 == 0 && PASSES >= 1 && PASSES <= 8, "epilogue passes");
    const bool second = p.n_split && n0 >= p.n_split;       // two layers in one launch: the tile belongs to ONE of them (host guarantees it)
    float* const yb = second ? p.y2 : p.y;
    const int y_ld = second ? p.ldy2 : p.ldy, y_act = second ? p.act2 : p.act, y_n0 = second ? p.n_split : 0;
    const int y_cols = p.n_split ? (second ? p.Cout - p.n_split : p.n_split) : p.Cout;
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yb, 0, (int)((size_t)p.M * y_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(PLANES ? p.y_planes : (void*)yb, 0, PLANES ? (int)((size_t)2 * p.M * p.Cout * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual : p.x), 0, p.residual ? (int)((size_t)p.M * p.ldres * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.mask ? p.mask : p.x), 0, p.mask ? (int)((size_t)p.M * p.Cout * 4) : 0, 0x00020000);
    const int prow = tid / C4, pcol = (tid % C4) * 4, n = n0 + pcol;
    // residual handed over as fp16 planes: value = (hi + lo / 2048) * 2^-e, e the producing launch's exponent (one device word)
    const __amdgpu_buffer_rsrc_t rprsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.res_planes ? p.res_planes : (const void*)p.x), 0, p.res_planes ? (int)((size_t)2 * p.M * p.Cout * 2) : 0, 0x00020000);
    float res_unscale = 1.0f;
    if (p.res_planes) { const unsigned b = (unsigned)(127 - *p.res_pexp) << 23; __builtin_memcpy(&res_unscale, &b, 4); }
    f32x4 sc = {1.0f, 1.0f, 1.0f, 1.0f}, sh = {0.0f, 0.0f, 0.0f, 0.0f};
    if (n < p.Cout) {
        if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
        if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n);
    }
    f32x4 rres[PASSES], rmask[PASSES];
    float vmax = 0.0f;
    unsigned pseen = 0u;
    auto fetch = [&](int h) {                                // the global reads of wave-row h: in flight while it goes through LDS
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const int m = m0 + h * HB + q * RPP + prow;
            const bool in = m < p.M && n < p.Cout;
            if (p.residual) rres[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, in ? (unsigned)(((size_t)m * p.ldres + n) * 4) : OOB_OFFSET, 0, 0));
            else if (p.res_planes) {                         // (8 + 8 bytes: the same bytes per element as the f32 tensor)
                typedef _Float16 f16x4r __attribute__((ext_vector_type(4)));
                typedef int i32x2r __attribute__((ext_vector_type(2)));
                const unsigned off = in ? (unsigned)(((size_t)m * p.Cout + n) * 2) : OOB_OFFSET;
                const f16x4r h = __builtin_bit_cast(f16x4r, (i32x2r)__builtin_amdgcn_raw_buffer_load_b64(rprsrc, off, 0, 0));
                const f16x4r l = __builtin_bit_cast(f16x4r, (i32x2r)__builtin_amdgcn_raw_buffer_load_b64(rprsrc, off == OOB_OFFSET ? OOB_OFFSET : off + (unsigned)((size_t)p.M * p.Cout * 2), 0, 0));
#pragma unroll
                for (int c = 0; c < 4; ++c) rres[q][c] = ((float)h[c] + (float)l[c] * (1.0f / 2048.0f)) * res_unscale;
            }
            if (p.mask) rmask[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrsrc, in ? (unsigned)(((size_t)m * p.Cout + n) * 4) : OOB_OFFSET, 0, 0));
        }
    };
#pragma unroll 1
    for (int h = 0; h < WM; ++h) {
        fetch(h);
        if (h) __syncthreads();                              // the previous wave-row has been read out (the caller synchronised before pass 0)
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (S16) {                     // four 16x16 blocks (epilogue()'s note): block e >> 2, col = lane & 15, row = 4 * (lane >> 4) + (e & 3)
                        const int lane = li + 32 * lh;
                        float* dst = smem + (i * 32 + 4 * (lane >> 4)) * LD + wn * TN * 32 + j * 32 + (lane & 15);
#pragma unroll
                        for (int e = 0; e < 16; ++e) dst[(16 * (e >> 3) + (e & 3)) * LD + 16 * ((e >> 2) & 1)] = acc[i][j][e];
                    } else {
                        float* dst = smem + (i * 32 + 4 * lh) * LD + wn * TN * 32 + j * 32 + li;
#pragma unroll
                        for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * LD] = acc[i][j][e];
                    }
                }
        }
        __syncthreads();
        const float* src = smem + prow * LD + pcol;
#pragma unroll
        for (int q = 0; q < PASSES; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + q * RPP * LD);
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float t = a[c] * sc[c] + sh[c];
                if (p.residual || p.res_planes) t += rres[q][c];
                if (p.mask && !(rmask[q][c] > 0.0f)) t = 0.0f;
                v[c] = activate(t, y_act);
            }
            const int ym = m0 + h * HB + q * RPP + prow, yn = n0 - y_n0 + pcol;
            const bool in = ym < p.M && yn < y_cols;
            const unsigned yoff = in ? (unsigned)(((size_t)ym * y_ld + yn) * 4) : OOB_OFFSET;
            if (!PLANES || yb) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v), yrsrc, yoff, 0, 0);
            if (in) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
            if constexpr (PLANES) {
                typedef _Float16 f16x4e __attribute__((ext_vector_type(4)));
                typedef int i32x2 __attribute__((ext_vector_type(2)));
                f16x4e h, l;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xs = v[c] * y_scale;
                    const _Float16 a1 = (_Float16)xs;
                    h[c] = a1; l[c] = (_Float16)((xs - (float)a1) * 2048.0f);
                }
                if (in) { const i32x2 hp = __builtin_bit_cast(i32x2, h); h3_see(pseen, (unsigned)hp[0]); h3_see(pseen, (unsigned)hp[1]); }
                const unsigned poff = in ? (unsigned)(((size_t)ym * p.Cout + yn) * 2) : OOB_OFFSET;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, h), prsrc, poff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, l), prsrc, poff == OOB_OFFSET ? OOB_OFFSET : poff + (unsigned)((size_t)p.M * p.Cout * 2), 0, 0);
            }
        }
    }
    if (float* rec = second ? p.y2_amax : p.y_amax) {
        amax_publish(rec, vmax);
        if constexpr (PLANES) h3_report(rec, pseen);      // planes written under a bound (bound_c * max|x| + bound_d) that the values exceeded
    }
}


// conv_x6.hip: the same GEMM on the bf16 matrix cores (exact three-way operand split); `a.w` = the three filter planes
int launch_conv_x6(const ConvArgs& a, int cfg, hipStream_t s);
int x6_tile_width(int cfg);
// conv_h3.hip: the same GEMM on the fp16 matrix cores (two-way split with a scaled low part); `a.w` = header + two filter planes
int launch_conv_h3(const ConvArgs& a, int cfg, hipStream_t s);
int h3_tile_width(int cfg);

}  // namespace frcnn
