// The ResNet stem as ONE launch on the bf16 matrix cores (gfx950), for the bf16 conv path (BASELINE configs[3] / [4]):
//   conv1 7x7, strides (2,2), padding 'same', 3 -> 64 channels      resnet.py:408 (resnet101: :565)
//   BatchNormalization(training=False) [+ Scale for ResNet-101]      resnet.py:410 (:566-567), folded to scale / shift
//   Activation('relu')                                               resnet.py:411
//   MaxPooling2D((3,3), strides=(2,2))  (VALID)                      resnet.py:412
//   the bf16 cast that opens the bf16 trunk (nets.ResNetBase)
// Until round 3 this was three launches in f32 -- the 3-channel implicit-GEMM kernel (69 TFLOP/s on v_mfma_f32_32x32x2_f32:
// 61 us per 600x1500 image), the pool (reads 57.6 MB, writes 13.3 MB per image) and the cast -- i.e. ~0.09 ms of a 1.06 ms
// image and 166 MB of the trunk's traffic.  Here a workgroup owns a 4 x 16 patch of POOLED pixels: it stages the 23 x 71 x 3
// input pixels they depend on in LDS (rounded to bf16 once: this is what makes the stem a bf16 conv), multiplies the
// 9 x 33 conv pixels under the patch against the whole 64-channel filter with v_mfma_f32_32x32x16_bf16 (f32 accumulate),
// applies scale / shift / ReLU in f32, rounds to bf16 into an LDS tile, takes the 3x3 / stride-2 maxima there and writes
// 16-byte pieces of the pooled NHWC tensor.  Per image: 10.8 MB read, 6.7 MB written, ~20 GFLOP-equivalent at the bf16 rate.
//
// max-pool and the bf16 rounding commute (round-to-nearest-even is monotonic), so rounding BEFORE the pool gives the bits
// the old pool -> cast order gave for the same conv values; the conv values themselves now come from bf16-rounded pixels
// and filter taps (the oracle's storage model follows: oracle/keras_ref.py `bf16_stem`).
//
// k order of the GEMM: k = r * 24 + s * 3 + c for filter row r, column s, channel c (21 real values per filter row, 3 zero
// columns), 7 rows + 1 zero row = 176 = 11 MFMA k-steps.  For a conv pixel (cy, cx) the 21 values of filter row r are the
// 21 CONTIGUOUS bf16 of input row 2 cy + r starting at pixel 2 cx, so lane (i, h) of k-step t reads the 8 bf16 at
// element (2 cy + r) * ROW + 6 cx + j0 with g = 2 t + h, r = g / 3, j0 = 8 (g % 3): four ds_read_b32 (4-byte aligned).
// Whatever lies behind the 21st value (the next pixel, the next row) meets a zero filter column.
#include "common.h"

namespace frcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int ST_PH = 4, ST_PW = 16;                         // pooled pixels per workgroup
constexpr int ST_CH = 2 * ST_PH + 1, ST_CW = 2 * ST_PW + 1;  // conv pixels under them: 9 x 33
constexpr int ST_CM = ST_CH * ST_CW;                         // 297 GEMM rows
constexpr int ST_MT = (ST_CM + 31) / 32;                     // 10 row tiles of 32
constexpr int ST_IH = 2 * ST_CH + 5, ST_IW = 2 * ST_CW + 5;  // input pixels: 23 x 71
constexpr int ST_ROW = 216;                                  // bf16 elements per staged input row (71 * 3 = 213, padded)
constexpr int ST_PROWS = ST_IH + 2;                          // + the zero filter row's reads and the tail
constexpr int ST_K = 176, ST_WLD = 184;                      // GEMM depth; filter row stride in LDS (368 B: 16-byte aligned, off the 256-B bank period)
constexpr int ST_CLD = 72;                                   // conv tile row stride in LDS (144 B)
constexpr size_t ST_LDS = (size_t)ST_PROWS * ST_ROW * 2 + (size_t)64 * ST_WLD * 2 + (size_t)ST_CM * ST_CLD * 2;

__global__ void __launch_bounds__(256) k_stem_bf16(const float* __restrict__ x, const __bf16* __restrict__ wp, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, int H, int W, int Hp, int Wp, __bf16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char st_smem[];
    __bf16* patch = reinterpret_cast<__bf16*>(st_smem);                                  // [ST_PROWS][ST_ROW]
    __bf16* wl = patch + ST_PROWS * ST_ROW;                                              // [64][ST_WLD]
    __bf16* ct = wl + 64 * ST_WLD;                                                       // [ST_CM][ST_CLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int px0 = blockIdx.x * ST_PW, py0 = blockIdx.y * ST_PH, img = blockIdx.z;
    // TF 'same' for 7x7 / 2: out = ceil(in / 2), pad_along = max((out - 1) * 2 + 7 - in, 0), before = pad_along / 2
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pad_t = max((Ho - 1) * 2 + 7 - H, 0) / 2, pad_l = max((Wo - 1) * 2 + 7 - W, 0) / 2;
    const int iy0 = 4 * py0 - pad_t, ix0 = 4 * px0 - pad_l;          // input pixel of patch element (0, 0)
    const float* xi = x + (size_t)img * H * W * 3;

    // ---- stage the input patch (f32 -> bf16, zeros outside the image and in the padding columns / rows).  All of a thread's
    // loads are issued before the first one is used: as a load -> convert -> store loop this phase was a chain of 22 dependent
    // memory round trips per thread and the whole kernel ran at its old f32 pace (59 us per 600x1500 image)
    constexpr int ST_NP = (ST_PROWS * ST_ROW + 255) / 256;
    float pv[ST_NP];
#pragma unroll
    for (int q = 0; q < ST_NP; ++q) {
        const int idx = tid + q * 256;
        const int row = idx / ST_ROW, col = idx - row * ST_ROW;
        const int gy = iy0 + row, gx = ix0 + col / 3;
        const bool ok = idx < ST_PROWS * ST_ROW && row < ST_IH && col < ST_IW * 3 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        pv[q] = ok ? xi[(size_t)gy * W * 3 + (ix0 * 3 + col)] : 0.0f;      // (gx >= 0 here, so the column offset is too)
    }
#pragma unroll
    for (int q = 0; q < ST_NP; ++q) {
        const int idx = tid + q * 256;
        if (idx < ST_PROWS * ST_ROW) patch[idx] = (__bf16)pv[q];
    }
    // ---- the whole filter: [64][176] bf16 -> LDS rows of ST_WLD
    for (int idx = tid; idx < 64 * (ST_K / 8); idx += 256) {
        const int n = idx / (ST_K / 8), c8 = idx - n * (ST_K / 8);
        *reinterpret_cast<i32x4*>(wl + n * ST_WLD + c8 * 8) = *reinterpret_cast<const i32x4*>(wp + n * ST_K + c8 * 8);
    }
    __syncthreads();

    // ---- 9 x 33 conv pixels x 64 channels on the matrix cores; wave w takes row tiles w, w + 4, w + 8
    for (int mt = wave; mt < ST_MT; mt += 4) {
        const int m = mt * 32 + li, mm = m < ST_CM ? m : 0;
        const int cy = mm / ST_CW, cx = mm - cy * ST_CW;
        const __bf16* arow = patch + (2 * cy) * ST_ROW + 6 * cx;
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.0f; acc1[e] = 0.0f; }
#pragma unroll
        for (int kt = 0; kt < ST_K / 16; ++kt) {
            const int g = 2 * kt + lh, r = g / 3, j0 = (g - 3 * r) * 8;
            const unsigned* ap = reinterpret_cast<const unsigned*>(arow + r * ST_ROW + j0);     // 4-byte aligned: 6 cx, 8 j and ST_ROW are even
            const i32x4 av = {(int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3]};
            const bf16x8 fa = __builtin_bit_cast(bf16x8, av);
            const bf16x8 fb0 = *reinterpret_cast<const bf16x8*>(wl + li * ST_WLD + kt * 16 + lh * 8);
            const bf16x8 fb1 = *reinterpret_cast<const bf16x8*>(wl + (32 + li) * ST_WLD + kt * 16 + lh * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb1, acc1, 0, 0, 0);
        }
        // scale / shift / ReLU in f32, one rounding, into the conv tile (C/D map: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5))
        const float sc0 = scale[li], sh0 = shift[li], sc1 = scale[32 + li], sh1 = shift[32 + li];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (row < ST_CM) {
                ct[row * ST_CLD + li] = (__bf16)fmaxf(acc0[e] * sc0 + sh0, 0.0f);
                ct[row * ST_CLD + 32 + li] = (__bf16)fmaxf(acc1[e] * sc1 + sh1, 0.0f);
            }
        }
    }
    __syncthreads();

    // ---- 3x3 / stride-2 maxima (values are >= +0 after the ReLU: bf16 order == unsigned order of the bit patterns)
    for (int item = tid; item < ST_PH * ST_PW * 8; item += 256) {
        const int pix = item >> 3, c8 = item & 7;
        const int ppy = pix / ST_PW, ppx = pix - ppy * ST_PW;
        const int py = py0 + ppy, px = px0 + ppx;
        if (py >= Hp || px >= Wp) continue;
        u16x8 best = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const u16x8 v = *reinterpret_cast<const u16x8*>(ct + ((2 * ppy + dy) * ST_CW + 2 * ppx + dx) * ST_CLD + c8 * 8);
#pragma unroll
                for (int q = 0; q < 8; ++q) best[q] = v[q] > best[q] ? v[q] : best[q];
            }
        *reinterpret_cast<u16x8*>(out + (((size_t)img * Hp + py) * Wp + px) * 64 + c8 * 8) = best;
    }
}

// HWIO f32 [7][7][3][64] -> bf16 [64][176], k = r * 24 + s * 3 + c, zeros elsewhere
__global__ void k_pack_stem_bf16(const float* w, __bf16* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * ST_K; i += gridDim.x * blockDim.x) {
        const int n = i / ST_K, k = i - n * ST_K, r = k / 24, j = k - r * 24, s = j / 3, c = j - s * 3;
        out[i] = (r < 7 && j < 21) ? (__bf16)w[((r * 7 + s) * 3 + c) * 64 + n] : (__bf16)0.0f;
    }
}

}  // namespace frcnn

using namespace frcnn;

extern "C" {

int frcnn_stem_bf16_packed_elems(void) { return 64 * ST_K; }

int frcnn_pack_stem_weights_bf16(const float* w_hwio, void* packed_bf16, void* stream) {
    if (!w_hwio || !packed_bf16) return fail(FRCNN_E_ARG, "pack_stem_weights_bf16: null pointer");
    k_pack_stem_bf16<<<44, 256, 0, as_stream(stream)>>>(w_hwio, (__bf16*)packed_bf16);
    return check_launch("pack_stem_weights_bf16");
}

int frcnn_stem_bf16_fwd(const float* x, int n, int h, int w, const void* w_packed_bf16, const float* scale, const float* shift,
                        void* out_bf16, void* stream) {
    if (!x || !w_packed_bf16 || !scale || !shift || !out_bf16) return fail(FRCNN_E_ARG, "stem_bf16_fwd: null pointer");
    if (n <= 0 || h < 7 || w < 7) return fail(FRCNN_E_ARG, "stem_bf16_fwd: bad shape");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2, hp = (ho - 3) / 2 + 1, wp = (wo - 3) / 2 + 1;
    if (hp <= 0 || wp <= 0) return fail(FRCNN_E_ARG, "stem_bf16_fwd: image too small for the 3x3 pool");
    if ((reinterpret_cast<uintptr_t>(w_packed_bf16) | reinterpret_cast<uintptr_t>(out_bf16)) & 15) return fail(FRCNN_E_ARG, "stem_bf16_fwd: 16-byte aligned tensors");
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_stem_bf16, ST_LDS, "stem_bf16_fwd")) return e;
    const dim3 grid((wp + ST_PW - 1) / ST_PW, (hp + ST_PH - 1) / ST_PH, n);
    if (grid.y > 65535 || grid.z > 65535) return fail(FRCNN_E_ARG, "stem_bf16_fwd: image too large");
    k_stem_bf16<<<grid, 256, ST_LDS, as_stream(stream)>>>(x, (const __bf16*)w_packed_bf16, scale, shift, h, w, hp, wp, (__bf16*)out_bf16);
    return check_launch("stem_bf16_fwd");
}

}  // extern "C"
