// The ResNet stem as ONE fp32-grade launch on the fp16 matrix cores (gfx950), for the f16x3 conv engine (csrc/conv_h3.hip):
//   conv1 7x7, strides (2,2), padding 'same', 3 -> 64 channels      resnet.py:408 (resnet101: :565)
//   BatchNormalization(training=False) [+ Scale for ResNet-101]      resnet.py:410 (:566-567), folded to scale / shift
//   Activation('relu')                                               resnet.py:411
//   MaxPooling2D((3,3), strides=(2,2))  (VALID)                      resnet.py:412
// f32 image in, f32 pooled map out.  Until round 5 the fp32 path ran this as two launches -- the 3-channel implicit-GEMM kernel on
// v_mfma_f32_32x32x2_f32 (51 us per 600x1000 image alone on the chip) and k_pool (38 MB read, 9.5 MB written) -- 131 + 185 us of
// kernel time per image with eight images in flight (profiles/round5_trace_default_kernel_stats.csv: 6 % of all kernel time, the
// 38 MB conv map of eight images at once does not stay in L2).  Here a workgroup owns a 4 x 16 patch of POOLED pixels, the structure of
// k_stem_bf16 (stem_bf16.hip): it stages the 23 x 71 x 3 input pixels they depend on in LDS -- as the f16x3 engine's TWO fp16 planes,
// hi = f16(a * 2^eA), lo = f16((a * 2^eA - hi) * 2^11), eA from the image's magnitude record -- multiplies the 9 x 33 conv pixels under
// the patch against the whole 64-channel filter (two planes as well) with three v_mfma_f32_32x32x16_f16 per block of products into
// two f32 accumulators, applies scale / shift / ReLU in f32 into an f32 LDS tile, takes the 3x3 / stride-2 maxima there and writes
// 16-byte pieces of the pooled NHWC tensor; max|y| goes to the output's magnitude record like every f16x3 launch's.
// The arithmetic per conv pixel is the f16x3 engine's (conv_h3.hip: operands to one unit in the last place, the al * bl term dropped,
// error against fp64 under the native f32 MFMA's); max-pooling selects, it does not round.
//
// k order of the GEMM (k_stem_bf16's): k = r * 24 + s * 3 + c for filter row r, column s, channel c (21 real values per filter row,
// 3 zero columns), 7 rows + 1 zero row = 176 = 11 MFMA k-steps; for a conv pixel (cy, cx) the 21 values of filter row r are 21
// CONTIGUOUS elements of input row 2 cy + r starting at pixel 2 cx, so a lane's 8-element fragment is four ds_read_b32 per plane.
#include "conv_f32_common.h"

namespace frcnn {

typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));

constexpr int SH_PH = 4, SH_PW = 16;                         // pooled pixels per workgroup
constexpr int SH_CH = 2 * SH_PH + 1, SH_CW = 2 * SH_PW + 1;  // conv pixels under them: 9 x 33
constexpr int SH_CM = SH_CH * SH_CW;                         // 297 GEMM rows
constexpr int SH_MT = (SH_CM + 31) / 32;                     // 10 row tiles of 32
constexpr int SH_IH = 2 * SH_CH + 5, SH_IW = 2 * SH_CW + 5;  // input pixels: 23 x 71
constexpr int SH_ROW = 216;                                  // fp16 elements per staged input row per plane (71 * 3 = 213, padded)
constexpr int SH_PROWS = SH_IH + 2;                          // + the zero filter row's reads and the tail
constexpr int SH_K = 176, SH_WLD = 184;                      // GEMM depth; filter row stride in LDS (368 B: 16-byte aligned, off the 256-B bank period)
constexpr int SH_CLD = 68;                                   // conv tile row stride in LDS, floats (272 B)
constexpr int SH_NT = 512;                                   // eight waves: one workgroup per CU (150 KB of LDS)
constexpr size_t SH_PATCH = (size_t)SH_PROWS * SH_ROW * 2, SH_FILT = (size_t)64 * SH_WLD * 2;       // bytes of ONE plane
constexpr size_t SH_LDS = 2 * SH_PATCH + 2 * SH_FILT + (size_t)SH_CM * SH_CLD * 4;
constexpr int SH_HEADER = 16;                                // in front of the packed filter planes: word 0 = max|w|

__host__ __device__ __forceinline__ int sh_exponent(float amax) {        // conv_h3.hip h3_exponent: amax * 2^e in [2^14, 2^15)
    unsigned b;
    __builtin_memcpy(&b, &amax, 4);
    const int e = 141 - (int)((b >> 23) & 0xffu);
    return e > 100 ? 100 : (e < -100 ? -100 : e);
}
__host__ __device__ __forceinline__ float sh_pow2(int e) {
    const unsigned b = (unsigned)(e + 127) << 23;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

__global__ void __launch_bounds__(SH_NT) k_stem_h3(const float* __restrict__ x, const float* __restrict__ x_amax, const char* __restrict__ wp,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int H, int W, int Hp, int Wp,
                                                   float* __restrict__ out, float* __restrict__ y_amax) {
    extern __shared__ __attribute__((aligned(16))) char sh_smem[];
    _Float16* patch = reinterpret_cast<_Float16*>(sh_smem);                               // [2][SH_PROWS][SH_ROW]
    _Float16* wl = reinterpret_cast<_Float16*>(sh_smem + 2 * SH_PATCH);                   // [2][64][SH_WLD]
    float* ct = reinterpret_cast<float*>(sh_smem + 2 * SH_PATCH + 2 * SH_FILT);          // [SH_CM][SH_CLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int px0 = blockIdx.x * SH_PW, py0 = blockIdx.y * SH_PH, img = blockIdx.z;
    // TF 'same' for 7x7 / 2: out = ceil(in / 2), pad_along = max((out - 1) * 2 + 7 - in, 0), before = pad_along / 2
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int pad_t = max((Ho - 1) * 2 + 7 - H, 0) / 2, pad_l = max((Wo - 1) * 2 + 7 - W, 0) / 2;
    const int iy0 = 4 * py0 - pad_t, ix0 = 4 * px0 - pad_l;          // input pixel of patch element (0, 0)
    const float* xi = x + (size_t)img * H * W * 3;

    // ---- the input patch: every load of a thread is issued before the first one is used (k_stem_bf16's lesson)
    constexpr int NP = (SH_PROWS * SH_ROW + SH_NT - 1) / SH_NT;
    float pv[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int idx = tid + q * SH_NT;
        const int row = idx / SH_ROW, col = idx - row * SH_ROW;
        const int gy = iy0 + row, gx = ix0 + col / 3;
        const bool ok = idx < SH_PROWS * SH_ROW && row < SH_IH && col < SH_IW * 3 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        pv[q] = ok ? xi[(size_t)gy * W * 3 + (ix0 * 3 + col)] : 0.0f;      // (gx >= 0 here, so the column offset is too)
    }
    h3_fp16_saturate();                                                     // (conv_f32_common.h: the engine's fences)
    const float x_max = amax_read(x_amax);
    const int eA = sh_exponent(x_max);
    const int eB = sh_exponent(*reinterpret_cast<const float*>(wp));
    const float sA = sh_pow2(eA);
    h3_check_record(x_amax, x_max);
    unsigned seen = 0u;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int idx = tid + q * SH_NT;
        if (idx < SH_PROWS * SH_ROW) {
            const float xs = pv[q] * sA;                                    // exact
            const _Float16 hi = (_Float16)xs;
            h3_see(seen, hi);
            patch[idx] = hi;
            patch[SH_PROWS * SH_ROW + idx] = (_Float16)((xs - (float)hi) * 2048.0f);
        }
    }
    h3_report(x_amax, seen);
    // ---- the whole filter, both planes: [2][64][176] fp16 -> LDS rows of SH_WLD
    const _Float16* wsrc = reinterpret_cast<const _Float16*>(wp + SH_HEADER);
    for (int idx = tid; idx < 2 * 64 * (SH_K / 8); idx += SH_NT) {
        const int pl = idx / (64 * (SH_K / 8)), r = idx - pl * (64 * (SH_K / 8)), n = r / (SH_K / 8), c8 = r - n * (SH_K / 8);
        *reinterpret_cast<i32x4*>(wl + pl * 64 * SH_WLD + n * SH_WLD + c8 * 8) = *reinterpret_cast<const i32x4*>(wsrc + pl * 64 * SH_K + n * SH_K + c8 * 8);
    }
    __syncthreads();

    // ---- 9 x 33 conv pixels x 64 channels: units (row tile, column tile) dealt round robin to the eight waves
    const float unscale_a = sh_pow2(-eA), unscale_b = sh_pow2(-eB);
    for (int u = wave; u < 2 * SH_MT; u += SH_NT / 64) {
        const int mt = u >> 1, ct_col = (u & 1) * 32;
        const int m = mt * 32 + li, mm = m < SH_CM ? m : 0;
        const int cy = mm / SH_CW, cx = mm - cy * SH_CW;
        const _Float16* arow = patch + (2 * cy) * SH_ROW + 6 * cx;
        const _Float16* brow = wl + (ct_col + li) * SH_WLD + lh * 8;
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.0f; acc1[e] = 0.0f; }
#pragma unroll
        for (int kt = 0; kt < SH_K / 16; ++kt) {
            const int g = 2 * kt + lh, r = g / 3, j0 = (g - 3 * r) * 8;
            const unsigned* ah = reinterpret_cast<const unsigned*>(arow + r * SH_ROW + j0);      // 4-byte aligned: 6 cx, 8 j and SH_ROW are even
            const unsigned* al = ah + SH_PROWS * SH_ROW / 2;
            const i32x4 avh = {(int)ah[0], (int)ah[1], (int)ah[2], (int)ah[3]}, avl = {(int)al[0], (int)al[1], (int)al[2], (int)al[3]};
            const sf16x8 fah = __builtin_bit_cast(sf16x8, avh), fal = __builtin_bit_cast(sf16x8, avl);
            const sf16x8 fbh = *reinterpret_cast<const sf16x8*>(brow + kt * 16);
            const sf16x8 fbl = *reinterpret_cast<const sf16x8*>(brow + 64 * SH_WLD + kt * 16);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal, fbh, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbl, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah, fbh, acc0, 0, 0, 0);
        }
        // scale / shift / ReLU in f32 into the conv tile (C/D map: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5))
        const float sc = scale[ct_col + li], sh = shift[ct_col + li];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (row < SH_CM) {
                const float conv = ((acc0[e] + acc1[e] * (1.0f / 2048.0f)) * unscale_a) * unscale_b;
                ct[row * SH_CLD + ct_col + li] = fmaxf(conv * sc + sh, 0.0f);
            }
        }
    }
    __syncthreads();

    // ---- 3x3 / stride-2 maxima, 16-byte pieces of the pooled NHWC tensor; max|y| for the magnitude record (values are >= 0)
    float vmax = 0.0f;
    for (int item = tid; item < SH_PH * SH_PW * 16; item += SH_NT) {
        const int pix = item >> 4, c4 = item & 15;
        const int ppy = pix / SH_PW, ppx = pix - ppy * SH_PW;
        const int py = py0 + ppy, px = px0 + ppx;
        if (py < Hp && px < Wp) {
            f32x4 best = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(ct + ((2 * ppy + dy) * SH_CW + 2 * ppx + dx) * SH_CLD + c4 * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) best[q] = fmaxf(best[q], v[q]);
                }
            *reinterpret_cast<f32x4*>(out + (((size_t)img * Hp + py) * Wp + px) * 64 + c4 * 4) = best;
            vmax = fmaxf(fmaxf(vmax, fmaxf(best[0], best[1])), fmaxf(best[2], best[3]));
        }
    }
    if (y_amax) amax_publish(y_amax, vmax);
}

// HWIO f32 [7][7][3][64] -> header (max|w|) + two fp16 planes [2][64][176], k = r * 24 + s * 3 + c, zeros elsewhere
__global__ void k_stem_h3_wmax(const float* w, unsigned* header) {
    float v = 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 7 * 7 * 3 * 64; i += gridDim.x * blockDim.x) v = fmaxf(v, fabsf(w[i]));
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0 && v > 0.0f) atomicMax(header, __float_as_uint(v));
}

__global__ void k_pack_stem_h3(const float* w, char* packed) {
    const float s = sh_pow2(sh_exponent(*reinterpret_cast<const float*>(packed)));
    _Float16* out = reinterpret_cast<_Float16*>(packed + SH_HEADER);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * SH_K; i += gridDim.x * blockDim.x) {
        const int n = i / SH_K, k = i - n * SH_K, r = k / 24, j = k - r * 24, sx = j / 3, c = j - sx * 3;
        const float v = (r < 7 && j < 21) ? w[((r * 7 + sx) * 3 + c) * 64 + n] * s : 0.0f;
        const _Float16 hi = (_Float16)v;
        out[i] = hi;
        out[64 * SH_K + i] = (_Float16)((v - (float)hi) * 2048.0f);
    }
}

__global__ void k_stem_h3_zero_header(unsigned* header) { if (threadIdx.x < SH_HEADER / 4) header[threadIdx.x] = 0u; }

}  // namespace frcnn

using namespace frcnn;

extern "C" {

size_t frcnn_stem_h3_packed_bytes(void) { return (size_t)SH_HEADER + (size_t)2 * 64 * SH_K * 2; }

int frcnn_pack_stem_weights_h3(const float* w_hwio, void* packed, void* stream) {
    if (!w_hwio || !packed) return fail(FRCNN_E_ARG, "pack_stem_weights_h3: null pointer");
    if (reinterpret_cast<uintptr_t>(packed) & 15) return fail(FRCNN_E_ARG, "pack_stem_weights_h3: 16-byte aligned buffer required");
    hipStream_t s = as_stream(stream);
    k_stem_h3_zero_header<<<1, 64, 0, s>>>(reinterpret_cast<unsigned*>(packed));
    k_stem_h3_wmax<<<37, 256, 0, s>>>(w_hwio, reinterpret_cast<unsigned*>(packed));
    k_pack_stem_h3<<<44, 256, 0, s>>>(w_hwio, reinterpret_cast<char*>(packed));
    return check_launch("pack_stem_weights_h3");
}

int frcnn_stem_h3_fwd(const float* x, const float* x_amax, int n, int h, int w, const void* w_packed, const float* scale, const float* shift,
                      float* out, float* y_amax, void* stream) {
    if (!x || !x_amax || !w_packed || !scale || !shift || !out) return fail(FRCNN_E_ARG, "stem_h3_fwd: null pointer (the image's magnitude record is required)");
    if (n <= 0 || h < 7 || w < 7) return fail(FRCNN_E_ARG, "stem_h3_fwd: bad shape");
    const int ho = (h + 1) / 2, wo = (w + 1) / 2, hp = (ho - 3) / 2 + 1, wp = (wo - 3) / 2 + 1;
    if (hp <= 0 || wp <= 0) return fail(FRCNN_E_ARG, "stem_h3_fwd: image too small for the 3x3 pool");
    if ((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(out)) & 15) return fail(FRCNN_E_ARG, "stem_h3_fwd: 16-byte aligned tensors");
    static std::atomic<uint64_t> lds_seen{0};
    if (int e = raise_lds_once(lds_seen, (const void*)k_stem_h3, SH_LDS, "stem_h3_fwd")) return e;
    const dim3 grid((wp + SH_PW - 1) / SH_PW, (hp + SH_PH - 1) / SH_PH, n);
    if (grid.y > 65535 || grid.z > 65535) return fail(FRCNN_E_ARG, "stem_h3_fwd: image too large");
    k_stem_h3<<<grid, SH_NT, SH_LDS, as_stream(stream)>>>(x, x_amax, reinterpret_cast<const char*>(w_packed), scale, shift, h, w, hp, wp, out, y_amax);
    return check_launch("stem_h3_fwd");
}

}  // extern "C"
