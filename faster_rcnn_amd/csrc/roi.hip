// RoI crop + bilinear resize (custom_layers.RoiResizeConv, custom_layers.py:35-56) for gfx950.
//
// The reference builds num_rois separate strided_slice + tf.image.resize_images nodes; here
// all RoIs go through ONE launch: a workgroup per output pixel (roi, py, px), lanes sweep the
// channel axis with 16-byte loads/stores (NHWC: the channel run of one feature-map pixel is
// contiguous), so the conv4 map (<= 10 MB, L2/MALL resident) is gathered at full line width
// and the [n][7][7][C] tile is written exactly once.  HBM bound: C*4 B read x4 taps (cache
// hits) + C*4 B written per output pixel.
//
// TF 1.3 resize_bilinear, align_corners=False, no half-pixel centres:
//   scale = in/out (f32); src = i*scale; lo = (int)src; hi = min(lo+1, in-1); t = src - lo
//   top = tl + (tr - tl)*tx; bot = bl + (br - bl)*tx; out = top + (bot - top)*ty
// Compiled with -ffp-contract=off so the three lerps round where TF's scalar code rounds.
#include "common.h"

namespace frcnn {

struct Taps { int y_lo, y_hi, x_lo, x_hi; float ty, tx; bool ok; };

__device__ __forceinline__ Taps roi_taps(const float4 roi, int py, int px, int pool, int rows, int cols) {
    Taps t;
    const int x1 = (int)roi.x, y1 = (int)roi.y, x2 = (int)roi.z, y2 = (int)roi.w;   // K.cast(.,'int32') truncates
    const int h = y2 - y1, w = x2 - x1;
    t.ok = h > 0 && w > 0 && x1 >= 0 && y1 >= 0 && x2 <= cols && y2 <= rows;
    const float sy = (float)h / (float)pool, sx = (float)w / (float)pool;
    const float fy = (float)py * sy, fx = (float)px * sx;
    const int ly = (int)fy, lx = (int)fx;
    t.ty = fy - (float)ly;
    t.tx = fx - (float)lx;
    t.y_lo = y1 + ly;
    t.y_hi = y1 + min(ly + 1, h - 1);
    t.x_lo = x1 + lx;
    t.x_hi = x1 + min(lx + 1, w - 1);
    return t;
}

// `fill` (optional, [C]): what an invalid RoI produces instead of zeros; `relu`: clamp the result at 0.  Both exist
// for the hoisted detector head (nets.ResNetHead): a 1x1 conv + folded BatchNorm commutes with this resampling
// (interpolation weights sum to 1), so the head applies res5a_branch2a / branch1 ONCE to the conv4 map and
// resamples their outputs; an all-zero crop would have produced the BatchNorm shift there.
// n_per_img > 0: `feat` holds one map per IMAGE, RoI r crops image r / n_per_img's (frcnn_roi_crop_resize_fwd_batch)
__global__ void __launch_bounds__(256) k_roi_fwd(const float4* feat, int rows, int cols, int C4,
                                                 const float4* rois, int pool, const float4* fill, int relu, int pos_major, float4* out, int n_per_img) {
    const int pix = blockIdx.x;                 // (roi, py, px)
    const int px = pix % pool, py = (pix / pool) % pool, r = pix / (pool * pool);
    if (n_per_img > 0) feat += (size_t)(r / n_per_img) * rows * cols * C4;
    const Taps t = roi_taps(rois[r], py, px, pool, rows, cols);
    // pos_major: out[py][px][roi][c] (frcnn_conv_desc.layout == 1) instead of out[roi][py][px][c]
    const size_t orow = pos_major ? (size_t)(py * pool + px) * (gridDim.x / (pool * pool)) + r : (size_t)pix;
    float4* o = out + orow * C4;
    if (!t.ok) {
        for (int c = threadIdx.x; c < C4; c += blockDim.x) {
            float4 v = fill ? fill[c] : make_float4(0, 0, 0, 0);
            if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
            o[c] = v;
        }
        return;
    }
    const float4* tl = feat + ((size_t)t.y_lo * cols + t.x_lo) * C4;
    const float4* tr = feat + ((size_t)t.y_lo * cols + t.x_hi) * C4;
    const float4* bl = feat + ((size_t)t.y_hi * cols + t.x_lo) * C4;
    const float4* br = feat + ((size_t)t.y_hi * cols + t.x_hi) * C4;
    for (int c = threadIdx.x; c < C4; c += blockDim.x) {
        const float4 a = tl[c], b = tr[c], d = bl[c], e = br[c];
        float4 v;
#define LERP2(f) { const float top = a.f + (b.f - a.f) * t.tx; const float bot = d.f + (e.f - d.f) * t.tx; v.f = top + (bot - top) * t.ty; }
        LERP2(x) LERP2(y) LERP2(z) LERP2(w)
#undef LERP2
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        o[c] = v;
    }
}

// The same resampling with the result written as the f16x3 engine's two fp16 planes (conv_h3.hip; frcnn_h3_planes): hi = f16(v * 2^e),
// lo = f16((v * 2^e - hi) * 2^11), e = *pexp -- derived BEFORE this launch from the map's magnitude record (a bilinear sample is a
// convex combination of map values and a rejected RoI yields the fill vector: max(|map|, |fill|) bounds every output,
// frcnn_amax_merge).  The consumer, res5a_branch2b's 3x3 over the crops (resnet.py:508-512), then stages the planes unchanged.
__global__ void __launch_bounds__(256) k_roi_fwd_planes(const float4* feat, int rows, int cols, int C4, const float4* rois, int pool,
                                                        const float4* fill, int relu, int pos_major, const int* pexp, unsigned* status,
                                                        _Float16* planes, size_t plane_elems, int n_per_img) {
    __builtin_amdgcn_s_setreg(1 | (23 << 6), 1u);         // MODE.FP16_OVFL: a value above the bound clamps to +-65504 (conv_f32_common.h, the engine's fences)
    unsigned seen = 0u;                                   // largest |hi| bit pattern written
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int pix = blockIdx.x;
    const int px = pix % pool, py = (pix / pool) % pool, r = pix / (pool * pool);
    if (n_per_img > 0) feat += (size_t)(r / n_per_img) * rows * cols * C4;
    const Taps t = roi_taps(rois[r], py, px, pool, rows, cols);
    const size_t orow = pos_major ? (size_t)(py * pool + px) * (gridDim.x / (pool * pool)) + r : (size_t)pix;
    const int e = *pexp;
    unsigned sb = (unsigned)(e + 127) << 23;
    float s;
    __builtin_memcpy(&s, &sb, 4);
    f16x4* hi = reinterpret_cast<f16x4*>(planes + orow * C4 * 4);
    f16x4* lo = reinterpret_cast<f16x4*>(planes + plane_elems + orow * C4 * 4);
    const float4* tl = feat + ((size_t)t.y_lo * cols + t.x_lo) * C4;
    const float4* tr = feat + ((size_t)t.y_lo * cols + t.x_hi) * C4;
    const float4* bl = feat + ((size_t)t.y_hi * cols + t.x_lo) * C4;
    const float4* br = feat + ((size_t)t.y_hi * cols + t.x_hi) * C4;
    for (int c = threadIdx.x; c < C4; c += blockDim.x) {
        float4 v;
        if (!t.ok) {
            v = fill ? fill[c] : make_float4(0, 0, 0, 0);
        } else {
            const float4 a = tl[c], b = tr[c], d = bl[c], g = br[c];
#define LERP2(f) { const float top = a.f + (b.f - a.f) * t.tx; const float bot = d.f + (g.f - d.f) * t.tx; v.f = top + (bot - top) * t.ty; }
            LERP2(x) LERP2(y) LERP2(z) LERP2(w)
#undef LERP2
        }
        if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        const float xs[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
        f16x4 h, l;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const _Float16 a1 = (_Float16)xs[q];
            h[q] = a1; l[q] = (_Float16)((xs[q] - (float)a1) * 2048.0f);
            unsigned short hb;
            __builtin_memcpy(&hb, &a1, 2);
            seen = seen > (unsigned)(hb & 0x7fffu) ? seen : (unsigned)(hb & 0x7fffu);
        }
        hi[c] = h; lo[c] = l;
    }
    if (status) {                                         // a sample at / above 2^15 under the planes' scale: 1 = above the bound, 2 = clamped, 4 = not finite
#pragma unroll
        for (int o = 32; o; o >>= 1) { const unsigned t = __shfl_xor(seen, o); seen = seen > t ? seen : t; }
        const unsigned bits = seen > 0x7bffu ? 7u : seen == 0x7bffu ? 3u : seen >= 0x7800u ? 1u : 0u;
        if ((threadIdx.x & 63) == 0 && bits) atomicOr(status, bits);
    }
}

// Gradient w.r.t. the feature map as a GATHER: one workgroup per feature cell lists the RoIs whose box contains the
// cell (ascending), walks those RoIs' pool x pool samples and lists the corner taps that land on the cell -- sample-major,
// then top-left, top-right, bottom-left, bottom-right: the order in which TF's CPU ResizeBilinearGrad walks them -- and
// every lane sums its channels over that list.  Fixed summation order (the f32-atomic scatter this replaces made a
// detector training step depend on the order in which workgroups happened to run), no zero-fill of dfeat, each cell
// written once.  Per tap ((wy * g) * wx), TF's order of operations.
__device__ __forceinline__ int block_excl_scan(int cnt, int lane, int wave, int* wave_tot, int& total) {
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    __syncthreads();                                         // the previous use of wave_tot is over
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int off = incl - cnt;
    total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) off += wave_tot[w]; total += wave_tot[w]; }
    return off;
}

template <typename T>
__global__ void __launch_bounds__(256) k_roi_bwd_gather(const T* dout, int rows, int cols, int C,
                                                        const float4* rois, int n, int pool, float* dfeat) {
    __shared__ int roi_list[256];
    __shared__ int e_s[1024];
    __shared__ float e_wy[1024], e_wx[1024];
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cell = blockIdx.x, cy = cell / cols, cx = cell % cols;
    const int pp = pool * pool;
    for (int cb = 0; cb < C; cb += 1024) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int r0 = 0; r0 < n; r0 += 256) {
            // the RoIs of this chunk that can touch the cell: every tap of a valid RoI lies inside its box
            bool covers = false;
            if (r0 + tid < n) {
                const float4 roi = rois[r0 + tid];
                const int x1 = (int)roi.x, y1 = (int)roi.y, x2 = (int)roi.z, y2 = (int)roi.w;
                covers = y2 - y1 > 0 && x2 - x1 > 0 && x1 >= 0 && y1 >= 0 && x2 <= cols && y2 <= rows &&
                         cy >= y1 && cy < y2 && cx >= x1 && cx < x2;
            }
            int m;
            const int slot = block_excl_scan(covers ? 1 : 0, lane, wave, wave_tot, m);
            if (covers) roi_list[slot] = r0 + tid;
            __syncthreads();
            for (int base = 0; base < m * pp; base += 256) {
                const int q = base + tid;
                int cnt = 0, smp = 0;
                float wy[4], wx[4];
                if (q < m * pp) {
                    const int r = roi_list[q / pp], rem = q % pp, py = rem / pool, px = rem % pool;
                    smp = r * pp + rem;
                    const Taps t = roi_taps(rois[r], py, px, pool, rows, cols);
                    if (t.y_lo == cy && t.x_lo == cx) { wy[cnt] = 1.0f - t.ty; wx[cnt] = 1.0f - t.tx; ++cnt; }
                    if (t.y_lo == cy && t.x_hi == cx) { wy[cnt] = 1.0f - t.ty; wx[cnt] = t.tx; ++cnt; }
                    if (t.y_hi == cy && t.x_lo == cx) { wy[cnt] = t.ty; wx[cnt] = 1.0f - t.tx; ++cnt; }
                    if (t.y_hi == cy && t.x_hi == cx) { wy[cnt] = t.ty; wx[cnt] = t.tx; ++cnt; }
                }
                int total;
                const int off = block_excl_scan(cnt, lane, wave, wave_tot, total);     // (its barriers also fence e_* reuse)
                for (int k = 0; k < cnt; ++k) { e_s[off + k] = smp; e_wy[off + k] = wy[k]; e_wx[off + k] = wx[k]; }
                __syncthreads();
                // eight taps' loads in flight at a time (a 1x1 RoI puts all its 196 taps on one cell: one at a time that
                // cell alone took > 100 us); the additions stay in list order
                for (int e0 = 0; e0 < total; e0 += 8) {
                    T gv[8][4];                              // unconditional loads (clamped index): no branch, no wait between them
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const T* g = dout + (size_t)e_s[min(e0 + u, total - 1)] * C;
#pragma unroll
                        for (int j = 0; j < 4; ++j) gv[u][j] = g[min(cb + tid + 256 * j, C - 1)];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (e0 + u < total) {
                            const float a = e_wy[e0 + u], b = e_wx[e0 + u];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] += (a * (float)gv[u][j]) * b;
                        }
                    }
                }
            }
            __syncthreads();                                 // roi_list is rewritten by the next chunk
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cb + tid + 256 * j;
            if (c < C) dfeat[(size_t)cell * C + c] = acc[j];
        }
    }
}

}  // namespace frcnn

using namespace frcnn;

extern "C" {

int frcnn_roi_crop_resize_fwd(const float* feat, int rows, int cols, int C, const float* rois, int n, int pool,
                              float* out, void* stream) {
    return frcnn_roi_crop_resize_fwd_ex(feat, rows, cols, C, rois, n, pool, nullptr, 0, 0, out, stream);
}

int frcnn_roi_crop_resize_fwd_ex(const float* feat, int rows, int cols, int C, const float* rois, int n, int pool,
                                 const float* fill, int relu, int layout, float* out, void* stream) {
    if (n < 0 || rows <= 0 || cols <= 0 || C <= 0 || (C & 3) || pool <= 0) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd: bad shape (C must be a multiple of 4)");
    if (n == 0) return FRCNN_OK;
    if (!feat || !rois || !out) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd: null pointer");
    k_roi_fwd<<<n * pool * pool, 256, 0, as_stream(stream)>>>((const float4*)feat, rows, cols, C / 4, (const float4*)rois, pool,
                                                              (const float4*)fill, relu, layout, (float4*)out, 0);
    return check_launch("roi_crop_resize_fwd");
}

int frcnn_roi_crop_resize_fwd_planes(const float* feat, int rows, int cols, int C, const float* rois, int n, int pool,
                                     const float* fill, int relu, int layout, const frcnn_h3_planes* out, void* stream) {
    if (n < 0 || rows <= 0 || cols <= 0 || C <= 0 || (C & 3) || pool <= 0) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_planes: bad shape (C %% 4 == 0)");
    if (n == 0) return FRCNN_OK;
    if (!feat || !rois || !out || !out->planes || !out->exponent) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_planes: null pointer");
    if (reinterpret_cast<uintptr_t>(out->planes) & 15) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_planes: 16-byte aligned planes required");
    k_roi_fwd_planes<<<n * pool * pool, 256, 0, as_stream(stream)>>>((const float4*)feat, rows, cols, C / 4, (const float4*)rois, pool, (const float4*)fill,
                                                                    relu, layout, out->exponent, (unsigned*)out->status, (_Float16*)out->planes, (size_t)n * pool * pool * C, 0);
    return check_launch("roi_crop_resize_fwd_planes");
}

int frcnn_roi_crop_resize_fwd_batch(const float* feat, int rows, int cols, int C, const float* rois, int n, int n_per_img, int pool,
                                    const float* fill, int relu, int layout, float* out, const frcnn_h3_planes* planes_out, void* stream) {
    if (n < 0 || n_per_img <= 0 || rows <= 0 || cols <= 0 || C <= 0 || (C & 3) || pool <= 0) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_batch: bad shape (C %% 4 == 0, n_per_img > 0)");
    if ((out != nullptr) == (planes_out != nullptr)) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_batch: exactly one of out / planes_out");
    if (n == 0) return FRCNN_OK;
    if (!feat || !rois) return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_batch: null pointer");
    if (out) {
        k_roi_fwd<<<n * pool * pool, 256, 0, as_stream(stream)>>>((const float4*)feat, rows, cols, C / 4, (const float4*)rois, pool,
                                                                  (const float4*)fill, relu, layout, (float4*)out, n_per_img);
        return check_launch("roi_crop_resize_fwd_batch");
    }
    if (!planes_out->planes || !planes_out->exponent || (reinterpret_cast<uintptr_t>(planes_out->planes) & 15))
        return fail(FRCNN_E_ARG, "roi_crop_resize_fwd_batch: planes and exponent required, planes 16-byte aligned");
    k_roi_fwd_planes<<<n * pool * pool, 256, 0, as_stream(stream)>>>((const float4*)feat, rows, cols, C / 4, (const float4*)rois, pool, (const float4*)fill,
                                                                    relu, layout, planes_out->exponent, (unsigned*)planes_out->status, (_Float16*)planes_out->planes, (size_t)n * pool * pool * C, n_per_img);
    return check_launch("roi_crop_resize_fwd_batch (planes)");
}

int frcnn_roi_crop_resize_bwd(const float* dout, int rows, int cols, int C, const float* rois, int n, int pool,
                              float* dfeat, void* stream) {
    if (n < 0 || rows <= 0 || cols <= 0 || C <= 0 || pool <= 0) return fail(FRCNN_E_ARG, "roi_crop_resize_bwd: bad shape");
    if (n == 0) {
        if (!dfeat) return fail(FRCNN_E_ARG, "roi_crop_resize_bwd: null pointer");
        return hipMemsetAsync(dfeat, 0, (size_t)rows * cols * C * sizeof(float), as_stream(stream)) == hipSuccess ? FRCNN_OK : fail(FRCNN_E_HIP, "roi_crop_resize_bwd: memset failed");
    }
    if (!dout || !rois || !dfeat) return fail(FRCNN_E_ARG, "roi_crop_resize_bwd: null pointer");
    k_roi_bwd_gather<float><<<rows * cols, 256, 0, as_stream(stream)>>>(dout, rows, cols, C, (const float4*)rois, n, pool, dfeat);
    return check_launch("roi_crop_resize_bwd");
}

int frcnn_roi_crop_resize_bwd_bf16(const void* dout_bf16, int rows, int cols, int C, const float* rois, int n, int pool,
                                   float* dfeat, void* stream) {
    if (n < 0 || rows <= 0 || cols <= 0 || C <= 0 || pool <= 0) return fail(FRCNN_E_ARG, "roi_crop_resize_bwd_bf16: bad shape");
    if (n == 0) {
        if (!dfeat) return fail(FRCNN_E_ARG, "roi_crop_resize_bwd_bf16: null pointer");
        return hipMemsetAsync(dfeat, 0, (size_t)rows * cols * C * sizeof(float), as_stream(stream)) == hipSuccess ? FRCNN_OK : fail(FRCNN_E_HIP, "roi_crop_resize_bwd_bf16: memset failed");
    }
    if (!dout_bf16 || !rois || !dfeat) return fail(FRCNN_E_ARG, "roi_crop_resize_bwd_bf16: null pointer");
    k_roi_bwd_gather<__bf16><<<rows * cols, 256, 0, as_stream(stream)>>>((const __bf16*)dout_bf16, rows, cols, C, (const float4*)rois, n, pool, dfeat);
    return check_launch("roi_crop_resize_bwd_bf16");
}

}  // extern "C"
