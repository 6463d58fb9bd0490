// Anchor generation, IoU, RPN target assignment, proposal decode, detector targets.
// gfx950 (CDNA4) only.  These kernels are HBM/latency bound integer + IEEE-float work:
// one thread per anchor / RoI, coalesced 16-byte stores, wave64 reductions for the
// per-GT arg-max.  This translation unit is compiled with -ffp-contract=off: every
// float expression below must round exactly where the reference's numpy code rounds.
#include "common.h"
#include <math.h>

namespace frcnn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------
// anchors.  image space: centre = int(stride*(x+.5)) (rpn_util.py:184-189), conv space:
// centre = cell index (det_util.py:165).  x1 = c - w//2, x2 = x1 + w (rpn_util.py:293-296).
__device__ __forceinline__ float4 anchor_box(int cx, int cy, int h, int w) {
    const int x1 = cx - (w >> 1), y1 = cy - (h >> 1);
    return make_float4((float)x1, (float)y1, (float)(x1 + w), (float)(y1 + h));
}

__global__ void k_anchors(int rows, int cols, AnchorTable t, int A, int stride, int image_space, float4* out) {
    const int n = rows * cols * A;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int a = i % A, cell = i / A, x = cell % cols, y = cell / cols;
        // int32(stride*(x+0.5)): exact in double, truncation toward zero (values are >= 0)
        const int cx = image_space ? (int)((double)stride * ((double)x + 0.5)) : x;
        const int cy = image_space ? (int)((double)stride * ((double)y + 0.5)) : y;
        out[i] = anchor_box(cx, cy, t.h[a], t.w[a]);
    }
}

// ------------------------------------------------------------------------------------
// IoU, no +1 convention (util.py:155-175).  All f32; the order of operations is the
// reference's: area1, area2, max/min, max(0,.), w*h, (area1 + area2) - inter, inter/union.
__device__ __forceinline__ float iou_f32(float4 b, float area_b, float4 g, float area_g) {
    const float ix1 = fmaxf(b.x, g.x), iy1 = fmaxf(b.y, g.y);
    const float ix2 = fminf(b.z, g.z), iy2 = fminf(b.w, g.w);
    const float iw = fmaxf(0.0f, ix2 - ix1), ih = fmaxf(0.0f, iy2 - iy1);
    const float inter = iw * ih;
    const float uni = area_b + area_g - inter;
    return inter / uni;      // IEEE correctly rounded division (no fast-math)
}

__device__ __forceinline__ float box_area(float4 b) { return (b.z - b.x) * (b.w - b.y); }

template <typename BoxT>
__device__ __forceinline__ void load_box(const BoxT* p, int i, float4& b, float& area);

template <>
__device__ __forceinline__ void load_box<float>(const float* p, int i, float4& b, float& area) {
    b = reinterpret_cast<const float4*>(p)[i];
    area = box_area(b);
}
// int16 boxes: numpy forms the area in int16 (util.py:152) and promotes to f32 when it
// meets the f32 GT; coordinates promote to f32 inside maximum/minimum.
template <>
__device__ __forceinline__ void load_box<int16_t>(const int16_t* p, int i, float4& b, float& area) {
    const short4 s = reinterpret_cast<const short4*>(p)[i];
    b = make_float4((float)s.x, (float)s.y, (float)s.z, (float)s.w);
    area = (float)(int16_t)((int16_t)(s.z - s.x) * (int16_t)(s.w - s.y));
}

template <typename BoxT>
__global__ void k_cross_ious(const BoxT* boxes1, int M, const float4* boxes2, int G, float* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M; i += gridDim.x * blockDim.x) {
        float4 b; float ab;
        load_box<BoxT>(boxes1, i, b, ab);
        for (int g = 0; g < G; ++g) {
            const float4 gb = boxes2[g];
            out[(size_t)i * G + g] = iou_f32(b, ab, gb, box_area(gb));
        }
    }
}

// ------------------------------------------------------------------------------------
// Regression targets (util.py:180-206) in f64 as numpy evaluates them for integer anchors.
// GT_F32: rpn_util.py:91 passes f32 GT -> centre sum and width are formed in f32.
__device__ __forceinline__ void reg_params(int ax1, int ay1, int ax2, int ay2,
                                           double gcx, double gcy, double gw, double gh, double t[4]) {
    const double acx = (double)(ax2 + ax1) / 2.0, acy = (double)(ay2 + ay1) / 2.0;
    const double aw = (double)(ax2 - ax1), ah = (double)(ay2 - ay1);
    t[0] = (gcx - acx) / aw;
    t[1] = (gcy - acy) / ah;
    t[2] = log(gw / aw);
    t[3] = log(gh / ah);
}

// pass 1: per anchor max/argmax over GT; per GT max/argmax over anchors via packed atomicMax.
// key = iou_bits << 32 | ~idx : larger IoU wins, ties -> smaller anchor index (np.argmax).
__global__ void k_rpn_pass1(int rows, int cols, AnchorTable t, int A, int stride,
                            const float4* gt, int G, float* max_iou, int32_t* argmax_gt,
                            unsigned long long* best_by_gt) {
    const int n = rows * cols * A;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    float4 b = make_float4(0, 0, 1, 1);
    if (live) {
        const int a = i % A, cell = i / A, x = cell % cols, y = cell / cols;
        b = anchor_box((int)((double)stride * ((double)x + 0.5)), (int)((double)stride * ((double)y + 0.5)), t.h[a], t.w[a]);
    }
    const float ab = box_area(b);
    float best = 0.0f; int barg = 0;
    for (int g = 0; g < G; ++g) {
        const float4 gb = gt[g];
        const float v = live ? iou_f32(b, ab, gb, box_area(gb)) : -1.0f;
        if (g == 0 || v > best) { best = v; barg = g; }     // first maximum wins (np.argmax)
        // wave-level arg-max, then one atomic per wave
        unsigned long long key = live ? (((unsigned long long)__float_as_uint(v)) << 32) | (unsigned)(~(unsigned)i) : 0ull;
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key, off);
            key = o > key ? o : key;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&best_by_gt[g], key);
    }
    if (live) { max_iou[i] = best; if (argmax_gt) argmax_gt[i] = barg; }
}

__global__ void k_rpn_pass2(int rows, int cols, AnchorTable t, int A, int stride,
                            const float4* gt, int G, int img_w, int img_h,
                            const float* max_iou, const int32_t* argmax_gt_ws,
                            const unsigned long long* best_by_gt,
                            uint8_t* can_use, uint8_t* is_pos, float4* bbreg) {
    const int n = rows * cols * A;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int a = i % A, cell = i / A, x = cell % cols, y = cell / cols;
    const int cx = (int)((double)stride * ((double)x + 0.5)), cy = (int)((double)stride * ((double)y + 0.5));
    const int w = t.w[a], h = t.h[a];
    const int x1 = cx - (w >> 1), y1 = cy - (h >> 1), x2 = x1 + w, y2 = y1 + h;
    const float m = G > 0 ? max_iou[i] : 0.0f;
    bool pos = G > 0 && m > 0.7f;                       // f32 compare (rpn_util.py:75)
    for (int g = 0; g < G; ++g) {                       // per-GT best anchor with IoU > 0 (:77-78)
        const unsigned long long key = best_by_gt[g];
        const float v = __uint_as_float((unsigned)(key >> 32));
        if (v > 0.0f && (unsigned)(~(unsigned)key) == (unsigned)i) pos = true;
    }
    float4 tgt = make_float4(0, 0, 0, 0);
    if (pos) {
        const float4 gb = gt[argmax_gt_ws[i]];
        // f32 sums / differences first (np.float32 scalars), halving exact, then f64
        const double gcx = (double)((gb.z + gb.x) / 2.0f), gcy = (double)((gb.w + gb.y) / 2.0f);
        const double gw = (double)(gb.z - gb.x), gh = (double)(gb.w - gb.y);
        double r[4];
        reg_params(x1, y1, x2, y2, gcx, gcy, gw, gh, r);
        // BBREG_MULTIPLIERS (f32) * f64 tuple -> f64 product, stored to f32 (rpn_util.py:93)
        tgt = make_float4((float)(10.0 * r[0]), (float)(10.0 * r[1]), (float)(5.0 * r[2]), (float)(5.0 * r[3]));
    }
    const bool neg = !pos && (G == 0 || m < 0.3f);        // :95
    const bool oob = x1 < 0 || y1 < 0 || x2 >= img_w || y2 >= img_h;   // :302-310
    can_use[i] = (pos || neg) && !oob;                  // can_use[oob]=0 but is_pos stays (:97)
    is_pos[i] = pos;
    bbreg[i] = tgt;
}

// ------------------------------------------------------------------------------------
// proposal decode: util.transform_np_inplace (util.py:111-142) in f32 with np.round
// (half-to-even), then det_util._sanitize_boxes_inplace (:179-192) and validity (:196-205).
__device__ __forceinline__ float exp_f32(float x) {
    // numpy's f32 exp is a <=2.5-ulp SIMD routine; no device libm reproduces it bit for
    // bit.  We return the correctly rounded f32 value (via f64), which differs from any
    // faithful expf by at most a couple of ulps: see DESIGN.md "decode rounding boundary".
    return (float)exp((double)x);
}

__device__ __forceinline__ float4 transform_f32(float4 c, float4 d) {
    float w = c.z - c.x, h = c.w - c.y;
    float cx = c.x + w / 2.0f, cy = c.y + h / 2.0f;
    cx = cx + d.x * w;
    cy = cy + d.y * h;
    w = w * exp_f32(d.z);
    h = h * exp_f32(d.w);
    float x1 = cx - w / 2.0f, y1 = cy - h / 2.0f;
    x1 = rintf(x1); y1 = rintf(y1); w = rintf(w); h = rintf(h);
    return make_float4(x1, y1, x1 + w, y1 + h);
}

__global__ void k_decode(const float4* regr, int rows, int cols, AnchorTable t, int A,
                         float4* rois, uint8_t* valid) {
    const int n = rows * cols * A;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int a = i % A, cell = i / A, x = cell % cols, y = cell / cols;
        const float4 anc = anchor_box(x, y, t.h[a], t.w[a]);
        float4 d = regr[i];
        d.x = d.x / 10.0f; d.y = d.y / 10.0f; d.z = d.z / 5.0f; d.w = d.w / 5.0f;   // det_util.py:376
        float4 r = transform_f32(anc, d);
        r.z = fmaxf(r.x + 1.0f, r.z);
        r.w = fmaxf(r.y + 1.0f, r.w);
        r.x = fmaxf(0.0f, r.x);
        r.y = fmaxf(0.0f, r.y);
        r.z = fminf((float)(cols - 1), r.z);
        r.w = fminf((float)(rows - 1), r.w);
        rois[i] = r;
        valid[i] = (r.z > r.x) && (r.w > r.y);
    }
}

__global__ void k_transform_inplace(float4* coords, const float4* deltas, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        coords[i] = transform_f32(coords[i], deltas[i]);
}

// ------------------------------------------------------------------------------------
// detector targets (det_util.py:310-366), one thread per RoI.
__global__ void k_roi_targets(const int16_t* rois, int E, const float4* gt32, const double* gt64,
                              const int32_t* gt_cls, int G, int bg_idx,
                              uint8_t* eligible, int32_t* cls, float4* targets) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= E) return;
    float4 b; float ab;
    load_box<int16_t>(rois, i, b, ab);
    float best = 0.0f; int barg = 0;
    for (int g = 0; g < G; ++g) {
        const float4 gb = gt32[g];
        const float v = iou_f32(b, ab, gb, box_area(gb));
        if (g == 0 || v > best) { best = v; barg = g; }
    }
    const bool elig = G > 0 && best >= 0.1f;            // f32 compares (det_util.py:317,320)
    const bool pos = G > 0 && best >= 0.5f;
    float4 tgt = make_float4(0, 0, 0, 0);
    if (pos) {
        const double* g = gt64 + 4 * barg;                // gt_box.corners are f64 (det_util.py:349)
        double r[4];
        reg_params((int)b.x, (int)b.y, (int)b.z, (int)b.w, (g[2] + g[0]) / 2.0, (g[3] + g[1]) / 2.0, g[2] - g[0], g[3] - g[1], r);
        // stored to f32 first (:350), THEN multiplied by the f32 multipliers (:351)
        tgt = make_float4((float)r[0] * 10.0f, (float)r[1] * 10.0f, (float)r[2] * 5.0f, (float)r[3] * 5.0f);
    }
    eligible[i] = elig;
    cls[i] = pos ? gt_cls[barg] : bg_idx;
    targets[i] = tgt;
}

static inline int grid_for(int n, int block = 256, int cap = 2048) {
    int g = (n + block - 1) / block;
    return g < 1 ? 1 : (g > cap ? cap : g);
}

// resnet.preprocess / vgg.preprocess (resnet.py:64-75, vgg.py:52-57) on the device: the host path computes
// float64(u8) - mean and the network input casts to f32; the same two roundings here, so the tensor is bit-identical
// while the upload shrinks from 12 to 3 bytes per pixel.
__global__ void k_preprocess_u8(const uint8_t* img, size_t n, double m0, double m1, double m2, float* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        out[i] = (float)((double)img[i] - (c == 0 ? m0 : (c == 1 ? m1 : m2)));
    }
}

// ---- padded canvases (round 6): images of DIFFERENT true sizes share one captured pass.  Each image sits near the top-left corner of
// a canvas [hc][wc] with EVEN sides, at offset (oy, ox) = (h & 1, w & 1): TF's SAME padding at stride 2 (resnet.py:408) puts
// pad_before = total / 2 zeros in front, total = 5 for an even size and 6 for an odd one under conv1's 7x7 window, i.e. one more zero row /
// column in front of an odd side -- the offset supplies it, so conv1's output cell (i, j) on the canvas is the true image's cell (i, j)
// whatever the parities (it is the only layer of the networks whose padding depends on the size).  Everything outside the image is zero,
// which is what the reference's padding puts there.  A convolution with taps (3x3, 7x7) reads zeros beyond the true border exactly
// where the reference reads its padding as long as its INPUT is zero there: the canvas is, and k_zero_outside restores that behind
// every layer whose output feeds such a convolution (bias / BatchNorm shift / ReLU make the outside non-zero again).  Pointwise
// layers and VALID pooling never look outside.
__global__ void k_preprocess_u8_canvas(const uint8_t* img, int h, int w, int hc, int wc, int oy, int ox, double m0, double m1, double m2, float* out) {
    const size_t n = (size_t)hc * wc * 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const size_t px = i / 3;
        const int x = (int)(px % wc) - ox, y = (int)(px / wc) - oy;
        out[i] = (x >= 0 && y >= 0 && x < w && y < h) ? (float)((double)img[((size_t)y * w + x) * 3 + c] - (c == 0 ? m0 : (c == 1 ? m1 : m2))) : 0.0f;
    }
}

// x [n][hc][wc][c16 pieces of 16 bytes]: zero the cells at or beyond image i's true extent hw[i] = {rows, cols} (device words).  One
// workgroup per canvas ROW: the pieces behind the true columns (the whole row below the true rows) are contiguous in memory.  (A
// workgroup per cell was 148 000 workgroups at stage 2 of a four-image pass, nearly all of which returned at once.)
__global__ void __launch_bounds__(256) k_zero_outside(int4* x, int hc, int wc, int c16, const int* hw) {
    const int y = blockIdx.x, img = blockIdx.y;
    const int x0 = y < hw[2 * img] ? min(hw[2 * img + 1], wc) : 0;
    int4* p = x + (((size_t)img * hc + y) * wc + x0) * c16;
    const int n = (wc - x0) * c16;
    for (int k = threadIdx.x; k < n; k += blockDim.x) p[k] = make_int4(0, 0, 0, 0);
}

// k_decode on a canvas: the RPN outputs are [rows_c][cols_c][4A]; image's true map is rc[0] x rc[1] (device words).  A cell outside it
// is no candidate (valid = 0); boxes are clipped to the TRUE extent (det_util.py:179-192 with the image's own rows / cols).  The linear
// order of the cells inside the true map is the true map's row-major order, so the top-K's tie rule (ascending index) picks as it would.
__global__ void k_decode_canvas(const float4* regr, int rows_c, int cols_c, AnchorTable t, int A, const int* rc, float4* rois, uint8_t* valid) {
    const int n = rows_c * cols_c * A, rows = rc[0], cols = rc[1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int a = i % A, cell = i / A, x = cell % cols_c, y = cell / cols_c;
        if (x >= cols || y >= rows) { rois[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); valid[i] = 0; continue; }
        const float4 anc = anchor_box(x, y, t.h[a], t.w[a]);
        float4 d = regr[i];
        d.x = d.x / 10.0f; d.y = d.y / 10.0f; d.z = d.z / 5.0f; d.w = d.w / 5.0f;   // det_util.py:376
        float4 r = transform_f32(anc, d);
        r.z = fmaxf(r.x + 1.0f, r.z);
        r.w = fmaxf(r.y + 1.0f, r.w);
        r.x = fmaxf(0.0f, r.x);
        r.y = fmaxf(0.0f, r.y);
        r.z = fminf((float)(cols - 1), r.z);
        r.w = fminf((float)(rows - 1), r.w);
        rois[i] = r;
        valid[i] = (r.z > r.x) && (r.w > r.y);
    }
}

// shapes.Image.data (shapes.py:19-29): cv2.resize(img, (w, h), interpolation=cv2.INTER_CUBIC) on the decoded uint8 frame --
// OpenCV's 8-bit path: per axis four taps (BORDER_REPLICATE), cubic coefficients (A = -0.75) in 11-bit fixed point, horizontal
// then vertical pass, (v + 2^21) >> 22, saturate.  No intermediate rounding between the passes, so one thread sums the 16
// products of a destination pixel directly: the same integer the two-pass form produces.  tab_[xy]: [dst][8] = 4 source
// indices (clamped), 4 coefficients (frcnn_resize_cubic_taps).  flip: Image.horizontal_flip (shapes.py:27: data[:, ::-1]).
__global__ void k_resize_cubic_u8(const uint8_t* src, int sw, const int32_t* tab_x, const int32_t* tab_y, int dh, int dw, int flip, uint8_t* dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const int32_t* tx = tab_x + 8 * x;
    const int32_t* ty = tab_y + 8 * y;
    long long acc0 = 0, acc1 = 0, acc2 = 0;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint8_t* row = src + (size_t)ty[ky] * sw * 3;
        int h0 = 0, h1 = 0, h2 = 0;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const uint8_t* px = row + (size_t)tx[kx] * 3;
            const int c = tx[4 + kx];
            h0 += c * px[0]; h1 += c * px[1]; h2 += c * px[2];
        }
        const long long cy = ty[4 + ky];
        acc0 += cy * h0; acc1 += cy * h1; acc2 += cy * h2;
    }
    auto fin = [](long long v) { v = (v + (1ll << 21)) >> 22; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    // flip bit 0: horizontal flip (shapes.py:27); bit 1 (round 6): the source is RGB as the JPEG decoder delivers it -- write B, G, R
    // (cv2.imread's order, shapes.py:23), so that the host does not spend ~0.8 ms per frame reversing the channels before the upload
    uint8_t* o = dst + ((size_t)y * dw + ((flip & 1) ? dw - 1 - x : x)) * 3;
    if (flip & 2) { o[0] = fin(acc2); o[1] = fin(acc1); o[2] = fin(acc0); }
    else { o[0] = fin(acc0); o[1] = fin(acc1); o[2] = fin(acc2); }
}

}  // namespace frcnn

using namespace frcnn;

extern "C" {

// HOST function: the tap table of one axis, in f32 exactly as OpenCV's resize computes it (and shapes._cubic_taps restates it):
// f = (float)((d + 0.5) * (src / dst) - 0.5) with the product in double, s = floor(f), x = f - s, the four cubic weights in f32
// (this translation unit is compiled with -ffp-contract=off: every operation rounds once), x 2048, round half to even, int16.
int frcnn_resize_cubic_taps(int dst, int src, int32_t* tab_h) {
    if (dst <= 0 || src <= 0 || !tab_h) return fail(FRCNN_E_ARG, "resize_cubic_taps: bad argument");
    const double scale = (double)src / (double)dst;
    const float A = -0.75f, one = 1.0f;
    for (int d = 0; d < dst; ++d) {
        const float f = (float)(((double)d + 0.5) * scale - 0.5);
        const float fl = floorf(f);
        const int s = (int)fl;
        const float x = f - fl;
        const float xp = x + one, xm = one - x;
        float c[4];
        c[0] = ((A * xp - 5.0f * A) * xp + 8.0f * A) * xp - 4.0f * A;
        c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + one;
        c[2] = ((A + 2.0f) * xm - (A + 3.0f)) * xm * xm + one;
        c[3] = one - c[0] - c[1] - c[2];
        for (int k = 0; k < 4; ++k) {
            int idx = s - 1 + k;
            idx = idx < 0 ? 0 : (idx > src - 1 ? src - 1 : idx);
            float v = nearbyintf(c[k] * 2048.0f);                 // default rounding mode: half to even (cvRound)
            v = v < -32768.0f ? -32768.0f : (v > 32767.0f ? 32767.0f : v);
            tab_h[8 * d + k] = idx;
            tab_h[8 * d + 4 + k] = (int32_t)v;
        }
    }
    return FRCNN_OK;
}

int frcnn_resize_cubic_u8(const uint8_t* src_hwc, int src_h, int src_w, const int32_t* tab_x, const int32_t* tab_y,
                          int dst_h, int dst_w, int flip, uint8_t* dst_hwc, void* stream) {
    if (!src_hwc || !tab_x || !tab_y || !dst_hwc) return fail(FRCNN_E_ARG, "resize_cubic_u8: null pointer");
    if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || dst_h > 65535) return fail(FRCNN_E_ARG, "resize_cubic_u8: bad size");
    k_resize_cubic_u8<<<dim3((dst_w + 127) / 128, dst_h), 128, 0, as_stream(stream)>>>(src_hwc, src_w, tab_x, tab_y, dst_h, dst_w, flip & 3, dst_hwc);
    return check_launch("resize_cubic_u8");
}

const char* frcnn_last_error(void) { return g_err; }
int frcnn_version(void) { return FRCNN_ABI_VERSION; }
int frcnn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int frcnn_anchors_image(int rows, int cols, const int32_t* anchor_hw_h, int A, int stride, float* out, void* stream) {
    AnchorTable t;
    if (int e = load_anchor_table(anchor_hw_h, A, &t)) return e;
    if (rows <= 0 || cols <= 0 || stride <= 0 || !out) return fail(FRCNN_E_ARG, "anchors_image: bad argument");
    k_anchors<<<grid_for(rows * cols * A), 256, 0, as_stream(stream)>>>(rows, cols, t, A, stride, 1, (float4*)out);
    return check_launch("anchors_image");
}

int frcnn_anchors_conv(int rows, int cols, const int32_t* anchor_hw_conv_h, int A, float* out, void* stream) {
    AnchorTable t;
    if (int e = load_anchor_table(anchor_hw_conv_h, A, &t)) return e;
    if (rows <= 0 || cols <= 0 || !out) return fail(FRCNN_E_ARG, "anchors_conv: bad argument");
    k_anchors<<<grid_for(rows * cols * A), 256, 0, as_stream(stream)>>>(rows, cols, t, A, 1, 0, (float4*)out);
    return check_launch("anchors_conv");
}

int frcnn_cross_ious_f32(const float* boxes1, int M, const float* boxes2, int G, float* out, void* stream) {
    if (M < 0 || G < 0) return fail(FRCNN_E_ARG, "cross_ious: negative size");
    if (M == 0 || G == 0) return FRCNN_OK;
    if (!boxes1 || !boxes2 || !out) return fail(FRCNN_E_ARG, "cross_ious: null pointer");
    k_cross_ious<float><<<grid_for(M), 256, 0, as_stream(stream)>>>(boxes1, M, (const float4*)boxes2, G, out);
    return check_launch("cross_ious_f32");
}

int frcnn_cross_ious_i16(const int16_t* boxes1, int M, const float* boxes2, int G, float* out, void* stream) {
    if (M < 0 || G < 0) return fail(FRCNN_E_ARG, "cross_ious: negative size");
    if (M == 0 || G == 0) return FRCNN_OK;
    if (!boxes1 || !boxes2 || !out) return fail(FRCNN_E_ARG, "cross_ious: null pointer");
    k_cross_ious<int16_t><<<grid_for(M), 256, 0, as_stream(stream)>>>(boxes1, M, (const float4*)boxes2, G, out);
    return check_launch("cross_ious_i16");
}

size_t frcnn_rpn_assign_workspace_bytes(int rows, int cols, int A, int G) {
    const size_t n = (size_t)rows * cols * A;
    return align_up((size_t)(G > 0 ? G : 1) * 8, 256) + align_up(n * 4, 256) + align_up(n * 4, 256);
}

int frcnn_rpn_assign(int rows, int cols, const int32_t* anchor_hw_h, int A, int stride,
                     const float* gt, int G, int img_w, int img_h,
                     uint8_t* can_use, uint8_t* is_pos, float* bbreg, int32_t* argmax_gt,
                     void* workspace, size_t workspace_bytes, void* stream) {
    AnchorTable t;
    if (int e = load_anchor_table(anchor_hw_h, A, &t)) return e;
    if (rows <= 0 || cols <= 0 || stride <= 0 || G < 0 || !can_use || !is_pos || !bbreg || (G > 0 && !gt))
        return fail(FRCNN_E_ARG, "rpn_assign: bad argument");
    if (!workspace || workspace_bytes < frcnn_rpn_assign_workspace_bytes(rows, cols, A, G))
        return fail(FRCNN_E_WORKSPACE, "rpn_assign: workspace needs %zu bytes", frcnn_rpn_assign_workspace_bytes(rows, cols, A, G));
    const int n = rows * cols * A;
    char* ws = (char*)workspace;
    unsigned long long* best = (unsigned long long*)ws;
    ws += align_up((size_t)(G > 0 ? G : 1) * 8, 256);
    float* max_iou = (float*)ws;
    ws += align_up((size_t)n * 4, 256);
    int32_t* arg_ws = (int32_t*)ws;
    hipStream_t s = as_stream(stream);
    if (hipMemsetAsync(best, 0, (size_t)(G > 0 ? G : 1) * 8, s) != hipSuccess) return fail(FRCNN_E_HIP, "rpn_assign: memset failed");
    const int blocks = (n + 255) / 256;
    if (G > 0) {
        k_rpn_pass1<<<blocks, 256, 0, s>>>(rows, cols, t, A, stride, (const float4*)gt, G, max_iou, arg_ws, best);
        if (int e = check_launch("rpn_assign pass1")) return e;
    }
    k_rpn_pass2<<<blocks, 256, 0, s>>>(rows, cols, t, A, stride, (const float4*)gt, G, img_w, img_h,
                                       max_iou, arg_ws, best, can_use, is_pos, (float4*)bbreg);
    if (int e = check_launch("rpn_assign pass2")) return e;
    if (argmax_gt) {
        if (G > 0) {
            if (hipMemcpyAsync(argmax_gt, arg_ws, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return fail(FRCNN_E_HIP, "rpn_assign: copy failed");
        } else if (hipMemsetAsync(argmax_gt, 0, (size_t)n * 4, s) != hipSuccess) {
            return fail(FRCNN_E_HIP, "rpn_assign: memset failed");
        }
    }
    return FRCNN_OK;
}

// ---- RPN batch sampling on the device side of the host RNG (rpn_util.py:324-350).  The reference's _apply_sampling needs, from the
// 21 546 - 64 296 anchors of an image, only HOW MANY usable positives / negatives there are: `random.sample(range(num_pos), ...)`
// draws POSITIONS in the ascending lists np.where returns.  frcnn_rpn_sample_lists builds those lists on the device and hands the
// two counts over (8 bytes instead of 1.2 MB of masks and targets); the host draws the positions to switch off (the global
// `random` stream, frcnn_host_mt_sample_range) and frcnn_rpn_pack_targets applies them and writes y_class / y_bbreg as
// rpn_y_true lays them out (rpn_util.py:126-140), as float32, straight into the training step's input tensors.
__global__ void __launch_bounds__(1024) k_rpn_sample_lists(const uint8_t* can_use, const uint8_t* is_pos, int n,
                                                           int32_t* pos_locs, int32_t* neg_locs, int32_t* counts) {
    __shared__ int s_pos[1024], s_neg[1024];
    const int tid = threadIdx.x, per = (n + 1023) / 1024, lo = min(tid * per, n), hi = min(lo + per, n);
    int np_ = 0, nn = 0;
    for (int i = lo; i < hi; ++i) {
        const bool u = can_use[i] == 1, p = is_pos[i] == 1;
        np_ += u && p;
        nn += u && is_pos[i] == 0;
    }
    s_pos[tid] = np_; s_neg[tid] = nn;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {             // inclusive scan (Hillis-Steele; one workgroup, once per image)
        const int a = tid >= off ? s_pos[tid - off] : 0, b = tid >= off ? s_neg[tid - off] : 0;
        __syncthreads();
        s_pos[tid] += a; s_neg[tid] += b;
        __syncthreads();
    }
    int wp = s_pos[tid] - np_, wn = s_neg[tid] - nn;       // exclusive prefix: where this thread's run starts in each list
    for (int i = lo; i < hi; ++i) {
        const bool u = can_use[i] == 1;
        if (u && is_pos[i] == 1) pos_locs[wp++] = i;
        else if (u && is_pos[i] == 0) neg_locs[wn++] = i;
    }
    if (tid == 1023) { counts[0] = s_pos[1023]; counts[1] = s_neg[1023]; }
}

__global__ void k_rpn_clear_sampled(uint8_t* can_use, const int32_t* locs, int n_locs, const int32_t* off, int n_off) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_off) {
        const int q = off[j];
        if (q >= 0 && q < n_locs) can_use[locs[q]] = 0;
    }
}

// one thread per (cell, anchor): y_class[cell] = [can_use x A | is_pos x A], y_bbreg[cell] = [(is_pos & can_use) repeated 4 x A | targets 4A]
__global__ void k_rpn_pack_targets(const uint8_t* can_use, const uint8_t* is_pos, const float4* bbreg, int cells, int A,
                                   float* y_class, float* y_bbreg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells * A) return;
    const int cell = i / A, a = i - cell * A;
    const bool u = can_use[i] != 0, p = is_pos[i] != 0;     // (np.concatenate of bool arrays: any non-zero byte is True)
    y_class[(size_t)cell * 2 * A + a] = u ? 1.0f : 0.0f;
    y_class[(size_t)cell * 2 * A + A + a] = p ? 1.0f : 0.0f;
    const float both = (u && p) ? 1.0f : 0.0f;
    float* yb = y_bbreg + (size_t)cell * 8 * A;
    const float4 t = bbreg[i];
    yb[4 * a] = both; yb[4 * a + 1] = both; yb[4 * a + 2] = both; yb[4 * a + 3] = both;
    yb[4 * A + 4 * a] = t.x; yb[4 * A + 4 * a + 1] = t.y; yb[4 * A + 4 * a + 2] = t.z; yb[4 * A + 4 * a + 3] = t.w;
}

int frcnn_rpn_sample_lists(const uint8_t* can_use, const uint8_t* is_pos, int n, int32_t* pos_locs, int32_t* neg_locs, int32_t* counts, void* stream) {
    if (!can_use || !is_pos || !pos_locs || !neg_locs || !counts || n <= 0) return fail(FRCNN_E_ARG, "rpn_sample_lists: bad argument");
    k_rpn_sample_lists<<<1, 1024, 0, as_stream(stream)>>>(can_use, is_pos, n, pos_locs, neg_locs, counts);
    return check_launch("rpn_sample_lists");
}

int frcnn_rpn_pack_targets(uint8_t* can_use, const uint8_t* is_pos, const float* bbreg, int cells, int A,
                           const int32_t* pos_locs, int n_pos, const int32_t* off_pos, int n_off_pos,
                           const int32_t* neg_locs, int n_neg, const int32_t* off_neg, int n_off_neg,
                           float* y_class, float* y_bbreg, void* stream) {
    if (!can_use || !is_pos || !bbreg || !y_class || !y_bbreg || cells <= 0 || A <= 0 || n_off_pos < 0 || n_off_neg < 0
        || (n_off_pos > 0 && (!pos_locs || !off_pos)) || (n_off_neg > 0 && (!neg_locs || !off_neg)))
        return fail(FRCNN_E_ARG, "rpn_pack_targets: bad argument");
    hipStream_t s = as_stream(stream);
    if (n_off_pos > 0) {
        k_rpn_clear_sampled<<<(n_off_pos + 255) / 256, 256, 0, s>>>(can_use, pos_locs, n_pos, off_pos, n_off_pos);
        if (int e = check_launch("rpn_pack_targets (positives)")) return e;
    }
    if (n_off_neg > 0) {
        k_rpn_clear_sampled<<<(n_off_neg + 255) / 256, 256, 0, s>>>(can_use, neg_locs, n_neg, off_neg, n_off_neg);
        if (int e = check_launch("rpn_pack_targets (negatives)")) return e;
    }
    k_rpn_pack_targets<<<(cells * A + 255) / 256, 256, 0, s>>>(can_use, is_pos, (const float4*)bbreg, cells, A, y_class, y_bbreg);
    return check_launch("rpn_pack_targets");
}

int frcnn_decode_proposals(const float* regr, int rows, int cols, const int32_t* anchor_hw_conv_h, int A,
                           float* rois, uint8_t* valid, void* stream) {
    AnchorTable t;
    if (int e = load_anchor_table(anchor_hw_conv_h, A, &t)) return e;
    if (rows <= 0 || cols <= 0 || !regr || !rois || !valid) return fail(FRCNN_E_ARG, "decode_proposals: bad argument");
    k_decode<<<grid_for(rows * cols * A), 256, 0, as_stream(stream)>>>((const float4*)regr, rows, cols, t, A, (float4*)rois, valid);
    return check_launch("decode_proposals");
}

int frcnn_decode_proposals_canvas(const float* regr, int rows_c, int cols_c, const int32_t* anchor_hw_conv_h, int A, const int32_t* true_rows_cols,
                                  float* rois, uint8_t* valid, void* stream) {
    AnchorTable t;
    if (int e = load_anchor_table(anchor_hw_conv_h, A, &t)) return e;
    if (rows_c <= 0 || cols_c <= 0 || !regr || !rois || !valid || !true_rows_cols) return fail(FRCNN_E_ARG, "decode_proposals_canvas: bad argument");
    k_decode_canvas<<<grid_for(rows_c * cols_c * A), 256, 0, as_stream(stream)>>>((const float4*)regr, rows_c, cols_c, t, A, true_rows_cols, (float4*)rois, valid);
    return check_launch("decode_proposals_canvas");
}

int frcnn_zero_outside(void* x, int n, int hc, int wc, int row_bytes, const int32_t* true_hw, void* stream) {
    if (!x || !true_hw || n <= 0 || hc <= 0 || wc <= 0 || row_bytes <= 0 || (row_bytes & 15) || (reinterpret_cast<uintptr_t>(x) & 15))
        return fail(FRCNN_E_ARG, "zero_outside: bad argument (a cell's channels must be a multiple of 16 bytes, 16-byte aligned)");
    k_zero_outside<<<dim3(hc, n), 256, 0, as_stream(stream)>>>((int4*)x, hc, wc, row_bytes / 16, true_hw);
    return check_launch("zero_outside");
}

int frcnn_preprocess_u8_canvas(const uint8_t* img_hwc, int h, int w, int hc, int wc, int oy, int ox, const double* mean3_h, float* out, void* stream) {
    if (!img_hwc || !mean3_h || !out || h <= 0 || w <= 0 || oy < 0 || ox < 0 || hc < h + oy || wc < w + ox)
        return fail(FRCNN_E_ARG, "preprocess_u8_canvas: bad argument (the image at its offset must fit the canvas)");
    size_t g = ((size_t)hc * wc * 3 + 255) / 256;
    if (g > 8192) g = 8192;
    k_preprocess_u8_canvas<<<(int)g, 256, 0, as_stream(stream)>>>(img_hwc, h, w, hc, wc, oy, ox, mean3_h[0], mean3_h[1], mean3_h[2], out);
    return check_launch("preprocess_u8_canvas");
}

int frcnn_transform_inplace(float* coords, const float* deltas, int n, void* stream) {
    if (n < 0) return fail(FRCNN_E_ARG, "transform_inplace: negative size");
    if (n == 0) return FRCNN_OK;
    if (!coords || !deltas) return fail(FRCNN_E_ARG, "transform_inplace: null pointer");
    k_transform_inplace<<<grid_for(n), 256, 0, as_stream(stream)>>>((float4*)coords, (const float4*)deltas, n);
    return check_launch("transform_inplace");
}

int frcnn_roi_targets(const int16_t* rois, int E, const float* gt_f32, const double* gt_f64,
                      const int32_t* gt_cls, int G, int bg_idx,
                      uint8_t* eligible, int32_t* cls, float* targets, void* stream) {
    if (E < 0 || G < 0) return fail(FRCNN_E_ARG, "roi_targets: negative size");
    if (E == 0) return FRCNN_OK;
    if (!rois || !eligible || !cls || !targets || (G > 0 && (!gt_f32 || !gt_f64 || !gt_cls)))
        return fail(FRCNN_E_ARG, "roi_targets: null pointer");
    k_roi_targets<<<(E + 255) / 256, 256, 0, as_stream(stream)>>>(rois, E, (const float4*)gt_f32, gt_f64, gt_cls, G, bg_idx,
                                                                  eligible, cls, (float4*)targets);
    return check_launch("roi_targets");
}

int frcnn_preprocess_u8(const uint8_t* img_hwc, size_t n_pixels, const double* mean3_h, float* out, void* stream) {
    if (!img_hwc || !mean3_h || !out) return fail(FRCNN_E_ARG, "preprocess_u8: null pointer");
    if (n_pixels == 0) return FRCNN_OK;
    const size_t n = n_pixels * 3;
    size_t g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    k_preprocess_u8<<<(int)g, 256, 0, as_stream(stream)>>>(img_hwc, n, mean3_h[0], mean3_h[1], mean3_h[2], out);
    return check_launch("preprocess_u8");
}

}  // extern "C"
