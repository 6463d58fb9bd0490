// Detection post-process (voc_dets.get_dets, voc_dets.py:51-88) on the device, gfx950.
//
// One workgroup handles one image's <= 512 scored RoIs entirely in LDS: per-RoI arg-max class,
// f64 box decode (util.transform, util.py:55-74) x stride, a global score ranking, the
// class-restricted suppression bit matrix, the greedy scan (one wave, bitmap in registers) and
// the reference's emission order (classes in first-seen order, NMS pick order inside a class).
// Latency bound, a few microseconds; what matters is that it removes the host round trip.
//
// dtype flow = the reference under its pinned numpy 1.13 (legacy scalar promotion):
//   cxa = f64(f32(x1+x2))/2, wa = f32(x2-x1), cx = f64(f32(tx*wa)) + cxa, w = exp(f64(tw))*f64(wa)
//   `confidence < det_threshold` (voc_dets.py:57): np.float32 scalar against a Python float = an f64 comparison there
//   (checked under numpy 1.26's legacy rules: np.float32(0.7) < 0.7 is True; NEP 50 compares in f32 and says False)
// Compiled with -ffp-contract=off.
#include "common.h"
#include <stdlib.h>

namespace frcnn {

typedef unsigned long long u64;
constexpr int DET_MAX = 512;
constexpr int DET_WORDS = DET_MAX / 64;
constexpr int DET_MAX_CLASSES = 128;

// readfirstlane returns a SIGNED int: go through unsigned or the low word sign-extends into the high one
__device__ __forceinline__ u64 det_uniform(u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ bool det_suppresses(const double4 a, const double4 b, double thresh) {
    const double area_a = (a.z - a.x + 1.0) * (a.w - a.y + 1.0);
    const double area_b = (b.z - b.x + 1.0) * (b.w - b.y + 1.0);
    const double w = fmax(0.0, fmin(a.z, b.z) - fmax(a.x, b.x) + 1.0);
    const double h = fmax(0.0, fmin(a.w, b.w) - fmax(a.y, b.y) + 1.0);
    const double inter = w * h;
    return !(inter / (area_a + area_b - inter) <= thresh);
}

__global__ void __launch_bounds__(DET_MAX) k_detections(
        const float4* rois, const int32_t* n_rois_ptr, int max_rows, const float* out_cls, const float* out_reg, int C,
        int bg_idx, double det_threshold, double stride, double resize_ratio, double nms_thresh,
        int32_t* det_cls, float* det_prob, int32_t* det_bbox, int32_t* det_roi, int32_t* n_dets,
        const double* dyn, int roi_batch) {
    __shared__ double4 s_box[DET_MAX];      // sorted by score
    __shared__ float s_prob[DET_MAX];
    __shared__ int s_cls[DET_MAX];
    __shared__ int s_roi[DET_MAX];
    __shared__ u64 s_mask[DET_MAX][DET_WORDS];
    __shared__ u64 s_kept[DET_WORDS];
    __shared__ int s_first[DET_MAX_CLASSES];
    __shared__ int s_nvalid;
    // unsorted staging aliases the mask storage (dead before the mask is written)
    double4* u_box = reinterpret_cast<double4*>(&s_mask[0][0]);
    float* u_prob = reinterpret_cast<float*>(u_box + DET_MAX);
    int* u_cls = reinterpret_cast<int*>(u_prob + DET_MAX);

    const int r = threadIdx.x;
    // frcnn_detections_dyn: the two per-image scalars live in device memory (a captured graph serves every image of its size)
    // and the reference's padded duplicates (voc_dets.py:42-46: the last batch filled with copies of ITS first RoI, which
    // frcnn_gather_rois has laid out) are scored like any other row (:51 loops over all num_rois rows of every batch)
    if (dyn) { resize_ratio = dyn[0]; det_threshold = dyn[1]; }
    const int n_live = *n_rois_ptr;
    const int n_scored = (roi_batch > 0 && n_live > 0) ? (n_live + roi_batch - 1) / roi_batch * roi_batch : n_live;
    const int n = min(min(n_scored, max_rows), DET_MAX);
    if (r < DET_MAX_CLASSES) s_first[r] = 0x7fffffff;
    if (r == 0) s_nvalid = 0;
    if (r < DET_WORDS) s_kept[r] = 0;
    __syncthreads();

    // ---- 1. per-RoI class / confidence / decoded box
    int cls = -1; float conf = 0.0f; double4 box = make_double4(0, 0, 0, 0);
    if (r < n) {
        const float* pc = out_cls + (size_t)r * C;
        int best = 0; float bv = pc[0];
#pragma unroll 8
        for (int c = 1; c < C; ++c) { const float v = pc[c]; if (v > bv) { bv = v; best = c; } }   // np.argmax: first max
        if (best != bg_idx && !((double)bv < det_threshold)) {
            cls = best; conf = bv;
            const float4 roi = rois[r];
            const float* pr = out_reg + (size_t)r * 4 * (C - 1) + 4 * best;
            const float tx = pr[0] / 10.0f, ty = pr[1] / 10.0f, tw = pr[2] / 5.0f, th = pr[3] / 5.0f;   // / BBREG_MULTIPLIERS (f32)
            const double cxa = (double)(roi.x + roi.z) / 2.0, cya = (double)(roi.y + roi.w) / 2.0;
            const float wa = roi.z - roi.x, ha = roi.w - roi.y;
            const double cx = (double)(tx * wa) + cxa, cy = (double)(ty * ha) + cya;
            const double w = exp((double)tw) * (double)wa, h = exp((double)th) * (double)ha;
            const double x = cx - w / 2.0, y = cy - h / 2.0;
            box = make_double4(stride * x, stride * y, stride * (x + w), stride * (y + h));
            atomicMin(&s_first[best], r);
        }
    }
    u_box[r] = box; u_prob[r] = conf; u_cls[r] = cls;
    const u64 vb = __ballot(cls >= 0);
    if ((r & 63) == 0 && vb) atomicAdd(&s_nvalid, __popcll(vb));
    __syncthreads();
    const int nv = s_nvalid;

    // ---- 2. rank by (score desc, roi index asc) among valid rows, scatter into sorted order
    int rank = -1;
    if (cls >= 0) {
        rank = 0;
#pragma unroll 8
        for (int j = 0; j < n; ++j) {
            const float pj = u_prob[j];
            rank += (u_cls[j] >= 0) && (pj > conf || (pj == conf && j < r));
        }
    }
    __syncthreads();
    const double4 mybox = box;
    if (rank >= 0) { s_box[rank] = mybox; s_prob[rank] = conf; s_cls[rank] = cls; s_roi[rank] = r; }
    __syncthreads();          // staging (aliasing s_mask) is dead from here on

    // ---- 3. class-restricted suppression bits (row i vs later rows j > i)
    const int W = (nv + 63) / 64;
    if (r < nv) {
        const double4 me = s_box[r]; const int mc = s_cls[r];
        for (int w = 0; w < W; ++w) {
            u64 bits = 0;
            const int j0 = w * 64, jn = min(64, nv - j0);
            if (j0 + 63 > r) {                              // words wholly at or below the diagonal are never read
#pragma unroll 4
                for (int j = max(0, r + 1 - j0); j < jn; ++j)
                    if (s_cls[j0 + j] == mc && det_suppresses(me, s_box[j0 + j], nms_thresh)) bits |= 1ull << j;
            }
            s_mask[r][w] = bits;
        }
    }
    __syncthreads();

    // ---- 4. greedy scan by wave 0 (max_boxes = 2000 in the reference never binds for <= 512 rows)
    if (r < 64) {
        u64 removed = 0;                     // lane w (< DET_WORDS) holds bitmap word w
        for (int c = 0; c < W; ++c) {
            const int base = c * 64, i = base + r;
            const u64 diag = i < nv ? s_mask[i][c] : 0ull;
            const unsigned rl = __builtin_amdgcn_readlane((unsigned)removed, c), rh = __builtin_amdgcn_readlane((unsigned)(removed >> 32), c);
            const int cnt = min(64, nv - base);
            const u64 vmask = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
            u64 alive = ~(((u64)rh << 32) | rl) & vmask;
            alive = det_uniform(alive);
            u64 kept = 0;
            while (alive) {
                const int b = __builtin_ctzll(alive);
                kept |= 1ull << b;
                alive &= ~(1ull << b);
                const unsigned dl = __builtin_amdgcn_readlane((unsigned)diag, b), dh = __builtin_amdgcn_readlane((unsigned)(diag >> 32), b);
                alive &= ~(((u64)dh << 32) | dl);
                alive = det_uniform(alive);
            }
            if (r == 0) s_kept[c] = kept;
            u64 k = kept;
            while (k) {                      // four kept rows per round: independent LDS reads in flight together
                u64 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool has = k != 0;
                    const int b = has ? __builtin_ctzll(k) : 0;
                    k = has ? k & (k - 1) : 0;
                    v[u] = (has && r > c && r < W) ? s_mask[base + b][r] : 0ull;
                }
                removed |= (v[0] | v[1]) | (v[2] | v[3]);
            }
        }
    }
    __syncthreads();

    // ---- 5. emission order: classes by first-seen RoI index, NMS pick order inside a class
    int* s_fj = reinterpret_cast<int*>(&s_mask[0][0]);     // the mask is dead after the scan: per-row class rank key
    const bool kept = r < nv && ((s_kept[r >> 6] >> (r & 63)) & 1);
    if (r < nv) s_fj[r] = kept ? s_first[s_cls[r]] : 0x7fffffff;      // dropped rows sort behind everything
    __syncthreads();
    if (kept) {
        const int mc = s_cls[r], mf = s_first[mc];
        int pos = 0;
#pragma unroll 8
        for (int j = 0; j < nv; ++j) {
            const int fj = s_fj[j];
            pos += (fj < mf) || (fj == mf && j < r);
        }
        const double4 b = s_box[r];
        det_cls[pos] = mc;
        det_prob[pos] = s_prob[r];
        det_roi[pos] = s_roi[r];
        // int(round(v / resize_ratio)): Python round() on a float is half-to-even
        det_bbox[4 * pos + 0] = (int)rint(b.x / resize_ratio);
        det_bbox[4 * pos + 1] = (int)rint(b.y / resize_ratio);
        det_bbox[4 * pos + 2] = (int)rint(b.z / resize_ratio);
        det_bbox[4 * pos + 3] = (int)rint(b.w / resize_ratio);
    }
    const u64 kb = __ballot(kept);
    __shared__ int s_total;
    if (r == 0) s_total = 0;
    __syncthreads();
    if ((r & 63) == 0 && kb) atomicAdd(&s_total, __popcll(kb));
    __syncthreads();
    for (int i = s_total + r; i < max_rows; i += DET_MAX) {   // rows past the last detection: the same "empty" values every call
        det_cls[i] = -1; det_prob[i] = 0.0f; det_roi[i] = -1;
        det_bbox[4 * i + 0] = det_bbox[4 * i + 1] = det_bbox[4 * i + 2] = det_bbox[4 * i + 3] = 0;
    }
    if (r == 0) {
        n_dets[0] = s_total;
        if (dyn) n_dets[1] = n_live;          // counts[1]: what voc_dets.get_dets prints as "num rois"
    }
}

}  // namespace frcnn

using namespace frcnn;

extern "C" int frcnn_detections(const float* rois, const int32_t* n_rois, int max_rows, const float* out_cls, const float* out_reg,
                                int num_classes, int bg_idx, double det_threshold, double stride, double resize_ratio, double nms_thresh,
                                int32_t* det_cls, float* det_prob, int32_t* det_bbox, int32_t* det_roi, int32_t* n_dets, void* stream) {
    if (!rois || !n_rois || !out_cls || !out_reg || !det_cls || !det_prob || !det_bbox || !det_roi || !n_dets)
        return fail(FRCNN_E_ARG, "detections: null pointer");
    if (max_rows <= 0 || max_rows > DET_MAX) return fail(FRCNN_E_ARG, "detections: max_rows=%d exceeds %d", max_rows, DET_MAX);
    if (num_classes < 2 || num_classes > DET_MAX_CLASSES) return fail(FRCNN_E_ARG, "detections: num_classes=%d out of range", num_classes);
    k_detections<<<1, DET_MAX, 0, as_stream(stream)>>>((const float4*)rois, n_rois, max_rows, out_cls, out_reg, num_classes, bg_idx,
                                                       det_threshold, stride, resize_ratio, nms_thresh, det_cls, det_prob, det_bbox, det_roi, n_dets,
                                                       nullptr, 0);
    return check_launch("detections");
}

extern "C" int frcnn_detections_dyn(const float* rois, const int32_t* n_rois, int roi_batch, int max_rows, const float* out_cls, const float* out_reg,
                                    int num_classes, int bg_idx, double stride, double nms_thresh, const double* dyn,
                                    int32_t* det_cls, float* det_prob, int32_t* det_bbox, int32_t* det_roi, int32_t* counts, void* stream) {
    if (!rois || !n_rois || !out_cls || !out_reg || !dyn || !det_cls || !det_prob || !det_bbox || !det_roi || !counts)
        return fail(FRCNN_E_ARG, "detections_dyn: null pointer");
    if (max_rows <= 0 || max_rows > DET_MAX) return fail(FRCNN_E_ARG, "detections_dyn: max_rows=%d exceeds %d", max_rows, DET_MAX);
    if (roi_batch < 0) return fail(FRCNN_E_ARG, "detections_dyn: roi_batch=%d", roi_batch);
    if (num_classes < 2 || num_classes > DET_MAX_CLASSES) return fail(FRCNN_E_ARG, "detections_dyn: num_classes=%d out of range", num_classes);
    if (((uintptr_t)dyn & 7) != 0) return fail(FRCNN_E_ARG, "detections_dyn: dyn must be 8-byte aligned");
    k_detections<<<1, DET_MAX, 0, as_stream(stream)>>>((const float4*)rois, n_rois, max_rows, out_cls, out_reg, num_classes, bg_idx,
                                                       0.0, stride, 1.0, nms_thresh, det_cls, det_prob, det_bbox, det_roi, counts, dyn, roi_batch);
    return check_launch("detections_dyn");
}
