/* A plain-C host on the drop-in boundary (include/frcnn_hip.h), with NO PyTorch in the process: the HIP runtime for device
 * memory and a stream, libfrcnn_hip.so for the arithmetic.  It is what a non-Python maintainer's binding does, and a check that
 * the library needs nothing from torch.  Walks a slice of the reference's proposal path on its own data and checks every
 * result against loops written here from the reference's formulas:
 *   rpn_util._get_all_anchor_coords (rpn_util.py:276-298)   frcnn_anchors_image
 *   util.cross_ious (util.py:146-177)                        frcnn_cross_ious_f32
 *   det_util.nms (det_util.py:209-256)                       frcnn_nms_f64
 * Build (tests/test_abi_host_gpu.py does it):  gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/tools/c_abi_host.c
 *        -Lfaster_rcnn_amd -lfrcnn_hip -L/opt/rocm/lib -lamdhip64 -lm -o c_abi_host
 * Exit code 0 = every check passed. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "frcnn_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)
#define CHECK_FR(x) do { int r_ = (x); if (r_ != FRCNN_OK) { printf("frcnn error %d at line %d: %s\n", r_, __LINE__, frcnn_last_error()); return 3; } } while (0)

static unsigned rng_state = 12345u;
static unsigned rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main(void) {
    if (frcnn_device_count() < 1) { printf("no HIP device\n"); return 1; }
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    int fails = 0;

    /* ---- anchors: idx = (y*cols + x)*A + a; centre = (int)(stride*(x + .5)); x1 = cx - w/2 (floor), x2 = x1 + w */
    const int rows = 5, cols = 7, A = 3, stride = 16;
    const int32_t hw[3][2] = {{128, 128}, {90, 181}, {181, 90}};
    const int N = rows * cols * A;
    float* d_anchors; CHECK_HIP(hipMalloc((void**)&d_anchors, (size_t)N * 4 * sizeof(float)));
    CHECK_FR(frcnn_anchors_image(rows, cols, &hw[0][0], A, stride, d_anchors, stream));
    float* anchors = (float*)malloc((size_t)N * 4 * sizeof(float));
    CHECK_HIP(hipMemcpyAsync(anchors, d_anchors, (size_t)N * 4 * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    for (int y = 0; y < rows; ++y) for (int x = 0; x < cols; ++x) for (int a = 0; a < A; ++a) {
        const int i = (y * cols + x) * A + a, h = hw[a][0], w = hw[a][1];
        const int cx = (int)(stride * (x + 0.5)), cy = (int)(stride * (y + 0.5));
        const int fx = cx >= 0 ? cx - w / 2 : 0, fy = cy - h / 2;               /* Python floor division on non-negative w, h */
        const float want[4] = {(float)fx, (float)fy, (float)(fx + w), (float)(fy + h)};
        if (memcmp(anchors + 4 * i, want, sizeof want)) { if (fails < 5) printf("anchor %d differs\n", i); ++fails; }
    }

    /* ---- IoU of the anchors against two boxes, no "+1": inter / (a1 + a2 - inter), all in f32 in numpy's order */
    const float gt[2][4] = {{10.f, 5.f, 90.f, 70.f}, {-40.f, -30.f, 60.f, 40.f}};
    float *d_gt, *d_iou; CHECK_HIP(hipMalloc((void**)&d_gt, sizeof gt)); CHECK_HIP(hipMalloc((void**)&d_iou, (size_t)N * 2 * sizeof(float)));
    CHECK_HIP(hipMemcpyAsync(d_gt, gt, sizeof gt, hipMemcpyHostToDevice, stream));
    CHECK_FR(frcnn_cross_ious_f32(d_anchors, N, d_gt, 2, d_iou, stream));
    float* iou = (float*)malloc((size_t)N * 2 * sizeof(float));
    CHECK_HIP(hipMemcpyAsync(iou, d_iou, (size_t)N * 2 * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < N; ++i) for (int g = 0; g < 2; ++g) {
        const float* b = anchors + 4 * i;
        const float iw = fmaxf(0.f, fminf(b[2], gt[g][2]) - fmaxf(b[0], gt[g][0])), ih = fmaxf(0.f, fminf(b[3], gt[g][3]) - fmaxf(b[1], gt[g][1]));
        const float inter = iw * ih, a1 = (b[2] - b[0]) * (b[3] - b[1]), a2 = (gt[g][2] - gt[g][0]) * (gt[g][3] - gt[g][1]);
        const float want = inter / (a1 + a2 - inter);
        if (iou[2 * i + g] != want) { if (fails < 5) printf("iou %d,%d: %.9g vs %.9g\n", i, g, iou[2 * i + g], want); ++fails; }
    }

    /* ---- greedy NMS on score-ordered f64 boxes, "+1" pixel convention, keep while overlap <= thresh (det_util.py:209-256) */
    const int K = 700, max_boxes = 50;
    const double thresh = 0.5;
    double* boxes = (double*)malloc((size_t)K * 4 * sizeof(double));
    for (int i = 0; i < K; ++i) {
        const double x1 = rnd() % 500, y1 = rnd() % 300, w = 20 + rnd() % 200, h = 20 + rnd() % 150;
        boxes[4 * i] = x1; boxes[4 * i + 1] = y1; boxes[4 * i + 2] = x1 + w; boxes[4 * i + 3] = y1 + h;
    }
    double* d_boxes; int32_t *d_n, *d_keep, *d_nkeep; void* d_ws;
    const size_t ws_bytes = frcnn_nms_workspace_bytes(K);
    CHECK_HIP(hipMalloc((void**)&d_boxes, (size_t)K * 4 * sizeof(double))); CHECK_HIP(hipMalloc((void**)&d_n, 4)); CHECK_HIP(hipMalloc((void**)&d_keep, max_boxes * 4));
    CHECK_HIP(hipMalloc((void**)&d_nkeep, 4)); CHECK_HIP(hipMalloc(&d_ws, ws_bytes));
    CHECK_HIP(hipMemcpyAsync(d_boxes, boxes, (size_t)K * 4 * sizeof(double), hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemcpyAsync(d_n, &K, 4, hipMemcpyHostToDevice, stream));
    CHECK_FR(frcnn_nms_f64(d_boxes, d_n, K, thresh, max_boxes, d_keep, d_nkeep, d_ws, ws_bytes, stream));
    int32_t keep[50], n_keep = -1;
    CHECK_HIP(hipMemcpyAsync(keep, d_keep, sizeof keep, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(&n_keep, d_nkeep, 4, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    char* dead = (char*)calloc(K, 1);
    int want_keep[50], nw = 0;
    for (int i = 0; i < K && nw < max_boxes; ++i) {
        if (dead[i]) continue;
        want_keep[nw++] = i;
        const double ai = (boxes[4 * i + 2] - boxes[4 * i] + 1) * (boxes[4 * i + 3] - boxes[4 * i + 1] + 1);
        for (int j = i + 1; j < K; ++j) {
            if (dead[j]) continue;
            const double w = fmin(boxes[4 * i + 2], boxes[4 * j + 2]) - fmax(boxes[4 * i], boxes[4 * j]) + 1;
            const double h = fmin(boxes[4 * i + 3], boxes[4 * j + 3]) - fmax(boxes[4 * i + 1], boxes[4 * j + 1]) + 1;
            if (w <= 0 || h <= 0) continue;
            const double aj = (boxes[4 * j + 2] - boxes[4 * j] + 1) * (boxes[4 * j + 3] - boxes[4 * j + 1] + 1);
            const double inter = w * h;
            if (inter / (ai + aj - inter) > thresh) dead[j] = 1;
        }
    }
    if (n_keep != nw) { printf("nms kept %d, expected %d\n", n_keep, nw); ++fails; }
    for (int k = 0; k < nw && k < n_keep; ++k) if (keep[k] != want_keep[k]) { if (fails < 5) printf("nms pick %d: %d vs %d\n", k, keep[k], want_keep[k]); ++fails; }

    /* ---- error path: a refused call leaves a message and no exception */
    if (frcnn_nms_f64(NULL, d_n, K, thresh, max_boxes, d_keep, d_nkeep, d_ws, ws_bytes, stream) == FRCNN_OK || !frcnn_last_error()[0]) { printf("null boxes accepted\n"); ++fails; }

    printf("c_abi_host: %d anchors, %d IoUs, %d NMS picks checked, %d failures (library version %d)\n", N, 2 * N, nw, fails, frcnn_version());
    return fails ? 4 : 0;
}
