/* A plain-C host on the drop-in boundary (include/frcnn_hip.h), with NO PyTorch in the process: the HIP runtime for device
 * memory and a stream, libfrcnn_hip.so for the arithmetic.  It is what a non-Python maintainer's binding does, and a check that
 * the library needs nothing from torch.  Walks a slice of the reference's proposal path on its own data and checks every
 * result against loops written here from the reference's formulas:
 *   rpn_util._get_all_anchor_coords (rpn_util.py:276-298)   frcnn_anchors_image
 *   util.cross_ious (util.py:146-177)                        frcnn_cross_ious_f32
 *   det_util.nms (det_util.py:209-256)                       frcnn_nms_f64
 *   one Conv2D + folded BatchNorm + ReLU group (resnet.py:150-176) on the matrix path the library's own policy names
 *   (frcnn_conv2d_engine -> frcnn_pack_conv_weights_h3, frcnn_amax_f32, frcnn_conv2d_fwd_h3), against a double-precision loop
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

    /* ---- one 1x1 convolution + per-channel scale / shift + ReLU over 12 288 pixels (64 -> 128 channels): the engine the POLICY picks
     * for a host that holds f16x3 planes, its filter planes, the input's magnitude record, the launch; y against a double loop */
    {
        const int cn = 1, ch = 96, cw = 128, cin = 64, cout = 128, M = cn * ch * cw;
        frcnn_conv_desc d;
        memset(&d, 0, sizeof d);
        d.n = cn; d.h = ch; d.w = cw; d.cin = cin; d.cout = cout; d.kh = d.kw = 1; d.stride = 1; d.ho = ch; d.wo = cw; d.act = FRCNN_ACT_RELU;
        const int engine = frcnn_conv2d_engine(&d, FRCNN_ENGINE_H3, 0);
        if (engine != FRCNN_ENGINE_H3) { printf("policy: engine %d for a 12 288 x 128 launch, expected f16x3\n", engine); ++fails; }
        frcnn_conv_desc small = d; small.h = 16; small.w = 16; small.ho = 16; small.wo = 16;
        if (frcnn_conv2d_engine(&small, FRCNN_ENGINE_H3, 0) != FRCNN_ENGINE_NATIVE || frcnn_conv2d_engine(&d, FRCNN_ENGINE_NATIVE, 1) != FRCNN_ENGINE_NATIVE
            || frcnn_conv2d_engine(&d, FRCNN_ENGINE_X6, 0) != FRCNN_ENGINE_X6 || frcnn_conv2d_engine(NULL, 0, 0) >= 0) { printf("policy: small grid / native / x6 answers differ\n"); ++fails; }
        const int kp = frcnn_conv_packed_k(1, 1, cin);
        float* x = (float*)malloc((size_t)M * cin * 4); float* wv = (float*)malloc((size_t)cin * cout * 4);
        float sc[128], sh[128];
        for (int i = 0; i < M * cin; ++i) x[i] = ((int)(rnd() % 2001) - 1000) * 1e-3f;
        for (int i = 0; i < cin * cout; ++i) wv[i] = ((int)(rnd() % 2001) - 1000) * 1e-4f;          /* HWIO: [cin][cout] for 1x1 */
        for (int c = 0; c < cout; ++c) { sc[c] = 1.f + (int)(rnd() % 100) * 1e-3f; sh[c] = ((int)(rnd() % 200) - 100) * 1e-3f; }
        float *dx, *dw, *dpk, *dsc, *dsh, *dy, *drec; void* dplanes;
        const size_t rec_bytes = (size_t)frcnn_amax_record_floats() * 4;
        CHECK_HIP(hipMalloc((void**)&dx, (size_t)M * cin * 4)); CHECK_HIP(hipMalloc((void**)&dw, (size_t)cin * cout * 4)); CHECK_HIP(hipMalloc((void**)&dpk, (size_t)cout * kp * 4));
        CHECK_HIP(hipMalloc((void**)&dsc, cout * 4)); CHECK_HIP(hipMalloc((void**)&dsh, cout * 4)); CHECK_HIP(hipMalloc((void**)&dy, (size_t)M * cout * 4));
        CHECK_HIP(hipMalloc((void**)&drec, 2 * rec_bytes)); CHECK_HIP(hipMalloc(&dplanes, frcnn_conv_h3_planes_bytes(cout, kp)));
        CHECK_HIP(hipMemcpyAsync(dx, x, (size_t)M * cin * 4, hipMemcpyHostToDevice, stream)); CHECK_HIP(hipMemcpyAsync(dw, wv, (size_t)cin * cout * 4, hipMemcpyHostToDevice, stream));
        CHECK_HIP(hipMemcpyAsync(dsc, sc, cout * 4, hipMemcpyHostToDevice, stream)); CHECK_HIP(hipMemcpyAsync(dsh, sh, cout * 4, hipMemcpyHostToDevice, stream));
        CHECK_FR(frcnn_pack_conv_weights(dw, 1, 1, cin, cout, dpk, stream));
        CHECK_FR(frcnn_pack_conv_weights_h3(dpk, cout, kp, dplanes, stream));
        CHECK_FR(frcnn_amax_clear(drec, 2, stream));
        CHECK_FR(frcnn_amax_f32(dx, (size_t)M * cin, drec, stream));
        float* y_rec = drec + frcnn_amax_record_floats();
        CHECK_FR(frcnn_conv2d_fwd_h3(&d, dx, drec, dplanes, dsc, dsh, NULL, NULL, dy, y_rec, NULL, 0, stream));
        float* y = (float*)malloc((size_t)M * cout * 4); float* rec = (float*)malloc(rec_bytes);
        CHECK_HIP(hipMemcpyAsync(y, dy, (size_t)M * cout * 4, hipMemcpyDeviceToHost, stream)); CHECK_HIP(hipMemcpyAsync(rec, y_rec, rec_bytes, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        double worst = 0; float ymax = 0, recmax = 0;
        for (int m = 0; m < M; m += 7) for (int c = 0; c < cout; ++c) {                                  /* every seventh pixel */
            double acc = 0, mag = 0;
            for (int k = 0; k < cin; ++k) { const double t = (double)x[(size_t)m * cin + k] * wv[(size_t)k * cout + c]; acc += t; mag += fabs(t); }
            double want = acc * sc[c] + sh[c]; if (want < 0) want = 0;
            const double e = fabs(y[(size_t)m * cout + c] - want) / (mag * fabs(sc[c]) + 1e-30);
            if (e > worst) worst = e;
        }
        for (size_t i = 0; i < (size_t)M * cout; ++i) if (y[i] > ymax) ymax = y[i];
        for (int i = 0; i < frcnn_amax_record_floats(); ++i) if (rec[i] > recmax) recmax = rec[i];
        if (worst > 4e-7) { printf("conv (f16x3): error %.3g of sum|ab|\n", worst); ++fails; }
        if (recmax != ymax) { printf("conv (f16x3): record holds %.9g, max|y| is %.9g\n", recmax, ymax); ++fails; }
        if (frcnn_conv2d_fwd_h3(&d, dx, NULL, dplanes, dsc, dsh, NULL, NULL, dy, NULL, NULL, 0, stream) == FRCNN_OK) { printf("conv (f16x3): a launch without a magnitude record was accepted\n"); ++fails; }
        printf("c_abi_host: conv 12288x128x64 on the f16x3 engine, max error %.3g of sum|ab|\n", worst);
    }

    /* ---- error path: a refused call leaves a message and no exception */
    if (frcnn_nms_f64(NULL, d_n, K, thresh, max_boxes, d_keep, d_nkeep, d_ws, ws_bytes, stream) == FRCNN_OK || !frcnn_last_error()[0]) { printf("null boxes accepted\n"); ++fails; }

    printf("c_abi_host: %d anchors, %d IoUs, %d NMS picks checked, %d failures (library version %d)\n", N, 2 * N, nw, fails, frcnn_version());
    return fails ? 4 : 0;
}
