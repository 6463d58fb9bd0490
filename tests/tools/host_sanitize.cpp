// Host-side sanitizer driver ("sanitizers on the CPU build").  Linked against the C-ABI objects compiled with
// -fsanitize=address,undefined (host code only: GPU ASan is not available on this pool) by scripts/sanitize_host.sh.
// It walks the HOST logic behind the entry points -- tile / split-K policy, workspace sizing, job-table packing, argument
// validation and error strings -- over a sweep of shapes.  No kernel needs to run: without a GPU every launch fails
// cleanly with a HIP error, which is itself one of the paths checked.
#include "../../include/frcnn_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

static unsigned rng_state = 12345u;
static unsigned rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static int pick(std::initializer_list<int> v) { return v.begin()[rnd() % v.size()]; }

int main() {
    int checked = 0, failures = 0;
    // ---- conv descriptors: workspace sizing runs choose_config / choose_splits for both precisions
    for (int it = 0; it < 20000; ++it) {
        frcnn_conv_desc d;
        memset(&d, 0, sizeof d);
        d.n = pick({1, 1, 1, 2, 5, 64, 300});
        d.h = 1 + rnd() % 160; d.w = 1 + rnd() % 260;
        if (d.n > 5) { d.h = pick({1, 7, 14}); d.w = d.h; }
        d.cin = pick({3, 4, 32, 64, 128, 256, 512, 1024, 2048, 96, 36});
        d.cout = pick({9, 36, 64, 101, 128, 256, 512, 1024, 2048, 84, 320});
        d.kh = d.kw = pick({1, 1, 3, 3, 7});
        d.stride = pick({1, 1, 2});
        const bool same = rnd() & 1;
        d.ho = same ? (d.h + d.stride - 1) / d.stride : (d.h - d.kh) / d.stride + 1;
        d.wo = same ? (d.w + d.stride - 1) / d.stride : (d.w - d.kw) / d.stride + 1;
        if (d.ho <= 0 || d.wo <= 0) continue;
        d.pad_top = same ? ((d.ho - 1) * d.stride + d.kh - d.h > 0 ? ((d.ho - 1) * d.stride + d.kh - d.h) / 2 : 0) : 0;
        d.pad_left = same ? ((d.wo - 1) * d.stride + d.kw - d.w > 0 ? ((d.wo - 1) * d.stride + d.kw - d.w) / 2 : 0) : 0;
        d.act = rnd() % 3;
        d.tile = pick({0, 0, 0, 50, 2, 21, 22, 23, 26, 61, 62, 123, 302, 42, 45, 46, 47, 48, 49, 60, 999});
        d.layout = (d.n > 5 && (d.cin % 32) == 0) ? (rnd() & 1) : 0;
        const size_t a = frcnn_conv2d_workspace_bytes(&d);
        const size_t b = frcnn_conv2d_workspace_bytes_bf16(&d);
        const size_t c = frcnn_conv2d_wgrad_workspace_bytes(&d);
        const size_t dual = frcnn_conv2d_dual_workspace_bytes(&d);          // round 3: the two-layer launch's split-K workspace
        if (dual > ((size_t)1 << 33) || (dual && (d.cin % 32))) { printf("implausible dual workspace %zu\n", dual); ++failures; }
        const int k = frcnn_conv_packed_k(d.kh, d.kw, d.cin);
        if (k < d.kh * d.kw * d.cin) { printf("packed_k too small for %dx%dx%d\n", d.kh, d.kw, d.cin); ++failures; }
        if (a > ((size_t)1 << 33) || b > ((size_t)1 << 33) || c > ((size_t)1 << 36)) { printf("implausible workspace %zu %zu %zu\n", a, b, c); ++failures; }
        ++checked;
    }
    // ---- weight-gradient job tables (sized on the host, never launched here)
    for (int it = 0; it < 200; ++it) {
        std::vector<frcnn_wgrad_job> jobs(1 + rnd() % 70);
        for (auto& j : jobs) {
            memset(&j, 0, sizeof j);
            j.d.n = 1; j.d.h = j.d.ho = 1 + rnd() % 40; j.d.w = j.d.wo = 1 + rnd() % 64;
            j.d.cin = pick({64, 128, 256, 512, 1024}); j.d.cout = pick({9, 36, 64, 256, 1024, 2048});
            j.d.kh = j.d.kw = pick({1, 3}); j.d.stride = 1;
            j.in_bf16 = rnd() & 1;
        }
        const size_t ws = frcnn_conv2d_wgrad_batch_workspace_bytes(jobs.data(), (int)jobs.size());
        if (ws == 0) { printf("empty batch workspace\n"); ++failures; }
        // null operands are refused before any launch
        if (frcnn_conv2d_wgrad_batch(jobs.data(), (int)jobs.size(), nullptr, 0, nullptr) == FRCNN_OK) { printf("wgrad_batch accepted a null workspace\n"); ++failures; }
        ++checked;
    }
    // ---- argument validation and the error string
    frcnn_conv_desc d;
    memset(&d, 0, sizeof d);
    d.n = 1; d.h = d.w = d.ho = d.wo = 8; d.cin = 64; d.cout = 64; d.kh = d.kw = 1; d.stride = 1;
    if (frcnn_conv2d_fwd(&d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == FRCNN_OK) { printf("conv2d_fwd accepted null tensors\n"); ++failures; }
    const char* msg = frcnn_last_error();
    if (!msg || !strstr(msg, "conv2d")) { printf("error string missing: %s\n", msg ? msg : "(null)"); ++failures; }
    if (frcnn_roi_crop_resize_bwd(nullptr, 0, 0, 0, nullptr, 1, 7, nullptr, nullptr) == FRCNN_OK) { printf("roi bwd accepted a bad shape\n"); ++failures; }
    if (frcnn_loss_rpn_cls_ws(nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr) == FRCNN_OK) { printf("loss accepted nulls\n"); ++failures; }
    if (frcnn_loss_workspace_bytes() < 1024) { printf("loss workspace too small\n"); ++failures; }
    if (frcnn_refresh_packed(nullptr, 3, nullptr) == FRCNN_OK) { printf("refresh_packed accepted a null table\n"); ++failures; }
    // round 3 entry points: refused before any launch
    float dummy = 0.0f;
    if (frcnn_conv2d_fwd_dual(&d, &dummy, &dummy, nullptr, nullptr, &dummy, 64, 1, &dummy, 0, nullptr, 0, nullptr) == FRCNN_OK) { printf("dual accepted n1 == cout\n"); ++failures; }
    if (frcnn_conv2d_fwd_dual(&d, &dummy, &dummy, nullptr, nullptr, &dummy, 32, 1, nullptr, 0, nullptr, 0, nullptr) == FRCNN_OK) { printf("dual accepted a null second output\n"); ++failures; }
    d.cin = 48;
    if (frcnn_conv2d_fwd_dual(&d, &dummy, &dummy, nullptr, nullptr, &dummy, 32, 1, &dummy, 0, nullptr, 0, nullptr) == FRCNN_OK) { printf("dual accepted cin %% 32 != 0\n"); ++failures; }
    if (frcnn_stem_bf16_fwd(nullptr, 1, 600, 1000, nullptr, nullptr, nullptr, nullptr, nullptr) == FRCNN_OK) { printf("stem accepted nulls\n"); ++failures; }
    if (frcnn_stem_bf16_fwd(&dummy, 1, 5, 5, &dummy, &dummy, &dummy, &dummy, nullptr) == FRCNN_OK) { printf("stem accepted a 5x5 image\n"); ++failures; }
    if (frcnn_stem_bf16_packed_elems() != 64 * 176) { printf("stem packed size\n"); ++failures; }
    if (frcnn_roi_crop_resize_fwd_bf16_batch(&dummy, 2, 38, 94, 12, &dummy, 300, 7, nullptr, 0, 1, &dummy, nullptr) == FRCNN_OK) { printf("roi batch accepted C %% 8 != 0\n"); ++failures; }
    if (frcnn_roi_crop_resize_fwd_bf16_batch(nullptr, 2, 38, 94, 16, nullptr, 300, 7, nullptr, 0, 1, nullptr, nullptr) == FRCNN_OK) { printf("roi batch accepted nulls\n"); ++failures; }
    checked += 14;
    // ---- round 4: the split-bf16 engine's host logic (split-K policy / workspace sizing over random descriptors), the resize tap
    // tables (a pure host function), argument validation of the new entry points
    for (int it = 0; it < 20000; ++it) {
        frcnn_conv_desc q;
        memset(&q, 0, sizeof q);
        q.n = 1 + rnd() % 4; q.h = 1 + rnd() % 200; q.w = 1 + rnd() % 200;
        q.cin = pick({32, 64, 128, 256, 512, 1024, 2048, 48, 3}); q.cout = pick({18, 45, 64, 128, 256, 512, 1024, 2048});
        q.kh = q.kw = pick({1, 3}); q.stride = pick({1, 2});
        q.ho = (q.h + q.stride - 1) / q.stride; q.wo = (q.w + q.stride - 1) / q.stride;
        q.tile = pick({0, 50, 71, 74, 76, 77, 78, 274, 374, 578, 1674, 23});
        q.layout = rnd() % 2;
        const size_t need = frcnn_conv2d_x6_workspace_bytes(&q);
        const long long rows = (long long)q.n * q.ho * q.wo;
        const long long tiles64 = ((rows + 63) / 64) * ((q.cout + 63) / 64), tiles128 = ((rows + 127) / 128) * ((q.cout + 127) / 128);
        // tickets + slices x tiles x one f32 tile, on 64x64 or (taller small grids) 128x128 tiles
        const bool ok64 = need >= 16384 && (need - 16384) % (64 * 64 * 4) == 0 && (long long)((need - 16384) / (64 * 64 * 4)) % tiles64 == 0;
        const bool ok128 = need >= 16384 && (need - 16384) % (128 * 128 * 4) == 0 && (long long)((need - 16384) / (128 * 128 * 4)) % tiles128 == 0;
        if (need && !ok64 && !ok128) { printf("x6 workspace size inconsistent\n"); ++failures; }
        if (need && (q.cin % 32)) { printf("x6 split-K offered for cin %% 32 != 0\n"); ++failures; }
        const int n1 = (rnd() % 3) ? 0 : pick({9, 64, 128, 512});
        const int xc = frcnn_conv2d_x6_config(&q, n1);
        const int asked = q.tile % 100;
        if (xc < 71 || xc > 77 || (asked >= 71 && asked <= 77 && xc != asked)) { printf("x6 tile code %d out of range / not the one asked for\n", xc); ++failures; }
        if (!(asked >= 71 && asked <= 77) && n1 > 0 && (n1 % 128) && xc != 74 && xc != 77) { printf("x6 paired launch: boundary %d on a 128-wide tile\n", n1); ++failures; }
        ++checked;
    }
    for (int it = 0; it < 300; ++it) {
        const int dst = 1 + rnd() % 1600, src = 1 + rnd() % 1600;
        std::vector<int32_t> tab((size_t)dst * 8 + 8, 0x5a5a5a5a);
        if (frcnn_resize_cubic_taps(dst, src, tab.data()) != FRCNN_OK) { printf("resize taps failed\n"); ++failures; }
        for (int d2 = 0; d2 < dst; ++d2)
            for (int k = 0; k < 4; ++k)
                if (tab[8 * d2 + k] < 0 || tab[8 * d2 + k] >= src || tab[8 * d2 + 4 + k] < -32768 || tab[8 * d2 + 4 + k] > 32767) { printf("resize tap out of range\n"); ++failures; }
        if (tab[(size_t)dst * 8] != 0x5a5a5a5a) { printf("resize taps wrote past the table\n"); ++failures; }
        ++checked;
    }
    if (frcnn_resize_cubic_taps(0, 5, nullptr) == FRCNN_OK) { printf("resize taps accepted a bad argument\n"); ++failures; }
    if (frcnn_resize_cubic_u8(nullptr, 1, 1, nullptr, nullptr, 1, 1, 0, nullptr, nullptr) == FRCNN_OK) { printf("resize accepted nulls\n"); ++failures; }
    if (frcnn_pack_conv_weights_x6(&dummy, 64, 48, &dummy, nullptr) == FRCNN_OK) { printf("pack x6 accepted k %% 32 != 0\n"); ++failures; }
    memset(&d, 0, sizeof d);
    d.n = 1; d.h = d.w = d.ho = d.wo = 8; d.cin = 48; d.cout = 64; d.kh = d.kw = 1; d.stride = 1;
    if (frcnn_conv2d_fwd_x6(&d, &dummy, &dummy, nullptr, nullptr, nullptr, nullptr, &dummy, nullptr, 0, nullptr) == FRCNN_OK) { printf("x6 accepted cin %% 32 != 0\n"); ++failures; }
    d.cin = 64;
    if (frcnn_conv2d_fwd_dual_x6(&d, &dummy, &dummy, nullptr, nullptr, &dummy, 64, 1, &dummy, 0, nullptr) == FRCNN_OK) { printf("dual x6 accepted n1 == cout\n"); ++failures; }
    {
        int32_t i32 = 0; double dyn_misaligned[2];
        if (frcnn_detections_dyn(&dummy, &i32, 64, 600, &dummy, &dummy, 21, 20, 16.0, 0.5, dyn_misaligned, &i32, &dummy, &i32, &i32, &i32, nullptr) == FRCNN_OK) { printf("detections_dyn accepted 600 rows\n"); ++failures; }
        if (frcnn_detections_dyn(&dummy, &i32, 64, 320, &dummy, &dummy, 21, 20, 16.0, 0.5, nullptr, &i32, &dummy, &i32, &i32, &i32, nullptr) == FRCNN_OK) { printf("detections_dyn accepted a null dyn\n"); ++failures; }
        if (frcnn_conv2d_dual_config(nullptr, 0) >= 0) { printf("dual_config accepted null\n"); ++failures; }
        if (frcnn_conv2d_x6_config(nullptr, 0) >= 0) { printf("x6_config accepted null\n"); ++failures; }
        frcnn_x6_job xj; xj.w_packed = &dummy; xj.planes_bf16 = &dummy; xj.rows = 4; xj.kpad = 40;
        if (frcnn_refresh_x6_planes(&xj, 1, nullptr) == FRCNN_OK) { printf("refresh_x6_planes accepted kpad %% 32 != 0\n"); ++failures; }
        if (frcnn_refresh_x6_planes(nullptr, 3, nullptr) == FRCNN_OK) { printf("refresh_x6_planes accepted a null table\n"); ++failures; }
        if (frcnn_refresh_x6_planes(nullptr, 0, nullptr) != FRCNN_OK) { printf("refresh_x6_planes refused an empty table\n"); ++failures; }
    }
    checked += 8;
    printf("host sanitizer driver: %d checks, %d failures\n", checked, failures);
    return failures ? 1 : 0;
}
