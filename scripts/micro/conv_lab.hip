// Conv-engine lab: the product conv source compiled into a standalone program, timed without Python in the
// launch path, optionally with per-workgroup phase timestamps (FRCNN_LAB_STAMPS).  Dev tool, GPU box only.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/conv_lab.hip -o gpurun_out/conv_lab
//   hipcc ... -DFRCNN_LAB_STAMPS scripts/micro/conv_lab.hip -o gpurun_out/conv_lab_stamps
//   conv_lab time  <set> <tile,tile,...>        per-shape table, us per launch (20 launches in one hipGraph, best of 4)
//   conv_lab_stamps stamps <layer> <tile>       one launch with timestamps: phase medians + start/end distribution
// <set>: trunk | head | all
#include "../../faster_rcnn_amd/csrc/conv_igemm.hip"
#include <algorithm>
#include <string>
#include <vector>
#include <string.h>

namespace frcnn {
static char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
}
extern "C" const char* frcnn_last_error(void) { return frcnn::g_err; }

// lab-only variants (tile codes >= 1000) plug in here
static int frcnn_conv2d_trunk_lab(const frcnn_conv_desc*, int, const float*, const float*, const float*, const float*, const float*, float*, void*, size_t, hipStream_t) { return -1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.0f);
    }
}
static float* dev_rand(size_t n, unsigned seed, float scale) {
    float* p; CK(hipMalloc(&p, n * sizeof(float)));
    k_fill<<<1024, 256>>>(p, n, seed, scale);
    return p;
}

struct Shape { std::string name; int cnt, n, h, w, cin, cout, k, stride; bool same, res; int set; };   // set 0 trunk, 1 head

static std::vector<Shape> shapes(int rois) {
    std::vector<Shape> s;
    s.push_back({"conv1", 1, 1, 600, 1000, 3, 64, 7, 2, true, false, 0});
    auto stage = [&](const char* tag, int& h, int& w, int cin, int f1, int f3, int nb, int stride) {
        const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
        std::string t(tag);
        s.push_back({t + "a_2a", 1, 1, h, w, cin, f1, 1, stride, false, false, 0});
        s.push_back({t + "a_1", 1, 1, h, w, cin, f3, 1, stride, false, false, 0});
        s.push_back({t + "_2b", nb, 1, ho, wo, f1, f1, 3, 1, true, false, 0});
        s.push_back({t + "_2c", nb, 1, ho, wo, f1, f3, 1, 1, false, true, 0});
        s.push_back({t + "x_2a", nb - 1, 1, ho, wo, f3, f1, 1, 1, false, false, 0});
        h = ho; w = wo;
    };
    int h = 149, w = 249;
    stage("s2", h, w, 64, 64, 256, 3, 1);
    stage("s3", h, w, 256, 128, 512, 4, 2);
    stage("s4", h, w, 512, 256, 1024, 6, 2);
    s.push_back({"rpn_conv1", 1, 1, h, w, 1024, 512, 3, 1, true, false, 0});
    s.push_back({"s5a_2a", 1, rois, 7, 7, 1024, 512, 1, 1, false, false, 1});
    s.push_back({"s5_2b", 3, rois, 7, 7, 512, 512, 3, 1, true, false, 1});
    s.push_back({"s5_2c", 3, rois, 7, 7, 512, 2048, 1, 1, false, true, 1});
    s.push_back({"s5x_2a", 2, rois, 7, 7, 2048, 512, 1, 1, false, false, 1});
    // training-only launch shapes (set 2): input-gradient convs and the detector head at 64 sampled RoIs
    s.push_back({"rpn_dgrad", 1, 1, h, w, 512, 1024, 3, 1, true, true, 2});            // dgrad of rpn_conv1: 512 -> 1024, 3x3
    s.push_back({"t5_2b", 6, 64, 7, 7, 512, 512, 3, 1, true, false, 2});              // stage-5 3x3 fwd / dgrad
    s.push_back({"t5_2c", 5, 64, 7, 7, 512, 2048, 1, 1, false, true, 2});             // 512 -> 2048 (2c fwd, 2a dgrad)
    s.push_back({"t5x_2a", 5, 64, 7, 7, 2048, 512, 1, 1, false, false, 2});           // 2048 -> 512 (2a fwd, 2c dgrad)
    s.push_back({"t5a_2a", 1, 64, 7, 7, 1024, 512, 1, 1, false, false, 2});
    s.push_back({"t5a_1", 1, 64, 7, 7, 1024, 2048, 1, 1, false, false, 2});
    s.push_back({"t5a_dg", 2, 64, 7, 7, 2048, 1024, 1, 1, false, true, 2});           // dgrad into the crops (from 2a / branch1)
    return s;
}

struct Problem {
    frcnn_conv_desc d; float *x, *w, *scale, *shift, *res, *y, *yref; void* ws; size_t ws_bytes; size_t M; double flops;
};

static Problem make(const Shape& sh) {
    Problem p; memset(&p, 0, sizeof p);
    frcnn_conv_desc& d = p.d;
    d.n = sh.n; d.h = sh.h; d.w = sh.w; d.cin = sh.cin; d.cout = sh.cout; d.kh = d.kw = sh.k; d.stride = sh.stride;
    if (sh.same) {
        d.ho = (sh.h + sh.stride - 1) / sh.stride; d.wo = (sh.w + sh.stride - 1) / sh.stride;
        const int th = std::max((d.ho - 1) * sh.stride + sh.k - sh.h, 0), tw = std::max((d.wo - 1) * sh.stride + sh.k - sh.w, 0);
        d.pad_top = th / 2; d.pad_left = tw / 2;
    } else { d.ho = (sh.h - sh.k) / sh.stride + 1; d.wo = (sh.w - sh.k) / sh.stride + 1; }
    d.act = FRCNN_ACT_RELU;
    p.M = (size_t)d.n * d.ho * d.wo;
    p.flops = 2.0 * p.M * d.cout * sh.k * sh.k * sh.cin;
    p.x = dev_rand((size_t)d.n * d.h * d.w * d.cin, 1, 1.0f);
    float* hwio = dev_rand((size_t)sh.k * sh.k * sh.cin * sh.cout, 2, sqrtf(6.0f / (sh.k * sh.k * sh.cin)));
    const int kp = frcnn_conv_packed_k(sh.k, sh.k, sh.cin);
    CK(hipMalloc(&p.w, (size_t)sh.cout * kp * 4));
    frcnn_pack_conv_weights(hwio, sh.k, sh.k, sh.cin, sh.cout, p.w, nullptr);
    p.scale = dev_rand(sh.cout, 3, 0.2f);
    p.shift = dev_rand(sh.cout, 4, 0.2f);
    p.res = sh.res ? dev_rand(p.M * sh.cout, 5, 1.0f) : nullptr;
    CK(hipMalloc(&p.y, p.M * sh.cout * 4)); CK(hipMalloc(&p.yref, p.M * sh.cout * 4));
    p.ws_bytes = 256u << 20;
    CK(hipMalloc(&p.ws, p.ws_bytes)); CK(hipMemset(p.ws, 0, p.ws_bytes));
    CK(hipDeviceSynchronize()); CK(hipFree(hwio));
    return p;
}
static void destroy(Problem& p) {
    for (void* q : {(void*)p.x, (void*)p.w, (void*)p.scale, (void*)p.shift, (void*)p.res, (void*)p.y, (void*)p.yref, p.ws}) if (q) (void)hipFree(q);
}

// tile codes < 1000: the product entry point; >= 1000: conv_trunk.hip variants (code - 1000)
static int launch(const Problem& p, int tile, float* y, hipStream_t s) {
    frcnn_conv_desc d = p.d;
    if (tile >= 1000) return frcnn_conv2d_trunk_lab(&d, tile - 1000, p.x, p.w, p.scale, p.shift, p.res, y, p.ws, p.ws_bytes, s);
    d.tile = tile;
    return frcnn_conv2d_fwd_ws(&d, p.x, p.w, p.scale, p.shift, p.res, nullptr, y, p.ws, p.ws_bytes, s);
}

static double time_graph(const Problem& p, int tile, hipStream_t s, int reps = 20) {
    if (launch(p, tile, p.y, s) != 0) return -1.0;
    CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; ++i) launch(p, tile, p.y, s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return ms * 1e3 / reps;
}

static void compare(const Problem& p, size_t n, double* max_abs, size_t* n_diff) {
    std::vector<float> a(n), b(n);
    CK(hipMemcpy(a.data(), p.y, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), p.yref, n * 4, hipMemcpyDeviceToHost));
    double m = 0; size_t nd = 0;
    for (size_t i = 0; i < n; ++i) { if (memcmp(&a[i], &b[i], 4)) ++nd; const double e = fabs((double)a[i] - b[i]); if (!(e <= m)) m = e; }
    *max_abs = m; *n_diff = nd;
}

static int cmd_time(int argc, char** argv) {
    const std::string set = argc > 2 ? argv[2] : "trunk";
    std::vector<int> tiles;
    { std::string t = argc > 3 ? argv[3] : "0"; size_t pos = 0; while (pos < t.size()) { size_t c = t.find(',', pos); if (c == std::string::npos) c = t.size(); tiles.push_back(atoi(t.substr(pos, c - pos).c_str())); pos = c + 1; } }
    const char* only = argc > 4 ? argv[4] : nullptr;
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("%-10s %3s %7s %5s %6s %8s |", "layer", "cnt", "M", "N", "K", "GFLOP");
    for (int t : tiles) printf("  t%-5d us / TF  ", t);
    printf("\n");
    std::vector<double> tot(tiles.size(), 0.0); double best_tot = 0, fl_tot = 0;
    for (const Shape& sh : shapes(300)) {
        if (set == "trunk" && sh.set != 0) continue;
        if (set == "head" && sh.set != 1) continue;
        if (set == "train" && sh.set != 2) continue;
        if (set != "train" && set != "all" && sh.set == 2) continue;
        if (only && sh.name != only) continue;
        Problem p = make(sh);
        std::vector<double> best(tiles.size(), -1.0);
        std::vector<std::string> note(tiles.size());
        // reference output: the product's auto choice with the plain 4-byte epilogue
        g_scalar_epilogue = true;
        launch(p, 0, p.yref, s); CK(hipStreamSynchronize(s));
        g_scalar_epilogue = getenv("FRCNN_SCALAR_EPILOGUE") != nullptr;
        for (size_t i = 0; i < tiles.size(); ++i) {
            CK(hipMemsetAsync(p.y, 0xff, p.M * sh.cout * 4, s));
            if (launch(p, tiles[i], p.y, s) != 0) { note[i] = "refused"; continue; }
            CK(hipStreamSynchronize(s));
            double ma; size_t nd; compare(p, p.M * sh.cout, &ma, &nd);
            char buf[64]; snprintf(buf, sizeof buf, nd ? "d%zu/%.1e" : "=", nd, ma); note[i] = buf;
        }
        for (int round = 0; round < 4; ++round)
            for (size_t i = 0; i < tiles.size(); ++i) {
                if (note[i] == "refused") continue;
                const double us = time_graph(p, tiles[i], s);
                if (us > 0 && (best[i] < 0 || us < best[i])) best[i] = us;
            }
        printf("%-10s %3d %7zu %5d %6d %8.3f |", sh.name.c_str(), sh.cnt, p.M, sh.cout, sh.k * sh.k * sh.cin, p.flops / 1e9);
        double mn = 1e30;
        for (size_t i = 0; i < tiles.size(); ++i) {
            if (best[i] < 0) { printf("  %-20s", "   -"); continue; }
            printf("  %6.1f/%5.1f %-7s", best[i], p.flops / best[i] / 1e6, note[i].c_str());
            tot[i] += best[i] * sh.cnt; mn = std::min(mn, best[i]);
        }
        printf("\n"); fflush(stdout);
        best_tot += mn * sh.cnt; fl_tot += p.flops * sh.cnt;
        destroy(p);
    }
    printf("total GFLOP %.1f; per-code totals (us):", fl_tot / 1e9);
    for (size_t i = 0; i < tiles.size(); ++i) printf(" t%d=%.0f", tiles[i], tot[i]);
    printf("; best-of %.0f us -> %.1f TF/s\n", best_tot, fl_tot / best_tot / 1e6);
    return 0;
}

#ifdef FRCNN_LAB_STAMPS
static int cmd_stamps(int argc, char** argv) {
    const std::string layer = argc > 2 ? argv[2] : "s4_2c";
    const int tile = argc > 3 ? atoi(argv[3]) : 0;
    hipStream_t s; CK(hipStreamCreate(&s));
    for (const Shape& sh : shapes(300)) {
        if (sh.name != layer) continue;
        Problem p = make(sh);
        const size_t max_wg = 1 << 16;
        unsigned long long* st; CK(hipMalloc(&st, max_wg * 64));
        for (int warm = 0; warm < 3; ++warm) launch(p, tile, p.y, s);
        CK(hipStreamSynchronize(s));
        // run behind a neighbour launch of the same kernel so the launch boundary is in the picture
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(st, 0, max_wg * 64));
            unsigned long long* nul = nullptr;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(frcnn::g_lab_stamps), &nul, sizeof nul));
            launch(p, tile, p.y, s); launch(p, tile, p.y, s);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(frcnn::g_lab_stamps), &st, sizeof st));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, s)); launch(p, tile, p.y, s); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(max_wg * 8);
            CK(hipMemcpy(h.data(), st, max_wg * 64, hipMemcpyDeviceToHost));
            size_t nwg = 0; unsigned long long t0 = ~0ull, tend = 0;
            for (size_t b = 0; b < max_wg; ++b) if (h[b * 8]) { nwg = b + 1; t0 = std::min(t0, h[b * 8]); }
            std::vector<double> start, pro, loop, sk, epi, drain, life, endt;
            std::vector<int> per_cu(8 * 64, 0);
            for (size_t b = 0; b < nwg; ++b) {
                const unsigned long long* q = &h[b * 8];
                if (!q[0]) continue;
                const unsigned long long last = q[6] ? q[6] : (q[3] ? q[3] : q[2]);
                tend = std::max(tend, last);
                start.push_back((q[0] - t0) * 0.01); pro.push_back((q[1] - q[0]) * 0.01); loop.push_back((q[2] - q[1]) * 0.01);
                if (q[3]) sk.push_back((q[3] - q[2]) * 0.01);
                if (q[6]) { const unsigned long long eb = q[4] ? q[4] : (q[3] ? q[3] : q[2]); epi.push_back((q[5] - eb) * 0.01); drain.push_back((q[6] - q[5]) * 0.01); }
                life.push_back((last - q[0]) * 0.01); endt.push_back((last - t0) * 0.01);
                const unsigned hw = (unsigned)q[7], xcc = (unsigned)(q[7] >> 32);
                if (q[6]) per_cu[(xcc & 7) * 64 + (((hw >> 13) & 7) * 16 + ((hw >> 8) & 15)) % 64]++;
            }
            auto pct = [](std::vector<double>& v, double f) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
            printf("%s tile %d rep %d: %zu workgroups, event time %.1f us, first start -> last end %.1f us\n", layer.c_str(), tile, rep, nwg, ms * 1e3, (tend - t0) * 0.01);
            auto row = [&](const char* nm, std::vector<double>& v) { printf("  %-28s n=%5zu  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", nm, v.size(), pct(v, .1), pct(v, .5), pct(v, .9), pct(v, 1.0)); };
            row("start after first start", start); row("prologue (entry->1st barrier)", pro); row("main loop", loop); row("split-K publish+ticket", sk);
            row("epilogue issue", epi); row("store drain", drain); row("workgroup lifetime", life); row("end after first start", endt);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        (void)hipFree(st); destroy(p);
    }
    return 0;
}
#endif

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: conv_lab time|stamps ...\n"); return 2; }
    if (!strcmp(argv[1], "time")) return cmd_time(argc, argv);
#ifdef FRCNN_LAB_STAMPS
    if (!strcmp(argv[1], "stamps")) return cmd_stamps(argc, argv);
#endif
    fprintf(stderr, "unknown command %s\n", argv[1]);
    return 2;
}
