// bf16 conv-engine lab: csrc/conv_bf16.hip compiled into a standalone program with per-workgroup phase timestamps
// (FRCNN_LAB_STAMPS) on the direct-to-LDS forms.  Dev tool, GPU box only.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFRCNN_LAB_STAMPS -Iinclude scripts/micro/bf16_stamps.hip -o scripts/micro/_bin/bf16_stamps
//   bf16_stamps <rows> <cin> <cout> <tile> <residual 0/1>      a 1x1 layer over <rows> rows (as (1, rows, 1, cin))
#include "../../faster_rcnn_amd/csrc/conv_bf16.hip"
#include <algorithm>
#include <map>
#include <vector>
#include <string.h>

namespace frcnn {
static char g_err[512];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
}
extern "C" const char* frcnn_last_error(void) { return frcnn::g_err; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill_b(__bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (__bf16)(((int)(h & 0xffff) - 32768) * (scale / 32768.0f));
    }
}
static __bf16* dev_rand(size_t n, unsigned seed, float scale) {
    __bf16* p; CK(hipMalloc(&p, n * 2));
    k_fill_b<<<1024, 256>>>(p, n, seed, scale);
    return p;
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: bf16_stamps rows cin cout tile residual\n"); return 2; }
    const int rows = atoi(argv[1]), cin = atoi(argv[2]), cout = atoi(argv[3]), tile = atoi(argv[4]), res = atoi(argv[5]);
    frcnn_conv_desc d; memset(&d, 0, sizeof d);
    d.n = 1; d.h = rows; d.w = 1; d.cin = cin; d.cout = cout; d.kh = d.kw = 1; d.stride = 1; d.ho = rows; d.wo = 1; d.act = 1; d.tile = tile;
    __bf16* x = dev_rand((size_t)rows * cin, 1, 1.0f);
    __bf16* w = dev_rand((size_t)cout * cin, 2, 0.05f);
    __bf16* r = res ? dev_rand((size_t)rows * cout, 3, 1.0f) : nullptr;
    __bf16* y; CK(hipMalloc(&y, (size_t)rows * cout * 2));
    hipStream_t s; CK(hipStreamCreate(&s));
    auto launch = [&]() {
        if (int e = frcnn_conv2d_fwd_bf16(&d, x, w, nullptr, nullptr, r, y, 0, s)) { fprintf(stderr, "launch: %d %s\n", e, frcnn_last_error()); exit(1); }
    };
    const size_t max_wg = 1 << 16;
    unsigned long long* st; CK(hipMalloc(&st, max_wg * 64));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipStreamSynchronize(s));
    {   // timing without stamps
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rows %d %d -> %d tile %d res %d: %.1f us per launch (10 back to back), %.1f TFLOP/s\n", rows, cin, cout, tile, res, ms * 100, 2.0 * rows * cin * cout / (ms * 100) / 1e6);
    }
    CK(hipMemset(st, 0, max_wg * 64));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(frcnn::g_lab_stamps_b), &st, sizeof st));
    launch();
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(max_wg * 8);
    CK(hipMemcpy(h.data(), st, max_wg * 64, hipMemcpyDeviceToHost));
    size_t nwg = 0; unsigned long long t0 = ~0ull, tend = 0;
    for (size_t b = 0; b < max_wg; ++b) if (h[b * 8]) { nwg = b + 1; t0 = std::min(t0, h[b * 8]); }
    std::vector<double> start, pro, loop, stage, issue, drain, life;
    struct Ev { double s, e; };
    std::map<unsigned, std::vector<Ev>> per_cu;
    for (size_t b = 0; b < nwg; ++b) {
        const unsigned long long* q = &h[b * 8];
        if (!q[0] || !q[5]) continue;
        tend = std::max(tend, q[5]);
        start.push_back((q[0] - t0) * 0.01); pro.push_back((q[1] - q[0]) * 0.01); loop.push_back((q[2] - q[1]) * 0.01);
        stage.push_back((q[3] - q[2]) * 0.01); issue.push_back((q[4] - q[3]) * 0.01); drain.push_back((q[5] - q[4]) * 0.01);
        life.push_back((q[5] - q[0]) * 0.01);
        const unsigned hw = (unsigned)q[7], xcc = (unsigned)(q[7] >> 32);
        per_cu[((xcc & 15) << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15)].push_back({(q[0] - t0) * 0.01, (q[5] - t0) * 0.01});
    }
    auto pct = [](std::vector<double>& v, double f) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("%zu workgroups on %zu CUs, first start -> last end %.1f us\n", nwg, per_cu.size(), (tend - t0) * 0.01);
    auto row = [&](const char* nm, std::vector<double>& v) { printf("  %-44s n=%6zu  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", nm, v.size(), pct(v, .1), pct(v, .5), pct(v, .9), pct(v, 1.0)); };
    row("entry -> first chunk landed (prologue)", pro); row("main loop", loop); row("accumulators -> LDS tile (+ barrier)", stage);
    row("row pieces: LDS reads, residual add, stores issued", issue); row("store drain (vmcnt 0)", drain); row("workgroup lifetime", life);
    // per CU: how many workgroups are resident over time, and the gap between an exit and the next entry
    std::vector<double> gaps, resident;
    double busy2 = 0, busy1 = 0, span = 0;
    for (auto& kv : per_cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [](const Ev& a, const Ev& b) { return a.s < b.s; });
        std::vector<std::pair<double, int>> ev;
        for (auto& e : v) { ev.push_back({e.s, +1}); ev.push_back({e.e, -1}); }
        std::sort(ev.begin(), ev.end());
        int live = 0; double last = ev.front().first;
        for (auto& e : ev) {
            const double dt = e.first - last;
            if (live >= 2) busy2 += dt; else if (live == 1) busy1 += dt;
            last = e.first; live += e.second;
        }
        span += ev.back().first - ev.front().first;
        // exit -> next entry: for every exit, the first entry at or after it
        std::vector<double> starts; for (auto& e : v) starts.push_back(e.s);
        for (auto& e : v) {
            auto it = std::lower_bound(starts.begin(), starts.end(), e.e);
            if (it != starts.end()) gaps.push_back(*it - e.e);
        }
        resident.push_back((double)v.size());
    }
    printf("  per CU: two or more workgroups resident %.1f %% of its span, one %.1f %%, none %.1f %%\n", 100 * busy2 / span, 100 * busy1 / span, 100 * (span - busy1 - busy2) / span);
    row("exit -> next entry on the same CU", gaps); row("workgroups per CU", resident);
    return 0;
}
