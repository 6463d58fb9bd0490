// Score ordering (top-K) and greedy NMS for gfx950.
//
// top-K: rank-by-counting over packed 64-bit keys (score bits | ~index) -- every key is
//        unique, so ranks are a permutation and the result is deterministic.  A 16-bit
//        histogram first finds the bin of the K-th key, so only the ~K candidates at or above
//        it are ranked against each other (C^2 compares, not N^2).
// NMS:   (1) all-pairs suppression bit matrix, upper triangle only, one 64x64 tile per
//        wave64 (a lane's 64 pair tests become one 64-bit word = one coalesced store);
//        (2) a single-wave sequential scan that keeps the `removed` bitmap in registers,
//        resolves each 64-candidate chunk with scalar bit tricks + v_readlane, ORs the kept
//        rows' words in with independent (pipelined) loads and stops at max_boxes exactly
//        like the reference loop does.
// Compiled with -ffp-contract=off (the f64 overlap ratio must round like numpy's).
#include "common.h"
#include <stdlib.h>

namespace frcnn {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned mono_f32(float f) {      // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ------------------------------------------------------------------------------------ top-K
// Exact top-K in four small passes: (1) histogram of the keys' top 16 bits (sign, exponent, 7 mantissa bits of the
// score), (2) one workgroup finds the bin b holding the K-th largest key, (3) keys in bins >= b are compacted
// into a candidate list (K <= C, typically C - K = one bin's population), (4) rank-by-counting among the
// candidates only: every non-candidate is smaller than every candidate, so a candidate's rank among the
// candidates IS its global rank.  C^2 compares instead of N^2 (64 296 anchors, K = 8000: 64x fewer); a
// degenerate score distribution (everything in one bin) degrades to the N^2 sweep, never to a wrong answer.
constexpr int TOPK_BINS = 65536;
constexpr int TOPK_BLOCK = 256;
struct TopkCtrl { int32_t n_cand; int32_t thr_bin; int32_t n_valid; int32_t pad[61]; };
constexpr int TOPK_DIRECT_N = 32768;        // at or below this many scores the plain all-pairs sweep is faster than the four passes

__device__ __forceinline__ u64 topk_key(float score, int i) { return (((u64)mono_f32(score)) << 32) | (unsigned)(~(unsigned)i); }

// RPN scores cluster (sigmoid outputs around one value): straight global atomics pile ~1600 adds onto each of a
// few dozen addresses (60 us at 64 296 anchors).  Each workgroup therefore counts into an LDS window of 4096
// bins starting at its smallest bin (32 binades: practically always wide enough) and flushes only the non-empty
// bins; a wider spread falls back to direct global adds.
constexpr int TOPK_WIN = 4096;
__global__ void __launch_bounds__(TOPK_BLOCK) k_topk_hist(const float* scores, const uint8_t* valid, int N, unsigned* hist, TopkCtrl* ctrl) {
    __shared__ unsigned win[TOPK_WIN];
    __shared__ int s_lo, s_hi;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool v = i < N && (!valid || valid[i]);
    const int bin = v ? (int)(mono_f32(scores[i]) >> 16) : -1;
    if (threadIdx.x == 0) { s_lo = TOPK_BINS; s_hi = -1; }
    for (int j = threadIdx.x; j < TOPK_WIN; j += TOPK_BLOCK) win[j] = 0u;
    __syncthreads();
    if (v) { atomicMin(&s_lo, bin); atomicMax(&s_hi, bin); }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    if (hi - lo < TOPK_WIN) {
        if (v) atomicAdd(&win[bin - lo], 1u);
        __syncthreads();
        for (int j = threadIdx.x; j <= hi - lo; j += TOPK_BLOCK) { const unsigned c = win[j]; if (c) atomicAdd(&hist[lo + j], c); }
    } else if (v) {
        atomicAdd(&hist[bin], 1u);
    }
    const u64 b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&ctrl->n_valid, __popcll(b));
}

// thread t owns the 64 bins [65536 - 64(t+1), 65536 - 64t): bins are walked from the largest key down
__global__ void __launch_bounds__(1024) k_topk_threshold(const unsigned* hist, int K, TopkCtrl* ctrl) {
    __shared__ unsigned tot[1024];
    const int t = threadIdx.x;
    const unsigned* mine = hist + (TOPK_BINS - 64 * (t + 1));
    unsigned sum = 0;
    const uint4* mine4 = reinterpret_cast<const uint4*>(mine);
#pragma unroll
    for (int b = 0; b < 16; ++b) { const uint4 v = mine4[b]; sum += (v.x + v.y) + (v.z + v.w); }
    tot[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {              // inclusive scan over t (t = 0 is the top of the key range)
        const unsigned add = t >= off ? tot[t - off] : 0u;
        __syncthreads();
        tot[t] += add;
        __syncthreads();
    }
    const int nv = ctrl->n_valid;
    const unsigned need = (unsigned)(nv < K ? nv : K);
    if (t == 0 && need == 0) ctrl->thr_bin = TOPK_BINS;     // nothing valid: no candidates
    const unsigned before = t ? tot[t - 1] : 0u;
    if (need > 0 && before < need && tot[t] >= need) {      // the K-th largest key lies in one of my bins
        unsigned acc = before;
        for (int b = 63; b >= 0; --b) {
            acc += mine[b];
            if (acc >= need) { ctrl->thr_bin = TOPK_BINS - 64 * (t + 1) + b; break; }
        }
    }
}

// direct != 0: no histogram ran -- every valid key is a candidate and this pass also counts them
__global__ void k_topk_compact(const float* scores, const uint8_t* valid, int N, TopkCtrl* ctrl, u64* cand, int32_t* rank, int direct) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool take = false;
    float sc = 0.0f;
    if (i < N && (!valid || valid[i])) { sc = scores[i]; take = direct || (int)(mono_f32(sc) >> 16) >= ctrl->thr_bin; }
    const u64 b = __ballot(take);
    if (!b) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long)b) - 1) {
        base = atomicAdd(&ctrl->n_cand, __popcll(b));     // one atomic per wave
        if (direct) atomicAdd(&ctrl->n_valid, __popcll(b));
    }
    base = __shfl(base, __ffsll((long long)b) - 1);
    if (take) {
        const int pos = base + __popcll(b & ((1ull << lane) - 1ull));
        cand[pos] = topk_key(sc, i);
        rank[pos] = 0;
    }
}

// 2-D decomposition: block (bi, bj) counts, for its 256 candidates i, how many candidates of j-slice bj are
// larger, and adds the partial rank with one atomic per key.  The grid is sized for N (static under hipGraph
// capture); blocks beyond the device-side candidate count exit at once.
__global__ void __launch_bounds__(TOPK_BLOCK) k_topk_rank(const u64* keys, const TopkCtrl* ctrl, int32_t* rank) {
    __shared__ u64 tile[TOPK_BLOCK];
    const int n = ctrl->n_cand;
    if ((int)blockIdx.x * TOPK_BLOCK >= n) return;
    const int slice = ((n + (int)gridDim.y - 1) / (int)gridDim.y + TOPK_BLOCK - 1) / TOPK_BLOCK * TOPK_BLOCK;
    const int i = blockIdx.x * TOPK_BLOCK + threadIdx.x;
    const u64 mine = i < n ? keys[i] : ~0ull;
    const int j_begin = blockIdx.y * slice, j_end = min(n, j_begin + slice);
    int cnt = 0;
    for (int t0 = j_begin; t0 < j_end; t0 += TOPK_BLOCK) {
        const int j = t0 + threadIdx.x;
        tile[threadIdx.x] = j < j_end ? keys[j] : 0ull;
        __syncthreads();
#pragma unroll 16
        for (int jj = 0; jj < TOPK_BLOCK; ++jj) cnt += tile[jj] > mine;       // LDS broadcast reads
        __syncthreads();
    }
    if (i < n && cnt) atomicAdd(&rank[i], cnt);
}

__global__ void k_zero_u32(unsigned* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

__global__ void k_topk_scatter(const u64* keys, const int32_t* rank, const TopkCtrl* ctrl, int K, int32_t* order, int32_t* n_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < ctrl->n_cand && rank[c] < K) order[rank[c]] = (int32_t)(~(unsigned)keys[c]);
    const int n = min(ctrl->n_valid, K);
    if (c >= n && c < K) order[c] = -1;                     // the tail no rank lands on
    if (c == 0) *n_out = n;
}

__global__ void k_gather_candidates(const float4* rois, const float* scores, const int32_t* order, const int32_t* n, int K,
                                    short4* cand, float* cand_scores) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    short4 o = make_short4(0, 0, 0, 0);
    float s = 0.0f;
    if (k < *n) {
        const int i = order[k];
        const float4 r = rois[i];
        o = make_short4((short)r.x, (short)r.y, (short)r.z, (short)r.w);   // astype('int16') (det_util.py:76)
        s = scores[i];
    }
    cand[k] = o;
    if (cand_scores) cand_scores[k] = s;
}

// ------------------------------------------------------------------------------------ NMS
// overlap test of det_util.nms (det_util.py:237-250): "+1" convention; numpy evaluates the
// int16 case in int16 and divides int/int in f64; the float case is f64 throughout.
__device__ __forceinline__ bool suppresses(const short4 a, int area_a, const short4 b, int area_b, double thresh) {
    const int w = max(0, min((int)a.z, (int)b.z) - max((int)a.x, (int)b.x) + 1);
    const int h = max(0, min((int)a.w, (int)b.w) - max((int)a.y, (int)b.y) + 1);
    const int inter = w * h;
    const double overlap = (double)inter / (double)(area_a + area_b - inter);
    return !(overlap <= thresh);
}
__device__ __forceinline__ int box_area1(const short4 b) { return ((int)b.z - b.x + 1) * ((int)b.w - b.y + 1); }

__device__ __forceinline__ bool suppresses(const double4 a, double area_a, const double4 b, double area_b, double thresh) {
    const double w = fmax(0.0, fmin(a.z, b.z) - fmax(a.x, b.x) + 1.0);
    const double h = fmax(0.0, fmin(a.w, b.w) - fmax(a.y, b.y) + 1.0);
    const double inter = w * h;
    const double overlap = inter / (area_a + area_b - inter);
    return !(overlap <= thresh);
}
__device__ __forceinline__ double box_area1(const double4 b) { return (b.z - b.x + 1.0) * (b.w - b.y + 1.0); }

// One wave64 per 64x64 tile (row block rb <= column block cb).  Lane = row; its 64 tests
// against the column block form one u64.  Bit j of mask[i][cb] <=> box i suppresses box
// cb*64+j; only bits with column index > i are ever consumed.
template <typename Box4>
__global__ void __launch_bounds__(64) k_nms_mask(const Box4* boxes, const int32_t* n_ptr, int K, int W, double thresh, u64* mask) {
    const int n = min(*n_ptr, K);
    // linear tile id -> (rb, cb) of the upper triangle, row-major
    int tid = blockIdx.x;
    int rb = 0;
    while (tid >= W - rb) { tid -= W - rb; ++rb; }
    const int cb = rb + tid;
    if (rb * 64 >= n || cb * 64 >= n) return;
    __shared__ Box4 cols[64];
    const int lane = threadIdx.x;
    const int j0 = cb * 64;
    cols[lane] = j0 + lane < n ? boxes[j0 + lane] : boxes[0];
    __syncthreads();
    const int i = rb * 64 + lane;
    if (i >= n) return;
    const Box4 me = boxes[i];
    const auto my_area = box_area1(me);
    u64 bits = 0;
    const int jmax = min(64, n - j0);
    for (int j = 0; j < jmax; ++j) {
        const Box4 o = cols[j];
        if (j0 + j > i && suppresses(me, my_area, o, box_area1(o), thresh)) bits |= 1ull << j;
    }
    mask[(size_t)i * W + cb] = bits;
}

constexpr int NMS_SLOTS = FRCNN_NMS_MAX_BOXES / 64 / 64;     // removed-bitmap words per lane (3)

__device__ __forceinline__ u64 uniform_u64(u64 v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readlane_u64(u64 v, int lane /*uniform*/) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}

__global__ void __launch_bounds__(64) k_nms_scan(const u64* mask, const int32_t* n_ptr, int K, int W, int max_boxes,
                                                 int32_t* keep, int32_t* n_keep) {
    const int lane = threadIdx.x;
    const int n = min(*n_ptr, K);
    u64 removed[NMS_SLOTS];                 // lane l, slot s holds bitmap word s*64 + l
#pragma unroll
    for (int s = 0; s < NMS_SLOTS; ++s) removed[s] = 0;
    int total = 0;
    const int chunks = (n + 63) / 64;
    u64 diag_next = lane < n ? mask[(size_t)lane * W] : 0ull;          // chunk 0's diagonal word of row `lane`
    for (int c = 0; c < chunks && total < max_boxes; ++c) {
        const int base = c * 64;
        const int i = base + lane;
        const u64 diag = diag_next;
        if (c + 1 < chunks) {                 // the next chunk's diagonal words do not depend on this chunk's outcome:
            const int in = i + 64;            // fetch them now, their latency hides behind the resolve below
            diag_next = in < n ? mask[(size_t)in * W + c + 1] : 0ull;
        }
        u64 rem = 0;
#pragma unroll
        for (int s = 0; s < NMS_SLOTS; ++s)
            if ((c >> 6) == s) rem = readlane_u64(removed[s], c & 63);
        const int cnt = min(64, n - base);
        const u64 vmask = cnt == 64 ? ~0ull : ((1ull << cnt) - 1);
        u64 alive = uniform_u64(~rem & vmask);
        u64 kept = 0;
        int room = max_boxes - total;
        while (alive && room > 0) {          // wave-uniform: iterates over KEPT boxes only
            const int b = __builtin_ctzll(alive);
            kept |= 1ull << b;
            --room;
            alive &= ~(1ull << b);
            alive &= ~readlane_u64(diag, b);
            alive = uniform_u64(alive);
        }
        if ((kept >> lane) & 1) keep[total + __popcll(kept & ((1ull << lane) - 1))] = i;
        total += __popcll(kept);
        if (total >= max_boxes) break;
        // OR the kept rows into the bitmap for all later chunks, EIGHT rows per round: their 24 loads are
        // independent and in flight together (one row per round paid one memory latency per kept box)
        u64 k = kept;
        while (k) {
            u64 v[8][NMS_SLOTS];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool has = k != 0;
                const int b = has ? __builtin_ctzll(k) : 0;
                k = has ? k & (k - 1) : 0;
                const u64* row = mask + (size_t)(base + b) * W;
#pragma unroll
                for (int s = 0; s < NMS_SLOTS; ++s) {
                    const int w = s * 64 + lane;
                    v[u][s] = (has && w > c && w < W) ? row[w] : 0ull;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int s = 0; s < NMS_SLOTS; ++s) removed[s] |= v[u][s];
        }
    }
    for (int i = total + lane; i < max_boxes; i += 64) keep[i] = -1;
    if (lane == 0) *n_keep = total;
}

__global__ void k_gather_rois(const short4* cand, const int32_t* keep, const int32_t* n_keep, int batch, int out_rows, float4* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= out_rows) return;
    const int n = *n_keep;
    float4 o = make_float4(0, 0, 0, 0);
    if (n > 0) {
        const int padded = (n + batch - 1) / batch * batch;
        int src = k < n ? k : (k < padded ? (k / batch) * batch : 0);
        const short4 b = cand[keep[src]];
        o = make_float4((float)b.x, (float)b.y, (float)b.z, (float)b.w);
    }
    out[k] = o;
}

template <typename Box4>
static int nms_launch(const Box4* boxes, const int32_t* n, int K, double thresh, int max_boxes,
                      int32_t* keep, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream, const char* what) {
    if (K < 0 || K > FRCNN_NMS_MAX_BOXES) return fail(FRCNN_E_ARG, "%s: K=%d exceeds %d", what, K, FRCNN_NMS_MAX_BOXES);
    if (max_boxes <= 0 || !n || !keep || !n_keep) return fail(FRCNN_E_ARG, "%s: bad argument", what);
    hipStream_t s = as_stream(stream);
    if (K == 0) {
        if (hipMemsetAsync(n_keep, 0, 4, s) != hipSuccess || hipMemsetAsync(keep, 0xFF, (size_t)max_boxes * 4, s) != hipSuccess)
            return fail(FRCNN_E_HIP, "%s: memset failed", what);
        return FRCNN_OK;
    }
    if (!boxes) return fail(FRCNN_E_ARG, "%s: null boxes", what);
    if (!workspace || workspace_bytes < frcnn_nms_workspace_bytes(K))
        return fail(FRCNN_E_WORKSPACE, "%s: workspace needs %zu bytes", what, frcnn_nms_workspace_bytes(K));
    const int W = (K + 63) / 64;
    u64* mask = (u64*)workspace;
    const int tiles = W * (W + 1) / 2;
    k_nms_mask<Box4><<<tiles, 64, 0, s>>>(boxes, n, K, W, thresh, mask);
    if (int e = check_launch(what)) return e;
    k_nms_scan<<<1, 64, 0, s>>>(mask, n, K, W, max_boxes, keep, n_keep);
    return check_launch(what);
}

}  // namespace frcnn

using namespace frcnn;

extern "C" {

size_t frcnn_topk_workspace_bytes(int N) {
    const size_t n = (size_t)(N > 0 ? N : 1);
    return (size_t)TOPK_BINS * 4 + sizeof(TopkCtrl) + align_up(n * 8, 256) + align_up(n * 4, 256);
}

int frcnn_topk_order(const float* scores, const uint8_t* valid, int N, int K, int32_t* order, int32_t* n_out,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (N < 0 || K <= 0 || !order || !n_out) return fail(FRCNN_E_ARG, "topk_order: bad argument");
    hipStream_t s = as_stream(stream);
    if (N == 0) {
        if (hipMemsetAsync(n_out, 0, 4, s) != hipSuccess || hipMemsetAsync(order, 0xFF, (size_t)K * 4, s) != hipSuccess)
            return fail(FRCNN_E_HIP, "topk_order: memset failed");
        return FRCNN_OK;
    }
    if (!scores) return fail(FRCNN_E_ARG, "topk_order: null scores");
    if (!workspace || workspace_bytes < frcnn_topk_workspace_bytes(N))
        return fail(FRCNN_E_WORKSPACE, "topk_order: workspace needs %zu bytes", frcnn_topk_workspace_bytes(N));
    unsigned* hist = (unsigned*)workspace;
    TopkCtrl* ctrl = (TopkCtrl*)((char*)workspace + (size_t)TOPK_BINS * 4);
    u64* cand = (u64*)((char*)ctrl + sizeof(TopkCtrl));
    int32_t* rank = (int32_t*)((char*)cand + align_up((size_t)N * 8, 256));
    const int bi = (N + TOPK_BLOCK - 1) / TOPK_BLOCK;
    const int direct = N <= TOPK_DIRECT_N;
    // The counters are cleared by a kernel, not hipMemsetAsync: captured into a hipGraph, a lone memset node between
    // two kernel nodes was observed (ROCm 7.2, gfx950) not to be ordered against its neighbours -- k_topk_compact
    // then appended behind the previous replay's count and ran off the end of the candidate list.
    if (direct) {
        k_zero_u32<<<1, 64, 0, s>>>((unsigned*)ctrl, (int)(sizeof(TopkCtrl) / 4));
        if (int e = check_launch("topk_order reset")) return e;
    } else {
        const int words = TOPK_BINS + (int)(sizeof(TopkCtrl) / 4);
        k_zero_u32<<<(words + 1023) / 1024, 1024, 0, s>>>(hist, words);
        if (int e = check_launch("topk_order reset")) return e;
        k_topk_hist<<<bi, TOPK_BLOCK, 0, s>>>(scores, valid, N, hist, ctrl);
        if (int e = check_launch("topk_order hist")) return e;
        k_topk_threshold<<<1, 1024, 0, s>>>(hist, K, ctrl);
        if (int e = check_launch("topk_order threshold")) return e;
    }
    k_topk_compact<<<bi, TOPK_BLOCK, 0, s>>>(scores, valid, N, ctrl, cand, rank, direct);
    if (int e = check_launch("topk_order compact")) return e;
    int split = 2048 / bi;
    split = split < 8 ? 8 : (split > 64 ? 64 : split);
    k_topk_rank<<<dim3(bi, split), TOPK_BLOCK, 0, s>>>(cand, ctrl, rank);
    if (int e = check_launch("topk_order rank")) return e;
    const int bs = ((N > K ? N : K) + TOPK_BLOCK - 1) / TOPK_BLOCK;     // covers every candidate and every output slot
    k_topk_scatter<<<bs, TOPK_BLOCK, 0, s>>>(cand, rank, ctrl, K, order, n_out);
    return check_launch("topk_order scatter");
}

int frcnn_gather_candidates(const float* rois, const float* scores, const int32_t* order, const int32_t* n, int K,
                            int16_t* cand, float* cand_scores, void* stream) {
    if (K <= 0 || !rois || !scores || !order || !n || !cand) return fail(FRCNN_E_ARG, "gather_candidates: bad argument");
    k_gather_candidates<<<(K + 255) / 256, 256, 0, as_stream(stream)>>>((const float4*)rois, scores, order, n, K, (short4*)cand, cand_scores);
    return check_launch("gather_candidates");
}

size_t frcnn_nms_workspace_bytes(int K) {
    const size_t W = (size_t)(K + 63) / 64;
    return align_up((size_t)(K > 0 ? K : 1) * W * 8, 256);
}

int frcnn_nms_i16(const int16_t* boxes, const int32_t* n, int K, double thresh, int max_boxes,
                  int32_t* keep, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream) {
    return nms_launch<short4>((const short4*)boxes, n, K, thresh, max_boxes, keep, n_keep, workspace, workspace_bytes, stream, "nms_i16");
}

int frcnn_nms_f64(const double* boxes, const int32_t* n, int K, double thresh, int max_boxes,
                  int32_t* keep, int32_t* n_keep, void* workspace, size_t workspace_bytes, void* stream) {
    return nms_launch<double4>((const double4*)boxes, n, K, thresh, max_boxes, keep, n_keep, workspace, workspace_bytes, stream, "nms_f64");
}

int frcnn_gather_rois(const int16_t* cand, const int32_t* keep, const int32_t* n_keep, int batch, int out_rows, float* out, void* stream) {
    if (batch <= 0 || out_rows <= 0 || !cand || !keep || !n_keep || !out) return fail(FRCNN_E_ARG, "gather_rois: bad argument");
    k_gather_rois<<<(out_rows + 255) / 256, 256, 0, as_stream(stream)>>>((const short4*)cand, keep, n_keep, batch, out_rows, (float4*)out);
    return check_launch("gather_rois");
}

}  // extern "C"
