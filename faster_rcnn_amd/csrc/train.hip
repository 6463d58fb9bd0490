// Training-side kernels for gfx950: the four reference losses (value + gradient), the small
// elementwise backward pieces around the conv engine, SGD-momentum / Adam and the L2 penalty.
// All HBM / latency bound.  Loss reductions run in ONE workgroup with f64 accumulation so that
// loss values and gradients are bitwise reproducible run to run (no float atomics).
//
// Loss semantics = loss_functions.py:15-76 evaluated by Keras 2.0.8 on the TF backend
// (SURVEY Appendix A.7), including the reference's quirks:
//   * N_CLS = 256 and N_REG = 2400 are constants, not the actual sample counts (:8-9);
//   * bbreg_loss_rpn multiplies the mask OUTSIDE K.sum (:44): the result is a tensor that Keras
//     then averages, i.e. loss = mean(mask) * 10 * S / 2400 with S summed over ALL anchors;
//   * bbreg_loss_det divides by sum(1e-4 + mask) (:65); cls_loss_det uses batch element 0 (:76).
// Keras clips probabilities to [1e-7, 1 - 1e-7] before the logarithm: a clipped probability has
// zero gradient.
#include "common.h"

namespace frcnn {

constexpr int LB = 1024;
constexpr float KERAS_EPS = 1e-7f;

__device__ __forceinline__ double block_sum(double v, double* scratch) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += scratch[w];      // fixed order
    return t;
}

__device__ __forceinline__ float smooth_l1(float d) { const float a = fabsf(d); return a <= 1.0f ? 0.5f * a * a : a - 0.5f; }
__device__ __forceinline__ float smooth_l1_grad(float d) { return fabsf(d) <= 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f); }

// cls_loss_rpn (loss_functions.py:15-28).  y_true [cells][2A] = [can_use | is_pos], p [cells][A] sigmoid
// outputs.  loss = sum(sel * BCE(is_pos, p)) / 256; g_logit = sel * (p - z) / 256 (0 where p was clipped).
__global__ void __launch_bounds__(LB) k_loss_rpn_cls(const float* y_true, const float* p, int cells, int A, float* loss, float* g_logit) {
    __shared__ double scratch[LB / 64];
    double acc = 0.0;
    const int n = cells * A;
#pragma unroll 4
    for (int i = threadIdx.x; i < n; i += LB) {         // unrolled: the loads of four iterations are in flight together
        const int cell = i / A, a = i % A;
        const float sel = y_true[(size_t)cell * 2 * A + a], z = y_true[(size_t)cell * 2 * A + A + a];
        const float pr = p[i];
        const float pc = fminf(fmaxf(pr, KERAS_EPS), 1.0f - KERAS_EPS);
        const float x = logf(pc / (1.0f - pc));                       // Keras goes back to logits
        const float bce = fmaxf(x, 0.0f) - x * z + log1pf(expf(-fabsf(x)));
        acc += (double)(sel * bce);
        const bool clipped = pr < KERAS_EPS || pr > 1.0f - KERAS_EPS;
        if (g_logit) g_logit[i] = clipped ? 0.0f : sel * (pc - z) / 256.0f;
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) *loss = (float)(t / 256.0);
}

// bbreg_loss_rpn (loss_functions.py:31-48).  y_true [cells][8A] = [mask(4A) | target(4A)], pred [cells][4A].
__global__ void __launch_bounds__(LB) k_loss_rpn_reg(const float* y_true, const float* pred, int cells, int A4, float* loss, float* g_pred) {
    __shared__ double scratch[LB / 64];
    double s = 0.0, msum = 0.0;
    const int n = cells * A4;
#pragma unroll 8
    for (int i = threadIdx.x; i < n; i += LB) {
        const int cell = i / A4, k = i % A4;
        msum += (double)y_true[(size_t)cell * 2 * A4 + k];
        s += (double)smooth_l1(y_true[(size_t)cell * 2 * A4 + A4 + k] - pred[i]);
    }
    const double S = block_sum(s, scratch);
    const double Msum = block_sum(msum, scratch);
    const double mean_mask = Msum / (double)n;
    if (threadIdx.x == 0) *loss = (float)(mean_mask * 10.0 * S / 2400.0);
    if (g_pred) {
        const float coef = (float)(mean_mask * 10.0 / 2400.0);
#pragma unroll 8
        for (int i = threadIdx.x; i < n; i += LB) {
            const int cell = i / A4, k = i % A4;
            g_pred[i] = -coef * smooth_l1_grad(y_true[(size_t)cell * 2 * A4 + A4 + k] - pred[i]);
        }
    }
}

// The two RPN losses over MANY workgroups (round 2): in one workgroup they cost 30 + 56 us of a 1.7-2.7 ms step, a
// latency-bound walk over 21 546 / 86 184 anchors.  Stage 1: LOSS_WGS workgroups take contiguous runs of anchors, write
// their gradients (classification) and an f64 partial per workgroup into the caller's workspace; stage 2 sums the
// partials IN INDEX ORDER (reproducible, no atomics) -- for the regression loss every stage-2 workgroup does so first,
// because its gradient needs mean(mask), then writes its run of the gradient.
constexpr int LOSS_WGS = 64, LOSS_T = 256;

__global__ void __launch_bounds__(LOSS_T) k_loss_rpn_cls_part(const float* y_true, const float* p, int cells, int A, double* part, float* g_logit) {
    __shared__ double scratch[LOSS_T / 64];
    const int n = cells * A, run = (n + LOSS_WGS - 1) / LOSS_WGS;
    const int i0 = blockIdx.x * run, i1 = min(n, i0 + run);
    double acc = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += LOSS_T) {
        const int cell = i / A, a = i % A;
        const float sel = y_true[(size_t)cell * 2 * A + a], z = y_true[(size_t)cell * 2 * A + A + a];
        const float pr = p[i];
        const float pc = fminf(fmaxf(pr, KERAS_EPS), 1.0f - KERAS_EPS);
        const float x = logf(pc / (1.0f - pc));
        const float bce = fmaxf(x, 0.0f) - x * z + log1pf(expf(-fabsf(x)));
        acc += (double)(sel * bce);
        const bool clipped = pr < KERAS_EPS || pr > 1.0f - KERAS_EPS;
        if (g_logit) g_logit[i] = clipped ? 0.0f : sel * (pc - z) / 256.0f;
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void k_loss_rpn_cls_final(const double* part, float* loss) {
    double t = 0.0;
    for (int w = 0; w < LOSS_WGS; ++w) t += part[w];
    *loss = (float)(t / 256.0);
}
__global__ void __launch_bounds__(LOSS_T) k_loss_rpn_reg_part(const float* y_true, const float* pred, int cells, int A4, double* part) {
    __shared__ double scratch[LOSS_T / 64];
    const int n = cells * A4, run = (n + LOSS_WGS - 1) / LOSS_WGS;
    const int i0 = blockIdx.x * run, i1 = min(n, i0 + run);
    double s = 0.0, msum = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += LOSS_T) {
        const int cell = i / A4, k = i % A4;
        msum += (double)y_true[(size_t)cell * 2 * A4 + k];
        s += (double)smooth_l1(y_true[(size_t)cell * 2 * A4 + A4 + k] - pred[i]);
    }
    const double S = block_sum(s, scratch);
    const double Msum = block_sum(msum, scratch);
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = S; part[2 * blockIdx.x + 1] = Msum; }
}
__global__ void __launch_bounds__(LOSS_T) k_loss_rpn_reg_final(const float* y_true, const float* pred, int cells, int A4, const double* part, float* loss, float* g_pred) {
    const int n = cells * A4, run = (n + LOSS_WGS - 1) / LOSS_WGS;
    double S = 0.0, Msum = 0.0;
    for (int w = 0; w < LOSS_WGS; ++w) { S += part[2 * w]; Msum += part[2 * w + 1]; }
    const double mean_mask = Msum / (double)n;
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(mean_mask * 10.0 * S / 2400.0);
    if (!g_pred) return;
    const float coef = (float)(mean_mask * 10.0 / 2400.0);
    const int i0 = blockIdx.x * run, i1 = min(n, i0 + run);
    for (int i = i0 + threadIdx.x; i < i1; i += LOSS_T) {
        const int cell = i / A4, k = i % A4;
        g_pred[i] = -coef * smooth_l1_grad(y_true[(size_t)cell * 2 * A4 + A4 + k] - pred[i]);
    }
}

// cls_loss_det (loss_functions.py:70-76): mean over RoIs of -sum_c y*log(clip(p/sum p)).  g w.r.t. the
// pre-softmax logits = (p - y)/n (the renormalisation is the identity on a softmax output).
__global__ void __launch_bounds__(LB) k_loss_det_cls(const float* y_true, const float* p, int n_rois, int C, float* loss, float* g_logit, int ldg) {
    __shared__ double scratch[LB / 64];
    double acc = 0.0;
    for (int r = threadIdx.x; r < n_rois; r += LB) {
        float sum = 0.0f;
        for (int c = 0; c < C; ++c) sum += p[(size_t)r * C + c];
        double l = 0.0;
        bool true_clipped = false;
        for (int c = 0; c < C; ++c) {
            const float y = y_true[(size_t)r * C + c];
            if (y != 0.0f) {
                const float q = p[(size_t)r * C + c] / sum;
                true_clipped |= q < KERAS_EPS || q > 1.0f - KERAS_EPS;
                l -= (double)(y * logf(fminf(fmaxf(q, KERAS_EPS), 1.0f - KERAS_EPS)));
            }
        }
        acc += l;
        if (g_logit)
            for (int c = 0; c < C; ++c)
                g_logit[(size_t)r * ldg + c] = true_clipped ? 0.0f : (p[(size_t)r * C + c] - y_true[(size_t)r * C + c]) / (float)n_rois;
    }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) *loss = (float)(t / (double)n_rois);
}

// bbreg_loss_det (loss_functions.py:51-67).  y_true [n][8K] = [mask(4K) | target(4K)], pred [n][4K].
__global__ void __launch_bounds__(LB) k_loss_det_reg(const float* y_true, const float* pred, int n_rois, int K4, float* loss, float* g_pred, int ldg) {
    __shared__ double scratch[LB / 64];
    double num = 0.0, den = 0.0;
    const int n = n_rois * K4;
    for (int i = threadIdx.x; i < n; i += LB) {
        const int r = i / K4, k = i % K4;
        const float m = y_true[(size_t)r * 2 * K4 + k];
        num += (double)(m * smooth_l1(y_true[(size_t)r * 2 * K4 + K4 + k] - pred[i]));
        den += (double)(1e-4f + m);
    }
    const double Num = block_sum(num, scratch);
    const double Den = block_sum(den, scratch);
    if (threadIdx.x == 0) *loss = (float)(Num / Den);
    if (g_pred) {
        const float inv = (float)(1.0 / Den);
        for (int i = threadIdx.x; i < n; i += LB) {
            const int r = i / K4, k = i % K4;
            const float m = y_true[(size_t)r * 2 * K4 + k];
            g_pred[(size_t)r * ldg + k] = -m * inv * smooth_l1_grad(y_true[(size_t)r * 2 * K4 + K4 + k] - pred[i]);
        }
    }
}

// g *= (y > 0)   (ReLU backward where no conv epilogue can fuse it, e.g. after the RoI-crop scatter)
__global__ void k_relu_bwd(float4* g, const float4* y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = g[i]; const float4 b = y[i];
        a.x = b.x > 0.0f ? a.x : 0.0f; a.y = b.y > 0.0f ? a.y : 0.0f; a.z = b.z > 0.0f ? a.z : 0.0f; a.w = b.w > 0.0f ? a.w : 0.0f;
        g[i] = a;
    }
}

// AveragePooling2D(k) over a k x k map followed by nothing: gx[n][h][w][c] = (y[n][h][w][c] > 0) * gp[n][c] / k^2
__global__ void k_avgpool_bwd_masked(const float4* gp, const float4* y, int hw, int C4, size_t total4, float inv, float4* gx) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const size_t n = i / ((size_t)hw * C4);
        const float4 g = gp[n * C4 + c], b = y[i];
        gx[i] = make_float4(b.x > 0.0f ? g.x * inv : 0.0f, b.y > 0.0f ? g.y * inv : 0.0f, b.z > 0.0f ? g.z * inv : 0.0f, b.w > 0.0f ? g.w * inv : 0.0f);
    }
}

// MaxPooling2D(k, stride=k) backward (vgg.py:100-128; TF MaxPoolGrad routes each window's gradient to its
// FIRST maximum in scan order).  One thread per input element; windows do not overlap (k == stride).
__global__ void k_maxpool_bwd(const float* x, const float* y, const float* gy, int n, int H, int W, int C, int k, float* gx) {
    const int Ho = H / k, Wo = W / k;
    const size_t total = (size_t)n * H * W * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int img = (int)(t / H);
        const int ho = h / k, wo = w / k;
        float g = 0.0f;
        if (ho < Ho && wo < Wo) {
            const size_t oi = (((size_t)img * Ho + ho) * Wo + wo) * C + c;
            const float m = y[oi], v = x[i];
            if (v == m) {
                bool first = true;                       // is there an earlier element of the window equal to the max?
                for (int r = ho * k; r <= h && first; ++r)
                    for (int q = wo * k; q < wo * k + k; ++q) {
                        if (r == h && q >= w) break;
                        if (x[(((size_t)img * H + r) * W + q) * C + c] == m) { first = false; break; }
                    }
                if (first) g = gy[oi];
            }
        }
        gx[i] = g;
    }
}

// Keras SGD(momentum, nesterov=False):  g += 2*l2*w;  v = momentum*v - lr*g;  w += v
__global__ void k_sgd_momentum(float* w, const float* g, float* v, size_t n, float lr, float momentum, float l2, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale + 2.0f * l2 * w[i];
        const float vi = momentum * v[i] - lr * gi;
        v[i] = vi;
        w[i] += vi;
    }
}
// the same update, four parameters per lane (16-byte loads and stores: the 4-byte form moved the flat buffers of an RPN
// step -- 47 MB of parameters, five passes -- at 2.7 TB/s, 88 us); per element the arithmetic is k_sgd_momentum's: bit-identical
__global__ void k_sgd_momentum_v4(float4* w, const float4* g, float4* v, size_t n4, float lr, float momentum, float l2, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 wi = w[i], vi = v[i];
        const float4 gg = g[i];
        const float g0 = gg.x * gscale + 2.0f * l2 * wi.x, g1 = gg.y * gscale + 2.0f * l2 * wi.y;
        const float g2 = gg.z * gscale + 2.0f * l2 * wi.z, g3 = gg.w * gscale + 2.0f * l2 * wi.w;
        vi.x = momentum * vi.x - lr * g0; vi.y = momentum * vi.y - lr * g1; vi.z = momentum * vi.z - lr * g2; vi.w = momentum * vi.w - lr * g3;
        wi.x += vi.x; wi.y += vi.y; wi.z += vi.z; wi.w += vi.w;
        v[i] = vi;
        w[i] = wi;
    }
}

// Keras Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t m/(sqrt(v)+eps)
__global__ void k_adam(float* w, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2, float eps, float l2, float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale + 2.0f * l2 * w[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// out = sum(w^2): 256 workgroups write f64 partials, one wave adds them in a fixed order (reproducible).
// The L2 penalty value l2 * sum(w^2) of Keras' regularizers.
constexpr int SUMSQ_BLOCKS = 1024;
__global__ void __launch_bounds__(256) k_sumsq_partial(const float* w, size_t n, double* partial) {
    __shared__ double scratch[4];
    double acc = 0.0;
    const size_t n4 = n / 4;                                 // 16-byte loads (the flat parameter buffer is 256-B aligned)
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)SUMSQ_BLOCKS * 256) {
        const float4 v = w4[i];
        acc += ((double)v.x * (double)v.x + (double)v.y * (double)v.y) + ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) { const double v = (double)w[4 * n4 + threadIdx.x]; acc += v * v; }
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
__global__ void __launch_bounds__(64) k_sumsq_final(const double* partial, float* out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < SUMSQ_BLOCKS; i += 64) acc += partial[i];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) *out = (float)acc;
}

// shift[n] = bias[n]*scale[n] + shift_const[n]: re-fold a TRAINABLE conv bias into the frozen BatchNorm epilogue
__global__ void k_fold_bias(const float* bias, const float* scale, const float* shift_const, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (bias ? bias[i] : 0.0f) * (scale ? scale[i] : 1.0f) + (shift_const ? shift_const[i] : 0.0f);
}

static inline int ew_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

}  // namespace frcnn

using namespace frcnn;

extern "C" {

int frcnn_loss_rpn_cls(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_logit, void* stream) {
    if (!y_true || !y_pred || !loss || cells <= 0 || A <= 0) return fail(FRCNN_E_ARG, "loss_rpn_cls: bad argument");
    k_loss_rpn_cls<<<1, LB, 0, as_stream(stream)>>>(y_true, y_pred, cells, A, loss, grad_logit);
    return check_launch("loss_rpn_cls");
}
int frcnn_loss_rpn_reg(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_pred, void* stream) {
    if (!y_true || !y_pred || !loss || cells <= 0 || A <= 0) return fail(FRCNN_E_ARG, "loss_rpn_reg: bad argument");
    k_loss_rpn_reg<<<1, LB, 0, as_stream(stream)>>>(y_true, y_pred, cells, 4 * A, loss, grad_pred);
    return check_launch("loss_rpn_reg");
}
size_t frcnn_loss_workspace_bytes(void) { return 2 * LOSS_WGS * sizeof(double); }
int frcnn_loss_rpn_cls_ws(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_logit, void* workspace, void* stream) {
    if (!y_true || !y_pred || !loss || !workspace || cells <= 0 || A <= 0) return fail(FRCNN_E_ARG, "loss_rpn_cls_ws: bad argument");
    k_loss_rpn_cls_part<<<LOSS_WGS, LOSS_T, 0, as_stream(stream)>>>(y_true, y_pred, cells, A, (double*)workspace, grad_logit);
    k_loss_rpn_cls_final<<<1, 1, 0, as_stream(stream)>>>((const double*)workspace, loss);
    return check_launch("loss_rpn_cls_ws");
}
int frcnn_loss_rpn_reg_ws(const float* y_true, const float* y_pred, int cells, int A, float* loss, float* grad_pred, void* workspace, void* stream) {
    if (!y_true || !y_pred || !loss || !workspace || cells <= 0 || A <= 0) return fail(FRCNN_E_ARG, "loss_rpn_reg_ws: bad argument");
    k_loss_rpn_reg_part<<<LOSS_WGS, LOSS_T, 0, as_stream(stream)>>>(y_true, y_pred, cells, 4 * A, (double*)workspace);
    k_loss_rpn_reg_final<<<grad_pred ? LOSS_WGS : 1, LOSS_T, 0, as_stream(stream)>>>(y_true, y_pred, cells, 4 * A, (const double*)workspace, loss, grad_pred);
    return check_launch("loss_rpn_reg_ws");
}
int frcnn_loss_det_cls(const float* y_true, const float* y_pred, int n_rois, int C, float* loss, float* grad_logit, int ldg, void* stream) {
    if (!y_true || !y_pred || !loss || n_rois <= 0 || C <= 1 || (grad_logit && ldg < C)) return fail(FRCNN_E_ARG, "loss_det_cls: bad argument");
    k_loss_det_cls<<<1, LB, 0, as_stream(stream)>>>(y_true, y_pred, n_rois, C, loss, grad_logit, ldg);
    return check_launch("loss_det_cls");
}
int frcnn_loss_det_reg(const float* y_true, const float* y_pred, int n_rois, int num_classes_excl_bg, float* loss, float* grad_pred, int ldg, void* stream) {
    if (!y_true || !y_pred || !loss || n_rois <= 0 || num_classes_excl_bg <= 0 || (grad_pred && ldg < 4 * num_classes_excl_bg)) return fail(FRCNN_E_ARG, "loss_det_reg: bad argument");
    k_loss_det_reg<<<1, LB, 0, as_stream(stream)>>>(y_true, y_pred, n_rois, 4 * num_classes_excl_bg, loss, grad_pred, ldg);
    return check_launch("loss_det_reg");
}
int frcnn_relu_bwd_inplace(float* g, const float* y, size_t n, void* stream) {
    if (!g || !y || (n & 3)) return fail(FRCNN_E_ARG, "relu_bwd_inplace: bad argument (n must be a multiple of 4)");
    if (n == 0) return FRCNN_OK;
    k_relu_bwd<<<ew_grid(n / 4), 256, 0, as_stream(stream)>>>((float4*)g, (const float4*)y, n / 4);
    return check_launch("relu_bwd_inplace");
}
int frcnn_avgpool_bwd_masked(const float* g_pooled, const float* y, int n, int k, int c, float* gx, void* stream) {
    if (!g_pooled || !y || !gx || n <= 0 || k <= 0 || c <= 0 || (c & 3)) return fail(FRCNN_E_ARG, "avgpool_bwd_masked: bad argument");
    const size_t total4 = (size_t)n * k * k * (c / 4);
    k_avgpool_bwd_masked<<<ew_grid(total4), 256, 0, as_stream(stream)>>>((const float4*)g_pooled, (const float4*)y, k * k, c / 4, total4, 1.0f / (float)(k * k), (float4*)gx);
    return check_launch("avgpool_bwd_masked");
}
int frcnn_maxpool_bwd(const float* x, const float* y, const float* gy, int n, int h, int w, int c, int k, float* gx, void* stream) {
    if (!x || !y || !gy || !gx || n <= 0 || h < k || w < k || c <= 0 || k <= 0) return fail(FRCNN_E_ARG, "maxpool_bwd: bad argument");
    k_maxpool_bwd<<<ew_grid((size_t)n * h * w * c), 256, 0, as_stream(stream)>>>(x, y, gy, n, h, w, c, k, gx);
    return check_launch("maxpool_bwd");
}
int frcnn_sgd_momentum(float* w, const float* g, float* v, size_t n, float lr, float momentum, float l2, float grad_scale, void* stream) {
    if (!w || !g || !v) return fail(FRCNN_E_ARG, "sgd_momentum: null pointer");
    if (n == 0) return FRCNN_OK;
    const size_t n4 = ((((uintptr_t)w | (uintptr_t)g | (uintptr_t)v) & 15) == 0) ? n / 4 : 0;
    if (n4) k_sgd_momentum_v4<<<ew_grid(n4), 256, 0, as_stream(stream)>>>((float4*)w, (const float4*)g, (float4*)v, n4, lr, momentum, l2, grad_scale);
    if (n > 4 * n4)                                              // the tail -- or all of it when a buffer is not 16-byte aligned
        k_sgd_momentum<<<n4 ? 1 : ew_grid(n), n4 ? 64 : 256, 0, as_stream(stream)>>>(w + 4 * n4, g + 4 * n4, v + 4 * n4, n - 4 * n4, lr, momentum, l2, grad_scale);
    return check_launch("sgd_momentum");
}
int frcnn_adam(float* w, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int t, float l2, float grad_scale, void* stream) {
    if (!w || !g || !m || !v || t < 1) return fail(FRCNN_E_ARG, "adam: bad argument");
    if (n == 0) return FRCNN_OK;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    k_adam<<<ew_grid(n), 256, 0, as_stream(stream)>>>(w, g, m, v, n, (float)lr_t, beta1, beta2, eps, l2, grad_scale);
    return check_launch("adam");
}
int frcnn_fold_bias(const float* bias, const float* scale, const float* shift_const, float* out, int n, void* stream) {
    if (!out || n <= 0) return fail(FRCNN_E_ARG, "fold_bias: bad argument");
    k_fold_bias<<<(n + 255) / 256, 256, 0, as_stream(stream)>>>(bias, scale, shift_const, out, n);
    return check_launch("fold_bias");
}
size_t frcnn_sumsq_workspace_bytes(void) { return SUMSQ_BLOCKS * sizeof(double); }
int frcnn_sumsq(const float* w, size_t n, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!w || !out) return fail(FRCNN_E_ARG, "sumsq: null pointer");
    if (!workspace || workspace_bytes < frcnn_sumsq_workspace_bytes()) return fail(FRCNN_E_WORKSPACE, "sumsq: workspace needs %zu bytes", frcnn_sumsq_workspace_bytes());
    if ((uintptr_t)w & 15) return fail(FRCNN_E_ARG, "sumsq: w must be 16-byte aligned");
    k_sumsq_partial<<<SUMSQ_BLOCKS, 256, 0, as_stream(stream)>>>(w, n, (double*)workspace);
    if (int e = check_launch("sumsq")) return e;
    k_sumsq_final<<<1, 64, 0, as_stream(stream)>>>((const double*)workspace, out);
    return check_launch("sumsq");
}

}  // extern "C"
