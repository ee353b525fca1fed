// Batch normalisation over channel-last rows for gfx950: the norm layer of the Cascade configs' ConvFCBBoxHead
// (configs/swin/cascade_mask_rcnn_swin_*: norm_cfg=dict(type='SyncBN'); convfc_bbox_head.py:99-107 builds
// ConvModule(conv3x3 -> norm -> ReLU)).  x is (R, C) = (RoIs * 7 * 7, 256) in the NHWC layout the conv kernels write.
//
//   forward:  det_bn_stats     per-channel {sum x, sum x^2, count}      -> [all-reduce over ranks for SyncBN, by the caller]
//             det_bn_finalize  mean, 1/sqrt(var+eps), running-stat update (momentum, unbiased variance)
//             det_bn_apply     y = relu?((x - mean) * invstd * gamma + beta)
//   backward: det_bn_bwd_reduce  per-channel {sum dy', sum dy' * xhat}, dy' = dy * [y > 0]  (= dbeta, dgamma of this rank)
//                                                                        -> [all-reduce for SyncBN]
//             det_bn_bwd_apply   dx = gamma * invstd * (dy' - sum_dy'/N - xhat * sum_dy'xhat/N)
// which is torch.nn.SyncBatchNorm's arithmetic (_functions.py: batch_norm_stats / gather_stats / backward_reduce /
// backward_elemt) with the ReLU of the ConvModule folded into the apply passes.  Every pass is one HBM sweep of the
// (R, C) activation (25.7 MB at 1024 RoIs in bf16) with 16-byte accesses; sums are fp32.
#include "common.h"

#define BN_MAXG 512

// Partial per-channel sums.  256 threads = RPP row slots x TPR threads per row (TPR = C / VN); block g walks rows
// g*RPP + slot, += G*RPP.  BWD: a = dy', b = dy' * xhat; else a = x, b = x^2.  part layout (G, 2, C).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy, int64_t R, int C,
                                                         const float* __restrict__ mean_invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int relu, float* __restrict__ part) {
    constexpr int VN = Vec16<T>::N;
    __shared__ float red[2][256][VN];
    const int TPR = C / VN, RPP = 256 / TPR;
    const int slot = threadIdx.x / TPR, tc = threadIdx.x - slot * TPR;
    float a[VN], b[VN];
#pragma unroll
    for (int j = 0; j < VN; ++j) { a[j] = 0.f; b[j] = 0.f; }
    if (slot < RPP) {
        float mu[VN], is[VN], ga[VN], be[VN];
        if (BWD) {
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                const int c = tc * VN + j;
                mu[j] = mean_invstd[c]; is[j] = mean_invstd[C + c]; ga[j] = gamma[c]; be[j] = beta[c];
            }
        }
        for (int64_t r = (int64_t)blockIdx.x * RPP + slot; r < R; r += (int64_t)gridDim.x * RPP) {
            Vec16<T> vx; vx.load(x + r * C + tc * VN);
            if (BWD) {
                Vec16<T> vd; vd.load(dy + r * C + tc * VN);
#pragma unroll
                for (int j = 0; j < VN; ++j) {
                    const float xh = (vx.get(j) - mu[j]) * is[j];
                    float d = vd.get(j);
                    if (relu && !(xh * ga[j] + be[j] > 0.f)) d = 0.f;
                    a[j] += d; b[j] += d * xh;
                }
            } else {
#pragma unroll
                for (int j = 0; j < VN; ++j) { const float v = vx.get(j); a[j] += v; b[j] += v * v; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VN; ++j) { red[0][threadIdx.x][j] = a[j]; red[1][threadIdx.x][j] = b[j]; }
    __syncthreads();
    if (slot == 0) {
        for (int s = 1; s < RPP; ++s) {
#pragma unroll
            for (int j = 0; j < VN; ++j) { a[j] += red[0][s * TPR + tc][j]; b[j] += red[1][s * TPR + tc][j]; }
        }
        float* p = part + (int64_t)blockIdx.x * 2 * C + tc * VN;
#pragma unroll
        for (int j = 0; j < VN; ++j) { p[j] = a[j]; p[C + j] = b[j]; }
    }
}

// sums[i] = sum_g part[g][i] for i in [0, 2C); sums[2C] = count (when count >= 0).  One block folds 32 columns: 8 row
// slices of the G partial rows run side by side (a single thread per column walking all G rows was latency-bound: 65 us).
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ part, int G, int C, float count, float* __restrict__ sums) {
    __shared__ float red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + tx;
    float s0 = 0.f, s1 = 0.f;
    if (i < 2 * C) {
        int g = ty;
        for (; g + 8 < G; g += 16) { s0 += part[(int64_t)g * 2 * C + i]; s1 += part[(int64_t)(g + 8) * 2 * C + i]; }
        if (g < G) s0 += part[(int64_t)g * 2 * C + i];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][tx];
        sums[i] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && count >= 0.f) sums[2 * C] = count;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, int C, float eps, float momentum,
                                                          float* __restrict__ mean_invstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float n = sums[2 * C];
    const float mean = sums[c] / n;
    const float var = fmaxf(sums[C + c] / n - mean * mean, 0.f);          // biased, as used for normalisation
    mean_invstd[c] = mean;
    mean_invstd[C + c] = rsqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
}

// BWD: out = dx (sums / count given); else out = y.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int64_t R,
                                                       int C, const float* __restrict__ mean_invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu, const float* __restrict__ sums,
                                                       const float* __restrict__ count) {
    constexpr int VN = Vec16<T>::N;
    extern __shared__ float sm[];                 // [6][C]: scale, shift, (bwd) mean_dy, mean_dy_xhat, gamma, beta
    float* sc = sm; float* sh = sm + C; float* m1 = sm + 2 * C; float* m2 = sm + 3 * C; float* ga = sm + 4 * C; float* be = sm + 5 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float mu = mean_invstd[c], is = mean_invstd[C + c];
        if (BWD) {
            const float inv_n = 1.f / count[0];
            sc[c] = is; sh[c] = -mu * is;                                   // xhat = x * sc + sh
            m1[c] = sums[c] * inv_n; m2[c] = sums[C + c] * inv_n;
            ga[c] = gamma[c]; be[c] = beta[c];
        } else {
            sc[c] = gamma[c] * is; sh[c] = beta[c] - mu * gamma[c] * is;
        }
    }
    __syncthreads();
    const int TPR = C / VN;
    const int64_t total = R * TPR, step = (int64_t)gridDim.x * 256;
    const int dstep = (int)(step % TPR);
    int cv = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % TPR) - dstep;
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < total; v += step) {
        cv += dstep; if (cv >= TPR) cv -= TPR; if (cv < 0) cv += TPR;
        const int c0 = cv * VN;
        Vec16<T> vx, vo; vx.load(x + v * VN);
        if (BWD) {
            Vec16<T> vd; vd.load(dy + v * VN);
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                const int c = c0 + j;
                const float xh = vx.get(j) * sc[c] + sh[c];
                const float g = ga[c];
                float d = vd.get(j);
                if (relu && !(xh * g + be[c] > 0.f)) d = 0.f;
                vo.set(j, g * sc[c] * (d - m1[c] - xh * m2[c]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < VN; ++j) {
                float y = vx.get(j) * sc[c0 + j] + sh[c0 + j];
                if (relu) y = fmaxf(y, 0.f);
                vo.set(j, y);
            }
        }
        vo.store(out + v * VN);
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
static bool bn_shape_ok(int64_t R, int C, int dtype) {
    const int vn = dtype == SWIN_F32 ? 4 : 8;
    return R > 0 && C > 0 && C % vn == 0 && C / vn <= 256 && (dtype == SWIN_F32 || dtype == SWIN_BF16);
}
static int bn_groups(int64_t R, int C, int dtype) {
    const int vn = dtype == SWIN_F32 ? 4 : 8, rpp = 256 / (C / vn);
    const int64_t want = (R + (int64_t)rpp * 8 - 1) / ((int64_t)rpp * 8);
    return (int)(want < 1 ? 1 : (want > BN_MAXG ? BN_MAXG : want));
}

extern "C" int64_t det_bn_workspace_bytes(int C) { return C > 0 ? (int64_t)BN_MAXG * 2 * C * 4 : 0; }

// sums (2C+1) f32 out: per-channel sum, sum of squares, and the row count R (as float) in the last slot.
extern "C" int det_bn_stats(const void* x, int64_t R, int C, float* sums, void* workspace, int dtype, void* stream) {
    if (!x || !sums || !workspace) return SWIN_ERR_BAD_ARG;
    if (!bn_shape_ok(R, C, dtype)) return SWIN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int G = bn_groups(R, C, dtype);
    float* part = (float*)workspace;
    if (dtype == SWIN_F32)
        bn_partial_kernel<float, false><<<G, 256, 0, s>>>((const float*)x, nullptr, R, C, nullptr, nullptr, nullptr, 0, part);
    else
        bn_partial_kernel<bf16, false><<<G, 256, 0, s>>>((const bf16*)x, nullptr, R, C, nullptr, nullptr, nullptr, 0, part);
    bn_fold_kernel<<<(2 * C + 31) / 32, 256, 0, s>>>(part, G, C, (float)R, sums);
    return swin_launch_status();
}

// sums (2C+1) (possibly all-reduced) -> mean_invstd (2C); running_mean / running_var (C) updated in place when non-NULL.
extern "C" int det_bn_finalize(const float* sums, int C, float eps, float momentum, float* mean_invstd, float* running_mean,
                               float* running_var, void* stream) {
    if (!sums || !mean_invstd || C <= 0 || !(eps > 0.f)) return SWIN_ERR_BAD_ARG;
    bn_finalize_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(sums, C, eps, momentum, mean_invstd, running_mean, running_var);
    return swin_launch_status();
}

extern "C" int det_bn_apply(const void* x, void* y, int64_t R, int C, const float* mean_invstd, const float* gamma, const float* beta,
                            int relu, int dtype, void* stream) {
    if (!x || !y || !mean_invstd || !gamma || !beta) return SWIN_ERR_BAD_ARG;
    if (!bn_shape_ok(R, C, dtype)) return SWIN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int vn = dtype == SWIN_F32 ? 4 : 8;
    const int64_t total = R * (C / vn);
    const int blocks = (int)(total / 256 / 4 + 1 > 2048 ? 2048 : total / 256 / 4 + 1);
    const size_t shm = (size_t)6 * C * sizeof(float);
    if (dtype == SWIN_F32)
        bn_apply_kernel<float, false><<<blocks, 256, shm, s>>>((const float*)x, nullptr, (float*)y, R, C, mean_invstd, gamma, beta, relu, nullptr, nullptr);
    else
        bn_apply_kernel<bf16, false><<<blocks, 256, shm, s>>>((const bf16*)x, nullptr, (bf16*)y, R, C, mean_invstd, gamma, beta, relu, nullptr, nullptr);
    return swin_launch_status();
}

// sums (2C) f32 out: this rank's sum dy' (= dbeta) and sum dy' * xhat (= dgamma).
extern "C" int det_bn_bwd_reduce(const void* x, const void* dy, int64_t R, int C, const float* mean_invstd, const float* gamma,
                                 const float* beta, int relu, float* sums, void* workspace, int dtype, void* stream) {
    if (!x || !dy || !mean_invstd || !gamma || !beta || !sums || !workspace) return SWIN_ERR_BAD_ARG;
    if (!bn_shape_ok(R, C, dtype)) return SWIN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int G = bn_groups(R, C, dtype);
    float* part = (float*)workspace;
    if (dtype == SWIN_F32)
        bn_partial_kernel<float, true><<<G, 256, 0, s>>>((const float*)x, (const float*)dy, R, C, mean_invstd, gamma, beta, relu, part);
    else
        bn_partial_kernel<bf16, true><<<G, 256, 0, s>>>((const bf16*)x, (const bf16*)dy, R, C, mean_invstd, gamma, beta, relu, part);
    bn_fold_kernel<<<(2 * C + 31) / 32, 256, 0, s>>>(part, G, C, -1.f, sums);
    return swin_launch_status();
}

// sums (2C): the (all-reduced) backward sums; count: DEVICE pointer to the (global) row count N as float.
extern "C" int det_bn_bwd_apply(const void* x, const void* dy, void* dx, int64_t R, int C, const float* mean_invstd, const float* gamma,
                                const float* beta, int relu, const float* sums, const float* count, int dtype, void* stream) {
    if (!x || !dy || !dx || !mean_invstd || !gamma || !beta || !sums || !count) return SWIN_ERR_BAD_ARG;
    if (!bn_shape_ok(R, C, dtype)) return SWIN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int vn = dtype == SWIN_F32 ? 4 : 8;
    const int64_t total = R * (C / vn);
    const int blocks = (int)(total / 256 / 4 + 1 > 2048 ? 2048 : total / 256 / 4 + 1);
    const size_t shm = (size_t)6 * C * sizeof(float);
    if (dtype == SWIN_F32)
        bn_apply_kernel<float, true><<<blocks, 256, shm, s>>>((const float*)x, (const float*)dy, (float*)dx, R, C, mean_invstd, gamma, beta, relu, sums, count);
    else
        bn_apply_kernel<bf16, true><<<blocks, 256, shm, s>>>((const bf16*)x, (const bf16*)dy, (bf16*)dx, R, C, mean_invstd, gamma, beta, relu, sums, count);
    return swin_launch_status();
}
