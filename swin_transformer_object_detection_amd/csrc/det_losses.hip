// Loss kernels of the detector heads for gfx950, forward value + input gradients, fixed-size samples.
// Reference arithmetic:
//   RPN:   AnchorHead.loss_single (mmdet/models/dense_heads/anchor_head.py:375-434): sigmoid cross entropy over the
//          sampled anchors + L1 on the positives' deltas, both divided by the number of sampled anchors of the batch;
//   bbox:  BBoxHead.loss (mmdet/models/roi_heads/bbox_heads/bbox_head.py:188-238): softmax cross entropy (avg_factor =
//          number of sampled RoIs), accuracy, class-specific L1 on the positives divided by the number of samples;
//   mask:  FCNMaskHead.loss / mask_cross_entropy (mmdet/models/losses/cross_entropy_loss.py): mean sigmoid BCE over
//          (positives x 28 x 28) of the labelled class channel.
// The reference spends 15-40 elementwise launches per loss (gathers, casts, BCE, masks, sums) and as many again in
// autograd; each loss here is one forward and one backward launch.  Accumulation in fp32; inputs bf16 or fp32.
#include "common.h"

__device__ __forceinline__ float block_sum(float v, float* red) {      // blockDim.x <= 1024, red[16]
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float bce_logits(float x, float y) {       // max(x,0) - x*y + log(1 + exp(-|x|))
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ RPN
// cls (B, A) logits, reg (B, A, 4) deltas; per image S sample slots: inds, flags (bit 0 used, bit 1 positive), tgt (S,4).
// out[0] = sum BCE / n, out[1] = sum L1 / n, out[2] = n (sampled anchors of the whole batch, >= 1).
template <typename T>
__global__ __launch_bounds__(1024) void rpn_loss_fwd_kernel(const T* __restrict__ cls, const T* __restrict__ reg, int B, int64_t A,
                                                            int S, const int64_t* __restrict__ inds,
                                                            const uint8_t* __restrict__ flags, const float* __restrict__ tgt,
                                                            float* __restrict__ out) {
    __shared__ float red[16];
    float lc = 0.f, lb = 0.f, cnt = 0.f;
    for (int i = threadIdx.x; i < B * S; i += blockDim.x) {
        const uint8_t f = flags[i];
        if (!(f & 1)) continue;
        const int b = i / S;
        const int64_t a = inds[i];
        const float y = (f & 2) ? 1.f : 0.f;
        lc += bce_logits(Elt<T>::ld(cls + b * A + a), y);
        cnt += 1.f;
        if (f & 2) {
            const T* r = reg + (b * A + a) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) lb += fabsf(Elt<T>::ld(r + k) - tgt[(int64_t)i * 4 + k]);
        }
    }
    lc = block_sum(lc, red); lb = block_sum(lb, red); cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        const float n = fmaxf(cnt, 1.f);
        out[0] = lc / n; out[1] = lb / n; out[2] = n;
    }
}

// dcls / dreg are ZEROED by the caller; sample indices of one image are distinct, so plain stores.
template <typename T>
__global__ __launch_bounds__(256) void rpn_loss_bwd_kernel(const T* __restrict__ cls, const T* __restrict__ reg, int B, int64_t A,
                                                           int S, const int64_t* __restrict__ inds,
                                                           const uint8_t* __restrict__ flags, const float* __restrict__ tgt,
                                                           const float* __restrict__ out, const float* __restrict__ gout,
                                                           T* __restrict__ dcls, T* __restrict__ dreg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * S) return;
    const uint8_t f = flags[i];
    if (!(f & 1)) return;
    const float inv = 1.f / out[2], g0 = gout[0] * inv, g1 = gout[1] * inv;
    const int b = i / S;
    const int64_t a = inds[i];
    const float y = (f & 2) ? 1.f : 0.f;
    Elt<T>::st(dcls + b * A + a, (sigmoidf(Elt<T>::ld(cls + b * A + a)) - y) * g0);
    if (f & 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = Elt<T>::ld(reg + (b * A + a) * 4 + k) - tgt[(int64_t)i * 4 + k];
            Elt<T>::st(dreg + (b * A + a) * 4 + k, (d > 0.f ? g1 : (d < 0.f ? -g1 : 0.f)));
        }
    }
}

// ------------------------------------------------------------------------------------------------ bbox head
// cls (n, nc+1) logits, bbox (n, 4 nc) class-specific deltas, labels (n) in [0, nc] (nc = background), tgt (n,4),
// flags (n) (bit 0 used, bit 1 positive).  out[0] = CE sum / nv, out[1] = accuracy (%), out[2] = L1 sum / nv, out[3] = nv.
template <typename T>
__global__ __launch_bounds__(1024) void bbox_loss_fwd_kernel(const T* __restrict__ cls, const T* __restrict__ bbox, int n, int nc,
                                                             const int64_t* __restrict__ labels, const float* __restrict__ tgt,
                                                             const uint8_t* __restrict__ flags, float* __restrict__ out,
                                                             float* __restrict__ lse) {
    __shared__ float red[16];
    float ce = 0.f, hit = 0.f, l1 = 0.f, cnt = 0.f;
    const int C = nc + 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const T* row = cls + (int64_t)i * C;
        float m = -3.0e38f; int am = 0;
        for (int c = 0; c < C; ++c) { const float v = Elt<T>::ld(row + c); if (v > m) { m = v; am = c; } }
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(Elt<T>::ld(row + c) - m);
        const float l = m + logf(s);
        lse[i] = l;
        const uint8_t f = flags[i];
        if (!(f & 1)) continue;
        const int lab = (int)labels[i];
        ce += l - Elt<T>::ld(row + lab);
        hit += (am == lab) ? 1.f : 0.f;
        cnt += 1.f;
        if ((f & 2) && lab < nc) {
            const T* p = bbox + ((int64_t)i * nc + lab) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) l1 += fabsf(Elt<T>::ld(p + k) - tgt[(int64_t)i * 4 + k]);
        }
    }
    ce = block_sum(ce, red); hit = block_sum(hit, red); l1 = block_sum(l1, red); cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        const float nv = fmaxf(cnt, 1.f);
        out[0] = ce / nv; out[1] = hit / nv * 100.f; out[2] = l1 / nv; out[3] = nv;
    }
}

// one thread per (row, column) of dcls and of dbbox: every element is written (zeros included)
template <typename T>
__global__ __launch_bounds__(256) void bbox_loss_bwd_kernel(const T* __restrict__ cls, const T* __restrict__ bbox, int n, int nc,
                                                            const int64_t* __restrict__ labels, const float* __restrict__ tgt,
                                                            const uint8_t* __restrict__ flags, const float* __restrict__ out,
                                                            const float* __restrict__ lse, const float* __restrict__ gout,
                                                            T* __restrict__ dcls, T* __restrict__ dbbox) {
    const int C = nc + 1, W = C + 4 * nc;
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= (int64_t)n * W) return;
    const int i = (int)(id / W), col = (int)(id - (int64_t)i * W);
    const uint8_t f = flags[i];
    const int lab = (int)labels[i];
    const float inv = 1.f / out[3];
    if (col < C) {
        float g = 0.f;
        if (f & 1) g = (expf(Elt<T>::ld(cls + (int64_t)i * C + col) - lse[i]) - (col == lab ? 1.f : 0.f)) * gout[0] * inv;
        Elt<T>::st(dcls + (int64_t)i * C + col, g);
    } else {
        const int q = col - C, c = q >> 2, k = q & 3;
        float g = 0.f;
        if ((f & 2) && c == lab) {
            const float d = Elt<T>::ld(bbox + (int64_t)i * 4 * nc + q) - tgt[(int64_t)i * 4 + k];
            g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * gout[2] * inv;
        }
        Elt<T>::st(dbbox + (int64_t)i * 4 * nc + q, g);
    }
}

// ------------------------------------------------------------------------------------------------ mask head
// pred (n, nc, P) logits (P = 28*28), target (n, P) in {0,1}, labels (n), valid (n) u8.
// out[0] = sum_valid mean_P BCE / max(#valid,1), out[1] = max(#valid,1).  grid = n blocks.
template <typename T>
__global__ __launch_bounds__(256) void mask_loss_fwd_kernel(const T* __restrict__ pred, int n, int nc, int P,
                                                            const float* __restrict__ target, const int64_t* __restrict__ labels,
                                                            const uint8_t* __restrict__ valid, float* __restrict__ per_roi) {
    __shared__ float red[16];
    const int i = blockIdx.x;
    float s = 0.f;
    if (valid[i]) {
        const T* p = pred + ((int64_t)i * nc + labels[i]) * P;
        const float* t = target + (int64_t)i * P;
        for (int k = threadIdx.x; k < P; k += 256) s += bce_logits(Elt<T>::ld(p + k), t[k]);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) per_roi[i] = s / (float)P;
}

__global__ __launch_bounds__(1024) void mask_loss_final_kernel(const float* __restrict__ per_roi, const uint8_t* __restrict__ valid,
                                                               int n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (valid[i]) { s += per_roi[i]; c += 1.f; }
    s = block_sum(s, red); c = block_sum(c, red);
    if (threadIdx.x == 0) { const float nv = fmaxf(c, 1.f); out[0] = s / nv; out[1] = nv; }
}

// dpred is ZEROED by the caller; only the labelled channel of valid RoIs is written.
template <typename T>
__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const T* __restrict__ pred, int n, int nc, int P,
                                                            const float* __restrict__ target, const int64_t* __restrict__ labels,
                                                            const uint8_t* __restrict__ valid, const float* __restrict__ out,
                                                            const float* __restrict__ gout, T* __restrict__ dpred) {
    const int i = blockIdx.x;
    if (!valid[i]) return;
    const float g = gout[0] / (out[1] * (float)P);
    const int64_t base = ((int64_t)i * nc + labels[i]) * P;
    const float* t = target + (int64_t)i * P;
    for (int k = threadIdx.x; k < P; k += 256)
        Elt<T>::st(dpred + base + k, (sigmoidf(Elt<T>::ld(pred + base + k)) - t[k]) * g);
}

// ------------------------------------------------------------------------------------------------ RPN flatten
// The fused RPN head GEMM yields, per level, (B, HW, CH) rows [A cls logits | 4A deltas | padding] (CH = 5A rounded up
// to 8).  loss() and get_bboxes() want the anchor-major concatenation over levels: cls_all (B, sum HW*A) and
// reg_all (B, sum HW*A, 4) (anchor_head.py:474-486, rpn_head.py:119-125).  With torch this is 10 strided slice copies
// + 2 concatenations forward and 10 zero fills + 10 strided copies + 5 adds backward; here one kernel each way.
struct FlatLevels { const void* y[8]; void* dy[8]; int hw[8]; int off[8]; int L; };

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void rpn_flatten_kernel(FlatLevels F, int B, int A, int CH, int64_t total_anchors,
                                                          T* __restrict__ cls_all, T* __restrict__ reg_all, int64_t total) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;        // over (b, level token, channel)
    if (id >= total) return;
    const int ch = (int)(id % CH);
    int64_t tok = id / CH;                                               // b * sum_hw + (level offset + hw)
    int sum_hw = F.off[F.L - 1] + F.hw[F.L - 1];
    const int b = (int)(tok / sum_hw);
    const int g = (int)(tok - (int64_t)b * sum_hw);
    int l = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) if (q < F.L && g >= F.off[q]) l = q;
    const int hw = g - F.off[l];
    const int64_t src = ((int64_t)b * F.hw[l] + hw) * CH + ch;
    const int64_t anchor0 = (int64_t)b * total_anchors + ((int64_t)F.off[l] + hw) * A;
    if (!BWD) {
        const T v = ((const T*)F.y[l])[src];
        if (ch < A) cls_all[anchor0 + ch] = v;
        else if (ch < 5 * A) reg_all[anchor0 * 4 + (ch - A)] = v;
    } else {
        T v = (T)0.f;
        if (ch < A) v = cls_all[anchor0 + ch];
        else if (ch < 5 * A) v = reg_all[anchor0 * 4 + (ch - A)];
        ((T*)F.dy[l])[src] = v;
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
    if ((dtype) == SWIN_F32) { CALL_F32; } else if ((dtype) == SWIN_BF16) { CALL_BF16; } else return SWIN_ERR_UNSUPPORTED;

extern "C" int det_rpn_loss_fwd(const void* cls, const void* reg, int B, int64_t A, int S, const int64_t* inds, const uint8_t* flags,
                                const float* targets, float* out3, int dtype, void* stream) {
    if (!cls || !reg || !inds || !flags || !targets || !out3 || B <= 0 || A <= 0 || S <= 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, (rpn_loss_fwd_kernel<float><<<1, 1024, 0, s>>>((const float*)cls, (const float*)reg, B, A, S, inds, flags, targets, out3)),
               (rpn_loss_fwd_kernel<bf16><<<1, 1024, 0, s>>>((const bf16*)cls, (const bf16*)reg, B, A, S, inds, flags, targets, out3)))
    return swin_launch_status();
}

// dcls (B,A) / dreg (B,A,4): zeroed by the caller; grad_out: 2 floats (d loss_cls, d loss_bbox); out3 from the forward.
extern "C" int det_rpn_loss_bwd(const void* cls, const void* reg, int B, int64_t A, int S, const int64_t* inds, const uint8_t* flags,
                                const float* targets, const float* out3, const float* grad_out, void* dcls, void* dreg, int dtype,
                                void* stream) {
    if (!cls || !reg || !inds || !flags || !targets || !out3 || !grad_out || !dcls || !dreg || B <= 0 || A <= 0 || S <= 0)
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (B * S + 255) / 256;
    DISPATCH_T(dtype, (rpn_loss_bwd_kernel<float><<<blocks, 256, 0, s>>>((const float*)cls, (const float*)reg, B, A, S, inds, flags, targets, out3, grad_out, (float*)dcls, (float*)dreg)),
               (rpn_loss_bwd_kernel<bf16><<<blocks, 256, 0, s>>>((const bf16*)cls, (const bf16*)reg, B, A, S, inds, flags, targets, out3, grad_out, (bf16*)dcls, (bf16*)dreg)))
    return swin_launch_status();
}

// out4: loss_cls, accuracy (%), loss_bbox, n_valid; lse (n) f32 scratch kept for the backward.
extern "C" int det_bbox_loss_fwd(const void* cls, const void* bbox, int n, int num_classes, const int64_t* labels, const float* targets,
                                 const uint8_t* flags, float* out4, float* lse, int dtype, void* stream) {
    if (!cls || !bbox || !labels || !targets || !flags || !out4 || !lse || n <= 0 || num_classes <= 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, (bbox_loss_fwd_kernel<float><<<1, 1024, 0, s>>>((const float*)cls, (const float*)bbox, n, num_classes, labels, targets, flags, out4, lse)),
               (bbox_loss_fwd_kernel<bf16><<<1, 1024, 0, s>>>((const bf16*)cls, (const bf16*)bbox, n, num_classes, labels, targets, flags, out4, lse)))
    return swin_launch_status();
}

// grad_out: 4 floats aligned with out4 (entries 0 and 2 are used); dcls (n, nc+1), dbbox (n, 4 nc): fully written.
extern "C" int det_bbox_loss_bwd(const void* cls, const void* bbox, int n, int num_classes, const int64_t* labels, const float* targets,
                                 const uint8_t* flags, const float* out4, const float* lse, const float* grad_out, void* dcls,
                                 void* dbbox, int dtype, void* stream) {
    if (!cls || !bbox || !labels || !targets || !flags || !out4 || !lse || !grad_out || !dcls || !dbbox || n <= 0 || num_classes <= 0)
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)n * (5 * num_classes + 1);
    const int blocks = (int)((total + 255) / 256);
    DISPATCH_T(dtype, (bbox_loss_bwd_kernel<float><<<blocks, 256, 0, s>>>((const float*)cls, (const float*)bbox, n, num_classes, labels, targets, flags, out4, lse, grad_out, (float*)dcls, (float*)dbbox)),
               (bbox_loss_bwd_kernel<bf16><<<blocks, 256, 0, s>>>((const bf16*)cls, (const bf16*)bbox, n, num_classes, labels, targets, flags, out4, lse, grad_out, (bf16*)dcls, (bf16*)dbbox)))
    return swin_launch_status();
}

// out2: loss, n_valid; per_roi (n) f32 scratch.
extern "C" int det_mask_loss_fwd(const void* pred, int n, int num_classes, int P, const float* target, const int64_t* labels,
                                 const uint8_t* valid, float* out2, float* per_roi, int dtype, void* stream) {
    if (!pred || !target || !labels || !valid || !out2 || !per_roi || n <= 0 || num_classes <= 0 || P <= 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, (mask_loss_fwd_kernel<float><<<n, 256, 0, s>>>((const float*)pred, n, num_classes, P, target, labels, valid, per_roi)),
               (mask_loss_fwd_kernel<bf16><<<n, 256, 0, s>>>((const bf16*)pred, n, num_classes, P, target, labels, valid, per_roi)))
    mask_loss_final_kernel<<<1, 1024, 0, s>>>(per_roi, valid, n, out2);
    return swin_launch_status();
}

// dpred (n, nc, P): zeroed by the caller.
extern "C" int det_mask_loss_bwd(const void* pred, int n, int num_classes, int P, const float* target, const int64_t* labels,
                                 const uint8_t* valid, const float* out2, const float* grad_out, void* dpred, int dtype, void* stream) {
    if (!pred || !target || !labels || !valid || !out2 || !grad_out || !dpred || n <= 0 || num_classes <= 0 || P <= 0)
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, (mask_loss_bwd_kernel<float><<<n, 256, 0, s>>>((const float*)pred, n, num_classes, P, target, labels, valid, out2, grad_out, (float*)dpred)),
               (mask_loss_bwd_kernel<bf16><<<n, 256, 0, s>>>((const bf16*)pred, n, num_classes, P, target, labels, valid, out2, grad_out, (bf16*)dpred)))
    return swin_launch_status();
}

static int rpn_flatten_launch(const void* const* ys, void* const* dys, const int* hw, int L, int B, int A, int CH, void* cls_all,
                              void* reg_all, int dtype, bool bwd, void* stream) {
    if (!hw || L <= 0 || L > 8 || B <= 0 || A <= 0 || CH < 5 * A || !cls_all || !reg_all || (!bwd && !ys) || (bwd && !dys))
        return SWIN_ERR_BAD_ARG;
    FlatLevels F;
    int off = 0;
    for (int l = 0; l < 8; ++l) {
        F.y[l] = (!bwd && l < L) ? ys[l] : nullptr;
        F.dy[l] = (bwd && l < L) ? dys[l] : nullptr;
        F.hw[l] = l < L ? hw[l] : 0;
        F.off[l] = off;
        if (l < L) { if (hw[l] <= 0 || (!bwd && !ys[l]) || (bwd && !dys[l])) return SWIN_ERR_BAD_ARG; off += hw[l]; }
    }
    F.L = L;
    const int64_t total = (int64_t)B * off * CH, total_anchors = (int64_t)off * A;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_F32) {
        if (bwd) rpn_flatten_kernel<float, true><<<blocks, 256, 0, s>>>(F, B, A, CH, total_anchors, (float*)cls_all, (float*)reg_all, total);
        else rpn_flatten_kernel<float, false><<<blocks, 256, 0, s>>>(F, B, A, CH, total_anchors, (float*)cls_all, (float*)reg_all, total);
    } else if (dtype == SWIN_BF16) {
        if (bwd) rpn_flatten_kernel<bf16, true><<<blocks, 256, 0, s>>>(F, B, A, CH, total_anchors, (bf16*)cls_all, (bf16*)reg_all, total);
        else rpn_flatten_kernel<bf16, false><<<blocks, 256, 0, s>>>(F, B, A, CH, total_anchors, (bf16*)cls_all, (bf16*)reg_all, total);
    } else return SWIN_ERR_UNSUPPORTED;
    return swin_launch_status();
}

// ys: HOST array of L (<= 8) device pointers to the per-level (B, hw[l], CH) head outputs; cls_all (B, sum hw*A),
// reg_all (B, sum hw*A, 4) out.
extern "C" int det_rpn_flatten_fwd(const void* const* ys, const int* hw, int L, int B, int A, int CH, void* cls_all, void* reg_all,
                                   int dtype, void* stream) {
    return rpn_flatten_launch(ys, nullptr, hw, L, B, A, CH, cls_all, reg_all, dtype, false, stream);
}

// dys: HOST array of L device pointers, fully written (padding channels zero); dcls_all / dreg_all in.
extern "C" int det_rpn_flatten_bwd(void* const* dys, const int* hw, int L, int B, int A, int CH, const void* dcls_all,
                                   const void* dreg_all, int dtype, void* stream) {
    return rpn_flatten_launch(nullptr, dys, hw, L, B, A, CH, const_cast<void*>(dcls_all), const_cast<void*>(dreg_all), dtype, true,
                              stream);
}
