// Loss kernels of the detector heads for gfx950, forward value + input gradients, fixed-size samples.
// Reference arithmetic:
//   RPN:   AnchorHead.loss_single (mmdet/models/dense_heads/anchor_head.py:375-434): sigmoid cross entropy over the
//          sampled anchors + L1 / SmoothL1(beta) on the positives' deltas, both divided by the number of sampled
//          anchors of the batch;
//   bbox:  BBoxHead.loss (mmdet/models/roi_heads/bbox_heads/bbox_head.py:188-238): softmax cross entropy (avg_factor =
//          number of sampled RoIs), accuracy, and on the positives -- divided by the number of samples -- class-specific
//          or class-agnostic L1 / SmoothL1 on the deltas, or GIoU on the decoded boxes (reg_decoded_bbox, Cascade configs);
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

// L1Loss / SmoothL1Loss element (losses/smooth_l1_loss.py:10-28, :31-45): beta <= 0 selects plain L1.
__device__ __forceinline__ float reg_elem_loss(float d, float beta) {
    const float a = fabsf(d);
    return a < beta ? 0.5f * a * a / beta : a - 0.5f * beta;
}
__device__ __forceinline__ float reg_elem_grad(float d, float beta) {
    if (fabsf(d) < beta) return d / beta;
    return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
}

// Regression term of BBoxHead.loss.  mode 0: L1 / SmoothL1(beta) on the encoded deltas; mode 2: GIoULoss on the DECODED
// boxes (reg_decoded_bbox=True, bbox_head.py:215-216 -> DeltaXYWHBBoxCoder.decode without max_shape; iou_loss.py:78-101 ->
// bbox_overlaps(mode='giou', is_aligned=True), iou2d_calculator.py:108-158).
struct RegCfg { int mode; int agnostic; float beta; float eps; float means[4]; float stds[4]; float max_ratio; };

// share of the gradient that max(a,b) / min(a,b) sends to a (torch.maximum / minimum split ties evenly)
__device__ __forceinline__ float sel_max(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float sel_min(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// decoded box o[4] of roi r[4] and raw deltas dl[4]; jac[4]: d(center)/d(dl0,dl1) and d(size)/d(dl2,dl3)
__device__ __forceinline__ void decode_box(const float* r, const float* dl, const RegCfg& c, float* o, float* jac) {
    const float dx = dl[0] * c.stds[0] + c.means[0], dy = dl[1] * c.stds[1] + c.means[1];
    const float dw0 = dl[2] * c.stds[2] + c.means[2], dh0 = dl[3] * c.stds[3] + c.means[3];
    const float dw = fminf(fmaxf(dw0, -c.max_ratio), c.max_ratio), dh = fminf(fmaxf(dh0, -c.max_ratio), c.max_ratio);
    const float px = (r[0] + r[2]) * 0.5f, py = (r[1] + r[3]) * 0.5f, pw = r[2] - r[0], ph = r[3] - r[1];
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = px + pw * dx, gy = py + ph * dy;
    o[0] = gx - gw * 0.5f; o[1] = gy - gh * 0.5f; o[2] = gx + gw * 0.5f; o[3] = gy + gh * 0.5f;
    if (jac) {
        jac[0] = pw * c.stds[0]; jac[1] = ph * c.stds[1];
        jac[2] = (dw0 >= -c.max_ratio && dw0 <= c.max_ratio) ? gw * c.stds[2] : 0.f;     // clamp passes gradient inside [min,max]
        jac[3] = (dh0 >= -c.max_ratio && dh0 <= c.max_ratio) ? gh * c.stds[3] : 0.f;
    }
}

// 1 - GIoU(p, t); g[4] (optional) = d loss / d p
__device__ __forceinline__ float giou_loss(const float* p, const float* t, float eps, float* g) {
    const float pw = p[2] - p[0], ph = p[3] - p[1];
    const float area1 = pw * ph, area2 = (t[2] - t[0]) * (t[3] - t[1]);
    const float w0 = fminf(p[2], t[2]) - fmaxf(p[0], t[0]), h0 = fminf(p[3], t[3]) - fmaxf(p[1], t[1]);
    const float w = fmaxf(w0, 0.f), h = fmaxf(h0, 0.f);
    const float ov = w * h;
    const float un0 = area1 + area2 - ov, un = fmaxf(un0, eps);
    const float iou = ov / un;
    const float ew0 = fmaxf(p[2], t[2]) - fminf(p[0], t[0]), eh0 = fmaxf(p[3], t[3]) - fminf(p[1], t[1]);
    const float ew = fmaxf(ew0, 0.f), eh = fmaxf(eh0, 0.f);
    const float ea0 = ew * eh, ea = fmaxf(ea0, eps);
    const float giou = iou - (ea - un) / ea;
    if (g) {
        // loss = 1 - ov/un + (ea - un)/ea
        const float d_un = (ov / (un * un) - 1.f / ea) * sel_max(un0, eps);
        const float d_ea = (un / (ea * ea)) * sel_max(ea0, eps);
        const float d_ov = -1.f / un - d_un;                      // un0 = area1 + area2 - ov
        const float d_w = (w0 >= 0.f) ? d_ov * h : 0.f, d_h = (h0 >= 0.f) ? d_ov * w : 0.f;
        const float d_ew = (ew0 >= 0.f) ? d_ea * eh : 0.f, d_eh = (eh0 >= 0.f) ? d_ea * ew : 0.f;
        g[0] = -d_un * ph - d_w * sel_max(p[0], t[0]) - d_ew * sel_min(p[0], t[0]);
        g[1] = -d_un * pw - d_h * sel_max(p[1], t[1]) - d_eh * sel_min(p[1], t[1]);
        g[2] = d_un * ph + d_w * sel_min(p[2], t[2]) + d_ew * sel_max(p[2], t[2]);
        g[3] = d_un * pw + d_h * sel_min(p[3], t[3]) + d_eh * sel_max(p[3], t[3]);
    }
    return 1.f - giou;
}

// ------------------------------------------------------------------------------------------------ RPN
// cls (B, A) logits, reg (B, A, 4) deltas; per image S sample slots: inds, flags (bit 0 used, bit 1 positive), tgt (S,4).
// out[0] = sum BCE / n, out[1] = sum L1 / n, out[2] = n (sampled anchors of the whole batch, >= 1).
template <typename T>
__global__ __launch_bounds__(1024) void rpn_loss_fwd_kernel(const T* __restrict__ cls, const T* __restrict__ reg, int B, int64_t A,
                                                            int S, const int64_t* __restrict__ inds,
                                                            const uint8_t* __restrict__ flags, const float* __restrict__ tgt,
                                                            float beta, float* __restrict__ out) {
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
            for (int k = 0; k < 4; ++k) lb += reg_elem_loss(Elt<T>::ld(r + k) - tgt[(int64_t)i * 4 + k], beta);
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
                                                           float beta, T* __restrict__ dcls, T* __restrict__ dreg) {
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
            Elt<T>::st(dreg + (b * A + a) * 4 + k, reg_elem_grad(d, beta) * g1);
        }
    }
}

// ------------------------------------------------------------------------------------------------ bbox head
// cls (n, nc+1) logits, bbox (n, 4 nc) class-specific or (n, 4) class-agnostic deltas, labels (n) in [0, nc] (nc =
// background), tgt (n,4) (encoded deltas, or gt boxes in GIoU mode), rois (n,4) (GIoU mode only), flags (n) (bit 0 used,
// bit 1 positive).  out[0] = CE sum / nv, out[1] = accuracy (%), out[2] = regression sum / nv, out[3] = nv.
// One WAVE per row, 16 rows per block: the lanes read the row's logits coalesced and reduce max / arg-max / sum with shuffles; a
// block leaves the sums of its 16 rows in part[block][4] and bbox_loss_final_kernel adds the blocks.  (History: one thread per row in
// ONE block -- 81 strided scalar loads and expf's in a serial loop -- 84 us; one wave per row in one block, 64 rows per wave each a
// chain of dependent global loads -- 140 us.  The work is 83 K logits.)
template <typename T>
__global__ __launch_bounds__(1024) void bbox_loss_rows_kernel(const T* __restrict__ cls, const T* __restrict__ bbox, int n, int nc,
                                                              const int64_t* __restrict__ labels, const float* __restrict__ tgt,
                                                              const uint8_t* __restrict__ flags, const float* __restrict__ rois,
                                                              RegCfg rc, float* __restrict__ part, float* __restrict__ lse) {
    __shared__ float red[16];
    float ce = 0.f, hit = 0.f, l1 = 0.f, cnt = 0.f;
    const int C = nc + 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 16 + wv;
    if (i < n) {
        const T* row = cls + (int64_t)i * C;
        float m = -3.0e38f; int am = 0x7fffffff;
        for (int c = lane; c < C; c += 64) { const float v = Elt<T>::ld(row + c); if (v > m) { m = v; am = c; } }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                       // max with the LOWEST index among equal maxima (as a serial scan)
            const float m2 = __shfl_xor(m, o); const int a2 = __shfl_xor(am, o);
            if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
        }
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(Elt<T>::ld(row + c) - m);
        s = wave_sum(s);
        const float l = m + logf(s);
        const uint8_t f = flags[i];
        if (lane == 0) {
            lse[i] = l;
            if (f & 1) {
                const int lab = (int)labels[i];
                ce = l - Elt<T>::ld(row + lab);
                hit = (am == lab) ? 1.f : 0.f;
                cnt = 1.f;
                if ((f & 2) && lab < nc) {
                    const T* p = rc.agnostic ? bbox + (int64_t)i * 4 : bbox + ((int64_t)i * nc + lab) * 4;
                    float dl[4], t4[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { dl[k] = Elt<T>::ld(p + k); t4[k] = tgt[(int64_t)i * 4 + k]; }
                    if (rc.mode == 2) {
                        float r4[4], o[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) r4[k] = rois[(int64_t)i * 4 + k];
                        decode_box(r4, dl, rc, o, nullptr);
                        l1 = giou_loss(o, t4, rc.eps, nullptr);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) l1 += reg_elem_loss(dl[k] - t4[k], rc.beta);
                    }
                }
            }
        }
    }
    ce = block_sum(ce, red); hit = block_sum(hit, red); l1 = block_sum(l1, red); cnt = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        float* o = part + (int64_t)blockIdx.x * 4;
        o[0] = ce; o[1] = hit; o[2] = l1; o[3] = cnt;
    }
}

// out[0] = CE sum / nv, out[1] = accuracy (%), out[2] = regression sum / nv, out[3] = nv from the blocks' partial sums (fixed order:
// the result does not depend on the launch's scheduling)
__global__ __launch_bounds__(256) void bbox_loss_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ float red[16];
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += part[(int64_t)b * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = block_sum(a[k], red);
    if (threadIdx.x == 0) {
        const float nv = fmaxf(a[3], 1.f);
        out[0] = a[0] / nv; out[1] = a[1] / nv * 100.f; out[2] = a[2] / nv; out[3] = nv;
    }
}

// one thread per (row, column) of dcls and of dbbox: every element is written (zeros included)
template <typename T>
__global__ __launch_bounds__(256) void bbox_loss_bwd_kernel(const T* __restrict__ cls, const T* __restrict__ bbox, int n, int nc,
                                                            const int64_t* __restrict__ labels, const float* __restrict__ tgt,
                                                            const uint8_t* __restrict__ flags, const float* __restrict__ rois,
                                                            RegCfg rc, const float* __restrict__ out,
                                                            const float* __restrict__ lse, const float* __restrict__ gout,
                                                            T* __restrict__ dcls, T* __restrict__ dbbox) {
    const int C = nc + 1, RW = rc.agnostic ? 4 : 4 * nc, W = C + RW;
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
        if ((f & 2) && lab < nc && (rc.agnostic || c == lab)) {
            const T* p = bbox + (int64_t)i * RW + (q - k);
            if (rc.mode == 2) {
                float dl[4], t4[4], r4[4], o[4], jac[4], gb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { dl[j] = Elt<T>::ld(p + j); t4[j] = tgt[(int64_t)i * 4 + j]; r4[j] = rois[(int64_t)i * 4 + j]; }
                decode_box(r4, dl, rc, o, jac);
                giou_loss(o, t4, rc.eps, gb);
                // o = (gx - gw/2, gy - gh/2, gx + gw/2, gy + gh/2)
                const float gk = k == 0 ? (gb[0] + gb[2]) * jac[0] : k == 1 ? (gb[1] + gb[3]) * jac[1]
                               : k == 2 ? 0.5f * (gb[2] - gb[0]) * jac[2] : 0.5f * (gb[3] - gb[1]) * jac[3];
                g = gk * gout[2] * inv;
            } else {
                g = reg_elem_grad(Elt<T>::ld(p + k) - tgt[(int64_t)i * 4 + k], rc.beta) * gout[2] * inv;
            }
        }
        Elt<T>::st(dbbox + (int64_t)i * RW + q, g);
    }
}

// ------------------------------------------------------------------------------------------------ mask head
// pred (n, nc, P) logits (P = 28*28), target (n, P) in {0,1}, labels (n), valid (n) u8.
// out[0] = sum_valid mean_P BCE / max(#valid,1), out[1] = max(#valid,1).  grid = n blocks.
// Element (i, c, pixel k) of the mask logits.  pw == 0: (n, nc, P) NCHW.  pw > 0 ("deconv order", pw = width of the
// 2H x 2W mask): the rows of the ConvTranspose2d(k=2,s=2)-as-GEMM output, (roi, h, w, ky, kx) with the class innermost --
// what FCNMaskHead produces BEFORE the 2x2 pixel shuffle, so the training path needs neither the shuffle copy of the
// 256-channel activation nor the NHWC->NCHW copy of the logits (fcn_mask_head.py:117-126 computes the same values).
__device__ __forceinline__ int64_t mask_elem(int i, int c, int k, int nc, int P, int pw) {
    if (pw == 0) return ((int64_t)i * nc + c) * P + k;
    const int Y = k / pw, X = k - Y * pw, W = pw >> 1;
    return (((int64_t)i * (P >> 2) + (Y >> 1) * W + (X >> 1)) * 4 + (Y & 1) * 2 + (X & 1)) * nc + c;
}

template <typename T>
__global__ __launch_bounds__(256) void mask_loss_fwd_kernel(const T* __restrict__ pred, int n, int nc, int P, int pw,
                                                            const float* __restrict__ target, const int64_t* __restrict__ labels,
                                                            const uint8_t* __restrict__ valid, float* __restrict__ per_roi) {
    __shared__ float red[16];
    const int i = blockIdx.x;
    float s = 0.f;
    if (valid[i]) {
        const int lab = (int)labels[i];
        const float* t = target + (int64_t)i * P;
        for (int k = threadIdx.x; k < P; k += 256) s += bce_logits(Elt<T>::ld(pred + mask_elem(i, lab, k, nc, P, pw)), t[k]);
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
__global__ __launch_bounds__(256) void mask_loss_bwd_kernel(const T* __restrict__ pred, int n, int nc, int P, int pw,
                                                            const float* __restrict__ target, const int64_t* __restrict__ labels,
                                                            const uint8_t* __restrict__ valid, const float* __restrict__ out,
                                                            const float* __restrict__ gout, T* __restrict__ dpred) {
    const int i = blockIdx.x;
    if (!valid[i]) return;
    const float g = gout[0] / (out[1] * (float)P);
    const int lab = (int)labels[i];
    const float* t = target + (int64_t)i * P;
    for (int k = threadIdx.x; k < P; k += 256) {
        const int64_t e = mask_elem(i, lab, k, nc, P, pw);
        Elt<T>::st(dpred + e, (sigmoidf(Elt<T>::ld(pred + e)) - t[k]) * g);
    }
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
                                const float* targets, float beta, float* out3, int dtype, void* stream) {
    if (!cls || !reg || !inds || !flags || !targets || !out3 || B <= 0 || A <= 0 || S <= 0 || !(beta >= 0.f)) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, (rpn_loss_fwd_kernel<float><<<1, 1024, 0, s>>>((const float*)cls, (const float*)reg, B, A, S, inds, flags, targets, beta, out3)),
               (rpn_loss_fwd_kernel<bf16><<<1, 1024, 0, s>>>((const bf16*)cls, (const bf16*)reg, B, A, S, inds, flags, targets, beta, out3)))
    return swin_launch_status();
}

// dcls (B,A) / dreg (B,A,4): zeroed by the caller; grad_out: 2 floats (d loss_cls, d loss_bbox); out3 from the forward.
extern "C" int det_rpn_loss_bwd(const void* cls, const void* reg, int B, int64_t A, int S, const int64_t* inds, const uint8_t* flags,
                                const float* targets, float beta, const float* out3, const float* grad_out, void* dcls, void* dreg,
                                int dtype, void* stream) {
    if (!cls || !reg || !inds || !flags || !targets || !out3 || !grad_out || !dcls || !dreg || B <= 0 || A <= 0 || S <= 0 ||
        !(beta >= 0.f))
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (B * S + 255) / 256;
    DISPATCH_T(dtype, (rpn_loss_bwd_kernel<float><<<blocks, 256, 0, s>>>((const float*)cls, (const float*)reg, B, A, S, inds, flags, targets, out3, grad_out, beta, (float*)dcls, (float*)dreg)),
               (rpn_loss_bwd_kernel<bf16><<<blocks, 256, 0, s>>>((const bf16*)cls, (const bf16*)reg, B, A, S, inds, flags, targets, out3, grad_out, beta, (bf16*)dcls, (bf16*)dreg)))
    return swin_launch_status();
}

// reg_mode 0: L1 (beta 0) / SmoothL1 (beta > 0) on encoded deltas; 2: GIoU on decoded boxes (rois, means, stds, eps used).
static int reg_cfg(RegCfg& rc, int reg_mode, int class_agnostic, float beta, float eps, const float* rois, const float* means,
                   const float* stds) {
    if (reg_mode != 0 && reg_mode != 2) return SWIN_ERR_UNSUPPORTED;
    if (!(beta >= 0.f)) return SWIN_ERR_BAD_ARG;
    rc.mode = reg_mode; rc.agnostic = class_agnostic ? 1 : 0; rc.beta = beta; rc.eps = eps;
    rc.max_ratio = fabsf(logf(16.f / 1000.f));                     // DeltaXYWHBBoxCoder default wh_ratio_clip
    for (int q = 0; q < 4; ++q) { rc.means[q] = 0.f; rc.stds[q] = 1.f; }
    if (reg_mode == 2) {
        if (!rois || !means || !stds || !(eps > 0.f)) return SWIN_ERR_BAD_ARG;
        for (int q = 0; q < 4; ++q) { rc.means[q] = means[q]; rc.stds[q] = stds[q]; }
    }
    return SWIN_OK;
}

// out4: loss_cls, accuracy (%), loss_bbox, n_valid; lse: n + 4 * ceil(n / 16) floats -- the first n (the rows' log-sum-exp) are kept
// for the backward, the rest is this call's scratch (one partial-sum row per thread block).
extern "C" int det_bbox_loss_fwd(const void* cls, const void* bbox, int n, int num_classes, const int64_t* labels, const float* targets,
                                 const uint8_t* flags, int reg_mode, int class_agnostic, float beta, float eps, const float* rois,
                                 const float* means, const float* stds, float* out4, float* lse, int dtype, void* stream) {
    if (!cls || !bbox || !labels || !targets || !flags || !out4 || !lse || n <= 0 || num_classes <= 0) return SWIN_ERR_BAD_ARG;
    RegCfg rc;
    const int st = reg_cfg(rc, reg_mode, class_agnostic, beta, eps, rois, means, stds);
    if (st != SWIN_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (n + 15) / 16;
    float* workspace = lse + n;
    DISPATCH_T(dtype, (bbox_loss_rows_kernel<float><<<nblk, 1024, 0, s>>>((const float*)cls, (const float*)bbox, n, num_classes, labels, targets, flags, rois, rc, workspace, lse)),
               (bbox_loss_rows_kernel<bf16><<<nblk, 1024, 0, s>>>((const bf16*)cls, (const bf16*)bbox, n, num_classes, labels, targets, flags, rois, rc, workspace, lse)))
    bbox_loss_final_kernel<<<1, 256, 0, s>>>(workspace, nblk, out4);
    return swin_launch_status();
}

// grad_out: 4 floats aligned with out4 (entries 0 and 2 are used); dcls (n, nc+1), dbbox (n, 4 nc | 4): fully written.
extern "C" int det_bbox_loss_bwd(const void* cls, const void* bbox, int n, int num_classes, const int64_t* labels, const float* targets,
                                 const uint8_t* flags, int reg_mode, int class_agnostic, float beta, float eps, const float* rois,
                                 const float* means, const float* stds, const float* out4, const float* lse, const float* grad_out,
                                 void* dcls, void* dbbox, int dtype, void* stream) {
    if (!cls || !bbox || !labels || !targets || !flags || !out4 || !lse || !grad_out || !dcls || !dbbox || n <= 0 || num_classes <= 0)
        return SWIN_ERR_BAD_ARG;
    RegCfg rc;
    const int st = reg_cfg(rc, reg_mode, class_agnostic, beta, eps, rois, means, stds);
    if (st != SWIN_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)n * (num_classes + 1 + (class_agnostic ? 4 : 4 * num_classes));
    const int blocks = (int)((total + 255) / 256);
    DISPATCH_T(dtype, (bbox_loss_bwd_kernel<float><<<blocks, 256, 0, s>>>((const float*)cls, (const float*)bbox, n, num_classes, labels, targets, flags, rois, rc, out4, lse, grad_out, (float*)dcls, (float*)dbbox)),
               (bbox_loss_bwd_kernel<bf16><<<blocks, 256, 0, s>>>((const bf16*)cls, (const bf16*)bbox, n, num_classes, labels, targets, flags, rois, rc, out4, lse, grad_out, (bf16*)dcls, (bf16*)dbbox)))
    return swin_launch_status();
}

// out2: loss, n_valid; per_roi (n) f32 scratch.  deconv_w: 0 = pred is (n, nc, P) NCHW; > 0 = "deconv order" rows with the
// class innermost, deconv_w = width of the square mask (P == deconv_w^2, even) -- see mask_elem.
extern "C" int det_mask_loss_fwd(const void* pred, int n, int num_classes, int P, int deconv_w, const float* target,
                                 const int64_t* labels, const uint8_t* valid, float* out2, float* per_roi, int dtype, void* stream) {
    if (!pred || !target || !labels || !valid || !out2 || !per_roi || n <= 0 || num_classes <= 0 || P <= 0) return SWIN_ERR_BAD_ARG;
    if (deconv_w != 0 && (deconv_w < 2 || (deconv_w & 1) || deconv_w * deconv_w != P)) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, (mask_loss_fwd_kernel<float><<<n, 256, 0, s>>>((const float*)pred, n, num_classes, P, deconv_w, target, labels, valid, per_roi)),
               (mask_loss_fwd_kernel<bf16><<<n, 256, 0, s>>>((const bf16*)pred, n, num_classes, P, deconv_w, target, labels, valid, per_roi)))
    mask_loss_final_kernel<<<1, 1024, 0, s>>>(per_roi, valid, n, out2);
    return swin_launch_status();
}

// dpred (same layout as pred): zeroed by the caller.
extern "C" int det_mask_loss_bwd(const void* pred, int n, int num_classes, int P, int deconv_w, const float* target,
                                 const int64_t* labels, const uint8_t* valid, const float* out2, const float* grad_out, void* dpred,
                                 int dtype, void* stream) {
    if (!pred || !target || !labels || !valid || !out2 || !grad_out || !dpred || n <= 0 || num_classes <= 0 || P <= 0)
        return SWIN_ERR_BAD_ARG;
    if (deconv_w != 0 && (deconv_w < 2 || (deconv_w & 1) || deconv_w * deconv_w != P)) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype, (mask_loss_bwd_kernel<float><<<n, 256, 0, s>>>((const float*)pred, n, num_classes, P, deconv_w, target, labels, valid, out2, grad_out, (float*)dpred)),
               (mask_loss_bwd_kernel<bf16><<<n, 256, 0, s>>>((const bf16*)pred, n, num_classes, P, deconv_w, target, labels, valid, out2, grad_out, (bf16*)dpred)))
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
