/* CPU restatement of the mmcv detection ops on the Swin hot path
 * -- TEST INFRASTRUCTURE ONLY (checker + CPU baseline; never shipped).
 *
 * PARITY UNPINNED: the arithmetic belongs to mmcv-full (pin 1.2.4 <= mmcv <= 1.4.0,
 * reference mmdet/__init__.py:18-19), whose sources are not in the reference tree.
 * This file restates mmcv's published algorithms as they are used at the
 * reference's call sites:
 *   RoIAlign      mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:49-55,
 *                 mmdet/core/mask/structures.py:353-354   (pool 'avg', aligned=True)
 *   nms           mmdet/models/dense_heads/rpn_head.py:233,
 *                 mmdet/core/post_processing/bbox_nms.py:84 (through batched_nms)
 * Build spec fixed by this project: stable descending sort
 * (ties -> lower index first), IoU test `inter > thr * (Sa + Sb - inter)` in fp32
 * in exactly this operation order.
 *
 * Plain C99, single thread.  Layout: NCHW fp32 features, rois (K,5) =
 * [batch_idx, x1, y1, x2, y2].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- bilinear sample with mmcv's border rules ---------------------------- */
static void bilinear_setup(int H, int W, float y, float x, int* yl, int* xl, int* yh, int* xh,
                           float* w1, float* w2, float* w3, float* w4, int* valid) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
        *valid = 0;
        *w1 = *w2 = *w3 = *w4 = 0.f;
        *yl = *xl = *yh = *xh = -1;
        return;
    }
    *valid = 1;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
    float ly = y - (float)y_low, lx = x - (float)x_low;
    float hy = 1.f - ly, hx = 1.f - lx;
    *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
    *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;
}

static void roi_geometry(const float* roi, float scale, int aligned, int ph, int pw, int sampling_ratio,
                         float* sw, float* sh, float* bw, float* bh, int* gh, int* gw) {
    float off = aligned ? 0.5f : 0.f;
    float roi_start_w = roi[1] * scale - off;
    float roi_start_h = roi[2] * scale - off;
    float roi_end_w = roi[3] * scale - off;
    float roi_end_h = roi[4] * scale - off;
    float roi_w = roi_end_w - roi_start_w;
    float roi_h = roi_end_h - roi_start_h;
    if (!aligned) {
        roi_w = fmaxf(roi_w, 1.f);
        roi_h = fmaxf(roi_h, 1.f);
    }
    *bh = roi_h / (float)ph;
    *bw = roi_w / (float)pw;
    *gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)ph);
    *gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)pw);
    *sw = roi_start_w;
    *sh = roi_start_h;
}

/* out: (K, C, ph, pw) */
void roi_align_fwd_ref(const float* input, const float* rois, float* out, int N, int C, int H, int W,
                       int K, int ph, int pw, float scale, int sampling_ratio, int aligned) {
    (void)N;
    for (int k = 0; k < K; ++k) {
        const float* roi = rois + 5 * k;
        int b = (int)roi[0];
        float sw, sh, bw, bh; int gh, gw;
        roi_geometry(roi, scale, aligned, ph, pw, sampling_ratio, &sw, &sh, &bw, &bh, &gh, &gw);
        int cnt_i = gh * gw; if (cnt_i < 1) cnt_i = 1;
        float count = (float)cnt_i;
        for (int c = 0; c < C; ++c) {
            const float* in = input + ((size_t)b * C + c) * H * W;
            for (int i = 0; i < ph; ++i)
                for (int j = 0; j < pw; ++j) {
                    float acc = 0.f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + (float)i * bh + ((float)iy + .5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + (float)j * bw + ((float)ix + .5f) * bw / (float)gw;
                            int yl, xl, yh, xh, valid; float w1, w2, w3, w4;
                            bilinear_setup(H, W, y, x, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4, &valid);
                            if (!valid) continue;
                            acc += w1 * in[yl * W + xl] + w2 * in[yl * W + xh] +
                                   w3 * in[yh * W + xl] + w4 * in[yh * W + xh];
                        }
                    }
                    out[(((size_t)k * C + c) * ph + i) * pw + j] = acc / count;
                }
        }
    }
}

/* grad_in: (N, C, H, W), zero-initialised by the caller; accumulated in double so the
 * oracle's result does not depend on summation order (the device result is compared
 * with a tolerance that covers fp32 atomics). */
void roi_align_bwd_ref(const float* grad_out, const float* rois, double* grad_in, int N, int C, int H, int W,
                       int K, int ph, int pw, float scale, int sampling_ratio, int aligned) {
    (void)N;
    for (int k = 0; k < K; ++k) {
        const float* roi = rois + 5 * k;
        int b = (int)roi[0];
        float sw, sh, bw, bh; int gh, gw;
        roi_geometry(roi, scale, aligned, ph, pw, sampling_ratio, &sw, &sh, &bw, &bh, &gh, &gw);
        int cnt_i = gh * gw; if (cnt_i < 1) cnt_i = 1;
        float count = (float)cnt_i;
        for (int c = 0; c < C; ++c) {
            double* gi = grad_in + ((size_t)b * C + c) * H * W;
            for (int i = 0; i < ph; ++i)
                for (int j = 0; j < pw; ++j) {
                    float g = grad_out[(((size_t)k * C + c) * ph + i) * pw + j];
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = sh + (float)i * bh + ((float)iy + .5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = sw + (float)j * bw + ((float)ix + .5f) * bw / (float)gw;
                            int yl, xl, yh, xh, valid; float w1, w2, w3, w4;
                            bilinear_setup(H, W, y, x, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4, &valid);
                            if (!valid) continue;
                            gi[yl * W + xl] += (double)(g * w1 / count);
                            gi[yl * W + xh] += (double)(g * w2 / count);
                            gi[yh * W + xl] += (double)(g * w3 / count);
                            gi[yh * W + xh] += (double)(g * w4 / count);
                        }
                    }
                }
        }
    }
}

/* ---- nms ----------------------------------------------------------------- */
typedef struct { float s; int64_t i; } sc_t;
static int cmp_desc_stable(const void* a, const void* b) {
    const sc_t* x = (const sc_t*)a; const sc_t* y = (const sc_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);   /* ties: lower index first */
}

static int iou_gt(const float* a, const float* b, float off, float thr) {
    float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
    float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
    float w = fmaxf(right - left + off, 0.f), h = fmaxf(bottom - top + off, 0.f);
    float inter = w * h;
    float sa = (a[2] - a[0] + off) * (a[3] - a[1] + off);
    float sb = (b[2] - b[0] + off) * (b[3] - b[1] + off);
    return inter > thr * (sa + sb - inter);
}

/* keep: capacity n; returns number kept.  keep[] = indices into the input in
 * descending-score order. */
int64_t nms_ref(const float* boxes, const float* scores, int64_t n, float thr, int offset, int64_t* keep) {
    if (n <= 0) return 0;
    sc_t* ord = (sc_t*)malloc(sizeof(sc_t) * (size_t)n);
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) { ord[i].s = scores[i]; ord[i].i = i; }
    qsort(ord, (size_t)n, sizeof(sc_t), cmp_desc_stable);
    int64_t m = 0;
    float off = (float)offset;
    for (int64_t a = 0; a < n; ++a) {
        if (dead[a]) continue;
        int64_t ia = ord[a].i;
        keep[m++] = ia;
        for (int64_t b = a + 1; b < n; ++b) {
            if (dead[b]) continue;
            if (iou_gt(boxes + 4 * ia, boxes + 4 * ord[b].i, off, thr)) dead[b] = 1;
        }
    }
    free(ord); free(dead);
    return m;
}
