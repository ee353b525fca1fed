// Test-time mask pasting for gfx950: FCNMaskHead.get_seg_masks + _do_paste_mask
// (mmdet/models/roi_heads/mask_heads/fcn_mask_head.py:169-300 and :303-377 of the reference) in ONE kernel:
// sigmoid of the class channel, bilinear resampling of the 28x28 mask into the detection box on the image grid
// (F.grid_sample, align_corners=False, zero padding) and the `>= mask_thr_binary` test, written as one byte per
// pixel.  The reference materialises N x img_h x img_w floats per chunk (100 detections on 800x1280: 410 MB) and
// thresholds them in a second pass; here nothing but the final (N, img_h, img_w) uint8 leaves the chip.
#include "common.h"
#pragma clang fp contract(off)

template <typename T>
__global__ __launch_bounds__(256) void paste_masks_kernel(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          const float4* __restrict__ boxes, int N, int num_classes, int mh,
                                                          int mw, int img_h, int img_w, float thr, int is_prob,
                                                          uint8_t* __restrict__ out) {
    const int n = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= img_w) return;
    const float4 b = boxes[n];
    const int64_t lab = labels[n];
    const T* m = logits + ((int64_t)n * num_classes + lab) * mh * mw;
    // img_x = (x + 0.5 - x0) / (x1 - x0) * 2 - 1 ; infinities (degenerate boxes) -> 0   (:350-362)
    float gx = ((float)x + 0.5f - b.x) / (b.z - b.x) * 2.f - 1.f;
    float gy = ((float)y + 0.5f - b.y) / (b.w - b.y) * 2.f - 1.f;
    if (isinf(gx)) gx = 0.f;
    if (isinf(gy)) gy = 0.f;
    // grid_sample, align_corners=False: pixel coordinate = ((g + 1) * size - 1) / 2
    const float ix = ((gx + 1.f) * (float)mw - 1.f) / 2.f;
    const float iy = ((gy + 1.f) * (float)mh - 1.f) / 2.f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
    float v = 0.f;
    if (!(isnan(ix) || isnan(iy))) {
        auto tap = [&](int yy, int xx) -> float {
            if (yy < 0 || yy >= mh || xx < 0 || xx >= mw) return 0.f;
            const float l = Elt<T>::ld(m + yy * mw + xx);
            return is_prob ? l : 1.f / (1.f + expf(-l));
        };
        // accumulation order of grid_sample's bilinear kernel: nw, ne, sw, se
        v = tap(y0, x0) * (wx0 * wy0);
        v += tap(y0, x1) * (wx1 * wy0);
        v += tap(y1, x0) * (wx0 * wy1);
        v += tap(y1, x1) * (wx1 * wy1);
    }
    out[((int64_t)n * img_h + y) * img_w + x] = (uint8_t)(v >= thr ? 1 : 0);
}

// mask_logits (N, num_classes, mh, mw) f32 or bf16; labels (N) i64; boxes (N,4) f32 in output-image coordinates;
// out (N, img_h, img_w) u8 = sigmoid-mask resampled into the box >= thr.
extern "C" int det_paste_masks(const void* mask_logits, const int64_t* labels, const float* boxes, int N, int num_classes,
                               int mh, int mw, int img_h, int img_w, float thr, int is_prob, int in_dtype, uint8_t* out, void* stream) {
    if (N == 0) return SWIN_OK;
    if (!mask_logits || !labels || !boxes || !out || N < 0 || num_classes <= 0 || mh <= 0 || mw <= 0 || img_h <= 0 || img_w <= 0)
        return SWIN_ERR_BAD_ARG;
    if (img_h > 65535 || N > 65535) return SWIN_ERR_UNSUPPORTED;
    dim3 grid((img_w + 255) / 256, img_h, N);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == SWIN_F32)
        paste_masks_kernel<float><<<grid, 256, 0, s>>>((const float*)mask_logits, labels, (const float4*)boxes, N, num_classes, mh,
                                                       mw, img_h, img_w, thr, is_prob, out);
    else if (in_dtype == SWIN_BF16)
        paste_masks_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)mask_logits, labels, (const float4*)boxes, N, num_classes, mh, mw,
                                                      img_h, img_w, thr, is_prob, out);
    else return SWIN_ERR_UNSUPPORTED;
    return swin_launch_status();
}
