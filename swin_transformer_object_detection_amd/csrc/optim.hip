// AdamW step over ALL parameters in one launch, with the bf16 operand copy ("shadow") of the GEMM/conv weights written
// in the same pass.  Replaces torch.optim.AdamW.step (the optimizer of configs/swin/*_coco.py:64-67, built by
// mmcv's DefaultOptimizerConstructor) followed by the master->half copy of apex O1 (mmdet/apis/train.py:82-89).
// torch's fused AdamW packs tensor lists into kernel arguments every step (8 launches + host packing for this model);
// here the segment table lives in device memory (parameter, gradient-bucket and state pointers are stable), so a step
// is one launch and no host work.  HBM-bound: 16 B read + 12 B (+2 B shadow) written per parameter.
#include "common.h"

struct AdamSeg {            // one parameter tensor
    float* p; const float* g; float* m; float* v; bf16* shadow;   // shadow may be null
    int64_t n; int group; int pad;
};
struct AdamHyper { float lr[8], wd[8]; };

#define ADAM_CHUNK 4096     // elements per block

__global__ __launch_bounds__(256) void adamw_kernel(const AdamSeg* __restrict__ segs, const int2* __restrict__ chunks,
                                                    AdamHyper hp, float beta1, float beta2, float eps, float bc1,
                                                    float bc2_sqrt) {
    const int2 ck = chunks[blockIdx.x];                 // (segment, chunk index within the segment)
    const AdamSeg s = segs[ck.x];
    const int64_t base = (int64_t)ck.y * ADAM_CHUNK;
    const float lr = hp.lr[s.group], wd = hp.wd[s.group];
    const float step_size = lr / bc1;
    const float decay = 1.f - lr * wd;
#pragma unroll 4
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i >= s.n) break;
        const float g = s.g[i];
        float p = s.p[i] * decay;                        // decoupled weight decay
        const float m = beta1 * s.m[i] + (1.f - beta1) * g;
        const float v = beta2 * s.v[i] + (1.f - beta2) * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + eps;
        p -= step_size * (m / denom);
        s.p[i] = p; s.m[i] = m; s.v[i] = v;
        if (s.shadow) s.shadow[i] = (bf16)p;
    }
}

// segs: device array of AdamSeg (layout above: 5 pointers, int64 n, int32 group, int32 pad = 56 bytes);
// chunks: device array of (segment, chunk) int32 pairs, one per 4096 elements; lr / wd: host arrays, n_groups <= 8;
// bias_correction1 = 1 - beta1^t, bias_correction2 = 1 - beta2^t for the step count t kept by the caller.
extern "C" int swin_adamw_step(const void* segs, const void* chunks, int n_chunks, const float* lr, const float* weight_decay,
                               int n_groups, float beta1, float beta2, float eps, float bias_correction1,
                               float bias_correction2, void* stream) {
    if (n_chunks == 0) return SWIN_OK;
    if (!segs || !chunks || n_chunks < 0 || !lr || !weight_decay || n_groups <= 0 || n_groups > 8 || bias_correction1 <= 0.f ||
        bias_correction2 <= 0.f)
        return SWIN_ERR_BAD_ARG;
    AdamHyper hp;
    for (int i = 0; i < 8; ++i) { hp.lr[i] = i < n_groups ? lr[i] : 0.f; hp.wd[i] = i < n_groups ? weight_decay[i] : 0.f; }
    adamw_kernel<<<n_chunks, 256, 0, (hipStream_t)stream>>>((const AdamSeg*)segs, (const int2*)chunks, hp, beta1, beta2, eps,
                                                            bias_correction1, sqrtf(bias_correction2));
    return swin_launch_status();
}

extern "C" int swin_adamw_chunk_elems(void) { return ADAM_CHUNK; }
