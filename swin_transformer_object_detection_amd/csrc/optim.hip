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

// Device-resident step state (swin_adamw_set_state writes it with a one-thread kernel, i.e. from kernel ARGUMENTS: no host->device
// copy, and under hipGraph replay the optimizer launch inside the graph reads this step's values).  32 floats:
//   [0..7] lr per group  [8..15] weight decay per group  [16] bias_correction1  [17] sqrt(bias_correction2)
//   [18] grad_scale: every gradient is multiplied by it (1 / loss scale; 1 without loss scaling)
//   [19] skip: != 0 -> the step leaves parameters and moments untouched (a non-finite gradient under fp16 loss scaling)
//   [20] loss scale  [21] clean steps since the last change of the scale  [22..31] reserved
struct AdamState { float lr[8], wd[8], bc1, bc2_sqrt, grad_scale, skip, loss_scale, good_steps, pad[10]; };

__global__ __launch_bounds__(256) void adamw_kernel(const AdamSeg* __restrict__ segs, const int2* __restrict__ chunks,
                                                    AdamHyper hp, float beta1, float beta2, float eps, float bc1,
                                                    float bc2_sqrt, const AdamState* __restrict__ st) {
    const int2 ck = chunks[blockIdx.x];                 // (segment, chunk index within the segment)
    const AdamSeg s = segs[ck.x];
    const int64_t base = (int64_t)ck.y * ADAM_CHUNK;
    float lr = hp.lr[s.group], wd = hp.wd[s.group], gscale = 1.f;
    if (st) {                                           // this step's values live on the device
        if (st->skip != 0.f) return;
        lr = st->lr[s.group]; wd = st->wd[s.group]; bc1 = st->bc1; bc2_sqrt = st->bc2_sqrt; gscale = st->grad_scale;
    }
    const float step_size = lr / bc1;
    const float decay = 1.f - lr * wd;
#pragma unroll 4
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i >= s.n) break;
        const float g = s.g[i] * gscale;
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
                                                            bias_correction1, sqrtf(bias_correction2), nullptr);
    return swin_launch_status();
}

__global__ void adamw_set_state_kernel(AdamState* st, AdamHyper hp, float bc1, float bc2_sqrt) {
    if (threadIdx.x < 8) { st->lr[threadIdx.x] = hp.lr[threadIdx.x]; st->wd[threadIdx.x] = hp.wd[threadIdx.x]; }
    if (threadIdx.x == 8) { st->bc1 = bc1; st->bc2_sqrt = bc2_sqrt; }
}

// this step's learning rates / weight decays / bias corrections -> the device-resident state (layout above; the loss-scaling
// fields 18..21 are left alone: swin_loss_scale_* own them).  Arguments travel as kernel arguments.
extern "C" int swin_adamw_set_state(void* state, const float* lr, const float* weight_decay, int n_groups, float bias_correction1,
                                    float bias_correction2, void* stream) {
    if (!state || !lr || !weight_decay || n_groups <= 0 || n_groups > 8 || bias_correction1 <= 0.f || bias_correction2 <= 0.f)
        return SWIN_ERR_BAD_ARG;
    AdamHyper hp;
    for (int i = 0; i < 8; ++i) { hp.lr[i] = i < n_groups ? lr[i] : 0.f; hp.wd[i] = i < n_groups ? weight_decay[i] : 0.f; }
    adamw_set_state_kernel<<<1, 64, 0, (hipStream_t)stream>>>((AdamState*)state, hp, bias_correction1, sqrtf(bias_correction2));
    return swin_launch_status();
}

// swin_adamw_step with every per-step scalar read from the device-resident state: the launch's arguments never change, so it can
// sit inside a captured hipGraph.
extern "C" int swin_adamw_step_dev(const void* segs, const void* chunks, int n_chunks, const void* state, float beta1, float beta2,
                                   float eps, void* stream) {
    if (n_chunks == 0) return SWIN_OK;
    if (!segs || !chunks || n_chunks < 0 || !state) return SWIN_ERR_BAD_ARG;
    AdamHyper hp = {};
    adamw_kernel<<<n_chunks, 256, 0, (hipStream_t)stream>>>((const AdamSeg*)segs, (const int2*)chunks, hp, beta1, beta2, eps, 1.f, 1.f,
                                                            (const AdamState*)state);
    return swin_launch_status();
}

__global__ void set_u64_kernel(unsigned long long* p, unsigned long long v) { *p = v; }

// *dst (device, 8-byte aligned) = value, as a kernel launch (no host->device copy: the per-step seed of the samplers)
extern "C" int swin_set_u64(void* dst, uint64_t value, void* stream) {
    if (!dst) return SWIN_ERR_BAD_ARG;
    set_u64_kernel<<<1, 1, 0, (hipStream_t)stream>>>((unsigned long long*)dst, (unsigned long long)value);
    return swin_launch_status();
}

extern "C" int swin_adamw_chunk_elems(void) { return ADAM_CHUNK; }

// ---- fp16 dynamic loss scaling on the device (the reference trains with apex O1: mmdet/apis/train.py:82-89 -> apex.amp's
// LossScaler("dynamic"): scale 2^16, halved when a step's gradients hold an inf / nan -- that step is skipped --, doubled after 2000
// clean steps).  Everything lives in the optimizer's device-resident state (layout above), so no step waits for the host:
//   swin_loss_scale_begin : skip = 0, grad_scale = 1 / loss_scale          (before the gradients are checked)
//   swin_grad_check_finite: skip = 1 if any of g[0..n) is not finite       (one call per gradient bucket)
//   swin_loss_scale_update: after the optimizer launch -- skipped step: scale *= backoff, counter = 0; clean step: counter += 1 and
//                           scale *= growth once it reaches growth_interval; the scale stays within [min_scale, max_scale].
__global__ void loss_scale_begin_kernel(AdamState* st) {
    st->skip = 0.f;
    st->grad_scale = 1.f / st->loss_scale;
}

__global__ __launch_bounds__(256) void grad_check_kernel(const float4* __restrict__ g, int64_t n4, const float* __restrict__ tail, int ntail,
                                                         AdamState* st) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = g[i];
        // x - x is 0 for finite x and nan for inf / nan
        const float z = (v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w);
        bad |= !(z == 0.f);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) { const float t = tail[threadIdx.x]; bad |= !((t - t) == 0.f); }
    if (__any(bad) && (threadIdx.x & 63) == 0) st->skip = 1.f;      // a plain store of the same value from every wave that saw one
}

__global__ void loss_scale_update_kernel(AdamState* st, float growth, float backoff, float interval, float lo, float hi) {
    float s = st->loss_scale, good = st->good_steps;
    if (st->skip != 0.f) { s *= backoff; good = 0.f; }
    else { good += 1.f; if (good >= interval) { s *= growth; good = 0.f; } }
    st->loss_scale = fminf(fmaxf(s, lo), hi);
    st->good_steps = good;
}

extern "C" int swin_loss_scale_begin(void* state, void* stream) {
    if (!state) return SWIN_ERR_BAD_ARG;
    loss_scale_begin_kernel<<<1, 1, 0, (hipStream_t)stream>>>((AdamState*)state);
    return swin_launch_status();
}

extern "C" int swin_grad_check_finite(const float* g, int64_t n, void* state, void* stream) {
    if (n == 0) return SWIN_OK;
    if (!g || n < 0 || !state || ((uintptr_t)g & 15)) return SWIN_ERR_BAD_ARG;
    const int64_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    grad_check_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>((const float4*)g, n4, g + n4 * 4, (int)(n - n4 * 4), (AdamState*)state);
    return swin_launch_status();
}

extern "C" int swin_loss_scale_update(void* state, float growth, float backoff, int growth_interval, float min_scale, float max_scale,
                                      void* stream) {
    if (!state || growth < 1.f || backoff <= 0.f || backoff > 1.f || growth_interval < 1 || !(min_scale > 0.f) || max_scale < min_scale)
        return SWIN_ERR_BAD_ARG;
    loss_scale_update_kernel<<<1, 1, 0, (hipStream_t)stream>>>((AdamState*)state, growth, backoff, (float)growth_interval, min_scale, max_scale);
    return swin_launch_status();
}
