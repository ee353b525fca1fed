// Greedy IoU NMS for gfx950: 64-wide suppression bitmask + ON-DEVICE reduction.
// Semantics: mmcv.ops.nms device part as reached from rpn_head.py:233 and bbox_nms.py:84 of the
// reference (through batched_nms).  mmcv copies the bitmask to the host and reduces it there
// (a stream sync per image); here one workgroup reduces it on the device, no sync.
//
// Build spec: boxes arrive sorted by descending score (stable); box j is
// suppressed by an earlier kept box i when  inter > thr * (area_i + area_j - inter)  in fp32,
// evaluated in exactly that order so the oracle and this kernel agree bit for bit.
#include "common.h"
#pragma clang fp contract(off)

__device__ __forceinline__ bool iou_gt(const float4 a, const float4 b, float off, float thr) {
    float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    float w = fmaxf(right - left + off, 0.f), h = fmaxf(bottom - top + off, 0.f);
    float inter = w * h;
    float sa = (a.z - a.x + off) * (a.w - a.y + off);
    float sb = (b.z - b.x + off) * (b.w - b.y + off);
    return inter > thr * (sa + sb - inter);
}

// grid (col_blocks, col_blocks); only cb >= rb does work.  64 threads: thread t owns row box rb*64+t.
__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, int n, float thr, float off,
                                                      uint64_t* __restrict__ mask, int col_blocks) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    boxes += (int64_t)blockIdx.z * n;                                    // image of a batched call
    mask += (int64_t)blockIdx.z * n * col_blocks;
    __shared__ float4 cbox[64];
    const int t = threadIdx.x;
    const int ncol = min(64, n - cb * 64);
    if (t < ncol) cbox[t] = boxes[cb * 64 + t];
    __syncthreads();
    const int row = rb * 64 + t;
    if (row >= n) return;
    const float4 me = boxes[row];
    uint64_t bits = 0;
    const int start = (rb == cb) ? t + 1 : 0;
    for (int j = start; j < ncol; ++j)
        if (iou_gt(me, cbox[j], off, thr)) bits |= 1ull << j;
    mask[(int64_t)row * col_blocks + cb] = bits;
}

// one workgroup of 1024 threads walks the 64-box blocks in order.  Per block: one lane does the greedy scan of the 64
// diagonal words (staged in LDS); then the kept rows are OR-ed into `remv` for the later column words with the
// (row, column) pairs spread over ALL threads -- 4 independent loads in flight per thread, LDS atomicOr to combine
// -- so a block costs about two memory latencies instead of cnt/4.  Stops as soon as `max_num` boxes are kept
// (> 0): the callers slice `dets[:max_per_img]` (rpn_head.py:235, bbox_nms.py:86-88), so later boxes never matter.
#define NMS_RT 1024
__global__ __launch_bounds__(NMS_RT) void nms_reduce_kernel(const uint64_t* __restrict__ mask, int n, int col_blocks,
                                                            uint8_t* __restrict__ keep, int32_t* __restrict__ num_kept,
                                                            int max_num, int32_t* __restrict__ kept_pos, int kept_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];   // col_blocks words
    mask += (int64_t)blockIdx.x * n * col_blocks;                        // image of a batched call
    keep += (int64_t)blockIdx.x * n;
    num_kept += blockIdx.x;
    if (kept_pos) kept_pos += (int64_t)blockIdx.x * kept_cap;
    __shared__ uint64_t kept_bits;
    __shared__ int kept_rows[64];
    __shared__ int kept_cnt, total;
    const int t = threadIdx.x;
    for (int j = t; j < col_blocks; j += NMS_RT) remv[j] = 0;
    if (kept_pos) for (int i = t; i < kept_cap; i += NMS_RT) kept_pos[i] = -1;     // fixed-size, -1 padded output
    if (t == 0) total = 0;
    __syncthreads();
    int b = 0;
    // wave 0 keeps the 64 diagonal words of the current block one per lane (prefetched one block ahead) and runs the
    // greedy scan on SCALAR registers: `r` is wave-uniform, the word of a kept row comes by v_readlane, and the loop
    // visits only the surviving bits (ctz) -- no LDS round trip and no 64 serial iterations on the critical path
    uint64_t dnext = 0;
    if (t < 64 && t < n) dnext = mask[(int64_t)t * col_blocks];
    for (; b < col_blocks; ++b) {
        const int lim = min(64, n - b * 64);
        if (t < 64) {
            const uint64_t dcur = dnext;
            if (b + 1 < col_blocks && (b + 1) * 64 + t < n) dnext = mask[(int64_t)((b + 1) * 64 + t) * col_blocks + b + 1];
            const unsigned dlo = (unsigned)dcur, dhi = (unsigned)(dcur >> 32);
            uint64_t r = remv[b], kb = 0;
            const uint64_t valid = lim == 64 ? ~0ull : ((1ull << lim) - 1);
            int cnt = 0;
            const int tot = total;
            uint64_t cand = ~r & valid;
            while (cand) {
                if (max_num > 0 && tot + cnt >= max_num) break;
                const int bit = __builtin_ctzll(cand);
                kb |= 1ull << bit;
                const uint64_t row = ((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)dhi, bit) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)dlo, bit);
                r |= row | (1ull << bit);
                if (t == 0) kept_rows[cnt] = b * 64 + bit;
                ++cnt;
                cand = ~r & valid;
            }
            if (t == 0) { kept_bits = kb; kept_cnt = cnt; total = tot + cnt; }
        }
        __syncthreads();
        const uint64_t kb = kept_bits;
        const int cnt = kept_cnt;
        if (t < lim) keep[b * 64 + t] = (uint8_t)((kb >> t) & 1);
        if (kept_pos && t < cnt && total - cnt + t < kept_cap) kept_pos[total - cnt + t] = kept_rows[t];
        if (max_num > 0 && total >= max_num) { ++b; break; }
        // work item = (column j, group of 4 kept rows); items spread over the whole workgroup
        const int ncol = col_blocks - (b + 1), ngrp = (cnt + 3) >> 2;
        for (int it = t; it < ncol * ngrp; it += NMS_RT) {
            const int j = b + 1 + it % ncol, k0 = (it / ncol) << 2;
            const uint64_t a0 = mask[(int64_t)kept_rows[k0] * col_blocks + j];
            const uint64_t a1 = k0 + 1 < cnt ? mask[(int64_t)kept_rows[k0 + 1] * col_blocks + j] : 0;
            const uint64_t a2 = k0 + 2 < cnt ? mask[(int64_t)kept_rows[k0 + 2] * col_blocks + j] : 0;
            const uint64_t a3 = k0 + 3 < cnt ? mask[(int64_t)kept_rows[k0 + 3] * col_blocks + j] : 0;
            const uint64_t acc = (a0 | a1) | (a2 | a3);
            if (acc) atomicOr(&remv[j], (unsigned long long)acc);
        }
        __syncthreads();
    }
    // early stop: everything after the last processed block is dropped
    for (int i = b * 64 + t; i < n; i += NMS_RT) keep[i] = 0;
    __syncthreads();
    if (t == 0) *num_kept = total;
}

extern "C" int64_t swin_nms_workspace_bytes(int64_t n) {
    if (n <= 0) return 8;
    int64_t cb = (n + 63) / 64;
    return n * cb * 8;
}

extern "C" int nms_sorted(const float* boxes_sorted, int64_t n, float iou_threshold, int offset, int max_num,
                          uint8_t* keep_flags, int32_t* num_kept, int32_t* kept_pos, int kept_cap, void* workspace,
                          void* stream) {
    if (n < 0 || !num_kept) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(num_kept, 0, sizeof(int32_t), s);
        if (e == hipSuccess && kept_pos && kept_cap > 0) e = hipMemsetAsync(kept_pos, 0xFF, sizeof(int32_t) * kept_cap, s);
        return e == hipSuccess ? SWIN_OK : SWIN_ERR_LAUNCH;
    }
    if (!boxes_sorted || !keep_flags || !workspace) return SWIN_ERR_BAD_ARG;
    int col_blocks = (int)((n + 63) / 64);
    if (col_blocks > 65535 || (size_t)col_blocks * 8 > 60000) return SWIN_ERR_UNSUPPORTED;   // n <= 480k
    dim3 grid(col_blocks, col_blocks);
    nms_mask_kernel<<<grid, 64, 0, s>>>((const float4*)boxes_sorted, (int)n, iou_threshold, (float)offset,
                                        (uint64_t*)workspace, col_blocks);
    nms_reduce_kernel<<<1, NMS_RT, (size_t)col_blocks * 8, s>>>((const uint64_t*)workspace, (int)n, col_blocks, keep_flags,
                                                            num_kept, max_num, kept_pos, kept_cap);
    return swin_launch_status();
}

// Batched form for fixed-size proposal lists: `batch` images with the SAME n, all buffers with a leading batch
// dimension (boxes (batch,n,4), keep_flags (batch,n), num_kept (batch), kept_pos (batch,kept_cap), workspace
// batch * swin_nms_workspace_bytes(n)).  One pair of launches for all images: the single-workgroup reductions of the
// images run side by side on different CUs instead of back to back.
extern "C" int nms_sorted_batch(const float* boxes_sorted, int batch, int64_t n, float iou_threshold, int offset, int max_num,
                                uint8_t* keep_flags, int32_t* num_kept, int32_t* kept_pos, int kept_cap, void* workspace,
                                void* stream) {
    if (batch <= 0 || n <= 0 || !boxes_sorted || !keep_flags || !num_kept || !workspace) return SWIN_ERR_BAD_ARG;
    int col_blocks = (int)((n + 63) / 64);
    if (col_blocks > 65535 || (size_t)col_blocks * 8 > 60000 || batch > 65535) return SWIN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(col_blocks, col_blocks, batch);
    nms_mask_kernel<<<grid, 64, 0, s>>>((const float4*)boxes_sorted, (int)n, iou_threshold, (float)offset, (uint64_t*)workspace,
                                        col_blocks);
    nms_reduce_kernel<<<batch, NMS_RT, (size_t)col_blocks * 8, s>>>((const uint64_t*)workspace, (int)n, col_blocks, keep_flags,
                                                                   num_kept, max_num, kept_pos, kept_cap);
    return swin_launch_status();
}

// ---- batched_nms front / back end on the device (round 2) -----------------------------------------------------------------
// mmcv's batched_nms (bbox_nms / rpn_head.py:233) offsets every box by idx * (max_coordinate + 1), nms() then sorts by score
// (descending; here: stable, ties -> lower index first) before the suppression scan.  One block per image does all of that:
// max over the image's coordinates, a bitonic sort of 64-bit keys (~orderable(score) << 32 | index: ascending order of the key
// = descending score, ascending index) in LDS, and the gather of the offset boxes into sorted order.  Replaces amax, two
// casts, mul, add, a segmented radix sort (8 launches) and a gather.  n <= 16384 (RPN: 8780).
constexpr int NP_MAX = 16384;

__global__ __launch_bounds__(1024) void nms_prepare_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                           const int64_t* __restrict__ idxs, int n, int np, float4* __restrict__ boxes_sorted,
                                                           int32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    __shared__ float red[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float4* bx = boxes + (size_t)b * n;
    const float* sc = scores + (size_t)b * n;
    float m = -INFINITY;
    for (int i = tid; i < np; i += 1024) {
        unsigned long long k = ~0ull;
        if (i < n) {
            const float4 v = bx[i];
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            unsigned u = __float_as_uint(sc[i]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // ascending unsigned order == ascending float order
            k = ((unsigned long long)(~u) << 32) | (unsigned)i;
        }
        keys[i] = k;
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    float maxc = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) maxc = fmaxf(maxc, red[w]);
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np >> 1); t += 1024) {
                const int i = 2 * t - (t & (j - 1)), l = i + j;
                const unsigned long long a = keys[i], c = keys[l];
                const bool up = (i & k) == 0;
                if ((a > c) == up) { keys[i] = c; keys[l] = a; }
            }
            __syncthreads();
        }
    }
    const float scale = maxc + 1.0f;
    const int64_t* id = idxs + (size_t)b * n;
    for (int i = tid; i < n; i += 1024) {
        const int src = (int)(unsigned)keys[i];
        const float off = (float)id[src] * scale;
        float4 v = bx[src];
        v.x += off; v.y += off; v.z += off; v.w += off;
        boxes_sorted[(size_t)b * n + i] = v;
        order[(size_t)b * n + i] = src;
    }
}

// dets (batch, cap, 5) = [original box | score] of the kept boxes in kept order, zero rows and valid = 0 after the last one
__global__ __launch_bounds__(256) void nms_gather_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores,
                                                         const int32_t* __restrict__ order, const int32_t* __restrict__ kept_pos, int n,
                                                         int cap, int total, float* __restrict__ dets, uint8_t* __restrict__ valid) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int b = t / cap;
    const int pos = kept_pos[t];
    float4 v = {0.f, 0.f, 0.f, 0.f};
    float s = 0.f;
    if (pos >= 0) {
        const int src = order[(size_t)b * n + pos];
        v = boxes[(size_t)b * n + src];
        s = scores[(size_t)b * n + src];
    }
    float* d = dets + (size_t)t * 5;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; d[4] = s;
    valid[t] = pos >= 0 ? 1 : 0;
}

extern "C" int nms_prepare_sorted_batch(const float* boxes, const float* scores, const int64_t* idxs, int batch, int64_t n,
                                        float* boxes_sorted, int32_t* order, void* stream) {
    if (batch <= 0 || n <= 0) return SWIN_OK;
    if (!boxes || !scores || !idxs || !boxes_sorted || !order) return SWIN_ERR_BAD_ARG;
    if (n > NP_MAX || batch > 65535) return SWIN_ERR_UNSUPPORTED;
    int np = 1024;
    while (np < n) np <<= 1;
    const size_t lds = (size_t)np * sizeof(unsigned long long);
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)nms_prepare_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(NP_MAX * sizeof(unsigned long long))) != hipSuccess) return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    nms_prepare_kernel<<<batch, 1024, lds, (hipStream_t)stream>>>((const float4*)boxes, scores, idxs, (int)n, np, (float4*)boxes_sorted,
                                                                 order);
    return swin_launch_status();
}

extern "C" int nms_gather_dets(const float* boxes, const float* scores, const int32_t* order, const int32_t* kept_pos, int batch,
                               int64_t n, int kept_cap, float* dets, uint8_t* valid, void* stream) {
    if (batch <= 0 || kept_cap <= 0) return SWIN_OK;
    if (!boxes || !scores || !order || !kept_pos || !dets || !valid || n <= 0) return SWIN_ERR_BAD_ARG;
    const int total = batch * kept_cap;
    nms_gather_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>((const float4*)boxes, scores, order, kept_pos, (int)n, kept_cap,
                                                                            total, dets, valid);
    return swin_launch_status();
}
