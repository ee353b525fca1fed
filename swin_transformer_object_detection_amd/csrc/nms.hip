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
// n_dyn (nullable): the number of boxes of batch entry blockIdx.x read from device memory (<= n_rows, the entry's row stride) --
// the per-group lists of nms_sorted_batch_grouped; col_stride: words per mask row.
__global__ __launch_bounds__(NMS_RT) void nms_reduce_kernel(const uint64_t* __restrict__ mask, int n_rows, int col_stride,
                                                            uint8_t* __restrict__ keep, int32_t* __restrict__ num_kept,
                                                            int max_num, int32_t* __restrict__ kept_pos, int kept_cap,
                                                            const int32_t* __restrict__ n_dyn) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long remv[];   // col_blocks words
    const int n = n_dyn ? min(n_dyn[blockIdx.x], n_rows) : n_rows;
    const int col_blocks = (n + 63) / 64;
    mask += (int64_t)blockIdx.x * n_rows * col_stride;                   // image of a batched call
    keep += (int64_t)blockIdx.x * n_rows;
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
    // wave 0 keeps the 64 diagonal words of the current block one per lane (prefetched one block ahead) and resolves the block's
    // greedy scan with wave-wide ORs (below)
    uint64_t dnext = 0;
    if (t < 64 && t < n) dnext = mask[(int64_t)t * col_stride];
    for (; b < col_blocks; ++b) {
        const int lim = min(64, n - b * 64);
        if (t < 64) {
            const uint64_t dcur = dnext;
            if (b + 1 < col_blocks && (b + 1) * 64 + t < n) dnext = mask[(int64_t)((b + 1) * 64 + t) * col_stride + b + 1];
            const uint64_t r = remv[b];
            const uint64_t valid = lim == 64 ? ~0ull : ((1ull << lim) - 1);
            const int tot = total;
            const uint64_t cand = ~r & valid;
            // The greedy scan of the block as a fixed point (round 3): K = cand & ~(OR of the diagonal words of the rows in K).  A
            // diagonal word only has bits ABOVE its own row, so the solution is unique and equals the greedy one (row j is decided once
            // the rows below it are), and the iteration K <- F(K) from K = cand reaches it in at most (longest suppression chain) steps
            // -- one or two for the RPN's lists -- each a wave-wide OR instead of a serial step per kept box (64 x ~70 cycles).
            uint64_t kb = cand;
            for (int iter = 0; iter < 64; ++iter) {
                uint64_t sup = ((kb >> t) & 1) ? dcur : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const unsigned lo = __shfl_xor((unsigned)sup, d), hi = __shfl_xor((unsigned)(sup >> 32), d);
                    sup |= ((uint64_t)hi << 32) | lo;
                }
                const uint64_t nk = cand & ~sup;
                if (nk == kb) break;
                kb = nk;
            }
            int cnt = __builtin_popcountll(kb);
            if (max_num > 0 && tot + cnt > max_num) {                // keep the first max_num - tot of them (wave-uniform)
                int keepn = max_num - tot;
                uint64_t m = kb, first = 0;
                while (keepn-- > 0 && m) { const uint64_t low = m & (~m + 1); first |= low; m ^= low; }
                kb = first;
                cnt = __builtin_popcountll(kb);
            }
            if ((kb >> t) & 1) kept_rows[__builtin_popcountll(kb & ((1ull << t) - 1))] = b * 64 + t;
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
            const uint64_t a0 = mask[(int64_t)kept_rows[k0] * col_stride + j];
            const uint64_t a1 = k0 + 1 < cnt ? mask[(int64_t)kept_rows[k0 + 1] * col_stride + j] : 0;
            const uint64_t a2 = k0 + 2 < cnt ? mask[(int64_t)kept_rows[k0 + 2] * col_stride + j] : 0;
            const uint64_t a3 = k0 + 3 < cnt ? mask[(int64_t)kept_rows[k0 + 3] * col_stride + j] : 0;
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
                                                            num_kept, max_num, kept_pos, kept_cap, nullptr);
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
                                                                   num_kept, max_num, kept_pos, kept_cap, nullptr);
    return swin_launch_status();
}

// ---- grouped form (round 3).  batched_nms offsets the boxes of class / level g by g * (max coordinate + 1) (rpn_head.py:233,
// mmcv batched_nms): boxes of different groups never overlap, so the greedy scan over the score-sorted list is the union of the
// scans of the groups' sub-lists -- same boxes, same IoU arithmetic (the OFFSET boxes are compared, as in the reference), same
// result.  The single scan walks ceil(n / 64) = 138 dependent steps for the RPN's 8780 candidates (142 us, one workgroup per image)
// and the mask is 138^2 blocks; five level lists of <= 2000 are 32 steps each, side by side, and 5 x 32^2 mask blocks.
//   partition: sorted position i -> (group, rank in its group), stable;   mask / reduce per (image, group);
//   finalize: flags back in sorted order, the first max_num kept positions (the callers slice dets[:max_per_img]).
constexpr int NMS_G = 8;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d);
        if (lane >= d) v += u;
    }
    return v;
}

struct NmsGroupWs { int32_t* gcount; int32_t* pk; int32_t* glist; uint8_t* gkeep; uint64_t* mask; int32_t* gkept; };

__global__ __launch_bounds__(1024) void nms_group_partition_kernel(const int32_t* __restrict__ order, const int64_t* __restrict__ idxs, int n,
                                                                   int G, int gmax, int32_t* __restrict__ gcount, int32_t* __restrict__ pk,
                                                                   int32_t* __restrict__ glist) {
    __shared__ int wtot[NMS_G][16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    order += (size_t)b * n; idxs += (size_t)b * n; pk += (size_t)b * n;
    glist += (size_t)b * G * gmax; gcount += (size_t)b * G;
    const int per = (n + 1023) / 1024, i0 = t * per, i1 = min(n, i0 + per);
    int cnt[NMS_G];
#pragma unroll
    for (int g = 0; g < NMS_G; ++g) cnt[g] = 0;
    for (int i = i0; i < i1; ++i) {
        const int g = min(max((int)idxs[order[i]], 0), G - 1);
#pragma unroll
        for (int q = 0; q < NMS_G; ++q) cnt[q] += (q == g);
    }
    int start[NMS_G];
#pragma unroll
    for (int g = 0; g < NMS_G; ++g) {
        const int incl = wave_incl_scan(cnt[g], lane);
        if (lane == 63) wtot[g][wave] = incl;
        start[g] = incl - cnt[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NMS_G; ++g) {
        int base = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int v = wtot[g][w]; base += w < wave ? v : 0; tot += v; }
        start[g] += base;
        if (t == 0 && g < G) gcount[g] = min(tot, gmax);             // (a group larger than the caller's bound loses its tail: the caller's contract)
    }
    for (int i = i0; i < i1; ++i) {
        const int g = min(max((int)idxs[order[i]], 0), G - 1);
        int k = 0;
#pragma unroll
        for (int q = 0; q < NMS_G; ++q)
            if (q == g) k = start[q]++;
        pk[i] = (g << 20) | min(k, (1 << 20) - 1);
        if (k < gmax) glist[(size_t)g * gmax + k] = i;
    }
}

// grid (cbg, cbg, batch * G); as nms_mask_kernel, the boxes through the group's list
__global__ __launch_bounds__(64) void nms_mask_grouped_kernel(const float4* __restrict__ boxes, int n, int G, int gmax, float thr, float off,
                                                              const int32_t* __restrict__ gcount, const int32_t* __restrict__ glist,
                                                              uint64_t* __restrict__ mask, int cbg) {
    const int rb = blockIdx.y, cb = blockIdx.x, bz = blockIdx.z, b = bz / G, g = bz - b * G;
    const int ng = gcount[bz];
    if (cb < rb || cb * 64 >= ng) return;
    boxes += (size_t)b * n;
    glist += (size_t)bz * gmax;
    mask += (size_t)bz * gmax * cbg;
    __shared__ float4 cbox[64];
    const int t = threadIdx.x;
    const int ncol = min(64, ng - cb * 64);
    if (t < ncol) cbox[t] = boxes[glist[cb * 64 + t]];
    __syncthreads();
    const int row = rb * 64 + t;
    if (row >= ng) return;
    const float4 me = boxes[glist[row]];
    uint64_t bits = 0;
    const int start = (rb == cb) ? t + 1 : 0;
    for (int j = start; j < ncol; ++j)
        if (iou_gt(me, cbox[j], off, thr)) bits |= 1ull << j;
    mask[(size_t)row * cbg + cb] = bits;
}

__global__ __launch_bounds__(1024) void nms_group_finalize_kernel(const int32_t* __restrict__ pk, const uint8_t* __restrict__ gkeep, int n, int G,
                                                                  int gmax, int max_num, uint8_t* __restrict__ keep, int32_t* __restrict__ num_kept,
                                                                  int32_t* __restrict__ kept_pos, int kept_cap) {
    __shared__ int wtot[16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    pk += (size_t)b * n; gkeep += (size_t)b * G * gmax; keep += (size_t)b * n;
    if (kept_pos) kept_pos += (size_t)b * kept_cap;
    const int per = (n + 1023) / 1024, i0 = t * per, i1 = min(n, i0 + per);
    int cnt = 0;
    for (int i = i0; i < i1; ++i) {
        const int v = pk[i], g = v >> 20, k = v & ((1 << 20) - 1);
        cnt += (k < gmax && gkeep[(size_t)g * gmax + k]) ? 1 : 0;
    }
    const int incl = wave_incl_scan(cnt, lane);
    if (lane == 63) wtot[wave] = incl;
    if (kept_pos) for (int i = t; i < kept_cap; i += 1024) kept_pos[i] = -1;     // fixed-size, -1 padded output
    __syncthreads();
    int r = incl - cnt, tot = 0;
    for (int w = 0; w < 16; ++w) { const int v = wtot[w]; r += w < wave ? v : 0; tot += v; }
    const int lim = max_num > 0 ? max_num : 0x7fffffff;
    for (int i = i0; i < i1; ++i) {
        const int v = pk[i], g = v >> 20, k = v & ((1 << 20) - 1);
        const bool f = k < gmax && gkeep[(size_t)g * gmax + k];
        const bool kp = f && r < lim;
        keep[i] = kp ? 1 : 0;
        if (kp && kept_pos && r < kept_cap) kept_pos[r] = i;
        r += f ? 1 : 0;
    }
    if (t == 0) num_kept[b] = min(tot, lim);
}

static NmsGroupWs nms_group_carve(void* workspace, int batch, int64_t n, int G, int gmax) {
    const int cbg = (gmax + 63) / 64;
    char* p = (char*)workspace;
    auto take = [&](size_t bytes) { char* q = p; p += (bytes + 255) / 256 * 256; return q; };
    NmsGroupWs w;
    w.gcount = (int32_t*)take((size_t)batch * G * 4);
    w.gkept = (int32_t*)take((size_t)batch * G * 4);
    w.pk = (int32_t*)take((size_t)batch * n * 4);
    w.glist = (int32_t*)take((size_t)batch * G * gmax * 4);
    w.gkeep = (uint8_t*)take((size_t)batch * G * gmax);
    w.mask = (uint64_t*)take((size_t)batch * G * gmax * cbg * 8);
    return w;
}

extern "C" int64_t nms_grouped_workspace_bytes(int batch, int64_t n, int groups, int gmax) {
    if (batch <= 0 || n <= 0 || groups <= 0 || gmax <= 0) return 256;
    const int64_t cbg = (gmax + 63) / 64;
    const int64_t parts[6] = {(int64_t)batch * groups * 4, (int64_t)batch * groups * 4, (int64_t)batch * n * 4, (int64_t)batch * groups * gmax * 4,
                              (int64_t)batch * groups * gmax, (int64_t)batch * groups * gmax * cbg * 8};
    int64_t tot = 0;
    for (int64_t v : parts) tot += (v + 255) / 256 * 256;
    return tot;
}

// boxes_sorted (batch, n, 4): the OFFSET boxes in score order and order (batch, n): their source indices (nms_prepare_sorted_batch);
// idxs (batch, n) int64 in [0, groups): the class / level of source box i; no group larger than gmax (the caller's bound: per-level
// candidate counts); groups <= 8, n < 2^20, iou_threshold > 0.  Outputs as nms_sorted_batch.
extern "C" int nms_sorted_batch_grouped(const float* boxes_sorted, const int32_t* order, const int64_t* idxs, int batch, int64_t n, int groups,
                                        int gmax, float iou_threshold, int offset, int max_num, uint8_t* keep_flags, int32_t* num_kept,
                                        int32_t* kept_pos, int kept_cap, void* workspace, void* stream) {
    if (batch <= 0 || n <= 0 || !boxes_sorted || !order || !idxs || !keep_flags || !num_kept || !workspace) return SWIN_ERR_BAD_ARG;
    if (groups < 1 || groups > NMS_G || gmax < 1 || n >= (1 << 20) || !(iou_threshold > 0.f) || batch * groups > 65535) return SWIN_ERR_UNSUPPORTED;
    if (gmax > n) gmax = (int)n;
    const int cbg = (gmax + 63) / 64;
    if ((size_t)cbg * 8 > 60000) return SWIN_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const NmsGroupWs w = nms_group_carve(workspace, batch, n, groups, gmax);
    nms_group_partition_kernel<<<batch, 1024, 0, s>>>(order, idxs, (int)n, groups, gmax, w.gcount, w.pk, w.glist);
    nms_mask_grouped_kernel<<<dim3(cbg, cbg, batch * groups), 64, 0, s>>>((const float4*)boxes_sorted, (int)n, groups, gmax, iou_threshold,
                                                                          (float)offset, w.gcount, w.glist, w.mask, cbg);
    // one reduction per (image, group): row stride gmax, the group's own count from device memory (gcount[image * groups + group])
    nms_reduce_kernel<<<batch * groups, NMS_RT, (size_t)cbg * 8, s>>>(w.mask, gmax, cbg, w.gkeep, w.gkept, max_num, nullptr, 0, w.gcount);
    nms_group_finalize_kernel<<<batch, 1024, 0, s>>>(w.pk, w.gkeep, (int)n, groups, gmax, max_num, keep_flags, num_kept, kept_pos, kept_cap);
    return swin_launch_status();
}

// ---- batched_nms front / back end on the device (round 2) -----------------------------------------------------------------
// mmcv's batched_nms (bbox_nms / rpn_head.py:233) offsets every box by idx * (max_coordinate + 1), nms() then sorts by score
// (descending; here: stable, ties -> lower index first) before the suppression scan.  One block per image does all of that:
// max over the image's coordinates, a bitonic sort of 64-bit keys (~orderable(score) << 32 | index: ascending order of the key
// = descending score, ascending index) in LDS, and the gather of the offset boxes into sorted order.  Replaces amax, two
// casts, mul, add, a segmented radix sort (8 launches) and a gather.  n <= 16384 (RPN: 8780).
constexpr int NP_MAX = 16384;
constexpr int SEG = 2048;                       // candidates per sorting block
constexpr int MAX_SEGS = NP_MAX / SEG;

__device__ __forceinline__ unsigned f2ord(float f) {             // ascending unsigned order == ascending float order
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// pass 1: block (segment, image) sorts its <= 2048 keys in LDS (bitonic: 66 passes over 16 KB; one block sorting all 8780
// candidates of an image moved 262 KB of LDS per pass x 105 passes = 150 us) and folds its coordinate maximum into maxc[image]
__global__ __launch_bounds__(1024) void nms_segsort_kernel(const float4* __restrict__ boxes, const float* __restrict__ scores, int n,
                                                           unsigned long long* __restrict__ skeys, unsigned* __restrict__ maxc) {
    __shared__ unsigned long long keys[SEG];
    __shared__ float red[16];
    const int seg = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int lo = seg * SEG;
    float m = -INFINITY;
    for (int t = tid; t < SEG; t += 1024) {
        const int i = lo + t;
        unsigned long long k = ~0ull;
        if (i < n) {
            const float4 v = boxes[(size_t)b * n + i];
            m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            k = ((unsigned long long)(~f2ord(scores[(size_t)b * n + i])) << 32) | (unsigned)i;
        }
        keys[t] = k;
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float mm = red[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) mm = fmaxf(mm, red[w]);
        atomicMax(maxc + b, f2ord(mm));
    }
    for (int k = 2; k <= SEG; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int t = tid;                                   // SEG / 2 == 1024 compare-exchanges per pass
            const int i = 2 * t - (t & (j - 1)), l = i + j;
            const unsigned long long a = keys[i], c = keys[l];
            const bool up = (i & k) == 0;
            if ((a > c) == up) { keys[i] = c; keys[l] = a; }
            __syncthreads();
        }
    }
    for (int t = tid; t < SEG; t += 1024) skeys[((size_t)b * gridDim.x + seg) * SEG + t] = keys[t];
}

// pass 2: a key's final position = its rank in its own segment + the number of smaller keys in every other segment (binary
// searches; keys are unique, so the merge is exact and stable); writes the offset box and the source index there
__global__ __launch_bounds__(256) void nms_merge_kernel(const float4* __restrict__ boxes, const int64_t* __restrict__ idxs, int n, int nseg,
                                                        const unsigned long long* __restrict__ skeys, const unsigned* __restrict__ maxc,
                                                        float4* __restrict__ boxes_sorted, int32_t* __restrict__ order) {
    const int b = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;               // (segment, rank) flattened
    if (g >= nseg * SEG) return;
    const int seg = g / SEG, r = g - seg * SEG;
    const unsigned long long* base = skeys + (size_t)b * nseg * SEG;
    const unsigned long long key = base[(size_t)seg * SEG + r];
    if (key == ~0ull) return;                                    // padding
    int pos = r;
    for (int s2 = 0; s2 < nseg; ++s2) {
        if (s2 == seg) continue;
        const unsigned long long* p = base + (size_t)s2 * SEG;
        int lo = 0, hi = SEG;                                    // first index with p[idx] >= key (padding ~0 is larger)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (p[mid] < key) lo = mid + 1; else hi = mid;
        }
        pos += lo;
    }
    const int src = (int)(unsigned)key;
    const float scale = ord2f(maxc[b]) + 1.0f;
    const float off = (float)idxs[(size_t)b * n + src] * scale;
    float4 v = boxes[(size_t)b * n + src];
    v.x += off; v.y += off; v.z += off; v.w += off;
    boxes_sorted[(size_t)b * n + pos] = v;
    order[(size_t)b * n + pos] = src;
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

extern "C" int64_t nms_prepare_workspace_bytes(int batch, int64_t n) {
    const int64_t nseg = (n + SEG - 1) / SEG;
    return (int64_t)batch * nseg * SEG * (int64_t)sizeof(unsigned long long) + (int64_t)batch * (int64_t)sizeof(unsigned) + 16;
}

extern "C" int nms_prepare_sorted_batch(const float* boxes, const float* scores, const int64_t* idxs, int batch, int64_t n,
                                        float* boxes_sorted, int32_t* order, void* workspace, void* stream) {
    if (batch <= 0 || n <= 0) return SWIN_OK;
    if (!boxes || !scores || !idxs || !boxes_sorted || !order || !workspace) return SWIN_ERR_BAD_ARG;
    if (n > NP_MAX || batch > 65535) return SWIN_ERR_UNSUPPORTED;
    const int nseg = (int)((n + SEG - 1) / SEG);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long* skeys = (unsigned long long*)workspace;
    unsigned* maxc = (unsigned*)(skeys + (size_t)batch * nseg * SEG);
    if (hipMemsetAsync(maxc, 0, (size_t)batch * sizeof(unsigned), s) != hipSuccess) return SWIN_ERR_LAUNCH;
    nms_segsort_kernel<<<dim3(nseg, batch), 1024, 0, s>>>((const float4*)boxes, scores, (int)n, skeys, maxc);
    nms_merge_kernel<<<dim3((nseg * SEG + 255) / 256, batch), 256, 0, s>>>((const float4*)boxes, idxs, (int)n, nseg, skeys, maxc,
                                                                          (float4*)boxes_sorted, order);
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
