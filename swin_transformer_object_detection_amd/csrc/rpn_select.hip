// RPN proposal selection on the device: RPNHead._get_bboxes, rpn_head.py:126-187.
//
// Per image and pyramid level the reference takes sigmoid of the objectness logits, sorts the level's anchors by score
// (descending) and keeps the first nms_pre, gathers their deltas and anchors, concatenates the levels and decodes
// (DeltaXYWHBBoxCoder.decode with max_shape).  Here ONE launch does all of it: a block per (level, image) finds the
// nms_pre-th largest score with a 3-pass radix select over the 32-bit score bits (11 + 11 + 10), then compacts the
// selected anchors IN ASCENDING ANCHOR ORDER, decoding each as it is written.
//
// Order contract: the reference hands batched_nms a list that is score-sorted inside each level; the NMS then sorts the
// whole list by score again (stable).  Elements with EQUAL scores therefore end up ordered by (level, anchor index) --
// exactly the order a stable sort produces from a level-major, index-ascending list.  At the nms_pre boundary a stable
// descending sort keeps, among the anchors tied with the threshold score, those with the LOWEST indices: the compaction
// below admits tied anchors in index order until the level's quota is full.  So the candidate SET and every later
// tie-break equal the reference's (checked against the CPU restatement of rpn_get_bboxes in the tests), without sorting here.
//
// HBM-bound integer/byte work: the logits are read once (2-4 B each), the 4-byte keys written once and re-read three
// times from L2 (768 KB per block at level 0); deltas and anchors are read for the selected anchors only.
#include "common.h"

namespace {

constexpr int RS_THREADS = 1024;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_MAX_LEVELS = 8;

struct RsLevels {
    int L;
    int off[RS_MAX_LEVELS + 1];       // first anchor of each level in the flattened (level, h, w, a) order
    int out_off[RS_MAX_LEVELS + 1];   // first output slot of each level: sum of min(n_l, nms_pre)
};
struct RsF4 { float v[4]; };

__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* sh /* RS_WAVES + 1 */, unsigned* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    __syncthreads();                      // sh may still be read from a previous call
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int i = 0; i < RS_WAVES; ++i) { const unsigned t = sh[i]; sh[i] = run; run += t; }
        sh[RS_WAVES] = run;
    }
    __syncthreads();
    if (total) *total = sh[RS_WAVES];
    return sh[w] + inc - v;
}

// one histogram pass: keys whose bits above `shift + bits` equal `prefix` are binned by their next `bits` bits.
// A wave whose active lanes all carry the same digit (scores of a random-init head share their exponent) adds once.
__device__ __forceinline__ void hist_pass(const unsigned* __restrict__ keys, int n, unsigned prefix, int shift, int bits,
                                          bool use_prefix, unsigned* hist) {
    const int lane = threadIdx.x & 63;
    const unsigned dmask = (1u << bits) - 1u;
    for (int base = 0; base < n; base += RS_THREADS) {
        const int i = base + threadIdx.x;
        unsigned key = 0;
        bool act = i < n;
        if (act) key = keys[i];
        if (use_prefix) act = act && ((key >> (shift + bits)) == prefix);
        const unsigned digit = (key >> shift) & dmask;
        const unsigned long long m = __ballot(act);
        if (m == 0) continue;
        const int first = __ffsll((long long)m) - 1;
        const unsigned d0 = __shfl(digit, first);
        if (__all(!act || digit == d0)) {
            if (lane == first) atomicAdd(&hist[d0], (unsigned)__popcll(m));
        } else if (act) {
            atomicAdd(&hist[digit], 1u);
        }
    }
}

// from the histogram (bins descending = larger keys first): the bin holding the k-th largest, and k within that bin
__device__ __forceinline__ void pick_bin(const unsigned* hist, int bits, unsigned k, unsigned* sh_scan, unsigned* sel /* [2] */) {
    const int nb = 1 << bits;                         // 2048 or 1024 bins, two or one per thread
    const int per = nb / RS_THREADS > 0 ? nb / RS_THREADS : 1;
    unsigned h0 = 0, h1 = 0;
    int b0 = -1, b1 = -1;
    if (per == 2) {
        b0 = nb - 1 - 2 * (int)threadIdx.x; b1 = b0 - 1;
        h0 = hist[b0]; h1 = hist[b1];
    } else {
        b0 = nb - 1 - (int)threadIdx.x;
        if (b0 >= 0) h0 = hist[b0];
    }
    const unsigned above = block_excl_scan(h0 + h1, sh_scan, nullptr);
    if (h0 > 0 && above < k && k <= above + h0) { sel[0] = (unsigned)b0; sel[1] = k - above; }
    else if (h1 > 0 && above + h0 < k && k <= above + h0 + h1) { sel[0] = (unsigned)b1; sel[1] = k - above - h0; }
    __syncthreads();
}

template <typename T>
__device__ __forceinline__ void decode_store(const T* __restrict__ reg, const float4* __restrict__ anchors, int64_t src, int64_t dst,
                                             float score, int level, RsF4 means, RsF4 stds, float max_h, float max_w, float max_ratio,
                                             float* __restrict__ out_scores, float4* __restrict__ out_boxes, int64_t* __restrict__ out_ids,
                                             int64_t img_off_in, int64_t img_off_out) {
    const T* d = reg + (img_off_in + src) * 4;
    const float4 r = anchors[src];
    // DeltaXYWHBBoxCoder.decode, delta_xywh_bbox_coder.py:189-237 (same operation order as det_delta2bbox)
    const float dx = Elt<T>::ld(d) * stds.v[0] + means.v[0], dy = Elt<T>::ld(d + 1) * stds.v[1] + means.v[1];
    float dw = Elt<T>::ld(d + 2) * stds.v[2] + means.v[2], dh = Elt<T>::ld(d + 3) * stds.v[3] + means.v[3];
    dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
    dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
    const float px = (r.x + r.z) * 0.5f, py = (r.y + r.w) * 0.5f, pw = r.z - r.x, ph = r.w - r.y;
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = px + pw * dx, gy = py + ph * dy;
    float4 o = {gx - gw * 0.5f, gy - gh * 0.5f, gx + gw * 0.5f, gy + gh * 0.5f};
    if (max_w > 0.f) {
        o.x = fminf(fmaxf(o.x, 0.f), max_w); o.z = fminf(fmaxf(o.z, 0.f), max_w);
        o.y = fminf(fmaxf(o.y, 0.f), max_h); o.w = fminf(fmaxf(o.w, 0.f), max_h);
    }
    out_scores[img_off_out + dst] = score;
    out_boxes[img_off_out + dst] = o;
    out_ids[img_off_out + dst] = level;
}

template <typename T>
__global__ __launch_bounds__(RS_THREADS) void rpn_topk_decode_kernel(
        const T* __restrict__ cls, const T* __restrict__ reg, const float4* __restrict__ anchors, RsLevels lv, int nms_pre,
        RsF4 means, RsF4 stds, float max_h, float max_w, float max_ratio, unsigned* __restrict__ keys_ws,
        float* __restrict__ out_scores, float4* __restrict__ out_boxes, int64_t* __restrict__ out_ids) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned sh_scan[RS_WAVES + 1];
    __shared__ unsigned sel[2];
    __shared__ unsigned wave_gt[RS_WAVES], wave_eq[RS_WAVES];
    const int level = blockIdx.x, img = blockIdx.y;
    const int n = lv.off[level + 1] - lv.off[level];
    const int64_t total = lv.off[lv.L], total_out = lv.out_off[lv.L];
    const int64_t in0 = (int64_t)img * total + lv.off[level];        // this (image, level)'s first logit / key
    const int64_t img_off_in = (int64_t)img * total, img_off_out = (int64_t)img * total_out;
    const T* c = cls + in0;
    unsigned* keys = keys_ws + in0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

    if (n <= nms_pre) {                   // rpn_head.py:162: no selection, anchor order kept
        for (int i = threadIdx.x; i < n; i += RS_THREADS) {
            const float s = 1.0f / (1.0f + expf(-Elt<T>::ld(c + i)));
            decode_store<T>(reg, anchors, lv.off[level] + i, lv.out_off[level] + i, s, level, means, stds, max_h, max_w, max_ratio,
                            out_scores, out_boxes, out_ids, img_off_in, img_off_out);
        }
        return;
    }
    // ---- keys: the fp32 bits of sigmoid(logit) order like the scores (they are >= 0) ----
    for (int i = threadIdx.x; i < n; i += RS_THREADS) {
        const float s = 1.0f / (1.0f + expf(-Elt<T>::ld(c + i)));
        keys[i] = __float_as_uint(s);
    }
    // every thread re-reads only what it wrote itself (same i -> thread map in every pass), so no fence is needed
    // ---- radix select of the nms_pre-th largest key ----
    unsigned prefix = 0, k = (unsigned)nms_pre;
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        for (int i = threadIdx.x; i < 2048; i += RS_THREADS) hist[i] = 0;
        __syncthreads();
        hist_pass(keys, n, prefix, shifts[p], nbits[p], p > 0, hist);
        __syncthreads();
        pick_bin(hist, nbits[p], k, sh_scan, sel);
        prefix = (prefix << nbits[p]) | sel[0];
        k = sel[1];
        __syncthreads();
    }
    const unsigned thr = prefix;          // the nms_pre-th largest key; `k` of the keys equal to it are admitted
    const unsigned need_eq = k;
    // ---- compaction in ascending anchor order: a contiguous segment per wave ----
    int seg = (n + RS_WAVES - 1) / RS_WAVES;
    seg = (seg + 63) & ~63;
    const int s0 = w * seg, s1 = min(n, s0 + seg);
    unsigned cgt = 0, ceq = 0;
    for (int base = s0; base < s1; base += 64) {
        const int i = base + lane;
        const unsigned key = i < s1 ? keys[i] : 0u;
        cgt += (unsigned)__popcll(__ballot(i < s1 && key > thr));
        ceq += (unsigned)__popcll(__ballot(i < s1 && key == thr));
    }
    if (lane == 0) { wave_gt[w] = cgt; wave_eq[w] = ceq; }
    __syncthreads();
    unsigned eq_base = 0, sel_base = 0;
    for (int v = 0; v < w; ++v) {
        const unsigned e = wave_eq[v];
        const unsigned room = need_eq > eq_base ? need_eq - eq_base : 0u;
        sel_base += wave_gt[v] + (e < room ? e : room);
        eq_base += e;
    }
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int base = s0; base < s1; base += 64) {
        const int i = base + lane;
        const bool in = i < s1;
        const unsigned key = in ? keys[i] : 0u;
        const bool gt = in && key > thr, eq = in && key == thr;
        const unsigned long long meq = __ballot(eq);
        const unsigned eq_rank = eq_base + (unsigned)__popcll(meq & lt_mask);
        const bool take = gt || (eq && eq_rank < need_eq);
        const unsigned long long msel = __ballot(take);
        if (take) {
            const unsigned pos = sel_base + (unsigned)__popcll(msel & lt_mask);
            decode_store<T>(reg, anchors, lv.off[level] + i, lv.out_off[level] + pos, __uint_as_float(key), level, means, stds, max_h,
                            max_w, max_ratio, out_scores, out_boxes, out_ids, img_off_in, img_off_out);
        }
        eq_base += (unsigned)__popcll(meq);
        sel_base += (unsigned)__popcll(msel);
    }
}

}  // namespace

extern "C" int64_t det_rpn_topk_decode_workspace_bytes(int64_t B, int64_t total_anchors) {
    return B * total_anchors * (int64_t)sizeof(unsigned);
}

// cls (B, total) logits and reg (B, total, 4) deltas, f32 | bf16, anchors in (level, h, w, a) order; anchors (total, 4) f32;
// level_sizes: HOST array of num_levels anchor counts.  Outputs (B, sum_l min(n_l, nms_pre)): scores f32 (sigmoid), boxes f32
// (decoded, clipped to [0,max_w] x [0,max_h] when max_w > 0), ids i64 (the level, = batched_nms's idxs).
extern "C" int det_rpn_topk_decode(const void* cls, const void* reg, const float* anchors, const int* level_sizes, int num_levels,
                                   int64_t B, int nms_pre, const float* means, const float* stds, float max_h, float max_w,
                                   void* workspace, float* out_scores, float* out_boxes, int64_t* out_ids, int dtype, void* stream) {
    if (B == 0 || num_levels == 0) return SWIN_OK;
    if (!cls || !reg || !anchors || !level_sizes || !means || !stds || !workspace || !out_scores || !out_boxes || !out_ids ||
        B < 0 || num_levels < 0 || num_levels > RS_MAX_LEVELS || nms_pre <= 0 || B > 65535)
        return SWIN_ERR_BAD_ARG;
    RsLevels lv;
    lv.L = num_levels;
    int64_t off = 0, ooff = 0;
    for (int l = 0; l < num_levels; ++l) {
        if (level_sizes[l] < 0) return SWIN_ERR_BAD_ARG;
        lv.off[l] = (int)off; lv.out_off[l] = (int)ooff;
        off += level_sizes[l];
        ooff += level_sizes[l] < nms_pre ? level_sizes[l] : nms_pre;
        if (off > 0x7fffffff) return SWIN_ERR_UNSUPPORTED;
    }
    lv.off[num_levels] = (int)off; lv.out_off[num_levels] = (int)ooff;
    for (int l = num_levels + 1; l <= RS_MAX_LEVELS; ++l) { lv.off[l] = (int)off; lv.out_off[l] = (int)ooff; }
    RsF4 m, sd;
    for (int q = 0; q < 4; ++q) { m.v[q] = means[q]; sd.v[q] = stds[q]; }
    const float mr = fabsf(logf(16.f / 1000.f));          // wh_ratio_clip of the coder (delta_xywh_bbox_coder.py:137)
    dim3 grid((unsigned)num_levels, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_F32)
        rpn_topk_decode_kernel<float><<<grid, RS_THREADS, 0, s>>>((const float*)cls, (const float*)reg, (const float4*)anchors, lv,
                                                                  nms_pre, m, sd, max_h, max_w, mr, (unsigned*)workspace, out_scores,
                                                                  (float4*)out_boxes, out_ids);
    else if (dtype == SWIN_BF16)
        rpn_topk_decode_kernel<bf16><<<grid, RS_THREADS, 0, s>>>((const bf16*)cls, (const bf16*)reg, (const float4*)anchors, lv,
                                                                 nms_pre, m, sd, max_h, max_w, mr, (unsigned*)workspace, out_scores,
                                                                 (float4*)out_boxes, out_ids);
    else return SWIN_ERR_UNSUPPORTED;
    return swin_launch_status();
}
