// RPN proposal selection on the device: RPNHead._get_bboxes, rpn_head.py:126-187.
//
// Per image and pyramid level the reference takes sigmoid of the objectness logits, sorts the level's anchors by score
// (descending) and keeps the first nms_pre, gathers their deltas and anchors, concatenates the levels and decodes
// (DeltaXYWHBBoxCoder.decode with max_shape).  Here: a radix select of the nms_pre-th largest score per (image, level) over
// the 32-bit score bits (11 + 11 + 10), then a compaction of the selected anchors IN ASCENDING ANCHOR ORDER that decodes each
// anchor as it is written.  Five small launches of 8192-anchor chunks (68 blocks at 2x800x1280) that meet through global
// histograms / chunk counts: the first version did everything in ONE block per (image, level) and took 350 us -- 192 000
// anchors x ~150 instructions on a single CU; the same work spread over the chip is ~10 us plus the launch gaps.
//
// Order contract: the reference hands batched_nms a list that is score-sorted inside each level; the NMS then sorts the
// whole list by score again (stable).  Elements with EQUAL scores therefore end up ordered by (level, anchor index) --
// exactly the order a stable sort produces from a level-major, index-ascending list.  At the nms_pre boundary a stable
// descending sort keeps, among the anchors tied with the threshold score, those with the LOWEST indices: the compaction
// below admits tied anchors in index order until the level's quota is full.  So the candidate SET and every later
// tie-break equal the reference's (checked against the CPU restatement of rpn_get_bboxes in the tests), without sorting here.
//
// HBM-bound integer/byte work: the logits are read once (2-4 B each), the 4-byte keys written once and re-read four
// times from L2; deltas and anchors are read for the selected anchors only.
#include "common.h"

namespace {

constexpr int RS_THREADS = 1024;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_MAX_LEVELS = 8;
constexpr int RS_PER = 8;                                   // anchors per thread and chunk
constexpr int RS_CHUNK = RS_THREADS * RS_PER;               // 8192

struct RsLevels {
    int L;
    int off[RS_MAX_LEVELS + 1];       // first anchor of each level in the flattened (level, h, w, a) order
    int out_off[RS_MAX_LEVELS + 1];   // first output slot of each level: sum of min(n_l, nms_pre)
    int chunk0[RS_MAX_LEVELS + 1];    // first chunk of each level (chunks of RS_CHUNK anchors, per image)
};
struct RsF4 { float v[4]; };

__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* sh /* RS_WAVES + 1 */) {
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
    return sh[w] + inc - v;
}

// from a 2^bits-bin histogram in LDS (bins descending = larger keys first): the bin holding the k-th largest key, and k
// within that bin -> sel[0], sel[1]
__device__ __forceinline__ void pick_bin(const unsigned* hist, int bits, unsigned k, unsigned* sh_scan, unsigned* sel) {
    const int nb = 1 << bits;                         // 2048 or 1024 bins, two or one per thread
    unsigned h0 = 0, h1 = 0;
    int b0 = -1, b1 = -1;
    if (nb == 2 * RS_THREADS) {
        b0 = nb - 1 - 2 * (int)threadIdx.x; b1 = b0 - 1;
        h0 = hist[b0]; h1 = hist[b1];
    } else {
        b0 = nb - 1 - (int)threadIdx.x;
        if (b0 >= 0) h0 = hist[b0];
    }
    const unsigned above = block_excl_scan(h0 + h1, sh_scan);
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

// PHASE 0: keys = bits of sigmoid(logit) (they order like the scores: >= 0) + histogram of bits 31..21   (levels with
//          n <= nms_pre are decoded here in anchor order: rpn_head.py:162 takes them all)
// PHASE 1: histogram of bits 20..10 among the keys that share the selected top digit
// PHASE 2: histogram of bits 9..0 among the keys that share the selected 22-bit prefix
// PHASE 3: per chunk, how many keys are above / equal to the nms_pre-th largest key
// PHASE 4: compaction + decode; a chunk's first output slot = the selected counts of the level's earlier chunks
// Every block re-derives the selected digits from the global histograms (a 2048-bin scan per digit): no host round trip and
// no extra "pick" launches.
template <typename T, int PHASE>
__global__ __launch_bounds__(RS_THREADS) void rpn_select_kernel(
        const T* __restrict__ cls, const T* __restrict__ reg, const float4* __restrict__ anchors, RsLevels lv, int nms_pre,
        RsF4 means, RsF4 stds, float max_h, float max_w, float max_ratio, unsigned* __restrict__ keys_ws,
        unsigned* __restrict__ ghist /* [B][L][3][2048] */, unsigned* __restrict__ counts /* [B][nchunks][2] */,
        float* __restrict__ out_scores, float4* __restrict__ out_boxes, int64_t* __restrict__ out_ids) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned sh_scan[RS_WAVES + 1];
    __shared__ unsigned sel[2];
    __shared__ unsigned wave_gt[RS_WAVES], wave_eq[RS_WAVES];
    const int chunk = blockIdx.x, img = blockIdx.y;
    int level = 0;
    while (level + 1 < lv.L && chunk >= lv.chunk0[level + 1]) ++level;
    const int n = lv.off[level + 1] - lv.off[level];
    const int lo = (chunk - lv.chunk0[level]) * RS_CHUNK, hi = min(n, lo + RS_CHUNK);
    const int64_t total = lv.off[lv.L], total_out = lv.out_off[lv.L];
    const int64_t in0 = (int64_t)img * total + lv.off[level];        // this (image, level)'s first logit / key
    const int64_t img_off_in = (int64_t)img * total, img_off_out = (int64_t)img * total_out;
    const T* c = cls + in0;
    unsigned* keys = keys_ws + in0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchunks = lv.chunk0[lv.L];

    if (n <= nms_pre) {                   // no selection for this level
        if (PHASE == 0) {
            for (int i = lo + threadIdx.x; i < hi; i += RS_THREADS) {
                const float s = 1.0f / (1.0f + expf(-Elt<T>::ld(c + i)));
                decode_store<T>(reg, anchors, lv.off[level] + i, lv.out_off[level] + i, s, level, means, stds, max_h, max_w,
                                max_ratio, out_scores, out_boxes, out_ids, img_off_in, img_off_out);
            }
        }
        return;
    }
    unsigned* gh = ghist + ((int64_t)img * lv.L + level) * 3 * 2048;
    // this thread's RS_PER keys: element u is anchor lo + u * RS_THREADS + tid
    unsigned kk[RS_PER];
    if (PHASE == 0) {
        float xv[RS_PER];
#pragma unroll
        for (int u = 0; u < RS_PER; ++u) {
            const int i = lo + u * RS_THREADS + threadIdx.x;
            xv[u] = Elt<T>::ld(c + (i < hi ? i : hi - 1));
        }
#pragma unroll
        for (int u = 0; u < RS_PER; ++u) {
            const int i = lo + u * RS_THREADS + threadIdx.x;
            kk[u] = __float_as_uint(1.0f / (1.0f + expf(-xv[u])));
            if (i < hi) keys[i] = kk[u];
        }
    } else if (PHASE < 4) {
#pragma unroll
        for (int u = 0; u < RS_PER; ++u) {
            const int i = lo + u * RS_THREADS + threadIdx.x;
            kk[u] = keys[i < hi ? i : hi - 1];
        }
    }
    // ---- the digits selected so far, from the global histograms of the earlier phases ----
    unsigned prefix = 0, k = (unsigned)nms_pre;
    constexpr int NPICK = PHASE < 3 ? PHASE : 3;
    const int nbits[3] = {11, 11, 10};
#pragma unroll
    for (int p = 0; p < NPICK; ++p) {
        for (int i = threadIdx.x; i < 2048; i += RS_THREADS) hist[i] = gh[p * 2048 + i];
        __syncthreads();
        pick_bin(hist, nbits[p], k, sh_scan, sel);
        prefix = (prefix << nbits[p]) | sel[0];
        k = sel[1];
        __syncthreads();
    }
    if (PHASE < 3) {
        // ---- histogram of this phase's digit over the chunk -> global ----
        const int shift = PHASE == 0 ? 21 : (PHASE == 1 ? 10 : 0), bits = nbits[PHASE];
        const unsigned dmask = (1u << bits) - 1u;
        for (int i = threadIdx.x; i < 2048; i += RS_THREADS) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RS_PER; ++u) {
            const int i = lo + u * RS_THREADS + threadIdx.x;
            const unsigned key = kk[u];
            bool act = i < hi;
            if (PHASE > 0) act = act && ((key >> (shift + bits)) == prefix);
            const unsigned digit = (key >> shift) & dmask;
            // lanes of a wave that carry the same digit add once (scores of a fresh head fall into two or three bins)
            unsigned long long m = __ballot(act);
#pragma unroll 1
            for (int round = 0; round < 3 && m != 0; ++round) {
                const int first = __ffsll((long long)m) - 1;
                const unsigned d0 = __shfl(digit, first);
                const unsigned long long same = __ballot(act && digit == d0);
                if (lane == first) atomicAdd(&hist[d0], (unsigned)__popcll(same));
                if (digit == d0) act = false;
                m &= ~same;
            }
            if (m != 0 && act) atomicAdd(&hist[digit], 1u);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2048; i += RS_THREADS) {
            const unsigned v = hist[i];
            if (v) atomicAdd(&gh[PHASE * 2048 + i], v);
        }
        return;
    }
    const unsigned thr = prefix;          // the nms_pre-th largest key; `k` of the keys equal to it are admitted
    const unsigned need_eq = k;
    if (PHASE == 3) {
        unsigned cgt = 0, ceq = 0;
#pragma unroll
        for (int u = 0; u < RS_PER; ++u) {
            const int i = lo + u * RS_THREADS + threadIdx.x;
            cgt += (unsigned)__popcll(__ballot(i < hi && kk[u] > thr));
            ceq += (unsigned)__popcll(__ballot(i < hi && kk[u] == thr));
        }
        if (lane == 0) { wave_gt[w] = cgt; wave_eq[w] = ceq; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned a = 0, b = 0;
            for (int v = 0; v < RS_WAVES; ++v) { a += wave_gt[v]; b += wave_eq[v]; }
            counts[((int64_t)img * nchunks + chunk) * 2] = a;
            counts[((int64_t)img * nchunks + chunk) * 2 + 1] = b;
        }
        return;
    }
    // ---- PHASE 4: compaction in ascending anchor order; a contiguous 512-anchor segment per wave ----
    unsigned eq_base = 0, sel_base = 0;
    for (int cprev = lv.chunk0[level]; cprev < chunk; ++cprev) {          // the level's earlier chunks (<= 24)
        const unsigned g = counts[((int64_t)img * nchunks + cprev) * 2], e = counts[((int64_t)img * nchunks + cprev) * 2 + 1];
        const unsigned room = need_eq > eq_base ? need_eq - eq_base : 0u;
        sel_base += g + (e < room ? e : room);
        eq_base += e;
    }
    const int s0 = lo + w * (RS_CHUNK / RS_WAVES), s1 = min(hi, s0 + RS_CHUNK / RS_WAVES);
    unsigned kq[RS_PER];
    unsigned cgt = 0, ceq = 0;
#pragma unroll
    for (int u = 0; u < RS_PER; ++u) {
        const int i = s0 + u * 64 + lane;
        kq[u] = keys[i < s1 ? i : (s1 > s0 ? s1 - 1 : lo)];
        cgt += (unsigned)__popcll(__ballot(i < s1 && kq[u] > thr));
        ceq += (unsigned)__popcll(__ballot(i < s1 && kq[u] == thr));
    }
    if (lane == 0) { wave_gt[w] = cgt; wave_eq[w] = ceq; }
    __syncthreads();
    for (int v = 0; v < w; ++v) {
        const unsigned e = wave_eq[v];
        const unsigned room = need_eq > eq_base ? need_eq - eq_base : 0u;
        sel_base += wave_gt[v] + (e < room ? e : room);
        eq_base += e;
    }
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int u = 0; u < RS_PER; ++u) {
        const int i = s0 + u * 64 + lane;
        const bool in = i < s1;
        const unsigned key = kq[u];
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

template <typename T>
int rpn_select_launch(const void* cls, const void* reg, const float* anchors, const RsLevels& lv, int64_t B, int nms_pre, RsF4 m, RsF4 sd,
                      float max_h, float max_w, float mr, void* workspace, float* out_scores, float* out_boxes, int64_t* out_ids,
                      hipStream_t s) {
    const int64_t total = lv.off[lv.L];
    const int nchunks = lv.chunk0[lv.L];
    unsigned* keys = (unsigned*)workspace;
    unsigned* ghist = keys + ((B * total + 3) / 4) * 4;
    const size_t hist_bytes = (size_t)B * lv.L * 3 * 2048 * sizeof(unsigned);
    unsigned* counts = ghist + (size_t)B * lv.L * 3 * 2048;
    if (hipMemsetAsync(ghist, 0, hist_bytes, s) != hipSuccess) return SWIN_ERR_LAUNCH;
    dim3 grid((unsigned)nchunks, (unsigned)B);
#define RS_ARGS (const T*)cls, (const T*)reg, (const float4*)anchors, lv, nms_pre, m, sd, max_h, max_w, mr, keys, ghist, counts, \
                out_scores, (float4*)out_boxes, out_ids
    rpn_select_kernel<T, 0><<<grid, RS_THREADS, 0, s>>>(RS_ARGS);
    rpn_select_kernel<T, 1><<<grid, RS_THREADS, 0, s>>>(RS_ARGS);
    rpn_select_kernel<T, 2><<<grid, RS_THREADS, 0, s>>>(RS_ARGS);
    rpn_select_kernel<T, 3><<<grid, RS_THREADS, 0, s>>>(RS_ARGS);
    rpn_select_kernel<T, 4><<<grid, RS_THREADS, 0, s>>>(RS_ARGS);
#undef RS_ARGS
    return swin_launch_status();
}

}  // namespace

static int64_t rs_chunks(int64_t n) { return (n + RS_CHUNK - 1) / RS_CHUNK; }

extern "C" int64_t det_rpn_topk_decode_workspace_bytes(int64_t B, int64_t total_anchors) {
    // keys (4 B per logit) + 3 histograms of 2048 bins per (image, level) + (above, equal) counts per chunk; the chunk count is
    // bounded by total / RS_CHUNK + one partial chunk per level
    const int64_t keys = ((B * total_anchors + 3) / 4) * 4;
    const int64_t hist = B * RS_MAX_LEVELS * 3 * 2048;
    const int64_t counts = B * (rs_chunks(total_anchors) + RS_MAX_LEVELS) * 2;
    return (keys + hist + counts) * (int64_t)sizeof(unsigned);
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
    int64_t off = 0, ooff = 0, ch = 0;
    for (int l = 0; l < num_levels; ++l) {
        if (level_sizes[l] < 0) return SWIN_ERR_BAD_ARG;
        lv.off[l] = (int)off; lv.out_off[l] = (int)ooff; lv.chunk0[l] = (int)ch;
        off += level_sizes[l];
        ooff += level_sizes[l] < nms_pre ? level_sizes[l] : nms_pre;
        ch += rs_chunks(level_sizes[l]);
        if (off > 0x7fffffff) return SWIN_ERR_UNSUPPORTED;
    }
    for (int l = num_levels; l <= RS_MAX_LEVELS; ++l) { lv.off[l] = (int)off; lv.out_off[l] = (int)ooff; lv.chunk0[l] = (int)ch; }
    if (ch == 0) return SWIN_OK;
    RsF4 m, sd;
    for (int q = 0; q < 4; ++q) { m.v[q] = means[q]; sd.v[q] = stds[q]; }
    const float mr = fabsf(logf(16.f / 1000.f));          // wh_ratio_clip of the coder (delta_xywh_bbox_coder.py:137)
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_F32)
        return rpn_select_launch<float>(cls, reg, anchors, lv, B, nms_pre, m, sd, max_h, max_w, mr, workspace, out_scores, out_boxes, out_ids, s);
    if (dtype == SWIN_BF16)
        return rpn_select_launch<bf16>(cls, reg, anchors, lv, B, nms_pre, m, sd, max_h, max_w, mr, workspace, out_scores, out_boxes, out_ids, s);
    return SWIN_ERR_UNSUPPORTED;
}
