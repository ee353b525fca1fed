// The small parameter-gradient reductions behind a Swin block's backward, as ONE launch.
//
// A block backward (swin_transformer.py:157-232 differentiated) ends in four reductions nobody on the data-gradient chain waits
// for: the [dgamma | dbeta] partial rows of norm2 and of the following norm (one row per LayerNorm-backward thread block), and the
// relative-position-bias gradient -- the attention backward's per-wave [key][query] slabs summed per head and scattered into the
// (169, nH) table (swin_transformer.py:105-110) together with the pad-token part of the qkv bias gradient.  As separate launches
// (2 x ln_param_reduce, dbias_slab_reduce behind a memset of its (nH,64,64) target, rel_bias_reduce) they were 60 of a step's 518
// launches at 5-7 us each, most of it launch latency.  Here they are problems of one table-driven launch: every entry point that
// would launch such a reduction pushes a SwinTailProb instead while a collection is open (swin_tail_collect), and the block
// runner flushes the table behind its data-gradient chain (swin_tail_flush).
#include <vector>

#include "common.h"

namespace {

constexpr int TAIL_MAX = 48;
struct TailTab {
    int n;
    SwinTailProb p[TAIL_MAX];
};

constexpr int W_TILE = 64, W_TOK = 49, W_HD = 32;      // window_attn.hip: TILE, NTOK, HD (static_asserts there keep them equal)

__global__ __launch_bounds__(1024) void tail_reduce_kernel(TailTab tab) {
    __shared__ float red[16][64];
    __shared__ float acc[169];
    int pi = 0;
    while (pi + 1 < tab.n && (int)blockIdx.x >= tab.p[pi + 1].blk0) ++pi;
    const SwinTailProb P = tab.p[pi];
    const int lb = blockIdx.x - P.blk0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (P.kind == SWIN_TAIL_COLSUM) {
        // dst0[i] / dst1[i - a0] += sum over the rows of src[rows][cols]; a block owns 64 columns, its 16 waves split the rows
        const int i = lb * 64 + lane;
        float a = 0.f;
        if (i < P.cols) {
            const float* p = P.src + i;
            const int64_t ld = P.cols;
            int b = w;
            for (; b + 48 < P.rows; b += 64) {                  // 4 independent loads in flight per lane
                const float v0 = p[b * ld], v1 = p[(b + 16) * ld], v2 = p[(b + 32) * ld], v3 = p[(b + 48) * ld];
                a += (v0 + v1) + (v2 + v3);
            }
            for (; b < P.rows; b += 16) a += p[b * ld];
        }
        red[w][lane] = a;
        __syncthreads();
        if (w == 0 && i < P.cols) {
            a = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) a += red[q][lane];
            float* dst = i < P.a0 ? P.dst0 + i : P.dst1 + (i - P.a0);
            *dst += a;                                           // the only writer of this element in this launch
        }
        return;
    }
    // SWIN_TAIL_RELBIAS: src = slabs [rows][cols] (slab s belongs to head s % nH; [key][query] tile, then 3*HD pad-token sums);
    // block (head, y) sums its share of the head's slabs and adds into dtable (169, nH) and dbias_pad (3, C)
    const int nH = P.a0, C = P.a1;
    const int per_head = P.rows / nH;
    const int ys = per_head < 16 ? per_head : 16;
    const int head = lb / ys, y = lb - head * ys;
    const int per = (per_head + ys - 1) / ys;
    const int k0 = y * per, k1 = min(k0 + per, per_head);
    if (threadIdx.x < 169) acc[threadIdx.x] = 0.f;
    __syncthreads();
    const int64_t ld = (int64_t)P.cols * nH;
    for (int e = threadIdx.x; e < W_TILE * W_TILE + 3 * W_HD; e += 1024) {
        const int k = e >> 6, q = e & 63;
        const bool tile = e < W_TILE * W_TILE;
        if (tile && (k >= W_TOK || q >= W_TOK)) continue;
        if (!tile && !P.dst1) continue;
        const float* p = P.src + (int64_t)head * P.cols + e;
        float a = 0.f;
        int s = k0;
        for (; s + 3 < k1; s += 4) a += (p[s * ld] + p[(s + 1) * ld]) + (p[(s + 2) * ld] + p[(s + 3) * ld]);
        for (; s < k1; ++s) a += p[s * ld];
        if (tile) {
            const int qh = q / 7, qw = q % 7, kh = k / 7, kw = k % 7;
            atomicAdd(&acc[(qh - kh + 6) * 13 + (qw - kw + 6)], a);      // swin_transformer.py:105-110
        } else {
            const int r = e - W_TILE * W_TILE;
            atomicAdd(P.dst1 + (r / W_HD) * C + head * W_HD + (r % W_HD), a);
        }
    }
    __syncthreads();
    if (threadIdx.x < 169 && k1 > k0) atomicAdd(P.dst0 + threadIdx.x * nH + head, acc[threadIdx.x]);
}

bool g_on[16] = {};
std::vector<SwinTailProb> g_tab[16];

int cur_dev() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? dev : -1;
}

int blocks_of(const SwinTailProb& p) {
    if (p.kind == SWIN_TAIL_COLSUM) return (p.cols + 63) / 64;
    const int per_head = p.rows / p.a0;
    return p.a0 * (per_head < 16 ? per_head : 16);
}

}  // namespace

int swin_tail_launch(const SwinTailProb* probs, int n, void* stream) {
    for (int i0 = 0; i0 < n; i0 += TAIL_MAX) {
        TailTab tab;
        tab.n = n - i0 < TAIL_MAX ? n - i0 : TAIL_MAX;
        int total = 0;
        for (int i = 0; i < tab.n; ++i) {
            tab.p[i] = probs[i0 + i];
            const SwinTailProb& p = tab.p[i];
            if (!p.src || !p.dst0 || p.rows <= 0 || p.cols <= 0) return SWIN_ERR_BAD_ARG;
            if (p.kind == SWIN_TAIL_COLSUM ? (p.a0 < p.cols && !p.dst1) : (p.a0 <= 0 || p.rows % p.a0 != 0 || p.cols < W_TILE * W_TILE + 3 * W_HD))
                return SWIN_ERR_BAD_ARG;
            tab.p[i].blk0 = total;
            total += blocks_of(p);
        }
        tail_reduce_kernel<<<total, 1024, 0, (hipStream_t)stream>>>(tab);
    }
    return swin_launch_status();
}

void swin_tail_collect(bool on) {
    const int dev = cur_dev();
    if (dev < 0) return;
    g_on[dev] = on;
    if (!on) g_tab[dev].clear();
}

bool swin_tail_push(const SwinTailProb& p) {
    const int dev = cur_dev();
    if (dev < 0 || !g_on[dev]) return false;
    g_tab[dev].push_back(p);
    return true;
}

int swin_tail_flush(void* stream) {
    const int dev = cur_dev();
    if (dev < 0) return SWIN_ERR_UNSUPPORTED;
    if (g_tab[dev].empty()) return SWIN_OK;
    const int rc = swin_tail_launch(g_tab[dev].data(), (int)g_tab[dev].size(), stream);
    g_tab[dev].clear();
    return rc;
}

// C ABI (tests, callers outside the block runner): `n` problems given as parallel arrays
extern "C" int swin_tail_reduce(const int* kind, const float* const* src, float* const* dst0, float* const* dst1, const int* rows,
                                const int* cols, const int* a0, const int* a1, int n, void* stream) {
    if (!kind || !src || !dst0 || !dst1 || !rows || !cols || !a0 || !a1 || n <= 0) return SWIN_ERR_BAD_ARG;
    std::vector<SwinTailProb> v((size_t)n);
    for (int i = 0; i < n; ++i) v[i] = SwinTailProb{src[i], dst0[i], dst1[i], kind[i], rows[i], cols[i], a0[i], a1[i], 0};
    return swin_tail_launch(v.data(), n, stream);
}
