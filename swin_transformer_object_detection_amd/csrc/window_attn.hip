// Window attention core for gfx950: cyclic shift + pad + window partition/reverse +
// relative-position bias + shift mask + softmax + PV in one kernel, reading the qkv
// projection and writing the attention output on the NATURAL (B,H,W) token grid.
//
// Reference semantics: mmdet/models/backbones/swin_transformer.py:129-150 (core),
// :214-231 / :237-247 (pad, roll, partition, reverse, crop), :371-389 (mask).
//
// bf16 path (one wave per (window, head), persistent over windows of one head):
//   S^T = K Q^T   as 2x2 tiles of v_mfma_f32_32x32x16_bf16 (49 -> 64 padding), query on the
//   lane, key on the accumulator registers, so the softmax reduction is in-lane plus one
//   lane^32 exchange and P^T is directly the B operand of  O^T = V^T P^T  (no LDS trip).
//   The expanded bias tile of the wave's head stays in 64 VGPRs across its windows.
// fp32 path: exact-fp32 VALU kernel (parity path for the 1e-4 gate).
#include <cstdlib>

#include "common.h"
#include "aux_defer.h"

extern "C" int swin_fork_stream(void* main, void* side);

#define NTOK 49
#define TILE 64
#define HD 32
#define LROW 40  // LDS row stride in bf16 elements (80 B: conflict-free ds_read_b128 of 32-wide rows)
#define NEG_BIG (-30000.0f)

struct WinGeom {
    int B, H, W, C, nH, shift, Hp, Wp, nWh, nWw, nW;
};

static inline WinGeom make_geom(int B, int H, int W, int C, int nH, int shift) {
    WinGeom g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.nH = nH; g.shift = shift;
    g.nWh = (H + 6) / 7; g.nWw = (W + 6) / 7;
    g.Hp = g.nWh * 7; g.Wp = g.nWw * 7; g.nW = g.nWh * g.nWw;
    return g;
}

// token `t` (0..48, row-major 7x7) of window (b, wr, wc) on the shifted padded grid ->
// linear token index on the natural grid, or -1 for a padded token.
__device__ __forceinline__ int token_src(const WinGeom& g, int b, int wr, int wc, int t) {
    int th = t / 7, tw = t - th * 7;
    int r = wr * 7 + th + g.shift; if (r >= g.Hp) r -= g.Hp;   // roll(-shift): shifted[r] = x[(r+shift)%Hp]
    int c = wc * 7 + tw + g.shift; if (c >= g.Wp) c -= g.Wp;
    if (r >= g.H || c >= g.W) return -1;
    return (b * g.H + r) * g.W + c;
}

// Window coordinates of a persistent wave's tasks.  Successive tasks of a wave are `n_waves / nH` windows apart; the
// (image, window row, window column) triple is advanced by that constant stride with carries instead of being
// re-derived from the task index with three integer divisions per task (and three more for the prefetched one).
struct WinPos { int b, wr, wc; };
struct WinStride { int db, dwr, dwc; };
__device__ __forceinline__ WinPos win_decode(const WinGeom& g, int win) {
    WinPos p;
    p.b = win / g.nW;
    const int wrem = win - p.b * g.nW;
    p.wr = wrem / g.nWw; p.wc = wrem - p.wr * g.nWw;
    return p;
}
__device__ __forceinline__ void win_advance(WinPos& p, const WinStride& s, const WinGeom& g) {
    p.wc += s.dwc;
    if (p.wc >= g.nWw) { p.wc -= g.nWw; p.wr += 1; }
    p.wr += s.dwr;
    if (p.wr >= g.nWh) { p.wr -= g.nWh; p.b += 1; }
    p.b += s.db;
}
__device__ __forceinline__ bool win_interior(const WinGeom& g, const WinPos& p) {
    return (p.wr * 7 + 6 + g.shift < g.H) && (p.wc * 7 + 6 + g.shift < g.W);
}

// ------------------------------------------------------------------------------------
// relative position bias: table (169,nH) <-> expanded (nH,64,64) [head][key][query]
// ------------------------------------------------------------------------------------
__global__ void rel_bias_expand_kernel(const float* __restrict__ table, float* __restrict__ out, int nH) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nH * TILE * TILE) return;
    int q = i & 63, k = (i >> 6) & 63, h = i >> 12;
    float v = NEG_BIG;
    if (k < NTOK) {
        v = 0.f;
        if (q < NTOK) {
            int qh = q / 7, qw = q % 7, kh = k / 7, kw = k % 7;
            int idx = (qh - kh + 6) * 13 + (qw - kw + 6);   // swin_transformer.py:105-110
            v = table[idx * nH + h];
        }
    }
    out[i] = v;
}

// one block per head: the 49 x 49 (query, key) gradients are scattered into a 169-entry LDS table (each entry collects
// <= 49 pairs) and added to dtable.  (The first version gave each of the 169 x nH outputs one thread that walked its
// <= 49 pairs serially: 13 us of dependent strided loads for 2 KB of output, twelve times per step.)
__global__ __launch_bounds__(256) void rel_bias_reduce_kernel(const float* __restrict__ dexp, float* __restrict__ dtable, int nH) {
    __shared__ float acc[169];
    const int h = blockIdx.x, t = threadIdx.x;
    if (t < 169) acc[t] = 0.f;
    __syncthreads();
    for (int i = t; i < NTOK * NTOK; i += 256) {
        const int k = i / NTOK, q = i - k * NTOK;
        const int qh = q / 7, qw = q % 7, kh = k / 7, kw = k % 7;
        atomicAdd(&acc[(qh - kh + 6) * 13 + (qw - kw + 6)], dexp[(h * TILE + k) * TILE + q]);   // swin_transformer.py:105-110
    }
    __syncthreads();
    if (t < 169) dtable[t * nH + h] += acc[t];
}

// ------------------------------------------------------------------------------------
// bf16 forward
// ------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 bias_to_bf16x8(const float* p) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16)p[e];
    return r;
}

// mask bits for the 64 accumulator elements a lane owns: bit (kt*2+qt)*16+reg set when query and key
// fall on different sides of the "last 3 rows" (resp. columns) boundary inside a window.
__device__ __forceinline__ void lane_mask_bits(int lane, uint64_t& mrow, uint64_t& mcol) {
    int c = lane & 31, h = lane >> 5;
    mrow = 0; mcol = 0;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int key = 32 * kt + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                int q = 32 * qt + c;
                int qh = q / 7, qw = q % 7, kh = key / 7, kw = key % 7;
                uint64_t bit = 1ull << ((kt * 2 + qt) * 16 + reg);
                if ((qh >= 4) != (kh >= 4)) mrow |= bit;
                if ((qw >= 4) != (kw >= 4)) mcol |= bit;
            }
}

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// transposed fragment: 8 rows {row0 + 8*(j>>2) + 4*h + (j&3)} of column `c` of a [rows][LROW] bf16 tile, via two
// ds_read_b64_tr_b16 (each 16-lane group fetches a 4-row x 16-column block and gets it column-major).
// Pairs with an accumulator-as-operand whose k order is the 32x32 C-layout row order.
__device__ __forceinline__ bf16x8 lds_tr_frag_perm(const bf16 (*tile)[LROW], int row0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, dh = g & 1;
    const bf16* a0 = &tile[row0 + 4 * h + q][16 * dh + 4 * p];
    bf16x4 lo = SWIN_DS_READ_TR16((lds_bf16x4*)a0);
    bf16x4 hi = SWIN_DS_READ_TR16((lds_bf16x4*)(a0 + 8 * LROW));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
}

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// combine a value with the one held by lane^32 using v_permlane32_swap (no LDS round trip)
__device__ __forceinline__ float half_swap_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Per-wave constant state of the forward kernel.
struct FwdLane {
    int lane, c, h;
    int tokr, which, part;        // staging role: token-in-round, q|k|v, 16-byte piece
    unsigned ld_off[10];          // interior fast path: element offset of staged piece i relative to the window base
    unsigned st_off[2];           // interior fast path: element offset of this lane's output row (per query tile)
};

// One (window, head) task.  INTERIOR windows (no wrap-around of the cyclic shift, no padding) use
// precomputed per-lane offsets from a wave-uniform base; the general path resolves every token.
template <bool INTERIOR>
__device__ __forceinline__ void fwd_issue_loads(bf16x8 (&stg)[10], const FwdLane& L, const WinGeom& g, const bf16* __restrict__ qkv,
                                                const float* __restrict__ qkv_bias, int head, int b, int wr, int wc) {
    const int C3 = 3 * g.C;
    const int ch = L.which * g.C + head * HD + L.part * 8;
    // Every lane issues every load (offsets of the unused tail pieces are clamped to the last token; only the LDS write
    // is predicated): a load inside a divergent `if` makes the compiler close the region with s_waitcnt vmcnt(0), i.e.
    // the wave would sit out the full latency of the prefetch it has just issued.
    if (INTERIOR) {
        const bf16* pb = qkv + ((size_t)(b * g.H + wr * 7 + g.shift) * g.W + wc * 7 + g.shift) * C3 + ch;
#pragma unroll
        for (int i = 0; i < 10; ++i) stg[i] = *(const bf16x8*)(pb + L.ld_off[i]);
    } else {
        // general path (wrap-around / padding): padded tokens read qkv.bias (their q|k|v); clamped address, fixed up by VALUE
        const bf16x8 padv = bias_to_bf16x8(qkv_bias + ch);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            int t = 5 * i + L.tokr; if (t >= NTOK) t = NTOK - 1;
            const int src = token_src(g, b, wr, wc, t);
            const bf16x8 v = *(const bf16x8*)(qkv + (size_t)(src >= 0 ? src : 0) * C3 + ch);
            stg[i] = src >= 0 ? v : padv;
        }
    }
}

__global__ __launch_bounds__(256, 2) void win_attn_fwd_bf16_kernel(
    const bf16* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_exp,
    bf16* __restrict__ out, float* __restrict__ lse, WinGeom g, float scale, int n_tasks, int wpb) {
    __shared__ __attribute__((aligned(16))) bf16 lds[4][3][TILE][LROW];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // wpb = waves of the block that take tasks (3 when nH == 3, else 4).  With three heads and four working waves every second
    // window has one head in another block -- usually on another XCD -- than the other two; the heads' 64-byte slices interleave
    // in the token rows (q.h0|q.h1, q.h2|k.h0, ... share 128-byte lines), so that head's block fetches every shared line a second
    // time from HBM: the 1.29x traffic of the stage-1 launch.  Three working waves keep a window's heads in one block (no block-wide
    // barrier in this kernel: the fourth wave just leaves).
    if (wave >= wpb) return;
    FwdLane L;
    L.lane = threadIdx.x & 63; L.c = L.lane & 31; L.h = L.lane >> 5;
    const int lane = L.lane, c = L.c, h = L.h;
    bf16(*Qs)[LROW] = lds[wave][0];
    bf16(*Ks)[LROW] = lds[wave][1];
    bf16(*Vs)[LROW] = lds[wave][2];
    {
        uint4 z = {0, 0, 0, 0};
        uint4* p = (uint4*)&lds[wave][0][0][0];
        for (int i = lane; i < 3 * TILE * LROW * 2 / 16; i += WAVE) p[i] = z;   // rows 49..63 stay zero
    }
    const int n_waves = gridDim.x * wpb;   // host guarantees n_waves % nH == 0
    int task = blockIdx.x * wpb + wave;
    if (task >= n_tasks) return;
    const int head = task % g.nH;

    // expanded bias of this head in accumulator layout, kept in registers in the RAW-logit domain (bias / scale): it is
    // the C operand of the first S MFMA, so the accumulators need no zero fill and no separate bias pass; the softmax
    // scale (and log2 e) is applied once, inside the exponent's fused multiply-add
    f32x16 biasr[2][2];
    {
        const float* bp = bias_exp + (size_t)head * TILE * TILE;
        const float rs = 1.0f / scale;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    int key = 32 * kt + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    biasr[kt][qt][reg] = bp[key * TILE + 32 * qt + c] * rs;
                }
    }
    uint64_t mrow = 0, mcol = 0;
    if (g.shift > 0) lane_mask_bits(lane, mrow, mcol);
    const float sl2 = scale * LOG2E;
    const float mask_raw = -100.0f / scale;

    // staging role: 5 tokens per round x 12 16-byte pieces (q|k|v x 4); lanes 60..63 duplicate lanes 48..51
    {
        const int sl = lane < 60 ? lane : lane - 12;
        L.tokr = sl / 12;
        const int rem = sl - L.tokr * 12;
        L.which = rem >> 2; L.part = rem & 3;
        const int C3 = 3 * g.C;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            int t = 5 * i + L.tokr; if (t >= NTOK) t = NTOK - 1;
            L.ld_off[i] = (unsigned)(((t / 7) * g.W + (t % 7)) * C3);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            int q = 32 * qt + c; if (q >= NTOK) q = NTOK - 1;
            L.st_off[qt] = (unsigned)(((q / 7) * g.W + (q % 7)) * g.C);
        }
    }
    const int which = L.which, part = L.part, tokr = L.tokr;

    WinPos cur = win_decode(g, task / g.nH);
    WinStride stride;
    {
        const WinPos d = win_decode(g, n_waves / g.nH);          // n_waves % nH == 0
        stride.db = d.b; stride.dwr = d.wr; stride.dwc = d.wc;
    }
    WinPos nxt = cur;
    win_advance(nxt, stride, g);

    bf16x8 stg[10];
    {
        if (win_interior(g, cur)) fwd_issue_loads<true>(stg, L, g, qkv, qkv_bias, head, cur.b, cur.wr, cur.wc);
        else fwd_issue_loads<false>(stg, L, g, qkv, qkv_bias, head, cur.b, cur.wr, cur.wc);
    }
    __builtin_amdgcn_wave_barrier();

    for (; task < n_tasks; task += n_waves, cur = nxt, win_advance(nxt, stride, g)) {
        const int b = cur.b, wr = cur.wr, wc = cur.wc;
        const bool interior = win_interior(g, cur);

        // ---- registers -> LDS (q,k,v head slices, 49 x 32 each), then prefetch the next task ----
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            int t = 5 * i + tokr;
            if (i < 9 || tokr < 4) *(bf16x8*)&lds[wave][which][t][part * 8] = stg[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (task + n_waves < n_tasks) {
            if (win_interior(g, nxt)) fwd_issue_loads<true>(stg, L, g, qkv, qkv_bias, head, nxt.b, nxt.wr, nxt.wc);
            else fwd_issue_loads<false>(stg, L, g, qkv, qkv_bias, head, nxt.b, nxt.wr, nxt.wc);
        }

        const bool edge = g.shift > 0 && (wr == g.nWh - 1 || wc == g.nWw - 1);
        uint64_t mbits = 0;
        if (edge) {
            if (wr == g.nWh - 1) mbits |= mrow;
            if (wc == g.nWw - 1) mbits |= mcol;
        }
        bf16x8 kf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) kf[kt][s] = *(const bf16x8*)&Ks[32 * kt + c][16 * s + 8 * h];

#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // ---- S^T = K Q^T for 32 queries --------------------------------------------------
            bf16x8 qf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) qf[s] = *(const bf16x8*)&Qs[32 * qt + c][16 * s + 8 * h];
            f32x16 sacc[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 a = SWIN_MFMA_32x32x16(kf[kt][0], qf[0], biasr[kt][qt], 0, 0, 0);   // + bias / scale
                sacc[kt] = SWIN_MFMA_32x32x16(kf[kt][1], qf[1], a, 0, 0, 0);
            }
            // ---- raw logits (q.k + bias / scale); softmax over keys with the scale folded into the exponent ------
            if (edge) {                                       // wave-uniform: only last-row / last-column windows
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        if (!(kt == 1 && reg >= 9) && ((mbits >> ((kt * 2 + qt) * 16 + reg)) & 1)) sacc[kt][reg] += mask_raw;   // -100 / scale  (:389)
            }
            // 4 independent partial reductions (ILP), then a VALU half-swap instead of an LDS bpermute
            // keys 49..63 are padding: in the accumulator layout (key = 32 kt + (reg & 3) + 8 (reg >> 2) + 4 h) that is
            // kt == 1, reg >= 9 for every lane (reg 8 is key 48 for h == 0) -- resolved at compile time, those 7 of the 32
            // elements never enter the max / exp / sum and are zero in P^T
            float m4[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    if (!(kt == 1 && reg >= 9)) m4[reg & 3] = fmaxf(m4[reg & 3], sacc[kt][reg]);
            float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
            m = half_swap_max(m);
            // p = exp2(s * sl2 - m * sl2): one packed fma + one exp per element pair / element; packed partial sums
            const f32x2 sc2 = {sl2, sl2};
            const float mc = -m * sl2;
            const f32x2 mc2 = {mc, mc};
            f32x2 sum2[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    if (kt == 1 && r2 >= 5) { sacc[kt][2 * r2] = 0.f; sacc[kt][2 * r2 + 1] = 0.f; continue; }   // padded keys
                    f32x2 x = {sacc[kt][2 * r2], sacc[kt][2 * r2 + 1]};
                    x = __builtin_elementwise_fma(x, sc2, mc2);
                    f32x2 p = {__builtin_amdgcn_exp2f(x[0]), (kt == 1 && r2 == 4) ? 0.f : __builtin_amdgcn_exp2f(x[1])};
                    sacc[kt][2 * r2] = p[0]; sacc[kt][2 * r2 + 1] = p[1];
                    sum2[r2 & 1] += p;
                }
            const f32x2 st = sum2[0] + sum2[1];
            float sum = half_swap_sum(st[0] + st[1]);
            const float inv = 1.0f / sum;
            m = m * sl2;                                      // log2-domain maximum for the saved log-sum-exp

            // ---- O^T = V^T P^T  (P^T accumulator registers are the B operand) -----------------------
            // two independent accumulation chains (one per key tile) instead of one chain of four dependent MFMAs
            f32x16 oacc2[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 o = {0};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 vf = lds_tr_frag_perm(Vs, 32 * kt + 16 * s, lane);
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (bf16)sacc[kt][8 * s + j];
                    o = SWIN_MFMA_32x32x16(vf, pf, o, 0, 0, 0);
                }
                oacc2[kt] = o;
            }
            const f32x16 oacc = oacc2[0] + oacc2[1];

            // ---- write O (window_reverse + roll back + crop == scatter to the source position) ----
            const int q = 32 * qt + c;
            bf16* op = nullptr;
            if (q < NTOK) {
                if (interior) {
                    op = out + ((size_t)(b * g.H + wr * 7 + g.shift) * g.W + wc * 7 + g.shift) * g.C + L.st_off[qt] + head * HD + 4 * h;
                } else {
                    int src = token_src(g, b, wr, wc, q);
                    if (src >= 0) op = out + (size_t)src * g.C + head * HD + 4 * h;
                }
                if (lse != nullptr && h == 0) lse[(size_t)task * TILE + q] = (m + __builtin_amdgcn_logf(sum)) * LN2;
            }
            // This lane holds the d = 4h + 8 gq + e elements of its query: 8-byte pieces interleaved with the partner lane's
            // (lane ^ 32).  Two v_permlane32_swap per piece pair regroup them so that the h = 0 lane owns d 0..15 and the
            // h = 1 lane d 16..31 -- two 16-byte stores per lane instead of four 8-byte ones (the scattered 8-byte stores
            // alone cost ~4 us of the 26 us launch).
            {
                union { bf16x4 v; unsigned u[2]; } ch[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ch[gq].v[e] = (bf16)(oacc[4 * gq + e] * inv);
                unsigned w[8];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)            // (gq, gq + 2): a' = {h0: own, h1: partner's gq+2}, b' = {h0: partner's gq, h1: own gq+2}
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        auto r = __builtin_amdgcn_permlane32_swap(ch[pr].u[d], ch[pr + 2].u[d], false, false);
                        w[4 * pr + d] = r[0];             // h0: chunk 2pr     | h1: chunk 2pr + 4
                        w[4 * pr + 2 + d] = r[1];         // h0: chunk 2pr + 1 | h1: chunk 2pr + 5
                    }
                if (op) {
                    bf16* o16 = op - 4 * h + 16 * h;      // row + head*HD + 16 h
                    *(uint4*)o16 = uint4{w[0], w[1], w[2], w[3]};
                    *(uint4*)(o16 + 8) = uint4{w[4], w[5], w[6], w[7]};
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------
// fp32 forward (parity path): one 64-thread block per (window, head), thread = query
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void win_attn_fwd_f32_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_exp,
    float* __restrict__ out, float* __restrict__ lse, WinGeom g, float scale) {
    __shared__ float Ks[NTOK][HD + 1];
    __shared__ float Vs[NTOK][HD + 1];
    __shared__ int srcs[NTOK];
    const int task = blockIdx.x, t = threadIdx.x;
    const int head = task % g.nH, win = task / g.nH;
    const int b = win / g.nW, wrem = win - b * g.nW;
    const int wr = wrem / g.nWw, wc = wrem - wr * g.nWw;
    const int C3 = 3 * g.C;
    float q[HD];
    int src = -1;
    if (t < NTOK) {
        src = token_src(g, b, wr, wc, t);
        srcs[t] = src;
        const float* base = src >= 0 ? qkv + (size_t)src * C3 : qkv_bias;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            q[d] = base[head * HD + d] * scale;                 // q = q * scale  (:132)
            Ks[t][d] = base[g.C + head * HD + d];
            Vs[t][d] = base[2 * g.C + head * HD + d];
        }
    }
    __syncthreads();
    if (t >= NTOK) return;
    const float* bp = bias_exp + (size_t)head * TILE * TILE;
    const bool lastr = g.shift > 0 && wr == g.nWh - 1, lastc = g.shift > 0 && wc == g.nWw - 1;
    const int qh = t / 7, qw = t % 7;
    float s[NTOK];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NTOK; ++k) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], Ks[k][d], a);
        a += bp[k * TILE + t];
        int kh = k / 7, kw = k % 7;
        if ((lastr && ((qh >= 4) != (kh >= 4))) || (lastc && ((qw >= 4) != (kw >= 4)))) a += -100.0f;
        s[k] = a;
        m = fmaxf(m, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NTOK; ++k) { s[k] = expf(s[k] - m); sum += s[k]; }
    float invs = 1.0f / sum;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int k = 0; k < NTOK; ++k) {
        float p = s[k] * invs;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = fmaf(p, Vs[k][d], o[d]);
    }
    if (src >= 0) {
        float* op = out + (size_t)src * g.C + head * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = o[d];
    }
    if (lse != nullptr) lse[(size_t)task * TILE + t] = m + logf(sum);
}

// ------------------------------------------------------------------------------------
// bf16 backward: one wave per (window, head), persistent over the windows of one head.
//   phase A (query on the lane):  S^T, P^T = exp2(S^T - lse), dP^T = V dO^T, delta = sum_k P dP,
//       dS^T = P^T (dP^T - delta);  dbias += dS^T (registers);  dQ^T = scale * K^T dS^T  with dS^T
//       taken straight from the accumulator registers as the B operand and K^T read with
//       ds_read_b64_tr_b16;
//   phase B: P and scale*dS go through LDS once as [query][key] bf16 tiles (8-byte writes of 4
//       consecutive keys) and come back through transposed reads as B operands of
//       dV^T = dO^T P  and  dK^T = Q^T dS  (dO^T, Q^T also transposed reads).
// ------------------------------------------------------------------------------------
#define PROW 72   // row stride (bf16 elements) of the 64x64 P / dS tiles: 144 B

// transposed fragment in natural k order: rows row0 + 8*h + j (j=0..7) of column (col0 + lane&31)
template <int STRIDE>
__device__ __forceinline__ bf16x8 lds_tr_frag_lin(const bf16* tile, int row0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, dh = g & 1;
    const bf16* a0 = tile + (row0 + 8 * h + q) * STRIDE + col0 + 16 * dh + 4 * p;
    bf16x4 lo = SWIN_DS_READ_TR16((lds_bf16x4*)a0);
    bf16x4 hi = SWIN_DS_READ_TR16((lds_bf16x4*)(a0 + 4 * STRIDE));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
}

#define SLAB (TILE * TILE + 3 * HD + 32)   // floats per wave in the workspace: dbias tile + pad-token bias gradient

// ------------------------------------------------------------------------------------
// bf16 backward, TWO waves per (window, head) task (round 2).  The first version ran one wave per task with 476 VGPRs and 38.5 KB
// of LDS per wave (one wave per SIMD, nothing hid a task's MFMA -> VALU -> LDS dependency chain: 94 us at stage 1 against 73 us
// here; removed in round 3, git history has it).  Here a task belongs to a
// PAIR of waves sharing one BwdLds2:
//   staging : each wave loads half of the task's q|k|v|dO rows (8 tokens per round over the pair, 7 rounds)
//   phase A : wave w owns query tile w (32 queries x 64 keys): S^T, dP^T, P, delta, dS, dQ -- independent per query tile, so
//             the bias / dbias register tiles halve (32 + 32 instead of 64 + 64)
//   phase B : wave w owns key tile w: dV^T = dO^T P and dK^T = Q^T dS over ALL 64 queries, reading the P / dS rows both
//             waves wrote
//   output  : wave w writes dq | dk | dv of tokens 32 w + lane
// with block-level barriers between the phases (every wave runs the same number of iterations; a pair without a task only
// meets the barriers).  <= 256 VGPRs per wave and the same 39 KB per task: 8 waves per CU instead of 4.
// ------------------------------------------------------------------------------------
struct BwdLds2 {
    bf16 t[4][TILE][LROW];     // Q, K, V, dO head slices
    bf16 p[TILE][PROW];        // P        [query][key]
    bf16 ds[TILE][PROW];       // scale*dS [query][key]
    float padacc[2][3 * HD + 32];   // per wave: gradient reaching qkv.bias through padded tokens (q|k|v x 32)
};

struct Bwd2Lane {
    int lane, c, h, w, tokr, which, part;
    unsigned ld_off[7];        // interior path: element offset of staged piece i (qkv stride 3C or dout stride C)
    unsigned st_off;           // interior path: element offset (x 3C) of this lane's token
};

template <bool INTERIOR>
__device__ __forceinline__ void bwd2_issue_loads(bf16x8 (&stg)[7], const Bwd2Lane& L, const WinGeom& g, const bf16* __restrict__ qkv,
                                                 const bf16* __restrict__ dout, const float* __restrict__ qkv_bias, int head,
                                                 int b, int wr, int wc) {
    const int C3 = 3 * g.C;
    const bool is_do = L.which == 3;
    const int ch = is_do ? head * HD + L.part * 8 : L.which * g.C + head * HD + L.part * 8;
    if (INTERIOR) {
        const size_t wbase = (size_t)(b * g.H + wr * 7 + g.shift) * g.W + wc * 7 + g.shift;
        const bf16* pb = is_do ? dout + wbase * g.C + ch : qkv + wbase * C3 + ch;
#pragma unroll
        for (int i = 0; i < 7; ++i) stg[i] = *(const bf16x8*)(pb + L.ld_off[i]);    // unpredicated: see fwd_issue_loads
    } else {
        bf16x8 padv;
        if (is_do) {
#pragma unroll
            for (int e = 0; e < 8; ++e) padv[e] = (bf16)0.f;
        } else padv = bias_to_bf16x8(qkv_bias + ch);
        const bf16* pb = is_do ? dout + ch : qkv + ch;
        const int stride = is_do ? g.C : C3;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            int t = 8 * i + L.tokr; if (t >= NTOK) t = NTOK - 1;
            const int src = token_src(g, b, wr, wc, t);
            const bf16x8 v = *(const bf16x8*)(pb + (size_t)(src >= 0 ? src : 0) * stride);
            stg[i] = src >= 0 ? v : padv;
        }
    }
}

__global__ __launch_bounds__(512) void win_attn_bwd2_bf16_kernel(
    const bf16* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_exp,
    const float* __restrict__ lse, const bf16* __restrict__ dout, bf16* __restrict__ dqkv,
    float* __restrict__ dbias_ws, float* __restrict__ dbias_pad, WinGeom g, float scale, int n_tasks, int iters, int ppb) {
    // ppb = wave pairs (tasks in flight) per block = blockDim.x / 128: 4, or 3 to keep the three heads of a window in one block
    // when nH == 3 (see win_attn_fwd_bf16_kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = wave >> 1, w = wave & 1;
    BwdLds2* Lm = reinterpret_cast<BwdLds2*>(smem_raw) + pair;
    Bwd2Lane L;
    L.lane = threadIdx.x & 63; L.c = L.lane & 31; L.h = L.lane >> 5; L.w = w;
    const int lane = L.lane, c = L.c, h = L.h;
    bf16(*Qs)[LROW] = Lm->t[0];
    bf16(*Ks)[LROW] = Lm->t[1];
    bf16(*Vs)[LROW] = Lm->t[2];
    bf16(*Ds)[LROW] = Lm->t[3];
    {
        uint4 z = {0, 0, 0, 0};
        uint4* p = (uint4*)Lm;
        for (int i = lane + WAVE * w; i < (int)(sizeof(BwdLds2) / 16); i += 2 * WAVE) p[i] = z;
    }
    const int n_pairs = gridDim.x * ppb;
    int task = blockIdx.x * ppb + pair;
    float* const slab = dbias_ws + (size_t)task * SLAB;
    const int head = task % g.nH;                      // n_pairs % nH == 0: constant over the pair's tasks
    const int C3 = 3 * g.C;

    // this wave's query tile (w): bias and the running bias gradient, [key tile][reg]
    float biasr[2][16], dbacc[2][16];
    {
        const float* bp = bias_exp + (size_t)head * TILE * TILE;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int key = 32 * kt + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                biasr[kt][reg] = bp[key * TILE + 32 * w + c] * LOG2E;
                dbacc[kt][reg] = 0.f;
            }
    }
    uint64_t mrow = 0, mcol = 0;
    if (g.shift > 0) lane_mask_bits(lane, mrow, mcol);
    const float sl2 = scale * LOG2E;

    // staging: 8 tokens per round over the pair (wave w: tokens 8 i + 4 w + lane/16), 16 pieces per token
    L.tokr = (lane >> 4) + 4 * w;
    L.which = (lane & 15) >> 2; L.part = lane & 3;
    {
        const int stride = L.which == 3 ? g.C : C3;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            int t = 8 * i + L.tokr; if (t >= NTOK) t = NTOK - 1;
            L.ld_off[i] = (unsigned)(((t / 7) * g.W + (t % 7)) * stride);
        }
        int q = 32 * w + c; if (q >= NTOK) q = NTOK - 1;
        L.st_off = (unsigned)(((q / 7) * g.W + (q % 7)) * C3);
    }
    const int which = L.which, part = L.part, tokr = L.tokr;

    WinPos cur = win_decode(g, task / g.nH);
    WinStride stride;
    {
        const WinPos d = win_decode(g, n_pairs / g.nH);
        stride.db = d.b; stride.dwr = d.wr; stride.dwc = d.wc;
    }
    WinPos nxt = cur;
    win_advance(nxt, stride, g);

    bf16x8 stg[7];
    float lse_n = 0.f;
    const int qtok = 32 * w + c;                      // this lane's query (phase A) and token (output)
    if (task < n_tasks) {
        lse_n = lse[(size_t)task * TILE + (qtok < NTOK ? qtok : 0)];
        if (win_interior(g, cur)) bwd2_issue_loads<true>(stg, L, g, qkv, dout, qkv_bias, head, cur.b, cur.wr, cur.wc);
        else bwd2_issue_loads<false>(stg, L, g, qkv, dout, qkv_bias, head, cur.b, cur.wr, cur.wc);
    }
    __syncthreads();                                  // LDS zeroed by both waves of every pair

    for (int it = 0; it < iters; ++it, task += n_pairs, cur = nxt, win_advance(nxt, stride, g)) {
        const bool act = task < n_tasks;              // wave-uniform (pair-uniform)
        const int b = cur.b, wr = cur.wr, wc = cur.wc;
        const bool interior = win_interior(g, cur);
        if (act) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                int t = 8 * i + tokr;
                if (t < NTOK) *(bf16x8*)&Lm->t[which][t][part * 8] = stg[i];
            }
        }
        __syncthreads();                              // (1) the task's tiles are staged
        const float lse_q = lse_n;
        if (task + n_pairs < n_tasks) {               // prefetch the next task (its lse row with it: see the one-wave kernel)
            lse_n = lse[(size_t)(task + n_pairs) * TILE + (qtok < NTOK ? qtok : 0)];
            if (win_interior(g, nxt)) bwd2_issue_loads<true>(stg, L, g, qkv, dout, qkv_bias, head, nxt.b, nxt.wr, nxt.wc);
            else bwd2_issue_loads<false>(stg, L, g, qkv, dout, qkv_bias, head, nxt.b, nxt.wr, nxt.wc);
        }
        f32x16 dq = {0};
        if (act) {
            // ---------------- phase A: query tile w ------------------------------------------------
            const bool edge = g.shift > 0 && (wr == g.nWh - 1 || wc == g.nWw - 1);
            uint64_t mbits = 0;
            if (edge) {
                if (wr == g.nWh - 1) mbits |= mrow;
                if (wc == g.nWw - 1) mbits |= mcol;
            }
            bf16x8 qf[2], df[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                qf[s] = *(const bf16x8*)&Qs[32 * w + c][16 * s + 8 * h];
                df[s] = *(const bf16x8*)&Ds[32 * w + c][16 * s + 8 * h];
            }
            f32x16 pacc[2], dpacc[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 a = {0}, d = {0};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 kf = *(const bf16x8*)&Ks[32 * kt + c][16 * s + 8 * h];
                    const bf16x8 vf = *(const bf16x8*)&Vs[32 * kt + c][16 * s + 8 * h];
                    a = SWIN_MFMA_32x32x16(kf, qf[s], a, 0, 0, 0);   // S^T
                    d = SWIN_MFMA_32x32x16(vf, df[s], d, 0, 0, 0);   // dP^T
                }
                pacc[kt] = a;
                dpacc[kt] = d;
            }
            const bool qv = qtok < NTOK;
            const float l2 = lse_q * LOG2E;
            float d4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    if (kt == 1 && reg >= 9) { pacc[kt][reg] = 0.f; continue; }   // keys 49..63 (padding): P = 0
                    float v = fmaf(pacc[kt][reg], sl2, biasr[kt][reg]);
                    if (edge && ((mbits >> ((kt * 2 + w) * 16 + reg)) & 1)) v += -100.0f * LOG2E;
                    float p = qv ? __builtin_amdgcn_exp2f(v - l2) : 0.f;
                    pacc[kt][reg] = p;
                    d4[reg & 3] = fmaf(p, dpacc[kt][reg], d4[reg & 3]);
                }
            const float delta = half_swap_sum((d4[0] + d4[1]) + (d4[2] + d4[3]));
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    bf16x4 p4, s4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int reg = 4 * gq + e;
                        if (kt == 1 && reg >= 9) { dpacc[kt][reg] = 0.f; p4[e] = (bf16)0.f; s4[e] = (bf16)0.f; continue; }
                        float ds = pacc[kt][reg] * (dpacc[kt][reg] - delta);
                        dbacc[kt][reg] += ds;
                        ds *= scale;
                        dpacc[kt][reg] = ds;
                        p4[e] = (bf16)pacc[kt][reg];
                        s4[e] = (bf16)ds;
                    }
                    *(bf16x4*)&Lm->p[qtok][32 * kt + 8 * gq + 4 * h] = p4;
                    *(bf16x4*)&Lm->ds[qtok][32 * kt + 8 * gq + 4 * h] = s4;
                }
            // dQ^T = K^T (scale dS^T)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 kc = lds_tr_frag_perm(Ks, 32 * kt + 16 * s, lane);
                    bf16x8 sf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) sf[j] = (bf16)dpacc[kt][8 * s + j];
                    dq = SWIN_MFMA_32x32x16(kc, sf, dq, 0, 0, 0);
                }
        }
        __syncthreads();                              // (2) P / dS rows of both query tiles are in LDS

        if (act) {
            // ---------------- phase B: key tile w: dV^T = dO^T P ,  dK^T = Q^T (scale dS) ----------
            f32x16 dv = {0}, dk = {0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 doc = lds_tr_frag_lin<LROW>(&Ds[0][0], 16 * ks, 0, lane);
                bf16x8 qc = lds_tr_frag_lin<LROW>(&Qs[0][0], 16 * ks, 0, lane);
                bf16x8 pb = lds_tr_frag_lin<PROW>(&Lm->p[0][0], 16 * ks, 32 * w, lane);
                bf16x8 sb = lds_tr_frag_lin<PROW>(&Lm->ds[0][0], 16 * ks, 32 * w, lane);
                dv = SWIN_MFMA_32x32x16(doc, pb, dv, 0, 0, 0);
                dk = SWIN_MFMA_32x32x16(qc, sb, dk, 0, 0, 0);
            }
            // ---------------- dq | dk | dv of token 32 w + c ----------------------------------------
            bool pad = false;
            if (qtok < NTOK) {
                bf16* op = nullptr;
                if (interior) {
                    op = dqkv + ((size_t)(b * g.H + wr * 7 + g.shift) * g.W + wc * 7 + g.shift) * C3 + L.st_off + head * HD + 4 * h;
                } else {
                    int src = token_src(g, b, wr, wc, qtok);
                    if (src >= 0) op = dqkv + (size_t)src * C3 + head * HD + 4 * h;
                    else pad = true;
                }
                if (op) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        bf16x4 a, bb, cc;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = (bf16)dq[4 * gq + e];
                            bb[e] = (bf16)dk[4 * gq + e];
                            cc[e] = (bf16)dv[4 * gq + e];
                        }
                        *(bf16x4*)(op + 8 * gq) = a;
                        *(bf16x4*)(op + g.C + 8 * gq) = bb;
                        *(bf16x4*)(op + 2 * g.C + 8 * gq) = cc;
                    }
                }
            }
            if (__ballot(pad)) {                       // wave-uniform; see the one-wave kernel for the history of this form
                float* bp = Lm->padacc[w] + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float a = pad ? dq[r] : 0.f, bq = pad ? dk[r] : 0.f, cv = pad ? dv[r] : 0.f;
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) {
                        a += __shfl_xor(a, o); bq += __shfl_xor(bq, o); cv += __shfl_xor(cv, o);
                    }
                    if (c == 0) {
                        const int d = (r & 3) + 8 * (r >> 2);
                        bp[d] += a; bp[HD + d] += bq; bp[2 * HD + d] += cv;
                    }
                }
            }
        }
        __syncthreads();                              // (3) all reads of this task's tiles are done
    }
    // this pair's partial relative-position-bias gradient -> its slab (wave w: query columns 32 w ..)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            int key = 32 * kt + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            slab[key * TILE + 32 * w + c] = dbacc[kt][reg];
        }
    if (w == 0)
        for (int i = lane; i < 3 * HD + 32; i += WAVE)
            slab[TILE * TILE + i] = i < 3 * HD ? Lm->padacc[0][i] + Lm->padacc[1][i] : 0.f;
}

// dbias_exp[head][key][query] += sum over the slabs of that head (slab s belongs to head s % nH);
// dbias_pad[which*C + head*32 + d] += the slabs' pad-token part.  grid.y splits the slab range.
__global__ __launch_bounds__(256) void dbias_slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dbias_exp,
                                                                float* __restrict__ dbias_pad, int n_slabs, int nH, int C) {
    const int per_head = TILE * TILE + 3 * HD;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nH * per_head) return;
    int head = i / per_head, e = i - head * per_head;
    const int per = (n_slabs / nH + gridDim.y - 1) / gridDim.y;        // slabs of this head handled per y-block
    const int k0 = blockIdx.y * per, k1 = min(k0 + per, n_slabs / nH);
    float acc = 0.f;
    for (int k = k0; k < k1; ++k) acc += ws[(size_t)(head + k * nH) * SLAB + e];
    if (k1 <= k0) return;
    if (e < TILE * TILE) atomicAdd(dbias_exp + head * TILE * TILE + e, acc);
    else if (dbias_pad) {
        int w = (e - TILE * TILE) / HD, d = (e - TILE * TILE) % HD;
        atomicAdd(dbias_pad + w * C + head * HD + d, acc);
    }
}

// ------------------------------------------------------------------------------------
// fp32 backward (parity path): 64-thread block per (window, head)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void win_attn_bwd_f32_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_exp,
    const float* __restrict__ lse, const float* __restrict__ dout, float* __restrict__ dqkv,
    float* __restrict__ dbias_exp, float* __restrict__ dbias_pad, WinGeom g, float scale) {
    __shared__ float Qs[NTOK][HD + 1], Ks[NTOK][HD + 1], Vs[NTOK][HD + 1], Ds[NTOK][HD + 1];
    __shared__ float Ps[NTOK][NTOK + 1], Ss[NTOK][NTOK + 1];
    const int task = blockIdx.x, t = threadIdx.x;
    const int head = task % g.nH, win = task / g.nH;
    const int b = win / g.nW, wrem = win - b * g.nW;
    const int wr = wrem / g.nWw, wc = wrem - wr * g.nWw;
    const int C3 = 3 * g.C;
    int src = -1;
    if (t < NTOK) {
        src = token_src(g, b, wr, wc, t);
        const float* base = src >= 0 ? qkv + (size_t)src * C3 : qkv_bias;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            Qs[t][d] = base[head * HD + d] * scale;
            Ks[t][d] = base[g.C + head * HD + d];
            Vs[t][d] = base[2 * g.C + head * HD + d];
            Ds[t][d] = src >= 0 ? dout[(size_t)src * g.C + head * HD + d] : 0.f;
        }
    }
    __syncthreads();
    const float* bp = bias_exp + (size_t)head * TILE * TILE;
    float* db = dbias_exp + (size_t)head * TILE * TILE;
    const bool lastr = g.shift > 0 && wr == g.nWh - 1, lastc = g.shift > 0 && wc == g.nWw - 1;
    float dq[HD];
    if (t < NTOK) {
        const int qh = t / 7, qw = t % 7;
        const float l = lse[(size_t)task * TILE + t];
        float p[NTOK], dp[NTOK];
        float delta = 0.f;
#pragma unroll
        for (int k = 0; k < NTOK; ++k) {
            float a = 0.f, e = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) { a = fmaf(Qs[t][d], Ks[k][d], a); e = fmaf(Ds[t][d], Vs[k][d], e); }
            a += bp[k * TILE + t];
            int kh = k / 7, kw = k % 7;
            if ((lastr && ((qh >= 4) != (kh >= 4))) || (lastc && ((qw >= 4) != (kw >= 4)))) a += -100.0f;
            p[k] = expf(a - l);
            dp[k] = e;
            delta += p[k] * e;
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = 0.f;
#pragma unroll
        for (int k = 0; k < NTOK; ++k) {
            float ds = p[k] * (dp[k] - delta);
            Ps[t][k] = p[k];
            Ss[t][k] = ds;
            atomicAdd(db + k * TILE + t, ds);
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, Ks[k][d], dq[d]);
        }
    }
    __syncthreads();
    if (t >= NTOK) return;
    float dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int q = 0; q < NTOK; ++q) {
        float pq = Ps[q][t], sq = Ss[q][t];
#pragma unroll
        for (int d = 0; d < HD; ++d) { dv[d] = fmaf(pq, Ds[q][d], dv[d]); dk[d] = fmaf(sq, Qs[q][d], dk[d]); }   // Qs holds q*scale
    }
    if (src >= 0) {
        float* op = dqkv + (size_t)src * C3 + head * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { op[d] = dq[d] * scale; op[g.C + d] = dk[d]; op[2 * g.C + d] = dv[d]; }
    } else {
        float* op = dbias_pad + head * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            atomicAdd(op + d, dq[d] * scale);
            atomicAdd(op + g.C + d, dk[d]);
            atomicAdd(op + 2 * g.C + d, dv[d]);
        }
    }
}

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
static int check_attn_args(const void* qkv, const float* qkv_bias, const float* bias_exp, const void* out,
                           int B, int H, int W, int C, int nH, int shift, int dtype) {
    if (!qkv || !bias_exp || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || nH <= 0) return SWIN_ERR_BAD_ARG;
    if (C != nH * HD || (shift != 0 && shift != 3) || (dtype != SWIN_F32 && dtype != SWIN_BF16))
        return SWIN_ERR_UNSUPPORTED;
    if ((H % 7 != 0 || W % 7 != 0) && !qkv_bias) return SWIN_ERR_BAD_ARG;
    return SWIN_OK;
}

// Grid sizing of the persistent kernels.  4 waves per block; the total wave count must be a multiple of nH so that a
// wave keeps its head (and its bias tile in registers): blocks % nH == 0.  Block counts were swept on one MI355X at the
// four Swin-T stage geometries of the 2x800x1280 workload (tools/microbench.py attn with SWIN_ATTN_{FWD,BWD}_BLOCKS):
//   forward (2 blocks resident per CU = 512): 448-480 blocks are best (stage 1: 28.4 us vs 30.8 at 513 and 33.8 at
//     256); rounding the cap UP to a multiple of nH (513 / 516 blocks) leaves blocks that start only when a resident
//     block has finished all its rounds -- a second, almost empty phase.
//   backward (1 block resident per CU = 256): many rounds per wave (stage 1, 8 rounds) -> stay below the resident
//     count (240: 96 us; 258: 106; 288: 123); few rounds (stages 2-4) -> MORE blocks than CUs, so the dispatcher
//     back-fills finished CUs and evens out the round quantisation (stage 2: 68 us at 320-384 blocks vs 91 at 258;
//     stage 3: 55-58 vs 63).
static int round_blocks(int want, int n_tasks, int nH) {
    const int all = ((n_tasks + 3) / 4 + nH - 1) / nH * nH;      // one task per wave
    int b = want >= nH ? want / nH * nH : nH;
    return b < all ? b : all;
}

// forward: working waves per block.  nH == 3 (stage 1 of Swin-T/S): 3, so that the three heads of a window share a block
// (see the kernel); SWIN_ATTN_FWD_WPB=4 restores the four-wave mapping for comparison.
static int attn_fwd_wpb(int nH) {
    static const int forced = swin_dev_int("SWIN_ATTN_FWD_WPB", 0);
    if (forced == 3 || forced == 4) return (forced == 3 && nH % 3 != 0) ? 4 : forced;
    return nH == 3 ? 3 : 4;
}

static int attn_grid_blocks(int n_tasks, int nH, int wpb) {
    static const int forced = swin_dev_int("SWIN_ATTN_FWD_BLOCKS", 0);   // development sweep
    if (wpb == 3) {                                  // nH % 3 == 0: any block count keeps (waves % nH == 0)
        const int all = (n_tasks + 2) / 3, want = forced > 0 ? forced : 512;        // swept: 480: 26.4 us, 512: 25.2, 640: 27.3 (stage 1)
        return want < all ? want : all;
    }
    return round_blocks(forced > 0 ? forced : 480, n_tasks, nH);
}

extern "C" int swin_window_attn_fwd(const void* qkv, const float* qkv_bias, const float* bias_exp, void* out,
                                    float* lse, int B, int H, int W, int C, int nH, int shift, float scale,
                                    int dtype, void* stream) {
    int st = check_attn_args(qkv, qkv_bias, bias_exp, out, B, H, W, C, nH, shift, dtype);
    if (st != SWIN_OK) return st;
    WinGeom g = make_geom(B, H, W, C, nH, shift);
    int n_tasks = B * g.nW * nH;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) {
        const int wpb = attn_fwd_wpb(nH);
        int blocks = attn_grid_blocks(n_tasks, nH, wpb);
        win_attn_fwd_bf16_kernel<<<blocks, 256, 0, s>>>((const bf16*)qkv, qkv_bias, bias_exp, (bf16*)out, lse, g,
                                                        scale, n_tasks, wpb);
    } else {
        win_attn_fwd_f32_kernel<<<n_tasks, 64, 0, s>>>((const float*)qkv, qkv_bias, bias_exp, (float*)out, lse, g,
                                                       scale);
    }
    return swin_launch_status();
}

extern "C" int swin_rel_bias_expand(const float* table, float* bias_exp, int nH, void* stream) {
    if (!table || !bias_exp || nH <= 0) return SWIN_ERR_BAD_ARG;
    int n = nH * TILE * TILE;
    rel_bias_expand_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(table, bias_exp, nH);
    return swin_launch_status();
}

// every block's table in one launch (the expanded tables are a function of the parameters alone: the caller rebuilds them once per
// optimizer step, not once per block forward)
struct ExpTab { int n; const float* t[48]; float* o[48]; int nH[48]; };
__global__ void rel_bias_expand_multi_kernel(ExpTab tab) {
    const int j = blockIdx.y;
    const int nH = tab.nH[j];
    const float* __restrict__ table = tab.t[j];
    float* __restrict__ out = tab.o[j];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nH * TILE * TILE; i += gridDim.x * blockDim.x) {
        int q = i & 63, k = (i >> 6) & 63, h = i >> 12;
        float v = NEG_BIG;
        if (k < NTOK) {
            v = 0.f;
            if (q < NTOK) {
                int qh = q / 7, qw = q % 7, kh = k / 7, kw = k % 7;
                v = table[((qh - kh + 6) * 13 + (qw - kw + 6)) * nH + h];   // swin_transformer.py:105-110
            }
        }
        out[i] = v;
    }
}

extern "C" int swin_rel_bias_expand_multi(const float* const* tables, float* const* outs, const int* nH, int n, void* stream) {
    if (!tables || !outs || !nH || n <= 0) return SWIN_ERR_BAD_ARG;
    for (int i0 = 0; i0 < n; i0 += 48) {
        ExpTab tab;
        tab.n = n - i0 < 48 ? n - i0 : 48;
        int mx = 0;
        for (int i = 0; i < tab.n; ++i) {
            if (!tables[i0 + i] || !outs[i0 + i] || nH[i0 + i] <= 0) return SWIN_ERR_BAD_ARG;
            tab.t[i] = tables[i0 + i]; tab.o[i] = outs[i0 + i]; tab.nH[i] = nH[i0 + i];
            mx = nH[i0 + i] > mx ? nH[i0 + i] : mx;
        }
        const int bx = mx * 16 < 64 ? mx * 16 : 64;
        rel_bias_expand_multi_kernel<<<dim3(bx, tab.n), 256, 0, (hipStream_t)stream>>>(tab);
    }
    return swin_launch_status();
}

extern "C" int swin_rel_bias_reduce(const float* dbias_exp, float* dtable, int nH, void* stream) {
    if (!dbias_exp || !dtable || nH <= 0) return SWIN_ERR_BAD_ARG;
    auto launch = [=](void* st) {
        rel_bias_reduce_kernel<<<nH, 256, 0, (hipStream_t)st>>>(dbias_exp, dtable, nH);
        return swin_launch_status();
    };
    if (swin_aux_push(launch)) return SWIN_OK;         // follows the slab reduce onto the auxiliary stream when one is set
    void* aux = swin_aux_stream();
    if (aux && aux != stream) {
        if (swin_fork_stream(stream, aux) != SWIN_OK) return SWIN_ERR_LAUNCH;
        stream = aux;
    }
    return launch(stream);
}

// backward (two-wave kernel): wave pairs per block.  nH == 3: 3, so that a window's heads share a block as in the forward kernel
// (stage 1: 73.4 -> 71.5 us with a quarter fewer waves); SWIN_ATTN_BWD_PPB=4 restores four pairs.
static int attn_bwd_ppb(int nH) {
    static const int forced = swin_dev_int("SWIN_ATTN_BWD_PPB", 0);
    if (forced == 4 || nH % 3 != 0) return 4;
    return (forced == 3 || nH == 3) ? 3 : 4;
}

static int attn_bwd_blocks(int n_tasks, int nH) {
    static const int forced = swin_dev_int("SWIN_ATTN_BWD_BLOCKS", 0);   // development sweep
    if (attn_bwd_ppb(nH) == 3) {                     // any block count keeps (pairs % nH == 0); swept at stage 1: 240: 74.5 us, 256: 71.5, 320: 89
        const int all = (n_tasks + 2) / 3, want = forced > 0 ? forced : 256;
        return want < all ? want : all;
    }
    if (forced > 0) return round_blocks(forced, n_tasks, nH);
    const bool many_rounds = n_tasks >= 6 * 960;                 // >= 6 rounds at 240 blocks
    // two-waves-per-task kernel, swept over 192..512 blocks (round 2): stage 1 (8004 tasks) 72.9 us at 240 (256: 73.2, 320+: 84-88);
    // stage 2 (4140) 56.5 at 256 (240: 58.4, 384: 57.9); stages 3 / 4 (2304 / 1152 tasks: whole rounds at 768 pairs) 42.0 / 39.5 at
    // 192 (256: 49.1 / 40.3, 384: 45.2 / 40.9).  One block per CU (158 KB of LDS): more blocks than CUs only add a second wave of blocks.
    return round_blocks(many_rounds ? 240 : (n_tasks <= 2400 ? 192 : 256), n_tasks, nH);
}

extern "C" int64_t swin_window_attn_bwd_workspace_bytes(int B, int H, int W, int nH, int dtype) {
    if (dtype != SWIN_BF16 || B <= 0 || H <= 0 || W <= 0 || nH <= 0) return 0;
    WinGeom g = make_geom(B, H, W, nH * HD, nH, 0);
    return (int64_t)attn_bwd_blocks(B * g.nW * nH, nH) * 4 * SLAB * sizeof(float);
}

static_assert(TILE == 64 && NTOK == 49 && HD == 32, "csrc/tail_reduce.hip restates these");

// the bf16 backward launch alone: dqkv written, the per-wave bias-gradient slabs left in `workspace`
int swin_window_attn_bwd_slabs(const void* qkv, const float* qkv_bias, const float* bias_exp, const float* lse, const void* dout,
                               void* dqkv, float* dqkv_bias_pad, void* workspace, int B, int H, int W, int C, int nH, int shift,
                               float scale, void* stream, int* n_slabs, int* slab_stride) {
    int st = check_attn_args(qkv, qkv_bias, bias_exp, dqkv, B, H, W, C, nH, shift, SWIN_BF16);
    if (st != SWIN_OK) return st;
    if (!lse || !dout || !workspace) return SWIN_ERR_BAD_ARG;
    if ((H % 7 != 0 || W % 7 != 0) && !dqkv_bias_pad) return SWIN_ERR_BAD_ARG;
    WinGeom g = make_geom(B, H, W, C, nH, shift);
    const int n_tasks = B * g.nW * nH;
    const int blocks = attn_bwd_blocks(n_tasks, nH);
    static bool attr_set[16] = {};                       // per device: the attribute belongs to the device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)win_attn_bwd2_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(4 * sizeof(BwdLds2))) != hipSuccess) return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const int ppb = attn_bwd_ppb(nH);
    const int iters = (n_tasks + blocks * ppb - 1) / (blocks * ppb);
    win_attn_bwd2_bf16_kernel<<<blocks, 128 * ppb, ppb * sizeof(BwdLds2), (hipStream_t)stream>>>(
        (const bf16*)qkv, qkv_bias, bias_exp, lse, (const bf16*)dout, (bf16*)dqkv, (float*)workspace, dqkv_bias_pad, g, scale, n_tasks,
        iters, ppb);
    if (n_slabs) *n_slabs = blocks * ppb;
    if (slab_stride) *slab_stride = SLAB;
    return swin_launch_status();
}

extern "C" int swin_window_attn_bwd(const void* qkv, const float* qkv_bias, const float* bias_exp, const float* lse,
                                    const void* dout, void* dqkv, float* dbias_exp, float* dqkv_bias_pad,
                                    void* workspace, int B, int H, int W, int C, int nH, int shift, float scale,
                                    int dtype, void* stream) {
    int st = check_attn_args(qkv, qkv_bias, bias_exp, dqkv, B, H, W, C, nH, shift, dtype);
    if (st != SWIN_OK) return st;
    if (!lse || !dout || !dbias_exp) return SWIN_ERR_BAD_ARG;
    if ((H % 7 != 0 || W % 7 != 0) && !dqkv_bias_pad) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) {
        int n_slabs = 0;
        st = swin_window_attn_bwd_slabs(qkv, qkv_bias, bias_exp, lse, dout, dqkv, dqkv_bias_pad, workspace, B, H, W, C, nH, shift, scale,
                                        stream, &n_slabs, nullptr);
        if (st != SWIN_OK) return st;
        int n = nH * (TILE * TILE + 3 * HD);
        dim3 rgrid((n + 255) / 256, 16);
        // bias-gradient reduce: off the data-gradient chain (see csrc/abi.hip)
        auto launch = [=](void* st) {
            dbias_slab_reduce_kernel<<<rgrid, 256, 0, (hipStream_t)st>>>((const float*)workspace, dbias_exp, dqkv_bias_pad, n_slabs, nH, C);
            return swin_launch_status();
        };
        if (!swin_aux_push(launch)) {
            void* rs = (void*)s;
            void* aux = swin_aux_stream();
            if (aux && aux != (void*)s) {
                if (swin_fork_stream((void*)s, aux) != SWIN_OK) return SWIN_ERR_LAUNCH;
                rs = aux;
            }
            return launch(rs);
        }
    } else {
        WinGeom g = make_geom(B, H, W, C, nH, shift);
        win_attn_bwd_f32_kernel<<<B * g.nW * nH, 64, 0, s>>>((const float*)qkv, qkv_bias, bias_exp, lse, (const float*)dout,
                                                       (float*)dqkv, dbias_exp, dqkv_bias_pad, g, scale);
    }
    return swin_launch_status();
}
