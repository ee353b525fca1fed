// Weight-gradient GEMM, second generation (round 3): dW[n1][n2] += sum_t dY[t][n1] * X[t][n2], bf16 in, fp32 out, with the
// operand tiles brought into LDS by LDS-DMA (global_load_lds_dwordx4) through a ring of stage buffers.
//
// Why a second kernel: rocprofv3 counters of the first one (csrc/wgrad_gemm.hip; profiles/r03_pmc_wgrad_summary.txt) show the
// MFMA pipe busy 10 % (Linear, stage 3) / 22 % (3x3 conv) of the kernel, 10 VALU + 2.5 LDS + 2.5 SALU instructions issued per
// MFMA (register staging: load -> mask -> ds_write with per-piece address arithmetic, the im2col row stepping, the bias
// column sums) and ONE stage of loads in flight per block.  Here a stage costs each wave 8 DMA instructions whose source
// addresses advance by one 64-bit add, nothing passes through VGPRs, the bias gradient is one more MFMA against a constant
// all-ones operand, and NBUF - 1 stages are in flight behind counted s_waitcnt vmcnt / raw s_barrier (guide section 5,
// "Pipelining across barriers").
//
// Tile 128 (n1) x 128 (n2), 4 waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x16_bf16 accumulators), stage = 64 rows of t.
// Both operands are read by COLUMNS of row-major [t][n] tiles (the contraction index t is the tile's row), i.e. with
// ds_read_b64_tr_b16.  LDS image of a stage: 64 rows x 256 B per operand, written linearly by the DMA (one wave instruction =
// 4 rows x 256 B); the 16-byte chunks of a row are XOR-permuted on the SOURCE address, chunk position = chunk ^ f(row),
// f(row) = ((row & 3) << 2) | ((row >> 2) & 3)  (guide T10, image (b): conflict-free transposed reads of the 32x32x16 operand).
// Rows beyond the split's t range and columns beyond N read a 16-byte zero line.
// X is a plain (T, N2) matrix (nn.Linear: swin_transformer.py:33,36,129,151,296) or the implicit im2col of a channels-last
// activation for a 3x3 / pad 1 convolution (fpn.py:195-197, rpn_head.py:43, fcn_mask_head.py:119-121) with Cin % 128 == 0, so
// that a 128-column tile lies inside ONE filter tap: the tap's pixel offset is block-uniform and the border test is per row.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "common.h"

namespace {

constexpr int ST = 64;               // rows of t per stage
constexpr int TN = 128;              // tile width (n1 and n2)
constexpr int ROWB = TN * 2;         // bytes per LDS row
constexpr int OPB = ST * ROWB;       // bytes per operand tile of a stage (16 KB)
constexpr int STAGEB = 2 * OPB;      // dY tile | X tile
constexpr int DPT = 8;               // DMA instructions per wave and stage (4 per operand)

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_d;
typedef __attribute__((address_space(3))) void* lptr_d;

// zero-initialised: the source of every masked 16-byte piece.  4096 lines, each lane reading its own: with one shared line a shape whose
// N is not a multiple of the tile (96, 288: a quarter to three quarters of the DMA lanes masked) sent all those lanes of all CUs to
// the same L2 channel (measured: 67 us against 26 us for the register-staged kernel at 96 x 96).
__device__ uint4 g_zero_line[4096];

__device__ __forceinline__ void glds16(uint64_t gsrc, unsigned lds_addr) {      // see csrc/conv_gemm.hip: asm, so that hipcc does not drain it
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// X operand sources.  A lane owns, for its 4 DMA instructions per stage, rows r_i = 16 wave + 4 i + (lane >> 4) of the stage and
// the chunk (lane & 15) ^ f(r_i) of the tile's 16 chunks.
struct PlainSrc {
    const bf16* x; int N2;
    struct St { uint64_t ptr[4]; bool colok[4]; };
    __device__ __forceinline__ void init(St& s, int64_t t0, int n2_0, const int (&row)[4], const int (&chunk)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n2_0 + chunk[i] * 8;
            s.colok[i] = n < N2;
            s.ptr[i] = (uint64_t)reinterpret_cast<uintptr_t>(x + (t0 + row[i]) * (int64_t)N2 + (s.colok[i] ? n : 0));
        }
    }
    __device__ __forceinline__ void advance(St& s, int dt) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) s.ptr[i] += (uint64_t)dt * (uint64_t)N2 * 2u;
    }
    __device__ __forceinline__ bool ok(const St& s, int i) const { return s.colok[i]; }
};

struct ConvSrc {
    const bf16* x; int H, W, Cin;
    int qd, rd;                         // a stage advances every row by dt pixels: dt / W and dt % W (host-computed)
    struct St { uint64_t ptr[4]; int y[4], xx[4]; int dy, dx; bool tapok; };
    __device__ __forceinline__ void init(St& s, int64_t t0, int n2_0, const int (&row)[4], const int (&chunk)[4]) const {
        const int tap = n2_0 / Cin, c0 = n2_0 - tap * Cin;           // block-uniform: Cin % 128 == 0
        s.tapok = tap < 9;
        s.dy = tap / 3 - 1; s.dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t t = t0 + row[i];
            s.xx[i] = (int)(t % W); s.y[i] = (int)((t / W) % H);
            s.ptr[i] = (uint64_t)reinterpret_cast<uintptr_t>(x + (t + (int64_t)s.dy * W + s.dx) * Cin + c0 + chunk[i] * 8);
        }
    }
    __device__ __forceinline__ void advance(St& s, int dt) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s.ptr[i] += (uint64_t)dt * (uint64_t)Cin * 2u;
            s.xx[i] += rd; s.y[i] += qd;
            if (s.xx[i] >= W) { s.xx[i] -= W; s.y[i] += 1; }
            if (s.y[i] >= H) { s.y[i] -= H; if (s.y[i] >= H) s.y[i] %= H; }
        }
    }
    __device__ __forceinline__ bool ok(const St& s, int i) const {
        return s.tapok && (unsigned)(s.y[i] + s.dy) < (unsigned)H && (unsigned)(s.xx[i] + s.dx) < (unsigned)W;
    }
};

// transposed fragment of the 32x32x16 operand from the swizzled image: k = rows 16 s + 8 h + j (j = 0..7), column col0 + (lane & 31).
// `base` = the lane's byte address for k-step 0, first / second 4-row half (the XOR term does not depend on s).
__device__ __forceinline__ bf16x8 tr_frag2(const char* lo_base, const char* hi_base, int s) {
    const bf16x4 lo = SWIN_DS_READ_TR16((lds_bf16x4_d*)(lo_base + s * (16 * ROWB)));
    const bf16x4 hi = SWIN_DS_READ_TR16((lds_bf16x4_d*)(hi_base + s * (16 * ROWB)));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
}

template <int N> __device__ __forceinline__ void wait_vm();
template <> __device__ __forceinline__ void wait_vm<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<8>() { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vm<16>() { asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }

// NBUF stage buffers per k-group (NBUF - 1 stages of DMA in flight); WKG k-groups of 4 waves per block, each walking every
// WKG-th stage of the block's t range with its own ring, folded through LDS before the atomics (halves the atomic bytes at the
// same number of resident waves: the float-atomic rate is a chip-wide byte rate, guide G12).
template <typename XSrc, int NBUF, int WKG>
__device__ __forceinline__ void wgrad2_body(char* lds2, const int block_id, const bf16* __restrict__ dy, XSrc X, float* __restrict__ dw,
                                            float* __restrict__ dbias, int64_t T, int N1, int N2, int64_t t_per_split, int g2, int g1,
                                            int splits, int xcd_map) {
    int bx, by, bz;
    {
        int L = block_id;
        const int tiles = g1 * g2;
        if (xcd_map & 1) {                  // a split's tiles on ONE XCD (blocks b and b + 8 share one): its dY / X rows are fetched into that
            const int xcd = L & 7, q = L >> 3;      // L2 once and hit by the other tiles.  splits % 8 == 0 (host).
            bz = (q / tiles) * 8 + xcd;
            L = q % tiles;
        } else { bz = L / tiles; L -= bz * tiles; }
        bx = L % g2; by = L / g2;
        if (bz >= splits) return;
    }
    const int n1_0 = by * TN, n2_0 = bx * TN;
    const int64_t t_begin = (int64_t)bz * t_per_split;
    const int64_t t_end = min(T, t_begin + t_per_split);
    const int tid = threadIdx.x & 255, lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);         // wave-uniform by construction; the DMA's LDS address (M0)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                // must be PROVABLY so
    const int w1 = wave >> 1, w2 = wave & 1;
    char* ring = lds2 + (size_t)grp * NBUF * STAGEB;
    const unsigned ring_addr = (unsigned)(uintptr_t)(lptr_d)ring;

    // ---- DMA sources of this lane
    int row[4], chunk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        row[i] = 16 * wave + 4 * i + (lane >> 4);
        chunk[i] = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | i);          // f(row) with row & 3 = lane >> 4, (row >> 2) & 3 = i
    }
    const int64_t t_first = t_begin + (int64_t)grp * ST;
    uint64_t aptr[4]; bool aok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n1_0 + chunk[i] * 8;
        aok[i] = n < N1;
        aptr[i] = (uint64_t)reinterpret_cast<uintptr_t>(dy + (t_first + row[i]) * (int64_t)N1 + (aok[i] ? n : 0));
    }
    typename XSrc::St xs;
    X.init(xs, t_first, n2_0, row, chunk);
    const uint64_t zero64 = (uint64_t)reinterpret_cast<uintptr_t>(g_zero_line + ((block_id * 256 + tid) & 4095));
    int64_t t_stage = t_first;                                       // first row of the next stage to issue
    auto dma_stage = [&](int buf) {
        const int left = (int)min((int64_t)ST, t_end - t_stage);     // rows of this stage inside the split's range (<= 0: none)
        const unsigned a_dst = ring_addr + buf * STAGEB + wave * (16 * ROWB), b_dst = a_dst + OPB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool rin = row[i] < left;
            glds16((rin && aok[i]) ? aptr[i] : zero64, a_dst + i * (4 * ROWB));
            aptr[i] += (uint64_t)(WKG * ST) * (uint64_t)N1 * 2u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool rin = row[i] < left;
            glds16((rin && X.ok(xs, i)) ? xs.ptr[i] : zero64, b_dst + i * (4 * ROWB));
        }
        X.advance(xs, WKG * ST);
        t_stage += WKG * ST;
    };

    // ---- fragment read addresses (byte offsets inside an operand tile, k-step 0)
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int rlo = 8 * (g >> 1) + q;                                 // first 4-row half; the second is rlo + 4
    auto frag_off = [&](int col0, int hi) {
        const int r = rlo + 4 * hi;
        const int c0 = (col0 + 16 * (g & 1)) >> 3;                    // first 16-byte chunk of this 16-lane group's 16 columns
        const int f = ((r & 3) << 2) | ((r >> 2) & 3);
        return r * ROWB + 16 * ((c0 + (p >> 1)) ^ f) + 8 * (p & 1);
    };
    int a_off[2][2], b_off[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            a_off[i][hi] = frag_off(w1 * 64 + 32 * i, hi);
            b_off[i][hi] = OPB + frag_off(w2 * 64 + 32 * i, hi);
        }

    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        accb[a] = f32x16{0};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    }
    // bias gradient = column sums of dY = dY^T x ones: one more MFMA per (n1 sub-tile, k-step) in the waves of the n2-tile-0 blocks
    // that own the left half (w2 == 0) -- no LDS read, no VALU
    const bool do_bias = dbias != nullptr && bx == 0 && w2 == 0;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    auto compute = [&](int buf) {
        const char* st = ring + buf * STAGEB;
        bf16x8 af[2][2], bfr[2][2];
        auto frags = [&](int s, int b) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[b][i] = tr_frag2(st + a_off[i][0], st + a_off[i][1], s);
                bfr[b][i] = tr_frag2(st + b_off[i][0], st + b_off[i][1], s);
            }
        };
        frags(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) frags(s + 1, (s + 1) & 1);                    // in flight under this k-step's MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = SWIN_MFMA_32x32x16(af[s & 1][i], bfr[s & 1][j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 2; ++i) accb[i] = SWIN_MFMA_32x32x16(af[s & 1][i], ones, accb[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the ring: stage `it` of this k-group has landed for this wave when at most DPT x (stages issued after it) DMA
    // instructions are outstanding; the barrier publishes it to the other waves and, every wave having finished compute(it - 1)
    // before arriving, frees buffer (it - 1) % NBUF for stage it + NBUF - 1.  Every group runs the same number of iterations
    // (block-wide barriers); a stage beyond t_end is all zero lines.
    const int64_t stages = (t_end - t_begin + ST - 1) / ST;
    const int iters = (int)((stages + WKG - 1) / WKG);
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
        if (i < iters) dma_stage(i);
    for (int it = 0; it < iters; ++it) {
        const int after = min(NBUF - 2, iters - 1 - it);            // stages issued after stage `it` so far
        if (NBUF >= 4 && after >= 2) wait_vm<16>();
        else if (NBUF >= 3 && after >= 1) wait_vm<8>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (it + NBUF - 1 < iters) dma_stage((it + NBUF - 1) % NBUF);
        compute(it % NBUF);
    }
    const int c = lane & 31, h = lane >> 5;
    // One split and nobody else adding into this dW / db during the launch (xcd_map bit 2, set by the grouped launcher): a tile has a
    // single writer, so its sums go out as plain read-modify-writes instead of float atomics (1.3 TB/s chip-wide, guide G12).
    const bool plain = (xcd_map & 4) != 0;
    // ---- fold the k-groups through LDS (the ring is free: every DMA has landed and been consumed)
    if (WKG > 1) {
        float* red = reinterpret_cast<float*>(lds2);             // [128][128] fp32 = 64 KB (+ 128 bias sums behind it)
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        red[(w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h) * TN + w2 * 64 + 32 * j + c] = acc[i][j][reg];
            if (do_bias && c == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) red[TN * TN + w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h] = accb[i][reg];
            }
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        acc[i][j][reg] += red[(w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h) * TN + w2 * 64 + 32 * j + c];
            if (do_bias && c == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) accb[i][reg] += red[TN * TN + w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h];
            }
        }
        if (grp != 0) return;
    }
    // ---- D[row n1][col n2]: lane = n2 column, registers = n1 rows -> a wave instruction adds two 128-byte row segments
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n2 = n2_0 + w2 * 64 + 32 * j + c;
            if (n2 >= N2) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n1 = n1_0 + w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (n1 >= N1) continue;
                float* d = dw + (int64_t)n1 * N2 + n2;
                if (plain) *d += acc[i][j][reg]; else atomicAdd(d, acc[i][j][reg]);
            }
        }
    if (do_bias && c == 0) {                     // every column of accb holds the same sums: lane 0 of each half adds its 16 rows
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n1 = n1_0 + w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (n1 >= N1) continue;
                if (plain) dbias[n1] += accb[i][reg]; else atomicAdd(dbias + n1, accb[i][reg]);
            }
    }
}

template <typename XSrc, int NBUF, int WKG>
__global__ __launch_bounds__(256 * WKG, (NBUF * WKG <= 2) ? 2 : 1) void wgrad2_kernel(const bf16* __restrict__ dy, XSrc X, float* __restrict__ dw,
                                                                                   float* __restrict__ dbias, int64_t T, int N1, int N2,
                                                                                   int64_t t_per_split, int g2, int g1, int splits, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) char lds2[];          // [WKG][NBUF][dY tile | X tile]
    wgrad2_body<XSrc, NBUF, WKG>(lds2, (int)blockIdx.x, dy, X, dw, dbias, T, N1, N2, t_per_split, g2, g1, splits, xcd_map);
}

// ---- several Linear weight gradients in ONE launch.  The backward of the backbone produces four of them per block, each a
// 10-40 us launch that is bound by its own latency chain and by splits x |dW| bytes of float atomics (stage 3: 11-15 splits to fill the
// chip with one problem's 27-36 tiles).  Deferred and grouped (mixed.wgrad_flush: a stage's worth at a time) the launch has hundreds of
// tiles, so a problem needs no or few splits, the loops are long enough for the DMA ring to stream, and the tail of one problem
// overlaps the next.  Block -> (problem, split, tile): first_block[] is the running block count.
constexpr int GMAX = 32;
struct GProb { const bf16* dy; const bf16* x; float* dw; float* db; int64_t T, per; int N1, N2, g1, g2, splits, pad; };
struct GTab { int n; int first_block[GMAX + 1]; GProb p[GMAX]; };

__global__ __launch_bounds__(256, 2) void wgrad2_group_kernel(const GTab tab) {
    extern __shared__ __attribute__((aligned(16))) char lds2[];
    int pi = 0;
    const int id = (int)blockIdx.x;
    while (pi + 1 < tab.n && id >= tab.first_block[pi + 1]) ++pi;        // block-uniform scalar search over <= 32 entries
    const GProb& q = tab.p[pi];
    PlainSrc X{q.x, q.N2};
    wgrad2_body<PlainSrc, 2, 1>(lds2, id - tab.first_block[pi], q.dy, X, q.dw, q.db, q.T, q.N1, q.N2, q.per, q.g2, q.g1, q.splits, q.pad ? 4 : 0);
}

template <typename XSrc> inline void set_step(XSrc&, int) {}
template <> inline void set_step<ConvSrc>(ConvSrc& X, int dt) { X.qd = dt / X.W; X.rd = dt % X.W; }

template <typename XSrc, int NBUF, int WKG>
int launch2(const bf16* dy, XSrc X, float* dw, float* dbias, int64_t T, int N1, int N2, int want_blocks, bool xcd, hipStream_t s) {
    set_step(X, WKG * ST);
    const size_t lds_bytes = (size_t)WKG * NBUF * STAGEB + (WKG > 1 && WKG * NBUF * STAGEB < TN * TN * 4 + 512 ? 512 : 0);
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)wgrad2_kernel<XSrc, NBUF, WKG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) !=
            hipSuccess)
            return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const int g1 = (N1 + TN - 1) / TN, g2 = (N2 + TN - 1) / TN, tiles = g1 * g2;
    const int64_t stages = (T + ST - 1) / ST;
    const int64_t max_splits = (stages + WKG - 1) / WKG;
    int64_t splits = (want_blocks + tiles - 1) / tiles;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (xcd) {
        int64_t s8 = (splits + 4) / 8 * 8;
        if (s8 < 8) s8 = 8;
        if (s8 > max_splits) s8 = max_splits / 8 * 8;
        if (s8 >= 8) splits = s8; else xcd = false;
    }
    int64_t per = (stages + splits - 1) / splits;
    per = ((per + WKG - 1) / WKG) * WKG * ST;
    splits = (T + per - 1) / per;
    if (xcd && splits % 8 != 0) splits = (splits + 7) / 8 * 8;       // the map deals splits to XCDs in eights; surplus splits are empty
    if (splits * tiles > (int64_t)1 << 30) return SWIN_ERR_UNSUPPORTED;
    const unsigned nblk = (unsigned)(splits * tiles);
    wgrad2_kernel<XSrc, NBUF, WKG><<<nblk, 256 * WKG, lds_bytes, s>>>(dy, X, dw, dbias, T, N1, N2, per, g2, g1, (int)splits, xcd ? 1 : 0);
    return swin_launch_status();
}

struct Plan2 { int nbuf, wkg, blocks; bool xcd; };

// Launch geometry by shape class, from a sweep on MI355X (tools/wgrad_sweep.py with the -DSWIN_DEV library, profiles/r03_wgrad_sweep.txt):
// two stage buffers at two blocks per CU beat deeper rings at one block per CU on every shape (the limit is operand traffic beyond
// L2 and the atomics' byte rate, not the DMA latency); the XCD-local tile order pays when a split has many tiles (its operand rows
// are then fetched into one L2 and hit by the other tiles).  blocks = thread blocks to aim for (the split count follows from it).
Plan2 choose(int64_t T, int tiles, bool conv) {
    Plan2 p;
    p.nbuf = 2; p.wkg = 1;
    if (conv) { p.blocks = T >= 100000 ? 768 : 384; p.xcd = T >= 100000 || T < 24000; }
    else { p.blocks = T >= 6000 ? 384 : 256; p.xcd = T >= 6000; }
    (void)tiles;
#ifdef SWIN_DEV
    const int nb = swin_dev_int("SWIN_WGRAD2_NBUF", 0), kg = swin_dev_int("SWIN_WGRAD2_WKG", 0), bl = swin_dev_int("SWIN_WGRAD2_BLOCKS", 0),
              xc = swin_dev_int("SWIN_WGRAD2_XCD", -1);
    if (nb) p.nbuf = nb;
    if (kg) p.wkg = kg;
    if (bl) p.blocks = bl;
    if (xc >= 0) p.xcd = xc != 0;
#endif
    return p;
}

template <typename XSrc>
int dispatch2(const bf16* dy, XSrc X, float* dw, float* dbias, int64_t T, int N1, int N2, bool conv, hipStream_t s) {
    const int tiles = ((N1 + TN - 1) / TN) * ((N2 + TN - 1) / TN);
    // Few output tiles (the stage-1 / stage-2 Linear layers: 1-16 tiles, t up to 128 000) stay on the register-staged kernel: those
    // shapes are bound by the HBM stream, which it already drives at 4.5 TB/s, and their N1 / N2 are not multiples of 128 -- here a
    // quarter of the DMA lanes would read the one zero line (measured: 50 vs 39 us at 384 x 96, 67 vs 26 us at 96 x 96).
    if (tiles <= 16 && swin_dev_int("SWIN_WGRAD2_FORCE", 0) == 0) return SWIN_ERR_UNSUPPORTED;
    const Plan2 p = choose(T, tiles, conv);
    const int blocks = p.blocks;
#ifdef SWIN_DEV             // the deeper rings / two k-groups exist only in development builds (sweeps)
    if (p.wkg == 2 && p.nbuf == 2) return launch2<XSrc, 2, 2>(dy, X, dw, dbias, T, N1, N2, blocks, p.xcd, s);
    if (p.wkg == 1 && p.nbuf == 3) return launch2<XSrc, 3, 1>(dy, X, dw, dbias, T, N1, N2, blocks, p.xcd, s);
    if (p.wkg == 1 && p.nbuf == 4) return launch2<XSrc, 4, 1>(dy, X, dw, dbias, T, N1, N2, blocks, p.xcd, s);
#endif
    return launch2<XSrc, 2, 1>(dy, X, dw, dbias, T, N1, N2, blocks, p.xcd, s);
}


// ------------------------------------------------------------------------------------------------------------------------------
// Third form (Linear layers with MANY output tiles and a short t axis: stages 3-4 of the backbone, the heads' fully connected
// layers): 64 x 64 output tiles, and the block's 8 waves split the CONTRACTION instead of the tile.  With 128 x 128 tiles those shapes
// need 7-15 splits of t to fill the chip, and splits x |dW| bytes of float atomics (26 MB per launch at 1536 x 384 = 20 us at the
// chip-wide 1.3 TB/s, PMC) -- four times as many tiles need a quarter of the splits.  Every wave owns the WHOLE tile (2 x 2
// accumulators) and every 8th 32-row slice of the block's t range, with a ring of two 8 KB stage buffers of its own: it reads only
// what its own DMA wrote, so the loop has NO block-level barrier -- a counted s_waitcnt vmcnt orders a wave's ds_reads behind its
// own LDS-DMA.  The eight partial tiles are folded through LDS once at the end (each wave sums and adds one eighth of the tile).
// LDS row (256 B) = [dY 64 columns | X 64 columns] with the same 16-chunk XOR as above.
constexpr int S3 = 32;               // rows of t per wave-stage
constexpr int W3 = 8;                // waves per block (k-split)
constexpr int T3 = 64;               // tile width (n1 and n2)
constexpr int STAGE3 = S3 * ROWB;    // 8 KB

__global__ __launch_bounds__(64 * W3, 2) void wgrad3_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ dbias, int64_t T, int N1, int N2, int64_t t_per_split,
                                                           int g2, int g1, int splits) {
    extern __shared__ __attribute__((aligned(16))) char lds3[];          // [W3][2][32 rows x 256 B] = 128 KB; reused for the fold
    int bx, by, bz;
    {
        // XCD-aware order (blocks b and b + 8 share an XCD): every XCD gets a contiguous run of (split, tile) pairs, n2 tiles fastest,
        // so the blocks that share an L2 read the same dY column block and neighbouring X column blocks of the same t range
        const int nblk = g1 * g2 * splits;
        int id = blockIdx.x;
        const int q = nblk / 8, r = nblk % 8, xcd = id % 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
        const int tiles = g1 * g2;
        bz = id / tiles; id -= bz * tiles;
        // the SHORTER tile axis runs fastest: an XCD's run of tiles then spans a near-square patch of the tile grid, i.e. the fewest
        // operand column blocks per tile (768 x 3072: 12 x 6 tiles = 18 column blocks per XCD instead of 1.5 x 48 = 49.5)
        if (g1 <= g2) { bx = id / g1; by = id - bx * g1; }
        else { by = id / g2; bx = id - by * g2; }
    }
    const int n1_0 = by * T3, n2_0 = bx * T3;
    const int64_t t_begin = (int64_t)bz * t_per_split;
    const int64_t t_end = min(T, t_begin + t_per_split);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = lds3 + (size_t)wave * 2 * STAGE3;
    const unsigned ring_addr = (unsigned)(uintptr_t)(lptr_d)ring;

    // ---- DMA sources: instruction i (0..7) of a stage covers rows 4 i + (lane >> 4); the lane's chunk position lane & 15 holds chunk
    // (lane & 15) ^ f(row), f(row) = ((lane >> 4) << 2) | (i & 3): instructions i and i + 4 read the same chunk 16 rows apart
    uint64_t ptr[4]; unsigned rs16[4]; bool cok[4];
    const int r0 = lane >> 4;
    const int64_t t_first = t_begin + (int64_t)wave * S3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = (lane & 15) ^ ((r0 << 2) | i);
        const bool isx = ch >= 8;
        const int n = (isx ? n2_0 : n1_0) + (ch & 7) * 8, N = isx ? N2 : N1;
        cok[i] = n < N;
        const bf16* base = isx ? x : dy;
        ptr[i] = (uint64_t)reinterpret_cast<uintptr_t>(base + (t_first + 4 * i + r0) * (int64_t)N + (cok[i] ? n : 0));
        rs16[i] = (unsigned)N * 32u;                                  // bytes per 16 rows
    }
    const uint64_t zero64 = (uint64_t)reinterpret_cast<uintptr_t>(g_zero_line + ((blockIdx.x * 512 + threadIdx.x) & 4095));
    int64_t t_stage = t_first;
    auto dma_stage = [&](int buf) {
        const int left = (int)min((int64_t)S3, t_end - t_stage);
        const unsigned dst = ring_addr + buf * STAGE3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = i & 3;
            const bool ok = cok[k] && (4 * i + r0) < left;
            glds16(ok ? ptr[k] + (uint64_t)((i >> 2) * rs16[k]) : zero64, dst + i * (4 * ROWB));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) ptr[k] += (uint64_t)rs16[k] * (uint64_t)(W3 * S3 / 16);
        t_stage += W3 * S3;
    };

    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int rlo = 8 * (g >> 1) + q;
    auto frag_off = [&](int col0, int hi) {
        const int r = rlo + 4 * hi;
        const int c0 = (col0 + 16 * (g & 1)) >> 3;
        const int f = ((r & 3) << 2) | ((r >> 2) & 3);
        return r * ROWB + 16 * ((c0 + (p >> 1)) ^ f) + 8 * (p & 1);
    };
    int a_off[2][2], b_off[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            a_off[i][hi] = frag_off(32 * i, hi);
            b_off[i][hi] = frag_off(64 + 32 * i, hi);
        }
    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        accb[a] = f32x16{0};
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    }
    const bool do_bias = dbias != nullptr && bx == 0;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    const int64_t slices = (t_end - t_begin + S3 - 1) / S3;
    const int iters = (int)((slices - wave + W3 - 1) / W3);             // this wave's slices: wave, wave + 8, ...
    if (iters > 0) dma_stage(0);
    if (iters > 1) dma_stage(1);
    for (int it = 0; it < iters; ++it) {
        if (it + 1 < iters) wait_vm<8>(); else wait_vm<0>();            // this wave's own DMA of stage `it` has landed
        const char* st = ring + (it & 1) * STAGE3;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = tr_frag2(st + a_off[i][0], st + a_off[i][1], s);
                bfr[i] = tr_frag2(st + b_off[i][0], st + b_off[i][1], s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = SWIN_MFMA_32x32x16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 2; ++i) accb[i] = SWIN_MFMA_32x32x16(af[i], ones, accb[i], 0, 0, 0);
            }
        }
        // every ds_read of this buffer has returned (the MFMAs consumed them): it may be overwritten
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it + 2 < iters) dma_stage(it & 1);
    }
    // ---- fold the 8 partial tiles through LDS: part[wave][row][col] fp32 (8 x 16 KB = the whole ring)
    const int c = lane & 31, h = lane >> 5;
    __syncthreads();                                                    // every wave is done with its ring
    float* part = reinterpret_cast<float*>(lds3);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                part[(wave * T3 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h) * T3 + 32 * j + c] = acc[i][j][reg];
    __syncthreads();
    {
        const int col = lane, n2 = n2_0 + col;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = 8 * wave + r, n1 = n1_0 + row;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < W3; ++w) v += part[(w * T3 + row) * T3 + col];
            if (n1 < N1 && n2 < N2) atomicAdd(dw + (int64_t)n1 * N2 + n2, v);
        }
    }
    if (do_bias && c == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n1 = n1_0 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (n1 < N1) atomicAdd(dbias + n1, accb[i][reg]);
            }
    }
}

int launch3(const bf16* dy, const bf16* x, float* dw, float* dbias, int64_t T, int N1, int N2, hipStream_t s) {
    const size_t lds_bytes = (size_t)W3 * 2 * STAGE3;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)wgrad3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const int g1 = (N1 + T3 - 1) / T3, g2 = (N2 + T3 - 1) / T3;
    const int64_t tiles = (int64_t)g1 * g2;
    const int64_t slices = (T + S3 - 1) / S3;
    // one 128 KB block per CU: as many splits as fill the chip once (development: SWIN_WGRAD3_BLOCKS), every wave with >= 2 slices
    const int want = swin_dev_int("SWIN_WGRAD3_BLOCKS", 256);
    int64_t splits = (want + tiles - 1) / tiles;
    const int64_t max_splits = (slices + 2 * W3 - 1) / (2 * W3);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t per = (slices + splits - 1) / splits;
    per = ((per + W3 - 1) / W3) * W3 * S3;
    splits = (T + per - 1) / per;
    if (splits * tiles > (int64_t)1 << 30) return SWIN_ERR_UNSUPPORTED;
    wgrad3_kernel<<<(unsigned)(splits * tiles), 64 * W3, lds_bytes, s>>>(dy, x, dw, dbias, T, N1, N2, per, g2, g1, (int)splits);
    return swin_launch_status();
}


// ---- recorded problems (swin_wgrad_record), per device; swin_wgrad_flush launches them as grouped kernels (<= GMAX problems each).
// The caller keeps every operand alive and unmodified until the flush and does not read the accumulators before it.
struct Pending { const bf16* dy; const bf16* x; float* dw; float* db; int64_t T; int N1, N2; };
std::vector<Pending> g_pending[16];

int cur_device() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? dev : -1;
}

int launch_group(const Pending* pp, int n, hipStream_t s) {
    static bool attr_set[16] = {};
    const int dev = cur_device();
    if (dev < 0) return SWIN_ERR_UNSUPPORTED;
    const size_t lds_bytes = 2 * STAGEB;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)wgrad2_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    // splits: every block should walk about the same number of stages, and the launch should have >= ~768 blocks (1.5 rounds of
    // two per CU) unless the problems are too short for that
    int64_t work = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t tiles = (int64_t)((pp[i].N1 + TN - 1) / TN) * ((pp[i].N2 + TN - 1) / TN);
        work += tiles * ((pp[i].T + ST - 1) / ST);
    }
    const int want_blocks = swin_dev_int("SWIN_WGRADG_BLOCKS", 768);
    int64_t L = work / want_blocks;
    if (L < 6) L = 6;                                   // stages per block to aim for
    GTab tab;
    tab.n = n;
    int64_t nblk = 0;
    for (int i = 0; i < n; ++i) {
        GProb& q = tab.p[i];
        q.dy = pp[i].dy; q.x = pp[i].x; q.dw = pp[i].dw; q.db = pp[i].db; q.T = pp[i].T; q.N1 = pp[i].N1; q.N2 = pp[i].N2;
        q.g1 = (q.N1 + TN - 1) / TN; q.g2 = (q.N2 + TN - 1) / TN; q.pad = 0;
        const int64_t stages = (q.T + ST - 1) / ST;
        int64_t splits = (stages + L - 1) / L;
        if (splits < 1) splits = 1;
        int64_t per = (stages + splits - 1) / splits * ST;
        splits = (q.T + per - 1) / per;
        q.per = per; q.splits = (int)splits;
        // pad = 1: plain read-modify-write epilogue -- one split, and no other problem of this launch accumulates into the same dW / db
        q.pad = splits == 1 ? 1 : 0;
        for (int k = 0; k < n && q.pad; ++k)
            if (k != i && (pp[k].dw == pp[i].dw || (pp[i].db && pp[k].db == pp[i].db))) q.pad = 0;
        tab.first_block[i] = (int)nblk;
        nblk += splits * q.g1 * q.g2;
        if (nblk > (int64_t)1 << 30) return SWIN_ERR_UNSUPPORTED;
    }
    tab.first_block[n] = (int)nblk;
    wgrad2_group_kernel<<<(unsigned)nblk, 256, lds_bytes, s>>>(tab);
    return swin_launch_status();
}

}  // namespace

extern "C" int wgrad_linear_bf16(const void* dy, const void* x, float* dw, float* dbias, int64_t T, int N1, int N2, void* stream);
// csrc/wgrad96.hip
int wgrad96_class(int64_t T, int N1, int N2);
int wgrad96_launch(const void* const* dy, const void* const* x, float* const* dw, float* const* db, const int64_t* T, const int* N1,
                   const int* N2, int n, int blocks, hipStream_t s);

// Called by the C ABI entry points of csrc/wgrad_gemm.hip; SWIN_ERR_UNSUPPORTED = not this kernel's shape (the caller falls
// back to the register-staged kernel).
int wgrad2_linear(const void* dy, const void* x, float* dw, float* dbias, int64_t T, int N1, int N2, void* stream) {
    if (N1 % 8 || N2 % 8 || T < 1) return SWIN_ERR_UNSUPPORTED;
    // many output tiles, short t: the 64 x 64 / wave-private-ring form (a quarter of the float-atomic bytes)
    const int form = swin_dev_int("SWIN_WGRAD_FORM", 0);            // development: 2 / 3 force a form
    const int64_t tiles128 = (int64_t)((N1 + TN - 1) / TN) * ((N2 + TN - 1) / TN);
    if (form == 3 || (form == 0 && tiles128 > 16 && T <= 16384))
        return launch3((const bf16*)dy, (const bf16*)x, dw, dbias, T, N1, N2, (hipStream_t)stream);
    PlainSrc X{(const bf16*)x, N2};
    return dispatch2((const bf16*)dy, X, dw, dbias, T, N1, N2, false, (hipStream_t)stream);
}

int wgrad2_conv3x3(const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W, int Cin, int Cout, void* stream) {
    if (Cin % TN || Cout % 8) return SWIN_ERR_UNSUPPORTED;
    const int64_t T = (int64_t)N * H * W;
    ConvSrc X{(const bf16*)x, H, W, Cin, 0, 0};
    return dispatch2((const bf16*)dy, X, dw, dbias, T, Cout, 9 * Cin, true, (hipStream_t)stream);
}

// ---- recording API (include/swin_hip.h)
extern "C" int swin_wgrad_record(const void* dy, const void* x, float* dw, float* dbias, int64_t T, int N1, int N2) {
    if (!dy || !x || !dw || T <= 0 || N1 <= 0 || N2 <= 0) return SWIN_ERR_BAD_ARG;
    if (N1 % 8 || N2 % 8) return SWIN_ERR_UNSUPPORTED;
    const int dev = cur_device();
    if (dev < 0) return SWIN_ERR_UNSUPPORTED;
    g_pending[dev].push_back(Pending{(const bf16*)dy, (const bf16*)x, dw, dbias, T, N1, N2});
    return SWIN_OK;
}

// problems recorded and not launched yet on the current device; *tiles (nullable) = their 128 x 128 output tiles in total
extern "C" int swin_wgrad_pending(int64_t* tiles) {
    const int dev = cur_device();
    if (dev < 0) return 0;
    if (tiles) {
        int64_t t = 0;
        for (const Pending& q : g_pending[dev]) t += (int64_t)((q.N1 + TN - 1) / TN) * ((q.N2 + TN - 1) / TN);
        *tiles = t;
    }
    return (int)g_pending[dev].size();
}

extern "C" int swin_wgrad_flush(void* stream) {
    const int dev = cur_device();
    if (dev < 0) return SWIN_ERR_UNSUPPORTED;
    std::vector<Pending>& all = g_pending[dev];
    int rc = SWIN_OK;
#ifdef SWIN_DEV
    if (swin_dev_int("SWIN_WGRAD_DUMP", 0)) {
        fprintf(stderr, "wgrad flush: %zu problems\n", all.size());
        for (const Pending& q : all) fprintf(stderr, "  T=%lld N1=%d N2=%d db=%d narrow=%d\n", (long long)q.T, q.N1, q.N2, q.db != nullptr, wgrad96_class(q.T, q.N1, q.N2));
    }
#endif
    // layers whose dimensions are multiples of 96 (Swin-T / Swin-S, every stage) go to csrc/wgrad96.hip, up to 32 problems per launch
    std::vector<Pending> v, nar;
    // (t >= 4096: at stage 4, t = 2000, the 128-tile form owns whole tiles with plain read-modify-write epilogues -- measured equal
    // alone and better next to the FPN laterals that share its launch)
    for (const Pending& q : all)
        (q.T >= swin_dev_int("SWIN_WGRAD96_MIN_T", 4096) && wgrad96_class(q.T, q.N1, q.N2) >= 0 && swin_dev_int("SWIN_WGRAD96", 1) ? nar : v).push_back(q);
    if (!nar.empty() && v.size() <= 2) {                         // a straggler or two of the other form (PatchMerging at t = 2000) ride along
        bool ok = true;
        for (const Pending& q : v) ok = ok && wgrad96_class(q.T, q.N1, q.N2) >= 0;
        if (ok) { nar.insert(nar.end(), v.begin(), v.end()); v.clear(); }
    }
    for (size_t i = 0; i < nar.size(); i += 32) {
        const int n = (int)std::min((size_t)32, nar.size() - i);
        const void* dy[32]; const void* x[32]; float* dw[32]; float* db[32]; int64_t T[32]; int N1[32], N2[32];
        for (int k = 0; k < n; ++k) {
            const Pending& q = nar[i + k];
            dy[k] = q.dy; x[k] = q.x; dw[k] = q.dw; db[k] = q.db; T[k] = q.T; N1[k] = q.N1; N2[k] = q.N2;
        }
        const int st = wgrad96_launch(dy, x, dw, db, T, N1, N2, n, swin_dev_int("SWIN_WGRAD96_BLOCKS", 0), (hipStream_t)stream);
        if (st != SWIN_OK && rc == SWIN_OK) rc = st;
    }
    // One or two problems of a few tiles left on their own (the patch embedding's 96 x 48 once the backbone's layers went the other
    // way): the grouped launch would cut each into hundreds of splits -- the single-problem entry streams them.
    int64_t left_tiles = 0;
    for (const Pending& q : v) left_tiles += (int64_t)((q.N1 + TN - 1) / TN) * ((q.N2 + TN - 1) / TN);
    if (!v.empty() && v.size() <= 2 && left_tiles <= 8) {
        for (const Pending& q : v) {
            const int st = wgrad_linear_bf16(q.dy, q.x, q.dw, q.db, q.T, q.N1, q.N2, stream);
            if (st != SWIN_OK && rc == SWIN_OK) rc = st;
        }
        v.clear();
    }
    for (size_t i = 0; i < v.size(); i += GMAX) {
        const int n = (int)std::min((size_t)GMAX, v.size() - i);
        const int st = launch_group(v.data() + i, n, (hipStream_t)stream);
        if (st != SWIN_OK && rc == SWIN_OK) rc = st;
    }
    all.clear();
    return rc;
}
