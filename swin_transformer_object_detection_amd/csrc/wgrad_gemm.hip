// Weight-gradient GEMM for gfx950: dW[n1][n2] (+)= sum_t dY[t][n1] * X[t][n2], bf16 in, fp32 out.
// The contraction runs over the TOKEN / PIXEL index t (10^4..10^5 long) while n1, n2 are a few hundred, so
//   * the t range is split over grid.z and the partial tiles are combined with contiguous fp32 atomics
//     (a 128x128 tile = 64 KB per split: far below the atomic-rate budget, guide G12);
//   * both MFMA operands are "column" reads of row-major [t][n] LDS tiles -> ds_read_b64_tr_b16.
// X is either a plain (T, N2) matrix (Linear layers: dW = dY^T X) or the implicit im2col of a channels-last
// activation for a 3x3/pad-1 convolution (N2 = 9*Cin ordered (ky,kx,cin); replaces torch.cat + library GEMM).
// Reference sites: the autograd of nn.Linear at swin_transformer.py:129,151,33,36,296 and of the 3x3 convs at
// fpn.py:195-197, rpn_head.py:43, fcn_mask_head.py:119-121.
#include <cstdlib>

#include "common.h"

#define WT 64          // t rows per stage
#define WN 128         // tile width (n1 and n2)
#define WROW 160       // LDS row stride in bf16 (320 B: conflict-free 4-row transposed reads)

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_w;

// X loaders keep an incremental per-row state (element offset of this thread's 8 columns at row t, plus the pixel
// coordinates for the conv): a stage advances it by a constant row step with 32-bit adds.  The first version
// recomputed t * N2 + n (64-bit multiply) and, for the conv, t % W and (t / W) % H (64-bit divisions) for every
// 16-byte piece of every stage -- more integer work than MFMA work in the loop.  The kernel loads from a clamped
// address and masks by value (a pointer select against a local zero becomes a flat load + full wait).
struct PlainX {
    static constexpr int kBlocksManyTiles = 384;
    const bf16* x; int64_t T; int N2;
    struct St { int64_t off; bool ok; };
    __device__ __forceinline__ void init(St& s, int64_t t, int n) const { s.ok = n < N2; s.off = t * N2 + n; }
    __device__ __forceinline__ void advance(St& s, int dt) const { s.off += (int64_t)dt * N2; }
    __device__ __forceinline__ int64_t offset(const St& s) const { return s.ok ? s.off : -1; }
};

struct ConvX {
    static constexpr int kBlocksManyTiles = 512;
    const bf16* x; int64_t T; int H, W, Cin;
    int qd, rd;                       // a stage advances every row by dt = WKG * WT pixels: dt / W and dt % W (host-computed)
    struct St { int64_t off; int y, x, dy, dx; bool ok; };
    __device__ __forceinline__ void init(St& s, int64_t t, int n) const {
        const int tap = n / Cin, c = n - tap * Cin;
        s.dy = tap / 3 - 1; s.dx = tap - (tap / 3) * 3 - 1;
        s.ok = tap < 9;
        s.x = (int)(t % W); s.y = (int)((t / W) % H);
        s.off = (t + (int64_t)s.dy * W + s.dx) * Cin + c;       // the tap's source pixel, this thread's 8 channels
    }
    // dt is the same for every stage of a launch, so the division is done once on the host: 6 VALU per row instead of an
    // integer division in a divergent branch (the first form spent 14 VALU per MFMA on this, PMC profiles/r02_pmc_wgrad*)
    __device__ __forceinline__ void advance(St& s, int dt) const {
        s.off += (int64_t)dt * Cin;
        s.x += rd; s.y += qd;
        if (s.x >= W) { s.x -= W; s.y += 1; }
        if (s.y >= H) { s.y -= H; if (s.y >= H) s.y %= H; }
    }
    __device__ __forceinline__ int64_t offset(const St& s) const {
        const unsigned yy = (unsigned)(s.y + s.dy), xx = (unsigned)(s.x + s.dx);
        return (s.ok && yy < (unsigned)H && xx < (unsigned)W) ? s.off : -1;
    }
};

// transposed fragment: rows t0 + 8*h + j (j = 0..7) of column (col0 + lane&31) of a [WT][WROW] tile
__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int t0, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, dh = g & 1;
    const bf16* a0 = tile + (t0 + 8 * h + q) * WROW + col0 + 16 * dh + 4 * p;
    bf16x4 lo = SWIN_DS_READ_TR16((lds_bf16x4_w*)a0);
    bf16x4 hi = SWIN_DS_READ_TR16((lds_bf16x4_w*)(a0 + 4 * WROW));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
}

// KG k-groups of 4 waves per block: group g takes the stages g, g + KG, ... of the block's t range with its own LDS
// stage buffer; the groups' accumulators are folded through LDS before the global atomics.  The float-atomic rate is
// a chip-wide byte rate (guide G12): blocks x tile bytes is what it prices, so KG = 2 halves that cost at the same
// number of resident waves.
template <typename XLoader, int WKG>
__global__ __launch_bounds__(256 * WKG, WKG == 1 ? 2 : 1) void wgrad_kernel(const bf16* __restrict__ dy, XLoader X, float* __restrict__ dw,
                                                         float* __restrict__ dbias, int64_t T, int N1, int N2,
                                                         int64_t t_per_split, int mode, int g2, int g1,
                                                         int splits) {
    extern __shared__ __attribute__((aligned(16))) bf16 lds_all[];         // [WKG][dY|X][WT * WROW]: 40 KB per group
    // XCD-aware block -> (tile, split) map.  Hardware deals consecutive block ids round-robin over the 8 XCDs (private L2s).
    // Every output tile of one split reads the SAME token range, so a split's tiles belong on ONE XCD: there its dY / X rows
    // are fetched into that L2 once and hit by the other tiles.  The plain (x, y, z) order put the tiles of a split on all
    // eight XCDs: each L2 fetched nearly every operand byte (PMC, round 2: 43 % L2 hit rate, 3-6x the unique bytes fetched).
    // (The opposite affinity -- all splits of a TILE on one XCD, so that the fp32 atomics onto its 64 KB stay in one L2 -- was
    // measured too, round 2: no gain on any shape, 38 -> 57 us at 768x192 and 262 -> 292 us on the P2 conv where it unbalances
    // the XCDs.  The atomics are not limited by lines migrating between L2s.)
    int bx, by, bz;
    if (g2 > 0) {
        const int L = blockIdx.x, xcd = L & 7, q = L >> 3, tiles = g1 * g2;
        bz = (q / tiles) * 8 + xcd;
        if (bz >= splits) return;
        const int tile = q % tiles;
        bx = tile % g2; by = tile / g2;
    } else { bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z; }
    const int n1_0 = by * WN, n2_0 = bx * WN;
    const int64_t t_begin = (int64_t)bz * t_per_split;
    const int64_t t_end = min(T, t_begin + t_per_split);
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    bf16* ldsA = lds_all + (size_t)grp * 2 * WT * WROW;
    bf16* ldsB = ldsA + WT * WROW;
    const int w1 = wave >> 1, w2 = wave & 1;                 // wave -> 64x64 sub-tile (n1, n2)
    const int c = lane & 31, h = lane >> 5;
    // staging: tile = 64 rows x 16 pieces of 16 B; thread handles rows (tid/16 + 16 i), piece tid%16
    const int srow = tid >> 4, spiece = tid & 15;
    uint4 ra[4], rb[4];
    // row states of this thread's 4 staging rows (t = first stage row + srow + 16 i), advanced by a constant step
    typename XLoader::St xs[4];
    int64_t aoff[4], trow[4];
    const int n1c = n1_0 + spiece * 8;
    const bool n1ok = n1c < N1;
    {
        const int64_t tfirst = t_begin + (int64_t)grp * WT + srow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            trow[i] = tfirst + 16 * i;
            aoff[i] = trow[i] * N1 + n1c;
            X.init(xs[i], trow[i], n2_0 + spiece * 8);
        }
    }
    // Loads are issued RAW from a clamped address and zeroed by value only when they are stored to LDS (gmask, after the
    // MFMAs of the current stage): a select right behind the load makes the compiler wait for the load on the spot
    // (s_waitcnt vmcnt(0)), i.e. the "prefetch" of the next stage was not overlapping anything.
    bool oka[4], okb[4];
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool tin = trow[i] < t_end;
            const int64_t ob = tin ? X.offset(xs[i]) : -1;
            oka[i] = tin && n1ok; okb[i] = ob >= 0;
            ra[i] = *(const uint4*)(dy + (oka[i] ? aoff[i] : 0));
            rb[i] = *(const uint4*)(X.x + (okb[i] ? ob : 0));
        }
    };
    auto gmask = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!oka[i]) ra[i] = uint4{0u, 0u, 0u, 0u};
            if (!okb[i]) rb[i] = uint4{0u, 0u, 0u, 0u};
        }
    };
    auto gadvance = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            trow[i] += WKG * WT;
            aoff[i] += (int64_t)WKG * WT * N1;
            X.advance(xs[i], WKG * WT);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = srow + 16 * i;
            *(uint4*)&ldsA[row * WROW + spiece * 8] = ra[i];
            *(uint4*)&ldsB[row * WROW + spiece * 8] = rb[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
    // bias gradient = column sums of dY: taken by the n2-tile-0 blocks from the pieces they stage anyway
    const bool do_bias = dbias != nullptr && bx == 0;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto bias_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 v = *(const bf16x8*)&ra[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += (float)v[e];
        }
    };

    // every group runs the same number of iterations (block-wide barriers); a stage beyond t_end loads zeros
    const int64_t stages = (t_end - t_begin + WT - 1) / WT;
    const int iters = (int)((stages + WKG - 1) / WKG);
    if (iters > 0) {
        gload();
        gmask();
        if (do_bias) bias_acc();
        lstore();
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            const bool more = it + 1 < iters;
            if (more) { gadvance(); gload(); }                // in flight during the MFMAs below
#pragma unroll
            for (int s = 0; s < 4; ++s) {                     // 16 t per MFMA k-step
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = tr_frag(ldsA, 16 * s, w1 * 64 + 32 * i, lane);     // A[row n1][k = t]
                    bfr[i] = tr_frag(ldsB, 16 * s, w2 * 64 + 32 * i, lane);    // B[k = t][col n2]
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = SWIN_MFMA_32x32x16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();                                  // everyone is done reading this stage
            if (more) { gmask(); if (do_bias) bias_acc(); lstore(); }
            __syncthreads();
        }
    }
    float* red = reinterpret_cast<float*>(lds_all);           // 80 KB: [128][128] tile or [WKG*16][128] bias rows
    if (do_bias) {           // 16 threads (srow) x WKG groups share a piece column: reduce through LDS, one atomic per channel
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(grp * 16 + srow) * WN + spiece * 8 + e] = bsum[e];
        __syncthreads();
        if (threadIdx.x < WN) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 16 * WKG; ++r) a += red[r * WN + threadIdx.x];
            if (n1_0 + (int)threadIdx.x < N1) atomicAdd(dbias + n1_0 + threadIdx.x, a);
        }
    }
    // fold the k-groups: groups > 0 park their accumulators in LDS ([n1][n2] fp32, 64 KB), group 0 adds them
    for (int g = 1; g < WKG; ++g) {
        __syncthreads();
        if (grp == g) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        red[(w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h) * WN + w2 * 64 + 32 * j + c] = acc[i][j][reg];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        acc[i][j][reg] += red[(w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h) * WN + w2 * 64 + 32 * j + c];
        }
    }
    if (grp != 0) return;
    // D[row n1][col n2]: lane = n2 column, registers = n1 rows -> 32 consecutive n2 per half-wave: 128-B segments.
    // mode 0: this block owns the tile (one split): plain read-modify-write.  mode 1: contiguous fp32 atomics.
    // (A third form -- plain-stored partial slabs + a reduce launch -- measured slower on every shape, round 2,
    // profiles/r02_wgrad_modes.txt: the atomics overlap the other resident blocks' MFMAs, a reduce launch does not.)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int n2 = n2_0 + w2 * 64 + 32 * j + c;
            if (n2 >= N2) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                int n1 = n1_0 + w1 * 64 + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (n1 >= N1) continue;
                float* p = dw + (int64_t)n1 * N2 + n2;
                if (mode == 1) atomicAdd(p, acc[i][j][reg]);
                else *p += acc[i][j][reg];
            }
        }
}

template <typename XLoader> static inline void set_stage_step(XLoader&, int) {}
template <> inline void set_stage_step<ConvX>(ConvX& X, int dt) { X.qd = dt / X.W; X.rd = dt % X.W; }

template <typename XLoader, int WKG>
static int wgrad_launch_kg(const bf16* dy, XLoader X, float* dw, float* dbias, int64_t T, int N1, int N2, hipStream_t s) {
    set_stage_step(X, WKG * WT);
    const size_t lds_bytes = (size_t)WKG * 2 * WT * WROW * sizeof(bf16);
    static bool attr_set[16] = {};
    {
        int dev0 = 0;
        hipGetDevice(&dev0);
        if (dev0 < 0 || dev0 >= 16) return SWIN_ERR_UNSUPPORTED;
        if (!attr_set[dev0]) {
            if (hipFuncSetAttribute((const void*)wgrad_kernel<XLoader, WKG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds_bytes) != hipSuccess)
                return SWIN_ERR_LAUNCH;
            attr_set[dev0] = true;
        }
    }
    int g1 = (N1 + WN - 1) / WN, g2 = (N2 + WN - 1) / WN;
    // enough splits to fill the chip with 8 waves per CU (2 blocks of 4 waves or 1 block of 8), each split a multiple
    // of WKG t-stages
    int64_t stages = (T + WT - 1) / WT;
    // blocks to aim for: 512 wave-quads for the two-k-group form (256 blocks of 8 waves); 384 for the many-tile form,
    // of the Linear layers, where fewer splits (less atomic traffic) outweigh the fuller chip (47 -> 42 us at 1536x384;
    // the conv form, 36 big-K tiles, prefers the 512: 303 vs 340 us at P2)
    static const int forced = swin_dev_int("SWIN_WGRAD_BLOCKS", 0);      // development sweep
    // ConvX (36 tiles): swept again with the overlapped loads -- 640 blocks for the long contractions (P2 291 -> 275 us, mask
    // head convs 112 -> 104 us), 384 for the short ones (T < 24000: P4 42 -> 40 us)
    const int many = XLoader::kBlocksManyTiles == 512 ? (T >= 24000 ? 640 : 384) : XLoader::kBlocksManyTiles;
    const int target = forced > 0 ? forced : (WKG == 1 ? many : 512);
    int splits = (int)((target / WKG + (int64_t)g1 * g2 - 1) / ((int64_t)g1 * g2));
    int64_t max_splits = (stages + WKG - 1) / WKG;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
    if (splits > 65535) splits = 65535;
    static const int xcd_map = swin_dev_int("SWIN_WGRAD_XCD", 1);           // development A/B
    // the XCD-aware map below keeps a split's tiles on one XCD: it needs the splits to spread evenly over the 8 XCDs
    // measured (profiles/r02_wgrad_modes.txt): +7 % on the P2 conv weight gradient (36 tiles x 16 splits), neutral to negative on the
    // Linear shapes (few tiles or short t ranges: there balance over the CUs matters more than L2 reuse) -> conv, long t only
    bool use_xcd = xcd_map == 1 && XLoader::kBlocksManyTiles == 512 && T >= 24000 && g1 * g2 > 1 && splits >= 8;
    if (use_xcd) {
        int s8 = (splits + 4) / 8 * 8;
        if (s8 < 8) s8 = 8;
        if (s8 > max_splits) s8 = (int)(max_splits / 8 * 8);
        if (s8 >= 8) splits = s8; else use_xcd = false;
    }
    int64_t per = ((stages + splits - 1) / splits);
    per = ((per + WKG - 1) / WKG) * WKG * WT;
    splits = (int)((T + per - 1) / per);
    dim3 grid(g2, g1, splits);
    int dev = 0;
    hipGetDevice(&dev);
    const int mode = 1;            // contiguous fp32 atomics (mode 0, plain read-modify-write for one-split shapes, measured slower: 88 vs 80 us)
    if (use_xcd && splits % 8 == 0) {
        const unsigned nblk = 8u * (unsigned)(g1 * g2) * (unsigned)((splits + 7) / 8);
        wgrad_kernel<XLoader, WKG><<<nblk, 256 * WKG, lds_bytes, s>>>(dy, X, dw, dbias, T, N1, N2, per, mode, g2, g1, splits);
    } else {
        wgrad_kernel<XLoader, WKG><<<grid, 256 * WKG, lds_bytes, s>>>(dy, X, dw, dbias, T, N1, N2, per, mode, 0, g1, splits);
    }
    return swin_launch_status();
}

template <typename XLoader>
static int wgrad_launch(const bf16* dy, XLoader X, float* dw, float* dbias, int64_t T, int N1, int N2, hipStream_t s) {
    // few output tiles and a long t axis: the split-T atomics dominate -> two k-groups per block (half the atomic bytes);
    // many tiles: two independent 4-wave blocks per CU overlap each other's barriers better
    const int tiles = ((N1 + WN - 1) / WN) * ((N2 + WN - 1) / WN);
    static const int kg2_tiles = swin_dev_int("SWIN_WGRAD_KG2_TILES", 16);      // development sweep
    if (tiles <= kg2_tiles) return wgrad_launch_kg<XLoader, 2>(dy, X, dw, dbias, T, N1, N2, s);
    return wgrad_launch_kg<XLoader, 1>(dy, X, dw, dbias, T, N1, N2, s);
}

// dw (N1, N2) f32 += dy(T, N1)^T x(T, N2);  dbias (N1) f32 += column sums of dy (NULL to skip).
// N1 % 8 == 0, N2 % 8 == 0.  ACCUMULATES (caller zeroes).
// csrc/wgrad_dma.hip: the LDS-DMA ring kernel (round 3); SWIN_ERR_UNSUPPORTED = not its shape
int wgrad2_linear(const void* dy, const void* x, float* dw, float* dbias, int64_t T, int N1, int N2, void* stream);
int wgrad2_conv3x3(const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W, int Cin, int Cout, void* stream);

extern "C" int wgrad_linear_bf16(const void* dy, const void* x, float* dw, float* dbias, int64_t T, int N1, int N2,
                                 void* stream) {
    if (!dy || !x || !dw || T <= 0 || N1 <= 0 || N2 <= 0) return SWIN_ERR_BAD_ARG;
    if (N1 % 8 || N2 % 8) return SWIN_ERR_UNSUPPORTED;
    const int gen = swin_dev_int("SWIN_WGRAD_GEN", 2);          // development A/B: 1 = the register-staged kernel below
    if (gen == 2) {
        const int st = wgrad2_linear(dy, x, dw, dbias, T, N1, N2, stream);
        if (st != SWIN_ERR_UNSUPPORTED) return st;
    }
    PlainX X{(const bf16*)x, T, N2};
    return wgrad_launch((const bf16*)dy, X, dw, dbias, T, N1, N2, (hipStream_t)stream);
}

// dw (Cout, 3, 3, Cin) f32 += conv-weight gradient; dy (N,H,W,Cout), x (N,H,W,Cin) bf16 channels-last.
extern "C" int wgrad_conv3x3_nhwc_bf16(const void* dy, const void* x, float* dw, float* dbias, int N, int H, int W, int Cin,
                                       int Cout, void* stream) {
    if (!dy || !x || !dw || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SWIN_ERR_BAD_ARG;
    if (Cin % 8 || Cout % 8) return SWIN_ERR_UNSUPPORTED;
    const int gen = swin_dev_int("SWIN_WGRAD_GEN", 2);
    if (gen == 2) {
        const int st = wgrad2_conv3x3(dy, x, dw, dbias, N, H, W, Cin, Cout, stream);
        if (st != SWIN_ERR_UNSUPPORTED) return st;
    }
    int64_t T = (int64_t)N * H * W;
    ConvX X{(const bf16*)x, T, H, W, Cin, 0, 0};
    return wgrad_launch((const bf16*)dy, X, dw, dbias, T, Cout, 9 * Cin, (hipStream_t)stream);
}
