// bf16 MFMA GEMM core for gfx950 with two A-operand loaders:
//   * implicit-GEMM 3x3 / pad 1 / stride 1 convolution over a channels-last (N,H,W,Cin) activation
//     (FPN output convs fpn.py:195-197, RPN conv rpn_head.py:43, FCN mask head convs fcn_mask_head.py:119-121)
//   * plain row-major A (token-major Linear layers).
// C[m][n] = sum_k A[m][k] * Wt[n][k] (+ bias[n]) (ReLU optional);  Wt is K-contiguous: for the conv it
// is the (Cout, ky, kx, Cin) weight, i.e. the channels_last memory of the (Cout,Cin,3,3) parameter.
//
// Tile 128(m) x 128(n) x 64(k), 256 threads = 2x2 waves of 64x64, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// LDS: one or two K-tiles (A 16 KB + W 16 KB each; see gemm_launch for the choice) filled by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no
// ds_write traffic -- register staging left the kernel bound by the VGPR->LDS write path at ~500 TFLOP/s); a wave
// instruction writes 1 KiB = 8 rows x 128 B linearly, so the 16-byte pieces are XOR-swizzled on the SOURCE
// address (slot = piece ^ ((row >> 1) & 7)) and the same XOR is applied by the ds_read_b128 fragment reads, which
// makes them bank-conflict free; the DMA of tile k+1 is in flight while tile k feeds the MFMAs (one barrier per
// K-tile); zero padding (conv borders, M/N tails) is read from a 16-byte zero buffer.
// The weight tile is the MFMA "A" operand and the pixel tile the "B" operand, so a lane owns one pixel
// and 4 consecutive output channels per accumulator group -> 8-byte stores into the NHWC output row.
#include <type_traits>

#include "common.h"
#include <cstdlib>

#define BM 128
#define BN 128
#define BK 64

struct ConvGeom { int N, H, W, Cin; };

// ---- A-operand loaders: element offset of 16-byte piece `piece` (0..7) of row `m` for K-tile `kt`, or -1 when
// the piece is zero padding.  (The load itself is done by the kernel from a CLAMPED address and masked by value:
// selecting between a global pointer and a local zero makes hipcc emit flat loads with a full wait after each.)
struct ConvA {
    const bf16* a; int64_t M; ConvGeom g; int cpt;   // cpt = Cin / BK (K-tiles per filter tap)
    __device__ __forceinline__ void prep(int64_t m, int64_t& base, int& y, int& x) const {
        if (m >= M) { base = -1; y = x = 0; return; }
        x = (int)(m % g.W); int64_t t = m / g.W; y = (int)(t % g.H);
        base = m * g.Cin;                              // pixel (n,y,x) itself
    }
    __device__ __forceinline__ int64_t offset(int64_t base, int y, int x, int kt, int piece) const {
        int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
        int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        int yy = y + dy, xx = x + dx;
        bool ok = base >= 0 && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        return ok ? base + ((int64_t)dy * g.W + dx) * g.Cin + c0 + piece * 8 : -1;
    }
    // bit t set: filter tap t of this pixel lies inside the image (border predication, once per row and lane)
    __device__ __forceinline__ unsigned tapmask(int64_t base, int y, int x) const {
        if (base < 0) return 0u;
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) m |= 1u << t;
        }
        return m;
    }
    // K-tile cursor for loops that walk the K-tiles in order: no division per tile (one wave per SIMD pays every scalar instruction)
    struct Cur { int tap, c, ty, tx; };
    __device__ __forceinline__ void seek(int kt, Cur& q) const { q.tap = kt / cpt; q.c = kt - q.tap * cpt; q.ty = q.tap / 3; q.tx = q.tap - 3 * q.ty; }
    __device__ __forceinline__ int64_t koff(const Cur& q) const { return ((int64_t)(q.ty - 1) * g.W + (q.tx - 1)) * g.Cin + q.c * BK; }
    __device__ __forceinline__ int ktile(const Cur& q) const { return q.tap * cpt + q.c; }
    __device__ __forceinline__ void next(Cur& q) const {
        if (++q.c == cpt) { q.c = 0; ++q.tap; if (++q.tx == 3) { q.tx = 0; ++q.ty; } }
    }
    __device__ __forceinline__ void kinfo(int kt, int64_t& koff, int& tap) const {
        tap = kt / cpt;
        const int c0 = (kt - tap * cpt) * BK, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        koff = ((int64_t)dy * g.W + dx) * g.Cin + c0;
    }
};

// Plain row-major A (M, K): the token-major activations of the Linear layers (swin_transformer.py:33,36,129,151,296) -- the same
// kernel as a GEMM  C = A Wt^T.  One "tap" that is always inside.
struct PlainA {
    const bf16* a; int64_t M; int K;
    __device__ __forceinline__ void prep(int64_t m, int64_t& base, int& y, int& x) const { y = x = 0; base = m < M ? m * K : -1; }
    __device__ __forceinline__ unsigned tapmask(int64_t base, int, int) const { return base < 0 ? 0u : 1u; }
    struct Cur { int tap, c; };
    __device__ __forceinline__ void seek(int kt, Cur& q) const { q.tap = 0; q.c = kt; }
    __device__ __forceinline__ int64_t koff(const Cur& q) const { return (int64_t)q.c * BK; }
    __device__ __forceinline__ int ktile(const Cur& q) const { return q.c; }
    __device__ __forceinline__ void next(Cur& q) const { ++q.c; }
};

__device__ __forceinline__ int swz(int row, int piece) { return piece ^ ((row >> 1) & 7); }

__device__ uint4 g_zero16[4];      // zero-initialised: source of padded 16-byte pieces

// LDS-DMA as inline asm (M0 = wave-uniform LDS byte address, saved and restored): the BUILTIN form makes hipcc drain the DMA with
// s_waitcnt vmcnt(0) in front of the next ds_read of the loop (it cannot tell which LDS bytes the DMA writes), which is exactly the
// wait the counted schedules below exist to avoid.  An asm DMA is invisible to that bookkeeping; its completion is waited for by hand.
__device__ __forceinline__ void glds16(uint64_t gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
// (Round 2: the 128-row kernel used the builtin until its ring of four buffers was found drained by such a compiler-inserted
// vmcnt(0) at every K-tile -- the 1 us per K-tile floor of the small maps.)

// WM = wave rows of the block: tile (64 WM) x 128 x 64 with 2 WM waves.  WM = 2 is the 128x128 tile (two blocks per
// CU); WM = 4 doubles the pixel rows per weight tile (one 8-wave block per CU): 48 KB instead of 64 KB of operand
// traffic per 4.2 MFLOP, for the large maps where the L2 -> LDS stream is the limit.
// EPI (epilogue): 0 = bias (+ ReLU, + gate);  1 = the fc1 of Mlp (swin_transformer.py:33-34): C = the product without bias (the
// pre-activation the backward reads), C2 = gelu_erf(product + bias);  2 = the data gradient through that GELU: C = product *
// gelu'(gate + bias), `gate` here being the saved pre-activation (no separate pass over the T x 4C tensors either way).  The
// product is rounded to bf16 BEFORE the activation, as in the reference under autocast (the Linear's output is a half tensor) and
// as the three-launch chain GEMM -> bias_gelu does: the two paths give the same bits.
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float v) {
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    return cdf + v * 0.39894228040143267794f * __expf(-0.5f * v * v);
}

template <typename ALoader, bool RELU, int WM, int NBUF, int EPI = 0>
__global__ __launch_bounds__(128 * WM, WM == 2 ? (NBUF == 1 ? 4 : (NBUF == 2 ? 2 : 1)) : 1) void gemm_bf16_kernel(ALoader A, const bf16* __restrict__ Wt,
                                                                              const float* __restrict__ bias, bf16* __restrict__ C,
                                                                              int64_t M, int Nn, int K, int mtiles, int ntiles,
                                                                              const bf16* __restrict__ gate, float* __restrict__ part, int ksplit,
                                                                              bf16* __restrict__ C2 = nullptr) {
    constexpr int TM = 64 * WM;                                          // tile rows (pixels)
    extern __shared__ __attribute__((aligned(16))) uint4 lds_raw[];      // [buf][A: TM*8 | W: BN*8]
    auto ldsA = [&](int buf) { return lds_raw + (size_t)buf * (TM + BN) * 8; };
    auto ldsW = [&](int buf) { return lds_raw + (size_t)buf * (TM + BN) * 8 + TM * 8; };
    // XCD-aware tile order: blocks that share an XCD (id % 8) get a contiguous run of tiles, n fastest
    const int nblk = mtiles * ntiles;
    int id = blockIdx.x;
    {
        int q = nblk / 8, r = nblk % 8, xcd = id % 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
    }
    const int mt_ = id / ntiles, nt_ = id - mt_ * ntiles;
    const int64_t m0 = (int64_t)mt_ * TM;
    const int n0 = nt_ * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c = lane & 31, h = lane >> 5;

    // DMA role: wave w fills rows [32w, 32w+32) of the A tile and rows [WROWS w, WROWS (w+1)) of the W tile,
    // 8 rows x 8 slots per instruction
    constexpr int WROWS = BN / (2 * WM), WI = WROWS / 8;                 // W rows / instructions per wave
    const int drow = 32 * wave + (lane >> 3), dslot = lane & 7;
    const int wrow = WROWS * wave + (lane >> 3);
    // per-lane constants of the DMA sources (computed once): byte address of this lane's 16-byte piece at K-tile 0 and
    // the border mask of its pixel row.  Per K-tile only a wave-uniform offset is added and the pointer is selected
    // against the zero line by value (v_cndmask) -- the former per-tile `offset()` + pointer select cost ~60
    // instructions and two branches per piece, which one block per CU (small maps) could not hide.
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const uint64_t zero64 = (uint64_t)reinterpret_cast<uintptr_t>(g_zero16);
    uint64_t aptr[4]; unsigned amask[4]; uint64_t wptr[WI]; bool wok[WI];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = drow + 8 * i;
        const int piece = dslot ^ ((row >> 1) & 7);                // slot s of LDS row holds piece s ^ f(row)
        int64_t base; int y, x;
        A.prep(m0 + row, base, y, x);
        amask[i] = A.tapmask(base, y, x);
        aptr[i] = (uint64_t)reinterpret_cast<uintptr_t>(A.a + (base < 0 ? 0 : base) + piece * 8);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = wrow + 8 * i;
        const int piece = dslot ^ ((row >> 1) & 7);
        const int n = n0 + row;
        wok[i] = n < Nn;
        wptr[i] = (uint64_t)reinterpret_cast<uintptr_t>(Wt + (wok[i] ? (int64_t)n * K : 0) + piece * 8);
    }
    const int nk = K / BK;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);          // LDS destinations of the DMA live in SGPRs (M0)
    typename ALoader::Cur cur_k;                                   // the next K-tile to stage (wave-uniform)
    auto dma_tile = [&](int buf) {                                 // stages K-tile cur_k into LDS buffer buf and advances the cursor
        const int tap = cur_k.tap;
        const uint64_t abytes = (uint64_t)(A.koff(cur_k) * 2), wbytes = (uint64_t)A.ktile(cur_k) * BK * 2;
        A.next(cur_k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t src = ((amask[i] >> tap) & 1u) ? aptr[i] + abytes : zero64;
            glds16(src, (unsigned)(uintptr_t)(lptr_t)(ldsA(buf) + (32 * wave_u + 8 * i) * 8));
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const uint64_t src = wok[i] ? wptr[i] + wbytes : zero64;
            glds16(src, (unsigned)(uintptr_t)(lptr_t)(ldsW(buf) + (WROWS * wave_u + 8 * i) * 8));
        }
    };
    // two MFMA forms: 32x32x16 (2x2 accumulator tiles per wave) and 16x16x32 (4x4 tiles, same 64 registers); the
    // second delivers more FLOP/s per cycle when every operand is re-read from LDS (guide section 3) -> the many-tile path
    constexpr bool MF16 = (NBUF == 1);
    f32x16 acc[2][2];
    f32x4 acc16[4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc16[a][b] = f32x4{0};
    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const uint4* As = ldsA(buf);
        const uint4* Ws = ldsW(buf);
        if constexpr (MF16) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {                 // 32 k per step: lane (fr, fq) holds row fr, k = 32 s + 8 fq .. +8
                bf16x8 wf[4], af[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int rw_ = wn * 64 + 16 * t + fr, ra_ = wm * 64 + 16 * t + fr;
                    uint4 u = Ws[rw_ * 8 + swz(rw_, 4 * s + fq)];
                    uint4 v = As[ra_ * 8 + swz(ra_, 4 * s + fq)];
                    wf[t] = *(bf16x8*)&u;
                    af[t] = *(bf16x8*)&v;
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        acc16[nt][mt] = SWIN_MFMA_16x16x32(wf[nt], af[mt], acc16[nt][mt], 0, 0, 0);
            }
        } else {
            // two fragment sets: the reads of k-step s+1 are in flight under the MFMAs of step s (these variants run one or two
            // blocks per CU -- nothing else hides the LDS latency; with one set a K-tile took ~0.7 us of which 0.2 us MFMA)
            bf16x8 wf[2][2], af[2][2];
            auto frag = [&](int s, int b) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    int rw_ = wn * 64 + 32 * t + c, ra_ = wm * 64 + 32 * t + c;
                    uint4 u = Ws[rw_ * 8 + swz(rw_, 2 * s + h)];
                    uint4 v = As[ra_ * 8 + swz(ra_, 2 * s + h)];
                    wf[b][t] = *(bf16x8*)&u;
                    af[b][t] = *(bf16x8*)&v;
                }
            };
            frag(0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) frag(s + 1, (s + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);         // (left alone, the scheduler folds the two sets back into one and serialises)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = SWIN_MFMA_32x32x16(wf[s & 1][nt], af[s & 1][mt], acc[nt][mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if constexpr (NBUF == 4) {
        // ring of 4 LDS buffers, 3 K-tiles of LDS-DMA in flight, ONE raw barrier per K-tile, counted vmcnt: for shapes
        // with at most one block per CU (small feature maps), where nothing else on the CU hides the DMA latency and a
        // 36-step K loop costs 36 memory latencies with the two-buffer schedule.  Each wave issues DPT DMA
        // instructions per tile, in order; tile kt has landed for this wave when at most DPT x (tiles issued after it)
        // are outstanding; the barrier then publishes it to the other waves (read AFTER the barrier that follows the
        // wait), and -- every wave having finished compute(kt-1) before arriving -- frees buffer (kt-1) % 4 for
        // tile kt+3.  No __syncthreads() here: with a DMA in flight its fence would drain the ring (guide section 5).
        constexpr int DPT = 4 + WI;
        static_assert(DPT == 8, "vmcnt literals below assume 8 DMA instructions per tile and wave");
        // split K (small maps, gemm_launch): blockIdx.y owns K-tiles [kb, kb + nkl) and stores its fp32 partial tile in `part`;
        // splitk_finish_kernel adds the slabs and applies bias / ReLU / gate.  The 36-tile chain of a 3x3 conv costs ~1.1 us per
        // tile however few tiles the map has (DMA latency / 3 tiles in flight): P4..P6 all took 40 us on 126 / 32 / 10 CUs.
        int kb = 0, nkl = nk;
        if (ksplit > 1) {
            kb = (int)((int64_t)nk * blockIdx.y / ksplit);
            nkl = (int)((int64_t)nk * (blockIdx.y + 1) / ksplit) - kb;
        }
        A.seek(kb, cur_k);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < nkl) dma_tile(i);
        for (int kt = 0; kt < nkl; ++kt) {
            if (kt + 2 < nkl) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (kt + 1 < nkl) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 3 < nkl) dma_tile((kt + 3) & 3);
            compute(kt & 3);
        }
    } else if constexpr (NBUF == 1) {
        // one LDS buffer, two barriers per K-tile: 32 KB per block -> 4 blocks per CU overlap each other's DMA and MFMA
        A.seek(0, cur_k);
        for (int kt = 0; kt < nk; ++kt) {
            dma_tile(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    } else {
        A.seek(0, cur_k);
        dma_tile(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) dma_tile(cur ^ 1);               // lands while this tile feeds the MFMAs
            compute(cur);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile kt+1 have landed
            __syncthreads();                                   // ... and everyone's; all reads of tile kt are done
            cur ^= 1;
        }
    }

    // epilogue: lane = pixel m, registers = output channels
    if constexpr (MF16) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int64_t m = m0 + wm * 64 + 16 * mt + fr;
            if (m >= M) continue;
            bf16* crow = C + m * Nn;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                int n = n0 + wn * 64 + 16 * nt + 4 * fq;            // C layout: row (= n) = 4 fq + j, column (= m) = fr
                if (n >= Nn) continue;
                bf16x4 o;
                if constexpr (EPI == 1) {
                    bf16x4 o2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (bf16)acc16[nt][mt][e]; o2[e] = (bf16)gelu_erf((float)o[e] + bias[n + e]); }
                    *(bf16x4*)(C2 + m * Nn + n) = o2;
                } else if constexpr (EPI == 2) {
                    const bf16x4 hp = *(const bf16x4*)(gate + m * Nn + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)((float)(bf16)acc16[nt][mt][e] * gelu_erf_grad((float)hp[e] + bias[n + e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc16[nt][mt][e] + (bias ? bias[n + e] : 0.f);
                        if (RELU) v = fmaxf(v, 0.f);
                        o[e] = (bf16)v;
                    }
                    if (gate) {                       // ReLU backward of the layer below: zero where its output was not positive
                        const bf16x4 gt = *(const bf16x4*)(gate + m * Nn + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (!((float)gt[e] > 0.f)) o[e] = (bf16)0.f;
                    }
                }
                *(bf16x4*)(crow + n) = o;
            }
        }
        return;
    }
    if constexpr (NBUF == 4) {
        if (ksplit > 1) {                          // fp32 partial tile of this K range, plain stores: slab [split][m][n]
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                int64_t m = m0 + wm * 64 + 32 * mt + c;
                if (m >= M) continue;
                float* prow = part + ((int64_t)blockIdx.y * M + m) * Nn;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        int n = n0 + wn * 64 + 32 * nt + 8 * gq + 4 * h;
                        if (n >= Nn) continue;
                        *(f32x4*)(prow + n) = f32x4{acc[nt][mt][4 * gq], acc[nt][mt][4 * gq + 1], acc[nt][mt][4 * gq + 2], acc[nt][mt][4 * gq + 3]};
                    }
            }
            return;
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int64_t m = m0 + wm * 64 + 32 * mt + c;
        if (m >= M) continue;
        bf16* crow = C + m * Nn;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                int n = n0 + wn * 64 + 32 * nt + 8 * gq + 4 * h;
                if (n >= Nn) continue;
                bf16x4 o;
                if constexpr (EPI == 1) {
                    bf16x4 o2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = (bf16)acc[nt][mt][4 * gq + e];
                        o2[e] = (bf16)gelu_erf((float)o[e] + bias[n + e]);
                    }
                    *(bf16x4*)(C2 + m * Nn + n) = o2;
                } else if constexpr (EPI == 2) {
                    const bf16x4 hp = *(const bf16x4*)(gate + m * Nn + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)((float)(bf16)acc[nt][mt][4 * gq + e] * gelu_erf_grad((float)hp[e] + bias[n + e]));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[nt][mt][4 * gq + e] + (bias ? bias[n + e] : 0.f);
                        if (RELU) v = fmaxf(v, 0.f);
                        o[e] = (bf16)v;
                    }
                    if (gate) {
                        const bf16x4 gt = *(const bf16x4*)(gate + m * Nn + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (!((float)gt[e] > 0.f)) o[e] = (bf16)0.f;
                    }
                }
                *(bf16x4*)(crow + n) = o;
            }
    }
}

// Second half of a split-K launch: C = bf16(relu?(sum over slabs + bias)), zeroed where gate is not positive.  One thread per four
// consecutive output channels of a pixel (Nn % 4 == 0).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int ksplit, int64_t MN, int Nn,
                                                            const float* __restrict__ bias, int relu, const bf16* __restrict__ gate,
                                                            bf16* __restrict__ C) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    f32x4 v = *(const f32x4*)(part + i);
    for (int s = 1; s < ksplit; ++s) {
        const f32x4 u = *(const f32x4*)(part + (int64_t)s * MN + i);
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    const int n = (int)(i % Nn);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e] + (bias ? bias[n + e] : 0.f);
        if (relu) x = fmaxf(x, 0.f);
        o[e] = (bf16)x;
    }
    if (gate) {
        const bf16x4 gt = *(const bf16x4*)(gate + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!((float)gt[e] > 0.f)) o[e] = (bf16)0.f;
    }
    *(bf16x4*)(C + i) = o;
}

// K split of the few-tile shapes: as many K ranges as fill the chip once (256 CUs, one 128 KB block each), at least 3 K-tiles per
// range.  1 = no split.  (P5: 32 tiles -> 8, P6: 10 -> 12.)  Kernel trace, 2 x 800 x 1280 batch: P5 40 -> 16.6 + 4.3 us (finish),
// P6 40 -> 11.3 + 4.3 us.  P4 (126 tiles) with two ranges: 41 -> 31.4 + 5.6 us and 32 MB of slab traffic -- not worth a second
// launch, hence the 64-tile limit.  A range costs ~9 us whatever its length (launch, per-lane border masks, first DMA latency).
static int conv_ksplit(int64_t M, int Nn, int K) {
    static const int off = swin_dev_int("SWIN_CONV_KSPLIT", -1);    // 0 / 1: never split; n > 1: forced
    const int64_t blocks = ((M + 127) / 128) * ((Nn + BN - 1) / BN);
    const int nk = K / BK;
    if (off == 0 || off == 1 || blocks > (off > 1 ? 128 : 64) || Nn % 4 != 0) return 1;
    int s = off > 1 ? off : (int)(256 / blocks);
    if (s > nk / 3) s = nk / 3;
    return s < 2 ? 1 : s;
}

template <typename ALoader, bool RELU, int WM, int NBUF>
static int gemm_launch_wm(ALoader A, const bf16* Wt, const float* bias, bf16* C, int64_t M, int Nn, int K, hipStream_t s, const bf16* gate,
                          float* part = nullptr, int ksplit = 1, int finish_relu = 0) {
    constexpr int TM = 64 * WM;
    const size_t lds_bytes = NBUF * (size_t)(TM + BN) * 8 * sizeof(uint4);
    static bool attr_set[16] = {};                           // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)gemm_bf16_kernel<ALoader, RELU, WM, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    int mtiles = (int)((M + TM - 1) / TM), ntiles = (Nn + BN - 1) / BN;
    if (ksplit > 1) {                                // (always the RELU = false instantiation: bias / ReLU / gate belong to the finish)
        gemm_bf16_kernel<ALoader, RELU, WM, NBUF><<<dim3(mtiles * ntiles, ksplit), 128 * WM, lds_bytes, s>>>(A, Wt, nullptr, C, M, Nn, K, mtiles,
                                                                                                          ntiles, nullptr, part, ksplit);
        if (int st = swin_launch_status()) return st;
        const int64_t MN = M * Nn;
        splitk_finish_kernel<<<(unsigned)((MN / 4 + 255) / 256), 256, 0, s>>>(part, ksplit, MN, Nn, bias, finish_relu, gate, C);
        return swin_launch_status();
    }
    gemm_bf16_kernel<ALoader, RELU, WM, NBUF><<<mtiles * ntiles, 128 * WM, lds_bytes, s>>>(A, Wt, bias, C, M, Nn, K, mtiles, ntiles, gate, nullptr, 1);
    return swin_launch_status();
}

static int g_conv_wm = 0;     // 0: choose by size; 2 / 4: forced (A/B experiments through SWIN_CONV_WM)

template <typename ALoader>
static int gemm_launch(ALoader A, const bf16* Wt, const float* bias, bf16* C, int64_t M, int Nn, int K, int relu, hipStream_t s,
                       const bf16* gate = nullptr, float* part = nullptr, int64_t part_bytes = 0) {
    static bool env_read = false;
    if (!env_read) { g_conv_wm = swin_dev_int("SWIN_CONV_WM", 0); env_read = true; }
    // (A 256 x 256 four-phase kernel with hand-placed LDS-DMA existed in round 2: 185 us against 192 us at P2 in isolation, no gain
    // inside the training step -- DESIGN section 9.2; removed in round 3, git history has it.)
    // large pixel counts: the 256-row tile (one 8-wave block per CU); otherwise the 128-row tile keeps the grid full
    const int64_t blocks128 = ((M + 127) / 128) * ((Nn + BN - 1) / BN);
    // the 256-row tile (8 waves, one block per CU, two buffers): for maps of 257..512 128-row tiles (P3 of the bench batch: 250 blocks
    // of 256 rows = one per CU): 56.6 us against 61-66 alone, 72 against 87 us inside the step (kernel trace); SWIN_CONV_MID=0
    // turns it off.  Everywhere else it measured no better than the 128-row tile.
    static const int mid = swin_dev_int("SWIN_CONV_MID", 1);
    const bool big = g_conv_wm == 4 || (g_conv_wm == 0 && mid && blocks128 > 256 && blocks128 <= 512);
    if (big) {
        if (relu) return gemm_launch_wm<ALoader, true, 4, 2>(A, Wt, bias, C, M, Nn, K, s, gate);
        return gemm_launch_wm<ALoader, false, 4, 2>(A, Wt, bias, C, M, Nn, K, s, gate);
    }
    // enough tiles for several blocks per CU: ONE LDS buffer (32 KB, two barriers per K-tile) at 4 blocks per CU -- the
    // co-resident blocks overlap each other's DMA and MFMA phases better than a block's own double buffer does
    // (P2 map: 535 -> 663 TFLOP/s, mask-head convs: 399 -> 593); few tiles: the double-buffered block hides more itself
    if (g_conv_wm == 1 || (g_conv_wm == 0 && blocks128 > 512)) {       // (SWIN_CONV_WM=2 forces the two-buffer variant below)
        if (relu) return gemm_launch_wm<ALoader, true, 2, 1>(A, Wt, bias, C, M, Nn, K, s, gate);
        return gemm_launch_wm<ALoader, false, 2, 1>(A, Wt, bias, C, M, Nn, K, s, gate);
    }
    if (g_conv_wm == 3 || (g_conv_wm == 0 && blocks128 <= 256)) {      // at most one block per CU: 4-buffer DMA ring
        if constexpr (std::is_same<ALoader, ConvA>::value) {
            const int ks = part ? conv_ksplit(M, Nn, K) : 1;
            if (ks > 1) {
                if (part_bytes < (int64_t)ks * M * Nn * (int64_t)sizeof(float)) return SWIN_ERR_BAD_ARG;
                return gemm_launch_wm<ALoader, false, 2, 4>(A, Wt, bias, C, M, Nn, K, s, gate, part, ks, relu);
            }
        }
        if (relu) return gemm_launch_wm<ALoader, true, 2, 4>(A, Wt, bias, C, M, Nn, K, s, gate);
        return gemm_launch_wm<ALoader, false, 2, 4>(A, Wt, bias, C, M, Nn, K, s, gate);
    }
    if (relu) return gemm_launch_wm<ALoader, true, 2, 2>(A, Wt, bias, C, M, Nn, K, s, gate);
    return gemm_launch_wm<ALoader, false, 2, 2>(A, Wt, bias, C, M, Nn, K, s, gate);
}

// The halo-staged form (csrc/conv_halo.hip) for the many-pixel maps; SWIN_ERR_UNSUPPORTED = not this shape, use the implicit GEMM.
static int conv_try_halo(const void* x, const void* w, const float* bias, const void* gate, void* y, int N, int H, int W, int Cin, int Cout,
                         int relu, hipStream_t s) {
    // from 400 tiles of 128 x 128 (P3 of the bench batch: 500; P4: 128 stays on the split-K implicit GEMM, where the two forms tie); 0 = never
    static const int min128 = swin_dev_int("SWIN_CONV_HALO_MIN", 400);
    const int64_t M = (int64_t)N * H * W;
    const int64_t blocks128 = ((M + 127) / 128) * ((Cout + BN - 1) / BN);
    if (min128 <= 0 || blocks128 < min128 || Cout % 128 != 0) return SWIN_ERR_UNSUPPORTED;
    static const int nt = swin_dev_int("SWIN_CONV_HALO_NT", 0);
    return swin_conv_halo((const bf16*)x, (const bf16*)w, bias, (const bf16*)gate, (bf16*)y, N, H, W, Cin, Cout, relu, nt, s);
}

// x (N,H,W,Cin) bf16 channels-last; w (Cout,3,3,Cin) bf16; bias (Cout) f32 or NULL; y (N,H,W,Cout) bf16.
extern "C" int conv3x3_nhwc_bf16(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin,
                                 int Cout, int relu, void* stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SWIN_ERR_BAD_ARG;
    if (Cin % BK != 0 || Cout % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    if (int st = conv_try_halo(x, w, bias, nullptr, y, N, H, W, Cin, Cout, relu, (hipStream_t)stream); st != SWIN_ERR_UNSUPPORTED) return st;
    int64_t M = (int64_t)N * H * W;
    ConvA A{(const bf16*)x, M, ConvGeom{N, H, W, Cin}, Cin / BK};
    return gemm_launch(A, (const bf16*)w, bias, (bf16*)y, M, Cout, 9 * Cin, relu, (hipStream_t)stream);
}

// Few-tile maps (the coarse pyramid levels): bytes of fp32 workspace with which conv3x3_nhwc_bf16_ws splits the contraction over
// blockIdx.y (0: this shape is not split -- use the plain entry points).
extern "C" int64_t conv3x3_splitk_workspace_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % BK != 0 || Cout % 4 != 0) return 0;
    const int64_t M = (int64_t)N * H * W;
    const int ks = conv_ksplit(M, Cout, 9 * Cin);
    return ks > 1 ? (int64_t)ks * M * Cout * (int64_t)sizeof(float) : 0;
}

// conv3x3_nhwc_bf16 (gate == NULL) / conv3x3_nhwc_bf16_gated (gate != NULL, relu ignored) with a caller-provided workspace of
// conv3x3_splitk_workspace_bytes(...) bytes; the result equals the unsplit kernel's up to the fp32 summation order.
extern "C" int conv3x3_nhwc_bf16_ws(const void* x, const void* w, const float* bias, const void* gate, void* y, int N, int H, int W,
                                    int Cin, int Cout, int relu, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (workspace && workspace_bytes <= 0)) return SWIN_ERR_BAD_ARG;
    if (Cin % BK != 0 || Cout % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    if (int st = conv_try_halo(x, w, bias, gate, y, N, H, W, Cin, Cout, gate ? 0 : relu, (hipStream_t)stream); st != SWIN_ERR_UNSUPPORTED) return st;
    int64_t M = (int64_t)N * H * W;
    ConvA A{(const bf16*)x, M, ConvGeom{N, H, W, Cin}, Cin / BK};
    return gemm_launch(A, (const bf16*)w, bias, (bf16*)y, M, Cout, 9 * Cin, gate ? 0 : relu, (hipStream_t)stream, (const bf16*)gate,
                       (float*)workspace, workspace_bytes);
}

// The same convolution with its output zeroed wherever gate (N,H,W,Cout) bf16 is not positive: used for data gradients,
// gate = the ReLU output that was this convolution's input in the forward pass, so the result is already the gradient
// at the ReLU's INPUT (torch: threshold_backward, one more pass over the map).
extern "C" int conv3x3_nhwc_bf16_gated(const void* x, const void* w, const float* bias, const void* gate, void* y, int N, int H,
                                       int W, int Cin, int Cout, void* stream) {
    if (!x || !w || !y || !gate || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SWIN_ERR_BAD_ARG;
    if (Cin % BK != 0 || Cout % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    if (int st = conv_try_halo(x, w, bias, gate, y, N, H, W, Cin, Cout, 0, (hipStream_t)stream); st != SWIN_ERR_UNSUPPORTED) return st;
    int64_t M = (int64_t)N * H * W;
    ConvA A{(const bf16*)x, M, ConvGeom{N, H, W, Cin}, Cin / BK};
    return gemm_launch(A, (const bf16*)w, bias, (bf16*)y, M, Cout, 9 * Cin, 0, (hipStream_t)stream, (const bf16*)gate);
}

// c (M,N) bf16 = a (M,K) bf16 x w (N,K)^T + bias (N) f32 [ReLU] on the same hand-written MFMA kernel (K % 64 == 0, N % 4 == 0):
// nn.Linear forward, and its data gradient when `w` is the transposed weight.
extern "C" int swin_linear_hip_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int relu,
                                    void* stream) {
    if (M == 0) return SWIN_OK;
    if (!a || !w || !c || M < 0 || N <= 0 || K <= 0) return SWIN_ERR_BAD_ARG;
    if (K % BK != 0 || N % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    PlainA A{(const bf16*)a, M, K};
    return gemm_launch(A, (const bf16*)w, bias, (bf16*)c, M, N, K, relu, (hipStream_t)stream);
}

// The two GELU-fused GEMMs of Mlp (EPI 1 / 2) for every width the token-stationary kernel of csrc/ts_mlp.hip does not cover:
// the same tile choice as gemm_launch, PlainA only.
template <int EPI, int WM, int NBUF>
static int gemm_launch_epi_wm(PlainA A, const bf16* Wt, const float* bias, bf16* C, bf16* C2, const bf16* aux, int64_t M, int Nn, int K,
                              hipStream_t s) {
    constexpr int TM = 64 * WM;
    const size_t lds_bytes = NBUF * (size_t)(TM + BN) * 8 * sizeof(uint4);
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)gemm_bf16_kernel<PlainA, false, WM, NBUF, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess)
            return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const int mtiles = (int)((M + TM - 1) / TM), ntiles = (Nn + BN - 1) / BN;
    gemm_bf16_kernel<PlainA, false, WM, NBUF, EPI><<<mtiles * ntiles, 128 * WM, lds_bytes, s>>>(A, Wt, bias, C, M, Nn, K, mtiles, ntiles, aux,
                                                                                          nullptr, 1, C2);
    return swin_launch_status();
}

template <int EPI>
static int gemm_launch_epi(PlainA A, const bf16* Wt, const float* bias, bf16* C, bf16* C2, const bf16* aux, int64_t M, int Nn, int K, hipStream_t s) {
    const int64_t blocks128 = ((M + 127) / 128) * ((Nn + BN - 1) / BN);
    if (blocks128 > 512) return gemm_launch_epi_wm<EPI, 2, 1>(A, Wt, bias, C, C2, aux, M, Nn, K, s);
    if (blocks128 <= 256) return gemm_launch_epi_wm<EPI, 2, 4>(A, Wt, bias, C, C2, aux, M, Nn, K, s);
    return gemm_launch_epi_wm<EPI, 2, 2>(A, Wt, bias, C, C2, aux, M, Nn, K, s);
}

// hpre (M,N) = a (M,K) w (N,K)^T (no bias);  h (M,N) = gelu_erf(that + bias).   K % 64 == 0, N % 4 == 0.
extern "C" int swin_linear_gelu_hip_bf16(const void* a, const void* w, const float* bias, void* hpre, void* h, int64_t M, int N, int K,
                                         void* stream) {
    if (M == 0) return SWIN_OK;
    if (!a || !w || !bias || !hpre || !h || M < 0 || N <= 0 || K <= 0) return SWIN_ERR_BAD_ARG;
    if (K % BK != 0 || N % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    PlainA A{(const bf16*)a, M, K};
    return gemm_launch_epi<1>(A, (const bf16*)w, bias, (bf16*)hpre, (bf16*)h, nullptr, M, N, K, (hipStream_t)stream);
}

// dhpre (M,N) = (dy (M,K) wt (N,K)^T) * gelu_erf'(hpre + bias): the data gradient through fc2 and the GELU in one launch; wt is the
// TRANSPOSED fc2 weight ((4C, C): linear_t_layout_multi).  K % 64 == 0, N % 4 == 0.
extern "C" int swin_linear_dgelu_hip_bf16(const void* dy, const void* wt, const void* hpre, const float* bias, void* dhpre, int64_t M,
                                          int N, int K, void* stream) {
    if (M == 0) return SWIN_OK;
    if (!dy || !wt || !hpre || !bias || !dhpre || M < 0 || N <= 0 || K <= 0) return SWIN_ERR_BAD_ARG;
    if (K % BK != 0 || N % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    PlainA A{(const bf16*)dy, M, K};
    return gemm_launch_epi<2>(A, (const bf16*)wt, bias, (bf16*)dhpre, nullptr, (const bf16*)hpre, M, N, K, (hipStream_t)stream);
}
