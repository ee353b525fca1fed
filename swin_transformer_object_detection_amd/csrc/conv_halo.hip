// 3x3 / pad 1 / stride 1 convolution over a channels-last bf16 activation, halo-staged form (round 3) -- the many-pixel maps of
// fpn.py:195-197 (FPN output convs), rpn_head.py:43 (RPN conv) and fcn_mask_head.py:119-121 (mask-head convs), forward and, with the
// re-laid-out weight, data gradient.
//
// Why a second kernel: the implicit-GEMM kernel (csrc/conv_gemm.hip) stages a 128-pixel x 64-channel A tile PER FILTER TAP, i.e.
// it brings every input pixel into LDS nine times per channel block, 32 KB of LDS-DMA per 2.1 MFLOP.  At P2 that stream runs at the
// ~11 TB/s the L2 / Infinity-Cache mix sustains chip-wide (21 B/clk/CU) with the MFMA pipe a third busy (831 TFLOP/s).  The nine
// taps of a pixel tile read the same input rows shifted by dy * W + dx pixels, so this kernel stages, once per 32-channel block, the
// THREE row segments a tile of 256 consecutive pixels needs (pixels p0 + dy W - 1 ... p0 + dy W + 256, dy = -1, 0, 1: 3 x 258 rows of
// 64 B = 48 KB) and then streams only the weight tiles (one 32-channel x TN slab per tap): 13.5 KB (TN = 128) or 21.5 KB (TN = 256)
// per 2.1 / 4.2 MFLOP -- 2.4x / 3x fewer bytes per flop.
//
//   block     8 waves = 4 (pixels: 64 each) x 2 (output channels: 32 NT each); tile 256 pixels x TN = 64 NT channels
//   K loop    channel blocks of 32 (outer) x 9 taps (inner, unrolled); per (block, tap) and wave 4 NT v_mfma_f32_32x32x16 (k = 32)
//   LDS       input image [2][3 segments][272 rows][64 B] (double-buffered over channel blocks: block cb + 1 is staged while cb is
//             consumed, one DMA instruction per wave and tap) + a ring of R weight tiles [TN rows][64 B] (R - 1 taps ahead)
//   swizzle   16-byte slot = piece ^ ((row >> 2) & 3) on the DMA SOURCE side; a fragment read (32 consecutive rows, any start --
//             the tap's dx shifts the rows) then covers all 64 banks once per 16 lanes.  272 % 16 == 0 keeps the rule segment-free.
//   borders   rows outside the image in y are staged as zeros (segment -1 holds no pixel of an image's last row, segment +1 none
//             of its first row: those would be read across the image boundary); the x borders depend on (pixel, dx) and are applied
//             to the fragment by value (a pixel in column 0 reads zeros for dx = -1, in column W - 1 for dx = +1).
//   sync      one raw s_barrier per tap, counted s_waitcnt vmcnt (inline-asm LDS-DMA, see csrc/conv_gemm.hip), no __syncthreads.
#include "common.h"

#ifndef HALO_ABL            // tools/halo_probe.hip compiles this file with ablation bits (timing probes; results are garbage): 1 no MFMAs, 2 no DMA
#define HALO_ABL 0         // in the loop, 4 no fragment reads, 8 no barrier / vmcnt wait, 16 no global stores, 32 no epilogue, 64 no prologue
#endif                     // DMA, 128 no vmcnt wait.  The library build has none of them (every use is an `if constexpr` on this constant).

namespace {

constexpr int HTM = 256;                 // pixels per tile
constexpr int KC = 32;                   // channels per K block
constexpr int ROWB = KC * 2;             // bytes per LDS row
constexpr int SEGROWS = 272;             // 17 DMA chunks of 16 rows (258 used)
constexpr int SEGB = SEGROWS * ROWB;     // 17 KB
constexpr int SEGCH = 17;                // chunks per segment
constexpr int XBUFS = 4;                 // segment buffers (three live + the one being staged)
constexpr int XLDS = XBUFS * SEGB + 1024;        // + one chunk that takes the surplus (dummy) DMA instructions

typedef __attribute__((address_space(3))) void* lptr_h;

__device__ uint4 g_halo_zero[4096];      // zero-initialised: the source of every masked 16-byte piece, one line per lane (no shared hot line)

__device__ __forceinline__ void glds16h(uint64_t gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmh() {
    static_assert(N >= 0 && N <= 9, "vmcnt literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}

template <int NT> struct HaloCfg {
    static constexpr int TN = 64 * NT;               // output channels per tile
    static constexpr int R = NT == 2 ? 6 : 5;        // weight-tile ring: tiles of R - 1 taps staged or in flight
    static constexpr int WTB = TN * ROWB;            // bytes per weight tile (one tap, 32 channels)
    static constexpr int WI = TN / 128;              // weight DMA instructions per wave and tap
    static constexpr int LDS = XLDS + R * WTB;
};

// Schedule (per wave; every wave issues the same number of DMA instructions per tap, so the vmcnt literals are compile-time):
//   iteration g = (channel block cb, tap t):
//     s_waitcnt vmcnt: weight tile g + 1 has landed (and every DMA issued before it)      s_barrier: ... for every wave; all waves are
//     done with iteration g - 1, i.e. with ring slot (g - 1) % R and with the input segment whose last tap was t - 1
//     DMA: one piece of the NEXT channel block's input segment t / 3 (into the buffer iteration g - 1 or earlier retired), then the
//          weight tile of tap g + R - 1 into slot (g - 1) % R
//     LDS reads: the k-step-1 fragments of tap g;  MFMAs of k-step 0 (fragments read during iteration g - 1)
//     LDS reads: the k-step-0 fragments of tap g + 1 (published by this iteration's barrier);  MFMAs of k-step 1
//   so no wave waits for an LDS read right behind the barrier, where all eight would queue at the LDS at once.
// Input segments rotate over four buffers: segment s of block cb lives in buffer (s - cb) & 3; the next block's segment 0 goes to
// the spare buffer during taps 0-2, its segment 1 to this block's segment-0 buffer during taps 3-5 (free after tap 2), its segment 2
// to this block's segment-1 buffer during taps 6-8 (free after tap 5).
template <int NT, bool RELU>
__global__ __launch_bounds__(512, 1) void conv_halo_kernel(const bf16* __restrict__ x, const bf16* __restrict__ Wt, const float* __restrict__ bias,
                                                           const bf16* __restrict__ gate, bf16* __restrict__ C, int64_t M, int H, int W, int Cin,
                                                           int Cout, int mtiles, int ntiles) {
    using Cfg = HaloCfg<NT>;
    constexpr int TN = Cfg::TN, R = Cfg::R, WTB = Cfg::WTB, WI = Cfg::WI;
    extern __shared__ __attribute__((aligned(1024))) char lds_h[];        // [4][SEGB] input segments | dummy chunk | [R][WTB] weight ring
    const unsigned xaddr = (unsigned)(uintptr_t)(lptr_h)lds_h;
    const unsigned waddr = xaddr + XLDS;
    // XCD-aware tile order (blocks b and b + 8 share an XCD): every XCD gets a contiguous run of tiles, n fastest -- the n tiles of a
    // pixel tile and its neighbours (whose halos overlap) hit the same L2
    const int nblk = mtiles * ntiles;
    int id = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = id % 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
    }
    const int mt_ = id / ntiles, nt_ = id - mt_ * ntiles;
    const int p0 = mt_ * HTM;                                  // M < 2^31 (host-checked): 32-bit pixel arithmetic
    const int n0 = nt_ * TN;
    const int K = 9 * Cin;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int c = lane & 31, h = lane >> 5;

    // ---- DMA sources.  Input segment s: chunks j = wave + 8 i, i = 0..2 (16 rows x 64 B per instruction; j >= 17: a dummy), lane -> row
    // (lane >> 2), slot lane & 3.  Kept as 32-bit byte offsets from x (the activation is < 4 GB, host-checked).
    const uint64_t zero64 = (uint64_t)reinterpret_cast<uintptr_t>(g_halo_zero + (((int)blockIdx.x * 512 + tid) & 4095));
    const uint64_t xbase = (uint64_t)reinterpret_cast<uintptr_t>(x);
    unsigned xo[3][3];
    unsigned xok = 0;
#pragma unroll
    for (int sg = 0; sg < 3; ++sg)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 8 * i;
            xo[sg][i] = 0;
            if (j < SEGCH) {
                const int row = j * 16 + (lane >> 2);
                const int piece = (lane & 3) ^ ((row >> 2) & 3);
                const int q = p0 + (sg - 1) * W - 1 + row;
                bool ok = q >= 0 && q < (int)M;
                if (ok && sg != 1) {
                    const int yq = (q / W) % H;
                    if (sg == 0 ? yq == H - 1 : yq == 0) ok = false;       // would be read across the top / bottom border of an image
                }
                if (ok) { xok |= 1u << (sg * 3 + i); xo[sg][i] = ((unsigned)q * (unsigned)Cin + piece * 8) * 2u; }
            }
        }
    uint64_t wptr[WI];
    bool wok[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int row = 128 * i + 16 * wave + (lane >> 2);
        const int piece = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        wok[i] = n < Cout;
        wptr[i] = (uint64_t)reinterpret_cast<uintptr_t>(Wt + (wok[i] ? (int64_t)n * K : 0) + piece * 8);
    }
    const int ncb = Cin / KC;
    auto issue_x = [&](int sg, int i, int cb) {               // piece i of segment sg of channel block cb (cb == ncb: a dummy, keeps the counts uniform)
        const int j = wave + 8 * i;
        const unsigned dst = xaddr + (j < SEGCH ? ((sg - cb) & 3) * SEGB + j * 1024 : XBUFS * SEGB);
        const bool live = ((xok >> (sg * 3 + i)) & 1u) && cb < ncb;
        glds16h(live ? xbase + xo[sg][i] + (unsigned)(cb * (KC * 2)) : zero64, dst);
    };
    int w_tap = 0, w_cb = 0, w_slot = 0;                      // the next weight tile to stage
    auto issue_w = [&]() {
        const uint64_t off = ((uint64_t)w_tap * Cin + (uint64_t)w_cb * KC) * 2;
        const bool live = w_cb < ncb;
#pragma unroll
        for (int i = 0; i < WI; ++i)
            glds16h(live && wok[i] ? wptr[i] + off : zero64, waddr + w_slot * WTB + (128 * i + 16 * wave) * ROWB);
        if (++w_tap == 9) { w_tap = 0; ++w_cb; }
        if (++w_slot == R) w_slot = 0;
    };

    // ---- fragment read offsets (bytes inside a weight tile / an input segment, k-step 0; k-step 1 = offset ^ 32)
    int woff[NT], xoff[2][3];
    bool lmask[2], rmask[2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int row = wn * 32 * NT + 32 * nt + c;
        woff[nt] = row * ROWB + 16 * (h ^ ((row >> 2) & 3));
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int i = wm * 64 + 32 * mt + c;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int row = i + dx;
            xoff[mt][dx] = row * ROWB + 16 * (h ^ ((row >> 2) & 3));
        }
        const int xc = (p0 + i) % W;
        lmask[mt] = xc == 0;
        rmask[mt] = xc == W - 1;
    }

    f32x16 acc[NT][2];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};

    const bf16x8 zfrag = bf16x8{0};
    struct Frag { bf16x8 w[NT], x[2]; };
    // fragments of tap t (compile-time) of channel block cb, weight ring slot `slot`, k-step ks: issued here, masked (x borders) at use
    auto load_frag = [&](Frag& f, auto tap_c, int cb, int slot, int ks) {
        constexpr int t = decltype(tap_c)::value, dy = t / 3, dx = t % 3;
        const char* xb = lds_h + ((dy - cb) & 3) * SEGB;
        const char* wb = lds_h + XLDS + slot * WTB;
        if constexpr ((HALO_ABL & 4) != 0) return;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) f.w[nt] = *(const bf16x8*)(wb + (woff[nt] ^ (ks * 32)));
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) f.x[mt] = *(const bf16x8*)(xb + (xoff[mt][dx] ^ (ks * 32)));
    };
    auto mask_x = [&](Frag& f, auto tap_c) {                  // the x borders of this tap, by value
        constexpr int dx = decltype(tap_c)::value % 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (dx == 0) f.x[mt] = lmask[mt] ? zfrag : f.x[mt];
            if (dx == 2) f.x[mt] = rmask[mt] ? zfrag : f.x[mt];
        }
    };
    auto mfmas = [&](const Frag& f, auto lo_c, auto hi_c) {      // MFMAs lo .. hi - 1 of the 2 NT of a k-step (index = nt * 2 + mt)
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
#pragma unroll
        for (int k = lo; k < hi; ++k) {
            const int nt = k >> 1, mt = k & 1;
            if constexpr ((HALO_ABL & 1) != 0) acc[nt][mt][0] += (float)f.w[nt][0] * (float)f.x[mt][0];
            else acc[nt][mt] = SWIN_MFMA_32x32x16(f.w[nt], f.x[mt], acc[nt][mt], 0, 0, 0);
        }
    };

    // ---- prologue: the three segments of channel block 0 and the weight tiles of taps 0 .. R - 2, paired the way the loop pairs them
    // (one input piece, then a weight tile) so that the loop's vmcnt literal holds from its first iteration
    if constexpr ((HALO_ABL & 64) == 0) {
        constexpr int first = 9 - (R - 1);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            issue_x(k / 3, k % 3, 0);
            if (k >= first) issue_w();
        }
    }
    wait_vmh<(R - 2) * (1 + WI)>();                          // tile 0 (and the pieces before it) landed; the rest of segment 2 follows in order
    Frag f0, f1;
    if constexpr ((HALO_ABL & 4) != 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { f0.w[nt] = zfrag + (bf16)(float)lane; f1.w[nt] = zfrag + (bf16)(float)wave; }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { f0.x[mt] = zfrag + (bf16)(float)c; f1.x[mt] = zfrag + (bf16)(float)h; }
    }
    int slot = 0;
    // the k-step-0 fragments of tap 0 need all of segment 0 only (dy = 0 of tap 0): pieces 0..2, issued before tile 0
    __builtin_amdgcn_s_barrier();
    load_frag(f0, std::integral_constant<int, 0>{}, 0, 0, 0);
    auto tap_iter = [&](auto tap_c, int cb) {
        constexpr int t = decltype(tap_c)::value, tn = (t + 1) % 9;
        if constexpr ((HALO_ABL & 8) == 0) {
            if constexpr ((HALO_ABL & 128) == 0) wait_vmh<(R - 3) * (1 + WI)>();
            __builtin_amdgcn_s_barrier();
        }
        const int slot_n = slot + 1 == R ? 0 : slot + 1;
        using I0 = std::integral_constant<int, 0>;
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        using IN = std::integral_constant<int, 2 * NT>;
        // the DMA issue (address selects, M0 moves) sits BETWEEN this wave's MFMAs: behind the barrier all eight waves are in the same
        // phase, so whatever a wave issues before its first MFMA is time the SIMD's MFMA pipe idles
        load_frag(f1, tap_c, cb, slot, 1);
        mask_x(f0, tap_c);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0, I0{}, I2{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((HALO_ABL & 2) == 0) issue_x(t / 3, t % 3, cb + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0, I2{}, I4{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((HALO_ABL & 2) == 0) issue_w();
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0, I4{}, IN{});
        __builtin_amdgcn_sched_barrier(0);
        load_frag(f0, std::integral_constant<int, tn>{}, t == 8 ? cb + 1 : cb, slot_n, 0);
        mask_x(f1, tap_c);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f1, I0{}, IN{});
        __builtin_amdgcn_sched_barrier(0);
        slot = slot_n;
    };
    for (int cb = 0; cb < ncb; ++cb) {
        tap_iter(std::integral_constant<int, 0>{}, cb);
        tap_iter(std::integral_constant<int, 1>{}, cb);
        tap_iter(std::integral_constant<int, 2>{}, cb);
        tap_iter(std::integral_constant<int, 3>{}, cb);
        tap_iter(std::integral_constant<int, 4>{}, cb);
        tap_iter(std::integral_constant<int, 5>{}, cb);
        tap_iter(std::integral_constant<int, 6>{}, cb);
        tap_iter(std::integral_constant<int, 7>{}, cb);
        tap_iter(std::integral_constant<int, 8>{}, cb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the surplus (dummy) DMA of the last taps: nothing may be in flight into LDS at exit

    // ---- epilogue through LDS.  In the accumulator layout a lane owns one pixel and 4 consecutive channels per register group: stored
    // straight from there a wave instruction writes 64 separate 8-byte pieces 2 * Cout bytes apart (measured: ~20 us per block for a
    // 256 x 256 tile, a quarter of the launch at P2).  The tile goes through LDS instead ([pixel][TN] bf16, row stride TN * 2 + 8 bytes:
    // conflict-free ds_write_b64 over 16 pixel rows) and leaves as 16-byte pieces of whole output rows.  bias / ReLU are applied before
    // the bf16 rounding, the gate after it -- the same values, bit for bit, as the direct epilogue of csrc/conv_gemm.hip.
    constexpr int RS = TN * 2 + 8;
    if constexpr ((HALO_ABL & 32) != 0) {                 // (probe: no epilogue at all beyond one store per lane that keeps the accumulators alive)
        float sacc = 0.f;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc += acc[a][b][e];
        if (sacc == 123.456f) C[tid] = (bf16)sacc;
        return;
    }
    f32x4 bv[NT][4];                                      // this lane's 4 NT x 4 bias values, as 16-byte loads (Cout % 8 == 0: whole or not at all)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int n = n0 + wn * 32 * NT + 32 * nt + 8 * gq + 4 * h;
            bv[nt][gq] = (bias && n < Cout) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    __builtin_amdgcn_s_barrier();                         // every wave has finished its fragment reads: the staging buffers are free
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int i = wm * 64 + 32 * mt + c;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int j = wn * 32 * NT + 32 * nt + 8 * gq + 4 * h;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[nt][mt][4 * gq + e] + bv[nt][gq][e];
                    if (RELU) v = fmaxf(v, 0.f);
                    o[e] = (bf16)v;
                }
                *(bf16x4*)(lds_h + i * RS + j * 2) = o;
            }
    }
    __syncthreads();
    constexpr int PPR = TN / 8;                           // 16-byte pieces per tile row
    const int jj = tid % PPR, r0 = tid / PPR;
    const int n = n0 + jj * 8;
    if (n < Cout) {                                       // Cout % 8 == 0 (host): a piece is inside or outside as a whole
#pragma unroll 4
        for (int k = 0; k < HTM * PPR / 512; ++k) {
            const int row = r0 + k * (512 / PPR);
            const int64_t m = p0 + row;
            if (m >= M) break;
            const bf16x4 lo = *(const bf16x4*)(lds_h + row * RS + jj * 16), hi = *(const bf16x4*)(lds_h + row * RS + jj * 16 + 8);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
            if constexpr ((HALO_ABL & 16) != 0) { if (o[0] != (bf16)123.456f) continue; }      // (probe: no global stores)
            if (gate) {                                   // ReLU backward of the layer below: zero where its output was not positive
                const bf16x8 gt = *(const bf16x8*)(gate + m * Cout + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) if (!((float)gt[e] > 0.f)) o[e] = (bf16)0.f;
            }
            *(bf16x8*)(C + m * Cout + n) = o;
        }
    }
}

template <int NT, bool RELU>
int launch_halo(const bf16* x, const bf16* Wt, const float* bias, const bf16* gate, bf16* C, int64_t M, int H, int W, int Cin, int Cout,
                hipStream_t s) {
    constexpr int TN = HaloCfg<NT>::TN;
    constexpr size_t epi_bytes = (size_t)HTM * (TN * 2 + 8);
    const size_t lds_bytes = HaloCfg<NT>::LDS > epi_bytes ? HaloCfg<NT>::LDS : epi_bytes;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)conv_halo_kernel<NT, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const int64_t mtiles = (M + HTM - 1) / HTM;
    const int ntiles = (Cout + TN - 1) / TN;
    if (mtiles * ntiles > (int64_t)1 << 30) return SWIN_ERR_UNSUPPORTED;
    conv_halo_kernel<NT, RELU><<<(unsigned)(mtiles * ntiles), 512, lds_bytes, s>>>(x, Wt, bias, gate, C, M, H, W, Cin, Cout, (int)mtiles, ntiles);
    return swin_launch_status();
}

}  // namespace

// nt: 2 (TN = 128) or 4 (TN = 256, Cout % 256 == 0); 0 = choose.  SWIN_ERR_UNSUPPORTED when the shape does not fit this form.
int swin_conv_halo(const bf16* x, const bf16* Wt, const float* bias, const bf16* gate, bf16* y, int N, int H, int W, int Cin, int Cout, int relu,
                   int nt, hipStream_t s) {
    if (Cin % KC != 0 || Cout % 8 != 0 || W < 2 || H < 1) return SWIN_ERR_UNSUPPORTED;
    const int64_t M = (int64_t)N * H * W;
    if ((M + 2 * (int64_t)W + 600) * Cin * 2 >= ((int64_t)1 << 32)) return SWIN_ERR_UNSUPPORTED;      // 32-bit byte offsets into x
    // 256-wide tiles once they fill more than half of the chip's 256 CUs (one block per CU): P2, the mask-head maps; below that the
    // 128-wide tile keeps twice as many blocks (P3: 45.8 us against 61.8 us and 55.6 us for the implicit GEMM; profiles/r03_conv_halo.txt)
    if (nt == 0) nt = (Cout % 256 == 0 && (M + HTM - 1) / HTM >= 160) ? 4 : 2;
    if (nt == 4 && Cout % 256 != 0) return SWIN_ERR_UNSUPPORTED;
    if (nt == 4) return relu ? launch_halo<4, true>(x, Wt, bias, gate, y, M, H, W, Cin, Cout, s) : launch_halo<4, false>(x, Wt, bias, gate, y, M, H, W, Cin, Cout, s);
    if (nt == 2) return relu ? launch_halo<2, true>(x, Wt, bias, gate, y, M, H, W, Cin, Cout, s) : launch_halo<2, false>(x, Wt, bias, gate, y, M, H, W, Cin, Cout, s);
    return SWIN_ERR_BAD_ARG;
}

// C ABI, always this form (parity tests, tools): SWIN_ERR_UNSUPPORTED instead of a fallback
extern "C" int conv3x3_halo_nhwc_bf16(const void* x, const void* w, const float* bias, const void* gate, void* y, int N, int H, int W, int Cin,
                                      int Cout, int relu, int nt, void* stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SWIN_ERR_BAD_ARG;
    return swin_conv_halo((const bf16*)x, (const bf16*)w, bias, (const bf16*)gate, (bf16*)y, N, H, W, Cin, Cout, relu, nt, (hipStream_t)stream);
}
