// Weight gradients of the backbone's Linear layers whose dimensions are multiples of 96 (Swin-T / Swin-S, every stage:
// swin_transformer.py:33,36 Mlp.fc1/fc2, :129 qkv, :151 proj, :296 PatchMerging.reduction), grouped:
// dW[n1][n2] += sum_t dY[t][n1] * X[t][n2], bf16 in, fp32 out.
//
// Why a third form next to csrc/wgrad_dma.hip (128 x 128 tiles).  In the step profile the grouped launches ran at 0.14-0.36 of
// the MFMA peak and 3-4 TB/s: at C = 96 a 96-wide operand fills three quarters of a 128-tile (a quarter of the DMA lanes read
// zero lines, 44 % of the MFMA work is padding) and is fetched once per 128-tile of the other operand; at C = 384 / 768 a
// 128 x 128 tile moves 32 KB through L2 -> LDS per 2.1 MFLOP, which is what bounds it (DESIGN, "Weight gradients, round 3");
// and whole-tile blocks with 1-2 splits of t leave the last round of blocks half empty.  Here
//   * a wave owns a 96 x 96 piece of dW (3 x 3 v_mfma_f32_32x32x16 accumulators); a block's A x B pieces form a tile of up to
//     384 x 192 (18 KB of operands per 2.4 MFLOP: half the bytes per flop), or the WHOLE dW of a C = 96 layer (every operand byte read
//     from HBM once); spare waves split the contraction (KG k-groups, folded through LDS before the atomics);
//   * a stage = R rows of both operands, brought in by LDS-DMA as whole 16-byte pieces of whole rows (no masked lanes), through a
//     ring of 3-4 stage buffers of 24-48 KB at one block per CU: 64-96 KB per CU in flight all the time;
//   * LDS rows are padded to a stride of 64 or 192 (mod 256) bytes, which the transposed fragment reads (ds_read_b64_tr_b16, four
//     rows x 32 bytes per 16-lane group) take without bank conflicts -- no swizzle needed;
//   * the work of a launch (up to 32 problems: a stage's worth of the backbone's backward) is ONE sequence of (problem, cluster of
//     <= 8 tiles, stage) units weighted by their DMA time, cut into equal ranges, one per GROUP of eight blocks ("stream-K" over
//     groups): the blocks of a group run on one XCD (blockIdx % 8) and walk the same stages of the <= 8 tiles of a cluster in step, so
//     the operand rows those tiles share are fetched from HBM once and hit in that XCD's L2 (with tile-by-tile ranges the timing matched 1.33 GB of
//     HBM reads for stage 3's 590 MB of operands: every tile streamed its operands on its own).  A problem with <= 4 tiles gives
//     each tile 8 / tiles sub-ranges of the group's stages, so no block idles: with one tile per problem (C = 96) this is plain
//     stream-K over blocks, with two (C = 192) pairs of blocks in step.  A segment ends with atomics into dW; the segments of different
//     groups end at different times, so the atomics mostly hide under other blocks' loops instead of forming a tail.
#include <algorithm>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lptr_n;

__device__ uint4 g_zero_rows[4096];
#ifdef SWIN_DEV
__device__ int g_w96_abl;       // development ablations: 1 no atomics, 2 DMA pointers do not advance, 4 no fragment reads / MFMAs, 8 no DMA, 16 no bias sums
#endif          // zero-initialised: source of rows beyond a split's range and of the padding slots

__device__ __forceinline__ void glds16n(uint64_t gsrc, unsigned lds_addr) {      // see csrc/conv_gemm.hip: asm, so that hipcc does not drain it
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmn() {
    static_assert(N >= 0 && N < 64, "vmcnt literal");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}

// acc + v[0] + v[1] in fp32 (v_dot2c_f32_bf16 / v_dot2c_f32_f16 against the constant pair (1, 1))
__device__ __forceinline__ float dot2_ones(bf16x2 v, float acc) {
    bf16x2 one; one[0] = (bf16)1.0f; one[1] = (bf16)1.0f;
#ifdef SWIN_HALF
    return __builtin_amdgcn_fdot2(v, one, acc, false);
#else
    return __builtin_amdgcn_fdot2_f32_bf16(v, one, acc, false);
#endif
}

constexpr int pad_stride(int cols) { return ((cols * 2) % 256 == 64 || (cols * 2) % 256 == 192) ? cols * 2 : cols * 2 + 64; }

// A x B pieces of 96 x 96, KG k-groups, R rows per stage, NBUF stage buffers
template <int A_, int B_, int KG_, int R_, int NBUF_>
struct NCfg {
    static constexpr int A = A_, B = B_, KG = KG_, R = R_, NBUF = NBUF_;
    static constexpr int NW = A * B * KG;                       // waves
    static constexpr int SA = pad_stride(96 * A), SB = pad_stride(96 * B);      // LDS row strides (bytes)
    static constexpr int CA = SA / 16, CB = SB / 16;            // 16-byte slots per LDS row
    static constexpr int RAW = R * (SA + SB);
    static constexpr int DPT = (RAW / 1024 + NW - 1) / NW;      // DMA instructions per wave and stage
    static constexpr int STAGEB = DPT * NW * 1024;
    static constexpr int KS = R / 16 / KG;                      // k-steps per wave and stage
    static_assert(NW <= 8 && RAW % 1024 == 0 && KS >= 1 && KS * KG * 16 == R, "geometry");
    static_assert(NBUF * STAGEB <= 163840, "LDS");
    static_assert((KG - 1) * A * B * 49 * 256 <= NBUF * STAGEB, "fold buffer");
};

// stages = ceil(T / R) of the class; sub = sub-ranges per tile when the problem has <= 4 tiles (8 / tiles, else 1); wt = weight of one
// stage of one cluster in the work sequence = (1 KB units of DMA per stage) x 8 / sub
struct NProb { const bf16* dy; const bf16* x; float* dw; float* db; int64_t T; int N1, N2, g1, g2, cls, stages, sub, wt; };
constexpr int NMAX = 32;
// group g of eight blocks walks [total g / ng, total (g + 1) / ng) of the sequence; cum[i] = first unit of problem i
struct NTab { int n, pad; int64_t total; int64_t cum[NMAX + 1]; NProb p[NMAX]; };

// one segment: stages [s0, s1) of tile `tile` of problem q.  Called by all eight waves of the block (block-uniform arguments); waves
// beyond the class's NW only keep the barriers company.
template <typename Cfg>
__device__ __forceinline__ void segment(char* lds, const NProb& q, const int tile, const int s0, const int s1) {
    constexpr int A = Cfg::A, B = Cfg::B, KG = Cfg::KG, R = Cfg::R, NBUF = Cfg::NBUF, NW = Cfg::NW, DPT = Cfg::DPT;
    constexpr int SA = Cfg::SA, SB = Cfg::SB, CA = Cfg::CA, CB = Cfg::CB, STAGEB = Cfg::STAGEB, KS = Cfg::KS;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool active = wave < NW;
    const int lane = threadIdx.x & 63;
    const int n1_0 = (tile / q.g2) * (96 * A), n2_0 = (tile % q.g2) * (96 * B);
    const int64_t t_begin = (int64_t)s0 * R, t_end = min(q.T, (int64_t)s1 * R);
    const int N1 = q.N1, N2 = q.N2;
    const unsigned ring_addr = (unsigned)(uintptr_t)(lptr_n)lds;
    __builtin_amdgcn_s_barrier();             // the previous segment's fold buffer (= this ring) has been read

    // ---- DMA: instruction j of this wave fills slots ((j NW + wave) 64 + lane) of the stage image [R][CA] | [R][CB] | spare.
    // Per lane and instruction: a source pointer, the stage row (8 bits, 255 = padding slot: always a zero line) and one bit
    // "X operand" (the pointer's step per stage)
    uint64_t ptr[DPT];
    unsigned rows_pk[(DPT + 3) / 4] = {}, xmask = 0;
    const unsigned step_a = (unsigned)(R * 2) * (unsigned)N1, step_b = (unsigned)(R * 2) * (unsigned)N2;
    auto zero_line = [&]() { return (uint64_t)reinterpret_cast<uintptr_t>(g_zero_rows + ((blockIdx.x * 512 + threadIdx.x) & 4095)); };
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int slot = (j * NW + wave) * 64 + lane;
        unsigned rowj = 255;
        ptr[j] = 0;
        if (active && slot < R * CA) {
            const int r = slot / CA, c = slot - r * CA;
            if (c < 12 * A) rowj = (unsigned)r;
            ptr[j] = (uint64_t)reinterpret_cast<uintptr_t>(q.dy + (t_begin + r) * (int64_t)N1 + n1_0 + min(c, 12 * A - 1) * 8);
        } else if (active && slot < R * (CA + CB)) {
            const int s2 = slot - R * CA;
            const int r = s2 / CB, c = s2 - r * CB;
            if (c < 12 * B) rowj = (unsigned)r;
            ptr[j] = (uint64_t)reinterpret_cast<uintptr_t>(q.x + (t_begin + r) * (int64_t)N2 + n2_0 + min(c, 12 * B - 1) * 8);
            xmask |= 1u << j;
        }
        rows_pk[j >> 2] |= rowj << (8 * (j & 3));
    }
    static_assert(R <= 128, "stage rows are packed in 8 bits");
    int64_t t_stage = t_begin;
#ifdef SWIN_DEV
    const int abl = __builtin_amdgcn_readfirstlane(g_w96_abl);
#endif
    auto dma_stage = [&](int buf) {
        const int left = (int)min((int64_t)R, t_end - t_stage);
        const unsigned dst = ring_addr + buf * STAGEB + wave * 1024;
        const uint64_t zero64 = zero_line();
#ifdef SWIN_DEV
        if (abl & 8) { t_stage += R; return; }
#endif
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
            const int rowj = (int)((rows_pk[j >> 2] >> (8 * (j & 3))) & 255u);
            glds16n(rowj < left ? ptr[j] : zero64, dst + j * (NW * 1024));
#ifdef SWIN_DEV
            if (!(abl & 2))
#endif
            ptr[j] += ((xmask >> j) & 1u) ? step_b : step_a;
        }
        t_stage += R;
    };

    // ---- this wave's piece and k-group
    const int wv = active ? wave : 0;                                  // idle waves: valid addresses, never used
    const int kg = wv / (A * B), piece = wv - kg * (A * B);
    const int pa = piece / B, pb = piece - pa * B;
    const int g4 = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    const int rlo = 8 * (g4 >> 1) + qq;                                // first 4-row half of the k-step; the second is rlo + 4
    const int colb = 2 * 16 * (g4 & 1) + 16 * (pp >> 1) + 8 * (pp & 1);       // byte offset of this lane's 8 bytes inside a 32-column fragment
    const int a_base = rlo * SA + 2 * 96 * pa + colb + kg * (16 * SA);
    const int b_base = R * SA + rlo * SB + 2 * 96 * pb + colb + kg * (16 * SB);

    f32x16 acc[3][3];
    float bsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x16{0};
    // bias gradient = column sums of dY: a lane's dY fragment holds 8 rows of ONE column, so three fp32 registers carry them (an MFMA
    // against ones would cost 48 accumulator registers on top of the 144; these problems are HBM-bound, the VALU adds are free)
#ifdef SWIN_DEV
    const bool do_bias = active && q.db != nullptr && n2_0 == 0 && pb == 0 && !(__builtin_amdgcn_readfirstlane(g_w96_abl) & 16);
#else
    const bool do_bias = active && q.db != nullptr && n2_0 == 0 && pb == 0;
#endif

    struct Frags { bf16x8 a[3], b[3]; };
    typedef __attribute__((address_space(3))) char lchar;
    lchar* const lbase = (lchar*)lds;
    // one run-time address per operand and k-step, every fragment at a compile-time offset from it (the ds_read offset field)
    auto read_frags = [&](Frags& f, int buf, int ks) {               // k-step ks of this wave in stage buffer buf
        lchar* const pa_ = lbase + (buf * STAGEB + a_base);
        lchar* const pb_ = lbase + (buf * STAGEB + b_base);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bf16x4 alo = SWIN_DS_READ_TR16(pa_ + (ks * KG * 16 * SA + 64 * i));
            const bf16x4 ahi = SWIN_DS_READ_TR16(pa_ + (ks * KG * 16 * SA + 64 * i + 4 * SA));
            const bf16x4 blo = SWIN_DS_READ_TR16(pb_ + (ks * KG * 16 * SB + 64 * i));
            const bf16x4 bhi = SWIN_DS_READ_TR16(pb_ + (ks * KG * 16 * SB + 64 * i + 4 * SB));
#pragma unroll
            for (int e = 0; e < 4; ++e) { f.a[i][e] = alo[e]; f.a[i][4 + e] = ahi[e]; f.b[i][e] = blo[e]; f.b[i][4 + e] = bhi[e]; }
        }
    };
    auto mma = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = SWIN_MFMA_32x32x16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
        if (do_bias) {                                               // v_dot2c_f32 against (1, 1): four instructions per fragment
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16x2 v; v[0] = f.a[i][2 * e]; v[1] = f.a[i][2 * e + 1];
                    bsum[i] = dot2_ones(v, bsum[i]);
                }
        }
    };

    // ---- the ring.  All NBUF buffers are filled ahead; a stage has landed for this wave when at most DPT x (stages issued after it) of
    // its DMA instructions are outstanding (counted s_waitcnt vmcnt), and the barrier publishes it.  The fragments run ONE k-step ahead
    // of the MFMAs, across the stage boundary: the barrier that publishes stage it + 1 sits before the LAST k-step of stage it, whose
    // fragments are in registers by then (lgkmcnt(0)), so the same barrier frees buffer it for stage it + NBUF and the first fragments
    // of stage it + 1 are read under the last MFMAs of stage it.  Two fragment sets alternate by name (no register copies), every
    // fragment address is one VGPR + an immediate: a k-step is 12 ds_reads, 9 MFMAs and a handful of VALU instructions.
    static_assert(KS == 1 || KS == 2, "k-steps per stage");
    const int iters = s1 - s0;
#ifdef SWIN_DEV
    const bool do_compute = !(abl & 4);
#else
    constexpr bool do_compute = true;
#endif
    auto wait_first = [&]() {
        const int after = min(NBUF - 1, iters - 1);
        if (after >= 3) wait_vmn<3 * DPT>();
        else if (after == 2) wait_vmn<2 * DPT>();
        else if (after == 1) wait_vmn<DPT>();
        else wait_vmn<0>();
    };
    auto wait_next = [&](int it) {                                    // stage it + 1
        const int after = min(NBUF - 2, iters - 2 - it);
        if (NBUF >= 4 && after >= 2) wait_vmn<2 * DPT>();
        else if (after >= 1) wait_vmn<DPT>();
        else wait_vmn<0>();
    };
    if (active) {
#pragma unroll
        for (int i = 0; i < NBUF; ++i)
            if (i < iters) dma_stage(i);
        wait_first();
        __builtin_amdgcn_s_barrier();
        Frags f0, f1;
        auto kstep = [&](Frags& cur, Frags& nxt, int it, int ks) {
            const int buf = it % NBUF;
            if (ks + 1 < KS) {
                if (do_compute) read_frags(nxt, buf, ks + 1);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every read of buffer `buf` by this wave has returned
                if (it + 1 < iters) {
                    wait_next(it);
                    __builtin_amdgcn_s_barrier();
                    if (it + NBUF < iters) dma_stage(buf);
                    if (do_compute) read_frags(nxt, (it + 1) % NBUF, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (do_compute) mma(cur);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (do_compute) read_frags(f0, 0, 0);
        if constexpr (KS == 2) {
            for (int it = 0; it < iters; ++it) { kstep(f0, f1, it, 0); kstep(f1, f0, it, 1); }
        } else {
            for (int it = 0; it < iters; it += 2) {
                kstep(f0, f1, it, 0);
                if (it + 1 < iters) kstep(f1, f0, it + 1, 0);
            }
        }
    } else {                                                          // the same barriers, nothing else
        __builtin_amdgcn_s_barrier();
        for (int it = 0; it + 1 < iters; ++it) __builtin_amdgcn_s_barrier();
    }

    // ---- fold the k-groups through LDS, one row of accumulator tiles at a time: [k-group - 1][piece][tile j x register | bias][lane]
    const int c = lane & 31, h = lane >> 5;
    if constexpr (KG > 1) {
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_s_barrier();                         // the ring (or the previous round's sums) has been consumed
            if (active && kg > 0) {
                float* w = red + ((size_t)((kg - 1) * (A * B) + piece) * 49) * 64 + lane;
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) w[(j * 16 + reg) * 64] = acc[i][j][reg];
                w[48 * 64] = bsum[i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (active && kg == 0) {
#pragma unroll 1
                for (int g = 1; g < KG; ++g) {
                    const float* r = red + ((size_t)((g - 1) * (A * B) + piece) * 49) * 64 + lane;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) acc[i][j][reg] += r[(j * 16 + reg) * 64];
                    bsum[i] += r[48 * 64];
                }
            }
        }
    }
    if (!active || kg != 0) return;
#ifdef SWIN_DEV
    if (abl & 1) return;
#endif
    // ---- D[row n1][col n2]: lane = n2 column, registers = n1 rows (csrc/wgrad_dma.hip)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n2 = n2_0 + 96 * pb + 32 * j + c;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int n1 = n1_0 + 96 * pa + 32 * i + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float* d = q.dw + (int64_t)n1 * N2 + n2;
                atomicAdd(d, acc[i][j][reg]);
            }
        }
    if (do_bias) {                               // lanes l and l ^ 32 hold the two row halves of column l & 31
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float t = bsum[i] + __shfl_xor(bsum[i], 32);
            if (h == 0) atomicAdd(q.db + n1_0 + 96 * pa + 32 * i + c, t);
        }
    }
}

// shape classes: (pieces of dY, pieces of X) per block tile, k-groups, rows per stage, ring depth
typedef NCfg<3, 1, 2, 32, 4> Cls0;      // N1 = 288 k, N2 = 96: qkv at C = 96 -- 24 KB stages
typedef NCfg<1, 1, 8, 128, 3> Cls1;     // any other multiple of 96 (proj at C = 96): 96 x 96 tiles, 48 KB stages, eight k-groups
typedef NCfg<4, 1, 2, 32, 4> Cls2;      // fc1 at C = 96 (384 x 96)
typedef NCfg<1, 4, 2, 32, 4> Cls3;      // fc2 at C = 96 (96 x 384)
typedef NCfg<3, 2, 1, 32, 4> Cls4;      // 288 x 192 tiles: qkv at C = 192
typedef NCfg<2, 2, 2, 32, 4> Cls5;      // 192 x 192 tiles: proj at C = 192
typedef NCfg<4, 2, 1, 32, 4> Cls6;      // 384 x 192 tiles: fc1 at C = 192; qkv / proj / fc1 / fc2 at C = 384, 768; PatchMerging.reduction
typedef NCfg<2, 4, 1, 32, 4> Cls7;      // 192 x 384 tiles: fc2 at C = 192, PatchMerging.reduction of stage 1
constexpr int W96_LDS = 163840;          // all of a CU's LDS: one block per CU

__device__ __forceinline__ void run_segment(char* lds, const NProb& q, int tile, int s0, int s1) {
    switch (q.cls) {
        case 0: segment<Cls0>(lds, q, tile, s0, s1); break;
        case 1: segment<Cls1>(lds, q, tile, s0, s1); break;
        case 2: segment<Cls2>(lds, q, tile, s0, s1); break;
        case 3: segment<Cls3>(lds, q, tile, s0, s1); break;
        case 4: segment<Cls4>(lds, q, tile, s0, s1); break;
        case 5: segment<Cls5>(lds, q, tile, s0, s1); break;
        case 6: segment<Cls6>(lds, q, tile, s0, s1); break;
        default: segment<Cls7>(lds, q, tile, s0, s1); break;
    }
}

__global__ __launch_bounds__(512, 1) void wgrad96_kernel(const NTab tab) {
    extern __shared__ __attribute__((aligned(16))) char lds_n[];
    const int L = (int)blockIdx.x, ng = (int)gridDim.x >> 3;
    int g, j;                                                        // group, and this block's place in it
    if ((ng & 7) == 0) { const int k = L >> 3; g = (L & 7) + 8 * (k >> 3); j = k & 7; }        // the eight blocks of a group: same blockIdx % 8
    else { g = L >> 3; j = L & 7; }                                                          // (small launches)
    // every quantity below is block-uniform: the loop runs the same number of times in every wave, and every wave reaches the end
    int64_t w = tab.total * g / ng;
    const int64_t w1 = tab.total * (g + 1) / ng;
    int guard = 0;
    while (w < w1 && guard++ < 80) {
        int pi = 0;
        while (pi + 1 < tab.n && w >= tab.cum[pi + 1]) ++pi;
        pi = __builtin_amdgcn_readfirstlane(pi);
        const NProb& q = tab.p[pi];
        const int64_t base = tab.cum[pi], per_cluster = (int64_t)q.stages * q.wt;
        const int cl = (int)((w - base) / per_cluster);
        const int64_t cl_begin = base + cl * per_cluster, cl_end = cl_begin + per_cluster;
        const int s0 = (int)((w - cl_begin) / q.wt);                  // floor at both ends: the neighbour's range ends / begins at the same stage
        const int s1 = w1 >= cl_end ? q.stages : (int)((w1 - cl_begin) / q.wt);
        if (s1 > s0) {
            const int ntc = min(8, q.g1 * q.g2 - 8 * cl);             // tiles of this cluster
            const int r = j / ntc, tile = 8 * cl + j - r * ntc;
            if (r < q.sub) {
                const int a = s0 + (int)((int64_t)(s1 - s0) * r / q.sub), e = s0 + (int)((int64_t)(s1 - s0) * (r + 1) / q.sub);
                if (e > a) run_segment(lds_n, q, tile, a, e);
            }
        }
        if (w1 < cl_end) break;
        w = cl_end;
    }
}

struct ClsInfo { int a, b, rows, u; };
const ClsInfo kCls[8] = {{3, 1, 32, Cls0::STAGEB / 1024}, {1, 1, 128, Cls1::STAGEB / 1024}, {4, 1, 32, Cls2::STAGEB / 1024},
                         {1, 4, 32, Cls3::STAGEB / 1024}, {3, 2, 32, Cls4::STAGEB / 1024}, {2, 2, 32, Cls5::STAGEB / 1024},
                         {4, 2, 32, Cls6::STAGEB / 1024}, {2, 4, 32, Cls7::STAGEB / 1024}};

}  // namespace

// class of a problem, -1 = not this kernel's.  Every (N1, N2) of multiples of 96 has one; the tile is the largest that divides both.
int wgrad96_class(int64_t T, int N1, int N2) {
    if (T < 1 || N1 < 96 || N2 < 96 || N1 % 96 || N2 % 96) return -1;
    const int a = N1 / 96, b = N2 / 96;
    if (a % 4 == 0 && b % 2 == 0) return 6;
    if (a % 2 == 0 && b % 4 == 0) return 7;
    if (a % 3 == 0 && b % 2 == 0) return 4;
    if (a % 2 == 0 && b % 2 == 0) return 5;
    if (a % 4 == 0 && b == 1) return 2;
    if (a == 1 && b % 4 == 0) return 3;
    if (a % 3 == 0 && b == 1) return 0;
    return 1;
}

// One grouped launch of n <= 32 problems.  blocks: 0 = one per CU.
int wgrad96_launch(const void* const* dy, const void* const* x, float* const* dw, float* const* db, const int64_t* T, const int* N1,
                   const int* N2, int n, int blocks, hipStream_t s) {
    if (n < 1 || n > NMAX) return SWIN_ERR_BAD_ARG;
    static bool attr_set[16] = {};
    static int cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)wgrad96_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W96_LDS) != hipSuccess)
            return SWIN_ERR_LAUNCH;
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_set[dev] = true;
    }
    if (blocks <= 0) blocks = cus[dev];
    NTab tab;
    tab.n = n; tab.pad = 0;
    int64_t units = 0;
    for (int i = 0; i < n; ++i) {
        NProb& q = tab.p[i];
        q.cls = wgrad96_class(T[i], N1[i], N2[i]);
        if (q.cls < 0) return SWIN_ERR_UNSUPPORTED;
        const ClsInfo& c = kCls[q.cls];
        q.dy = (const bf16*)dy[i]; q.x = (const bf16*)x[i]; q.dw = dw[i]; q.db = db ? db[i] : nullptr; q.T = T[i]; q.N1 = N1[i]; q.N2 = N2[i];
        q.g1 = N1[i] / (96 * c.a); q.g2 = N2[i] / (96 * c.b);
        const int64_t stages = (T[i] + c.rows - 1) / c.rows, nt = (int64_t)q.g1 * q.g2;
        if (stages > (1 << 30) || nt > (1 << 20)) return SWIN_ERR_UNSUPPORTED;
        q.stages = (int)stages;
        q.sub = nt == 1 ? 8 : nt == 2 ? 4 : nt <= 4 ? 2 : 1;
        q.wt = c.u * 8 / q.sub;
        tab.cum[i] = units;
        units += (nt + 7) / 8 * stages * q.wt;
    }
    tab.cum[n] = units; tab.total = units;
    if (units <= 0) return SWIN_ERR_UNSUPPORTED;
    // a group should have at least ~8 stages of an eight-tile cluster to walk
    int groups = (int)std::min<int64_t>(blocks / 8, std::max<int64_t>(1, units / (8 * 40 * 8)));
    if (groups > 8) groups = groups / 8 * 8;                    // the kernel's XCD map deals groups to blockIdx % 8 in eights
    if (groups < 1) groups = 1;
#ifdef SWIN_DEV
    { const int abl = swin_dev_int("SWIN_WGRAD96_ABL", 0); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w96_abl), &abl, sizeof(int)); }
#endif
    wgrad96_kernel<<<(unsigned)groups * 8, 512, W96_LDS, s>>>(tab);
    return swin_launch_status();
}

// C ABI (include/swin_hip.h): the grouped launch on its own -- tests and microbenchmarks; training goes through swin_wgrad_record / _flush
extern "C" int swin_wgrad96_group(const void* const* dy, const void* const* x, float* const* dw, float* const* db, const int64_t* T,
                                  const int* N1, const int* N2, int n, void* stream) {
    if (!dy || !x || !dw || !T || !N1 || !N2) return SWIN_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i)
        if (!dy[i] || !x[i] || !dw[i] || T[i] <= 0) return SWIN_ERR_BAD_ARG;
    return wgrad96_launch(dy, x, dw, db, T, N1, N2, n, swin_dev_int("SWIN_WGRAD96_BLOCKS", 0), (hipStream_t)stream);
}
