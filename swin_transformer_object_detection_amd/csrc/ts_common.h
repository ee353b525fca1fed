// Shared pieces of the token-stationary kernels (csrc/ts_mlp.hip, csrc/ts_linear.hip): the MFMA wrapper, the row permutation of an
// accumulator-fed operand, and the padded weight images in LDS with their LDS-DMA staging.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return SWIN_MFMA_32x32x16(a, b, c, 0, 0, 0);
}

// swap bits 2 and 3 of a row index (an involution on 0..15, applied inside each 16-row group)
__device__ __forceinline__ int pi16(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// Weight chunk images in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, the next chunk lands while
// this one feeds the MFMAs).  An image is a sequence of 16-byte slots, ROWS x (COLS/8 + 1) of them (the last slot of a row is
// the pad); a wave instruction writes 64 consecutive slots, every lane reading the source piece of ITS slot (the pad slot
// re-reads the row's last piece; slots past the image re-read the last row and land in the image's rounded-up tail).
template <int ROWS, int COLS> struct WImg {
    static constexpr int SPR = COLS / 8 + 1;                      // slots per row
    static constexpr int RS = COLS + 8;                           // row stride in bf16
    static constexpr int SLOTS = ROWS * SPR;
    static constexpr int SLOTS_PAD = (SLOTS + 63) / 64 * 64;
    static constexpr int BYTES = SLOTS_PAD * 16;
    static_assert((SPR & 1) == 1, "row stride must be an odd number of 16-byte slots (conflict-free ds_read_b128)");
};
typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA piece (16 B per lane, 1 KiB per wave) as inline asm: hipcc waits vmcnt(0) at the first ds_read after a
// __builtin_amdgcn_global_load_lds (an LDS-DMA is a pending LDS write it cannot disambiguate), which would drain the
// prefetch of the NEXT chunk before the MFMAs of this one.  An asm DMA is invisible to that bookkeeping; its completion is
// waited for by hand (s_waitcnt vmcnt(0) + barrier at the end of the chunk).  M0 = wave-uniform LDS byte address, written
// in the statement that reads it and restored (guide 5.7).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

// src: element pointer of (row 0, col 0) of the chunk; ld: source row stride in elements
template <int ROWS, int COLS, int WAVES>
__device__ __forceinline__ void dma_image(const bf16* src, int64_t ld, unsigned char* img, int wave, int lane) {
    using I = WImg<ROWS, COLS>;
    constexpr int ROUNDS = (I::SLOTS_PAD / 64 + WAVES - 1) / WAVES;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const int blk = i * WAVES + wave;                         // wave-uniform
        if (blk * 64 < I::SLOTS_PAD) {
            const int p = blk * 64 + lane;
            int row = p / I::SPR, sl = p - row * I::SPR;
            if (sl == I::SPR - 1) sl = I::SPR - 2;
            if (row >= ROWS) row = ROWS - 1;
            const bf16* g = src + (int64_t)row * ld + sl * 8;
            glds16(g, (unsigned)(uintptr_t)(lptr_t)(img + blk * 1024));
        }
    }
}

}  // namespace
