// Token-stationary Linear layers of the attention branch for the narrow stages (C = 96 / 128 / 192 / 256), round 3:
//
//   swin_ts_linear_bf16       y = x W^T + b                     qkv = norm1(x) Wqkv^T + bqkv          (swin_transformer.py:129)
//   swin_ts_proj_add_ln_bf16  x1 = x + dp * (o W^T + b);  n2 = LayerNorm(x1)                          (swin_transformer.py:150-151, 252-253)
//
// At these widths the GEMMs move 100 MB for 7 GFLOP: they are HBM-bound, and as library calls they run at ~3 TB/s and leave the
// projection's output to a separate residual + LayerNorm launch (write y, read y and x, write x1 and n2).  Here a WAVE owns 32 tokens
// (as in csrc/ts_mlp.hip): its token fragments stay in registers as the B operand, the weight rows come from an LDS image staged
// by LDS-DMA, and an output tile has the token on the lane and 16 channels in the registers -- so the projection's whole output
// row of a token sits in TWO lanes (lane r and r + 32), and residual, DropPath scale, LayerNorm statistics (one cross-lane add
// each) and both stores happen in the epilogue.  Same arithmetic as the two-launch chain: the product is rounded to bf16 before the
// residual add (the reference's Linear output is a half tensor under autocast), x1 is rounded before the statistics are taken
// (csrc/layernorm.hip does the same), two-pass variance.
#include "ts_common.h"

namespace {

// register q of an output tile <-> channel offset inside the tile (ts_mlp.hip: registers 0..7 = channels 8h.., 8..15 = 16 + 8h..)
__device__ __forceinline__ int tile_ch(int q, int h) { return (q < 8 ? 8 * h + q : 16 + 8 * h + (q - 8)); }

template <int C, int WAVES, int CH>
__global__ __launch_bounds__(WAVES * 64) void ts_linear_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W, const bf16* __restrict__ bias,
                                                               bf16* __restrict__ Y, int64_t T, int N, int relu) {
    using I = WImg<CH, C>;
    constexpr int KS = C / 16, HT = CH / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // two chunk images | bias as f32 [N]
    float* bs = (float*)(smem + 2 * I::BYTES);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tok = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32 + r;
    const int64_t tokc = tok < T ? tok : T - 1;
    // blockIdx.y takes an equal share of the N / CH output chunks (few tokens, wide layers: stage 3 has 8000 tokens = 32 blocks of
    // 256, the grid's second dimension fills the chip)
    const int nch_all = N / CH;
    const int jb = (int)((int64_t)nch_all * blockIdx.y / gridDim.y), nch = (int)((int64_t)nch_all * (blockIdx.y + 1) / gridDim.y) - jb;
    if (nch <= 0) return;
    W += (int64_t)jb * CH * C;
    dma_image<CH, C, WAVES>(W, C, smem, wave, lane);
    for (int i = tid; i < N; i += WAVES * 64) bs[i] = bias ? (float)bias[i] : 0.f;
    bf16x8 xf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xf[s] = *(const bf16x8*)(X + tokc * C + 16 * s + 8 * h);
    const int pr = pi16(r & 15) | (r & 16);
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" :: "v"(xf[s]));                // consumed here: the wait below covers them (see ts_mlp.hip)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < nch; ++j) {
        if (j + 1 < nch) dma_image<CH, C, WAVES>(W + (int64_t)(j + 1) * CH * C, C, smem + ((j + 1) & 1) * I::BYTES, wave, lane);
        const bf16* Ws = (const bf16*)(smem + (j & 1) * I::BYTES);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
            f32x16 a;
            const int n0 = (jb + j) * CH + 32 * t;
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = bs[n0 + tile_ch(q, h)];
            const bf16* wrow = Ws + (32 * t + pr) * I::RS + 8 * h;
#pragma unroll
            for (int s = 0; s < KS; ++s) a = mfma32(*(const bf16x8*)(wrow + 16 * s), xf[s], a);
            if (tok < T) {
                bf16x8 o0, o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o0[e] = (bf16)(relu ? fmaxf(a[e], 0.f) : a[e]);
                    o1[e] = (bf16)(relu ? fmaxf(a[8 + e], 0.f) : a[8 + e]);
                }
                *(bf16x8*)(Y + tok * N + n0 + 8 * h) = o0;
                *(bf16x8*)(Y + tok * N + n0 + 16 + 8 * h) = o1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next chunk's DMA has landed (this wave's part; also drains the stores)
        __syncthreads();                                            // ... everyone's; and everyone is done with this buffer
    }
}

template <int C, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ts_proj_add_ln_kernel(const bf16* __restrict__ O, const bf16* __restrict__ W, const bf16* __restrict__ bias,
                                                                    const bf16* __restrict__ X, const float* __restrict__ dp, int64_t rows_per_sample,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    bf16* __restrict__ X1, bf16* __restrict__ N2, float* __restrict__ mean_out,
                                                                    float* __restrict__ rstd_out, int64_t T, float eps) {
    using I = WImg<C, C>;
    constexpr int KS = C / 16, CT = C / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tok = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32 + r;
    const int64_t tokc = tok < T ? tok : T - 1;
    dma_image<C, C, WAVES>(W, C, smem, wave, lane);
    bf16x8 of[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) of[s] = *(const bf16x8*)(O + tokc * C + 16 * s + 8 * h);
    // the residual row of this lane's channels (two 16-byte pieces per tile), in flight under the MFMAs
    bf16x8 xr[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        xr[ct][0] = *(const bf16x8*)(X + tokc * C + 32 * ct + 8 * h);
        xr[ct][1] = *(const bf16x8*)(X + tokc * C + 32 * ct + 16 + 8 * h);
    }
    const float sc = dp ? dp[tokc / rows_per_sample] : 1.f;
    const int pr = pi16(r & 15) | (r & 16);
#pragma unroll
    for (int s = 0; s < KS; ++s) asm volatile("" :: "v"(of[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bf16* Ws = (const bf16*)smem;
    float v[CT][16];
    float sum = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        f32x16 a;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = bias ? (float)bias[32 * ct + tile_ch(q, h)] : 0.f;
        const bf16* wrow = Ws + (32 * ct + pr) * I::RS + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) a = mfma32(*(const bf16x8*)(wrow + 16 * s), of[s], a);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float y = (float)(bf16)a[q];                                   // the Linear's output is a 16-bit tensor
            const float x1 = (float)(bf16)((float)xr[ct][q >> 3][q & 7] + sc * y);  // ... and so is the residual stream
            v[ct][q] = x1;
            sum += x1;
        }
    }
    sum += __shfl_xor(sum, 32);                                     // the token's other 16 channels of every tile live in lane r ^ 32
    const float mean = sum / (float)C;
    float qs = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) { const float d = v[ct][q] - mean; qs += d * d; }
    qs += __shfl_xor(qs, 32);
    const float rstd = rsqrtf(qs / (float)C + eps);
    if (tok >= T) return;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c0 = 32 * ct + 16 * half + 8 * h;
            const float4 g0 = *(const float4*)(gamma + c0), g1 = *(const float4*)(gamma + c0 + 4);
            const float4 b0 = *(const float4*)(beta + c0), b1 = *(const float4*)(beta + c0 + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            bf16x8 o1, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x1 = v[ct][8 * half + e];
                o1[e] = (bf16)x1;
                o2[e] = (bf16)((x1 - mean) * rstd * gg[e] + bb[e]);
            }
            *(bf16x8*)(X1 + tok * C + c0) = o1;
            *(bf16x8*)(N2 + tok * C + c0) = o2;
        }
    if (h == 0) { mean_out[tok] = mean; rstd_out[tok] = rstd; }
}

template <int C, int WAVES, int CH>
int launch_linear(const void* x, const void* w, const void* bias, void* y, int64_t T, int N, int relu, hipStream_t s) {
    if (N % CH != 0) return SWIN_ERR_UNSUPPORTED;
    const size_t lds = 2 * (size_t)WImg<CH, C>::BYTES + (size_t)N * sizeof(float);
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    auto kern = ts_linear_kernel<C, WAVES, CH>;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const unsigned blocks = (unsigned)((T + WAVES * 32 - 1) / (WAVES * 32));
    // one block per CU (LDS: two chunk images): with fewer token blocks than CUs the output chunks are dealt over blockIdx.y
    unsigned ny = blocks >= 256 ? 1u : 256u / blocks;
    if (ny > (unsigned)(N / CH)) ny = (unsigned)(N / CH);
    if (ny < 1) ny = 1;
    kern<<<dim3(blocks, ny), WAVES * 64, lds, s>>>((const bf16*)x, (const bf16*)w, (const bf16*)bias, (bf16*)y, T, N, relu);
    return swin_launch_status();
}

template <int C, int WAVES>
int launch_proj(const void* o, const void* w, const void* bias, const void* x, const float* dp, int64_t rps, const float* gamma, const float* beta,
                void* x1, void* n2, float* mean, float* rstd, int64_t T, float eps, hipStream_t s) {
    const size_t lds = (size_t)WImg<C, C>::BYTES;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    auto kern = ts_proj_add_ln_kernel<C, WAVES>;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const unsigned blocks = (unsigned)((T + WAVES * 32 - 1) / (WAVES * 32));
    kern<<<blocks, WAVES * 64, lds, s>>>((const bf16*)o, (const bf16*)w, (const bf16*)bias, (const bf16*)x, dp, rps, gamma, beta, (bf16*)x1, (bf16*)n2,
                                         mean, rstd, T, eps);
    return swin_launch_status();
}

}  // namespace

// y (T, N) = [relu](x (T, C) w (N, C)^T + bias (N; 16-bit, may be NULL)).  C in {96, 128, 192, 256, 384}, N a multiple of 64 (of 96 / 128 at
// C = 96, 192 / 128 the whole-chunk variants are used: the qkv projection's 3C); else SWIN_ERR_UNSUPPORTED.
extern "C" int swin_ts_linear_bf16(const void* x, const void* w, const void* bias, void* y, int64_t T, int N, int C, int relu, void* stream) {
    if (T == 0) return SWIN_OK;
    if (!x || !w || !y || T < 0 || N <= 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 96: return N % 96 == 0 ? launch_linear<96, 8, 96>(x, w, bias, y, T, N, relu, s) : launch_linear<96, 8, 64>(x, w, bias, y, T, N, relu, s);
        case 128: return N % 128 == 0 ? launch_linear<128, 8, 128>(x, w, bias, y, T, N, relu, s) : launch_linear<128, 8, 64>(x, w, bias, y, T, N, relu, s);
        case 192: return N % 96 == 0 ? launch_linear<192, 8, 96>(x, w, bias, y, T, N, relu, s) : launch_linear<192, 8, 64>(x, w, bias, y, T, N, relu, s);
        case 256: return launch_linear<256, 8, 64>(x, w, bias, y, T, N, relu, s);
        case 384: return launch_linear<384, 8, 64>(x, w, bias, y, T, N, relu, s);      // stage 3 of Swin-T / -S: 96 fragment registers per lane
        default: return SWIN_ERR_UNSUPPORTED;
    }
}

// x1 (T, C) = x + dp[row / rows_per_sample] * (o w^T + bias)   (dp NULL: 1);   n2 = LayerNorm(x1; gamma, beta, eps);  mean / rstd (T) f32.
// o, x, x1, n2 16-bit; w (C, C), bias (C) 16-bit (bias may be NULL); gamma, beta f32.  C in {96, 128, 192, 256}.
extern "C" int swin_ts_proj_add_ln_bf16(const void* o, const void* w, const void* bias, const void* x, const float* dp, int64_t rows_per_sample,
                                        const float* gamma, const float* beta, void* x1, void* n2, float* mean, float* rstd, int64_t T, int C,
                                        float eps, void* stream) {
    if (T == 0) return SWIN_OK;
    if (!o || !w || !x || !gamma || !beta || !x1 || !n2 || !mean || !rstd || T < 0 || rows_per_sample <= 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 96: return launch_proj<96, 8>(o, w, bias, x, dp, rows_per_sample, gamma, beta, x1, n2, mean, rstd, T, eps, s);
        case 128: return launch_proj<128, 8>(o, w, bias, x, dp, rows_per_sample, gamma, beta, x1, n2, mean, rstd, T, eps, s);
        case 192: return launch_proj<192, 4>(o, w, bias, x, dp, rows_per_sample, gamma, beta, x1, n2, mean, rstd, T, eps, s);
        case 256: return launch_proj<256, 4>(o, w, bias, x, dp, rows_per_sample, gamma, beta, x1, n2, mean, rstd, T, eps, s);
        default: return SWIN_ERR_UNSUPPORTED;
    }
}
