// Token-stationary fused MLP for gfx950 (Mlp.forward swin_transformer.py:32-38 and its backward).
//
//   forward :  y = fc2(gelu(fc1(x) + b1)) + b2                 one launch, the 4C hidden activation never leaves the CU
//   backward:  hpre recomputed from x;  dh = dy W2;  dhpre = dh * gelu'(hpre);  dx = dhpre W1
//              h and dhpre are written once (bf16) for the two weight-gradient GEMMs -- nothing else of size T x 4C moves
//
// Decomposition.  A wave OWNS 32 tokens for the whole kernel and computes everything TRANSPOSED:
//     Ht[hid x tok]  = W1[hid x C]  . Xt[C x tok]        A = weight rows from LDS,  B = the wave's token fragments (registers)
//     Yt[C   x tok] += W2[C x hid]  . gelu(Ht)           B = the accumulator tile of the first product, as it stands
// A 32x32 accumulator of v_mfma_f32_32x32x16_bf16 has its column (token) on the lane and its rows in the registers, so it IS
// the B operand of a product that sums over its rows (guide: "an accumulator tile as the next MFMA's operand"): the hidden
// tile goes from the first GEMM to the second through 8 cvt_pk per lane -- no LDS round trip, no barrier, no cross-wave
// dependency.  Only the weights are shared: a block stages them chunk by chunk (CH hidden units) into LDS.
// The k order of an accumulator-fed operand is permuted (row 16s + 8(j>>2) + 4h + (j&3) for element j of lane half h); the
// permutation is absorbed by which weight ROW a lane feeds to the first product (pi = swap bits 2,3 of the row index):
// the second product then reads its weight fragment with one natural 16-byte LDS read, and every lane ends up holding
// 8 CONTIGUOUS channels / hidden units per register octet -> all global stores are 16-byte.
// LDS rows are padded so that (row stride / 16 B) is odd: ds_read_b128 of 16 different rows is conflict-free.
//
// Roofline: 16 T C^2 flops over 2 T C bpe bytes -> MFMA-bound on paper (AI = 4C >= 384); at C = 96 the exact-erf GELU
// (one v_exp + one v_rcp + 14 plain VALU per element) makes the VALU pipe the co-limit.
#include <cstdlib>

#include "common.h"
#include "ts_common.h"

namespace {

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7) on z = |x| / sqrt(2); returns q = 1 - erf(z) = poly(t) * exp(-z^2)
// and e = exp(-x^2 / 2).  gelu(x) = x * Phi(x),  Phi(x) = x > 0 ? 1 - q/2 : q/2;  gelu'(x) = Phi(x) + x e / sqrt(2 pi).
__device__ __forceinline__ void erfc_parts(float x, float& q, float& e) {
    const float z = fabsf(x) * 0.70710678118f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    e = __builtin_amdgcn_exp2f(-z * z * 1.44269504089f);
    q = p * e;
}
__device__ __forceinline__ float gelu_f(float x) {
    float q, e;
    erfc_parts(x, q, e);
    const float cdf = x > 0.f ? fmaf(-0.5f, q, 1.0f) : 0.5f * q;
    return x * cdf;
}
__device__ __forceinline__ void gelu_fg(float x, float& g, float& dg) {
    float q, e;
    erfc_parts(x, q, e);
    const float cdf = x > 0.f ? fmaf(-0.5f, q, 1.0f) : 0.5f * q;
    g = x * cdf;
    dg = fmaf(x * e, 0.3989422804f, cdf);
}

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int s) {
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (bf16)a[8 * s + e];
    return f;
}

// transposed A fragment from a row-major [k][n] LDS image: lane (r = lane&31, h = lane>>5) gets img[k0 + 8h + e][n0 + pi(r)],
// e = 0..7 (two ds_read_b64_tr_b16; the pi column permutation is a permutation of the 4-column quads a lane addresses)
__device__ __forceinline__ bf16x8 tr_frag_pi(const bf16* img, int stride, int k0, int n0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int h = g >> 1, dh = g & 1;
    const int pp = ((p & 1) << 1) | (p >> 1);
    const bf16* a0 = img + (k0 + 8 * h + q) * stride + n0 + 16 * dh + 4 * pp;
    bf16x4 lo = SWIN_DS_READ_TR16((lds_bf16x4*)a0);
    bf16x4 hi = SWIN_DS_READ_TR16((lds_bf16x4*)(a0 + 4 * stride));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = lo[e]; f[4 + e] = hi[e]; }
    return f;
}

template <int C, int CH> struct TsGeom {
    static constexpr int HID = 4 * C;
    static constexpr int NCH = HID / CH;
    static constexpr int KS = C / 16;          // k-steps over the channels
    static constexpr int CT = C / 32;          // 32-row output tiles over the channels
    static constexpr int HT = CH / 32;         // hidden tiles per chunk
    static_assert(C % 32 == 0 && CH % 32 == 0 && HID % CH == 0, "tile sizes");
};

// ------------------------------------------------------------------------------------------------ forward
// Optional epilogue of the forward kernel (x1 != null): the block's second residual and the NEXT LayerNorm in the same launch --
//   x2 = x1 + dp[row / rows_per_sample] * y;   nn = LayerNorm(x2; gamma, beta)   (swin_transformer.py:253 and :211 of the next block)
// -- instead of storing y for a separate swin_add_layernorm_fwd.  A token's output row sits in lanes r and r + 32 (16 channels per
// 32-channel tile each), so the statistics cost one cross-lane add each.  gamma == null: residual only (x2 stored, no norm).
struct MlpEpi {
    const bf16* x1; const float* dp; int64_t rows_per_sample; const float* gamma; const float* beta;
    bf16* x2; bf16* nn; float* mean; float* rstd; float eps;
};

template <int C, int WAVES, int CH, int OCC, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64, OCC) void ts_mlp_fwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ W1,
                                                               const float* __restrict__ b1, const bf16* __restrict__ W2,
                                                               const float* __restrict__ b2, bf16* __restrict__ Y, int64_t T, MlpEpi E) {
    using G = TsGeom<C, CH>;
    using I1 = WImg<CH, C>;                                          // W1 chunk: CH hidden rows x C
    using I2 = WImg<C, CH>;                                          // W2 chunk: C rows x CH hidden
    constexpr int NT = WAVES * 64;
    constexpr int BUF = I1::BYTES + I2::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* b1s = (float*)(smem + 2 * BUF);                           // [4C]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tok = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32 + r;
    const int64_t tokc = tok < T ? tok : T - 1;

    auto stage = [&](int j, int buf) {
        dma_image<CH, C, WAVES>(W1 + (int64_t)j * CH * C, C, smem + buf * BUF, wave, lane);
        dma_image<C, CH, WAVES>(W2 + (int64_t)j * CH, G::HID, smem + buf * BUF + I1::BYTES, wave, lane);
    };
    stage(0, 0);
    for (int i = tid; i < G::HID; i += NT) b1s[i] = b1[i];
    bf16x8 xf[G::KS];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) xf[s] = *(const bf16x8*)(X + tokc * C + 16 * s + 8 * h);
    f32x16 yacc[G::CT];
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) yacc[ct][q] = 0.f;
    const int pr = pi16(r & 15) | (r & 16);
    // the token fragments are consumed here, once: hipcc then waits for them HERE and not at their first MFMA inside the
    // chunk loop, where its vmcnt would also drain the (invisible to it, younger) DMA of the next chunk every iteration
#pragma unroll
    for (int s = 0; s < G::KS; ++s) asm volatile("" :: "v"(xf[s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int j = 0; j < ((ABL & 16) ? 0 : G::NCH); ++j) {
        if (j + 1 < G::NCH && !(ABL & 8)) stage(j + 1, (j + 1) & 1);              // lands under this chunk's MFMAs
        if ((ABL & 32) && wave >= WAVES / 2) __builtin_amdgcn_s_sleep(9);      // stagger the SIMD partners by ~half a tile
        const bf16* W1s = (const bf16*)(smem + (j & 1) * BUF);
        const bf16* W2s = (const bf16*)(smem + (j & 1) * BUF + I1::BYTES);
        // software pipeline over the hidden tiles: the W1 fragments of tile t+1 and the W2 fragments of tile t are read from
        // LDS while the GELU of tile t runs on the VALU (hipcc otherwise issues read -> wait -> MFMA one fragment at a time)
        bf16x8 wf[G::KS];
        {
            const bf16* w1row = W1s + pr * I1::RS + 8 * h;
#pragma unroll
            for (int s = 0; s < G::KS; ++s) wf[s] = *(const bf16x8*)(w1row + 16 * s);
        }
#pragma unroll 1
        for (int t = 0; t < G::HT; ++t) {
            // ---- Ht tile = W1 rows (pi order) . Xt, bias as the initial accumulator ----
            f32x16 a;
            {
                const float4* bp = (const float4*)&b1s[j * CH + 32 * t + 8 * h];
                const float4 v0 = bp[0], v1 = bp[1], v2 = bp[4], v3 = bp[5];      // hidden 8h..8h+7 and 16+8h..16+8h+7
                a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
                a[8] = v2.x; a[9] = v2.y; a[10] = v2.z; a[11] = v2.w; a[12] = v3.x; a[13] = v3.y; a[14] = v3.z; a[15] = v3.w;
            }
#pragma unroll
            for (int s = 0; s < G::KS; ++s) a = mfma32(wf[s], xf[s], a);
            bf16x8 w2f[G::CT][2];
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct) {
                const bf16* w2row = W2s + (32 * ct + pr) * I2::RS + 32 * t + 8 * h;
                w2f[ct][0] = *(const bf16x8*)(w2row);
                w2f[ct][1] = *(const bf16x8*)(w2row + 16);
            }
            {
                const int tn = t + 1 < G::HT ? t + 1 : t;
                const bf16* w1row = W1s + (32 * tn + pr) * I1::RS + 8 * h;
#pragma unroll
                for (int s = 0; s < G::KS; ++s) wf[s] = *(const bf16x8*)(w1row + 16 * s);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- GELU on the accumulators, packed straight into the next product's B fragments ----
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = (ABL & 1) ? a[q] * 0.5f : gelu_f(a[q]);
            const bf16x8 pf0 = pack8(a, 0), pf1 = pack8(a, 1);
            // ---- Yt += W2 rows (pi order) . gelu(Ht) ----
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct) {
                yacc[ct] = mfma32(w2f[ct][0], pf0, yacc[ct]);
                yacc[ct] = mfma32(w2f[ct][1], pf1, yacc[ct]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the next chunk's DMA has landed (this wave's part)
        __syncthreads();                                            // ... everyone's; and everyone is done with this buffer
    }
    if (E.x1) {
        // ---- residual + next LayerNorm epilogue.  Rounding points of the three-launch chain: y to 16 bits, x2 to 16 bits, statistics
        // of the rounded x2, two-pass variance (csrc/layernorm.hip)
        const float sc = E.dp ? E.dp[tokc / E.rows_per_sample] : 1.f;
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c0 = 32 * ct + 16 * s + 8 * h;
                const bf16x8 xr = *(const bf16x8*)(E.x1 + tokc * C + c0);
                const float4 ba = *(const float4*)(b2 + c0), bb = *(const float4*)(b2 + c0 + 4);
                const float bv[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = (float)(bf16)(yacc[ct][8 * s + e] + bv[e]);
                    const float x2 = (float)(bf16)((float)xr[e] + sc * y);
                    yacc[ct][8 * s + e] = x2;
                    sum += x2;
                }
            }
        sum += __shfl_xor(sum, 32);
        const float mean = sum / (float)C;
        float qs = 0.f;
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int q = 0; q < 16; ++q) { const float d = yacc[ct][q] - mean; qs += d * d; }
        qs += __shfl_xor(qs, 32);
        const float rstd = rsqrtf(qs / (float)C + E.eps);
        if (tok >= T) return;
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c0 = 32 * ct + 16 * s + 8 * h;
                bf16x8 o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) o1[e] = (bf16)yacc[ct][8 * s + e];
                *(bf16x8*)(E.x2 + tok * C + c0) = o1;
                if (E.gamma) {
                    const float4 g0 = *(const float4*)(E.gamma + c0), g1 = *(const float4*)(E.gamma + c0 + 4);
                    const float4 t0 = *(const float4*)(E.beta + c0), t1 = *(const float4*)(E.beta + c0 + 4);
                    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                    const float tt[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                    bf16x8 o2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o2[e] = (bf16)((yacc[ct][8 * s + e] - mean) * rstd * gg[e] + tt[e]);
                    *(bf16x8*)(E.nn + tok * C + c0) = o2;
                }
            }
        if (E.gamma && h == 0) { E.mean[tok] = mean; E.rstd[tok] = rstd; }
        return;
    }
    // ---- epilogue: + b2, bf16, two 16-byte stores per channel tile (registers 0..7 = channels 8h.., 8..15 = 16+8h..) ----
    if (tok < T) {
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c0 = 32 * ct + 16 * s + 8 * h;
                const float4 ba = *(const float4*)(b2 + c0), bb = *(const float4*)(b2 + c0 + 4);
                bf16x8 o;
                o[0] = (bf16)(yacc[ct][8 * s + 0] + ba.x); o[1] = (bf16)(yacc[ct][8 * s + 1] + ba.y);
                o[2] = (bf16)(yacc[ct][8 * s + 2] + ba.z); o[3] = (bf16)(yacc[ct][8 * s + 3] + ba.w);
                o[4] = (bf16)(yacc[ct][8 * s + 4] + bb.x); o[5] = (bf16)(yacc[ct][8 * s + 5] + bb.y);
                o[6] = (bf16)(yacc[ct][8 * s + 6] + bb.z); o[7] = (bf16)(yacc[ct][8 * s + 7] + bb.w);
                *(bf16x8*)(Y + tok * C + c0) = o;
            }
        }
    }
}

template <int C, int WAVES, int CH, int OCC, int ABL = 0>
int launch_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int64_t T, hipStream_t s,
               MlpEpi epi = MlpEpi{}) {
    using G = TsGeom<C, CH>;
    const size_t lds = 2 * (size_t)(WImg<CH, C>::BYTES + WImg<C, CH>::BYTES) + G::HID * sizeof(float);
    static bool attr_set[16] = {};
    int dev = 0;
    hipGetDevice(&dev);
    auto kern = ts_mlp_fwd_kernel<C, WAVES, CH, OCC, ABL>;
    if (dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const unsigned blocks = (unsigned)((T + WAVES * 32 - 1) / (WAVES * 32));
    kern<<<blocks, WAVES * 64, lds, s>>>((const bf16*)x, (const bf16*)w1, b1, (const bf16*)w2, b2, (bf16*)y, T, epi);
    return swin_launch_status();
}


// ------------------------------------------------------------------------------------------------ backward
// Per hidden tile (32 hidden units x the wave's 32 tokens), everything transposed as in the forward:
//   Hpre_t = W1 rows (pi) . Xt + b1          (recomputed: the forward saved nothing of size T x 4C)
//   dH_t   = W2^T rows (pi) . dYt             A = transposed read of the W2 chunk image [C][CH]
//   h = gelu(Hpre), dhpre = dH * gelu'(Hpre)  -> both written once, bf16, 16-byte stores (8 contiguous hidden units per lane)
//   dXt   += W1^T rows (pi) . dhpre           A = transposed read of the W1 chunk image [CH][C]; B = the dhpre accumulators
// Optional epilogue of the backward kernel (x1 != null): the backward of norm2 and of the first residual (swin_transformer.py:252)
// on the data gradient dn2 the kernel has just produced, instead of storing dn2 for a separate swin_layernorm_bwd:
//   xh = (x1 - mean) rstd;  g = dn2 gamma;  dx = rstd (g - mean_c(g) - xh mean_c(g xh)) + dres;  dy = dx * dp[row / rows_per_sample]
// and the block's partial sums of dgamma = sum_t dn2 xh, dbeta = sum_t dn2 as one row [dgamma | dbeta] of `partials` per thread
// block (the block runner's tail launch adds the rows).  dn2 is rounded to 16 bits first, as the stored tensor was.
struct MlpBwdEpi {
    const bf16* x1; const float* mean; const float* rstd; const float* gamma; const bf16* dres; const float* dp; int64_t rows_per_sample;
    bf16* dx; bf16* dy; float* partials;
    // Optional PROLOGUE (p_dnn != null): the backward of the block's NEXT norm and of its second residual (swin_transformer.py:253 and
    // the following :211) produces the kernel's dY operand instead of reading it:
    //   dx1 = LayerNorm-backward(dnn; x2, mean3, rstd3, gamma3) + dres3;   dY = dx1 * dp1[row / rows_per_sample]
    // dx1 is stored (p_dx1: it is `dres` of the epilogue), dY is stored when it differs from dx1 (p_dy2 != null: the fc2 weight
    // gradient reads it), and the block's [dgamma3 | dbeta3] partial row goes to p_partials.
    const bf16* p_dnn; const bf16* p_x2; const float* p_mean; const float* p_rstd; const float* p_gamma; const bf16* p_dres;
    const float* p_dp; bf16* p_dx1; bf16* p_dy2; float* p_partials;
};

template <int C, int WAVES, int CH, int OCC>
__global__ __launch_bounds__(WAVES * 64, OCC) void ts_mlp_bwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY,
                                                               const bf16* __restrict__ W1, const float* __restrict__ b1,
                                                               const bf16* __restrict__ W2, bf16* __restrict__ dX,
                                                               bf16* __restrict__ Hout, bf16* __restrict__ dHpre, int64_t T, MlpBwdEpi E) {
    using G = TsGeom<C, CH>;
    using I1 = WImg<CH, C>;
    using I2 = WImg<C, CH>;
    constexpr int NT = WAVES * 64;
    constexpr int BUF = I1::BYTES + I2::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* b1s = (float*)(smem + 2 * BUF);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tok = (int64_t)blockIdx.x * (WAVES * 32) + wave * 32 + r;
    const int64_t tokc = tok < T ? tok : T - 1;
    auto stage = [&](int j, int buf) {
        dma_image<CH, C, WAVES>(W1 + (int64_t)j * CH * C, C, smem + buf * BUF, wave, lane);
        dma_image<C, CH, WAVES>(W2 + (int64_t)j * CH, G::HID, smem + buf * BUF + I1::BYTES, wave, lane);
    };
    bf16x8 xf[G::KS], df[G::KS];
    if (E.p_dnn) {
        // ---- prologue (before the first weight DMA: the reduction scratch below is the weight images' LDS)
        // A lane's fragment pieces are channels 16 s + 8 h .. + 7 of its token, s = 0 .. KS - 1: half of the row, the other half in lane
        // r ^ 32 -- the same split as the accumulator layout of the epilogue (channel tile ct, half s' <-> s = 2 ct + s').
        float* red = (float*)smem + wave * (2 * 32 * 33);
        float* wsum = (float*)smem + WAVES * (2 * 32 * 33);
        const bool live = tok < T;
        const float m3 = E.p_mean[tokc], rs3 = E.p_rstd[tokc];
        const float sc1 = E.p_dp ? E.p_dp[tokc / E.rows_per_sample] : 1.f;
        float g[G::KS][8];
        float s1 = 0.f, s2 = 0.f;
        float colsum[G::CT];
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct) {
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                const int s = 2 * ct + sp, c0 = 16 * s + 8 * h;
                const bf16x8 dn = *(const bf16x8*)(E.p_dnn + tokc * C + c0);
                const bf16x8 xr = *(const bf16x8*)(E.p_x2 + tokc * C + c0);
                const float4 g0 = *(const float4*)(E.p_gamma + c0), g1 = *(const float4*)(E.p_gamma + c0 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = live ? (float)dn[e] : 0.f;
                    const float xh = ((float)xr[e] - m3) * rs3;
                    const float gv = d * gg[e];
                    s1 += gv; s2 += gv * xh;
                    g[s][e] = gv;
                    red[r * 33 + 16 * sp + 8 * h + e] = d * xh;
                    red[32 * 33 + r * 33 + 16 * sp + 8 * h + e] = d;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const float* col = red + (lane >> 5) * (32 * 33) + (lane & 31);
                float a = 0.f;
#pragma unroll
                for (int tkn = 0; tkn < 32; ++tkn) a += col[tkn * 33];
                colsum[ct] = a;
            }
            __builtin_amdgcn_wave_barrier();
        }
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        s1 /= (float)C; s2 /= (float)C;
#pragma unroll
        for (int s = 0; s < G::KS; ++s) {
            const int c0 = 16 * s + 8 * h;
            const bf16x8 xr = *(const bf16x8*)(E.p_x2 + tokc * C + c0);
            bf16x8 rr;
            if (E.p_dres) rr = *(const bf16x8*)(E.p_dres + tokc * C + c0);
            bf16x8 o, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = ((float)xr[e] - m3) * rs3;
                float d = rs3 * (g[s][e] - s1 - xh * s2);
                if (E.p_dres) d += (float)rr[e];
                o[e] = (bf16)d;
                o2[e] = (bf16)(d * sc1);
            }
            if (live) {
                *(bf16x8*)(E.p_dx1 + tok * C + c0) = o;
                if (E.p_dy2) *(bf16x8*)(E.p_dy2 + tok * C + c0) = o2;
            }
            df[s] = o2;
        }
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct) wsum[wave * (2 * C) + (lane >> 5) * C + 32 * ct + (lane & 31)] = colsum[ct];
        __syncthreads();
        for (int i = tid; i < 2 * C; i += NT) {
            float a = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WAVES; ++w2) a += wsum[w2 * (2 * C) + i];
            E.p_partials[(int64_t)blockIdx.x * (2 * C) + i] = a;
        }
        __syncthreads();                                            // the scratch becomes the weight images
    }
    stage(0, 0);
    for (int i = tid; i < G::HID; i += NT) b1s[i] = b1[i];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
        xf[s] = *(const bf16x8*)(X + tokc * C + 16 * s + 8 * h);
        if (!E.p_dnn) df[s] = *(const bf16x8*)(dY + tokc * C + 16 * s + 8 * h);
    }
    f32x16 xacc[G::CT];
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) xacc[ct][q] = 0.f;
    const int pr = pi16(r & 15) | (r & 16);
#pragma unroll
    for (int s = 0; s < G::KS; ++s) { asm volatile("" :: "v"(xf[s])); asm volatile("" :: "v"(df[s])); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int j = 0; j < G::NCH; ++j) {
        if (j + 1 < G::NCH) stage(j + 1, (j + 1) & 1);
        const bf16* W1s = (const bf16*)(smem + (j & 1) * BUF);
        const bf16* W2s = (const bf16*)(smem + (j & 1) * BUF + I1::BYTES);
#pragma unroll 1
        for (int t = 0; t < G::HT; ++t) {
            f32x16 a, d;
            {
                const float4* bp = (const float4*)&b1s[j * CH + 32 * t + 8 * h];
                const float4 v0 = bp[0], v1 = bp[1], v2 = bp[4], v3 = bp[5];
                a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
                a[8] = v2.x; a[9] = v2.y; a[10] = v2.z; a[11] = v2.w; a[12] = v3.x; a[13] = v3.y; a[14] = v3.z; a[15] = v3.w;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] = 0.f;
            const bf16* w1row = W1s + (32 * t + pr) * I1::RS + 8 * h;
#pragma unroll
            for (int s = 0; s < G::KS; ++s) a = mfma32(*(const bf16x8*)(w1row + 16 * s), xf[s], a);
            // dHt: A[row = hidden pi(r)][k = channel 16 s + 8 h + e] = W2s[channel][hidden]
#pragma unroll
            for (int s = 0; s < G::KS; ++s) d = mfma32(tr_frag_pi(W2s, I2::RS, 16 * s, 32 * t, lane), df[s], d);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float g, dg;
                gelu_fg(a[q], g, dg);
                a[q] = g;
                d[q] *= dg;
            }
            const bf16x8 h0 = pack8(a, 0), h1 = pack8(a, 1), p0 = pack8(d, 0), p1 = pack8(d, 1);
            if (tok < T) {
                const int64_t o = tok * G::HID + j * CH + 32 * t + 8 * h;        // registers 0..7: hidden 8h.., 8..15: 16+8h..
                *(bf16x8*)(Hout + o) = h0;
                *(bf16x8*)(Hout + o + 16) = h1;
                *(bf16x8*)(dHpre + o) = p0;
                *(bf16x8*)(dHpre + o + 16) = p1;
            }
            // dXt += W1^T rows (pi) . dhpre: A[row = channel pi(r)][k = hidden 16 s2 + 8 h + e] = W1s[hidden][channel]
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct) {
                xacc[ct] = mfma32(tr_frag_pi(W1s, I1::RS, 32 * t, 32 * ct, lane), p0, xacc[ct]);
                xacc[ct] = mfma32(tr_frag_pi(W1s, I1::RS, 32 * t + 16, 32 * ct, lane), p1, xacc[ct]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (E.x1) {
        // (the weight images are dead: the last chunk's barrier has passed) per-wave reduction scratch: [2 arrays][32 tokens][33] floats
        float* red = (float*)smem + wave * (2 * 32 * 33);
        float* wsum = (float*)smem + WAVES * (2 * 32 * 33);               // [WAVES][2 C] column sums of the waves
        const bool live = tok < T;
        const float m = E.mean[tokc], rs = E.rstd[tokc];
        const float sc = E.dp ? E.dp[tokc / E.rows_per_sample] : 1.f;
        float s1 = 0.f, s2 = 0.f;
        float colsum[G::CT];
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct) {
            float ag[16], ab[16];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int c0 = 32 * ct + 16 * s + 8 * h;
                const bf16x8 xr = *(const bf16x8*)(E.x1 + tokc * C + c0);
                const float4 g0 = *(const float4*)(E.gamma + c0), g1 = *(const float4*)(E.gamma + c0 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = live ? (float)(bf16)xacc[ct][8 * s + e] : 0.f;
                    const float xh = ((float)xr[e] - m) * rs;
                    ag[8 * s + e] = d * xh; ab[8 * s + e] = d;
                    const float g = d * gg[e];
                    s1 += g; s2 += g * xh;
                    xacc[ct][8 * s + e] = g;
                }
            }
            // column sums over the wave's 32 tokens of this tile's 32 channels: through LDS (rows padded to 33 floats)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int ch = (q < 8 ? 8 * h + q : 16 + 8 * h + (q - 8));
                red[r * 33 + ch] = ag[q];
                red[32 * 33 + r * 33 + ch] = ab[q];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const float* col = red + (lane >> 5) * (32 * 33) + (lane & 31);      // lanes 0-31: dgamma of channel lane; 32-63: dbeta
                float a = 0.f;
#pragma unroll
                for (int tkn = 0; tkn < 32; ++tkn) a += col[tkn * 33];
                colsum[ct] = a;
            }
            __builtin_amdgcn_wave_barrier();                        // the scratch is rewritten for the next tile
        }
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);          // the token's other half of every tile lives in lane r ^ 32
        s1 /= (float)C; s2 /= (float)C;
        if (live) {
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int c0 = 32 * ct + 16 * s + 8 * h;
                    const bf16x8 xr = *(const bf16x8*)(E.x1 + tok * C + c0);
                    bf16x8 rr;
                    if (E.dres) rr = *(const bf16x8*)(E.dres + tok * C + c0);
                    bf16x8 o, o2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xh = ((float)xr[e] - m) * rs;
                        float d = rs * (xacc[ct][8 * s + e] - s1 - xh * s2);
                        if (E.dres) d += (float)rr[e];
                        o[e] = (bf16)d;
                        o2[e] = (bf16)(d * sc);
                    }
                    *(bf16x8*)(E.dx + tok * C + c0) = o;
                    if (E.dy) *(bf16x8*)(E.dy + tok * C + c0) = o2;
                }
        }
        // the block's [dgamma | dbeta] partial row: waves -> LDS -> one row of `partials`
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct) wsum[wave * (2 * C) + (lane >> 5) * C + 32 * ct + (lane & 31)] = colsum[ct];
        __syncthreads();
        for (int i = tid; i < 2 * C; i += NT) {
            float a = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WAVES; ++w2) a += wsum[w2 * (2 * C) + i];
            E.partials[(int64_t)blockIdx.x * (2 * C) + i] = a;
        }
        return;
    }
    if (tok < T) {
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct) {
#pragma unroll
            for (int s = 0; s < 2; ++s) *(bf16x8*)(dX + tok * C + 32 * ct + 16 * s + 8 * h) = pack8(xacc[ct], s);
        }
    }
}

template <int C, int WAVES, int CH, int OCC>
int launch_bwd(const void* x, const void* dy, const void* w1, const float* b1, const void* w2, void* dx, void* hout, void* dhpre,
               int64_t T, hipStream_t s, MlpBwdEpi epi = MlpBwdEpi{}) {
    using G = TsGeom<C, CH>;
    const size_t lds = 2 * (size_t)(WImg<CH, C>::BYTES + WImg<C, CH>::BYTES) + G::HID * sizeof(float);
    static bool attr_set[16] = {};
    int dev = 0;
    hipGetDevice(&dev);
    auto kern = ts_mlp_bwd_kernel<C, WAVES, CH, OCC>;
    if (dev < 0 || dev >= 16) return SWIN_ERR_UNSUPPORTED;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SWIN_ERR_LAUNCH;
        attr_set[dev] = true;
    }
    const unsigned blocks = (unsigned)((T + WAVES * 32 - 1) / (WAVES * 32));
    static_assert(2 * (size_t)(WImg<CH, C>::BYTES + WImg<C, CH>::BYTES) >= (size_t)WAVES * (2 * 32 * 33 + 2 * C) * sizeof(float),
                  "the epilogue's reduction scratch reuses the weight images");
    kern<<<blocks, WAVES * 64, lds, s>>>((const bf16*)x, (const bf16*)dy, (const bf16*)w1, b1, (const bf16*)w2, (bf16*)dx, (bf16*)hout,
                                         (bf16*)dhpre, T, epi);
    return swin_launch_status();
}

}  // namespace

// y (T,C) = fc2(gelu(fc1(x))) : x (T,C) bf16, w1 (4C,C) bf16, b1 (4C) f32, w2 (C,4C) bf16, b2 (C) f32.  C in {96, 192}.
extern "C" int swin_mlp_fwd_bf16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int64_t T,
                                 int C, void* stream) {
    if (T == 0) return SWIN_OK;
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || T < 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 96: {
#ifdef SWIN_DEV      // ablation instantiations (some compute WRONG results on purpose): development builds only
            static const int abl = swin_dev_int("SWIN_MLP_ABL", 0);     // development ablations
            if (abl == 1) return launch_fwd<96, 8, 128, 2, 1>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 2) return launch_fwd<96, 4, 128, 1, 0>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 3) return launch_fwd<96, 4, 64, 1, 0>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 8) return launch_fwd<96, 8, 128, 2, 8>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 32) return launch_fwd<96, 8, 128, 2, 32>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 16) return launch_fwd<96, 8, 128, 2, 16>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 9) return launch_fwd<96, 8, 128, 2, 9>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 4) return launch_fwd<96, 4, 32, 3, 0>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 5) return launch_fwd<96, 4, 32, 4, 0>(x, w1, b1, w2, b2, y, T, s);
            if (abl == 6) return launch_fwd<96, 4, 32, 2, 0>(x, w1, b1, w2, b2, y, T, s);
#endif
            return launch_fwd<96, 8, 128, 2>(x, w1, b1, w2, b2, y, T, s);
        }
        case 192: return launch_fwd<192, 4, 64, 1>(x, w1, b1, w2, b2, y, T, s);
        default: return SWIN_ERR_UNSUPPORTED;
    }
}


// x, dy (T,C) bf16 -> dx (T,C), h (T,4C), dhpre (T,4C) bf16 (see include/swin_hip.h).  C in {96, 192}.
extern "C" int swin_mlp_bwd_bf16(const void* x, const void* dy, const void* w1, const float* b1, const void* w2, void* dx, void* h,
                                 void* dhpre, int64_t T, int C, void* stream) {
    if (T == 0) return SWIN_OK;
    if (!x || !dy || !w1 || !b1 || !w2 || !dx || !h || !dhpre || T < 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 96: return launch_bwd<96, 8, 128, 2>(x, dy, w1, b1, w2, dx, h, dhpre, T, s);
        case 192: return launch_bwd<192, 4, 64, 1>(x, dy, w1, b1, w2, dx, h, dhpre, T, s);
        default: return SWIN_ERR_UNSUPPORTED;
    }
}

// swin_mlp_fwd_bf16 with the block's second residual and the next LayerNorm in its epilogue (csrc/ts_mlp.hip, MlpEpi):
//   x2 (T,C) = x1 + dp[row / rows_per_sample] * Mlp(x);   nn = LayerNorm(x2; gamma, beta, eps), mean / rstd (T) f32 -- gamma NULL:
//   residual only (nn, mean, rstd unused).  C in {96, 192}.
extern "C" int swin_mlp_add_ln_fwd_bf16(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* x1,
                                        const float* dp, int64_t rows_per_sample, const float* gamma, const float* beta, void* x2, void* nn,
                                        float* mean, float* rstd, int64_t T, int C, float eps, void* stream) {
    if (T == 0) return SWIN_OK;
    if (!x || !w1 || !b1 || !w2 || !b2 || !x1 || !x2 || T < 0 || rows_per_sample <= 0) return SWIN_ERR_BAD_ARG;
    if (gamma && (!beta || !nn || !mean || !rstd)) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const MlpEpi epi{(const bf16*)x1, dp, rows_per_sample, gamma, beta, (bf16*)x2, (bf16*)nn, mean, rstd, eps};
    switch (C) {
        case 96: return launch_fwd<96, 8, 128, 2>(x, w1, b1, w2, b2, nullptr, T, s, epi);
        case 192: return launch_fwd<192, 4, 64, 1>(x, w1, b1, w2, b2, nullptr, T, s, epi);
        default: return SWIN_ERR_UNSUPPORTED;
    }
}

// rows of the `partials` workspace of swin_mlp_ln_bwd_bf16 (one [dgamma | dbeta] row of 2 C floats per thread block)
extern "C" int64_t swin_mlp_ln_bwd_partial_rows(int64_t T, int C) {
    if (T <= 0) return 0;
    return C == 96 ? (T + 255) / 256 : (C == 192 ? (T + 127) / 128 : 0);
}

// swin_mlp_bwd_bf16 with the backward of norm2 and of the first residual in its epilogue (MlpBwdEpi): dn2 is not stored; instead
//   dx (T,C) = LayerNorm-backward(dn2; x1, mean, rstd, gamma) + dres,   dy (T,C) = dx * dp[row / rows_per_sample] (dy NULL: not wanted;
//   dp NULL: scale 1), and `partials` (swin_mlp_ln_bwd_partial_rows(T, C) rows of 2 C floats) receives the per-block sums of
//   [dgamma | dbeta] -- add the rows (swin_tail_reduce kind 0).  h / dhpre as in swin_mlp_bwd_bf16.  C in {96, 192}.
extern "C" int swin_mlp_ln_bwd_bf16(const void* x, const void* dy2, const void* w1, const float* b1, const void* w2, void* h, void* dhpre,
                                    const void* x1, const float* mean, const float* rstd, const float* gamma, const void* dres, const float* dp,
                                    int64_t rows_per_sample, void* dx, void* dy, float* partials, int64_t T, int C, void* stream) {
    if (T == 0) return SWIN_OK;
    if (!x || !dy2 || !w1 || !b1 || !w2 || !h || !dhpre || !x1 || !mean || !rstd || !gamma || !dx || !partials || T < 0 || rows_per_sample <= 0)
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const MlpBwdEpi epi{(const bf16*)x1, mean, rstd, gamma, (const bf16*)dres, dp, rows_per_sample, (bf16*)dx, (bf16*)dy, partials,
                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    switch (C) {
        case 96: return launch_bwd<96, 8, 128, 2>(x, dy2, w1, b1, w2, nullptr, h, dhpre, T, s, epi);
        case 192: return launch_bwd<192, 4, 64, 1>(x, dy2, w1, b1, w2, nullptr, h, dhpre, T, s, epi);
        default: return SWIN_ERR_UNSUPPORTED;
    }
}

// swin_mlp_ln_bwd_bf16 with, in addition, the backward of the block's NEXT norm and second residual as its PROLOGUE (MlpBwdEpi p_*): the
// MLP half of a Swin block's backward in ONE launch.  dnn: gradient of the next norm's output; x2 / mean3 / rstd3 / gamma3: that norm's
// saved input, statistics and weight; dres3: gradient of the residual stream behind it (NULL: none); dp1: the second DropPath scale (NULL: 1).
// Writes dx1 (T,C) (and reads it back as the epilogue's residual gradient), dy2 (T,C) when dp1 != NULL (the fc2 weight gradient's
// operand; with dp1 NULL it equals dx1), partials3: swin_mlp_ln_bwd_partial_rows(T, C) rows of [dgamma3 | dbeta3].
extern "C" int swin_mlp_ln2_bwd_bf16(const void* x, const void* w1, const float* b1, const void* w2, void* h, void* dhpre, const void* x1,
                                     const float* mean, const float* rstd, const float* gamma, const float* dp, int64_t rows_per_sample, void* dx,
                                     void* dy, float* partials, const void* dnn, const void* x2, const float* mean3, const float* rstd3,
                                     const float* gamma3, const void* dres3, const float* dp1, void* dx1, void* dy2, float* partials3, int64_t T,
                                     int C, void* stream) {
    if (T == 0) return SWIN_OK;
    if (!x || !w1 || !b1 || !w2 || !h || !dhpre || !x1 || !mean || !rstd || !gamma || !dx || !partials || !dnn || !x2 || !mean3 || !rstd3 || !gamma3 ||
        !dx1 || !partials3 || (dp1 && !dy2) || T < 0 || rows_per_sample <= 0)
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const MlpBwdEpi epi{(const bf16*)x1, mean, rstd, gamma, (const bf16*)dx1, dp, rows_per_sample, (bf16*)dx, (bf16*)dy, partials,
                        (const bf16*)dnn, (const bf16*)x2, mean3, rstd3, gamma3, (const bf16*)dres3, dp1, (bf16*)dx1, dp1 ? (bf16*)dy2 : nullptr, partials3};
    switch (C) {
        case 96: return launch_bwd<96, 8, 128, 2>(x, nullptr, w1, b1, w2, nullptr, h, dhpre, T, s, epi);
        case 192: return launch_bwd<192, 4, 64, 1>(x, nullptr, w1, b1, w2, nullptr, h, dhpre, T, s, epi);
        default: return SWIN_ERR_UNSUPPORTED;
    }
}
