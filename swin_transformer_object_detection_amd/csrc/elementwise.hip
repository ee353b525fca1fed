// HBM-bound elementwise / gather kernels of the Swin block and the FPN top-down path.
// All of them move 16 bytes per lane per access (coalesced dwordx4).
#include <algorithm>
#include <cstdlib>

#include "common.h"

static inline int ew_blocks(int64_t n_items) {
    int64_t b = (n_items + 255) / 256;
    return (int)(b < 4096 ? (b > 0 ? b : 1) : 4096);
}

// ---------------------------------------------------------------- bias + GELU(erf)
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float v) {
    float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * __expf(-0.5f * v * v);
    return cdf + v * pdf;
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bias_gelu_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                        const T* __restrict__ dy, T* __restrict__ out, float* __restrict__ dbias,
                                                        int64_t nvec, int vec_per_row) {
    // The host sizes the grid so that (gridDim.x * 256) % vec_per_row == 0: a thread then visits ONE column slot
    // on every row, so the bias (forward/backward) and the bias-gradient partial sums (backward) live in registers;
    // the block combines them in LDS and issues one global atomic per channel.
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float colsum[];       // vec_per_row * VEC floats (BWD && dbias)
    const bool want_db = BWD && dbias != nullptr;
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int col = (int)(i0 % vec_per_row) * VEC;
    float bv[VEC], acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { bv[e] = bias ? bias[col + e] : 0.f; acc[e] = 0.f; }
    if (want_db) {
        for (int i = threadIdx.x; i < vec_per_row * VEC; i += 256) colsum[i] = 0.f;
        __syncthreads();
    }
    for (int64_t i = i0; i < nvec; i += (int64_t)gridDim.x * 256) {
        Vec16<T> v, o, g;
        v.load(x + i * VEC);
        if (BWD) g.load(dy + i * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float a = v.get(e) + bv[e];
            float r = BWD ? g.get(e) * gelu_grad_f(a) : gelu_f(a);
            o.set(e, r);
            if (BWD) acc[e] += r;
        }
        o.store(out + i * VEC);
    }
    if (want_db) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) atomicAdd(&colsum[col + e], acc[e]);
        __syncthreads();
        for (int i = threadIdx.x; i < vec_per_row * VEC; i += 256) atomicAdd(dbias + i, colsum[i]);
    }
}

// Backward with the bias gradient.  The flat mapping above would make EVERY block add to EVERY channel of dbias
// (2048 same-address fp32 atomics per channel: measured 3-10x over the HBM time).  Here a block owns a column chunk
// (cw 16-byte vectors) and a strided set of rows, keeps its column sums in registers, folds the 256/cw row lanes
// through LDS and issues cw*VEC atomics: a few hundred per channel, and rows * C / (rows per block) in total.
template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_bwd_cols_kernel(const T* __restrict__ x, const float* __restrict__ bias,
                                                                 const T* __restrict__ dy, T* __restrict__ out,
                                                                 float* __restrict__ dbias, int64_t rows, int vec_per_row, int cw) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red[256 * VEC];
    const int lc = threadIdx.x % cw, lr = threadIdx.x / cw, rpb = 256 / cw;
    const int cv = blockIdx.x * cw + lc;                       // column vector of this thread
    const int col = cv * VEC;
    float bv[VEC], acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { bv[e] = bias ? bias[col + e] : 0.f; acc[e] = 0.f; }
    const int64_t rstep = (int64_t)gridDim.y * rpb;
    int64_t r = (int64_t)blockIdx.y * rpb + lr;
    for (; r + rstep < rows; r += 2 * rstep) {                 // two independent rows in flight
        const int64_t i0 = r * vec_per_row + cv, i1 = (r + rstep) * vec_per_row + cv;
        Vec16<T> v0, g0, v1, g1, o0, o1;
        v0.load(x + i0 * VEC); g0.load(dy + i0 * VEC);
        v1.load(x + i1 * VEC); g1.load(dy + i1 * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float r0 = g0.get(e) * gelu_grad_f(v0.get(e) + bv[e]);
            float r1 = g1.get(e) * gelu_grad_f(v1.get(e) + bv[e]);
            o0.set(e, r0); o1.set(e, r1);
            acc[e] += r0 + r1;
        }
        o0.store(out + i0 * VEC); o1.store(out + i1 * VEC);
    }
    if (r < rows) {
        const int64_t i0 = r * vec_per_row + cv;
        Vec16<T> v0, g0, o0;
        v0.load(x + i0 * VEC); g0.load(dy + i0 * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float r0 = g0.get(e) * gelu_grad_f(v0.get(e) + bv[e]);
            o0.set(e, r0);
            acc[e] += r0;
        }
        o0.store(out + i0 * VEC);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[lr * cw * VEC + lc * VEC + e] = acc[e];
    __syncthreads();
    if ((int)threadIdx.x < cw * VEC) {
        float a = 0.f;
        for (int q = 0; q < rpb; ++q) a += red[q * cw * VEC + threadIdx.x];
        atomicAdd(dbias + blockIdx.x * cw * VEC + threadIdx.x, a);
    }
}

static inline int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

static int gelu_cap() {                                     // SWIN_GELU_BLOCKS: development sweep (default 2048)
    static const int n = swin_dev_int("SWIN_GELU_BLOCKS", 2048);
    return n < 1 ? 1 : n;
}

// grid with (blocks * 256) % vec_per_row == 0, about gelu_cap() blocks at most
static int bias_gelu_blocks(int64_t nvec, int vpr) {
    int unit = vpr / gcd_i(256, vpr);
    int64_t want = (nvec + 255) / 256;
    if (want > gelu_cap()) want = gelu_cap();
    int64_t b = (want + unit - 1) / unit * unit;
    return (int)(b > 0 ? b : unit);
}

template <bool BWD>
static int bias_gelu_launch(const void* x, const float* bias, const void* dy, void* out, float* dbias, int64_t rows, int C,
                            int dtype, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0 || (BWD && !dy)) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    size_t shm = (BWD && dbias) ? (size_t)C * sizeof(float) : 0;
    if (shm > 60000) return SWIN_ERR_UNSUPPORTED;
    if (BWD && dbias) {
        const int vec = dtype == SWIN_BF16 ? 8 : 4;
        if (dtype != SWIN_BF16 && dtype != SWIN_F32) return SWIN_ERR_UNSUPPORTED;
        if (C % vec) return SWIN_ERR_UNSUPPORTED;
        const int vpr = C / vec, cw = gcd_i(vpr, 64), rpb = 256 / cw, nchunk = vpr / cw;
        int64_t ny = (rows + rpb - 1) / rpb;
        int64_t cap = (gelu_cap() + nchunk - 1) / nchunk;      // ~2048 blocks: 8 per CU
        if (ny > cap) ny = cap;
        dim3 grid(nchunk, (unsigned)ny);
        if (dtype == SWIN_BF16)
            bias_gelu_bwd_cols_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)x, bias, (const bf16*)dy, (bf16*)out, dbias, rows, vpr, cw);
        else
            bias_gelu_bwd_cols_kernel<float><<<grid, 256, 0, s>>>((const float*)x, bias, (const float*)dy, (float*)out, dbias, rows, vpr, cw);
        return swin_launch_status();
    }
    if (dtype == SWIN_BF16) {
        if (C % 8) return SWIN_ERR_UNSUPPORTED;
        int64_t nvec = rows * (C / 8);
        int blocks = bias_gelu_blocks(nvec, C / 8);
        bias_gelu_kernel<bf16, BWD><<<blocks, 256, shm, s>>>((const bf16*)x, bias, (const bf16*)dy, (bf16*)out, dbias, nvec,
                                                            C / 8);
    } else if (dtype == SWIN_F32) {
        if (C % 4) return SWIN_ERR_UNSUPPORTED;
        int64_t nvec = rows * (C / 4);
        int blocks = bias_gelu_blocks(nvec, C / 4);
        bias_gelu_kernel<float, BWD><<<blocks, 256, shm, s>>>((const float*)x, bias, (const float*)dy, (float*)out, dbias,
                                                              nvec, C / 4);
    } else return SWIN_ERR_UNSUPPORTED;
    return swin_launch_status();
}

extern "C" int swin_bias_gelu_fwd(const void* x, const float* bias, void* y, int64_t rows, int C, int dtype, void* stream) {
    return bias_gelu_launch<false>(x, bias, nullptr, y, nullptr, rows, C, dtype, stream);
}
extern "C" int swin_bias_gelu_bwd(const void* dy, const void* x, const float* bias, void* dx, float* dbias, int64_t rows,
                                  int C, int dtype, void* stream) {
    return bias_gelu_launch<true>(x, bias, dy, dx, dbias, rows, C, dtype, stream);
}

// ---------------------------------------------------------------- PatchEmbed im2row (k=4, s=4, 3 channels)
// one thread per (row, c, ky): 4 consecutive pixels -> 4 consecutive row elements
template <typename T>
__global__ __launch_bounds__(256) void patch_im2row_kernel(const float* __restrict__ img, T* __restrict__ rows, int B, int Hi,
                                                           int Wi, int Ho, int Wo) {
    int64_t n = (int64_t)B * Ho * Wo * 12;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int sub = (int)(i % 12);
        int64_t r = i / 12;
        int c = sub >> 2, ky = sub & 3;
        int xo = (int)(r % Wo);
        int64_t t = r / Wo;
        int yo = (int)(t % Ho);
        int b = (int)(t / Ho);
        int y = yo * 4 + ky;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (y < Hi) {
            const float* p = img + (((int64_t)b * 3 + c) * Hi + y) * Wi + xo * 4;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
                if (xo * 4 + kx < Wi) v[kx] = p[kx];
        }
        T* o = rows + r * 48 + c * 16 + ky * 4;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) Elt<T>::st(o + kx, v[kx]);
    }
}

extern "C" int swin_patch_im2row(const float* img, void* rows, int B, int Hi, int Wi, int dtype, void* stream) {
    if (!img || !rows || B <= 0 || Hi <= 0 || Wi <= 0) return SWIN_ERR_BAD_ARG;
    int Ho = (Hi + 3) / 4, Wo = (Wi + 3) / 4;
    int64_t n = (int64_t)B * Ho * Wo * 12;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) patch_im2row_kernel<bf16><<<ew_blocks(n), 256, 0, s>>>(img, (bf16*)rows, B, Hi, Wi, Ho, Wo);
    else if (dtype == SWIN_F32) patch_im2row_kernel<float><<<ew_blocks(n), 256, 0, s>>>(img, (float*)rows, B, Hi, Wi, Ho, Wo);
    else return SWIN_ERR_UNSUPPORTED;
    return swin_launch_status();
}

// ---------------------------------------------------------------- FPN top-down: fine += nearest_up(coarse)
// torch 'nearest' with an explicit size: src = min(floor(dst * (float)in/out), in-1)   (fpn.py:188-191)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// channels-last memory (N,H,W,C): one thread per 16-byte channel vector
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_nhwc_kernel(const T* fine, T* out, const T* __restrict__ coarse, int N, int C,
                                                                int Hf, int Wf, int Hc, int Wc, float sh, float sw) {
    constexpr int VEC = Vec16<T>::N;
    const int vpc = C / VEC;
    int64_t n = (int64_t)N * Hf * Wf * vpc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int cv = (int)(i % vpc);
        int64_t t = i / vpc;
        int x = (int)(t % Wf); t /= Wf;
        int y = (int)(t % Hf);
        int b = (int)(t / Hf);
        int ys = nearest_src(y, sh, Hc), xs = nearest_src(x, sw, Wc);
        Vec16<T> f, c;
        f.load(fine + i * VEC);
        c.load(coarse + ((((int64_t)b * Hc + ys) * Wc + xs) * vpc + cv) * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) f.set(e, f.get(e) + c.get(e));
        f.store(out + i * VEC);
    }
}

// contiguous NCHW memory: one thread per element
template <typename T>
__global__ __launch_bounds__(256) void upsample_add_nchw_kernel(const T* fine, T* out, const T* __restrict__ coarse, int64_t NC,
                                                                int Hf, int Wf, int Hc, int Wc, float sh, float sw) {
    int64_t n = NC * Hf * Wf;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int x = (int)(i % Wf);
        int64_t t = i / Wf;
        int y = (int)(t % Hf);
        int64_t nc = t / Hf;
        int ys = nearest_src(y, sh, Hc), xs = nearest_src(x, sw, Wc);
        Elt<T>::st(out + i, Elt<T>::ld(fine + i) + Elt<T>::ld(coarse + (nc * Hc + ys) * Wc + xs));
    }
}

// backward: dcoarse[yc,xc] += sum of dfine over the fine cells that read (yc,xc)
__device__ __forceinline__ void footprint(int c, float scale, int fine_size, int coarse_size, int& lo, int& hi) {
    // candidates around c/scale; exact membership is re-tested with nearest_src by the caller
    lo = (int)floorf((float)c / scale) - 1; if (lo < 0) lo = 0;
    hi = (int)ceilf((float)(c + 1) / scale) + 1; if (hi > fine_size) hi = fine_size;
    (void)coarse_size;
}

template <typename T, bool NHWC>
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const T* __restrict__ dfine, T* __restrict__ dcoarse, int N, int C,
                                                               int Hf, int Wf, int Hc, int Wc, float sh, float sw, int fresh) {
    constexpr int VEC = NHWC ? Vec16<T>::N : 1;
    const int vpc = NHWC ? C / VEC : 1;
    int64_t n = NHWC ? (int64_t)N * Hc * Wc * vpc : (int64_t)N * C * Hc * Wc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int cv = 0, xc, yc;
        int64_t outer;
        int64_t t = i;
        if (NHWC) { cv = (int)(t % vpc); t /= vpc; }
        xc = (int)(t % Wc); t /= Wc;
        yc = (int)(t % Hc); outer = t / Hc;   // NHWC: batch index; NCHW: n*C + c
        int y0, y1, x0, x1;
        footprint(yc, sh, Hf, Hc, y0, y1);
        footprint(xc, sw, Wf, Wc, x0, x1);
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        if (!fresh) {                      // fresh: dcoarse is an uninitialised output, nothing to add to
            if (NHWC) {
                Vec16<T> cur; cur.load(dcoarse + i * Vec16<T>::N);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = cur.get(e);
            } else {
                acc[0] = Elt<T>::ld(dcoarse + i);
            }
        }
        for (int y = y0; y < y1; ++y) {
            if (nearest_src(y, sh, Hc) != yc) continue;
            for (int x = x0; x < x1; ++x) {
                if (nearest_src(x, sw, Wc) != xc) continue;
                if (NHWC) {
                    Vec16<T> d; d.load(dfine + (((outer * Hf + y) * Wf + x) * vpc + cv) * Vec16<T>::N);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) acc[e] += d.get(e);
                } else {
                    acc[0] += Elt<T>::ld(dfine + (outer * Hf + y) * Wf + x);
                }
            }
        }
        if (NHWC) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
            o.store(dcoarse + i * Vec16<T>::N);
        } else {
            Elt<T>::st(dcoarse + i, acc[0]);
        }
    }
}

template <typename T>
static int upsample_launch(void* fine, const void* coarse, int N, int C, int Hf, int Wf, int Hc, int Wc,
                           int channels_last, bool bwd, hipStream_t s, void* out = nullptr, bool fresh = false) {
    if (!out) out = fine;
    float sh = (float)Hc / (float)Hf, sw = (float)Wc / (float)Wf;
    if (channels_last && C % Vec16<T>::N) return SWIN_ERR_UNSUPPORTED;
    if (!bwd) {
        if (channels_last) {
            int64_t n = (int64_t)N * Hf * Wf * (C / Vec16<T>::N);
            upsample_add_nhwc_kernel<T><<<ew_blocks(n), 256, 0, s>>>((const T*)fine, (T*)out, (const T*)coarse, N, C, Hf, Wf, Hc, Wc, sh, sw);
        } else {
            int64_t n = (int64_t)N * C * Hf * Wf;
            upsample_add_nchw_kernel<T><<<ew_blocks(n), 256, 0, s>>>((const T*)fine, (T*)out, (const T*)coarse, (int64_t)N * C, Hf, Wf, Hc,
                                                                    Wc, sh, sw);
        }
    } else {
        // here `fine` is dfine (read) and `coarse` is dcoarse (read-modify-write)
        if (channels_last) {
            int64_t n = (int64_t)N * Hc * Wc * (C / Vec16<T>::N);
            upsample_add_bwd_kernel<T, true><<<ew_blocks(n), 256, 0, s>>>((const T*)fine, (T*)coarse, N, C, Hf, Wf, Hc, Wc, sh, sw, fresh ? 1 : 0);
        } else {
            int64_t n = (int64_t)N * C * Hc * Wc;
            upsample_add_bwd_kernel<T, false><<<ew_blocks(n), 256, 0, s>>>((const T*)fine, (T*)coarse, N, C, Hf, Wf, Hc, Wc, sh,
                                                                          sw, fresh ? 1 : 0);
        }
    }
    return swin_launch_status();
}

extern "C" int fpn_upsample_add_fwd(void* fine, const void* coarse, int N, int C, int Hf, int Wf, int Hc, int Wc,
                                    int channels_last, int dtype, void* stream) {
    if (!fine || !coarse || N <= 0 || C <= 0 || Hf <= 0 || Wf <= 0 || Hc <= 0 || Wc <= 0) return SWIN_ERR_BAD_ARG;
    if (dtype == SWIN_BF16) return upsample_launch<bf16>(fine, coarse, N, C, Hf, Wf, Hc, Wc, channels_last, false, (hipStream_t)stream);
    if (dtype == SWIN_F32) return upsample_launch<float>(fine, coarse, N, C, Hf, Wf, Hc, Wc, channels_last, false, (hipStream_t)stream);
    return SWIN_ERR_UNSUPPORTED;
}

// out = fine + nearest_upsample(coarse) without touching `fine` (no clone in the caller); dcoarse = footprint sums of dfine
// written from scratch (no memset in the caller).
extern "C" int fpn_upsample_add_out_fwd(const void* fine, const void* coarse, void* out, int N, int C, int Hf, int Wf, int Hc, int Wc,
                                        int channels_last, int dtype, void* stream) {
    if (!fine || !coarse || !out || N <= 0 || C <= 0 || Hf <= 0 || Wf <= 0 || Hc <= 0 || Wc <= 0) return SWIN_ERR_BAD_ARG;
    if (dtype == SWIN_BF16) return upsample_launch<bf16>((void*)fine, coarse, N, C, Hf, Wf, Hc, Wc, channels_last, false, (hipStream_t)stream, out);
    if (dtype == SWIN_F32) return upsample_launch<float>((void*)fine, coarse, N, C, Hf, Wf, Hc, Wc, channels_last, false, (hipStream_t)stream, out);
    return SWIN_ERR_UNSUPPORTED;
}
extern "C" int fpn_upsample_add_out_bwd(const void* dfine, void* dcoarse, int N, int C, int Hf, int Wf, int Hc, int Wc,
                                        int channels_last, int dtype, void* stream) {
    if (!dfine || !dcoarse || N <= 0 || C <= 0 || Hf <= 0 || Wf <= 0 || Hc <= 0 || Wc <= 0) return SWIN_ERR_BAD_ARG;
    if (dtype == SWIN_BF16) return upsample_launch<bf16>((void*)dfine, dcoarse, N, C, Hf, Wf, Hc, Wc, channels_last, true, (hipStream_t)stream, nullptr, true);
    if (dtype == SWIN_F32) return upsample_launch<float>((void*)dfine, dcoarse, N, C, Hf, Wf, Hc, Wc, channels_last, true, (hipStream_t)stream, nullptr, true);
    return SWIN_ERR_UNSUPPORTED;
}

extern "C" int fpn_upsample_add_bwd(const void* dfine, void* dcoarse, int N, int C, int Hf, int Wf, int Hc, int Wc,
                                    int channels_last, int dtype, void* stream) {
    if (!dfine || !dcoarse || N <= 0 || C <= 0 || Hf <= 0 || Wf <= 0 || Hc <= 0 || Wc <= 0) return SWIN_ERR_BAD_ARG;
    if (dtype == SWIN_BF16) return upsample_launch<bf16>((void*)dfine, dcoarse, N, C, Hf, Wf, Hc, Wc, channels_last, true, (hipStream_t)stream);
    if (dtype == SWIN_F32) return upsample_launch<float>((void*)dfine, dcoarse, N, C, Hf, Wf, Hc, Wc, channels_last, true, (hipStream_t)stream);
    return SWIN_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------
// Weights of the data-gradient convolution for a batch of 3x3 conv weights, one launch: dst[ci][2-ky][2-kx][co] =
// src[co][ky][kx][ci] (bf16; src = the resident (Cout,3,3,Cin) layout).  The data gradient of y = conv(x, w) is
// conv(dy, rot180(w) with in/out swapped) (the same implicit-GEMM kernel); with torch this re-layout cost a flip and a
// permuted copy per layer and step.  One 32x32 (co, ci) tile per block through LDS: both sides move whole 64-byte rows.
// ------------------------------------------------------------------------------------
#define DGL_MAX 16
struct DgradLayoutTable {
    const bf16* src[DGL_MAX];
    bf16* dst[DGL_MAX];
    int co[DGL_MAX], ci[DGL_MAX], taps[DGL_MAX], tile0[DGL_MAX + 1];       // taps: 9 (3x3 conv, rotated) or 1 (a plain transpose)
    int n;
};

__global__ __launch_bounds__(256) void conv_dgrad_layout_kernel(DgradLayoutTable t) {
    __shared__ bf16 tile[32][34];
    int k = 0;
    while (k + 1 < t.n && (int)blockIdx.x >= t.tile0[k + 1]) ++k;
    const int co = t.co[k], ci = t.ci[k], taps = t.taps[k];
    const int tci = (ci + 31) / 32, tco = (co + 31) / 32;
    int id = blockIdx.x - t.tile0[k];
    const int tap = id / (tci * tco); id -= tap * (tci * tco);
    const int co0 = (id / tci) * 32, ci0 = (id % tci) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const bf16* src = t.src[k];
    bf16* dst = t.dst[k];
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int o = co0 + r, i = ci0 + tx;
        if (o < co && i < ci) tile[r][tx] = src[((size_t)o * taps + tap) * ci + i];
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int i = ci0 + r, o = co0 + tx;
        if (o < co && i < ci) dst[((size_t)i * taps + (taps - 1 - tap)) * co + o] = tile[tx][r];
    }
}

// srcs / dsts: HOST arrays of n device pointers; couts / cins: HOST arrays of n ints.
extern "C" int conv_dgrad_layout_multi(const void* const* srcs, void* const* dsts, const int* couts, const int* cins, int n,
                                       void* stream) {
    if (n == 0) return SWIN_OK;
    if (!srcs || !dsts || !couts || !cins || n < 0) return SWIN_ERR_BAD_ARG;
    for (int base = 0; base < n; base += DGL_MAX) {
        DgradLayoutTable t;
        t.n = n - base < DGL_MAX ? n - base : DGL_MAX;
        int tiles = 0;
        for (int k = 0; k < t.n; ++k) {
            if (!srcs[base + k] || !dsts[base + k] || couts[base + k] <= 0 || cins[base + k] <= 0) return SWIN_ERR_BAD_ARG;
            t.src[k] = (const bf16*)srcs[base + k]; t.dst[k] = (bf16*)dsts[base + k];
            t.co[k] = couts[base + k]; t.ci[k] = cins[base + k]; t.taps[k] = 9;
            t.tile0[k] = tiles;
            tiles += 9 * ((t.co[k] + 31) / 32) * ((t.ci[k] + 31) / 32);
        }
        t.tile0[t.n] = tiles;
        conv_dgrad_layout_kernel<<<tiles, 256, 0, (hipStream_t)stream>>>(t);
        int st = swin_launch_status();
        if (st != SWIN_OK) return st;
    }
    return SWIN_OK;
}

// dsts[k] (cols, rows) = srcs[k] (rows, cols)^T for n bf16 matrices in as few launches as the table allows (16 per launch): the
// K-contiguous weights of the Linear layers' data-gradient GEMMs on the hand-written kernel (dx = dy W needs W^T as its
// (N, K) operand), rebuilt once per optimizer step.
extern "C" int linear_t_layout_multi(const void* const* srcs, void* const* dsts, const int* rows, const int* cols, int n, void* stream) {
    if (n == 0) return SWIN_OK;
    if (!srcs || !dsts || !rows || !cols || n < 0) return SWIN_ERR_BAD_ARG;
    for (int base = 0; base < n; base += DGL_MAX) {
        DgradLayoutTable t;
        t.n = n - base < DGL_MAX ? n - base : DGL_MAX;
        int tiles = 0;
        for (int k = 0; k < t.n; ++k) {
            if (!srcs[base + k] || !dsts[base + k] || rows[base + k] <= 0 || cols[base + k] <= 0) return SWIN_ERR_BAD_ARG;
            t.src[k] = (const bf16*)srcs[base + k]; t.dst[k] = (bf16*)dsts[base + k];
            t.co[k] = rows[base + k]; t.ci[k] = cols[base + k]; t.taps[k] = 1;
            t.tile0[k] = tiles;
            tiles += ((t.co[k] + 31) / 32) * ((t.ci[k] + 31) / 32);
        }
        t.tile0[t.n] = tiles;
        conv_dgrad_layout_kernel<<<tiles, 256, 0, (hipStream_t)stream>>>(t);
        int st = swin_launch_status();
        if (st != SWIN_OK) return st;
    }
    return SWIN_OK;
}

// ------------------------------------------------------------------------------------
// Data gradient of a narrow head behind a ReLU: dx[t][c] = gate[t][c] > 0 ? sum_k dy[t][k] w[k][c] : 0 with K <= 64
// (the RPN's fused cls | reg 1x1 heads, rpn_head.py:41-47: K = 5A padded to 16, C = 256; gate = the ReLU output of
// rpn_conv).  As a GEMM this is 16 deep -- a bandwidth problem, not a matrix-core one: the weights sit in LDS as fp32, a
// thread owns one 16-byte piece of four consecutive tokens, and the ReLU backward of the layer below (torch:
// threshold_backward, another pass over the (T, C) map) is the store predicate.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void narrow_dgrad_gated_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ w,
                                                                 const bf16* __restrict__ gate, bf16* __restrict__ dx, int64_t T,
                                                                 int K, int C) {
    extern __shared__ float wl[];                 // [K][C]
    for (int i = threadIdx.x; i < K * C; i += 256) wl[i] = (float)w[i];
    __syncthreads();
    const int cpt = C / 8;
    const int64_t groups = (T + 3) / 4, total = groups * cpt;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t t0 = (idx / cpt) * 4;
        const int c0 = (int)(idx % cpt) * 8;
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 8) {
            bf16x8 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t t = t0 + q < T ? t0 + q : T - 1;
                d[q] = *(const bf16x8*)(dy + t * K + k0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 w0 = *(const float4*)(wl + (k0 + j) * C + c0), w1 = *(const float4*)(wl + (k0 + j) * C + c0 + 4);
                const float wr[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dv = (float)d[q][j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[q][e] = fmaf(dv, wr[e], acc[q][e]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (t0 + q >= T) break;
            const bf16x8 g = *(const bf16x8*)(gate + (t0 + q) * C + c0);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (float)g[e] > 0.f ? (bf16)acc[q][e] : (bf16)0.f;
            *(bf16x8*)(dx + (t0 + q) * C + c0) = o;
        }
    }
}

// The same on the matrix cores (round 3), K in {16, 32}, C % 32 == 0: in the step the fp32-FMA form above ran 72 us on P2's 128 000
// tokens -- 524 M multiply-adds plus 32 LDS reads per thread, VALU-bound -- for 134 MB of traffic (22 us at HBM speed).  A wave owns
// 32 tokens: their dy rows are the B operand (one 16-byte load per lane and k-step), the weight columns the A operand (gathered
// once per wave: 8 scalar loads per tile and k-step, L2-resident), the accumulator tile has the token on the lane and 16 channels in
// the registers (csrc/ts_linear.hip), so the gate test and the store are two 16-byte pieces per tile.  fp32 accumulation in the MFMA's
// order instead of k = 0 .. K-1: an output may differ by an ulp of the 16-bit type.
template <int KS>
__global__ __launch_bounds__(256) void narrow_dgrad_gated_mfma_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ w,
                                                                      const bf16* __restrict__ gate, bf16* __restrict__ dx, int64_t T,
                                                                      int C) {
    constexpr int K = 16 * KS;
    extern __shared__ __attribute__((aligned(16))) bf16 wt[];                  // w transposed: [C][K + 8] (row stride of an odd number of 16-byte slots)
    constexpr int RS = K + 8;
    for (int i = threadIdx.x; i < K * C; i += 256) { const int k = i / C, c = i - k * C; wt[c * RS + k] = w[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int pr = ((r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1));            // rows of the A operand in the order that puts channels
    const int ct_n = C / 32;                                                   // 8 h + q / 16 + 8 h + (q - 8) into the registers
    const int64_t groups = (T + 31) / 32;
    for (int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); grp < groups; grp += (int64_t)gridDim.x * 4) {
        const int64_t tok = grp * 32 + r, tokc = tok < T ? tok : T - 1;
        bf16x8 xf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = *(const bf16x8*)(dy + tokc * K + 16 * s + 8 * h);
        for (int ct = 0; ct < ct_n; ++ct) {
            const bf16* wrow = wt + (32 * ct + pr) * RS + 8 * h;               // this lane's weight column
            f32x16 a = f32x16{0};
#pragma unroll
            for (int s = 0; s < KS; ++s) a = SWIN_MFMA_32x32x16(*(const bf16x8*)(wrow + 16 * s), xf[s], a, 0, 0, 0);
            if (tok < T) {
                const int c0 = 32 * ct + 8 * h;
                const bf16x8 g0 = *(const bf16x8*)(gate + tok * C + c0), g1 = *(const bf16x8*)(gate + tok * C + c0 + 16);
                bf16x8 o0, o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o0[e] = (float)g0[e] > 0.f ? (bf16)a[e] : (bf16)0.f;
                    o1[e] = (float)g1[e] > 0.f ? (bf16)a[8 + e] : (bf16)0.f;
                }
                *(bf16x8*)(dx + tok * C + c0) = o0;
                *(bf16x8*)(dx + tok * C + c0 + 16) = o1;
            }
        }
    }
}

// dy (T,K) bf16, w (K,C) bf16, gate (T,C) bf16 -> dx (T,C) bf16.  K % 8 == 0, K <= 64, C % 8 == 0, K*C*4 <= 64 KB.
extern "C" int narrow_dgrad_gated_bf16(const void* dy, const void* w, const void* gate, void* dx, int64_t T, int K, int C,
                                       void* stream) {
    if (T == 0) return SWIN_OK;
    if (!dy || !w || !gate || !dx || T < 0 || K <= 0 || C <= 0) return SWIN_ERR_BAD_ARG;
    if (K % 8 != 0 || K > 64 || C % 8 != 0 || (size_t)K * C * sizeof(float) > 65536) return SWIN_ERR_UNSUPPORTED;
    if ((K == 16 || K == 32) && C % 32 == 0) {
        const int64_t groups = (T + 31) / 32;
        const unsigned nb = (unsigned)std::min<int64_t>((groups + 3) / 4, 256 * 4);       // a block transposes the weights into LDS once
        const size_t lds = (size_t)C * (K + 8) * 2;
        if (lds > 65536) return SWIN_ERR_UNSUPPORTED;
        if (K == 16) narrow_dgrad_gated_mfma_kernel<1><<<nb, 256, lds, (hipStream_t)stream>>>((const bf16*)dy, (const bf16*)w, (const bf16*)gate, (bf16*)dx, T, C);
        else narrow_dgrad_gated_mfma_kernel<2><<<nb, 256, lds, (hipStream_t)stream>>>((const bf16*)dy, (const bf16*)w, (const bf16*)gate, (bf16*)dx, T, C);
        return swin_launch_status();
    }
    const int64_t total = ((T + 3) / 4) * (C / 8);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    narrow_dgrad_gated_kernel<<<(unsigned)blocks, 256, (size_t)K * C * sizeof(float), (hipStream_t)stream>>>(
        (const bf16*)dy, (const bf16*)w, (const bf16*)gate, (bf16*)dx, T, K, C);
    return swin_launch_status();
}
