// LayerNorm family for gfx950 (HBM-bound; one pass over the row held in registers).
//
// A row of C channels is owned by a group of G lanes (G = power of two <= 64); each lane keeps
// NV 16-byte vectors of the row in registers, so x is read once and y written once.
// Variants selected by the `Src` policy:
//   PlainSrc      : row r is contiguous at x + r*C              (nn.LayerNorm on (rows, C))
//   AddSrc        : row = x + scale[b]*y, written back as xo    (residual + DropPath + LN fused,
//                                                                swin_transformer.py:252-253 / :211)
//   MergeSrc      : row = concat of the 2x2 neighbourhood        (PatchMerging, :284-295)
// Statistics in fp32: mean first, then variance around the mean (two reductions over registers).
#include <cstdlib>

#include "common.h"
#include "aux_defer.h"

extern "C" int swin_fork_stream(void* main, void* side);

template <typename T> struct PlainSrc {
    const T* x; int C;
    __device__ __forceinline__ bool load(int64_t row, int vidx, Vec16<T>& v) const {
        v.load(x + row * C + vidx * Vec16<T>::N);
        return true;
    }
};

// PatchMerging: output row (b,i,j) gathers x0=(2i,2j) x1=(2i+1,2j) x2=(2i,2j+1) x3=(2i+1,2j+1)
template <typename T> struct MergeSrc {
    const T* x; int C, H, W, Ho, Wo;   // C = input channels; row length is 4C
    __device__ __forceinline__ int64_t src_token(int64_t row, int quad) const {
        int j = (int)(row % Wo); int64_t t = row / Wo; int i = (int)(t % Ho); int64_t b = t / Ho;
        int r = 2 * i + (quad & 1), c = 2 * j + (quad >> 1);
        if (r >= H || c >= W) return -1;
        return (b * H + r) * W + c;
    }
    __device__ __forceinline__ bool load(int64_t row, int vidx, Vec16<T>& v) const {
        const int vpc = C / Vec16<T>::N;           // vectors per source token
        int quad = vidx / vpc, within = vidx - quad * vpc;
        int64_t tok = src_token(row, quad);
        if (tok < 0) {
#pragma unroll
            for (int e = 0; e < Vec16<T>::N; ++e) v.set(e, 0.f);
            return false;
        }
        v.load(x + tok * C + within * Vec16<T>::N);
        return true;
    }
};

struct LnLaunch { int G, NV, rows_per_block, blocks; };

template <typename T> static LnLaunch ln_plan(int64_t rows, int C) {
    // chunks <= 64: 16 lanes per row, up to 4 vectors per lane; else a full wave per row
    const int VEC = Vec16<T>::N;
    int chunks = C / VEC;
    int G = chunks <= 64 ? 16 : 64;
    int NV = (chunks + G - 1) / G;
    static const int allowed[] = {1, 2, 3, 4, 6, 8, 12, 16};
    int nv2 = -1;
    for (int a : allowed) if (a >= NV) { nv2 = a; break; }
    LnLaunch p; p.G = G; p.NV = nv2; p.rows_per_block = 256 / G;
    int64_t nb = (rows + p.rows_per_block - 1) / p.rows_per_block;
    static const int cap = swin_dev_int("SWIN_LN_BLOCKS", 2048);      // development sweep
    p.blocks = (int)(nb < cap ? nb : cap);
    return p;
}

template <int G> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------- forward
template <typename T, int G, int NV, typename Src, bool ADD>
__global__ __launch_bounds__(256) void ln_fwd_kernel(Src src, const T* __restrict__ addy, const float* __restrict__ scale,
                                                     int64_t rows_per_sample, T* __restrict__ xo,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     T* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int64_t rows, int C, float eps) {
    constexpr int VEC = Vec16<T>::N;
    const int chunks = C / VEC;
    const int lig = threadIdx.x % G, gid = threadIdx.x / G;
    const int rpb = 256 / G;
    for (int64_t row0 = (int64_t)blockIdx.x * rpb; row0 < rows; row0 += (int64_t)gridDim.x * rpb) {
        const int64_t row = row0 + gid;
        const bool rv = row < rows;
        Vec16<T> v[NV];
        float s = 0.f;
        float sc = 1.f;
        if (ADD && rv && scale) sc = scale[row / rows_per_sample];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lig + i * G;
            if (rv && vi < chunks) {
                src.load(row, vi, v[i]);
                if (ADD) {
                    Vec16<T> a; a.load(addy + row * C + vi * VEC);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[i].set(e, v[i].get(e) + sc * a.get(e));
                    v[i].store(xo + row * C + vi * VEC);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) s += v[i].get(e);
            }
        }
        if (y == nullptr) continue;              // residual-only call
        float mean = group_sum<G>(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lig + i * G;
            if (rv && vi < chunks) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { float d = v[i].get(e) - mean; q += d * d; }
            }
        }
        float rstd = rsqrtf(group_sum<G>(q) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lig + i * G;
            if (rv && vi < chunks) {
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    int cidx = vi * VEC + e;
                    o.set(e, (v[i].get(e) - mean) * rstd * gamma[cidx] + beta[cidx]);
                }
                o.store(y + row * C + vi * VEC);
            }
        }
        if (rv && lig == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
    }
}

// ---------------------------------------------------------------------------- backward
// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ; (+ dres) ; optional dy2 = scale*dx
template <typename T, int G, int NV, typename Src, bool MERGE>
__global__ __launch_bounds__(256) void ln_bwd_kernel(Src src, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const T* __restrict__ dres, T* __restrict__ dx,
                                                     T* __restrict__ dy2, const float* __restrict__ scale,
                                                     int64_t rows_per_sample, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int64_t rows, int C, int use_slab,
                                                     float* __restrict__ partials) {
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // NSLAB x (dgamma[C] | dbeta[C]), or one such row
    const int chunks = C / VEC;
    const int lig = threadIdx.x % G, gid = threadIdx.x / G;
    const int rpb = 256 / G;
    if (!use_slab) {
        for (int i = threadIdx.x; i < 2 * C; i += 256) sm[i] = 0.f;
        __syncthreads();
    }
    float ag[NV][VEC], ab[NV][VEC];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < VEC; ++e) { ag[i][e] = 0.f; ab[i][e] = 0.f; }

    for (int64_t row0 = (int64_t)blockIdx.x * rpb; row0 < rows; row0 += (int64_t)gridDim.x * rpb) {
        const int64_t row = row0 + gid;
        const bool rv = row < rows;
        float mean = 0.f, rstd = 0.f;
        if (rv) { mean = mean_in[row]; rstd = rstd_in[row]; }
        Vec16<T> xv[NV], gv[NV], rsv[NV];
        bool real[NV];
        float s1 = 0.f, s2 = 0.f;
        const bool has_res = !MERGE && dres != nullptr;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lig + i * G;
            real[i] = false;
            if (rv && vi < chunks) {
                real[i] = src.load(row, vi, xv[i]);
                gv[i].load(dy + row * C + vi * VEC);
                if (has_res) rsv[i].load(dres + row * C + vi * VEC);      // issued with the other loads: one latency
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float xh = (xv[i].get(e) - mean) * rstd;
                    float d = gv[i].get(e);
                    ag[i][e] += d * xh;
                    ab[i][e] += d;
                    float gg = d * gamma[vi * VEC + e];
                    s1 += gg; s2 += gg * xh;
                }
            }
        }
        s1 = group_sum<G>(s1) / (float)C;
        s2 = group_sum<G>(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lig + i * G;
            if (rv && vi < chunks) {
                Vec16<T> o, o2;
                float sc = 1.f;
                if (!MERGE && dy2 && scale) sc = scale[row / rows_per_sample];
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    float xh = (xv[i].get(e) - mean) * rstd;
                    float gg = gv[i].get(e) * gamma[vi * VEC + e];
                    float d = rstd * (gg - s1 - xh * s2);
                    if (has_res) d += rsv[i].get(e);
                    o.set(e, d);
                    if (!MERGE && dy2) o2.set(e, d * sc);
                }
                if constexpr (MERGE) {
                    if (real[i]) {
                        const MergeSrc<T>& ms = *reinterpret_cast<const MergeSrc<T>*>(&src);
                        const int vpc = ms.C / VEC;
                        int quad = vi / vpc, within = vi - quad * vpc;
                        int64_t tok = ms.src_token(row, quad);
                        o.store(dx + tok * ms.C + within * VEC);
                    }
                } else {
                    o.store(dx + row * C + vi * VEC);
                    if (dy2) o2.store(dy2 + row * C + vi * VEC);
                }
            }
        }
    }
    // block reduction of the parameter gradients.  slab form (use_slab): row groups that share a wave are folded with one
    // shuffle step, then every remaining group stores its partial row [2C] with plain LDS writes and the block sums the
    // NSLAB rows -- LDS float atomics here are 16-way same-address conflicts and cost more than the streaming loop.
    constexpr int NSLAB = (G == 16) ? 8 : 256 / G;
    if (use_slab) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            int vi = lig + i * G;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float a = ag[i][e], b = ab[i][e];
                if (G == 16) { a += __shfl_xor(a, 16); b += __shfl_xor(b, 16); }
                if (vi < chunks && (G != 16 || (gid & 1) == 0)) {
                    const int slab = (G == 16) ? (gid >> 1) : gid;
                    sm[slab * 2 * C + vi * VEC + e] = a;
                    sm[slab * 2 * C + C + vi * VEC + e] = b;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < NSLAB; ++q) a += sm[q * 2 * C + i];
            if (partials) partials[(int64_t)blockIdx.x * 2 * C + i] = a;      // plain store; ln_param_reduce_kernel sums
            else atomicAdd(i < C ? &dgamma[i] : &dbeta[i - C], a);
        }
        return;
    }
    // wide rows whose slabs would not fit the LDS: LDS atomics (4-way conflicts at G = 64), one global atomic per channel
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int vi = lig + i * G;
        if (vi < chunks) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                atomicAdd(&sm[vi * VEC + e], ag[i][e]);
                atomicAdd(&sm[C + vi * VEC + e], ab[i][e]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
        atomicAdd(&dgamma[i], sm[i]);
        atomicAdd(&dbeta[i], sm[C + i]);
    }
}

// ---------------------------------------------------------------------------- dispatch
#define LN_CASE(Gv, NVv, ...) \
    if (p.G == Gv && p.NV == NVv) { constexpr int G = Gv; constexpr int NV = NVv; __VA_ARGS__; } else
#define LN_DISPATCH(...)                                                                         \
    LN_CASE(16, 1, __VA_ARGS__) LN_CASE(16, 2, __VA_ARGS__) LN_CASE(16, 3, __VA_ARGS__)          \
    LN_CASE(16, 4, __VA_ARGS__) LN_CASE(64, 2, __VA_ARGS__) LN_CASE(64, 3, __VA_ARGS__)          \
    LN_CASE(64, 4, __VA_ARGS__) LN_CASE(64, 6, __VA_ARGS__) LN_CASE(64, 8, __VA_ARGS__)          \
    LN_CASE(64, 12, __VA_ARGS__) LN_CASE(64, 16, __VA_ARGS__) { return SWIN_ERR_UNSUPPORTED; }

template <typename T, typename Src, bool ADD>
static int ln_fwd_launch(Src src, const T* addy, const float* scale, int64_t rps, T* xo, const float* gamma,
                         const float* beta, T* y, float* mean, float* rstd, int64_t rows, int C, float eps,
                         hipStream_t s) {
    if (C % Vec16<T>::N != 0) return SWIN_ERR_UNSUPPORTED;
    LnLaunch p = ln_plan<T>(rows, C);
    LN_DISPATCH((ln_fwd_kernel<T, G, NV, Src, ADD><<<p.blocks, 256, 0, s>>>(src, addy, scale, rps, xo, gamma, beta, y,
                                                                            mean, rstd, rows, C, eps)))
    return swin_launch_status();
}

// dgamma / dbeta += sum over the blocks' partial rows [nblk][2C] (written with plain stores: hundreds of blocks adding
// to the same 2C addresses with float atomics run an order of magnitude below the atomic rate, guide G12)
__global__ __launch_bounds__(1024) void ln_param_reduce_kernel(const float* __restrict__ partials, int nblk, int C,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                      // column of [dgamma | dbeta]; the 16 waves split the rows
    float a = 0.f;
    if (i < 2 * C) {
        const float* p = partials + i;
        int b = w;
        for (; b + 48 < nblk; b += 64) {                       // 4 independent loads in flight per lane
            const float v0 = p[(int64_t)b * 2 * C], v1 = p[(int64_t)(b + 16) * 2 * C];
            const float v2 = p[(int64_t)(b + 32) * 2 * C], v3 = p[(int64_t)(b + 48) * 2 * C];
            a += (v0 + v1) + (v2 + v3);
        }
        for (; b < nblk; b += 16) a += p[(int64_t)b * 2 * C];
    }
    red[w][lane] = a;
    __syncthreads();
    if (w == 0 && i < 2 * C) {
        a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += red[q][lane];
        float* dst = i < C ? dgamma + i : dbeta + (i - C);
        *dst += a;
    }
}

static int ln_bwd_cap() {                                   // SWIN_LN_BWD_BLOCKS: development sweep; 768: 33.3 -> 29.9 us at stage 1 vs 512
    static const int n = swin_dev_int("SWIN_LN_BWD_BLOCKS", 768);
    return n < 1 ? 1 : n;
}
template <typename T> static int ln_bwd_blocks(int64_t rows, int C) {
    LnLaunch p = ln_plan<T>(rows, C);
    return p.blocks < ln_bwd_cap() ? p.blocks : ln_bwd_cap();
}

template <typename T, typename Src, bool MERGE>
static int ln_bwd_launch(Src src, const T* dy, const float* gamma, const float* mean, const float* rstd,
                         const T* dres, T* dx, T* dy2, const float* scale, int64_t rps, float* dgamma, float* dbeta,
                         int64_t rows, int C, hipStream_t s, float* workspace = nullptr) {
    if (C % Vec16<T>::N != 0) return SWIN_ERR_UNSUPPORTED;
    LnLaunch p = ln_plan<T>(rows, C);
    int blocks = p.blocks < ln_bwd_cap() ? p.blocks : ln_bwd_cap();
    const int nslab = p.G == 16 ? 8 : 256 / p.G;
    size_t shm = (size_t)nslab * 2 * C * sizeof(float);
    int use_slab = 1;
    if (shm > 60000) { use_slab = 0; shm = 2 * (size_t)C * sizeof(float); }
    float* partials = use_slab ? workspace : nullptr;
    LN_DISPATCH((ln_bwd_kernel<T, G, NV, Src, MERGE><<<blocks, 256, shm, s>>>(src, dy, gamma, mean, rstd, dres, dx, dy2,
                                                                              scale, rps, dgamma, dbeta, rows, C, use_slab,
                                                                              partials)))
    if (partials) {
        // a block backward is collecting its small reductions into one launch (csrc/tail_reduce.hip)
        if (swin_tail_push(SwinTailProb{partials, dgamma, dbeta, SWIN_TAIL_COLSUM, blocks, 2 * C, C, 0, 0})) return swin_launch_status();
        // nobody on this stream consumes the parameter gradients: with an auxiliary stream set, reduce them there
        auto launch = [=](void* st) {
            ln_param_reduce_kernel<<<(2 * C + 63) / 64, 1024, 0, (hipStream_t)st>>>(partials, blocks, C, dgamma, dbeta);
            return swin_launch_status();
        };
        if (!swin_aux_push(launch)) {
            void* aux = swin_aux_stream();
            void* rs = (void*)s;
            if (aux && aux != (void*)s) {
                if (swin_fork_stream((void*)s, aux) != SWIN_OK) return SWIN_ERR_LAUNCH;
                rs = aux;
            }
            return launch(rs);
        }
    }
    return swin_launch_status();
}

extern "C" int swin_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                  float* rstd, int64_t rows, int C, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) {
        PlainSrc<bf16> src{(const bf16*)x, C};
        return ln_fwd_launch<bf16, PlainSrc<bf16>, false>(src, nullptr, nullptr, 1, nullptr, gamma, beta, (bf16*)y, mean,
                                                          rstd, rows, C, eps, s);
    } else if (dtype == SWIN_F32) {
        PlainSrc<float> src{(const float*)x, C};
        return ln_fwd_launch<float, PlainSrc<float>, false>(src, nullptr, nullptr, 1, nullptr, gamma, beta, (float*)y,
                                                            mean, rstd, rows, C, eps, s);
    }
    return SWIN_ERR_UNSUPPORTED;
}

extern "C" int swin_add_layernorm_fwd(const void* x, const void* yadd, const float* scale, int64_t rows_per_sample,
                                      const float* gamma, const float* beta, void* xo, void* n, float* mean,
                                      float* rstd, int64_t rows, int C, float eps, int dtype, void* stream) {
    if (!x || !yadd || !xo || rows <= 0 || C <= 0 || rows_per_sample <= 0) return SWIN_ERR_BAD_ARG;
    if (n && (!gamma || !beta)) return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) {
        PlainSrc<bf16> src{(const bf16*)x, C};
        return ln_fwd_launch<bf16, PlainSrc<bf16>, true>(src, (const bf16*)yadd, scale, rows_per_sample, (bf16*)xo, gamma,
                                                         beta, (bf16*)n, mean, rstd, rows, C, eps, s);
    } else if (dtype == SWIN_F32) {
        PlainSrc<float> src{(const float*)x, C};
        return ln_fwd_launch<float, PlainSrc<float>, true>(src, (const float*)yadd, scale, rows_per_sample, (float*)xo,
                                                           gamma, beta, (float*)n, mean, rstd, rows, C, eps, s);
    }
    return SWIN_ERR_UNSUPPORTED;
}

// bytes of the optional parameter-gradient workspace of swin_layernorm_bwd (C = row width) and
// swin_patch_merge_ln_bwd (pass 4 * C): one partial [dgamma | dbeta] row per thread block
extern "C" int64_t swin_layernorm_bwd_workspace_bytes(int64_t rows, int C, int dtype) {
    if (rows <= 0 || C <= 0) return 16;
    int blocks = dtype == SWIN_BF16 ? ln_bwd_blocks<bf16>(rows, C) : ln_bwd_blocks<float>(rows, C);
    return (int64_t)blocks * 2 * C * (int64_t)sizeof(float);
}

extern "C" int swin_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                  const float* rstd, const void* dres, void* dx, void* dx_scaled, const float* scale,
                                  int64_t rows_per_sample, float* dgamma, float* dbeta, int64_t rows, int C,
                                  int dtype, void* workspace, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0)
        return SWIN_ERR_BAD_ARG;
    if (rows_per_sample <= 0) rows_per_sample = 1;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) {
        PlainSrc<bf16> src{(const bf16*)x, C};
        return ln_bwd_launch<bf16, PlainSrc<bf16>, false>(src, (const bf16*)dy, gamma, mean, rstd, (const bf16*)dres,
                                                          (bf16*)dx, (bf16*)dx_scaled, scale, rows_per_sample, dgamma,
                                                          dbeta, rows, C, s, (float*)workspace);
    } else if (dtype == SWIN_F32) {
        PlainSrc<float> src{(const float*)x, C};
        return ln_bwd_launch<float, PlainSrc<float>, false>(src, (const float*)dy, gamma, mean, rstd,
                                                            (const float*)dres, (float*)dx, (float*)dx_scaled, scale,
                                                            rows_per_sample, dgamma, dbeta, rows, C, s, (float*)workspace);
    }
    return SWIN_ERR_UNSUPPORTED;
}

extern "C" int swin_patch_merge_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                       float* rstd, int B, int H, int W, int C, float eps, int dtype, void* stream) {
    if (!x || !gamma || !beta || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0) return SWIN_ERR_BAD_ARG;
    int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    int64_t rows = (int64_t)B * Ho * Wo;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) {
        if (C % 8) return SWIN_ERR_UNSUPPORTED;
        MergeSrc<bf16> src{(const bf16*)x, C, H, W, Ho, Wo};
        return ln_fwd_launch<bf16, MergeSrc<bf16>, false>(src, nullptr, nullptr, 1, nullptr, gamma, beta, (bf16*)y, mean,
                                                          rstd, rows, 4 * C, eps, s);
    } else if (dtype == SWIN_F32) {
        if (C % 4) return SWIN_ERR_UNSUPPORTED;
        MergeSrc<float> src{(const float*)x, C, H, W, Ho, Wo};
        return ln_fwd_launch<float, MergeSrc<float>, false>(src, nullptr, nullptr, 1, nullptr, gamma, beta, (float*)y,
                                                            mean, rstd, rows, 4 * C, eps, s);
    }
    return SWIN_ERR_UNSUPPORTED;
}

extern "C" int swin_patch_merge_ln_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                       const float* rstd, void* dx, float* dgamma, float* dbeta, int B, int H, int W,
                                       int C, int dtype, void* workspace, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || B <= 0 || H <= 0 || W <= 0 || C <= 0)
        return SWIN_ERR_BAD_ARG;
    int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    int64_t rows = (int64_t)B * Ho * Wo;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SWIN_BF16) {
        if (C % 8) return SWIN_ERR_UNSUPPORTED;
        MergeSrc<bf16> src{(const bf16*)x, C, H, W, Ho, Wo};
        return ln_bwd_launch<bf16, MergeSrc<bf16>, true>(src, (const bf16*)dy, gamma, mean, rstd, nullptr, (bf16*)dx,
                                                         nullptr, nullptr, 1, dgamma, dbeta, rows, 4 * C, s, (float*)workspace);
    } else if (dtype == SWIN_F32) {
        if (C % 4) return SWIN_ERR_UNSUPPORTED;
        MergeSrc<float> src{(const float*)x, C, H, W, Ho, Wo};
        return ln_bwd_launch<float, MergeSrc<float>, true>(src, (const float*)dy, gamma, mean, rstd, nullptr, (float*)dx,
                                                           nullptr, nullptr, 1, dgamma, dbeta, rows, 4 * C, s, (float*)workspace);
    }
    return SWIN_ERR_UNSUPPORTED;
}
