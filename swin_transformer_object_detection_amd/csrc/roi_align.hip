// RoIAlign ('avg' pooling, aligned flag, adaptive sampling grid) for gfx950.
// Semantics: mmcv.ops.roi_align as called at base_roi_extractor.py:49-55 /
// single_level_roi_extractor.py:93-97 / structures.py:353-354 of the reference.
//
// Two memory layouts for the feature map and the output:
//   channels_last = 0 : input (N,C,H,W), output (K,C,ph,pw)   -- mmcv's layout, one thread per output element
//   channels_last = 1 : input (N,H,W,C), output (K,ph,pw,C)   -- lanes run along C: every bilinear corner is one
//                       coalesced read, and the backward's fp32 atomics are 256-byte contiguous per wave
// Features may be fp32 or bf16 (converted on load, which is what the reference's force_fp32 does);
// output and all arithmetic are fp32.  Gradient w.r.t. features is fp32 (atomics), caller zeroes it.
#include <algorithm>

#include "common.h"

struct RoiGeom {
    float start_w, start_h, bin_w, bin_h;
    int grid_h, grid_w, batch;
    float count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int aligned, int ph, int pw, int sampling_ratio) {
    RoiGeom g;
    g.batch = (int)roi[0];
    float off = aligned ? 0.5f : 0.0f;
    g.start_w = roi[1] * scale - off;
    g.start_h = roi[2] * scale - off;
    float end_w = roi[3] * scale - off;
    float end_h = roi[4] * scale - off;
    float rw = end_w - g.start_w, rh = end_h - g.start_h;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    g.bin_h = rh / (float)ph;
    g.bin_w = rw / (float)pw;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw);
    int c = g.grid_h * g.grid_w;
    g.count = (float)(c > 1 ? c : 1);
    return g;
}

struct Bilin { int yl, xl, yh, xh; float w1, w2, w3, w4; bool valid; };

__device__ __forceinline__ Bilin bilin_setup(int H, int W, float y, float x) {
    Bilin b;
    b.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (!b.valid) { b.yl = b.xl = b.yh = b.xh = 0; b.w1 = b.w2 = b.w3 = b.w4 = 0.f; return b; }
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
    b.yl = yl; b.xl = xl; b.yh = yh; b.xh = xh;
    b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
    return b;
}

// ----------------------------------------------------------------------------- NCHW
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_nchw(const T* __restrict__ in, const float* __restrict__ rois,
                                                          float* __restrict__ out, int C, int H, int W, int64_t total,
                                                          int ph, int pw, float scale, int sr, int aligned) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int j = (int)(idx % pw), i = (int)((idx / pw) % ph);
        int c = (int)((idx / pw / ph) % C);
        int64_t k = idx / pw / ph / C;
        RoiGeom g = roi_geom(rois + 5 * k, scale, aligned, ph, pw, sr);
        const T* p = in + ((int64_t)g.batch * C + c) * H * W;
        float acc = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
            float y = g.start_h + (float)i * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                float x = g.start_w + (float)j * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                Bilin b = bilin_setup(H, W, y, x);
                if (!b.valid) continue;
                acc += b.w1 * Elt<T>::ld(p + b.yl * W + b.xl) + b.w2 * Elt<T>::ld(p + b.yl * W + b.xh) +
                       b.w3 * Elt<T>::ld(p + b.yh * W + b.xl) + b.w4 * Elt<T>::ld(p + b.yh * W + b.xh);
            }
        }
        out[idx] = acc / g.count;
    }
}

// NCHW forward for few outputs with large adaptive grids -- the mask targets (mask_target.py:66-122: 28x28 bins over RoIs hundreds
// of pixels wide on the full-resolution 1-channel gt mask), where a thread per bin would walk hundreds of samples serially.
// Round 2 gave every bin a wave (lanes = samples): 152 us per step, VALU-bound -- every wave instruction of the per-bin set-up costs a
// full issue slot, and 196 samples x (set-up + 4 loads) per bin.  Round 3: the sum is SEPARABLE.  A bin's value is the sum over its
// sampling grid of bilinear interpolations; the bilinear weight of pixel (r, c) for sample (y, x) is wy(y, r) * wx(x, c) and the validity
// test is per axis, so   out[i][j] = sum_ix [ wl(x) V_i[lo(x)] + wh(x) V_i[hi(x)] ] / count,   V_i[c] = sum_r WY_i[r] P[r][c],
// WY_i[r] = sum over bin-row i's y samples of wy(y, r).  One wave per (RoI, channel, bin-row i): WY_i once (lanes = y samples),
// V_i over the RoI's columns (lanes = columns, rows in the loop), then lanes = the row's pw bins walking their x samples.  ~30 wave
// instructions per bin instead of ~400.  Same clamping rules as bilin_setup (roi_align_cuda_kernel.cuh:28-70 bilinear_interpolate);
// fp32 sums in a different order (tests: 1e-4).  Rows whose footprint exceeds the LDS vectors (> 64 pixel rows per bin, > 1408
// columns per RoI) take the sample-by-sample loop.
struct Axis1 { int lo, hi; float wl, wh; bool valid; };
__device__ __forceinline__ Axis1 axis_setup(int n, float v) {
    Axis1 a;
    a.valid = !(v < -1.0f || v > (float)n);
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= n - 1) { hi = lo = n - 1; v = (float)lo; } else hi = lo + 1;
    a.lo = lo; a.hi = hi;
    a.wh = v - (float)lo; a.wl = 1.f - a.wh;
    return a;
}

constexpr int RA_VMAX = 1408;

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_nchw_rows(const T* __restrict__ in, const float* __restrict__ rois,
                                                               float* __restrict__ out, int C, int H, int W, int tasks,
                                                               int ph, int pw, float scale, int sr, int aligned) {
    __shared__ float wy_s[4][64];
    __shared__ float v_s[4][RA_VMAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* WY = wy_s[wv];
    float* V = v_s[wv];
    for (unsigned task = blockIdx.x * 4u + wv; task < (unsigned)tasks; task += gridDim.x * 4u) {
        const unsigned q1 = task / (unsigned)ph, k = q1 / (unsigned)C;
        const int i = (int)(task - q1 * ph), c = (int)(q1 - k * C);
        const RoiGeom g = roi_geom(rois + 5 * k, scale, aligned, ph, pw, sr);
        const T* p = in + ((int64_t)g.batch * C + c) * H * W;
        float* o = out + (((int64_t)k * C + c) * ph + i) * pw;
        const float ys = g.bin_h / (float)g.grid_h, xs = g.bin_w / (float)g.grid_w;
        const float y_base = g.start_h + (float)i * g.bin_h;
        const int r0 = axis_setup(H, y_base + .5f * ys).lo, r1 = axis_setup(H, y_base + ((float)(g.grid_h - 1) + .5f) * ys).hi;
        const int c0 = axis_setup(W, g.start_w + .5f * xs).lo;
        const int c1 = axis_setup(W, g.start_w + (float)(pw - 1) * g.bin_w + ((float)(g.grid_w - 1) + .5f) * xs).hi;
        const int R = r1 - r0 + 1, Cs = c1 - c0 + 1;
        if (g.grid_h < 1 || g.grid_w < 1) {                          // degenerate RoI: no samples (count = 1)
            if (lane < pw) o[lane] = 0.f;
            continue;
        }
        if (R > 64 || Cs > RA_VMAX || R < 1 || Cs < 1) {              // (wave-uniform) sample by sample, a bin at a time
            for (int j = 0; j < pw; ++j) {
                float acc = 0.f;
                const int ns = g.grid_h * g.grid_w;
                for (int s = lane; s < ns; s += 64) {
                    const int iy = s / g.grid_w, ix = s - iy * g.grid_w;
                    const float y = y_base + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                    const float x = g.start_w + (float)j * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                    const Bilin b = bilin_setup(H, W, y, x);
                    if (!b.valid) continue;
                    acc += b.w1 * Elt<T>::ld(p + b.yl * W + b.xl) + b.w2 * Elt<T>::ld(p + b.yl * W + b.xh) +
                           b.w3 * Elt<T>::ld(p + b.yh * W + b.xl) + b.w4 * Elt<T>::ld(p + b.yh * W + b.xh);
                }
                acc = wave_sum(acc);
                if (lane == 0) o[j] = acc / g.count;
            }
            continue;
        }
        // ---- WY: lanes = y samples of bin-row i
        WY[lane] = 0.f;
        __builtin_amdgcn_wave_barrier();
        for (int iy = lane; iy < g.grid_h; iy += 64) {
            const Axis1 a = axis_setup(H, y_base + ((float)iy + .5f) * g.bin_h / (float)g.grid_h);
            if (a.valid) { atomicAdd(&WY[a.lo - r0], a.wl); atomicAdd(&WY[a.hi - r0], a.wh); }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- V[c] = sum_r WY[r] P[r0 + r][c0 + c]: lanes = columns (coalesced rows), four rows in flight
        for (int cc = lane; cc < Cs; cc += 64) {
            const T* q = p + (int64_t)r0 * W + c0 + cc;
            float acc = 0.f;
            int r = 0;
            for (; r + 4 <= R; r += 4) {
                const float va = Elt<T>::ld(q + (int64_t)r * W), vb = Elt<T>::ld(q + (int64_t)(r + 1) * W);
                const float vc = Elt<T>::ld(q + (int64_t)(r + 2) * W), vd = Elt<T>::ld(q + (int64_t)(r + 3) * W);
                acc += WY[r] * va; acc += WY[r + 1] * vb; acc += WY[r + 2] * vc; acc += WY[r + 3] * vd;
            }
            for (; r < R; ++r) acc += WY[r] * Elt<T>::ld(q + (int64_t)r * W);
            V[cc] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lanes = bins of the row: their x samples against V
        for (int j = lane; j < pw; j += 64) {
            float acc = 0.f;
            const float x_base = g.start_w + (float)j * g.bin_w;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const Axis1 a = axis_setup(W, x_base + ((float)ix + .5f) * g.bin_w / (float)g.grid_w);
                if (a.valid) acc += a.wl * V[a.lo - c0] + a.wh * V[a.hi - c0];
            }
            o[j] = acc / g.count;
        }
        __builtin_amdgcn_wave_barrier();                              // everyone is done with WY / V before the next task rewrites them
    }
}

__global__ __launch_bounds__(256) void roi_align_bwd_nchw(const float* __restrict__ gout, const float* __restrict__ rois,
                                                          float* __restrict__ gin, int C, int H, int W, int64_t total,
                                                          int ph, int pw, float scale, int sr, int aligned) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int j = (int)(idx % pw), i = (int)((idx / pw) % ph);
        int c = (int)((idx / pw / ph) % C);
        int64_t k = idx / pw / ph / C;
        RoiGeom g = roi_geom(rois + 5 * k, scale, aligned, ph, pw, sr);
        float* p = gin + ((int64_t)g.batch * C + c) * H * W;
        float go = gout[idx];
        for (int iy = 0; iy < g.grid_h; ++iy) {
            float y = g.start_h + (float)i * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                float x = g.start_w + (float)j * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                Bilin b = bilin_setup(H, W, y, x);
                if (!b.valid) continue;
                atomicAdd(p + b.yl * W + b.xl, go * b.w1 / g.count);
                atomicAdd(p + b.yl * W + b.xh, go * b.w2 / g.count);
                atomicAdd(p + b.yh * W + b.xl, go * b.w3 / g.count);
                atomicAdd(p + b.yh * W + b.xh, go * b.w4 / g.count);
            }
        }
    }
}

// ----------------------------------------------------------------------------- NHWC
// one thread per (k, i, j, 4-channel group); a wave covers 256 consecutive channels of one bin
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_nhwc(const T* __restrict__ in, const float* __restrict__ rois,
                                                          float* __restrict__ out, int C, int H, int W, int64_t total,
                                                          int ph, int pw, float scale, int sr, int aligned) {
    const int cg = C / 4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int c4 = (int)(idx % cg) * 4;
        int64_t t = idx / cg;
        int j = (int)(t % pw); t /= pw;
        int i = (int)(t % ph);
        int64_t k = t / ph;
        RoiGeom g = roi_geom(rois + 5 * k, scale, aligned, ph, pw, sr);
        const T* p = in + (int64_t)g.batch * H * W * C + c4;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < g.grid_h; ++iy) {
            float y = g.start_h + (float)i * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                float x = g.start_w + (float)j * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                Bilin b = bilin_setup(H, W, y, x);
                if (!b.valid) continue;
                const T* p1 = p + ((int64_t)b.yl * W + b.xl) * C;
                const T* p2 = p + ((int64_t)b.yl * W + b.xh) * C;
                const T* p3 = p + ((int64_t)b.yh * W + b.xl) * C;
                const T* p4 = p + ((int64_t)b.yh * W + b.xh) * C;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[e] += b.w1 * Elt<T>::ld(p1 + e) + b.w2 * Elt<T>::ld(p2 + e) + b.w3 * Elt<T>::ld(p3 + e) +
                              b.w4 * Elt<T>::ld(p4 + e);
            }
        }
        float4 o = {acc[0] / g.count, acc[1] / g.count, acc[2] / g.count, acc[3] / g.count};
        *(float4*)(out + idx * 4) = o;
    }
}

// NHWC backward.  grid (K, S): the blocks of one RoI share its footprint rows (y % S), threads = channels
// (256-byte contiguous atomics per wave).  The bilinear weights are separable: per bin row / bin column the block
// first sums the 1-D weights every sample puts on each feature row / column (LDS tables), then walks the PIXELS of
// the RoI footprint, gathers for each pixel the <= 2x2 bins whose footprints contain it and issues ONE atomic per
// (pixel, channel).  The atomic rate bounds this kernel (guide G12): 4*grid^2 atomics per bin in the per-sample
// form, (grid+1)^2 per bin when scattering per bin, ~grid^2 here (adjacent bins share their border pixels).
#define RA_MAXP 16       // pooled size limit of the fast path (7 and 14 in every swin config)
#define RA_MAXT 352      // RoI footprint rows/cols covered by the tables (1333 / 4 = 334)
#define RA_MAXW (RA_MAXT + 3 * RA_MAXP)   // ragged per-bin weight table of one axis

// one sample of bilinear_interpolate along one axis: false = contributes nothing; else pixels lo/hi, weights hgh/l
__device__ __forceinline__ bool axis_sample(float t, int size, int& lo, int& hi, float& l, float& hgh) {
    if (t < -1.0f || t > (float)size) return false;
    if (t <= 0.f) t = 0.f;
    lo = (int)t;
    if (lo >= size - 1) { hi = lo = size - 1; t = (float)lo; } else hi = lo + 1;
    l = t - (float)lo; hgh = 1.f - l;
    return true;
}

struct RoiBwdLds {
    float Wy[RA_MAXW], Wx[RA_MAXW];                       // ragged: bin i owns [off[i], off[i] + N[i])
    int Y0[RA_MAXP], NY[RA_MAXP], YO[RA_MAXP], X0[RA_MAXP], NX[RA_MAXP], XO[RA_MAXP];
    unsigned char ylo[RA_MAXT], yhi[RA_MAXT], xlo[RA_MAXT], xhi[RA_MAXT];
    int ymin, ymax, xmin, xmax, slow;
    float R[RA_MAXP][256];      // per-thread (channel) row intermediates of the separable form
};

// pixel extent [p0, p0 + np) touched by the samples of bin b along one axis
__device__ __forceinline__ void axis_extent(float start, float bin, int b, int grid, int size, int& p0, int& np) {
    int first = 1 << 30, last = -1;
    for (int g = 0; g < grid; ++g) {
        int lo, hi; float l, h;
        if (!axis_sample(start + (float)b * bin + ((float)g + .5f) * bin / (float)grid, size, lo, hi, l, h)) continue;
        first = lo < first ? lo : first; last = hi > last ? hi : last;
    }
    if (last < 0) { p0 = 0; np = 0; } else { p0 = first; np = last - first + 1; }
}

__device__ __forceinline__ void axis_fill(float start, float bin, int b, int grid, int size, float* Wt, int p0, int np) {
    for (int f = 0; f < np; ++f) Wt[f] = 0.f;
    for (int g = 0; g < grid; ++g) {
        int lo, hi; float l, h;
        if (!axis_sample(start + (float)b * bin + ((float)g + .5f) * bin / (float)grid, size, lo, hi, l, h)) continue;
        Wt[lo - p0] += h; Wt[hi - p0] += l;
    }
}

// bins (first, last+1) whose footprint [P0[i], P0[i]+NP[i]) contains pixel p
__device__ __forceinline__ void bin_range(const int* P0, const int* NP, int nb, int p, unsigned char& lo, unsigned char& hi) {
    int l = nb, h = 0;
    for (int i = 0; i < nb; ++i)
        if (p >= P0[i] && p < P0[i] + NP[i]) { l = i < l ? i : l; h = i + 1; }
    lo = (unsigned char)(l < h ? l : 0); hi = (unsigned char)h;
}

template <typename TG>
__device__ __forceinline__ void roi_bwd_body(RoiBwdLds& L, float* __restrict__ gin, const TG* __restrict__ gout,
                                             const float* __restrict__ roi, int64_t k, int C, int H, int W, int ph, int pw,
                                             float scale, int sr, int aligned) {
    const int t = threadIdx.x;
    RoiGeom g = roi_geom(roi, scale, aligned, ph, pw, sr);
    const bool fast = ph <= RA_MAXP && pw <= RA_MAXP;
    if (fast) {
        if (t < ph) axis_extent(g.start_h, g.bin_h, t, g.grid_h, H, L.Y0[t], L.NY[t]);
        else if (t >= 32 && t < 32 + pw) axis_extent(g.start_w, g.bin_w, t - 32, g.grid_w, W, L.X0[t - 32], L.NX[t - 32]);
    }
    __syncthreads();
    if (t == 0) {
        int slow = fast ? 0 : 1;
        if (fast) {
            int y0 = 1 << 30, y1 = 0, x0 = 1 << 30, x1 = 0, oy = 0, ox = 0;
            for (int i = 0; i < ph; ++i) {
                L.YO[i] = oy; oy += L.NY[i];
                if (L.NY[i] > 0) { y0 = min(y0, L.Y0[i]); y1 = max(y1, L.Y0[i] + L.NY[i]); }
            }
            for (int j = 0; j < pw; ++j) {
                L.XO[j] = ox; ox += L.NX[j];
                if (L.NX[j] > 0) { x0 = min(x0, L.X0[j]); x1 = max(x1, L.X0[j] + L.NX[j]); }
            }
            if (y1 <= y0 || x1 <= x0) { y0 = y1 = x0 = x1 = 0; }
            L.ymin = y0; L.ymax = y1; L.xmin = x0; L.xmax = x1;
            if (y1 - y0 > RA_MAXT || x1 - x0 > RA_MAXT || oy > RA_MAXW || ox > RA_MAXW) slow = 1;
        }
        L.slow = slow;
    }
    __syncthreads();
    if (!L.slow) {
        if (t < ph) axis_fill(g.start_h, g.bin_h, t, g.grid_h, H, L.Wy + L.YO[t], L.Y0[t], L.NY[t]);
        else if (t >= 32 && t < 32 + pw) axis_fill(g.start_w, g.bin_w, t - 32, g.grid_w, W, L.Wx + L.XO[t - 32], L.X0[t - 32], L.NX[t - 32]);
    }
    float* base = gin + (int64_t)g.batch * H * W * C;
    const TG* gob = gout + k * ph * pw * C;
    if (!L.slow) {
        const int ny = L.ymax - L.ymin, nx = L.xmax - L.xmin;
        for (int q = t; q < ny; q += 256) bin_range(L.Y0, L.NY, ph, L.ymin + q, L.ylo[q], L.yhi[q]);
        for (int q = t; q < nx; q += 256) bin_range(L.X0, L.NX, pw, L.xmin + q, L.xlo[q], L.xhi[q]);
        __syncthreads();
        const float inv = 1.0f / g.count;
        for (int c = t; c < C; c += 256) {
            for (int ry = blockIdx.y; ry < ny; ry += gridDim.y) {
                const int y = L.ymin + ry, ilo = L.ylo[ry], ihi = L.yhi[ry];
                // separable step 1: R[j] = sum_i Wy[i][y] * gout[i][j][c] for this feature row (pw independent loads
                // per bin row, issued back to back); kept in this thread's LDS column
                if (ihi - ilo <= 2) {
                    const int i0 = ilo, i1 = (ilo + 1 < ihi) ? ilo + 1 : ilo;
                    const int f0 = y - L.Y0[i0], f1 = y - L.Y0[i1];
                    const float w0 = (ihi > ilo && f0 >= 0 && f0 < L.NY[i0]) ? L.Wy[L.YO[i0] + f0] : 0.f;
                    const float w1 = (i1 != i0 && f1 >= 0 && f1 < L.NY[i1]) ? L.Wy[L.YO[i1] + f1] : 0.f;
                    const TG* g0 = gob + (int64_t)i0 * pw * C + c;
                    const TG* g1 = gob + (int64_t)i1 * pw * C + c;
                    for (int j = 0; j < pw; ++j) L.R[j][t] = w0 * Elt<TG>::ld(g0 + j * C) + w1 * Elt<TG>::ld(g1 + j * C);
                } else {
                    for (int j = 0; j < pw; ++j) {
                        float r = 0.f;
                        for (int i = ilo; i < ihi; ++i) {
                            const int fy = y - L.Y0[i];
                            if (fy >= 0 && fy < L.NY[i]) r += L.Wy[L.YO[i] + fy] * Elt<TG>::ld(gob + ((int64_t)i * pw + j) * C + c);
                        }
                        L.R[j][t] = r;
                    }
                }
                // step 2: one atomic per pixel of the row
                float* rowp = base + ((int64_t)y * W + L.xmin) * C + c;
                for (int rx = 0; rx < nx; ++rx) {
                    const int x = L.xmin + rx, jlo = L.xlo[rx], jhi = L.xhi[rx];
                    float acc = 0.f;
                    for (int j = jlo; j < jhi; ++j) {
                        const int fx = x - L.X0[j];
                        if (fx >= 0 && fx < L.NX[j]) acc += L.Wx[L.XO[j] + fx] * L.R[j][t];
                    }
                    if (acc != 0.f) atomicAdd(rowp + (int64_t)rx * C, acc * inv);
                }
            }
        }
    } else if (blockIdx.y == 0) {        // per-sample path (very large sampling grids / footprints)
        for (int c = t; c < C; c += 256) {
            const TG* go = gob + c;
            for (int i = 0; i < ph; ++i)
                for (int j = 0; j < pw; ++j) {
                    const float gv = Elt<TG>::ld(go + (i * pw + j) * C);
                    for (int iy = 0; iy < g.grid_h; ++iy) {
                        float y = g.start_h + (float)i * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                        for (int ix = 0; ix < g.grid_w; ++ix) {
                            float x = g.start_w + (float)j * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                            Bilin b = bilin_setup(H, W, y, x);
                            if (!b.valid) continue;
                            float* p = base + c;
                            atomicAdd(p + ((int64_t)b.yl * W + b.xl) * C, gv * b.w1 / g.count);
                            atomicAdd(p + ((int64_t)b.yl * W + b.xh) * C, gv * b.w2 / g.count);
                            atomicAdd(p + ((int64_t)b.yh * W + b.xl) * C, gv * b.w3 / g.count);
                            atomicAdd(p + ((int64_t)b.yh * W + b.xh) * C, gv * b.w4 / g.count);
                        }
                    }
                }
        }
    }
}

__global__ __launch_bounds__(256) void roi_align_bwd_nhwc(const float* __restrict__ gout, const float* __restrict__ rois,
                                                          float* __restrict__ gin, int C, int H, int W, int K,
                                                          int ph, int pw, float scale, int sr, int aligned) {
    __shared__ RoiBwdLds L;
    const int64_t k = blockIdx.x;
    roi_bwd_body<float>(L, gin, gout, rois + 5 * k, k, C, H, W, ph, pw, scale, sr, aligned);
}

// ----------------------------------------------------------------------------- multi-level (FPN) NHWC
// SingleRoIExtractor.forward (single_level_roi_extractor.py:82-97) in ONE launch: every RoI reads the pyramid level
// `lvl[k]` the host mapped it to (map_roi_levels, :47-51), so there is no per-level nonzero / gather / scatter and
// no host synchronisation.  RoIs with lvl < 0 are skipped (output rows stay zero): slots of a fixed-size sample.
struct MLFeats { const void* p[4]; int H[4], W[4]; float scale[4]; };

template <typename T, typename TO>
__global__ __launch_bounds__(256) void roi_align_ml_fwd_nhwc(MLFeats F, const float* __restrict__ rois, const int* __restrict__ lvl,
                                                             TO* __restrict__ out, int C, int64_t total, int ph, int pw,
                                                             int sr, int aligned) {
    const int cg = C / 4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int c4 = (int)(idx % cg) * 4;
        int64_t t = idx / cg;
        int j = (int)(t % pw); t /= pw;
        int i = (int)(t % ph);
        int64_t k = t / ph;
        const int l = lvl[k];
        float4 o = {0.f, 0.f, 0.f, 0.f};
        if (l >= 0) {
            const int H = F.H[l], W = F.W[l];
            RoiGeom g = roi_geom(rois + 5 * k, F.scale[l], aligned, ph, pw, sr);
            const T* p = (const T*)F.p[l] + (int64_t)g.batch * H * W * C + c4;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int iy = 0; iy < g.grid_h; ++iy) {
                float y = g.start_h + (float)i * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                for (int ix = 0; ix < g.grid_w; ++ix) {
                    float x = g.start_w + (float)j * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                    Bilin b = bilin_setup(H, W, y, x);
                    if (!b.valid) continue;
                    const T* p1 = p + ((int64_t)b.yl * W + b.xl) * C;
                    const T* p2 = p + ((int64_t)b.yl * W + b.xh) * C;
                    const T* p3 = p + ((int64_t)b.yh * W + b.xl) * C;
                    const T* p4 = p + ((int64_t)b.yh * W + b.xh) * C;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[e] += b.w1 * Elt<T>::ld(p1 + e) + b.w2 * Elt<T>::ld(p2 + e) + b.w3 * Elt<T>::ld(p3 + e) +
                                  b.w4 * Elt<T>::ld(p4 + e);
                }
            }
            o = float4{acc[0] / g.count, acc[1] / g.count, acc[2] / g.count, acc[3] / g.count};
        }
        if constexpr (sizeof(TO) == 4) {
            *(float4*)(out + idx * 4) = o;
        } else {                                      // bf16 output: the RoI heads consume it directly (no cast pass)
            bf16x4 ob = {(bf16)o.x, (bf16)o.y, (bf16)o.z, (bf16)o.w};
            *(bf16x4*)(out + idx * 4) = ob;
        }
    }
}

// The same forward, separably, for C = 256 (the FPN's width: the 64 four-channel threads of a bin are exactly one wave; round 3).
// In the kernel above all 64 lanes of a bin run the identical bilinear set-up for each of its grid_h x grid_w samples (about half of
// the instructions) and load 4 pixels per sample.  A bin's value is sum_r sum_c WY[r] WX[c] P[r][c] with WY / WX the summed 1-D
// weights of its y / x samples (validity and clamping are per axis: see roi_align_fwd_nchw_rows): the wave builds the two weight
// vectors once (lanes = samples) and walks the (bin_h + 2) x (bin_w + 2) pixel footprint -- 81 loads instead of 256 for an 8 x 8 grid,
// one multiply per pixel instead of a set-up per sample.  fp32 sums in a different order than sample by sample (tests: 1e-4).
template <typename T, typename TO>
__global__ __launch_bounds__(256) void roi_align_ml_fwd_nhwc_sep(MLFeats F, const float* __restrict__ rois, const int* __restrict__ lvl,
                                                                 TO* __restrict__ out, int bins, int ph, int pw, int sr, int aligned) {
    constexpr int C = 256;
    __shared__ float wts[4][2][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c4 = lane * 4;
    float* WY = wts[wv][0];
    float* WX = wts[wv][1];
    for (unsigned bin = blockIdx.x * 4u + wv; bin < (unsigned)bins; bin += gridDim.x * 4u) {
        const unsigned q1 = bin / (unsigned)pw, k = q1 / (unsigned)ph;
        const int j = (int)(bin - q1 * pw), i = (int)(q1 - k * ph);
        const int l = lvl[k];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        float cnt = 1.f;
        if (l >= 0) {
            const int H = F.H[l], W = F.W[l];
            const RoiGeom g = roi_geom(rois + 5 * k, F.scale[l], aligned, ph, pw, sr);
            cnt = g.count;
            const T* p = (const T*)F.p[l] + (int64_t)g.batch * H * W * C + c4;
            const float ys = g.bin_h / (float)g.grid_h, xs = g.bin_w / (float)g.grid_w;
            const float y_base = g.start_h + (float)i * g.bin_h, x_base = g.start_w + (float)j * g.bin_w;
            const int r0 = axis_setup(H, y_base + .5f * ys).lo, r1 = axis_setup(H, y_base + ((float)(g.grid_h - 1) + .5f) * ys).hi;
            const int c0 = axis_setup(W, x_base + .5f * xs).lo, c1 = axis_setup(W, x_base + ((float)(g.grid_w - 1) + .5f) * xs).hi;
            const int R = r1 - r0 + 1, Cn = c1 - c0 + 1;
            if (g.grid_h < 1 || g.grid_w < 1) {
                // no samples: zeros
            } else if (R > 64 || Cn > 64 || R < 1 || Cn < 1) {           // (wave-uniform) sample by sample
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    const float y = y_base + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        const float x = x_base + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
                        const Bilin b = bilin_setup(H, W, y, x);
                        if (!b.valid) continue;
                        const T* p1 = p + ((int64_t)b.yl * W + b.xl) * C;
                        const T* p2 = p + ((int64_t)b.yl * W + b.xh) * C;
                        const T* p3 = p + ((int64_t)b.yh * W + b.xl) * C;
                        const T* p4 = p + ((int64_t)b.yh * W + b.xh) * C;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[e] += b.w1 * Elt<T>::ld(p1 + e) + b.w2 * Elt<T>::ld(p2 + e) + b.w3 * Elt<T>::ld(p3 + e) +
                                      b.w4 * Elt<T>::ld(p4 + e);
                    }
                }
            } else {
                WY[lane] = 0.f;
                WX[lane] = 0.f;
                __builtin_amdgcn_wave_barrier();
                for (int iy = lane; iy < g.grid_h; iy += 64) {
                    const Axis1 a = axis_setup(H, y_base + ((float)iy + .5f) * g.bin_h / (float)g.grid_h);
                    if (a.valid) { atomicAdd(&WY[a.lo - r0], a.wl); atomicAdd(&WY[a.hi - r0], a.wh); }
                }
                for (int ix = lane; ix < g.grid_w; ix += 64) {
                    const Axis1 a = axis_setup(W, x_base + ((float)ix + .5f) * g.bin_w / (float)g.grid_w);
                    if (a.valid) { atomicAdd(&WX[a.lo - c0], a.wl); atomicAdd(&WX[a.hi - c0], a.wh); }
                }
                __builtin_amdgcn_wave_barrier();
                for (int r = 0; r < R; ++r) {
                    const float wy = WY[r];
                    if (wy == 0.f) continue;                                  // wave-uniform
                    const T* q = p + ((int64_t)(r0 + r) * W + c0) * C;
                    for (int cc = 0; cc < Cn; cc += 4) {                       // four pixels in flight (a weight of zero: the pixel at cc again)
                        float w[4];
                        const T* qq[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool in = cc + u < Cn;
                            w[u] = in ? wy * WX[cc + u] : 0.f;
                            qq[u] = q + (int64_t)(in ? cc + u : cc) * C;
                        }
                        float v[4][4];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[u][e] = Elt<T>::ld(qq[u] + e);
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[e] += w[u] * v[u][e];
                    }
                }
                __builtin_amdgcn_wave_barrier();                              // the weights are read before the next bin zeroes them
            }
        }
        const float4 o = float4{acc[0] / cnt, acc[1] / cnt, acc[2] / cnt, acc[3] / cnt};
        const int64_t oi = ((int64_t)bin * 64 + lane) * 4;
        if constexpr (sizeof(TO) == 4) {
            *(float4*)(out + oi) = o;
        } else {
            bf16x4 ob = {(bf16)o.x, (bf16)o.y, (bf16)o.z, (bf16)o.w};
            *(bf16x4*)(out + oi) = ob;
        }
    }
}

struct MLGrads { float* p[4]; int H[4], W[4]; float scale[4]; };

template <typename TG>
__global__ __launch_bounds__(256) void roi_align_ml_bwd_nhwc(MLGrads G, const TG* __restrict__ gout, const float* __restrict__ rois,
                                                             const int* __restrict__ lvl, int C, int K, int ph, int pw, int sr,
                                                             int aligned) {
    __shared__ RoiBwdLds L;
    const int64_t k = blockIdx.x;
    const int l = lvl[k];
    if (l < 0) return;
    roi_bwd_body<TG>(L, G.p[l], gout, rois + 5 * k, k, C, G.H[l], G.W[l], ph, pw, G.scale[l], sr, aligned);
}

// row splits per RoI: enough blocks to fill the chip when K is small
static inline int ra_row_splits(int K) {
    int s = 4096 / (K > 0 ? K : 1);
    return s < 1 ? 1 : (s > 16 ? 16 : s);
}

static inline int ra_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 65535 ? (b > 0 ? b : 1) : 65535);
}

extern "C" int roi_align_fwd(const void* input, const float* rois, float* output, int N, int C, int H, int W, int K,
                             int ph, int pw, float spatial_scale, int sampling_ratio, int aligned, int channels_last,
                             int in_dtype, void* stream) {
    if (K == 0) return SWIN_OK;
    if (!input || !rois || !output || N <= 0 || C <= 0 || H <= 0 || W <= 0 || K < 0 || ph <= 0 || pw <= 0)
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (channels_last) {
        if (C % 4) return SWIN_ERR_UNSUPPORTED;
        int64_t total = (int64_t)K * ph * pw * (C / 4);
        if (in_dtype == SWIN_F32)
            roi_align_fwd_nhwc<float><<<ra_blocks(total), 256, 0, s>>>((const float*)input, rois, output, C, H, W, total, ph, pw,
                                                                      spatial_scale, sampling_ratio, aligned);
        else if (in_dtype == SWIN_BF16)
            roi_align_fwd_nhwc<bf16><<<ra_blocks(total), 256, 0, s>>>((const bf16*)input, rois, output, C, H, W, total, ph, pw,
                                                                     spatial_scale, sampling_ratio, aligned);
        else return SWIN_ERR_UNSUPPORTED;
    } else {
        int64_t total = (int64_t)K * C * ph * pw;
        const bool wave = sampling_ratio <= 0 && total <= (1 << 18);      // few bins, adaptive grid: wave per bin
        if (in_dtype != SWIN_F32 && in_dtype != SWIN_BF16) return SWIN_ERR_UNSUPPORTED;
        if (wave) {
            const int tasks = K * C * ph;                               // a wave per (RoI, channel, bin-row)
            int blocks = std::min((tasks + 3) / 4, 256 * 8);
            if (in_dtype == SWIN_F32)
                roi_align_fwd_nchw_rows<float><<<blocks, 256, 0, s>>>((const float*)input, rois, output, C, H, W, tasks, ph, pw,
                                                                      spatial_scale, sampling_ratio, aligned);
            else
                roi_align_fwd_nchw_rows<bf16><<<blocks, 256, 0, s>>>((const bf16*)input, rois, output, C, H, W, tasks, ph, pw,
                                                                     spatial_scale, sampling_ratio, aligned);
        } else if (in_dtype == SWIN_F32)
            roi_align_fwd_nchw<float><<<ra_blocks(total), 256, 0, s>>>((const float*)input, rois, output, C, H, W, total, ph, pw,
                                                                      spatial_scale, sampling_ratio, aligned);
        else
            roi_align_fwd_nchw<bf16><<<ra_blocks(total), 256, 0, s>>>((const bf16*)input, rois, output, C, H, W, total, ph, pw,
                                                                     spatial_scale, sampling_ratio, aligned);
    }
    return swin_launch_status();
}

extern "C" int roi_align_bwd(const float* grad_output, const float* rois, float* grad_input, int N, int C, int H, int W,
                             int K, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned,
                             int channels_last, void* stream) {
    if (K == 0) return SWIN_OK;
    if (!grad_output || !rois || !grad_input || N <= 0 || C <= 0 || H <= 0 || W <= 0 || K < 0 || ph <= 0 || pw <= 0)
        return SWIN_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    int64_t total = (int64_t)K * C * ph * pw;
    if (channels_last)
        roi_align_bwd_nhwc<<<dim3(K, ra_row_splits(K)), 256, 0, s>>>(grad_output, rois, grad_input, C, H, W, K, ph, pw, spatial_scale,
                                             sampling_ratio, aligned);
    else
        roi_align_bwd_nchw<<<ra_blocks(total), 256, 0, s>>>(grad_output, rois, grad_input, C, H, W, total, ph, pw,
                                                           spatial_scale, sampling_ratio, aligned);
    return swin_launch_status();
}

// Multi-level RoIAlign over an FPN pyramid in channels-last memory (see roi_align_ml_fwd_nhwc above).
//   feats[l]: (N, H[l], W[l], C) in_dtype, l < n_levels <= 4;  lvl (K) int32 level per RoI, < 0 = skip (zero row);
//   output (K, ph, pw, C) f32, or bf16 (out_dtype) for bf16 features: one rounding of the fp32 result.
extern "C" int roi_align_multilevel_fwd(const void* const* feats, const int* Hs, const int* Ws, const float* scales,
                                        int n_levels, const float* rois, const int* lvl, void* output, int C, int K,
                                        int ph, int pw, int sampling_ratio, int aligned, int in_dtype, int out_dtype,
                                        void* stream) {
    if (K == 0) return SWIN_OK;
    if (!feats || !Hs || !Ws || !scales || !rois || !lvl || !output || n_levels <= 0 || n_levels > 4 || C <= 0 || K < 0)
        return SWIN_ERR_BAD_ARG;
    if (C % 4) return SWIN_ERR_UNSUPPORTED;
    MLFeats F;
    for (int l = 0; l < 4; ++l) {
        int m = l < n_levels ? l : n_levels - 1;
        F.p[l] = feats[m]; F.H[l] = Hs[m]; F.W[l] = Ws[m]; F.scale[l] = scales[m];
        if (!F.p[l]) return SWIN_ERR_BAD_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t total = (int64_t)K * ph * pw * (C / 4);
    if (C == 256 && (int64_t)K * ph * pw < ((int64_t)1 << 30)) {         // a wave per bin: the separable form
        const int bins = K * ph * pw;
        const int blocks = std::min((bins + 3) / 4, 256 * 16);
        if (in_dtype == SWIN_F32 && out_dtype == SWIN_F32)
            roi_align_ml_fwd_nhwc_sep<float, float><<<blocks, 256, 0, s>>>(F, rois, lvl, (float*)output, bins, ph, pw, sampling_ratio, aligned);
        else if (in_dtype == SWIN_BF16 && out_dtype == SWIN_F32)
            roi_align_ml_fwd_nhwc_sep<bf16, float><<<blocks, 256, 0, s>>>(F, rois, lvl, (float*)output, bins, ph, pw, sampling_ratio, aligned);
        else if (in_dtype == SWIN_BF16 && out_dtype == SWIN_BF16)
            roi_align_ml_fwd_nhwc_sep<bf16, bf16><<<blocks, 256, 0, s>>>(F, rois, lvl, (bf16*)output, bins, ph, pw, sampling_ratio, aligned);
        else return SWIN_ERR_UNSUPPORTED;
        return swin_launch_status();
    }
    if (in_dtype == SWIN_F32 && out_dtype == SWIN_F32)
        roi_align_ml_fwd_nhwc<float, float><<<ra_blocks(total), 256, 0, s>>>(F, rois, lvl, (float*)output, C, total, ph, pw, sampling_ratio, aligned);
    else if (in_dtype == SWIN_BF16 && out_dtype == SWIN_F32)
        roi_align_ml_fwd_nhwc<bf16, float><<<ra_blocks(total), 256, 0, s>>>(F, rois, lvl, (float*)output, C, total, ph, pw, sampling_ratio, aligned);
    else if (in_dtype == SWIN_BF16 && out_dtype == SWIN_BF16)
        roi_align_ml_fwd_nhwc<bf16, bf16><<<ra_blocks(total), 256, 0, s>>>(F, rois, lvl, (bf16*)output, C, total, ph, pw, sampling_ratio, aligned);
    else return SWIN_ERR_UNSUPPORTED;
    return swin_launch_status();
}

// grads[l]: (N, H[l], W[l], C) f32, zeroed by the caller; fp32 atomics.
extern "C" int roi_align_multilevel_bwd(float* const* grads, const int* Hs, const int* Ws, const float* scales, int n_levels,
                                        const void* grad_output, const float* rois, const int* lvl, int C, int K, int ph,
                                        int pw, int sampling_ratio, int aligned, int grad_dtype, void* stream) {
    if (K == 0) return SWIN_OK;
    if (!grads || !Hs || !Ws || !scales || !grad_output || !rois || !lvl || n_levels <= 0 || n_levels > 4 || C <= 0 || K < 0)
        return SWIN_ERR_BAD_ARG;
    MLGrads G;
    for (int l = 0; l < 4; ++l) {
        int m = l < n_levels ? l : n_levels - 1;
        G.p[l] = grads[m]; G.H[l] = Hs[m]; G.W[l] = Ws[m]; G.scale[l] = scales[m];
        if (!G.p[l]) return SWIN_ERR_BAD_ARG;
    }
    const dim3 grid(K, ra_row_splits(K));
    if (grad_dtype == SWIN_F32)
        roi_align_ml_bwd_nhwc<float><<<grid, 256, 0, (hipStream_t)stream>>>(G, (const float*)grad_output, rois, lvl, C, K, ph, pw, sampling_ratio, aligned);
    else if (grad_dtype == SWIN_BF16)
        roi_align_ml_bwd_nhwc<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(G, (const bf16*)grad_output, rois, lvl, C, K, ph, pw, sampling_ratio, aligned);
    else return SWIN_ERR_UNSUPPORTED;
    return swin_launch_status();
}

// ----------------------------------------------------------------------------- multi-level NHWC backward, GATHER form (round 3)
// The scatter form above is bound by the chip's float-atomic rate (~400 K footprint pixels x 256 channels x 4 B per RoI set at
// 1.3 TB/s), needs the whole pyramid's fp32 accumulator zero-filled first (174 MB at 2x800x1280) and cast afterwards.  Here every
// 8 x 8-pixel tile of every pyramid level is OWNED by thread blocks that walk the list of RoIs touching it and write the tile's
// gradient ONCE, in the feature dtype: no atomics on the data path, no zero fill, no cast, and a summation order that does not
// depend on scheduling (the lists are sorted).  single_level_roi_extractor.py:93-97 differentiated; all RoI sets pooled from the
// same pyramid (the bbox head's 7x7 and the mask head's 14x14 RoIs of an R-CNN stage) in one pass.
//   1 roi_gather_count   one thread per RoI: ++count[tile] for every tile its (conservative) footprint rectangle overlaps
//   2 roi_gather_scan    one block: offsets = exclusive scan of the counts; counts and cursors zeroed for the next use
//   3 roi_gather_fill    one thread per RoI: list[offset[tile] + cursor[tile]++] = (set, roi)
//   4 roi_gather_main    block = (tile, 256-channel chunk), thread = (4 consecutive channels, pair of tile rows): per listed RoI the separable
//                        weights Wy[8][ph], Wx[8][pw] of the tile's rows / columns (the forward's sampling grid, summed per pixel),
//                        then acc[y][x] += Wx[x][j] * sum_i Wy[y][i] * gout[i][j][c].
#define RG_TS 8            // tile side in pixels
#define RG_CH 256          // channels per block: a thread owns FOUR consecutive channels (8- / 16-byte loads and stores)
#define RG_MAXL 1024       // list entries sorted in LDS (longer lists are walked unsorted)

__device__ __forceinline__ f32x4 rg_load4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 rg_load4(const bf16* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void rg_store4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void rg_store4(bf16* p, f32x4 v) { *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}; }

struct RGLevels { int H[4], W[4], ty[4], tx[4], base[4]; float scale[4]; int n_levels, N, total; };
struct RGSets { const void* gout[4]; const float* rois[4]; const int* lvl[4]; int K[4], ph[4], pw[4], first[4]; int n, Ktot; };

// footprint rectangle of a RoI on its level in tile units; false: it touches nothing
__device__ __forceinline__ bool rg_rect(const RGLevels& Lv, const float* roi, int l, int ph, int pw, int sr, int aligned, int& n,
                                        int& ty0, int& ty1, int& tx0, int& tx1) {
    const float scale = l == 0 ? Lv.scale[0] : (l == 1 ? Lv.scale[1] : (l == 2 ? Lv.scale[2] : Lv.scale[3]));
    const RoiGeom g = roi_geom(roi, scale, aligned, ph, pw, sr);
    n = g.batch;
    if (n < 0 || n >= Lv.N || g.grid_h <= 0 || g.grid_w <= 0) return false;
    const int H = l == 0 ? Lv.H[0] : (l == 1 ? Lv.H[1] : (l == 2 ? Lv.H[2] : Lv.H[3]));
    const int W = l == 0 ? Lv.W[0] : (l == 1 ? Lv.W[1] : (l == 2 ? Lv.W[2] : Lv.W[3]));
    const float y0 = g.start_h, y1 = g.start_h + g.bin_h * (float)ph, x0 = g.start_w, x1 = g.start_w + g.bin_w * (float)pw;
    if (!(y1 >= -1.0f) || !(y0 <= (float)H) || !(x1 >= -1.0f) || !(x0 <= (float)W)) return false;
    const int ylo = min(max((int)floorf(y0), 0), H - 1), yhi = min(max((int)floorf(y1) + 1, 0), H - 1);
    const int xlo = min(max((int)floorf(x0), 0), W - 1), xhi = min(max((int)floorf(x1) + 1, 0), W - 1);
    ty0 = ylo / RG_TS; ty1 = yhi / RG_TS; tx0 = xlo / RG_TS; tx1 = xhi / RG_TS;
    return true;
}

// 64 threads per RoI: thread q takes tiles q, q + 64, ... of the RoI's rectangle, so the atomics of a RoI (up to ~40 tiles) are in
// flight together instead of one round trip after the other (one thread per RoI: 39 us for the fill pass of 1280 RoIs)
template <bool FILL>
__global__ __launch_bounds__(256) void roi_gather_bin(RGLevels Lv, RGSets S, int sr, int aligned, int* __restrict__ count,
                                                      const int* __restrict__ offs, int* __restrict__ cursor, int* __restrict__ list) {
    const int id = blockIdx.x * 4 + (threadIdx.x >> 6), q0 = threadIdx.x & 63;
    if (id >= S.Ktot) return;
    const int s = (S.n > 1 && id >= S.first[1]) ? ((S.n > 2 && id >= S.first[2]) ? ((S.n > 3 && id >= S.first[3]) ? 3 : 2) : 1) : 0;
    const int first_s = s == 0 ? 0 : (s == 1 ? S.first[1] : (s == 2 ? S.first[2] : S.first[3]));
    const int k = id - first_s;
    // selects instead of S.x[s] with a run-time s: indexing the kernel-argument struct dynamically faulted in roi_gather_main
    // (hipcc 7.2, gfx950: 'Memory access fault' as soon as a second set existed; tools/rg_stage.py) -- same rule here
    const int* lvl_s = s == 0 ? S.lvl[0] : (s == 1 ? S.lvl[1] : (s == 2 ? S.lvl[2] : S.lvl[3]));
    const float* rois_s = s == 0 ? S.rois[0] : (s == 1 ? S.rois[1] : (s == 2 ? S.rois[2] : S.rois[3]));
    const int ph = s == 0 ? S.ph[0] : (s == 1 ? S.ph[1] : (s == 2 ? S.ph[2] : S.ph[3]));
    const int pw = s == 0 ? S.pw[0] : (s == 1 ? S.pw[1] : (s == 2 ? S.pw[2] : S.pw[3]));
    const int l = lvl_s[k];
    if (l < 0 || l >= Lv.n_levels) return;
    int n, ty0, ty1, tx0, tx1;
    if (!rg_rect(Lv, rois_s + 5 * (int64_t)k, l, ph, pw, sr, aligned, n, ty0, ty1, tx0, tx1)) return;
    const int nx = tx1 - tx0 + 1, nt = (ty1 - ty0 + 1) * nx;
    const int ltx = l == 0 ? Lv.tx[0] : (l == 1 ? Lv.tx[1] : (l == 2 ? Lv.tx[2] : Lv.tx[3]));
    const int lty = l == 0 ? Lv.ty[0] : (l == 1 ? Lv.ty[1] : (l == 2 ? Lv.ty[2] : Lv.ty[3]));
    const int lbase = l == 0 ? Lv.base[0] : (l == 1 ? Lv.base[1] : (l == 2 ? Lv.base[2] : Lv.base[3]));
    for (int q = q0; q < nt; q += 64) {
        const int ty = ty0 + q / nx, tx = tx0 + q % nx;
        const int t = lbase + (n * lty + ty) * ltx + tx;
        if (FILL) list[offs[t] + atomicAdd(cursor + t, 1)] = (s << 24) | k;
        else atomicAdd(count + t, 1);
    }
}

// offs[0 .. total] = exclusive scan of count; count and cursor are left zero
__global__ __launch_bounds__(1024) void roi_gather_scan(int* __restrict__ count, int* __restrict__ offs, int* __restrict__ cursor, int total) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (total + 1023) / 1024;
    const int b = t * per, e = min(b + per, total);
    int s = 0;
    for (int i = b; i < e; ++i) s += count[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = b; i < e; ++i) { const int c = count[i]; offs[i] = run; run += c; count[i] = 0; cursor[i] = 0; }
    if (t == 1023) offs[total] = part[1023];
}

// block = (tile, pair of tile rows, 256-channel chunk); its four WAVES take different entries of the tile's list (entry i goes to
// wave i % 4), each with wave-private weight tables and no block barrier inside the list walk, and fold their accumulators through
// LDS at the end.  A thread owns 4 consecutive channels of both rows and all 8 columns (64 accumulators).  (First version: the four
// waves were the four row pairs and walked the whole list together behind one barrier per entry -- 182 us on spread-out RoIs but
// 573 us inside the training step, where the sampled RoIs pile up on the ground-truth boxes: the longest list sets the time.)
template <typename TG, typename TO>
__global__ __launch_bounds__(256) void roi_gather_main(RGLevels Lv, RGSets S, int C, int sr, int aligned, const int* __restrict__ offs,
                                                       const int* __restrict__ list, TO* o0, TO* o1, TO* o2, TO* o3) {
    __shared__ __attribute__((aligned(16))) char pool[16384];       // list walk: raw | ent | RoIs | meta;  afterwards: the fold buffer
    __shared__ float wtab[4][160];                                  // per wave: Wy[2][16] | Wx[8][16] of its current entry
    int* raw = (int*)pool;                                          // [RG_MAXL]
    int* ent = raw + RG_MAXL;                                       // [RG_MAXL]
    float (*roi_lds)[5] = (float (*)[5])(pool + 8192);              // [256][5]
    int (*meta_lds)[2] = (int (*)[2])(pool + 8192 + 5120);          // [256][2]
    const int chunks = (C + RG_CH - 1) / RG_CH;
    int bid = blockIdx.x;
    const int chunk = bid % chunks; bid /= chunks;
    const int rp = bid & 3, tile = bid >> 2;
    int l = 0;
    while (l + 1 < Lv.n_levels && tile >= Lv.base[l + 1]) ++l;
    const int H = Lv.H[l], W = Lv.W[l];
    const float lscale = Lv.scale[l];
    int r = tile - Lv.base[l];
    const int tx = r % Lv.tx[l]; r /= Lv.tx[l];
    const int ty = r % Lv.ty[l];
    const int n = r / Lv.ty[l];
    const int y0 = ty * RG_TS + 2 * rp, x0 = tx * RG_TS;            // this block's two rows
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, c = chunk * RG_CH + 4 * lane;      // C % 4 == 0 (host-checked)
    if (y0 >= H) return;                                            // (block-uniform: the tile's lower rows are outside the map)
    const int b = offs[tile], cnt = offs[tile + 1] - b;
    const bool sorted = cnt <= RG_MAXL;
    if (sorted) {                                                   // rank sort in LDS: the same summation order in every run
        for (int i = t; i < cnt; i += 256) raw[i] = list[b + i];
        __syncthreads();
        for (int i = t; i < cnt; i += 256) {
            const int v = raw[i];
            int rank = 0;
            for (int q = 0; q < cnt; ++q) rank += raw[q] < v ? 1 : 0;
            ent[rank] = v;                                         // (set, roi) pairs are distinct within a tile's list
        }
    }
    f32x4 acc[2][RG_TS];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int x = 0; x < RG_TS; ++x) acc[a][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    volatile float* wt = wtab[wv];
    for (int e0 = 0; e0 < cnt; e0 += 256) {
        const int nb = min(256, cnt - e0);
        __syncthreads();                                            // (ent is complete; the previous batch's RoIs are no longer read)
        if (t < nb) {                                               // the batch's RoIs into LDS together: one memory round trip
            const int v = sorted ? ent[e0 + t] : list[b + e0 + t];
            const int s = v >> 24, k = v & 0xffffff;
            // (selects, not S.x[s]: indexing the kernel-argument struct with a run-time index faulted, see roi_gather_bin)
            const float* rois_s = s == 0 ? S.rois[0] : (s == 1 ? S.rois[1] : (s == 2 ? S.rois[2] : S.rois[3]));
            const int ph = s == 0 ? S.ph[0] : (s == 1 ? S.ph[1] : (s == 2 ? S.ph[2] : S.ph[3]));
            const int pw = s == 0 ? S.pw[0] : (s == 1 ? S.pw[1] : (s == 2 ? S.pw[2] : S.pw[3]));
            const float* rpn = rois_s + 5 * (int64_t)k;
#pragma unroll
            for (int q = 0; q < 5; ++q) roi_lds[t][q] = rpn[q];
            meta_lds[t][0] = ph | (pw << 8) | (s << 16);
            meta_lds[t][1] = k;
        }
        __syncthreads();
        for (int i = wv; i < nb; i += 4) {                          // this wave's entries; no block barrier in here
            const int m0 = meta_lds[i][0], k = meta_lds[i][1];
            const int ph = m0 & 0xff, pw = (m0 >> 8) & 0xff, s = __builtin_amdgcn_readfirstlane(m0 >> 16);
            const void* gout_s = s == 0 ? S.gout[0] : (s == 1 ? S.gout[1] : (s == 2 ? S.gout[2] : S.gout[3]));
            float rv[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) rv[q] = roi_lds[i][q];
            const RoiGeom g = roi_geom(rv, lscale, aligned, ph, pw, sr);
            // weight tables: entry idx of [Wy row 0 | Wy row 1 | Wx col 0..7] x 16 bins; lane handles idx = lane, lane + 64, lane + 128
            unsigned iu = 0, jm = 0;
#pragma unroll
            for (int pass = 0; pass < 3; ++pass) {
                const int idx = lane + 64 * pass;
                const bool live = idx < 160;
                const bool isx = idx >= 32;
                const int q = isx ? idx - 32 : idx;
                const int row = q >> 4, bin = q & 15;
                const int np = isx ? pw : ph, grid = isx ? g.grid_w : g.grid_h, size = isx ? W : H, pix = (isx ? x0 : y0) + row;
                const float start = isx ? g.start_w : g.start_h, bsz = isx ? g.bin_w : g.bin_h;
                float wsum = 0.f;
                if (live && bin < np && pix < size) {
                    for (int s2 = 0; s2 < grid; ++s2) {
                        int lo, hi; float lw, hw;
                        if (!axis_sample(start + (float)bin * bsz + ((float)s2 + .5f) * bsz / (float)grid, size, lo, hi, lw, hw)) continue;
                        if (lo == pix) wsum += hw;
                        if (hi == pix) wsum += lw;
                    }
                }
                if (live) wt[idx] = wsum;
                const unsigned long long nz = __ballot(wsum != 0.f);
                const unsigned fold = (unsigned)((nz | (nz >> 16) | (nz >> 32) | (nz >> 48)) & 0xffffu);
                if (pass == 0) {                                    // lanes 0-31: Wy rows 0, 1;  lanes 32-63: Wx columns 0, 1
                    iu = (unsigned)((nz | (nz >> 16)) & 0xffffu);
                    jm = (unsigned)(((nz >> 32) | (nz >> 48)) & 0xffffu);
                } else jm |= fold;                                  // Wx columns 2-5, 6-7
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (iu != 0 && jm != 0 && c < C) {
                const TG* gob = (const TG*)gout_s + (int64_t)k * ph * pw * C + c;
                const float inv = 1.0f / g.count;
                unsigned jr = jm;
                while (jr) {
                    const int j = __builtin_ctz(jr); jr &= jr - 1;
                    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};
                    unsigned im = iu;
                    while (im) {
                        const int i2 = __builtin_ctz(im); im &= im - 1;
                        const f32x4 gv = rg_load4(gob + ((int64_t)i2 * pw + j) * C);
                        r0 += wt[i2] * gv; r1 += wt[16 + i2] * gv;
                    }
                    r0 *= inv; r1 *= inv;
#pragma unroll
                    for (int x = 0; x < RG_TS; ++x) { const float wx = wt[32 + 16 * x + j]; acc[0][x] += wx * r0; acc[1][x] += wx * r1; }
                }
            }
            __builtin_amdgcn_wave_barrier();                        // the table is rewritten for the wave's next entry
        }
    }
    // ---- fold the four waves: round q handles columns 2q, 2q + 1 -- every wave leaves its partial sums in LDS, wave q adds and stores
    TO* out = l == 0 ? o0 : (l == 1 ? o1 : (l == 2 ? o2 : o3));
    f32x4* fold = (f32x4*)pool;                                     // [4 waves][4 values][64 lanes] = 16 KB
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        __syncthreads();                                            // (first round: everybody is done with raw / ent / RoIs)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int xx = 0; xx < 2; ++xx) fold[(wv * 4 + a * 2 + xx) * 64 + lane] = acc[a][2 * q + xx];
        __syncthreads();
        if (wv == q && c < C) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int y = y0 + a;
                if (y >= H) continue;
#pragma unroll
                for (int xx = 0; xx < 2; ++xx) {
                    const int x = x0 + 2 * q + xx;
                    if (x >= W) continue;
                    f32x4 v = fold[(0 * 4 + a * 2 + xx) * 64 + lane];
#pragma unroll
                    for (int w2 = 1; w2 < 4; ++w2) v += fold[(w2 * 4 + a * 2 + xx) * 64 + lane];
                    rg_store4(out + (((int64_t)n * H + y) * W + x) * C + c, v);
                }
            }
        }
    }
}

static int rg_levels(RGLevels& Lv, const int* Hs, const int* Ws, const float* scales, int n_levels, int N) {
    int total = 0;
    for (int l = 0; l < 4; ++l) {
        const int m = l < n_levels ? l : n_levels - 1;
        Lv.H[l] = Hs[m]; Lv.W[l] = Ws[m]; Lv.scale[l] = scales[m];
        if (Lv.H[l] <= 0 || Lv.W[l] <= 0) return -1;
        Lv.ty[l] = (Lv.H[l] + RG_TS - 1) / RG_TS; Lv.tx[l] = (Lv.W[l] + RG_TS - 1) / RG_TS;
        Lv.base[l] = total;
        if (l < n_levels) total += N * Lv.ty[l] * Lv.tx[l];
    }
    Lv.n_levels = n_levels; Lv.N = N; Lv.total = total;
    return total;
}

// bytes of the persistent workspace of roi_align_multilevel_bwd_gather: [count | cursor | offsets (+1) | list].  The list is sized
// for the worst case (every RoI covering its level's whole map), so it cannot overflow.  Zero the workspace once; every call
// leaves the counts zero again.
extern "C" int64_t roi_align_gather_workspace_bytes(const int* Hs, const int* Ws, int n_levels, int N, int K_total) {
    if (!Hs || !Ws || n_levels <= 0 || n_levels > 4 || N <= 0 || K_total < 0) return 0;
    RGLevels Lv;
    const float one[4] = {1.f, 1.f, 1.f, 1.f};
    const int total = rg_levels(Lv, Hs, Ws, one, n_levels, N);
    if (total <= 0) return 0;
    int per_roi = 0;
    for (int l = 0; l < n_levels; ++l) per_roi = per_roi > Lv.ty[l] * Lv.tx[l] ? per_roi : Lv.ty[l] * Lv.tx[l];
    return ((int64_t)3 * total + 16 + (int64_t)K_total * per_roi) * (int64_t)sizeof(int);
}

// grads[l]: (N, H[l], W[l], C) in out_dtype (SWIN_BF16 or SWIN_F32), FULLY WRITTEN (zeros where no RoI reaches) -- no zero fill by
// the caller, no cast afterwards.  n_sets RoI sets pooled from the same pyramid: gouts[s] (K[s], ph[s], pw[s], C) channels-last in
// grad_dtype, rois[s] (K[s], 5), lvls[s] (K[s]) (< 0: skipped).  ph, pw <= 16; K[s] < 2^24.
extern "C" int roi_align_multilevel_bwd_gather(void* const* grads, const int* Hs, const int* Ws, const float* scales, int n_levels, int N,
                                               int n_sets, const void* const* gouts, const float* const* rois, const int* const* lvls,
                                               const int* Ks, const int* phs, const int* pws, int C, int sampling_ratio, int aligned,
                                               int grad_dtype, int out_dtype, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!grads || !Hs || !Ws || !scales || n_levels <= 0 || n_levels > 4 || N <= 0 || n_sets <= 0 || n_sets > 4 || !gouts || !rois || !lvls ||
        !Ks || !phs || !pws || C <= 0 || !workspace)
        return SWIN_ERR_BAD_ARG;
    RGLevels Lv;
    const int total = rg_levels(Lv, Hs, Ws, scales, n_levels, N);
    if (total <= 0) return SWIN_ERR_BAD_ARG;
    RGSets S;
    int ktot = 0;
    for (int s = 0; s < 4; ++s) {
        const int m = s < n_sets ? s : n_sets - 1;
        S.gout[s] = gouts[m]; S.rois[s] = rois[m]; S.lvl[s] = lvls[m]; S.K[s] = s < n_sets ? Ks[m] : 0; S.ph[s] = phs[m]; S.pw[s] = pws[m];
        S.first[s] = ktot;
        if (s < n_sets) {
            if (Ks[m] < 0 || Ks[m] >= (1 << 24) || phs[m] <= 0 || pws[m] <= 0) return SWIN_ERR_BAD_ARG;
            if (phs[m] > 16 || pws[m] > 16 || C % 4 != 0) return SWIN_ERR_UNSUPPORTED;
            if (Ks[m] > 0 && (!gouts[m] || !rois[m] || !lvls[m])) return SWIN_ERR_BAD_ARG;
            ktot += Ks[m];
        }
    }
    S.n = n_sets; S.Ktot = ktot;
    if (workspace_bytes < roi_align_gather_workspace_bytes(Hs, Ws, n_levels, N, ktot)) return SWIN_ERR_BAD_ARG;
    for (int l = 0; l < n_levels; ++l) if (!grads[l]) return SWIN_ERR_BAD_ARG;
    int* count = (int*)workspace;
    int* cursor = count + total;
    int* offs = cursor + total;
    int* list = offs + total + 16;
    hipStream_t st = (hipStream_t)stream;
    const int stage = swin_dev_int("SWIN_RG_STAGE", 4);            // development builds: stop after stage n (fault localisation)
    if (ktot > 0) roi_gather_bin<false><<<(ktot + 3) / 4, 256, 0, st>>>(Lv, S, sampling_ratio, aligned, count, offs, cursor, list);
    if (stage < 2) return swin_launch_status();
    roi_gather_scan<<<1, 1024, 0, st>>>(count, offs, cursor, total);
    if (stage < 3) return swin_launch_status();
    if (ktot > 0) roi_gather_bin<true><<<(ktot + 3) / 4, 256, 0, st>>>(Lv, S, sampling_ratio, aligned, count, offs, cursor, list);
    if (stage < 4) return swin_launch_status();
    const int chunks = (C + RG_CH - 1) / RG_CH;
    void* o[4];
    for (int l = 0; l < 4; ++l) o[l] = grads[l < n_levels ? l : n_levels - 1];
#define RG_LAUNCH(TG, TO) roi_gather_main<TG, TO><<<total * 4 * chunks, 256, 0, st>>>(Lv, S, C, sampling_ratio, aligned, offs, list, (TO*)o[0], (TO*)o[1], (TO*)o[2], (TO*)o[3])
    if (grad_dtype == SWIN_BF16 && out_dtype == SWIN_BF16) RG_LAUNCH(bf16, bf16);
    else if (grad_dtype == SWIN_F32 && out_dtype == SWIN_F32) RG_LAUNCH(float, float);
    else if (grad_dtype == SWIN_BF16 && out_dtype == SWIN_F32) RG_LAUNCH(bf16, float);
    else if (grad_dtype == SWIN_F32 && out_dtype == SWIN_BF16) RG_LAUNCH(float, bf16);
    else return SWIN_ERR_UNSUPPORTED;
#undef RG_LAUNCH
    return swin_launch_status();
}
