// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/swin_hip.h"

// The 16-bit storage / MFMA-operand type.  The library is built TWICE from these sources: libswin_hip.so with bfloat16 and
// libswin_hip_f16.so (-DSWIN_HALF) with IEEE half -- the reference's mixed precision is fp16 (apex O1, mmdet/apis/train.py:82-89;
// configs/swin/mask_rcnn_swin_small_*: use_fp16=True).  The two 16-bit MFMA forms have the same shape, rate and operand maps on
// gfx950, so nothing but the element type, the MFMA / transposed-read builtins and the library GEMM's type enum differs; the type
// keeps the name `bf16` in the sources ("the 16-bit type"), and SWIN_BF16 as a dtype code means "the library's 16-bit type".
#ifdef SWIN_HALF
typedef _Float16 bf16;
#define SWIN_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define SWIN_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define SWIN_HIP_R_16 HIP_R_16F
#else
typedef __bf16 bf16;
#define SWIN_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define SWIN_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define SWIN_HIP_R_16 HIP_R_16BF
#endif
typedef __attribute__((ext_vector_type(8))) bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) bf16 bf16x2;

// ds_read_b64_tr_b16 (guide T10) for the library's 16-bit type: p = this lane's 8-byte-aligned LDS address
typedef __attribute__((address_space(3))) bf16x4 swin_lds_h4;
__device__ __forceinline__ bf16x4 swin_ds_read_tr16(swin_lds_h4* p) {
#ifdef SWIN_HALF
    typedef __fp16 swin_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
    typedef __attribute__((address_space(3))) swin_fp16x4 swin_lds_fp16x4;
    const swin_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((swin_lds_fp16x4*)p);
    return __builtin_bit_cast(bf16x4, v);
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
#endif
}
#define SWIN_DS_READ_TR16(p) swin_ds_read_tr16((swin_lds_h4*)(p))
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define WAVE 64

// Development switches (launch-geometry sweeps, A/B toggles, ablations): the environment is read ONLY in a -DSWIN_DEV build
// (SWIN_DEV_BUILD=1 python -m swin_transformer_object_detection_amd.build; tools/microbench.py).  The shipped library takes
// the default and never looks at the environment, so a stray variable cannot change what a training process computes.
#ifdef SWIN_DEV
#include <cstdlib>
static inline int swin_dev_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline constexpr int swin_dev_int(const char*, int dflt) { return dflt; }
#endif

static inline int swin_launch_status() {
    return hipGetLastError() == hipSuccess ? SWIN_OK : SWIN_ERR_LAUNCH;
}

// csrc/abi.hip: the per-device auxiliary stream (null = none) and the event fork used to order work on it
void* swin_aux_stream(void);
void swin_aux_defer(bool on);                                 // block runner: collect the reductions, flush once
int swin_aux_flush(void* main, void* side);

// csrc/conv_halo.hip: the halo-staged 3x3 convolution (nt = 0: choose the tile width); SWIN_ERR_UNSUPPORTED when the shape does not fit
int swin_conv_halo(const bf16* x, const bf16* Wt, const float* bias, const bf16* gate, bf16* y, int N, int H, int W, int Cin, int Cout, int relu,
                   int nt, hipStream_t s);

// csrc/tail_reduce.hip: the small parameter-gradient reductions of a block backward collected into one table-driven launch
enum { SWIN_TAIL_COLSUM = 0, SWIN_TAIL_RELBIAS = 1 };
struct SwinTailProb {
    const float* src;      // COLSUM: partial rows [rows][cols];  RELBIAS: attention-backward slabs [rows][cols] (cols = slab stride)
    float* dst0;           // COLSUM: columns [0, a0) are added here;  RELBIAS: dtable (169, nH)
    float* dst1;           // COLSUM: columns [a0, cols) are added here;  RELBIAS: dbias_pad (3, C) or null
    int kind, rows, cols;
    int a0, a1;            // COLSUM: a0 = split column;  RELBIAS: a0 = nH, a1 = C
    int blk0;              // first thread block of this problem (filled in by the launcher)
};
void swin_tail_collect(bool on);                              // open / close (and drop) a collection on the current device
bool swin_tail_push(const SwinTailProb& p);                   // true: queued for swin_tail_flush; false: no collection open
int swin_tail_flush(void* stream);
int swin_tail_launch(const SwinTailProb* probs, int n, void* stream);
// swin_window_attn_bwd without its bias-gradient reduce: the slabs stay in `workspace` (n_slabs of slab_stride floats)
int swin_window_attn_bwd_slabs(const void* qkv, const float* qkv_bias, const float* bias_exp, const float* lse, const void* dout,
                               void* dqkv, float* dqkv_bias_pad, void* workspace, int B, int H, int W, int C, int nH, int shift,
                               float scale, void* stream, int* n_slabs, int* slab_stride);

template <typename T> struct Elt;
template <> struct Elt<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elt<bf16> {
    static __device__ __forceinline__ float ld(const bf16* p) { return (float)*p; }
    static __device__ __forceinline__ void st(bf16* p, float v) { *p = (bf16)v; }
};

// 16-byte vector of T (4 floats or 8 bf16) <-> floats
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    float v[4];
    __device__ __forceinline__ void load(const float* p) { *(float4*)v = *(const float4*)p; }
    __device__ __forceinline__ void store(float* p) const { *(float4*)p = *(const float4*)v; }
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ void load(const bf16* p) { v = *(const bf16x8*)p; }
    __device__ __forceinline__ void store(bf16* p) const { *(bf16x8*)p = v; }
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}
