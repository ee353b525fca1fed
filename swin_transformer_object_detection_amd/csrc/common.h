// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/swin_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
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
