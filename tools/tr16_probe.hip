// Probe of ds_read_b64_tr_b16 lane mapping on gfx950 (development aid; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short t[64][40];
    for (int i = threadIdx.x; i < 64 * 40; i += 64) t[i / 40][i % 40] = (short)((i / 40) * 100 + (i % 40));
    __syncthreads();
    int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    int h = g >> 1, dh = g & 1;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)&t[4 * h + q][16 * dh + 4 * p]);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    k<<<1, 64>>>(d);
    short hbuf[256]; hipMemcpy(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, hbuf[l * 4], hbuf[l * 4 + 1], hbuf[l * 4 + 2], hbuf[l * 4 + 3]);
    return 0;
}
