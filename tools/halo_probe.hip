// Ablation timings of the halo-staged conv kernel: compiled once per HALO_ABL value (tools/halo_probe.sh), each binary prints the
// time of the P2 convolution (2 x 200 x 320, 256 -> 256) for both tile widths.  Development tool; results are not checked.
#include <cstdio>
#include <vector>
#include "../swin_transformer_object_detection_amd/csrc/conv_halo.hip"

int main() {
    const int N = 2, H = 200, W = 320, Cin = 256, Cout = 256;
    const size_t nx = (size_t)N * H * W * Cin, nw = (size_t)Cout * 9 * Cin, ny = (size_t)N * H * W * Cout;
    bf16 *x, *w, *y; float* b;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2); hipMalloc(&b, Cout * 4);
    std::vector<unsigned short> hx(nx), hw(nw);
    unsigned s = 12345;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff)) ^ (unsigned short)((s >> 31) << 15); }
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3800 + ((s >> 16) & 0x3ff)) ^ (unsigned short)((s >> 31) << 15); }
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemset(b, 0, Cout * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("HALO_ABL=%d:", HALO_ABL);
    for (int nt : {2, 4}) {
        for (int i = 0; i < 3; ++i) swin_conv_halo(x, w, b, nullptr, y, N, H, W, Cin, Cout, 0, nt, 0);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) swin_conv_halo(x, w, b, nullptr, y, N, H, W, Cin, Cout, 0, nt, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("  nt=%d %7.1f us", nt, best * 100.f);
    }
    printf("\n");
    return 0;
}
