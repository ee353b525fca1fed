// bf16 MFMA GEMM core for gfx950 with two A-operand loaders:
//   * implicit-GEMM 3x3 / pad 1 / stride 1 convolution over a channels-last (N,H,W,Cin) activation
//     (FPN output convs fpn.py:195-197, RPN conv rpn_head.py:43, FCN mask head convs fcn_mask_head.py:119-121)
//   * plain row-major A (token-major Linear layers).
// C[m][n] = sum_k A[m][k] * Wt[n][k] (+ bias[n]) (ReLU optional);  Wt is K-contiguous: for the conv it
// is the (Cout, ky, kx, Cin) weight, i.e. the channels_last memory of the (Cout,Cin,3,3) parameter.
//
// Tile 128(m) x 128(n) x 64(k), 256 threads = 2x2 waves of 64x64, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// LDS: two K-tiles (A 16 KB + W 16 KB each), 16-byte pieces XOR-swizzled by (row & 7) so the
// ds_read_b128 fragment reads are bank-conflict free; global->register->LDS staging with the next tile's
// loads issued before the MFMAs of the current one (one barrier per K-tile).
// The weight tile is the MFMA "A" operand and the pixel tile the "B" operand, so a lane owns one pixel
// and 4 consecutive output channels per accumulator group -> 8-byte stores into the NHWC output row.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64

struct ConvGeom { int N, H, W, Cin; };

// ---- A-operand loaders: 16-byte piece `piece` (0..7) of row `m` for K-tile `kt` --------------------
struct PlainA {
    const bf16* a; int64_t M; int K;
    __device__ __forceinline__ void prep(int64_t m, int64_t& base, int& y, int& x) const { base = m < M ? m * K : -1; y = x = 0; }
    __device__ __forceinline__ uint4 load(int64_t base, int y, int x, int kt, int piece) const {
        uint4 z = {0, 0, 0, 0};
        if (base < 0) return z;
        return *(const uint4*)(a + base + kt * BK + piece * 8);
    }
};

struct ConvA {
    const bf16* a; int64_t M; ConvGeom g; int cpt;   // cpt = Cin / BK (K-tiles per filter tap)
    __device__ __forceinline__ void prep(int64_t m, int64_t& base, int& y, int& x) const {
        if (m >= M) { base = -1; y = x = 0; return; }
        x = (int)(m % g.W); int64_t t = m / g.W; y = (int)(t % g.H);
        base = m * g.Cin;                              // pixel (n,y,x) itself
    }
    __device__ __forceinline__ uint4 load(int64_t base, int y, int x, int kt, int piece) const {
        uint4 z = {0, 0, 0, 0};
        if (base < 0) return z;
        int tap = kt / cpt, c0 = (kt - tap * cpt) * BK;
        int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= g.H || xx < 0 || xx >= g.W) return z;
        return *(const uint4*)(a + base + ((int64_t)dy * g.W + dx) * g.Cin + c0 + piece * 8);
    }
};

__device__ __forceinline__ int swz(int row, int piece) { return piece ^ (row & 7); }

template <typename ALoader, bool RELU>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(ALoader A, const bf16* __restrict__ Wt, const float* __restrict__ bias,
                                                           bf16* __restrict__ C, int64_t M, int Nn, int K, int mtiles, int ntiles) {
    __shared__ __attribute__((aligned(16))) uint4 lds[2][2][BM * 8];     // [buf][A|W][row*8 + piece]
    // XCD-aware tile order: blocks that share an XCD (id % 8) get a contiguous run of tiles, n fastest
    const int nblk = mtiles * ntiles;
    int id = blockIdx.x;
    {
        int q = nblk / 8, r = nblk % 8, xcd = id % 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + id / 8;
    }
    const int mt_ = id / ntiles, nt_ = id - mt_ * ntiles;
    const int64_t m0 = (int64_t)mt_ * BM;
    const int n0 = nt_ * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int c = lane & 31, h = lane >> 5;

    // staging role: 4 rows (tid/8 + 32 i), piece tid%8, for both operands
    const int srow = tid >> 3, spiece = tid & 7;
    int64_t abase[4]; int ay[4], ax[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) A.prep(m0 + srow + 32 * i, abase[i], ay[i], ax[i]);
    const bf16* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int n = n0 + srow + 32 * i;
        wrow[i] = n < Nn ? Wt + (int64_t)n * K + spiece * 8 : nullptr;
    }
    const int nk = K / BK;
    uint4 ra[4], rw[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = A.load(abase[i], ay[i], ax[i], kt, spiece);
            uint4 z = {0, 0, 0, 0};
            rw[i] = wrow[i] ? *(const uint4*)(wrow[i] + kt * BK) : z;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row = srow + 32 * i;
            lds[buf][0][row * 8 + swz(row, spiece)] = ra[i];
            lds[buf][1][row * 8 + swz(row, spiece)] = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x16{0};

    gload(0);
    lstore(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
        const uint4* As = lds[cur][0];
        const uint4* Ws = lds[cur][1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 wf[2], af[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int rw_ = wn * 64 + 32 * t + c, ra_ = wm * 64 + 32 * t + c;
                uint4 u = Ws[rw_ * 8 + swz(rw_, 2 * s + h)];
                uint4 v = As[ra_ * 8 + swz(ra_, 2 * s + h)];
                wf[t] = *(bf16x8*)&u;
                af[t] = *(bf16x8*)&v;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // epilogue: lane = pixel m, registers = output channels
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int64_t m = m0 + wm * 64 + 32 * mt + c;
        if (m >= M) continue;
        bf16* crow = C + m * Nn;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                int n = n0 + wn * 64 + 32 * nt + 8 * gq + 4 * h;
                if (n >= Nn) continue;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[nt][mt][4 * gq + e] + (bias ? bias[n + e] : 0.f);
                    if (RELU) v = fmaxf(v, 0.f);
                    o[e] = (bf16)v;
                }
                *(bf16x4*)(crow + n) = o;
            }
    }
}

template <typename ALoader>
static int gemm_launch(ALoader A, const bf16* Wt, const float* bias, bf16* C, int64_t M, int Nn, int K, int relu, hipStream_t s) {
    int mtiles = (int)((M + BM - 1) / BM), ntiles = (Nn + BN - 1) / BN;
    int blocks = mtiles * ntiles;
    if (relu) gemm_bf16_kernel<ALoader, true><<<blocks, 256, 0, s>>>(A, Wt, bias, C, M, Nn, K, mtiles, ntiles);
    else gemm_bf16_kernel<ALoader, false><<<blocks, 256, 0, s>>>(A, Wt, bias, C, M, Nn, K, mtiles, ntiles);
    return swin_launch_status();
}

// x (N,H,W,Cin) bf16 channels-last; w (Cout,3,3,Cin) bf16; bias (Cout) f32 or NULL; y (N,H,W,Cout) bf16.
extern "C" int conv3x3_nhwc_bf16(const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int Cin,
                                 int Cout, int relu, void* stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SWIN_ERR_BAD_ARG;
    if (Cin % BK != 0 || Cout % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    int64_t M = (int64_t)N * H * W;
    ConvA A{(const bf16*)x, M, ConvGeom{N, H, W, Cin}, Cin / BK};
    return gemm_launch(A, (const bf16*)w, bias, (bf16*)y, M, Cout, 9 * Cin, relu, (hipStream_t)stream);
}

// a (M,K) bf16 row-major; w (N,K) bf16 (nn.Linear weight layout); c (M,N) bf16 = a w^T + bias.
extern "C" int gemm_nt_bf16(const void* a, const void* w, const float* bias, void* c, int64_t M, int N, int K, int relu,
                            void* stream) {
    if (!a || !w || !c || M <= 0 || N <= 0 || K <= 0) return SWIN_ERR_BAD_ARG;
    if (K % BK != 0 || N % 4 != 0) return SWIN_ERR_UNSUPPORTED;
    PlainA A{(const bf16*)a, M, K};
    return gemm_launch(A, (const bf16*)w, bias, (bf16*)c, M, N, K, relu, (hipStream_t)stream);
}
