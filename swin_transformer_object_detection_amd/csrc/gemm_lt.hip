// Plain bf16 GEMMs of the hot path (nn.Linear forward and data gradient: swin_transformer.py:33-36,129,151,296;
// fpn.py lateral 1x1 convs; the heads' fully connected layers) on hipBLASLt, called directly with cached plans.
// The host mirror used torch.nn.functional.linear / torch.mm for these; on this stack each such call costs 25-30 us of
// host time (dispatcher + descriptor set-up + heuristic look-up), 139 calls per training step -- with the kernels
// around them tuned, that alone kept the step host-bound.  Here descriptors, layouts and the selected algorithm are
// built once per (M, N, K, layout, bias) and a call is one hipblasLtMatmul.
#include <hipblaslt/hipblaslt.h>

#include <cstdlib>
#include <mutex>
#include <vector>
#include <unordered_map>

#include "common.h"

namespace {
struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulHeuristicResult_t algo;
    std::vector<hipblasLtMatmulHeuristicResult_t> cand;      // the heuristic's candidate list the algorithm was chosen from
    int chosen = 0;                                          // index of `algo` in `cand`
    bool ok = false;
};
struct Key {
    int64_t M; int N, K, layout, bias, dev;                // plans (and their tuned algorithm) belong to a device
    bool operator==(const Key& o) const {
        return M == o.M && N == o.N && K == o.K && layout == o.layout && bias == o.bias && dev == o.dev;
    }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        size_t h = (size_t)k.M * 1000003u;
        h ^= (size_t)k.N * 7919u + ((size_t)k.K << 20) + (size_t)k.layout * 31u + (size_t)k.bias + (size_t)k.dev * 131071u;
        return h;
    }
};
std::mutex g_mu;
hipblasLtHandle_t g_handles[16] = {};                    // one per device
thread_local hipblasLtHandle_t g_handle = nullptr;       // the current call's handle (set by get_plan)
std::unordered_map<Key, Plan, KeyHash> g_plans;
const size_t kWorkspace = 32u << 20;

// First use of a shape: time the library's top candidates on the caller's own buffers and keep the fastest (the
// heuristic's first choice is not always the best for these tall-skinny shapes).  One-off host synchronisation per
// shape, during warm-up; SWIN_GEMM_TUNE=0 keeps the heuristic's first choice.
static const int kMaxCandidates = 64;
static int candidates() {                                    // SWIN_GEMM_CANDIDATES: development sweep (default 12)
    static const int n = swin_dev_int("SWIN_GEMM_CANDIDATES", 12);
    return n < 1 ? 1 : (n > kMaxCandidates ? kMaxCandidates : n);
}

static float time_algo(Plan& p, const hipblasLtMatmulHeuristicResult_t& h, const void* a, const void* b, const void* bias,
                       void* c, void* workspace, hipStream_t s) {
    const float alpha = 1.f, beta = 0.f;
    if (h.workspaceSize > kWorkspace) return -1.f;
    if (bias) hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f;
    float best = -1.f;
    for (int rep = 0; rep < 3; ++rep) {                      // rep 0 = warm-up
        hipEventRecord(e0, s);
        hipblasStatus_t st = hipblasLtMatmul(g_handle, p.desc, &alpha, b, p.a, a, p.b, &beta, c, p.c, c, p.c, &h.algo, workspace,
                                             kWorkspace, s);
        hipEventRecord(e1, s);
        if (st != HIPBLAS_STATUS_SUCCESS || hipEventSynchronize(e1) != hipSuccess) { best = -1.f; break; }
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && (best < 0.f || ms < best)) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

Plan* get_plan(int64_t M, int N, int K, int layout, int bias, const void* a = nullptr, const void* b = nullptr,
               const void* bias_ptr = nullptr, void* c = nullptr, void* workspace = nullptr, hipStream_t stream = nullptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!g_handles[dev] && hipblasLtCreate(&g_handles[dev]) != HIPBLAS_STATUS_SUCCESS) return nullptr;
    g_handle = g_handles[dev];
    Key key{M, N, K, layout, bias, dev};
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second.ok ? &it->second : nullptr;
    Plan& p = g_plans[key];
    // row-major C(M,N) = A(M,K) op(B)  ==  column-major C^T(N,M) = op'(B) A^T
    if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return nullptr;
    const int32_t ta = layout == 0 ? HIPBLAS_OP_T : HIPBLAS_OP_N, tb = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
    if (bias) {
        const uint32_t ep = HIPBLASLT_EPILOGUE_BIAS;
        const int32_t bt = SWIN_HIP_R_16;
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
    }
    bool good = true;
    if (layout == 0)   // B given as (N,K) row-major = (K,N) column-major, transposed by the op
        good &= hipblasLtMatrixLayoutCreate(&p.a, SWIN_HIP_R_16, (uint64_t)K, (uint64_t)N, K) == HIPBLAS_STATUS_SUCCESS;
    else               // B given as (K,N) row-major = (N,K) column-major
        good &= hipblasLtMatrixLayoutCreate(&p.a, SWIN_HIP_R_16, (uint64_t)N, (uint64_t)K, N) == HIPBLAS_STATUS_SUCCESS;
    good &= hipblasLtMatrixLayoutCreate(&p.b, SWIN_HIP_R_16, (uint64_t)K, (uint64_t)M, K) == HIPBLAS_STATUS_SUCCESS;
    good &= hipblasLtMatrixLayoutCreate(&p.c, SWIN_HIP_R_16, (uint64_t)N, (uint64_t)M, N) == HIPBLAS_STATUS_SUCCESS;
    if (!good) return nullptr;
    hipblasLtMatmulPreference_t pref;
    if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return nullptr;
    uint64_t ws = kWorkspace;
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
    int found = 0;
    // the bias pointer must be set for the heuristic of a bias epilogue on some versions; any non-null value does
    if (bias) {
        const void* dummy = bias_ptr ? bias_ptr : (const void*)0x1000;
        hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &dummy, sizeof(dummy));
    }
    hipblasLtMatmulHeuristicResult_t cand[kMaxCandidates];
    static const bool tune = swin_dev_int("SWIN_GEMM_TUNE", 1) != 0;
    const int want = (tune && a && b && c && workspace) ? candidates() : 1;
    hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.a, p.b, p.c, p.c, pref, want, cand, &found);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found < 1) return nullptr;
    int best = 0;
    if (found > 1) {
        float best_ms = -1.f;
        for (int i = 0; i < found; ++i) {
            const float ms = time_algo(p, cand[i], a, b, bias_ptr, c, workspace, stream);
            if (ms >= 0.f && (best_ms < 0.f || ms < best_ms)) { best_ms = ms; best = i; }
        }
    }
    p.algo = cand[best];
    p.cand.assign(cand, cand + found);
    p.chosen = best;
    p.ok = true;
    return &p;
}
}  // namespace

extern "C" int64_t swin_gemm_workspace_bytes(void) { return (int64_t)kWorkspace; }

// c (M,N) bf16 row-major = a (M,K) bf16 row-major x op(b) [+ bias (N) bf16], fp32 accumulation.
//   b_layout 0: b is (N,K) row-major -- an nn.Linear weight, c = a b^T  (forward)
//   b_layout 1: b is (K,N) row-major -- c = a b                        (data gradient: dx = dy w)
// workspace: swin_gemm_workspace_bytes() bytes of device scratch.  Not thread-safe across streams sharing `workspace`.
extern "C" int swin_gemm_bf16(const void* a, const void* b, const void* bias, void* c, int64_t M, int N, int K, int b_layout,
                              void* workspace, void* stream) {
    if (M == 0) return SWIN_OK;
    if (!a || !b || !c || !workspace || M < 0 || N <= 0 || K <= 0 || (b_layout != 0 && b_layout != 1)) return SWIN_ERR_BAD_ARG;
    Plan* p = get_plan(M, N, K, b_layout, bias != nullptr, a, b, bias, c, workspace, (hipStream_t)stream);
    if (!p) return SWIN_ERR_UNSUPPORTED;
    const float alpha = 1.f, beta = 0.f;
    hipblasStatus_t st;
    {
        // the bias pointer lives in the (shared) descriptor: set + launch under the lock
        std::lock_guard<std::mutex> lk(g_mu);
        if (bias) hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
        st = hipblasLtMatmul(g_handle, p->desc, &alpha, b, p->a, a, p->b, &beta, c, p->c, c, p->c, &p->algo.algo, workspace,
                             kWorkspace, (hipStream_t)stream);
    }
    return st == HIPBLAS_STATUS_SUCCESS ? SWIN_OK : SWIN_ERR_LAUNCH;
}

// The algorithm of a plan is picked by timing on first use, so two processes can pick differently for the same shape.  Data-parallel
// ranks exchange rank 0's choices after warm-up so that every rank runs the same GEMM kernels:
//   swin_gemm_plans_export: up to `cap` records of 6 int64 {M, N, K, b_layout, has_bias, chosen candidate index} for the current
//     device's plans -> `out` (HOST memory); returns the number of plans (may exceed cap: call again with a larger buffer).
//   swin_gemm_plans_import: for every record whose plan exists on the current device and whose candidate list is long enough,
//     select that candidate; returns the number of plans changed.  Plans not built yet are left to their own first use.
extern "C" int swin_gemm_plans_export(int64_t* out, int cap) {
    std::lock_guard<std::mutex> lk(g_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    int n = 0;
    for (auto& kv : g_plans) {
        if (kv.first.dev != dev || !kv.second.ok) continue;
        if (out && n < cap) {
            int64_t* r = out + 6 * (int64_t)n;
            r[0] = kv.first.M; r[1] = kv.first.N; r[2] = kv.first.K; r[3] = kv.first.layout; r[4] = kv.first.bias; r[5] = kv.second.chosen;
        }
        ++n;
    }
    return n;
}

extern "C" int swin_gemm_plans_import(const int64_t* in, int n) {
    if (n < 0 || (n > 0 && !in)) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    int changed = 0;
    for (int i = 0; i < n; ++i) {
        const int64_t* r = in + 6 * (int64_t)i;
        auto it = g_plans.find(Key{r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], dev});
        if (it == g_plans.end() || !it->second.ok) continue;
        Plan& p = it->second;
        const int idx = (int)r[5];
        if (idx >= 0 && idx < (int)p.cand.size() && idx != p.chosen) { p.algo = p.cand[idx]; p.chosen = idx; ++changed; }
    }
    return changed;
}
