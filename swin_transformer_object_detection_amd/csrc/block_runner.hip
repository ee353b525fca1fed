// One C call per Swin block and direction: the launch sequence of SwinTransformerBlock.forward
// (swin_transformer.py:204-255) and of its backward, issued from native code.  The kernels are the library's own entry
// points (include/swin_hip.h); what this removes is the host cost of issuing them one by one from Python (13 forward /
// 17 backward launches per block: ~150 / ~220 us of interpreter + ctypes + allocator time, against ~20 us here) --
// the training step had become bound by exactly that.  All buffers are provided by the caller (pointer tables below);
// nothing is allocated, nothing synchronises.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/swin_hip.h"
#include "common.h"

#define CHK(expr) do { int st_ = (expr); if (st_ != SWIN_OK) return st_; } while (0)

// Forward pointer table (all device pointers; bf16 activations / weights unless noted):
//  0 x            1 n1           2 dp0 (B f32|null)   3 dp1 (B f32|null)
//  4 wqkv (3C,C)  5 bqkv16|null  6 bqkv32 (3C f32)    7 table (169,nH f32)|null   8 bias_exp (nH,64,64 f32; out, or in when 7 is null)
//  9 wproj (C,C)  10 bproj16|null 11 n2w f32 12 n2b f32 13 w1 (4C,C) 14 b1 (4C f32) 15 w2 (C,4C) 16 b216|null
//  17 nnw f32|null 18 nnb f32|null
//  outputs / saved: 19 qkv (T,3C) 20 lse (f32) 21 o 22 y (tmp) 23 x1 24 n2 25 mean2 26 rstd2 27 hpre (T,4C) 28 h (T,4C)
//  29 y2 (tmp) 30 x2 31 nn|null 32 mean3|null 33 rstd3|null 34 gemm workspace 35 b2 (C f32; fused MLP only)
// ints: B, H, W, C, nH, shift, fused_mlp;  floats: scale, eps
// fused_mlp != 0 (C in {96,192}): fc1 -> GELU -> fc2 is ONE launch (swin_mlp_fwd_bf16, csrc/ts_mlp.hip); hpre / h (27, 28) are
// neither written nor saved -- the backward recomputes them.
extern "C" int swin_block_fwd(const void* const* p, const int64_t* iv, const float* fv, void* stream) {
    if (!p || !iv || !fv) return SWIN_ERR_BAD_ARG;
    const int B = (int)iv[0], H = (int)iv[1], W = (int)iv[2], C = (int)iv[3], nH = (int)iv[4], shift = (int)iv[5];
    const float scale = fv[0], eps = fv[1];
    const int64_t L = (int64_t)H * W, T = (int64_t)B * L;
    void* ws = const_cast<void*>(p[34]);
    // narrow stages: the token-stationary kernels of csrc/ts_linear.hip (qkv; proj + residual + norm2 in one launch)
    const bool ts = C == 96 || C == 128 || C == 192 || C == 256;
    const bool ts_gemm = ts || C == 384;          // C = 384: the GEMMs only (a token's 384-wide row does not fit two lanes' registers next to them)
    if (ts_gemm) CHK(swin_ts_linear_bf16(p[1], p[4], p[5], const_cast<void*>(p[19]), T, 3 * C, C, 0, stream));
    else CHK(swin_gemm_bf16(p[1], p[4], p[5], const_cast<void*>(p[19]), T, 3 * C, C, 0, ws, stream));
    if (p[7]) CHK(swin_rel_bias_expand((const float*)p[7], (float*)p[8], nH, stream));     // null: 8 already holds this step's expansion
    CHK(swin_window_attn_fwd(p[19], (const float*)p[6], (const float*)p[8], const_cast<void*>(p[21]), (float*)p[20], B, H, W, C, nH,
                             shift, scale, SWIN_BF16, stream));
    if (ts) {
        CHK(swin_ts_proj_add_ln_bf16(p[21], p[9], p[10], p[0], (const float*)p[2], L, (const float*)p[11], (const float*)p[12],
                                     const_cast<void*>(p[23]), const_cast<void*>(p[24]), (float*)p[25], (float*)p[26], T, C, eps, stream));
    } else {
        if (ts_gemm) CHK(swin_ts_linear_bf16(p[21], p[9], p[10], const_cast<void*>(p[22]), T, C, C, 0, stream));
        else CHK(swin_gemm_bf16(p[21], p[9], p[10], const_cast<void*>(p[22]), T, C, C, 0, ws, stream));
        CHK(swin_add_layernorm_fwd(p[0], p[22], (const float*)p[2], L, (const float*)p[11], (const float*)p[12], const_cast<void*>(p[23]),
                                   const_cast<void*>(p[24]), (float*)p[25], (float*)p[26], T, C, eps, SWIN_BF16, stream));
    }
    if (iv[6]) {
        // fused MLP with the second residual and the next norm in its epilogue: the block's last launch
        return swin_mlp_add_ln_fwd_bf16(p[24], p[13], (const float*)p[14], p[15], (const float*)p[35], p[23], (const float*)p[3], L,
                                        (const float*)p[17], (const float*)p[18], const_cast<void*>(p[30]), const_cast<void*>(p[31]),
                                        (float*)p[32], (float*)p[33], T, C, eps, stream);
    } else {
        // fc1 with the GELU in its epilogue (hand-written GEMM: hpre and h leave the kernel together) when the width allows
        if (C % 64 == 0) {
            CHK(swin_linear_gelu_hip_bf16(p[24], p[13], (const float*)p[14], const_cast<void*>(p[27]), const_cast<void*>(p[28]), T, 4 * C, C, stream));
        } else {
            CHK(swin_gemm_bf16(p[24], p[13], nullptr, const_cast<void*>(p[27]), T, 4 * C, C, 0, ws, stream));
            CHK(swin_bias_gelu_fwd(p[27], (const float*)p[14], const_cast<void*>(p[28]), T, 4 * C, SWIN_BF16, stream));
        }
        CHK(swin_gemm_bf16(p[28], p[15], p[16], const_cast<void*>(p[29]), T, C, 4 * C, 0, ws, stream));
    }
    if (p[17])
        CHK(swin_add_layernorm_fwd(p[23], p[29], (const float*)p[3], L, (const float*)p[17], (const float*)p[18],
                                   const_cast<void*>(p[30]), const_cast<void*>(p[31]), (float*)p[32], (float*)p[33], T, C, eps,
                                   SWIN_BF16, stream));
    else
        CHK(swin_add_layernorm_fwd(p[23], p[29], (const float*)p[3], L, nullptr, nullptr, const_cast<void*>(p[30]), nullptr, nullptr,
                                   nullptr, T, C, eps, SWIN_BF16, stream));
    return SWIN_OK;
}

// Backward pointer table:
//  saved:  0 n1  1 qkv  2 bias_exp  3 lse  4 o  5 x1  6 mean2  7 rstd2  8 n2  9 hpre  10 h  11 x2  12 mean3|null  13 rstd3|null
//          14 dp0|null  15 dp1|null  16 wqkv  17 wproj  18 w1  19 w2  20 n2w  21 nnw|null  22 b1 (f32)  23 qkv_bias (f32)
//  grads in: 24 dx2 (residual-stream gradient; with a next norm it may be null)   25 dnn (gradient of the next norm's output;
//          required with a next norm)
//  out:    26 dx   27 dn1
//  temporaries: 28 dx1  29 dy2 (INPUT when there is no next norm: dx2 scaled by DropPath, or dx2 itself)  30 dh (T,4C)
//          31 dhpre (T,4C)  32 dn2  33 dy  34 do  35 dqkv (T,3C)  36 unused (was dbexp)
//  fp32 gradient accumulators (null = not wanted): 37 dWqkv 38 dbqkv 39 dbqkv_pad 40 dWproj 41 dbproj 42 dW1 43 db1 44 dW2
//          45 db2 46 dn2w 47 dn2b 48 dnnw 49 dnnb 50 dtable
//  workspaces: 51 attention backward  52 LayerNorm backward (norm2)  53 LayerNorm backward (next norm)  54 gemm
//  56 w2t|null: the TRANSPOSED fc2 weight (4C, C) -- with it (and C % 64 == 0, no fused MLP) the fc2 data gradient and the GELU backward are
//     ONE launch (swin_linear_dgelu_hip_bf16), dh (30) is never written and db1 (43) comes from the dW1 launch
//  57 wproj^T|null: the TRANSPOSED proj weight (C, C) at the narrow widths -- the proj data gradient then runs on the token-stationary
//     kernel (swin_ts_linear_bf16)
//  55 weight-gradient stream (a hipStream_t, or null = `stream`): the four weight-gradient GEMMs are enqueued there, each
//     after an event recorded on `stream` behind the kernel that produced its dY operand; the caller joins the two
//     streams before anything reads the accumulators and keeps the operands alive until then.
// ints: B, H, W, C, nH, shift, fused_mlp, record_wgrads;  floats: scale
// fused_mlp != 0: 9 (hpre) and 10 (h) were not saved; swin_mlp_bwd_bf16 recomputes them from n2 (8), writes dn2 (32) and, as
// temporaries for the two weight-gradient GEMMs, h into 30 and dhpre into 31; db1 (43) then comes from the dW1 launch.
struct AuxScope {                 // the auxiliary stream is set (and its launches collected) for the duration of one block backward
    explicit AuxScope(void* side) : on(side != nullptr) { swin_set_aux_stream(side); if (on) swin_aux_defer(true); }
    ~AuxScope() { if (on) swin_aux_defer(false); swin_set_aux_stream(nullptr); }
    bool on;
};
struct TailScope {                // the block's small reductions are collected and flushed as one launch (csrc/tail_reduce.hip)
    TailScope() { swin_tail_collect(true); }
    ~TailScope() { swin_tail_collect(false); }
};

extern "C" int swin_block_bwd(const void* const* p, const int64_t* iv, const float* fv, void* stream) {
    if (!p || !iv || !fv) return SWIN_ERR_BAD_ARG;
    // Everything nothing on `stream` waits for -- the four weight-gradient GEMMs and the parameter-gradient reductions behind
    // LayerNorm / attention backward -- is enqueued on the second stream (entry 55) AFTER the block's data-gradient chain, behind
    // ONE event: a fork per launch was two runtime calls each, 16 of a block's 47, with the step host-bound.  On the GPU that
    // work then overlaps the NEXT block's data-gradient chain instead of this one's.
    void* const side = const_cast<void*>(p[55]);
    void* const wst = side ? side : stream;
    static const bool aux_on = swin_dev_int("SWIN_AUX_REDUCE", 1) != 0;      // development A/B (-DSWIN_DEV builds only)
    AuxScope aux(side && aux_on ? side : nullptr);
    TailScope tail;
    const int B = (int)iv[0], H = (int)iv[1], W = (int)iv[2], C = (int)iv[3], nH = (int)iv[4], shift = (int)iv[5];
    const float scale = fv[0];
    const int64_t L = (int64_t)H * W, T = (int64_t)B * L;
    void* gws = const_cast<void*>(p[54]);
    auto M = [&](int i) { return const_cast<void*>(p[i]); };
    const void* dx1 = p[28];
    const void* dy2 = p[29];
    // GELU backward in the epilogue of the fc2 data-gradient GEMM (entry 56 = the transposed fc2 weight): db1 then comes from the dW1
    // launch, as with the fused MLP
    const bool gelu_epi = !iv[6] && p[56] && C % 64 == 0;
    const bool db1_from_wgrad = iv[6] || gelu_epi;
    if ((p[45] && !p[44]) || (p[43] && db1_from_wgrad && !p[42]) || (p[41] && !p[40]) || (p[38] && !p[37])) return SWIN_ERR_UNSUPPORTED;
    // fused MLP and a next norm: the backward of that norm and of the second residual is the PROLOGUE of the fused MLP backward
    const bool mlp_half_fused = iv[6] && p[21];
    if (p[21]) {                              // second residual + next norm
        if (!p[25] || !p[48] || !p[49]) return SWIN_ERR_BAD_ARG;
        if (!mlp_half_fused)
            CHK(swin_layernorm_bwd(p[25], p[11], (const float*)p[21], (const float*)p[12], (const float*)p[13], p[24], M(28),
                                   p[15] ? M(29) : nullptr, (const float*)p[15], L, (float*)p[48], (float*)p[49], T, C, SWIN_BF16, M(53),
                                   stream));
        if (!p[15]) dy2 = p[28];
    } else {
        if (!p[24]) return SWIN_ERR_BAD_ARG;
        dx1 = p[24];
    }
    const void* hbuf;                          // the hidden activation h the fc2 weight gradient contracts with
    // dy = the gradient entering the attention branch: dx scaled by DropPath, or dx itself.  With a second stream it is ALWAYS the
    // private copy (33): dx (26) is handed to autograd, which adds a second gradient into it IN PLACE when x has two consumers
    // (the first block of a stage: x also feeds norm1) -- while the proj weight gradient would still be reading it on the other
    // stream (found by test_weight_gradient_stream_gives_the_same_gradients: 30 % error on those four proj.weight gradients).
    const bool record = iv[7] != 0;
    const bool own_dy = p[14] || side || record;        // (recorded weight gradients read dy long after dx was handed to autograd)
    if (!p[46] || !p[47]) return SWIN_ERR_BAD_ARG;
    if (mlp_half_fused) {
        // next-norm backward -> fused MLP backward -> norm2 backward: ONE launch (csrc/ts_mlp.hip); dx1 (28), dy2 (29, with DropPath) and the
        // two sets of [dgamma | dbeta] partial rows (52: norm2, 53: next norm) are its only outputs besides dx / dy / h / dhpre
        const int rows = (int)swin_mlp_ln_bwd_partial_rows(T, C);
        CHK(swin_mlp_ln2_bwd_bf16(p[8], p[18], (const float*)p[22], p[19], M(30), M(31), p[5], (const float*)p[6], (const float*)p[7],
                                  (const float*)p[20], (const float*)p[14], L, M(26), own_dy ? M(33) : nullptr, (float*)M(52), p[25], p[11],
                                  (const float*)p[12], (const float*)p[13], (const float*)p[21], p[24], (const float*)p[15], M(28), M(29),
                                  (float*)M(53), T, C, stream));
        swin_tail_push(SwinTailProb{(const float*)p[53], (float*)p[48], (float*)p[49], SWIN_TAIL_COLSUM, rows, 2 * C, C, 0, 0});
        swin_tail_push(SwinTailProb{(const float*)p[52], (float*)p[46], (float*)p[47], SWIN_TAIL_COLSUM, rows, 2 * C, C, 0, 0});
        hbuf = p[30];
    } else if (iv[6]) {
        // fused MLP backward with the backward of norm2 and of the first residual in its epilogue (csrc/ts_mlp.hip): dn2 (32) is never
        // stored; the block's [dgamma | dbeta] partial rows go to the tail launch
        CHK(swin_mlp_ln_bwd_bf16(p[8], dy2, p[18], (const float*)p[22], p[19], M(30), M(31), p[5], (const float*)p[6], (const float*)p[7],
                                 (const float*)p[20], dx1, (const float*)p[14], L, M(26), own_dy ? M(33) : nullptr, (float*)M(52), T, C, stream));
        swin_tail_push(SwinTailProb{(const float*)p[52], (float*)p[46], (float*)p[47], SWIN_TAIL_COLSUM, (int)swin_mlp_ln_bwd_partial_rows(T, C),
                                    2 * C, C, 0, 0});
        hbuf = p[30];
    } else {
        // fc2: dh = dy2 w2; GELU; fc1: dn2 = dhpre w1
        if (gelu_epi) {
            CHK(swin_linear_dgelu_hip_bf16(dy2, p[56], p[9], (const float*)p[22], M(31), T, 4 * C, C, stream));
        } else {
            CHK(swin_gemm_bf16(dy2, p[19], nullptr, M(30), T, 4 * C, C, 1, gws, stream));
            CHK(swin_bias_gelu_bwd(p[30], p[9], (const float*)p[22], M(31), (float*)p[43], T, 4 * C, SWIN_BF16, stream));
        }
        CHK(swin_gemm_bf16(p[31], p[18], nullptr, M(32), T, C, 4 * C, 1, gws, stream));
        hbuf = p[10];
    }
    // first residual + norm2 (already done in the fused MLP's epilogue)
    if (!iv[6])
        CHK(swin_layernorm_bwd(p[32], p[5], (const float*)p[20], (const float*)p[6], (const float*)p[7], dx1, M(26), own_dy ? M(33) : nullptr,
                               (const float*)p[14], L, (float*)p[46], (float*)p[47], T, C, SWIN_BF16, M(52), stream));
    const void* dy = own_dy ? p[33] : p[26];
    // proj
    if (p[57]) CHK(swin_ts_linear_bf16(dy, p[57], nullptr, M(34), T, C, C, 0, stream));       // do = dy Wproj as a Linear with Wproj^T (entry 57)
    else CHK(swin_gemm_bf16(dy, p[17], nullptr, M(34), T, C, C, 1, gws, stream));
    // window attention
    {   // the per-wave bias-gradient slabs go straight into the (169, nH) table in the block's tail launch
        int n_slabs = 0, slab_stride = 0;
        CHK(swin_window_attn_bwd_slabs(p[1], (const float*)p[23], (const float*)p[2], (const float*)p[3], p[34], M(35), (float*)p[39],
                                       M(51), B, H, W, C, nH, shift, scale, stream, &n_slabs, &slab_stride));
        if (p[50]) swin_tail_push(SwinTailProb{(const float*)p[51], (float*)p[50], (float*)p[39], SWIN_TAIL_RELBIAS, n_slabs, slab_stride,
                                               nH, C, 0});
    }
    // qkv
    CHK(swin_gemm_bf16(p[35], p[16], nullptr, M(27), T, C, 3 * C, 1, gws, stream));

    // ---- off the chain: one fork, then the collected reductions and the weight gradients (dW += dY^T X, db += colsum dY)
    if (aux.on) CHK(swin_aux_flush(stream, side));
    else if (side) CHK(swin_fork_stream(stream, side));
    CHK(swin_tail_flush(wst));
    if (record) {                 // iv[7]: recorded for the caller's next grouped launch (swin_wgrad_flush) instead of four launches here
        if (p[44]) CHK(swin_wgrad_record(dy2, hbuf, (float*)p[44], (float*)p[45], T, C, 4 * C));
        if (p[42]) CHK(swin_wgrad_record(p[31], p[8], (float*)p[42], db1_from_wgrad ? (float*)p[43] : nullptr, T, 4 * C, C));
        if (p[40]) CHK(swin_wgrad_record(dy, p[4], (float*)p[40], (float*)p[41], T, C, C));
        if (p[37]) CHK(swin_wgrad_record(p[35], p[0], (float*)p[37], (float*)p[38], T, 3 * C, C));
        return SWIN_OK;
    }
    if (p[44]) CHK(wgrad_linear_bf16(dy2, hbuf, (float*)p[44], (float*)p[45], T, C, 4 * C, wst));                           // fc2
    if (p[42]) CHK(wgrad_linear_bf16(p[31], p[8], (float*)p[42], db1_from_wgrad ? (float*)p[43] : nullptr, T, 4 * C, C, wst));       // fc1
    if (p[40]) CHK(wgrad_linear_bf16(dy, p[4], (float*)p[40], (float*)p[41], T, C, C, wst));                                // proj
    if (p[37]) CHK(wgrad_linear_bf16(p[35], p[0], (float*)p[37], (float*)p[38], T, 3 * C, C, wst));                         // qkv
    return SWIN_OK;
}
