"""Kernel name -> category (shared by prof_summary.py and step_trace.py)."""


def cat(n):
    if 'win_attn' in n or 'dbias_slab' in n or 'rel_bias' in n: return 'attention'
    if 'wgrad' in n: return 'wgrad'
    if 'ts_linear' in n or 'ts_proj_add_ln' in n: return 'linear GEMM (mine)'
    if 'conv_halo' in n or 'gemm_bf16_kernel<ConvA' in n or 'gemm_bf16_kernelI5ConvA' in n or 'splitk_finish' in n or 'conv_dgrad_layout' in n:
        return 'conv3x3 (mine)'
    if 'gemm_bf16' in n or 'narrow_dgrad' in n or 'linear_t_layout' in n: return 'linear GEMM (mine)'
    if 'tail_reduce' in n: return 'block tail reduce (mine)'
    if 'Cijk' in n: return 'hipBLASLt'
    if 'ln_' in n or 'layernorm' in n or 'patch_merge' in n: return 'layernorm'
    if 'gelu' in n: return 'bias_gelu'
    if 'roi_align' in n or 'roi_gather' in n: return 'roi_align'
    if 'nms' in n: return 'nms'
    if 'rpn_topk' in n: return 'rpn select (mine)'
    if 'ts_mlp' in n: return 'fused mlp (mine)'
    if 'assign_' in n or 'sample_' in n: return 'targets'
    if 'upsample' in n or 'im2row' in n: return 'fpn/embed (mine)'
    if 'bn_' in n: return 'batchnorm'
    if 'loss' in n or 'regress' in n or 'bbox_targets' in n or 'delta2bbox' in n or 'rpn_flatten' in n: return 'losses/targets (mine)'
    if 'adamw_kernel' in n: return 'optimizer (mine)'
    if 'multi_tensor' in n: return 'optimizer (torch)'
    if 'rocprim' in n or 'sort' in n.lower() or 'topk' in n.lower(): return 'sort/topk (torch)'
    if 'rocclr' in n: return 'memcpy/fill (runtime)'
    if 'rpn_' in n or 'flat_' in n or 'anchor' in n or 'mask_target' in n or 'zero_segments' in n: return 'losses/targets (mine)'
    return 'elementwise (torch)'
