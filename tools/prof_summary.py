"""Summarise a rocprofv3 kernel_stats.csv by category (development aid).  usage: prof_summary.py stats.csv steps [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40


def cat(n):
    if 'win_attn' in n or 'dbias_slab' in n or 'rel_bias' in n: return 'attention'
    if 'wgrad' in n: return 'wgrad'
    if 'gemm_bf16' in n: return 'conv/gemm (mine)'
    if 'Cijk' in n: return 'hipBLASLt'
    if 'ln_' in n or 'layernorm' in n or 'patch_merge' in n: return 'layernorm'
    if 'gelu' in n: return 'bias_gelu'
    if 'roi_align' in n: return 'roi_align'
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
    return 'elementwise (torch)'


d = collections.defaultdict(lambda: [0.0, 0])
tot = cnt = 0
for r in rows:
    ns = float(r['TotalDurationNs']); k = cat(r['Name'])
    d[k][0] += ns; d[k][1] += int(r['Calls']); tot += ns; cnt += int(r['Calls'])
print("# note: bench.py's live roofline measurements (35 attention + 5 x 23 GEMM-class launches per run) are inside this trace")
print(f"{'category':26s} {'ms/step':>8s} {'launches/step':>14s}")
for k, v in sorted(d.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:26s} {v[0] / steps / 1e6:8.3f} {v[1] / steps:14.1f}")
print(f"{'TOTAL':26s} {tot / steps / 1e6:8.3f} {cnt / steps:14.1f}")
print()
for r in rows[:top]:
    n = re.sub(r'at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_400001_NS::detail::', '', r['Name'])
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms {int(r['Calls']) / steps:6.1f}x {float(r['AverageNs']) / 1e3:8.1f} us  {n[:150]}")
