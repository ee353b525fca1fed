#!/usr/bin/env python
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (``python tests/golden/make_golden.py``): it
imports ``/root/reference/mmdet/models/backbones/swin_transformer.py`` and
``.../necks/fpn.py`` *by file path* with tiny stand-ins for the non-arithmetic
third-party names they import (timm DropPath/to_2tuple/trunc_normal_,
mmcv ConvModule/auto_fp16/xavier_init, the registry decorators), runs them on
seeded inputs/weights and stores inputs + outputs (+ gradients) as ``.npz``.
Nothing of the reference's source is copied; the fixtures are data.

Weights are NOT stored: they are regenerated from a seed by
``oracle.swin_oracle.make_params`` / ``oracle.fpn_oracle.make_params`` and
loaded into the reference module through its own ``load_state_dict``; a
checksum of every parameter set is stored so a drifted RNG is detected.

The reference never travels to the GPU box; tests only read the ``.npz`` files.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SWIN_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import fpn_oracle, swin_oracle  # noqa: E402

DP_LOG = []  # per DropPath call: the per-sample factors actually applied


def _install_shims():
    class DropPath(nn.Module):
        """timm.models.layers.DropPath semantics; records the factors it applied."""

        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                DP_LOG.append(None)
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            r = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
            r.floor_()
            DP_LOG.append((r / keep).flatten().clone())
            return x.div(keep) * r

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls

    mod("timm"); mod("timm.models")
    mod("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x),
        trunc_normal_=torch.nn.init.trunc_normal_)
    mod("mmcv_custom", load_checkpoint=lambda *a, **k: None)
    mod("mmdet"); mod("mmdet.utils", get_root_logger=lambda *a, **k: None)
    mod("mmdet.models")
    mod("mmdet.models.builder", BACKBONES=_Reg(), NECKS=_Reg())
    mod("mmdet.models.backbones")
    mod("mmdet.models.necks")

    class ConvModule(nn.Module):
        """mmcv.cnn.ConvModule with norm_cfg=None, act_cfg=None == Conv2d(bias=True) under .conv"""

        def __init__(self, cin, cout, k, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=None,
                     inplace=False):
            super().__init__()
            assert norm_cfg is None and act_cfg is None and conv_cfg is None
            self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=True)

        def forward(self, x):
            return self.conv(x)

    def xavier_init(m, gain=1, bias=0, distribution="normal"):
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(m.weight, gain=gain)
        if m.bias is not None:
            nn.init.constant_(m.bias, bias)

    mod("mmcv"); mod("mmcv.cnn", ConvModule=ConvModule, xavier_init=xavier_init)
    mod("mmcv.runner", auto_fp16=lambda *a, **k: (lambda f: f))


def _load(path, name, package):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    m.__package__ = package
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def _checksum(p):
    return float(sum(float(v.double().abs().sum()) for v in p.values()))


def _np(t):
    return t.detach().cpu().numpy()


def gen_swin(ref, name, cfg, img_shape, seed, train=False, drop_path_rate=0.0, grads=True, store_inputs=True):
    oi = tuple(range(len(cfg["depths"])))
    p = swin_oracle.make_params(cfg["embed_dim"], cfg["depths"], cfg["num_heads"], seed=seed,
                                out_indices=oi, randomize_norm=True)
    m = ref.SwinTransformer(embed_dim=cfg["embed_dim"], depths=list(cfg["depths"]),
                            num_heads=list(cfg["num_heads"]), drop_path_rate=drop_path_rate,
                            out_indices=oi)
    sd = m.state_dict()
    for k, v in p.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
    missing, unexpected = m.load_state_dict(p, strict=False)
    assert not unexpected and all(k.endswith("relative_position_index") for k in missing), (missing, unexpected)
    m.train(train)
    g = torch.Generator().manual_seed(seed + 1000)
    img = torch.randn(*img_shape, generator=g)
    img.requires_grad_(grads)
    torch.manual_seed(seed + 2000)
    DP_LOG.clear()
    # capture per-block outputs
    blk_out = {}
    hooks = []
    for i, layer in enumerate(m.layers):
        for j, blk in enumerate(layer.blocks):
            hooks.append(blk.register_forward_hook(
                lambda mod_, inp, out, key=f"blk_{i}_{j}": blk_out.__setitem__(key, _np(out))))
    pe = {}
    hooks.append(m.patch_embed.register_forward_hook(lambda mod_, inp, out: pe.__setitem__("pe", _np(out))))
    outs = m(img)
    for h in hooks:
        h.remove()
    data = {"param_checksum": np.float64(_checksum(p)), "seed": np.int64(seed),
            "embed_dim": np.int64(cfg["embed_dim"]), "depths": np.asarray(cfg["depths"]),
            "num_heads": np.asarray(cfg["num_heads"]), "img_shape": np.asarray(img_shape),
            "train": np.int64(train)}
    if store_inputs:
        data["img"] = _np(img)
        data["patch_embed_out"] = pe["pe"]
    for i, o in enumerate(outs):
        data[f"out{i}"] = _np(o)
    # keep only the first two and the last block outputs (size)
    keys = sorted(blk_out)
    for k in (keys[:2] + keys[-1:]) if store_inputs else ():
        data[k] = blk_out[k]
    # DropPath factors, indexed 2*block + {0: attention branch, 1: MLP branch}; blocks whose
    # drop_path is nn.Identity (rate 0) draw nothing
    log = list(DP_LOG)
    n = 0
    for layer in m.layers:
        for blk in layer.blocks:
            if not isinstance(blk.drop_path, nn.Identity):
                for br in range(2):
                    f = log.pop(0)
                    if f is not None:
                        data[f"dp_{2 * n + br}"] = _np(f)
            n += 1
    assert not log
    if grads:
        gw = torch.Generator().manual_seed(seed + 3000)
        loss = sum((o * torch.randn(o.shape, generator=gw)).sum() for o in outs)
        names = ["patch_embed.proj.weight", "layers.0.blocks.1.attn.relative_position_bias_table",
                 "layers.0.blocks.1.attn.qkv.bias", "layers.0.blocks.1.attn.qkv.weight",
                 "layers.0.blocks.0.norm1.weight", "layers.0.blocks.1.mlp.fc1.weight",
                 "layers.0.downsample.reduction.weight", "layers.0.downsample.norm.bias",
                 "layers.1.blocks.1.attn.relative_position_bias_table", "layers.1.blocks.1.attn.qkv.bias",
                 "norm0.weight", f"norm{len(cfg['depths']) - 1}.bias"]
        params = dict(m.named_parameters())
        gs = torch.autograd.grad(loss, [img] + [params[n] for n in names])
        data["grad_img"] = _np(gs[0])
        for n, gg in zip(names, gs[1:]):
            data["grad__" + n] = _np(gg)
        data["loss"] = np.float64(loss.item())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, {k: getattr(v, "shape", None) for k, v in data.items() if k.startswith("out")})


def gen_window_attention(ref, seed=7):
    """WindowAttention.forward alone (masked and unmasked), with grads."""
    C, nH, ws, B, nW = 64, 2, 7, 2, 6
    m = ref.WindowAttention(C, (ws, ws), nH)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g) * 0.2)
    x = torch.randn(B * nW, ws * ws, C, generator=g, requires_grad=True)
    # the mask the reference builds for a 14 x 21 padded grid (2 x 3 windows)
    mask = swin_oracle.shift_attn_mask(14, 21, ws, 3)
    assert mask.shape[0] == nW
    data = {"x": _np(x), "mask": _np(mask)}
    for k, v in m.state_dict().items():
        data["p__" + k] = _np(v)
    for tag, mk in (("nomask", None), ("mask", mask)):
        y = m(x, mk)
        w = torch.randn(y.shape, generator=g)
        gs = torch.autograd.grad((y * w).sum(), [x] + list(m.parameters()))
        data[f"y_{tag}"] = _np(y)
        data[f"w_{tag}"] = _np(w)
        data[f"gx_{tag}"] = _np(gs[0])
        for (n, _), gg in zip(m.named_parameters(), gs[1:]):
            data[f"g_{tag}__{n}"] = _np(gg)
    np.savez_compressed(os.path.join(HERE, "window_attention.npz"), **data)
    print("window_attention ok")


def gen_fpn(reffpn, seed=11, name="fpn_small", in_ch=(8, 16, 32, 64), oc=16, shapes=((20, 28), (10, 14), (5, 7), (3, 4)),
            bf16_operands=False):
    """bf16_operands: parameters and inputs are rounded to bf16-representable values first (the reference still computes in fp32):
    the fixture for the bf16 HIP path, whose MFMA conv then sees exactly these operand values."""
    p = fpn_oracle.make_params(in_ch, oc, seed=seed)
    if bf16_operands:
        p = {k: v.bfloat16().float() for k, v in p.items()}
    m = reffpn.FPN(list(in_ch), oc, 5)
    m.init_weights()
    m.load_state_dict(p, strict=True)
    g = torch.Generator().manual_seed(seed + 1)
    xs = [torch.randn(2, c, h, w, generator=g) for c, (h, w) in zip(in_ch, shapes)]
    if bf16_operands:
        xs = [x.bfloat16().float() for x in xs]
    xs = [x.requires_grad_(True) for x in xs]
    outs = m(tuple(xs))
    ws = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * w).sum() for o, w in zip(outs, ws))
    params = dict(m.named_parameters())
    pn = ["lateral_convs.0.conv.weight", "lateral_convs.3.conv.bias", "fpn_convs.1.conv.weight", "fpn_convs.0.conv.bias"]
    gs = torch.autograd.grad(loss, xs + [params[n] for n in pn])
    data = {"param_checksum": np.float64(_checksum(p)), "seed": np.int64(seed)}
    for i, x in enumerate(xs):
        data[f"in{i}"] = _np(x)
        data[f"gin{i}"] = _np(gs[i])
    for i, (o, w) in enumerate(zip(outs, ws)):
        data[f"out{i}"] = _np(o)
        data[f"w{i}"] = _np(w)
    for n, gg in zip(pn, gs[len(xs):]):
        data["grad__" + n] = _np(gg)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "ok", [tuple(o.shape) for o in outs])


def main():
    torch.set_num_threads(8)
    _install_shims()
    ref = _load(os.path.join(REF, "mmdet/models/backbones/swin_transformer.py"),
                "mmdet.models.backbones.swin_transformer", "mmdet.models.backbones")
    reffpn = _load(os.path.join(REF, "mmdet/models/necks/fpn.py"), "mmdet.models.necks.fpn", "mmdet.models.necks")
    # (a) mini: pads at every stage, odd patch-merge, shifted blocks, both batch items
    mini = dict(embed_dim=32, depths=(2, 2, 2), num_heads=(1, 2, 4))
    gen_swin(ref, "swin_mini_eval", mini, (2, 3, 76, 100), seed=1)
    # (b) mini in train mode with DropPath firing (factors recorded)
    gen_swin(ref, "swin_mini_train_dp", mini, (2, 3, 76, 100), seed=2, train=True, drop_path_rate=0.5)
    # (c) BASELINE cfg1: Swin-T, 1x3x224x224, eval; image regenerated from its seed, outputs stored
    tiny = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24))
    gen_swin(ref, "swin_tiny_224", tiny, (1, 3, 224, 224), seed=3, grads=False, store_inputs=False)
    gen_window_attention(ref)
    gen_fpn(reffpn)
    gen_fpn_c64(reffpn)


def gen_fpn_c64(reffpn):
    # (round 3) FPN wide enough for the hand-written MFMA 3x3 conv (Cin % 64 == 0), operands bf16-representable
    gen_fpn(reffpn, seed=12, name="fpn_c64", in_ch=(64, 128, 192, 256), oc=64, shapes=((16, 24), (8, 12), (4, 6), (2, 3)),
            bf16_operands=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fpn_c64":      # only the fixture added in round 3 (the others are unchanged)
        torch.set_num_threads(8)
        _install_shims()
        gen_fpn_c64(_load(os.path.join(REF, "mmdet/models/necks/fpn.py"), "mmdet.models.necks.fpn", "mmdet.models.necks"))
    else:
        main()
