"""``SwinTransformer`` backbone on the HIP hot path.

Drop-in for the class registered at ``mmdet/models/backbones/swin_transformer.py:448`` of the
reference: same registry name, constructor kwargs (:478-497), ``forward(x) -> tuple`` of
NCHW feature maps (:600-625), ``init_weights`` (:574-598), ``train`` / frozen stages
(:557-572, :627-630) and state_dict keys, so ``configs/swin/*.py`` and
reference / ImageNet checkpoints load unchanged.

What differs is the execution plan.  The reference pads, rolls,
partitions, attends, reverses, un-rolls and crops with eight full-tensor copies per block;
here every per-token op (LN, qkv / proj / MLP linears, residuals) runs on the natural
(B, H*W, C) token grid, and ONE kernel (``ops.window_attention``) does
pad + roll + partition + bias + mask + softmax + PV + reverse + un-roll + crop by address
arithmetic.  Padded tokens are zero *after* norm1 (:211-218), so their q|k|v equal the qkv bias:
the kernel substitutes it instead of running them through the Linear.  Residual adds and DropPath
are fused with the following LayerNorm (``ops.add_layer_norm``).

The module itself is glue: submodules only hold parameters under the reference's names.
"""
import math

import torch

from ._lib import half_dtype as _H
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from . import mixed, ops
from .registry import BACKBONES

_WS = 7
_HEAD_DIM = 32


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class _Params(nn.Module):
    """A parameter holder; never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; the owning SwinTransformer runs the fused plan")


class Mlp(_Params):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention(_Params):
    """Parameters of swin_transformer.py:87-119 (table, index buffer, qkv, proj)."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing='ij'))
        cf = torch.flatten(coords, 1)
        rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1))   # kept for checkpoint compatibility
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinTransformerBlock(_Params):
    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_path=0.):
        super().__init__()
        assert 0 <= shift_size < window_size, "shift_size must in 0-window_size"
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.drop_path_prob = float(drop_path)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, _to_2tuple(window_size), num_heads, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchMerging(_Params):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)


class BasicLayer(_Params):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, qk_scale, drop_path, downsample,
                 use_checkpoint):
        super().__init__()
        self.window_size, self.shift_size, self.depth, self.use_checkpoint = window_size, window_size // 2, depth, use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio,
                                 qkv_bias, qk_scale, drop_path[i] if isinstance(drop_path, list) else drop_path)
            for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None


class PatchEmbed(_Params):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm=True):
        super().__init__()
        self.patch_size, self.in_chans, self.embed_dim = _to_2tuple(patch_size), in_chans, embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = nn.LayerNorm(embed_dim) if norm else None


def _lin(x, weight, bias, dtype):
    """Linear on the compute-dtype copies of the fp32 master parameters (library GEMM forward / dgrad,
    hand-written weight+bias gradient kernel)."""
    return ops.linear(x, weight, bias, dtype)


@BACKBONES.register_module()
class SwinTransformer(nn.Module):
    """See module docstring.  Extra (non-reference) kwarg: ``compute_dtype`` -- torch.float32
    (parity path, default) or torch.bfloat16 (training path: bf16 activations/MFMA, fp32 master
    params, fp32 softmax / LN statistics / accumulation)."""

    def __init__(self, pretrain_img_size=224, patch_size=4, in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, ape=False, patch_norm=True,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False, compute_dtype=torch.float32):
        super().__init__()
        if window_size != _WS:
            raise NotImplementedError("the HIP window-attention kernel is specialised for window_size=7 "
                                      "(every configs/swin/*.py)")
        if any(embed_dim * 2 ** i != h * _HEAD_DIM for i, h in enumerate(num_heads)):
            raise NotImplementedError("the HIP window-attention kernel is specialised for head_dim=32 (Swin-T/S/B)")
        if _to_2tuple(patch_size) != (4, 4) or in_chans != 3:
            raise NotImplementedError("patch embedding kernel is specialised for patch_size=4, in_chans=3")
        if drop_rate != 0. or attn_drop_rate != 0.:
            raise NotImplementedError("drop_rate / attn_drop_rate are 0 in every swin config")
        if norm_layer is not nn.LayerNorm or qk_scale is not None:
            raise NotImplementedError("norm_layer must be nn.LayerNorm and qk_scale None")
        self.pretrain_img_size = pretrain_img_size
        self.num_layers = len(depths)
        self.embed_dim, self.ape, self.patch_norm = embed_dim, ape, patch_norm
        self.out_indices, self.frozen_stages = out_indices, frozen_stages
        if compute_dtype in (torch.bfloat16, torch.float16):
            from . import _lib
            _lib.set_half_dtype(compute_dtype)    # the process's 16-bit type (and library build)
        self.compute_dtype = compute_dtype
        self.fused_blocks = True       # one autograd node per block on the bf16 GPU path (ops/swin_block.py)
        self._dp_replay = None
        self._dp_pool = []
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, patch_norm)
        if ape:
            pis, ps = _to_2tuple(pretrain_img_size), _to_2tuple(patch_size)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, pis[0] // ps[0], pis[1] // ps[1]))
            trunc_normal_(self.absolute_pos_embed, std=.02)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]   # :525
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio,
                                          qkv_bias, qk_scale, dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                          i < self.num_layers - 1, use_checkpoint))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i in out_indices:
            self.add_module(f'norm{i}', nn.LayerNorm(self.num_features[i]))
        self._freeze_stages()

    def parameters_in_forward_order(self):
        """The parameters in the order the forward pass first uses them.  Registration order puts the output norms (norm0..3)
        after all stages; in backward they arrive between the stages, so a gradient bucket formed by reverse REGISTRATION order
        holds norm0 next to stage-4 parameters and cannot be reduced until the very end of backward."""
        out, seen = [], set()

        def add(ps):
            for p in ps:
                if id(p) not in seen:
                    seen.add(id(p)); out.append(p)
        add(self.patch_embed.parameters())
        if getattr(self, 'ape', False):
            add([self.absolute_pos_embed])
        for i, layer in enumerate(self.layers):
            for blk in layer.blocks:
                add(blk.parameters())
            if i in self.out_indices:
                add(getattr(self, f'norm{i}').parameters())
            if getattr(layer, 'downsample', None) is not None:
                add(layer.downsample.parameters())
        add(self.parameters())                       # anything not named above, in registration order
        return out

    # ------------------------------------------------------------------ reference plumbing
    def _freeze_stages(self):                                   # swin_transformer.py:557-572
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1 and self.ape:
            self.absolute_pos_embed.requires_grad = False
        if self.frozen_stages >= 2:
            for i in range(0, self.frozen_stages - 1):
                m = self.layers[i]
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False

    def init_weights(self, pretrained=None):                    # swin_transformer.py:574-598
        def _init(m):
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if isinstance(pretrained, str):
            self.apply(_init)
            from .checkpoint import load_checkpoint
            load_checkpoint(self, pretrained, strict=False)
        elif pretrained is None:
            self.apply(_init)
        else:
            raise TypeError('pretrained must be a str or None')

    def train(self, mode=True):                                 # :627-630 (returns self, as nn.Module.train / .eval() do)
        super().train(mode)
        self._freeze_stages()
        return self

    # ------------------------------------------------------------------ the fused plan
    def _dp_scale(self, blk, B, device):
        """One DropPath draw: per-sample factor floor(keep + U)/keep (timm), or None."""
        p = blk.drop_path_prob
        if p == 0. or not (self.training and blk.training):
            return None
        if self._dp_replay is not None:          # tests replay the factors a reference run drew
            f = self._dp_replay.pop(0)
            return None if f is None else f.to(device=device, dtype=torch.float32)
        if self._dp_pool:                        # all draws of this forward were made by ONE rand kernel
            return self._dp_pool.pop(0)
        keep = 1.0 - p
        return torch.floor(keep + torch.rand(B, device=device, dtype=torch.float32)) / keep

    def _draw_drop_paths(self, B, device):
        """Every DropPath factor of this forward in three kernels instead of four per draw: rows of a (n_draws, B)
        uniform tensor, in the order the blocks consume them (two per block, swin_transformer.py:252-253)."""
        self._dp_pool = []
        if self._dp_replay is not None or not self.training:
            return
        probs = []
        for layer in self.layers:
            for blk in layer.blocks:
                if blk.drop_path_prob > 0. and blk.training:
                    probs += [blk.drop_path_prob, blk.drop_path_prob]
        if not probs:
            return
        ck = (tuple(probs), str(device), B)
        if getattr(self, '_dp_keep_key', None) != ck:            # constant over training: one host->device copy, ever
            keep = 1.0 - torch.tensor(probs, device=device, dtype=torch.float32)[:, None]
            self._dp_keep = keep.expand(len(probs), B).contiguous()
            self._dp_inv = 1.0 / keep
            self._dp_keep_key = ck
        # floor(keep + U[0,1)) / keep of the reference (swin_transformer.py drop_path) == Bernoulli(keep) / keep: two launches
        f = torch.bernoulli(self._dp_keep).mul_(self._dp_inv)
        self._dp_pool = list(f.unbind(0))

    def _block(self, x, n1, blk, B, H, W, next_norm, dp):
        """x: residual stream (B,L,C); n1 = norm1(x) already computed.  Returns (x_out, n_next) where
        n_next = next_norm(x_out) (fused with the second residual) or None."""
        dt = self.compute_dtype
        a = blk.attn
        L = H * W
        if (self.fused_blocks and dt == _H() and x.is_cuda and a.qkv.bias is not None and blk.mlp.fc1.bias is not None
                and B * L >= ops.functional._MIN_T and blk.dim % 8 == 0):
            # the same kernels in the same order inside ONE autograd node (ops/swin_block.py): host overhead only
            from .ops.swin_block import swin_block
            return swin_block(x, n1, dp, (B, H, W, blk.num_heads, blk.shift_size), blk, next_norm, dt)
        qkv = _lin(n1, a.qkv.weight, a.qkv.bias, dt)                                     # :129
        qkv_bias = a.qkv.bias if a.qkv.bias is not None else torch.zeros(3 * blk.dim, device=x.device)
        o = ops.window_attention(qkv, qkv_bias, a.relative_position_bias_table, B, H, W, blk.num_heads,
                                 blk.shift_size)                                         # :214-247 fused
        y = _lin(o, a.proj.weight, a.proj.bias, dt)                                      # :151
        x, n2 = ops.add_layer_norm(x, y, dp[0], L, blk.norm2.weight, blk.norm2.bias)     # :252 + norm2
        h = _lin(n2, blk.mlp.fc1.weight, None, dt)                                       # :33
        h = ops.bias_gelu(h, blk.mlp.fc1.bias)                                           # :33-34
        y2 = _lin(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias, dt)                           # :36
        if next_norm is not None:
            return ops.add_layer_norm(x, y2, dp[1], L, next_norm.weight, next_norm.bias)  # :253 + next LN
        return ops.add_scaled(x, y2, dp[1], L), None

    def forward(self, x):
        """x (B,3,H,W) float -> tuple of (B, C_i, H_i, W_i) feature maps (channels_last memory: the
        token-major buffers are returned as permuted views, no NCHW copy -- cf. :622)."""
        dt = self.compute_dtype
        B, _, Hi, Wi = x.shape
        self._draw_drop_paths(B, x.device)
        pe = self.patch_embed
        rows = ops.patch_im2row(x.float().contiguous(), dt)                              # :433-438
        Wh, Ww = (Hi + 3) // 4, (Wi + 3) // 4
        # the 4x4 / stride-4 conv as a GEMM over the (c, ky, kx) patch rows; through ops.linear its weight / bias gradients
        # (a 96 x 48 x 128000 contraction: 213 us on the library's 16x16 tile, + a reduce launch) run on the split-T kernel
        t = ops.linear(rows, pe.proj.weight, pe.proj.bias, dt)
        if pe.norm is not None:
            t = ops.layer_norm(t, pe.norm.weight, pe.norm.bias)                          # :441-443
        t = t.view(B, Wh * Ww, self.embed_dim)
        if self.ape:                                                                     # :605-608
            ape = F.interpolate(self.absolute_pos_embed, size=(Wh, Ww), mode='bicubic')
            t = (t.float() + ape.flatten(2).transpose(1, 2)).to(dt).contiguous()
        outs = []
        H, W = Wh, Ww
        for i, layer in enumerate(self.layers):
            blocks = layer.blocks
            out_norm = getattr(self, f'norm{i}') if i in self.out_indices else None
            n = ops.layer_norm(t, blocks[0].norm1.weight, blocks[0].norm1.bias)           # :211 of block 0
            x_normed = None
            for j, blk in enumerate(blocks):
                last = j == len(blocks) - 1
                next_norm = out_norm if last else blocks[j + 1].norm1
                dp = (self._dp_scale(blk, B, t.device), self._dp_scale(blk, B, t.device))
                if layer.use_checkpoint and self.training:
                    t, n = checkpoint.checkpoint(self._block, t, n, blk, B, H, W, next_norm, dp, use_reentrant=False)
                else:
                    t, n = self._block(t, n, blk, B, H, W, next_norm, dp)
                if last:
                    x_normed = n
            if out_norm is not None:                                                     # :618-623
                C = self.num_features[i]
                outs.append(x_normed.view(B, H, W, C).permute(0, 3, 1, 2))
            if layer.downsample is not None:                                             # :397-400
                d = layer.downsample
                m = ops.patch_merge_layer_norm(t, d.norm.weight, d.norm.bias, B, H, W)   # :284-295
                t = _lin(m, d.reduction.weight, None, dt)                                # :296
                H, W = (H + 1) // 2, (W + 1) // 2
        return tuple(outs)
