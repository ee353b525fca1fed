"""Torch-CPU fp32 restatement of the Swin backbone forward -- TEST INFRASTRUCTURE ONLY.

Follows ``mmdet/models/backbones/swin_transformer.py`` of the reference
(line numbers cited per function).  Written functionally over a flat
``params`` dict that uses the reference's state_dict keys
so that the golden fixtures (produced by the reference itself, see
tests/golden/make_golden.py) can be replayed key by key.

Pinned: yes -- tests/test_oracle_swin.py checks every function here against
the reference-generated golden vectors.

Autograd works through every function (they are plain torch ops), which is how
the backward parity tests obtain reference gradients.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-5  # nn.LayerNorm default, swin_transformer.py:492


# ----------------------------------------------------------------------------
# index / integer helpers (numpy)
# ----------------------------------------------------------------------------
def relative_position_index(ws=7):
    """swin_transformer.py:101-110 -- (dh+ws-1)*(2ws-1) + (dw+ws-1), query minus key."""
    coords = np.stack(np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij"))  # 2,ws,ws
    flat = coords.reshape(2, -1)
    rel = flat[:, :, None] - flat[:, None, :]  # 2, N, N
    rel = rel.transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).astype(np.int64)  # N, N


def padded_hw(H, W, ws=7):
    """swin_transformer.py:371-372."""
    return int(np.ceil(H / ws)) * ws, int(np.ceil(W / ws)) * ws


def region_ids(Hp, Wp, ws=7, shift=3):
    """swin_transformer.py:373-384 -- 3x3 region id image on the padded grid."""
    img = np.zeros((Hp, Wp), dtype=np.float32)
    slices = (slice(0, -ws), slice(-ws, -shift), slice(-shift, None))
    cnt = 0
    for h in slices:
        for w in slices:
            img[h, w] = cnt
            cnt += 1
    return img


def shift_attn_mask(H, W, ws=7, shift=3):
    """swin_transformer.py:371-389 -- (nW, N, N) fp32 mask, 0 / -100."""
    Hp, Wp = padded_hw(H, W, ws)
    img = torch.from_numpy(region_ids(Hp, Wp, ws, shift))[None, :, :, None]
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


# ----------------------------------------------------------------------------
# layout helpers
# ----------------------------------------------------------------------------
def window_partition(x, ws):
    """swin_transformer.py:41-53."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws, H, W):
    """swin_transformer.py:56-70."""
    B = int(windows.shape[0] / (H * W / ws / ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


# ----------------------------------------------------------------------------
# modules
# ----------------------------------------------------------------------------
def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def gelu(x):
    """nn.GELU (exact erf form), swin_transformer.py:23,34."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def mlp(x, p, pre):
    """Mlp.forward swin_transformer.py:32-38 (dropouts p=0)."""
    h = F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])
    h = gelu(h)
    return F.linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def window_attention_core(qkv, bias_table, num_heads, mask=None, ws=7):
    """q,k,v -> o part of WindowAttention.forward, swin_transformer.py:129-150.

    qkv: (B_, N, 3C) already projected (``self.qkv(x)``).  Returns (B_, N, C).
    """
    B_, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // num_heads
    scale = hd ** -0.5                                              # :94
    qkv = qkv.reshape(B_, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)  # :129
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * scale                                                   # :132
    attn = q @ k.transpose(-2, -1)                                  # :133
    idx = torch.from_numpy(relative_position_index(ws)).view(-1)
    rpb = bias_table[idx].view(N, N, -1).permute(2, 0, 1).contiguous()  # :135-137
    attn = attn + rpb.unsqueeze(0)                                  # :138
    if mask is not None:                                            # :140-143
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, num_heads, N, N)
    attn = torch.softmax(attn, dim=-1)                              # :144/146
    return (attn @ v).transpose(1, 2).reshape(B_, N, C)             # :150


def window_attention(x, p, pre, num_heads, mask=None, ws=7):
    """WindowAttention.forward swin_transformer.py:121-153. x: (B_, N, C)."""
    qkv = F.linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"])
    o = window_attention_core(qkv, p[pre + "relative_position_bias_table"], num_heads, mask, ws)
    return F.linear(o, p[pre + "proj.weight"], p[pre + "proj.bias"])  # :151


def drop_path(x, keep_mask_scale):
    """timm DropPath as used at swin_transformer.py:190,252-253.

    ``keep_mask_scale`` is the per-sample factor ``floor(keep + U)/keep`` of shape
    (B,) (or None = identity).  The random draw is made by the caller so the
    oracle and the HIP path consume the same factors.
    """
    if keep_mask_scale is None:
        return x
    return x * keep_mask_scale.view(-1, *([1] * (x.dim() - 1)))


def swin_block(x, H, W, p, pre, num_heads, shift, mask, ws=7, dp_scale=None):
    """SwinTransformerBlock.forward swin_transformer.py:198-255. x: (B, L, C)."""
    B, L, C = x.shape
    assert L == H * W, "input feature has wrong size"              # :208
    shortcut = x
    x = layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"])  # :211
    x = x.view(B, H, W, C)
    pad_r = (ws - W % ws) % ws                                      # :216
    pad_b = (ws - H % ws) % ws                                      # :217
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))                        # :218 zeros AFTER LN
    _, Hp, Wp, _ = x.shape
    if shift > 0:                                                   # :222-227
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
        attn_mask = mask
    else:
        attn_mask = None
    xw = window_partition(x, ws).view(-1, ws * ws, C)               # :230-231
    aw = window_attention(xw, p, pre + "attn.", num_heads, attn_mask, ws)  # :234
    x = window_reverse(aw.view(-1, ws, ws, C), ws, Hp, Wp)          # :237-238
    if shift > 0:                                                   # :241-244
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    if pad_r > 0 or pad_b > 0:                                      # :246-247
        x = x[:, :H, :W, :].contiguous()
    x = x.view(B, H * W, C)
    dp1, dp2 = (None, None) if dp_scale is None else dp_scale       # two independent draws
    x = shortcut + drop_path(x, dp1)                                # :252
    y = mlp(layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp.")
    return x + drop_path(y, dp2)                                    # :253


def patch_merging(x, H, W, p, pre):
    """PatchMerging.forward swin_transformer.py:271-298."""
    B, L, C = x.shape
    assert L == H * W, "input feature has wrong size"              # :279
    x = x.view(B, H, W, C)
    if (H % 2 == 1) or (W % 2 == 1):                                # :284-286
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x0 = x[:, 0::2, 0::2, :]                                        # :288-291
    x1 = x[:, 1::2, 0::2, :]
    x2 = x[:, 0::2, 1::2, :]
    x3 = x[:, 1::2, 1::2, :]
    x = torch.cat([x0, x1, x2, x3], -1).view(B, -1, 4 * C)          # :292-293
    x = layer_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"])  # :295
    return F.linear(x, p[pre + "reduction.weight"])                 # :296 (no bias)


def patch_embed(img, p, patch=4):
    """PatchEmbed.forward swin_transformer.py:429-445 -> (B, C, Wh, Ww)."""
    _, _, H, W = img.shape
    if W % patch != 0:                                              # :433-434
        img = F.pad(img, (0, patch - W % patch))
    if H % patch != 0:                                              # :435-436
        img = F.pad(img, (0, 0, 0, patch - H % patch))
    x = F.conv2d(img, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=patch)  # :438
    if "patch_embed.norm.weight" in p:                              # :439-443
        B, C, Wh, Ww = x.shape
        x = x.flatten(2).transpose(1, 2)
        x = layer_norm(x, p["patch_embed.norm.weight"], p["patch_embed.norm.bias"])
        x = x.transpose(1, 2).view(-1, C, Wh, Ww)
    return x


def basic_layer(x, H, W, p, i, depth, num_heads, has_down, ws=7, dp_scales=None):
    """BasicLayer.forward swin_transformer.py:362-402."""
    shift = ws // 2                                                 # :336
    mask = shift_attn_mask(H, W, ws, shift)                         # :371-389
    for j in range(depth):                                          # :391-396
        s = 0 if j % 2 == 0 else shift                              # :346
        dps = None if dp_scales is None else dp_scales[j]
        x = swin_block(x, H, W, p, f"layers.{i}.blocks.{j}.", num_heads, s, mask, ws, dps)
    if has_down:                                                    # :397-400
        xd = patch_merging(x, H, W, p, f"layers.{i}.downsample.")
        return x, H, W, xd, (H + 1) // 2, (W + 1) // 2
    return x, H, W, x, H, W                                         # :402


def swin_forward(img, p, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), ws=7,
                 out_indices=(0, 1, 2, 3), dp_scales=None):
    """SwinTransformer.forward swin_transformer.py:600-625 (ape=False).

    ``dp_scales``: None (eval / drop_path off) or a list over all blocks of
    ``None`` / ``(attn_factors, mlp_factors)`` -- the two per-sample DropPath
    factor vectors (B,) a block draws (one per ``self.drop_path`` call, :252-253).
    Returns a tuple of NCHW tensors.
    """
    x = patch_embed(img, p)                                         # :602
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)                                # :610
    outs = []
    blk0 = 0
    for i, d in enumerate(depths):
        dps = None if dp_scales is None else dp_scales[blk0:blk0 + d]
        blk0 += d
        x_out, H, W, x, Wh, Ww = basic_layer(x, Wh, Ww, p, i, d, num_heads[i],
                                             i < len(depths) - 1, ws, dps)  # :616
        if i in out_indices:                                        # :618-623
            C = x_out.shape[-1]
            xo = layer_norm(x_out, p[f"norm{i}.weight"], p[f"norm{i}.bias"])
            outs.append(xo.view(-1, H, W, C).permute(0, 3, 1, 2).contiguous())
    return tuple(outs)


# ----------------------------------------------------------------------------
# parameter construction (reference init rules, swin_transformer.py:582-589,118)
# ----------------------------------------------------------------------------
def make_params(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), ws=7,
                mlp_ratio=4.0, in_chans=3, patch=4, out_indices=(0, 1, 2, 3), seed=0,
                randomize_norm=False):
    """Seeded parameter dict with the reference's keys/shapes.

    Linear: trunc_normal std .02, bias 0; LayerNorm 1/0; bias table trunc_normal
    std .02; patch_embed conv: torch default (kaiming-uniform) like the reference,
    which never re-initialises the conv.  ``randomize_norm`` perturbs LN affine
    params and biases so tests exercise them (the reference init leaves them at
    1/0, which would hide a swapped weight/bias).
    """
    g = torch.Generator().manual_seed(seed)

    def tn(*shape):
        t = torch.empty(*shape)
        torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
        return t

    def nw(n):
        return 1.0 + 0.1 * torch.randn(n, generator=g) if randomize_norm else torch.ones(n)

    def nb(n):
        return 0.1 * torch.randn(n, generator=g) if randomize_norm else torch.zeros(n)

    p = {}
    fan_in = in_chans * patch * patch
    bound = 1.0 / math.sqrt(fan_in)
    p["patch_embed.proj.weight"] = (torch.rand(embed_dim, in_chans, patch, patch, generator=g) * 2 - 1) * bound
    p["patch_embed.proj.bias"] = (torch.rand(embed_dim, generator=g) * 2 - 1) * bound
    p["patch_embed.norm.weight"] = nw(embed_dim)
    p["patch_embed.norm.bias"] = nb(embed_dim)
    for i, d in enumerate(depths):
        C = embed_dim * 2 ** i
        hid = int(C * mlp_ratio)
        for j in range(d):
            pre = f"layers.{i}.blocks.{j}."
            p[pre + "norm1.weight"] = nw(C)
            p[pre + "norm1.bias"] = nb(C)
            p[pre + "attn.relative_position_bias_table"] = tn((2 * ws - 1) ** 2, num_heads[i])
            p[pre + "attn.qkv.weight"] = tn(3 * C, C)
            p[pre + "attn.qkv.bias"] = nb(3 * C)
            p[pre + "attn.proj.weight"] = tn(C, C)
            p[pre + "attn.proj.bias"] = nb(C)
            p[pre + "norm2.weight"] = nw(C)
            p[pre + "norm2.bias"] = nb(C)
            p[pre + "mlp.fc1.weight"] = tn(hid, C)
            p[pre + "mlp.fc1.bias"] = nb(hid)
            p[pre + "mlp.fc2.weight"] = tn(C, hid)
            p[pre + "mlp.fc2.bias"] = nb(C)
        if i < len(depths) - 1:
            pre = f"layers.{i}.downsample."
            p[pre + "reduction.weight"] = tn(2 * C, 4 * C)
            p[pre + "norm.weight"] = nw(4 * C)
            p[pre + "norm.bias"] = nb(4 * C)
    for i in out_indices:
        C = embed_dim * 2 ** i
        p[f"norm{i}.weight"] = nw(C)
        p[f"norm{i}.bias"] = nb(C)
    return p
