"""Swin-aware checkpoint loading (reference: ``mmcv_custom/checkpoint.py:286-356``): local files only
(no network in this deployment), key surgery for DDP / MoBY prefixes, absolute-pos-embed reshape and
bicubic resize of ``relative_position_bias_table`` when the window size differs, non-strict load.
Save format of the reference runner (``mmcv_custom/runner/checkpoint.py:19-85``):
``{'meta': ..., 'state_dict': ..., 'optimizer': ...}``.
"""
import torch
import torch.nn.functional as F


def _strip(sd, prefix):
    if all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return sd


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    if '://' in filename:
        raise IOError(f"{filename}: remote checkpoint schemes are not available offline; pass a local path")
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    sd = ckpt.get('state_dict', ckpt.get('model', ckpt))
    sd = _strip(sd, 'module.')                                     # :319-320
    if any(k.startswith('encoder') for k in sd):                   # MoBY :323-324
        sd = {k.replace('encoder.', ''): v for k, v in sd.items() if k.startswith('encoder.')}
    if any(k.startswith('backbone.') for k in sd) and not any(k.startswith('backbone.') for k in model.state_dict()):
        sd = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    own = model.state_dict()
    if sd.get('absolute_pos_embed') is not None and 'absolute_pos_embed' in own:      # :327-335
        ape = sd['absolute_pos_embed']
        N2, C2, H, W = own['absolute_pos_embed'].shape
        if ape.dim() == 3:
            N1, L, C1 = ape.shape
            if N1 == N2 and C1 == C2 and L == H * W:
                sd['absolute_pos_embed'] = ape.view(N2, H, W, C2).permute(0, 3, 1, 2)
    for k in [k for k in sd if 'relative_position_bias_table' in k]:                  # :337-352
        if k not in own:
            continue
        t, cur = sd[k], own[k]
        (L1, nH1), (L2, nH2) = t.shape, cur.shape
        if nH1 == nH2 and L1 != L2:
            S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
            r = F.interpolate(t.permute(1, 0).view(1, nH1, S1, S1), size=(S2, S2), mode='bicubic')
            sd[k] = r.view(nH2, L2).permute(1, 0)
    # mmcv's load_state_dict (mmcv_custom/checkpoint.py:41-106) reports size mismatches instead of raising when not strict
    # (e.g. a bias table with another head count, :342-343; a detection head with another class count)
    mismatched = [k for k, v in sd.items() if k in own and tuple(v.shape) != tuple(own[k].shape)]
    if mismatched and not strict:
        sd = {k: v for k, v in sd.items() if k not in mismatched}
    missing, unexpected = model.load_state_dict(sd, strict=strict)
    from . import mixed
    mixed.refresh_all()            # bf16 shadows / constants / re-laid-out weights of a live training setup follow the new masters
    if logger is not None and (missing or unexpected or mismatched):
        logger.warning(f'missing keys: {missing}; unexpected keys: {unexpected}; size mismatch (skipped): {mismatched}')
    return ckpt


def save_checkpoint(model, filename, optimizer=None, meta=None):
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    ckpt = {'meta': meta or {}, 'state_dict': sd}
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    torch.save(ckpt, filename)
