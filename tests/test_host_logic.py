"""CPU tests of the host-side mirror: registry, config loader, module construction / state_dict schema,
box utilities, assigner and sampler (pure torch, device independent), against the oracle's numpy callers
and, when the reference tree is present, against the reference's config files."""
import glob
import os

import numpy as np
import pytest
import torch

import swin_transformer_object_detection_amd as pkg
from oracle import callers_oracle as CO
from swin_transformer_object_detection_amd import config, detector, presets, registry

REF = "/root/reference"
has_ref = os.path.isdir(os.path.join(REF, "configs", "swin"))


def test_registry_and_build_from_cfg():
    assert "SwinTransformer" in registry.BACKBONES and "FPN" in registry.NECKS
    assert "SingleRoIExtractor" in registry.ROI_EXTRACTORS and "MaskRCNN" in registry.DETECTORS
    m = registry.build_backbone(dict(type="SwinTransformer", embed_dim=32, depths=[2, 2], num_heads=[1, 2],
                                     out_indices=(0, 1)))
    assert isinstance(m, pkg.SwinTransformer) and m.num_features == [32, 64]
    with pytest.raises(KeyError):
        registry.build_backbone(dict(type="ResNet"))
    with pytest.raises(TypeError):
        m.init_weights(pretrained=3)                      # swin_transformer.py:598
    with pytest.raises(NotImplementedError):
        registry.build_backbone(dict(type="SwinTransformer", window_size=12))


def test_swin_tiny_state_dict_schema():
    """189 entries, 27.52 M parameters, the reference's key names."""
    m = pkg.SwinTransformer()
    sd = m.state_dict()
    assert len(sd) == 189
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 27.52) < 0.01
    assert sd["layers.0.blocks.1.attn.relative_position_bias_table"].shape == (169, 3)
    assert sd["layers.2.blocks.5.attn.relative_position_index"].dtype == torch.int64
    assert sd["layers.0.blocks.0.attn.relative_position_index"][0, :8].tolist() == [84, 83, 82, 81, 80, 79, 78, 71]
    assert sd["layers.1.downsample.reduction.weight"].shape == (384, 768)
    assert sd["patch_embed.proj.weight"].shape == (96, 3, 4, 4) and "norm3.bias" in sd
    f = pkg.FPN([96, 192, 384, 768], 256, 5)
    assert sorted(f.state_dict())[:2] == ["fpn_convs.0.conv.bias", "fpn_convs.0.conv.weight"]
    assert "lateral_convs.3.conv.weight" in f.state_dict()


def test_frozen_stages_and_train_mode():
    m = pkg.SwinTransformer(embed_dim=32, depths=[2, 2, 2], num_heads=[1, 2, 4], out_indices=(0, 1, 2), frozen_stages=2)
    m.train()
    assert not any(p.requires_grad for p in m.patch_embed.parameters())
    assert not any(p.requires_grad for p in m.layers[0].parameters())
    assert all(p.requires_grad for p in m.layers[1].parameters())
    assert not m.layers[0].training and m.layers[1].training


def test_config_loader_merge_and_delete(tmp_path):
    (tmp_path / "base.py").write_text("model = dict(type='A', backbone=dict(type='R', depth=50), neck=dict(k=1))\nlr = 0.1\n")
    (tmp_path / "child.py").write_text("_base_ = './base.py'\nmodel = dict(backbone=dict(_delete_=True, type='S', dim=96), neck=dict(j=2))\n")
    c = config.Config.fromfile(str(tmp_path / "child.py"))
    assert c.model.backbone == dict(type='S', dim=96) and c.model.neck == dict(k=1, j=2) and c.lr == 0.1
    c.merge_from_dict({"model.backbone.use_checkpoint": True})
    assert c.model.backbone.use_checkpoint is True and c.model.type == 'A'


@pytest.mark.skipif(not has_ref, reason="reference tree not present (GPU box)")
def test_reference_swin_configs_load_unchanged():
    files = sorted(glob.glob(os.path.join(REF, "configs", "swin", "*.py")))
    assert len(files) == 7
    for f in files:
        cfg = config.Config.fromfile(f)
        bb = registry.build_backbone(cfg.model.backbone)
        nk = registry.build_neck(cfg.model.neck)
        assert bb.num_features[-1] == cfg.model.neck.in_channels[-1] and nk.num_outs == 5
        assert cfg.optimizer.type == 'AdamW' and cfg.runner.type == 'EpochBasedRunnerAmp'
        det = pkg.build_detector(cfg.model)
        if cfg.model.type == 'MaskRCNN':
            assert abs(sum(p.numel() for p in det.parameters()) / 1e6 - {96: 47.8, }.get(cfg.model.backbone.embed_dim, 0)) < 0.2 \
                or cfg.model.backbone.depths[2] == 18
        else:                                   # Cascade Mask R-CNN: 3 ConvFCBBoxHead (4conv1fc + SyncBN) and 3 mask heads
            assert cfg.model.type == 'CascadeRCNN' and len(det.roi_head.bbox_head) == 3 and len(det.roi_head.mask_head) == 3
            keys = set(det.state_dict())
            assert {'roi_head.bbox_head.0.shared_convs.3.conv.weight', 'roi_head.bbox_head.2.shared_convs.0.bn.running_var',
                    'roi_head.bbox_head.1.shared_fcs.0.weight', 'roi_head.mask_head.2.conv_logits.bias'} <= keys
            assert 'roi_head.bbox_head.0.shared_convs.0.conv.bias' not in keys          # ConvModule bias='auto' with a norm
            assert det.roi_head.bbox_head[0].shared_fcs[0].weight.shape == (1024, 256 * 49)


@pytest.mark.skipif(not has_ref, reason="reference tree not present (GPU box)")
def test_presets_equal_reference_config():
    def plain(d):
        if isinstance(d, dict):
            return {k: plain(v) for k, v in d.items()}
        if isinstance(d, (list, tuple)):
            return [plain(v) for v in d]
        return d
    cfg = config.Config.fromfile(os.path.join(
        REF, "configs/swin/mask_rcnn_swin_tiny_patch4_window7_mstrain_480-800_adamw_1x_coco.py"))
    assert plain(cfg.to_dict()["model"]) == plain(presets.mask_rcnn_swin("tiny"))
    assert cfg.optimizer.lr == presets.OPTIMIZER["lr"] and cfg.optimizer.weight_decay == presets.OPTIMIZER["weight_decay"]
    assert set(cfg.optimizer.paramwise_cfg.custom_keys) == set(presets.OPTIMIZER["no_decay_keys"])
    for variant, name in (("base", "cascade_mask_rcnn_swin_base_patch4_window7_mstrain_480-800_giou_4conv1f_adamw_3x_coco.py"),
                          ("tiny", "cascade_mask_rcnn_swin_tiny_patch4_window7_mstrain_480-800_giou_4conv1f_adamw_3x_coco.py")):
        cfg = config.Config.fromfile(os.path.join(REF, "configs/swin", name))
        assert plain(cfg.to_dict()["model"]) == plain(presets.cascade_mask_rcnn_swin(variant))


def test_box_utils_match_oracle_callers():
    rng = np.random.RandomState(0)
    rois = np.array([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]], np.float32)
    deltas = np.array([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]], np.float32)
    got = detector.delta2bbox(torch.from_numpy(rois), torch.from_numpy(deltas), max_shape=(32, 32))
    np.testing.assert_allclose(got.numpy(), CO.delta2bbox(rois, deltas, max_shape=(32, 32)), atol=1e-6)
    ag = detector.AnchorGenerator([4, 8], [0.5, 1.0, 2.0], [8])
    a = ag.grid_anchors([(3, 5), (2, 2)], torch.device("cpu"))
    np.testing.assert_allclose(a[0].numpy(), CO.grid_anchors(3, 5, 4), atol=1e-4)
    np.testing.assert_allclose(a[1].numpy(), CO.grid_anchors(2, 2, 8), atol=1e-4)
    # encode / decode round trip
    xy = rng.rand(50, 2).astype(np.float32) * 100
    p = torch.from_numpy(np.concatenate([xy, xy + rng.rand(50, 2).astype(np.float32) * 50 + 2], 1))
    xy = rng.rand(50, 2).astype(np.float32) * 100
    g = torch.from_numpy(np.concatenate([xy, xy + rng.rand(50, 2).astype(np.float32) * 50 + 2], 1))
    d = detector.bbox2delta(p, g, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2))
    back = detector.delta2bbox(p, d, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2))
    np.testing.assert_allclose(back.numpy(), g.numpy(), atol=1e-3)
    rois5 = torch.tensor([[0, 0, 0, 10, 10], [0, 0, 0, 112, 112], [1, 0, 0, 448, 448]], dtype=torch.float32)
    ex = detector.SingleRoIExtractor(dict(type='RoIAlign', output_size=7, sampling_ratio=0), 256, [4, 8, 16, 32])
    assert ex.map_roi_levels(rois5, 4).tolist() == CO.map_roi_levels(rois5.numpy(), 4).tolist() == [0, 1, 3]
    assert ex.roi_layers[2].spatial_scale == 1 / 16 and ex.roi_layers[0].output_size == (7, 7)


def test_max_iou_assigner_semantics():
    """Known-answer case of the reference's tests/test_utils/test_assigner.py:14-35."""
    bboxes = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [5, 5, 15, 15], [32, 32, 38, 42]])
    gt = torch.FloatTensor([[0, 0, 10, 9], [0, 10, 10, 19]])
    assigned, _, labels = detector.max_iou_assign(bboxes, gt, 0.5, 0.5, 0.0, True, torch.LongTensor([2, 3]))   # min_pos_iou default .0
    assert assigned.tolist() == [1, 0, 2, 0]
    assert labels.tolist() == [2, -1, 3, -1]
    a0, _, _ = detector.max_iou_assign(bboxes, torch.zeros(0, 4), 0.5, 0.5, 0.5)
    assert a0.tolist() == [0, 0, 0, 0]
    torch.manual_seed(0)
    big = torch.cat([torch.ones(300, dtype=torch.long), torch.zeros(700, dtype=torch.long), -torch.ones(50, dtype=torch.long)])
    pos, neg = detector.random_sample(big, 256, 0.5)
    assert pos.numel() == 128 and neg.numel() == 128 and (big[pos] > 0).all() and (big[neg] == 0).all()
    pos, neg = detector.random_sample(big[280:], 512, 0.25)
    assert pos.numel() == 20 and neg.numel() == 492


def test_shadow_params_and_gradient_gather():
    """bf16 shadow leaves (mixed.ShadowParams) + the reducer's bucket gather: masters receive fp32 gradients in
    the flat buckets, shadows follow the masters after refresh(), fp32 small parameters keep direct gradients."""
    import torch.nn as nn
    import torch.nn.functional as F
    from swin_transformer_object_detection_amd import ddp, mixed
    torch.manual_seed(0)
    lin1, lin2 = nn.Linear(64, 32), nn.Linear(32, 64)          # weights 2048 elements -> shadowed; biases are not
    model = nn.ModuleList([lin1, lin2])
    sh = mixed.ShadowParams(model, torch.bfloat16)
    try:
        assert len(sh.shadows) == 2 and sh.flat.dtype == torch.bfloat16
        assert mixed.weight(lin1.weight, torch.bfloat16) is sh.shadows[0]
        assert mixed.weight(lin1.bias, torch.bfloat16).dtype == torch.bfloat16      # plain cast
        red = ddp.BucketedGradReducer(model.parameters(), bucket_bytes=4096, leaf_of=sh.leaf_of)
        x = torch.randn(5, 64)
        for _ in range(2):
            red.zero_grad()
            h = F.linear(x.bfloat16(), mixed.weight(lin1.weight, torch.bfloat16), mixed.weight(lin1.bias, torch.bfloat16))
            y = F.linear(h, mixed.weight(lin2.weight, torch.bfloat16), mixed.weight(lin2.bias, torch.bfloat16))
            y.float().square().mean().backward()
            red.finish()
            # reference gradient in fp32 on the same (bf16-rounded) weights
            w1, w2 = sh.shadows[0].detach().float().requires_grad_(True), sh.shadows[1].detach().float().requires_grad_(True)
            yr = F.linear(F.linear(x.bfloat16().float(), w1, lin1.bias.detach()), w2, lin2.bias.detach())
            yr.square().mean().backward()
            for p, r in ((lin1.weight, w1), (lin2.weight, w2)):
                assert p.grad.dtype == torch.float32
                np.testing.assert_allclose(p.grad.numpy(), r.grad.numpy(), atol=0.05 * float(r.grad.abs().max()) + 1e-4)
            assert lin1.bias.grad is not None and lin1.bias.grad.dtype == torch.float32
            assert all(s.grad is None for s in sh.shadows)
        with torch.no_grad():
            lin1.weight.add_(1.0)
        sh.refresh()
        np.testing.assert_allclose(sh.shadows[0].detach().float().numpy(), lin1.weight.detach().bfloat16().float().numpy())
    finally:
        sh.release()


def test_khwc_resident_conv_weights_layout_follows_through():
    """mixed.khwc_resident_: eligible 3x3 conv weights keep their (Cout,Cin,3,3) shape with channels-last strides; the bf16
    shadow, the reducer's bucket view and (by stride equality) the optimizer state share that memory layout, values
    unchanged; other parameters stay contiguous; gradients through plain autograd land in the bucket correctly."""
    import torch.nn as nn
    import torch.nn.functional as F
    from swin_transformer_object_detection_amd import ddp, mixed
    torch.manual_seed(0)
    c3, c1, odd = nn.Conv2d(64, 64, 3, padding=1), nn.Conv2d(64, 16, 1), nn.Conv2d(24, 8, 3, padding=1)     # odd: Cin % 64 != 0
    model = nn.ModuleList([c3, c1, odd])
    w0 = c3.weight.detach().clone()
    assert mixed.khwc_resident_(model) == 1
    assert tuple(c3.weight.shape) == (64, 64, 3, 3) and mixed.is_khwc(c3.weight) and torch.equal(c3.weight.detach(), w0)
    assert c1.weight.is_contiguous() and odd.weight.is_contiguous()
    assert c3.weight.permute(0, 2, 3, 1).is_contiguous()
    sh = mixed.ShadowParams(model, torch.bfloat16)
    try:
        s3 = mixed.shadow_of(c3.weight)
        assert s3.stride() == c3.weight.stride() and torch.equal(s3.detach(), w0.bfloat16())
        red = ddp.BucketedGradReducer(model.parameters(), bucket_bytes=1 << 20, leaf_of=sh.leaf_of)
        assert c3.weight.grad.stride() == c3.weight.stride() and c1.weight.grad.is_contiguous()
        x = torch.randn(2, 64, 5, 6)
        red.zero_grad()
        y = F.conv2d(x.bfloat16(), mixed.weight(c3.weight, torch.bfloat16), None, padding=1)
        y.float().square().mean().backward()
        red.finish()
        wr = w0.bfloat16().float().requires_grad_(True)
        F.conv2d(x.bfloat16().float(), wr, None, padding=1).square().mean().backward()
        np.testing.assert_allclose(c3.weight.grad.numpy(), wr.grad.numpy(), atol=0.05 * float(wr.grad.abs().max()) + 1e-4)
        # the memory of the bucket view is (Cout,3,3,Cin): what the weight-gradient kernel writes
        b0 = red.buckets[0]
        i3 = [i for i, q in enumerate(b0['params']) if q is c3.weight][0]
        off = sum(q.numel() for q in b0['params'][:i3])
        np.testing.assert_array_equal(b0['flat'][off:off + c3.weight.numel()].numpy(),
                                      c3.weight.grad.permute(0, 2, 3, 1).contiguous().view(-1).numpy())
        red.release()
    finally:
        sh.release()


def test_sample_static_counts_and_order():
    """Fixed-size sampler: same counts as RandomSampler (random_sampler.py:31-78); positives first, then negatives,
    then invalid padding; never picks ignored (-1) entries."""
    from swin_transformer_object_detection_amd.detector import sample_static
    torch.manual_seed(0)
    for n_pos, n_neg, n_ign, num, frac in [(10, 1000, 50, 256, 0.5), (300, 1000, 0, 256, 0.5), (5, 20, 3, 512, 0.25),
                                           (0, 40, 0, 64, 0.25), (200, 30, 0, 256, 0.5)]:
        a = torch.cat([torch.randint(1, 5, (n_pos,)), torch.zeros(n_neg, dtype=torch.long), -torch.ones(n_ign, dtype=torch.long)])
        a = a[torch.randperm(a.numel())]
        idx, is_pos, valid = sample_static(a, num, frac)
        k = min(num, a.numel())
        assert idx.shape == (k,)
        exp_pos = min(n_pos, int(num * frac))
        exp_neg = min(n_neg, num - exp_pos)
        assert int(is_pos.sum()) == exp_pos and int(valid.sum()) == exp_pos + exp_neg
        assert bool((a[idx[is_pos]] > 0).all()) and bool((a[idx[valid & ~is_pos]] == 0).all())
        assert bool(is_pos[:exp_pos].all()) and bool(valid[:exp_pos + exp_neg].all())
        assert idx[valid].unique().numel() == exp_pos + exp_neg


# ---- Swin-aware checkpoint I/O (mmcv_custom/checkpoint.py:286-356, mmcv_custom/runner/checkpoint.py:19-85) ----
def _tiny_swin(**kw):
    cfg = dict(type='SwinTransformer', embed_dim=32, depths=[1, 1], num_heads=[1, 2], window_size=7, out_indices=(0, 1), **kw)
    return registry.build_backbone(cfg)


def test_checkpoint_round_trip_and_key_surgery(tmp_path):
    from swin_transformer_object_detection_amd import checkpoint
    torch.manual_seed(0)
    src = _tiny_swin()
    opt = torch.optim.AdamW(src.parameters(), lr=1e-3)
    f = str(tmp_path / "epoch_1.pth")
    checkpoint.save_checkpoint(src, f, optimizer=opt, meta=dict(epoch=1, iter=10))
    raw = torch.load(f, weights_only=False)
    assert set(raw) == {'meta', 'state_dict', 'optimizer'} and raw['meta']['epoch'] == 1          # runner/checkpoint.py:49-85
    assert all(v.device.type == 'cpu' for v in raw['state_dict'].values())
    dst = _tiny_swin()
    ck = checkpoint.load_checkpoint(dst, f)
    assert ck['meta']['iter'] == 10
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # DDP 'module.' prefix (:319-320), MoBY online branch 'encoder.' (:323-324), bare {'model': ...} files (:312-313)
    sd = src.state_dict()
    for wrapped in ({'state_dict': {'module.' + k: v for k, v in sd.items()}},
                    {'model': {**{'encoder.' + k: v for k, v in sd.items()}, **{'projector.w': torch.zeros(2)}}},
                    dict(sd)):
        g = str(tmp_path / "w.pth")
        torch.save(wrapped, g)
        dst = _tiny_swin()
        checkpoint.load_checkpoint(dst, g)
        assert all(torch.equal(a, b) for a, b in zip(sd.values(), dst.state_dict().values()))
    with pytest.raises(IOError):
        checkpoint.load_checkpoint(dst, "https://example.invalid/swin_tiny.pth")
    torch.save([1, 2, 3], str(tmp_path / "bad.pth"))
    with pytest.raises(RuntimeError, match="No state_dict"):
        checkpoint.load_checkpoint(dst, str(tmp_path / "bad.pth"))
    # init_weights(pretrained=<path>) goes through the same loader (swin_transformer.py:590-594); wrong type raises (:597-598)
    dst = _tiny_swin()
    dst.init_weights(pretrained=f)
    assert all(torch.equal(a, b) for a, b in zip(sd.values(), dst.state_dict().values()))
    with pytest.raises(TypeError, match="pretrained must be a str or None"):
        dst.init_weights(pretrained=123)


def test_checkpoint_bias_table_resize_and_ape(tmp_path):
    """relative_position_bias_table from another window size is resized bicubically per head (:337-352); a table with another
    head count is skipped, not fatal (:342-343 + non-strict load); a (1, L, C) absolute_pos_embed is reshaped (:327-335)."""
    import torch.nn.functional as F
    from swin_transformer_object_detection_amd import checkpoint
    torch.manual_seed(1)
    dst = _tiny_swin(ape=True, pretrain_img_size=56)
    own = dst.state_dict()
    sd = {k: v.clone() for k, v in own.items()}
    k0, k1 = 'layers.0.blocks.0.attn.relative_position_bias_table', 'layers.1.blocks.0.attn.relative_position_bias_table'
    big = torch.randn(23 * 23, 1)                            # window 12 -> (2*12-1)^2 entries
    sd[k0] = big
    sd[k1] = torch.randn(169, 5)                             # wrong head count
    N2, C2, H, W = own['absolute_pos_embed'].shape
    ape = torch.randn(1, H * W, C2)
    sd['absolute_pos_embed'] = ape
    f = str(tmp_path / "pre.pth")
    torch.save({'state_dict': sd}, f)
    before1 = own[k1].clone()
    checkpoint.load_checkpoint(dst, f)
    got = dst.state_dict()
    want = F.interpolate(big.permute(1, 0).view(1, 1, 23, 23), size=(13, 13), mode='bicubic').view(1, 169).permute(1, 0)
    assert got[k0].shape == (169, 1) and torch.allclose(got[k0], want)
    assert torch.equal(got[k1], before1)
    assert torch.equal(got['absolute_pos_embed'], ape.view(N2, H, W, C2).permute(0, 3, 1, 2))
    const = torch.full((23 * 23, 1), 0.25)                   # bicubic weights sum to one: a constant table stays constant
    sd[k0] = const
    torch.save({'state_dict': sd}, f)
    checkpoint.load_checkpoint(dst, f)
    assert torch.allclose(dst.state_dict()[k0], torch.full((169, 1), 0.25), atol=1e-6)
