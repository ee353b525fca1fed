"""Synthetic COCO-shaped training inputs: seeded images, 8 GT boxes / image drawn the way
the reference's ``_demo_mm_inputs`` does (tests/test_models/test_forward.py:369-376: centre and size
~ U(0,1), corners clipped to the image), labels ~ randint(0, 80), box-filled rectangle masks."""
import numpy as np
import torch


def synthetic_batch(batch, height, width, device, seed=0, num_boxes=8, num_classes=80, min_size=8.0):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(batch, 3, height, width, generator=g).to(device)
    rng = np.random.RandomState(seed)
    gt_bboxes, gt_labels, gt_masks, metas = [], [], [], []
    for _ in range(batch):
        cx, cy, bw, bh = rng.rand(num_boxes, 4).T
        tl_x = ((cx * width) - (width * bw / 2)).clip(0, width)
        tl_y = ((cy * height) - (height * bh / 2)).clip(0, height)
        br_x = ((cx * width) + (width * bw / 2)).clip(0, width)
        br_y = ((cy * height) + (height * bh / 2)).clip(0, height)
        # degenerate boxes make log(gw/pw) = -inf in the box coder: keep at least `min_size` pixels
        br_x = np.maximum(br_x, np.minimum(tl_x + min_size, width)); tl_x = np.minimum(tl_x, br_x - min_size)
        br_y = np.maximum(br_y, np.minimum(tl_y + min_size, height)); tl_y = np.minimum(tl_y, br_y - min_size)
        boxes = np.stack([tl_x, tl_y, br_x, br_y], 1).astype(np.float32)
        labels = rng.randint(0, num_classes, num_boxes).astype(np.int64)
        masks = np.zeros((num_boxes, height, width), np.uint8)
        for k, (x1, y1, x2, y2) in enumerate(boxes):
            masks[k, int(y1):max(int(np.ceil(y2)), int(y1) + 1), int(x1):max(int(np.ceil(x2)), int(x1) + 1)] = 1
        gt_bboxes.append(torch.from_numpy(boxes).to(device))
        gt_labels.append(torch.from_numpy(labels).to(device))
        gt_masks.append(torch.from_numpy(masks).to(device))
        metas.append(dict(img_shape=(height, width, 3), pad_shape=(height, width, 3), ori_shape=(height, width, 3),
                          scale_factor=1.0, flip=False))
    return dict(img=img, img_metas=metas, gt_bboxes=gt_bboxes, gt_labels=gt_labels, gt_masks=gt_masks)
