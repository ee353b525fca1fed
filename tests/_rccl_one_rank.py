"""Child process of test_gpu_backbone.test_reducer_over_rccl_one_rank: the reducer's N > 1 code path (launch stream,
pre-division, asynchronous all-reduce, wait in finish) over a real RCCL ("nccl") process group of ONE rank -- the
only RCCL configuration a one-GPU box can run.  The reducer is told the world has two ranks, so every bucket must come
out as exactly half of the one-process gradients.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from swin_transformer_object_detection_amd import data, ddp, detector, mixed, presets
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    cfg = presets.mask_rcnn_swin("tiny")
    cfg["backbone"]["drop_path_rate"] = 0.0
    torch.manual_seed(0)
    model = detector.build_detector(cfg, compute_dtype=torch.bfloat16).to(dev).train()
    batch = data.synthetic_batch(2, 384, 512, dev, seed=5, num_boxes=6)
    sh = mixed.ShadowParams(model, torch.bfloat16)
    red = ddp.BucketedGradReducer(model.parameters_in_forward_order(), leaf_of=sh.leaf_of, bucket_bytes=8 << 20)
    step_stream = torch.cuda.Stream(device=dev)
    step_stream.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(step_stream)

    def run(world):
        red.world = world
        outs = []
        for it in range(3):
            torch.manual_seed(100 + it)
            red.zero_grad()
            loss, _ = model.parse_losses(model.forward_train(**batch))
            red.mark_backward_start()
            loss.backward()
            red.finish()
            outs.append([b['flat'].clone() for b in red.buckets])
        torch.cuda.synchronize()
        return outs, [list(r) for r in red.timeline]

    ref, _ = run(1)
    again, _ = run(1)
    got, tl = run(2)
    worst = 0.0
    for it in range(3):
        for fr, fa, fg in zip(ref[it], again[it], got[it]):
            scale = float(fr.abs().max())
            noise = float((fr - fa).abs().max())
            err = float((fr - 2.0 * fg).abs().max())
            worst = max(worst, err / (4 * noise + 2e-3 * scale + 1e-7))
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(dict(buckets=len(red.buckets), reduced=len(tl), all_done=all(r[2] is not None for r in tl),
                          worst_over_bound=worst, loss_finite=bool(torch.isfinite(loss_probe(model, batch))))))


def loss_probe(model, batch):
    with torch.no_grad():
        loss, _ = model.parse_losses(model.forward_train(**batch))
    return loss


if __name__ == "__main__":
    main()
