"""Where the device-to-device copy nodes of one training step come from: torch.profiler with python stacks, aten::copy_ / clone /
contiguous / fill_ / zero_ calls grouped by the innermost frame under pseldnets_amd.  python tools/copy_trace.py [--chunks 192]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pseldnets_amd.models import multi_accdoa  # noqa: E402
from pseldnets_amd.trainer import FusedTrainer  # noqa: E402
from pseldnets_amd.utils.config import get_afextractor  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--chunks', type=int, default=192)
    ap.add_argument('--group', action='store_true', help='train through a one-rank RCCL group with sync-BN (the data-parallel code path)')
    args = ap.parse_args()
    device = torch.device('cuda:0')
    group = None
    if args.group:
        import torch.distributed as dist
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29878', rank=0, world_size=1, device_id=device)
        group = dist.group.WORLD
    cfg = bench.make_cfg()
    torch.manual_seed(2024)
    net = multi_accdoa.HTSAT(cfg, bench.CLASSES, 7, pretrained_path=None)
    net.compute_dtype = torch.bfloat16
    net.to(device)
    wave, target = bench.synthetic_batch(32, device, 2024, chunks=args.chunks)
    trainer = FusedTrainer(net, get_afextractor(cfg).to(device), 'adpit', lr=1e-4, max_norm=1.0, process_group=group, sync_bn=group is not None,
                           use_graph=False)
    step = lambda: trainer.training_step(wave, target, next_x=wave)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    names = ('c10d::allreduce_', 'aten::empty', 'aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::fill_', 'aten::zero_', 'aten::_to_copy', 'aten::cat', 'aten::index_copy_')
    by = collections.Counter()
    for ev in prof.events():
        if ev.name in names:
            frame = next((f for f in ev.stack if 'pseldnets_amd' in f or 'bench.py' in f), ev.stack[0] if ev.stack else '?')
            by[(ev.name, frame.strip()[-110:])] += 1
    for (n, f), c in by.most_common(60):
        print(f'{c:5d}  {n:18s} {f}')
    gpu = collections.Counter(ev.name[:60] for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA and ('emcpy' in ev.name or 'emset' in ev.name or 'copyBuffer' in ev.name or 'fillBuffer' in ev.name))
    print(gpu.most_common(10))


if __name__ == '__main__':
    main()
