"""Probe for DESIGN.md's 'two half-batches on two streams': two INDEPENDENT half-size training steps (96 chunks each, own
network) enqueued on two streams, against one 192-chunk step. Not a product path (two networks): it only measures what the
concurrency of two dependent chains returns on this device.  python tools/two_chains_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pseldnets_amd.models import multi_accdoa
from pseldnets_amd.trainer import FusedTrainer
from pseldnets_amd.utils.config import get_afextractor
dev = torch.device('cuda:0')


class A(dict):
    __getattr__ = dict.__getitem__


cfg = A(data=A(nfft=1024, hoplen=240, window='hann', n_mels=64, sample_rate=24000, audio_feature='logmelIV'), adapt=A())


def make(chunks, seed):
    net = multi_accdoa.HTSAT(cfg, bench.CLASSES, 7, pretrained_path=None, embed_dim=96, depths=[2, 2, 6, 2], num_heads=[4, 8, 16, 32],
                             drop_path_rate=0.1)
    net.compute_dtype = torch.bfloat16
    net = net.to(dev)
    tr = FusedTrainer(net, get_afextractor({'data': dict(cfg.data)}).to(dev), 'adpit', lr=1e-4, max_norm=1.0)
    wave, target = bench.synthetic_batch(32, dev, seed, chunks=chunks)
    return tr, wave, target


def run(trainers, steps=20, warmup=5):
    streams = [torch.cuda.Stream() for _ in trainers] if len(trainers) > 1 else [torch.cuda.current_stream()]
    def one():
        for (tr, w, t), st in zip(trainers, streams):
            with torch.cuda.stream(st):
                tr.training_step(w, t, next_x=w)
    for _ in range(warmup): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


full = [make(192, 1)]
ms1 = run(full)
del full
torch.cuda.empty_cache()
halves = [make(96, 1), make(96, 2)]
ms2 = run(halves)
print(f"one 192-chunk step: {ms1:.2f} ms; two concurrent 96-chunk steps: {ms2:.2f} ms  ({ms1 / ms2:.3f}x)")
one_half = run(halves[:1])
print(f"one 96-chunk step alone: {one_half:.2f} ms (2x = {2 * one_half:.2f})")
