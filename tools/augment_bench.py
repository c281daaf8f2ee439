"""Cost of the AugMix front end at the bench batch (192 ten-second chunks -> 576 after data_copy): waveform augmentations,
feature extraction of the tripled batch, feature augmentations.  python tools/augment_bench.py"""
import os, sys, random, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd.models.model_module import SELDModelModule
from pseldnets_amd.train import SyntheticDataset, compose, synthetic_batch
dev = torch.device('cuda:0')
torch.manual_seed(0); np.random.seed(0); random.seed(0)
cfg = compose(['experiment=synth_maccdoa', 'augment=augmix', 'model.batch_size=192'])
m = SELDModelModule(cfg, SyntheticDataset(cfg))
m.af_extractor.to(dev)
gen = torch.Generator(device=dev).manual_seed(1)
batch = synthetic_batch(cfg, cfg.model.method, dev, gen)
target = {k: v for k, v in batch.items() if 'data' not in k}
for _ in range(2):
    m.augment_step(batch['data'], target)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m.augment_step(batch['data'], target)
torch.cuda.synchronize()
print(f"augment_step (AugMix, 192 -> 576 chunks, incl. feature extraction of 576 chunks): {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
t0 = time.perf_counter()
for _ in range(5):
    m.standardize(batch['data'])
torch.cuda.synchronize()
print(f"feature extraction of 192 chunks alone: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
