"""Seeded inputs shared by the epoch-end golden generator and the tests (no reference code involved)."""
from collections import OrderedDict

import numpy as np
import torch

C = 5
PATHS = OrderedDict([('foa/room1/mix_a.flac', 330), ('foa/room1/mix_b.flac', 100), ('foa/room2/mix_c.flac', 215)])   # label frames


def epoch_inputs(method):
    """(list of per-step prediction dicts as the test loop appends them, paths_dict, ground-truth DCASE dictionaries).
    One test chunk = 100 prediction frames; recording r owns ceil(frames_r / 100) consecutive chunks; steps of 3 chunks."""
    n_chunks = sum(int(np.ceil(v / 100)) for v in PATHS.values())
    g = torch.Generator().manual_seed(11)
    if method == 'multi_accdoa':
        full = {'multi_accdoa': torch.randn(n_chunks, 100, 9 * C, generator=g) * 0.6}
    elif method == 'accdoa':
        full = {'accdoa': torch.randn(n_chunks, 100, 3 * C, generator=g) * 0.6}
    else:
        full = {'sed': torch.randn(n_chunks, 100, 3, C, generator=g) * 2.0, 'doa': torch.randn(n_chunks, 100, 3, 3, generator=g)}
    steps = [{k: v[i:i + 3].clone() for k, v in full.items()} for i in range(0, n_chunks, 3)]
    rng = np.random.default_rng(12)
    gts = OrderedDict()
    for path, frames in PATHS.items():
        d = {}
        for f in range(frames):
            for c in range(C):
                if rng.random() < 0.15:
                    d.setdefault(f, []).append([c, float(rng.integers(-180, 180)), float(rng.integers(-80, 80))])
        gts[path] = d
    return steps, PATHS, gts
