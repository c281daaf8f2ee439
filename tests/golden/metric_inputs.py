"""Seeded inputs shared by the metrics golden generator and the tests (no reference code involved)."""
import numpy as np


def metric_inputs(seed, nb_classes=5, num_frames=57):
    """Random DCASE dictionaries (prediction, reference) with shared, missed, spurious and multi-instance events."""
    rng = np.random.default_rng(seed)
    gt, pred = {}, {}
    for f in range(num_frames):
        for c in range(nb_classes):
            n = int(rng.choice([0, 0, 1, 1, 2, 3]))
            if n:
                doas = [[float(rng.uniform(-180, 180)), float(rng.uniform(-80, 80))] for _ in range(n)]
                gt.setdefault(f, []).extend([[c, a, e] for a, e in doas])
                for a, e in doas:
                    u = rng.random()
                    if u < 0.6:
                        pred.setdefault(f, []).append([c, a + float(rng.normal(0, 12)), e + float(rng.normal(0, 8))])
                    elif u < 0.7:
                        pred.setdefault(f, []).append([c, a + 90.0, -e])
            if rng.random() < 0.08:
                pred.setdefault(f, []).append([c, float(rng.uniform(-180, 180)), float(rng.uniform(-80, 80))])
    return pred, gt, num_frames
