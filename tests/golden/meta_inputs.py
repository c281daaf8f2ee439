"""Seeded DCASE metadata files shared by the label-extraction golden generator and the tests (no reference code involved)."""
import numpy as np


def meta_rows(seed, num_frames=60, num_classes=5):
    """Rows (frame, class, track, azimuth, elevation): 0-4 events per frame, up to four of the same class, integer angles."""
    rng = np.random.default_rng(seed)
    rows = []
    for f in range(num_frames):
        n = int(rng.choice([0, 1, 1, 2, 3, 4]))
        if f == num_frames - 1:
            n = max(n, 1)                                    # the last frame defines the length of the recording
        cls = rng.integers(0, num_classes, n) if rng.random() < 0.6 else np.full(n, rng.integers(0, num_classes))
        for k in range(n):
            rows.append([f, int(cls[k]), k, int(rng.integers(-180, 180)), int(rng.integers(-90, 90))])
    return rows


def write_meta(path, rows):
    with open(path, 'w') as f:
        for r in rows:
            f.write(','.join(str(v) for v in r) + '\n')
