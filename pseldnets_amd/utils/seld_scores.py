"""SELD scores (location-sensitive detection ER / F at a DOA threshold, class-sensitive localisation LE / LR, and their mean)
on DCASE-format dictionaries `{frame: [[class, azimuth_deg, elevation_deg], ...]}` — the quality metric of the reference's
validation loop (`utils/SELD_metrics.py:20-233` fed through `utils/data_utilities.py:107-151 to_metrics_format`,
`models/components/model_module.py:243-262`). Host-side numpy (as in the reference: the metric walks variable-length event
lists per one-second segment and class and matches same-class events with the Hungarian algorithm); the device hands over the
dictionaries of `pseldnets_amd.inference`. Written from the metric's definition; pinned to the reference class on seeded
random inputs (tests/golden/metrics.npz).
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

_EPS = np.finfo(float).eps


def _great_circle_deg(az1, el1, az2, el2):
    d = np.sin(el1) * np.sin(el2) + np.cos(el1) * np.cos(el2) * np.cos(np.abs(az1 - az2))
    return np.arccos(np.clip(d, -1, 1)) * 180 / np.pi


def _segments(events, num_frames, frames_per_segment):
    """[{class: {frame_in_segment: [[az, el], ...]}}] per one-second segment, frames ascending, events in listing order."""
    n_seg = int(np.ceil(num_frames / float(frames_per_segment)))
    segs = [dict() for _ in range(n_seg)]
    for s in range(n_seg):
        for frame in range(s * frames_per_segment, (s + 1) * frames_per_segment):
            for ev in events.get(frame, ()):
                segs[s].setdefault(ev[0], {}).setdefault(frame - s * frames_per_segment, []).append(list(ev[1:]))
    return segs


class SeldScores:
    def __init__(self, doa_threshold=20, nb_classes=13, label_resolution=0.1):
        self.nb_classes, self.threshold = nb_classes, doa_threshold
        self.frames_per_segment = int(1 / label_resolution)
        self.reset()

    def reset(self):
        z = lambda: np.zeros(self.nb_classes)
        self.TP, self.FP, self.FP_spatial, self.FN, self.Nref = z(), z(), z(), z(), z()
        self.total_DE, self.DE_TP, self.DE_FP, self.DE_FN = z(), z(), z(), z()
        self.S = self.D = self.I = 0

    def update(self, pred, gt, num_frames):
        """pred / gt: DCASE dictionaries of ONE recording in degrees; num_frames: its number of label frames."""
        ps = _segments(pred, num_frames, self.frames_per_segment)
        gs = _segments(gt, num_frames, self.frames_per_segment)
        for seg_p, seg_g in zip(ps, gs):
            loc_fn = loc_fp = 0
            for c in range(self.nb_classes):
                g, p = seg_g.get(c), seg_p.get(c)
                n_g = max(len(v) for v in g.values()) if g else None
                n_p = max(len(v) for v in p.values()) if p else None
                if g:
                    self.Nref[c] += n_g
                if g and p:
                    per_track = {}                                   # reference track (its index inside the frame) -> distances
                    for frame, g_doas in g.items():
                        if frame not in p:
                            continue
                        ga, pa = np.array(g_doas) * np.pi / 180., np.array(p[frame]) * np.pi / 180.
                        cost = _great_circle_deg(ga[:, None, 0], ga[:, None, 1], pa[None, :, 0], pa[None, :, 1])
                        rows, cols = linear_sum_assignment(cost)
                        for r_, c_ in zip(rows, cols):
                            per_track.setdefault(r_, []).append(cost[r_, c_])
                    if not per_track:                                 # no frame in common
                        loc_fn += n_p; self.FN[c] += n_p; self.DE_FN[c] += n_p
                    else:
                        for dists in per_track.values():
                            avg = sum(dists) / len(dists)
                            self.total_DE[c] += avg; self.DE_TP[c] += 1
                            if avg <= self.threshold:
                                self.TP[c] += 1
                            else:
                                loc_fp += 1; self.FP_spatial[c] += 1
                        if n_p > n_g:
                            loc_fp += n_p - n_g; self.FP[c] += n_p - n_g; self.DE_FP[c] += n_p - n_g
                        elif n_p < n_g:
                            loc_fn += n_g - n_p; self.FN[c] += n_g - n_p; self.DE_FN[c] += n_g - n_p
                elif g:
                    loc_fn += n_g; self.FN[c] += n_g; self.DE_FN[c] += n_g
                elif p:
                    loc_fp += n_p; self.FP[c] += n_p; self.DE_FP[c] += n_p
            self.S += min(loc_fp, loc_fn)
            self.D += max(0, loc_fn - loc_fp)
            self.I += max(0, loc_fp - loc_fn)

    def compute(self, average='macro'):
        """{'ER','F','LE','LR','SELD_scr'} — micro: pooled counts; macro: mean over the classes that were localised at least once."""
        ER = (self.S + self.D + self.I) / (self.Nref.sum() + _EPS)
        mix = lambda er, f, le, lr: np.mean([er, 1 - f, le / 180, 1 - lr], 0)
        if average == 'micro':
            F = self.TP.sum() / (_EPS + self.TP.sum() + self.FP_spatial.sum() + 0.5 * (self.FP.sum() + self.FN.sum()))
            LE = self.total_DE.sum() / float(self.DE_TP.sum() + _EPS) if self.DE_TP.sum() else 180
            LR = self.DE_TP.sum() / (_EPS + self.DE_TP.sum() + self.DE_FN.sum())
            score = mix(ER, F, LE, LR)
        elif average == 'macro':
            F = self.TP / (_EPS + self.TP + self.FP_spatial + 0.5 * (self.FP + self.FN))
            LE = self.total_DE / (self.DE_TP + _EPS)
            LE[self.DE_TP == 0] = 180.0
            LR = self.DE_TP / (_EPS + self.DE_TP + self.DE_FN)
            score = mix(np.repeat(ER, self.nb_classes), F, LE, LR)
            keep = LE != 180.0
            F, LE, LR, score = F[keep], LE[keep], LR[keep], score[keep]
            F = F.mean() if keep.any() else -1.
            LE = LE.mean() if keep.any() else 180.0
            LR = LR.mean() if keep.any() else -1.
            score = score.mean() if keep.any() else 1.0
        else:
            raise ValueError(average)
        return {'ER': float(ER), 'F': float(F), 'LE': float(LE), 'LR': float(LR), 'SELD_scr': float(score)}
