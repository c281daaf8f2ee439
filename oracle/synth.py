"""Deterministic (RNG-free) synthetic inputs shared by the golden generator and the parity tests
(test infrastructure). Shapes follow the tensor contract of SURVEY.md §8b/§8d."""
import math

import torch


def formula_wave(B, C, L, sr=24000.0):
    """A few chirps/tones + a deterministic pseudo-noise term; amplitude ~0.1 like the bench's N(0, 0.1)."""
    t = torch.arange(L, dtype=torch.float64) / sr
    out = torch.zeros(B, C, L, dtype=torch.float64)
    for b in range(B):
        for c in range(C):
            f0 = 180.0 * (1 + c) + 37.0 * b
            x = 0.05 * torch.sin(2 * math.pi * (f0 * t + 900.0 * (1 + 0.3 * c) * t * t))
            x += 0.03 * torch.sin(2 * math.pi * (2500.0 + 410.0 * c) * t + 0.7 * b)
            n = torch.arange(L, dtype=torch.float64)
            x += 0.04 * torch.sin(n * (1.2345 + 0.11 * c) + 0.5 * torch.sin(n * 0.01 * (b + 1)) * 40.0)
            out[b, c] = x
    return out.to(torch.float32)


def _unit(v):
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-9)


def formula_adpit_label(B, T=100, C=3):
    """[B, T, 6, 4, C]: sparse activity with every ADPIT case present (A0 only; B0+B1; C0+C1+C2)."""
    lab = torch.zeros(B, T, 6, 4, C)
    for b in range(B):
        for t in range(T):
            for c in range(C):
                r = (7 * b + 3 * t + 5 * c) % 11
                if r == 0:
                    tracks = [0]
                elif r == 1:
                    tracks = [1, 2]
                elif r == 2 and t % 2 == 0:
                    tracks = [3, 4, 5]
                else:
                    continue
                for k in tracks:
                    v = _unit(torch.tensor([math.sin(0.3 * t + k + b), math.cos(0.2 * t + c + 0.5 * k), math.sin(0.1 * t * (k + 1)) + 0.2]))
                    lab[b, t, k, 0, c] = 1.0
                    lab[b, t, k, 1:, c] = v
    return lab


def formula_accdoa_label(B, T=100, C=3):
    """[B, T, 3*C] activity-coupled Cartesian DOA (layout xyz-major like the reference's accdoa label)."""
    lab = torch.zeros(B, T, 3, C)
    for b in range(B):
        for t in range(T):
            for c in range(C):
                if (5 * b + 2 * t + 3 * c) % 7 == 0:
                    lab[b, t, :, c] = _unit(torch.tensor([math.sin(0.3 * t + b), math.cos(0.2 * t + c), 0.3 + math.sin(0.05 * t)]))
    return lab.reshape(B, T, 3 * C)


def formula_einv2_label(B, T=100, C=3):
    """sed_label [B, T, 3, C] one-hot per active track, doa_label [B, T, 3, 3]."""
    sed = torch.zeros(B, T, 3, C)
    doa = torch.zeros(B, T, 3, 3)
    for b in range(B):
        for t in range(T):
            n_act = (b + t // 5) % 4          # 0..3 simultaneously active tracks
            for k in range(min(n_act, 3)):
                c = (t + 2 * k + b) % C
                sed[b, t, k, c] = 1.0
                doa[b, t, k] = _unit(torch.tensor([math.sin(0.3 * t + k), math.cos(0.2 * t + b + k), 0.1 + math.sin(0.07 * t * (k + 1))]))
    return sed, doa


def formula_pred(shape, phase=0.0, scale=0.7):
    n = 1
    for d in shape:
        n *= d
    i = torch.arange(n, dtype=torch.float64)
    v = scale * (torch.sin(i * 0.618 + phase) * 0.7 + 0.3 * torch.cos(i * 1.7 + 2 * phase))
    return v.reshape(shape).to(torch.float32)
