"""Seeded inputs shared by the decoding golden generator and the tests (no reference code involved)."""
import torch


def decode_inputs(seed=5):
    """Predictions [frames, 9C] with every unification case: single, two close / far, three with 0, 1 (each pair) and >= 2
    close pairs; C = 6."""
    g = torch.Generator().manual_seed(seed)
    C, Fm = 6, 48
    pred = torch.zeros(Fm, 9 * C)
    base = torch.randn(Fm, C, 3, generator=g)
    base = base / base.norm(dim=-1, keepdim=True)
    pattern = torch.randint(0, 9, (Fm, C), generator=g)
    for f in range(Fm):
        for c in range(C):
            b = base[f, c]
            far1 = torch.tensor([b[1], -b[0], b[2]]) * 0.9                  # >= 60 degrees away (well clear of the 15-degree rule)
            far2 = -b * 0.8
            near = lambda s: (b + 0.05 * torch.randn(3, generator=g)) * s   # a few degrees away
            tr = [torch.zeros(3)] * 3
            pt = int(pattern[f, c])
            if pt == 1: tr = [b * 0.9, torch.zeros(3), torch.zeros(3)]
            elif pt == 2: tr = [b * 0.9, near(0.8), torch.zeros(3)]
            elif pt == 3: tr = [b * 0.9, torch.zeros(3), far1]
            elif pt == 4: tr = [b * 0.9, far1, far2]
            elif pt == 5: tr = [b * 0.9, near(0.7), far1]
            elif pt == 6: tr = [far1, b * 0.9, near(0.7)]
            elif pt == 7: tr = [b * 0.9, far1, near(0.7)]
            elif pt == 8: tr = [b * 0.9, near(0.8), near(0.7)]
            for k in range(3):
                for a in range(3):
                    pred[f, (3 * k + a) * C + c] = tr[k][a]
    pred = pred + 0.02 * torch.randn(pred.shape, generator=g)
    acc = torch.randn(Fm, 3 * C, generator=g) * 0.45
    return pred, acc, C


def toy_forward(x, C):
    """Stand-in network of the ACS goldens: x [B, 4, L] -> {'multi_accdoa': [B, 5, 9C], 'accdoa': [B, 5, 3C]}, a fixed
    nonlinear map that mixes the four channels (so that a wrong rotation bookkeeping changes the result)."""
    toy_w = torch.linspace(-1, 1, 4 * 9 * C, device=x.device).reshape(4, 9 * C)
    B = x.shape[0]
    z = x.reshape(B, 4, 5, -1).mean(-1).transpose(1, 2)                  # [B, 5, 4]
    return {'multi_accdoa': torch.tanh(z @ toy_w + 0.3 * (z ** 2) @ toy_w.flip(0)), 'accdoa': torch.tanh(z @ toy_w[:, :3 * C])}
