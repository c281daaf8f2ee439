"""Seeded inputs shared by the augmentation golden generator and the tests (no reference code involved)."""
import torch


def aug_inputs(kind, seed=7):
    """Seeded inputs of the augmentation goldens (shared with the tests): features [6,3,40,12], waveforms [6,4,240], labels
    for 5 classes over 4 label frames (xy_ratio 10), 'ov' strings. Labels of the single-source samples use track 0 only."""
    g = torch.Generator().manual_seed(seed)
    N, C, Tn, Fq, Ty, K = 6, 3, 40, 12, 4, 5
    feat = torch.randn(N, C, Tn, Fq, generator=g)
    wave = torch.randn(N, 4, 240, generator=g)
    ov = ['1', '2', '1', '1', '2', '1']
    tgt = {'ov': list(ov)}
    if kind == 'adpit':
        lab = torch.zeros(N, Ty, 6, 4, K)
        for n in range(N):
            ntr = 1 if ov[n] == '1' else 3          # ov '2': tracks B0, B1 (same class) plus an A0 of another class
            act = (torch.rand(Ty, K, generator=g) < 0.5).float()
            xyz = torch.randn(Ty, 3, K, generator=g)
            if ntr == 1:
                lab[n, :, 0, 0] = act; lab[n, :, 0, 1:] = xyz * act[:, None]
            else:
                a0 = act.clone(); a0[:, ::2] = 0
                b = act.clone(); b[:, 1::2] = 0
                lab[n, :, 0, 0] = a0; lab[n, :, 0, 1:] = xyz * a0[:, None]
                xyz2 = torch.randn(Ty, 3, K, generator=g)
                lab[n, :, 1, 0] = b; lab[n, :, 1, 1:] = xyz * b[:, None]
                lab[n, :, 2, 0] = b; lab[n, :, 2, 1:] = xyz2 * b[:, None]
        tgt['adpit_label'] = lab
    elif kind == 'accdoa':
        tgt['accdoa_label'] = torch.randn(N, Ty, 3 * K, generator=g) * (torch.rand(N, Ty, 3 * K, generator=g) < 0.4)
    else:
        sed = (torch.rand(N, Ty, 3, K, generator=g) < 0.3).float()
        sed[:, :, 2] = 0
        for n in range(N):
            if ov[n] == '1':
                sed[n, :, 1] = 0
        tgt['sed_label'] = sed
        tgt['doa_label'] = torch.randn(N, Ty, 3, 3, generator=g) * sed.amax(-1, keepdim=True)
    return feat, wave, tgt
