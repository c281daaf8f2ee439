"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the CPU oracle."""
import torch


def run():
    from oracle import feature as of
    from oracle import htsat as oh
    from oracle import losses as ol
    from oracle import synth
    from pseldnets_amd import ops
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.utils.feature import LogmelIV_Extractor

    dev = torch.device('cuda:0')

    class A(dict):
        __getattr__ = dict.__getitem__
    cfg = A(data=A(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'), adapt=A())
    tiny = dict(embed_dim=48, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), drop_path_rate=0.0)
    wave = synth.formula_wave(2, 4, 240000)
    feat = LogmelIV_Extractor({'data': dict(cfg.data)}).to(dev)(wave.to(dev))
    feat_ref = of.logmel_iv(wave)
    # The formula wave is a handful of pure tones: most FFT bins are near-silent, so the -100 dB floor and the per-bin
    # normalised intensity vectors are dominated by fp32 round-off (the fp32 oracle itself differs from a float64
    # evaluation by ~0.03). The kernel is therefore held to the float64 evaluation with the fp32 oracle's own
    # round-off as the yardstick, and to 2e-3 relative on the log-mel POWER.
    import numpy as np
    f_hip, f_ref = feat.cpu().double(), feat_ref.double()
    f_64 = torch.from_numpy(np.asarray(of.logmel_iv_f64(wave.numpy()))).double()
    noise = (f_ref - f_64).abs().max().item()
    assert (f_hip - f_64).abs().max().item() <= 2 * noise + 1e-3, "feature kernel deviates from the float64 oracle"
    p_hip, p_64 = 10.0 ** (f_hip[:, :4] / 10), 10.0 ** (f_64[:, :4] / 10)
    assert ((p_hip - p_64).abs() <= 2e-3 * p_64 + 1e-9).all(), "feature kernel (log-mel power) deviates from the oracle"
    net = multi_accdoa.HTSAT(cfg, 3, 7, pretrained_path=None, embed_dim=48, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16],
                             drop_path_rate=0.0)
    sd = oh.formula_state('multi_accdoa', 3, 7, tiny)
    net.load_state_dict(sd, strict=False)
    net.to(dev).train()
    net._materialize(dev)
    lab = synth.formula_adpit_label(2, 100, 3)
    y, saved = net._forward_impl(feat, True)
    loss, dpred = ops.adpit_loss(y, lab.to(dev))
    net.zero_grad_arena()
    net._backward_impl(saved, (dpred,))
    net.fused_adamw_step(1e-4, max_norm=1.0)
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    out = oh.accdoa_htsat_forward(feat_ref, p, tiny, training=True, key='multi_accdoa')
    lo = ol.adpit(out, {'adpit_label': lab})['loss_all']
    err = abs(loss.item() - lo.item()) / abs(lo.item())
    assert err < 1e-3, f"loss {loss.item()} vs oracle {lo.item()}"
    print(f"smoke ok: loss {loss.item():.6f} (oracle {lo.item():.6f}), rel err {err:.2e}")
