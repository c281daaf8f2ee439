"""EINV2 networks on MI355X — mirror of the reference's `models/einv2.py` registry module:
HTSAT (:189-327, dual SED/DOA Swin encoders coupled by CrossStitch before every stage) and HTSAT_SEDDOA (:329-442,
one encoder, two heads). Outputs {'sed': f32[B,100,3,C] (logits), 'doa': f32[B,100,3,3] (tanh)}; state-dict keys
are the reference's. CRNN / ConvConformer / PASST variants are not built on this path."""
from .. import ops
from . import accdoa
from .components.htsat import SwinEncoder, TscamHead
from .components.seld_net import HTSATNetBase


class HTSAT(HTSATNetBase):
    def __init__(self, cfg, num_classes, in_channels=7, audioset_pretrain=True,
                 pretrained_path='ckpts/HTSAT-fullset-imagenet-768d-32000hz.ckpt', **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.sed_in_channels, self.doa_in_channels = 4, in_channels
        self._init_common(cfg, in_channels)
        self.sed_enc = SwinEncoder(self.arena, 'sed_encoder.', self.sed_in_channels, mel_bins=self.mel_bins, **kwargs)
        self.doa_enc = SwinEncoder(self.arena, 'doa_encoder.', self.doa_in_channels, mel_bins=self.mel_bins, **kwargs)
        E, nl = self.sed_enc.E, self.sed_enc.nl
        for li in range(nl):
            self.arena.add(f'stitch1.{li}.weight', (E * 2 ** li, 2, 2))
        self.sed_head = TscamHead(self.arena, 'sed_tscam_conv.', self.sed_enc.num_features, num_classes * 3, False)
        self.doa_head = TscamHead(self.arena, 'doa_tscam_conv.', self.doa_enc.num_features, 9, True)
        self._finish_init()
        if pretrained_path:
            raise NotImplementedError("checkpoint adaptation for EINV2 (einv2.py:239-272) is not mirrored yet: "
                                      "pass pretrained_path=None and load a state dict")

    def _forward_impl(self, x, training):
        B, dt = x.shape[0], self.compute_dtype
        a = self.arena
        mean_rstd, scale_shift = self._bn_front(x, training)
        drop_s = self._drop_scales(B, self.sed_enc, x.device, training)
        drop_d = self._drop_scales(B, self.doa_enc, x.device, training)
        xs, sp_s = self.sed_enc.forward_patch(x, scale_shift, dt, 0)
        xd, sp_d = self.doa_enc.forward_patch(x, scale_shift, dt, 0)
        layers = []
        for li in range(self.sed_enc.nl):
            xs_in, xd_in = xs, xd
            xs, xd = ops.cross_stitch_fwd(xs_in, xd_in, a.p(f'stitch1.{li}.weight').view(-1, 4))
            xs, ss = self.sed_enc.forward_layer(li, xs, B, drop_s)
            xd, sd = self.doa_enc.forward_layer(li, xd, B, drop_d)
            layers.append(dict(xs_in=xs_in, xd_in=xd_in, sed=ss, doa=sd))
        xs, fs = self.sed_enc.forward_final(xs)
        xd, fd = self.doa_enc.forward_final(xd)
        ys, hs = self.sed_head.forward(xs, B)
        yd, hd = self.doa_head.forward(xd, B)
        saved = dict(feat=x, mean_rstd=mean_rstd, sp_s=sp_s, sp_d=sp_d, layers=layers, fs=fs, fd=fd, hs=hs, hd=hd, B=B)
        return (ys.view(B, 100, 3, -1), yd.view(B, 100, 3, 3)), saved

    def _backward_impl(self, saved, douts, on_range_done=None):
        dsed, ddoa = douts
        B, dt, a = saved['B'], self.compute_dtype, self.arena
        dxs = self.sed_enc.backward_final(self.sed_head.backward(dsed.reshape(B, 100, -1), saved['hs'], B, dt), saved['fs'])
        dxd = self.doa_enc.backward_final(self.doa_head.backward(ddoa.reshape(B, 100, -1), saved['hd'], B, dt), saved['fd'])
        for li in reversed(range(self.sed_enc.nl)):
            s = saved['layers'][li]
            dxs = self.sed_enc.backward_layer(li, dxs, s['sed'], B)
            dxd = self.doa_enc.backward_layer(li, dxd, s['doa'], B)
            dxs, dxd = ops.cross_stitch_bwd(s['xs_in'], s['xd_in'], a.p(f'stitch1.{li}.weight').view(-1, 4), dxs, dxd,
                                            a.g(f'stitch1.{li}.weight').view(-1, 4))
        dw, db = self._bn_grads()
        self.doa_enc.backward_patch(dxd, saved['sp_d'], saved['feat'], saved['mean_rstd'], dw, db, accumulate_bn=False)
        self.sed_enc.backward_patch(dxs, saved['sp_s'], saved['feat'], saved['mean_rstd'], dw, db, accumulate_bn=True)
        if on_range_done is not None:
            on_range_done(0, a.size)

    def forward(self, x):
        sed, doa = self._run(x)
        return {'sed': sed, 'doa': doa}


class HTSAT_SEDDOA(HTSATNetBase):
    def __init__(self, cfg, num_classes, in_channels=7, audioset_pretrain=True,
                 pretrained_path='ckpts/HTSAT-fullset-imagenet-768d-32000hz.ckpt', **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self._init_common(cfg, in_channels)
        self.enc = SwinEncoder(self.arena, 'encoder.', in_channels, mel_bins=self.mel_bins, **kwargs)
        self.sed_head = TscamHead(self.arena, 'sed_tscam_conv.', self.enc.num_features, num_classes * 3, False)
        self.doa_head = TscamHead(self.arena, 'doa_tscam_conv.', self.enc.num_features, 9, True)
        self._finish_init()
        if pretrained_path:
            raise NotImplementedError("pass pretrained_path=None and load a state dict")

    def _forward_impl(self, x, training):
        B, dt = x.shape[0], self.compute_dtype
        mean_rstd, scale_shift = self._bn_front(x, training)
        drop = self._drop_scales(B, self.enc, x.device, training)
        tok, sp = self.enc.forward_patch(x, scale_shift, dt)
        layers = []
        for li in range(self.enc.nl):
            tok, s = self.enc.forward_layer(li, tok, B, drop)
            layers.append(s)
        xn, fin = self.enc.forward_final(tok)
        ys, hs = self.sed_head.forward(xn, B)
        yd, hd = self.doa_head.forward(xn, B)
        return (ys.view(B, 100, 3, -1), yd.view(B, 100, 3, 3)), dict(feat=x, mean_rstd=mean_rstd, patch=sp, layers=layers,
                                                                  fin=fin, hs=hs, hd=hd, B=B)

    def _backward_impl(self, saved, douts, on_range_done=None):
        dsed, ddoa = douts
        B, dt = saved['B'], self.compute_dtype
        dxn = self.sed_head.backward(dsed.reshape(B, 100, -1), saved['hs'], B, dt)
        dxn2 = self.doa_head.backward(ddoa.reshape(B, 100, -1), saved['hd'], B, dt)
        dxn = ops.add(dxn, dxn2)  # two consumers of the final tokens
        dx = self.enc.backward_final(dxn, saved['fin'])
        for li in reversed(range(self.enc.nl)):
            dx = self.enc.backward_layer(li, dx, saved['layers'][li], B)
        dw, db = self._bn_grads()
        self.enc.backward_patch(dx, saved['patch'], saved['feat'], saved['mean_rstd'], dw, db, accumulate_bn=False)
        if on_range_done is not None:
            on_range_done(0, self.arena.size)

    def forward(self, x):
        sed, doa = self._run(x)
        return {'sed': sed, 'doa': doa}


CRNN = ConvConformer = PASST = accdoa._NotBuilt
