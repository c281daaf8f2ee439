"""EINV2 networks on MI355X — mirror of the reference's `models/einv2.py` registry module:
HTSAT (:189-327, dual SED/DOA Swin encoders coupled by CrossStitch before every stage) and HTSAT_SEDDOA (:329-442,
one encoder, two heads). Outputs {'sed': f32[B,100,3,C] (logits), 'doa': f32[B,100,3,3] (tanh)}; state-dict keys
are the reference's. CRNN (:17-174: dual CNN8 / CNN12 stacks stitched after every ConvBlock, three Decoder pairs) and
PASST (:446-575: dual PaSST stitched before every ps_gap-th block) share the three-track tail `EinTracks`.
einv2.ConvConformer (:177-187) cannot be constructed in the reference (its super().__init__ call passes the arguments
of an older signature and fails on `cfg.data`), so there is no behaviour to mirror and it stays refused."""
import torch

from .. import ops
from . import accdoa
from .accdoa import StaticBufferMixin, decoder_config, make_decoder
from .components.crnn import ConvEncoder
from .components.gru import GRUDecoder
from .components.htsat import SwinEncoder, TscamHead
from .components.passt import PasstEncoder
from .components.seld_net import HTSATNetBase


class EinTracks:
    """The three-track tail shared by einv2.CRNN (:51-64,129-168) and einv2.PASST (:470-480,556-566): per track a
    (SED, DOA) Decoder pair on the shared encoder outputs, a CrossStitch, optionally the 'repeat' interpolation +
    10-frame mean of the CRNN, then Linear(feats, C) -> SED logits and Linear(feats, 3) -> tanh -> DOA.
    `names` maps a role ('sed_dec', 'doa_dec', 'stitch', 'fc_sed', 'fc_doa') to a function track -> key prefix."""

    def __init__(self, arena, feats, num_classes, decoder, n_layers, names, register_stitch=True):
        self.arena, self.feats, self.C, self.names = arena, feats, num_classes, names
        self.Cp, self.Dp = (num_classes + 7) // 8 * 8, 8
        self.sed_dec = [make_decoder(arena, names['sed_dec'](t), decoder, feats, n_layers) for t in range(3)]
        self.doa_dec = [make_decoder(arena, names['doa_dec'](t), decoder, feats, n_layers) for t in range(3)]
        for t in range(3 if register_stitch else 0):
            arena.add(names['stitch'](t) + 'weight', (feats, 2, 2))
        for t in range(3):
            arena.add(names['fc_sed'](t) + 'weight', (num_classes, feats), pad_rows=self.Cp)
            arena.add(names['fc_sed'](t) + 'bias', (num_classes,), pad_rows=self.Cp)
        for t in range(3):
            arena.add(names['fc_doa'](t) + 'weight', (3, feats), pad_rows=self.Dp)
            arena.add(names['fc_doa'](t) + 'bias', (3,), pad_rows=self.Dp)

    def decoders(self):
        return [d for d in self.sed_dec + self.doa_dec if d is not None]

    def _fc(self, prefix, x, y_t, D, act):
        a, dt = self.arena, x.dtype
        z = ops.linear_fwd(x, a.w(prefix + 'weight', dt, padded=True), a.p(prefix + 'bias', padded=True))
        ops.fc_out_fwd(z, y_t, D, act)

    def _fc_bwd(self, prefix, dy_t, y_t, x, Dp, act):
        a, dt = self.arena, x.dtype
        dz = ops.fc_out_bwd(dy_t, y_t, Dp, dt, act)
        ops.linear_wgrad(dz, x, a.g(prefix + 'weight', padded=True), dbias=a.g(prefix + 'bias', padded=True))
        return ops.linear_dgrad(dz, a.w(prefix + 'weight', dt, padded=True), wt=a.wt(prefix + 'weight', dt, padded=True))

    def forward(self, xs, xd, B, T, training, buffers, taps=None):
        """xs / xd: [B*T, feats] rows of the SED / DOA encoders. Returns (sed f32[B,T',3,C], doa f32[B,T',3,3], saved)."""
        a, n = self.arena, self.names
        rows = B * (taps['n_out'] if taps is not None else T)
        sed = torch.empty((rows, 3, self.C), dtype=torch.float32, device=xs.device)
        doa = torch.empty((rows, 3, 3), dtype=torch.float32, device=xs.device)
        saved = []
        many = None
        if isinstance(self.sed_dec[0], GRUDecoder):          # the six recurrences advance together, one launch per timestep
            many = GRUDecoder.forward_many(self.sed_dec + self.doa_dec, [xs] * 3 + [xd] * 3, B, T)
        for t in range(3):
            if many is not None:
                (ps, pd), (s_s, s_d) = (many[0][t], many[0][3 + t]), (many[1][t], many[1][3 + t])
            else:
                ps, s_s = self.sed_dec[t].forward(xs, B, T, training, buffers) if self.sed_dec[t] is not None else (xs, None)
                pd, s_d = self.doa_dec[t].forward(xd, B, T, training, buffers) if self.doa_dec[t] is not None else (xd, None)
            qs, qd = ops.cross_stitch_fwd(ps, pd, a.p(n['stitch'](t) + 'weight').view(-1, 4))
            if taps is not None:
                qs, qd = ops.rows_pool_fwd(qs, taps, B), ops.rows_pool_fwd(qd, taps, B)
            self._fc(n['fc_sed'](t), qs, sed[:, t], self.C, False)
            self._fc(n['fc_doa'](t), qd, doa[:, t], 3, True)
            saved.append(dict(ps=ps, pd=pd, s_s=s_s, s_d=s_d, qs=qs, qd=qd))
        return sed.view(B, -1, 3, self.C), doa.view(B, -1, 3, 3), dict(tracks=saved, doa=doa, taps=taps)

    def backward(self, dsed, ddoa, saved, B):
        """Returns (dxs, dxd): the gradients of the two encoder outputs, summed over the three tracks."""
        a, n = self.arena, self.names
        dsed = dsed.contiguous().float().view(-1, 3, self.C)
        ddoa = ddoa.contiguous().float().view(-1, 3, 3)
        doa, taps = saved['doa'], saved['taps']
        dxs = dxd = None
        gru = isinstance(self.sed_dec[0], GRUDecoder)
        pend_s, pend_d = [], []
        for t in range(3):
            s = saved['tracks'][t]
            dqs = self._fc_bwd(n['fc_sed'](t), dsed[:, t], None, s['qs'], self.Cp, False)
            dqd = self._fc_bwd(n['fc_doa'](t), ddoa[:, t], doa[:, t], s['qd'], self.Dp, True)
            if taps is not None:
                dqs, dqd = ops.rows_pool_bwd(dqs, taps, B), ops.rows_pool_bwd(dqd, taps, B)
            dps, dpd = ops.cross_stitch_bwd(s['ps'], s['pd'], a.p(n['stitch'](t) + 'weight').view(-1, 4), dqs, dqd,
                                            a.g(n['stitch'](t) + 'weight').view(-1, 4))
            if gru:
                pend_s.append(dps); pend_d.append(dpd)
                continue
            if self.sed_dec[t] is not None:
                dps = self.sed_dec[t].backward(dps, s['s_s'], B)
            if self.doa_dec[t] is not None:
                dpd = self.doa_dec[t].backward(dpd, s['s_d'], B)
            dxs = dps if dxs is None else ops.add(dxs, dps)
            dxd = dpd if dxd is None else ops.add(dxd, dpd)
        if gru:
            tr = saved['tracks']
            d = GRUDecoder.backward_many(self.sed_dec + self.doa_dec, pend_s + pend_d,
                                         [tr[t]['s_s'] for t in range(3)] + [tr[t]['s_d'] for t in range(3)], B)
            dxs, dxd = ops.add(ops.add(d[0], d[1]), d[2]), ops.add(ops.add(d[3], d[4]), d[5])
        return dxs, dxd


def _load_htsat_ckpt(net, pretrained_path, audioset_pretrain, encoders):
    ck = torch.load(pretrained_path, map_location='cpu')['state_dict']
    own = net.state_dict()
    if audioset_pretrain:
        ck = {k.replace('sed_model.', ''): v for k, v in ck.items()}
        for prefix, cin in encoders:
            for key in own:
                if not key.startswith(prefix):
                    continue
                src = key[len(prefix):]
                if src == 'patch_embed.proj.weight':
                    own[key].copy_(ck[src].repeat(1, cin, 1, 1) / cin)
                else:
                    own[key].copy_(ck[src])
        for c in range(net.in_channels):
            for leaf in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                own[f'scalar.{c}.{leaf}'].copy_(ck[f'bn0.{leaf}'])
    else:
        ck = {k.replace('net.', '').replace('_orig_mod.', ''): v for k, v in ck.items()}
        for key, value in ck.items():
            if key.startswith(('sed_tscam_conv.', 'head', 'af_extractor')):
                continue
            if key in own:
                own[key].copy_(value)
            elif 'relative_position_index' not in key and 'attn_mask' not in key:       # index buffers this path does not keep
                raise KeyError(f'{pretrained_path}: unexpected entry {key}')
    net.shadow_trusted = False


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
            self.load_ckpts(pretrained_path, audioset_pretrain)

    def load_ckpts(self, pretrained_path, audioset_pretrain=True):
        """einv2.py:239-272: AudioSet HTS-AT checkpoints into both encoders (one-channel patch embedding replicated /
        in_channels, bn0 copied into every scalar) or PSELDNets checkpoints (sed_tscam_conv / head / af_extractor skipped)."""
        _load_htsat_ckpt(self, pretrained_path, audioset_pretrain,
                         (('sed_encoder.', self.sed_in_channels), ('doa_encoder.', self.doa_in_channels)))

    def _forward_impl(self, x, training):
        B, dt = x.shape[0], self.compute_dtype
        a = self.arena
        box = []
        mean_rstd, scale_shift = self._bn_front(x, training, overlap=lambda: box.extend((self._drop_scales(B, self.sed_enc, x.device, training), self._drop_scales(B, self.doa_enc, x.device, training))))
        drop_s, drop_d = box
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
            self.load_ckpts(pretrained_path, audioset_pretrain)

    def load_ckpts(self, pretrained_path, audioset_pretrain=True):
        """einv2.py:369-396: as einv2.HTSAT.load_ckpts with the single encoder."""
        _load_htsat_ckpt(self, pretrained_path, audioset_pretrain, (('encoder.', self.in_channels),))

    def _forward_impl(self, x, training):
        B, dt = x.shape[0], self.compute_dtype
        box = []
        mean_rstd, scale_shift = self._bn_front(x, training, overlap=lambda: box.append(self._drop_scales(B, self.enc, x.device, training)))
        drop = box[0]
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


class CRNN(StaticBufferMixin, HTSATNetBase):
    """einv2.py:17-174: scalar BatchNorms -> CNN8 / CNN12 twice (SED on the first 4 channels, DOA on all), a CrossStitch
    after every ConvBlock but the last -> frequency mean -> three (SED, DOA) Decoder pairs, each stitched ->
    'repeat' x8 interpolation + 10-frame mean -> per-track Linear heads. The SED stack reads the same NHWC input rows
    as the DOA stack: its first-layer weight copy has zero columns for the channels it must not see."""

    def __init__(self, cfg, num_classes, in_channels=7, encoder='CNN8', pretrained_path=None, audioset_pretrain=True,
                 num_features=[32, 64, 128, 256]):
        super().__init__()
        decoder, n_layers = decoder_config(cfg)
        self.num_classes = num_classes
        self.sed_in_channels, self.doa_in_channels = 4, in_channels
        self.interpolate_time_ratio = 2 ** 3
        self._init_common(cfg, in_channels)
        nf = list(num_features)
        self.num_features = nf
        for i, c in enumerate(nf + [nf[-1], nf[-1]]):
            self.arena.add(f'stitch.{i}.weight', (c, 2, 2))
        self.sed_enc = ConvEncoder(self.arena, 'sed_convs.', self.sed_in_channels, encoder, nf)
        self.doa_enc = ConvEncoder(self.arena, 'doa_convs.', self.doa_in_channels, encoder, nf)
        if self.sed_enc.cin_p != self.doa_enc.cin_p:
            raise NotImplementedError("the SED and DOA stacks share one padded input: in_channels must be <= 8")
        if len(self.sed_enc.pools) != len(nf):
            raise ValueError(f'{encoder} needs {len(self.sed_enc.pools)} feature widths')
        # stitch[-3:] couple the three tracks; stitch[:len-1] the ConvBlocks (einv2.py:36-38,119-124,138-140)
        n_st = len(nf) + 2
        self.tracks = EinTracks(self.arena, nf[-1], num_classes, decoder, n_layers, dict(
            sed_dec=lambda t: f'sed_track{t + 1}.decoder.', doa_dec=lambda t: f'doa_track{t + 1}.decoder.',
            stitch=lambda t: f'stitch.{n_st - 3 + t}.', fc_sed=lambda t: f'fc_sed_track{t + 1}.',
            fc_doa=lambda t: f'fc_doa_track{t + 1}.'), register_stitch=False)
        self._finish_init()
        self._taps = None
        if pretrained_path:
            self.load_ckpts(pretrained_path, audioset_pretrain)

    def load_ckpts(self, pretrained_path, audioset_pretrain=True):
        """einv2.py:69-96: PANNs CNN14 checkpoints into both stacks (first conv replicated / in_channels, bn0 copied
        into every scalar) or PSELDNets checkpoints (fc_sed* skipped)."""
        own = self.state_dict()
        if audioset_pretrain:
            ck = torch.load(pretrained_path, map_location='cpu')['model']
            for pre, cin in (('sed_convs.', self.sed_in_channels), ('doa_convs.', self.doa_in_channels)):
                for key in own:
                    if not key.startswith(pre):
                        continue
                    src = key[len(pre):]
                    if src == 'conv_block1.conv1.weight':
                        own[key].copy_(ck[src].repeat(1, cin, 1, 1) / cin)
                    else:
                        own[key].copy_(ck[src])
            for c in range(self.in_channels):
                for leaf in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                    own[f'scalar.{c}.{leaf}'].copy_(ck[f'bn0.{leaf}'])
        else:
            ck = torch.load(pretrained_path, map_location='cpu')['state_dict']
            ck = {k.replace('net.', '').replace('_orig_mod.', ''): v for k, v in ck.items()}
            for key in own:
                if not key.startswith('fc_sed'):
                    own[key].copy_(ck[key])
        self.shadow_trusted = False

    def _encoders(self):
        return [self.sed_enc, self.doa_enc] + self.tracks.decoders()

    def _pool(self, device, n_in):
        if self._taps is None or self._taps['i0'].device != device or self._taps['n_in'] != n_in:
            taps = ops.pool_taps(n_in=n_in, ratio=self.interpolate_time_ratio, n_keep=self.tgt_output_frames * self.pred_res,
                                 group=self.pred_res, method='repeat')
            self._taps = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in taps.items()}
        return self._taps

    def _forward_impl(self, x, training):
        B, _, T, F = x.shape
        dt, a, bufs = self.compute_dtype, self.arena, self._bn_bufs
        mean_rstd, scale_shift = self._bn_front(x, training)
        x0 = ops.cnn_input(x, scale_shift, dt, self.doa_enc.cin_p)
        xs = xd = x0
        blocks, nb = [], len(self.num_features)
        for i in range(nb):
            xs, ss, _, _ = self.sed_enc.forward_block(i, xs, B, T, F, dt, training, bufs)
            xd, sd, T, F = self.doa_enc.forward_block(i, xd, B, T, F, dt, training, bufs)
            s = dict(sed=ss, doa=sd)
            if i < nb - 1:
                s['xs'], s['xd'] = xs, xd
                xs, xd = ops.cross_stitch_fwd(xs, xd, a.p(f'stitch.{i}.weight').view(-1, 4))
            blocks.append(s)
        xs, xd = self.sed_enc.forward_tail(xs, B, T, F), self.doa_enc.forward_tail(xd, B, T, F)
        if T * self.interpolate_time_ratio != self.tgt_output_frames * self.pred_res:
            raise NotImplementedError(f"{T} encoder frames x {self.interpolate_time_ratio} do not cover "
                                      f"{self.tgt_output_frames} x {self.pred_res} output frames")
        sed, doa, s_tr = self.tracks.forward(xs, xd, B, T, training, bufs, taps=self._pool(x.device, T))
        return (sed, doa), dict(feat=x, mean_rstd=mean_rstd, blocks=blocks, tracks=s_tr, T=T, F=F, B=B)

    def _backward_impl(self, saved, douts, on_range_done=None):
        dsed, ddoa = douts
        B, dt, a = saved['B'], self.compute_dtype, self.arena
        dxs, dxd = self.tracks.backward(dsed, ddoa, saved['tracks'], B)
        dxs = self.sed_enc.backward_tail(dxs, B, saved['T'], saved['F'])
        dxd = self.doa_enc.backward_tail(dxd, B, saved['T'], saved['F'])
        nb = len(self.num_features)
        for i in reversed(range(nb)):
            s = saved['blocks'][i]
            if i < nb - 1:
                dxs, dxd = ops.cross_stitch_bwd(s['xs'], s['xd'], a.p(f'stitch.{i}.weight').view(-1, 4), dxs, dxd,
                                                a.g(f'stitch.{i}.weight').view(-1, 4))
            dxs = self.sed_enc.backward_block(i, dxs, s['sed'], B, dt)
            dxd = self.doa_enc.backward_block(i, dxd, s['doa'], B, dt)
        dw, db = self._bn_grads()
        ops.cnn_input_bwd(saved['feat'], saved['mean_rstd'], ops.add(dxs, dxd), dw, db)
        if on_range_done is not None:
            on_range_done(0, a.size)

    def forward(self, x):
        sed, doa = self._run(x)
        return {'sed': sed, 'doa': doa}


class PASST(StaticBufferMixin, HTSATNetBase):
    """einv2.py:446-575: scalar BatchNorms -> PaSST twice (SED on the first 4 channels, DOA on all), a CrossStitch in
    front of every `cfg.model.ps_gap`-th block -> three (SED, DOA) Decoder pairs on the [B, 100, E] feature maps, each
    stitched -> per-track Linear heads."""

    def __init__(self, cfg, num_classes, in_channels=7, pretrained_path=None, audioset_pretrain=True, **kwargs):
        super().__init__()
        decoder, n_layers = decoder_config(cfg)
        model = cfg.model if hasattr(cfg, 'model') else cfg['model']
        self.ps_gap = model.ps_gap if hasattr(model, 'ps_gap') else model['ps_gap']
        self.num_classes = num_classes
        self.sed_in_channels, self.doa_in_channels = 4, in_channels
        self._init_common(cfg, in_channels)
        self.sed_enc = PasstEncoder(self.arena, 'sed_encoder.', self.sed_in_channels, mel_bins=self.mel_bins, **kwargs)
        self.doa_enc = PasstEncoder(self.arena, 'doa_encoder.', self.doa_in_channels, mel_bins=self.mel_bins, **kwargs)
        E = self.sed_enc.E
        for i in range((self.sed_enc.depth - 1) // self.ps_gap + 1):
            self.arena.add(f'stitch1.{i}.weight', (E, 2, 2))
        self.tracks = EinTracks(self.arena, E, num_classes, decoder, n_layers, dict(
            sed_dec=lambda t: f'sed_decoder.{t}.decoder.', doa_dec=lambda t: f'doa_decoder.{t}.decoder.',
            stitch=lambda t: f'stitch2.{t}.', fc_sed=lambda t: f'fc_sed.{t}.', fc_doa=lambda t: f'fc_doa.{t}.'))
        self._finish_init()
        if pretrained_path:
            self.load_ckpts(pretrained_path, audioset_pretrain)

    def load_ckpts(self, pretrained_path, audioset_pretrain=True):
        """einv2.py:486-533: AudioSet PaSST checkpoints into both encoders (the rules of accdoa.PASST.load_ckpts) or
        PSELDNets checkpoints (fc_sed* skipped)."""
        own = self.state_dict()
        if audioset_pretrain:
            ck = torch.load(pretrained_path, map_location='cpu')
            for pre, cin in (('sed_encoder.', self.sed_in_channels), ('doa_encoder.', self.doa_in_channels)):
                for key in own:
                    if key.startswith(pre):
                        accdoa.copy_passt_entry(own[key], key[len(pre):], ck, cin)
        else:
            ck = torch.load(pretrained_path, map_location='cpu')['state_dict']
            ck = {k.replace('net.', '').replace('_orig_mod.', ''): v for k, v in ck.items()}
            for key in own:
                if not key.startswith('fc_sed'):
                    own[key].copy_(ck[key])
        self.shadow_trusted = False

    def _encoders(self):
        return [self.sed_enc, self.doa_enc] + self.tracks.decoders()

    def _forward_impl(self, x, training):
        B, dt, a = x.shape[0], self.compute_dtype, self.arena
        box = []
        mean_rstd, scale_shift = self._bn_front(x, training, overlap=lambda: box.extend((self._drop_scales(B, self.sed_enc, x.device, training), self._drop_scales(B, self.doa_enc, x.device, training))))
        drop_s, drop_d = box
        xs, fs = self.sed_enc.forward_front(x, scale_shift, dt, training)
        xd, fd = self.doa_enc.forward_front(x, scale_shift, dt, training)
        blocks = []
        for i in range(self.sed_enc.depth):
            s = {}
            if i % self.ps_gap == 0:
                s['xs'], s['xd'] = xs, xd
                xs, xd = ops.cross_stitch_fwd(xs, xd, a.p(f'stitch1.{i // self.ps_gap}.weight').view(-1, 4))
            xs, s['sed'] = self.sed_enc.forward_block(i, xs, B, drop_s)
            xd, s['doa'] = self.doa_enc.forward_block(i, xd, B, drop_d)
            blocks.append(s)
        ms, bs = self.sed_enc.forward_back(xs, B, fs['maps'])
        md, bd = self.doa_enc.forward_back(xd, B, fd['maps'])
        sed, doa, s_tr = self.tracks.forward(ms, md, B, self.sed_enc.Tg, training, self._bn_bufs)
        return (sed, doa), dict(feat=x, mean_rstd=mean_rstd, fs=fs, fd=fd, blocks=blocks, bs=bs, bd=bd, tracks=s_tr, B=B)

    def _backward_impl(self, saved, douts, on_range_done=None):
        dsed, ddoa = douts
        B, a = saved['B'], self.arena
        dms, dmd = self.tracks.backward(dsed, ddoa, saved['tracks'], B)
        dxs = self.sed_enc.backward_back(dms, saved['bs'], B)
        dxd = self.doa_enc.backward_back(dmd, saved['bd'], B)
        for i in reversed(range(self.sed_enc.depth)):
            s = saved['blocks'][i]
            dxs = self.sed_enc.backward_block(i, dxs, s['sed'], B)
            dxd = self.doa_enc.backward_block(i, dxd, s['doa'], B)
            if i % self.ps_gap == 0:
                w = f'stitch1.{i // self.ps_gap}.weight'
                dxs, dxd = ops.cross_stitch_bwd(s['xs'], s['xd'], a.p(w).view(-1, 4), dxs, dxd, a.g(w).view(-1, 4))
        dw, db = self._bn_grads()
        self.doa_enc.backward_front(dxd, saved['fd'], saved['feat'], saved['mean_rstd'], dw, db, B, accumulate_bn=False)
        self.sed_enc.backward_front(dxs, saved['fs'], saved['feat'], saved['mean_rstd'], dw, db, B, accumulate_bn=True)
        ops.join_wgrads(dxs.device)                    # the blocks' weight gradients ran on the second stream (ops.linear_wgrad_side)
        if on_range_done is not None:
            on_range_done(0, a.size)

    def forward(self, x):
        sed, doa = self._run(x)
        return {'sed': sed, 'doa': doa}


ConvConformer = accdoa._NotBuilt
