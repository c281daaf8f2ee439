"""ACCDOA networks on MI355X — mirror of the reference's `models/accdoa.py` registry module (HTSAT :107-246).
Constructor signature, forward contract ({'accdoa': f32[B, 100, 3*C]}) and state-dict keys are the reference's;
the arithmetic is the HIP path (components/htsat.py). PASST (:249-329) runs on the same kernels plus the global-attention ones; the CRNN / ConvConformer backbones of the
reference registry are not built on this path yet and raise NotImplementedError."""
import torch

from .components.htsat import SwinEncoder, TscamHead
from .components.passt import FcTanhHead, PasstEncoder
from .components.crnn import ConvEncoder
from .components.conformer import ConformerDecoder
from .components.gru import GRUDecoder
from .components.transformer import TransformerDecoder
from .. import ops
from .components.seld_net import HTSATNetBase


class HTSAT(HTSATNetBase):
    out_key = 'accdoa'
    tracks_axes = 3

    def __init__(self, cfg, num_classes, in_channels=7, audioset_pretrain=True,
                 pretrained_path='ckpts/HTSAT-fullset-imagenet-768d-32000hz.ckpt', **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self._init_common(cfg, in_channels)
        cfg_adapt = cfg.adapt if hasattr(cfg, 'adapt') else (cfg.get('adapt', {}) if isinstance(cfg, dict) else {})
        cfg_adapt = dict(cfg_adapt or {})
        self.enc = SwinEncoder(self.arena, 'encoder.', in_channels, mel_bins=self.mel_bins, cfg_adapt=cfg_adapt, **kwargs)
        self.head = TscamHead(self.arena, 'tscam_conv.', self.enc.num_features, num_classes * self.tracks_axes, True)
        self._finish_init()
        if pretrained_path:
            self.load_ckpts(pretrained_path, audioset_pretrain)
        self.freeze_layers_if_needed(str(cfg_adapt.get('method', '') or ''))
        for n_, p_ in self.named_parameters():
            if n_.startswith('tscam_conv.'):
                p_.requires_grad_(True)                       # accdoa.py:146

    def freeze_layers_if_needed(self, adapt_method):
        """accdoa.py:148-170: adapter fine-tuning trains the biases (every parameter whose name contains 'bias', the
        relative-position bias tables included), the adapters and the head; 'mono_adapter' without adapters trains all."""
        if 'lora' in adapt_method:             # model_utilities_adapt.py:91,141: the base weight of every LoRA layer is frozen
            for name, param in self.named_parameters():
                if name.endswith('.weight') and (name[:-len('weight')] + 'lora_A' in self.arena.offsets
                                                 or name[:-len('weight')] + 'lora_A.weight' in self.arena.offsets):
                    param.requires_grad_(False)
        if 'adapter' not in adapt_method:
            return
        found = False
        self.requires_grad_(False)
        for name, param in self.named_parameters():
            if 'bias' in name:
                param.requires_grad_(True)
            if 'adapter' in name or 'lora' in name:
                found = True
                param.requires_grad_(True)
        if adapt_method == 'mono_adapter' and not found:
            self.requires_grad_(True)
        else:
            self.enc.frozen_weights = True     # the backward skips the frozen weight-gradient GEMMs (bias gradients stay)

    def load_ckpts(self, pretrained_path, audioset_pretrain=True):
        """accdoa.py:172-202: AudioSet HTS-AT checkpoints (1-channel patch-embed replicated / in_channels, bn0 copied
        into every scalar) or PSELDNets checkpoints (heads skipped)."""
        ck = torch.load(pretrained_path, map_location='cpu')['state_dict']
        own = self.state_dict()
        if audioset_pretrain:
            ck = {k.replace('sed_model.', ''): v for k, v in ck.items()}
            for key in own:
                if not key.startswith('encoder.'):
                    continue
                src = key[len('encoder.'):]
                if src == 'patch_embed.proj.weight':
                    own[key].copy_(ck[src].repeat(1, self.in_channels, 1, 1) / self.in_channels)
                elif src in ck and 'tscam_conv' not in src and 'head' not in src:
                    own[key].copy_(ck[src])
            for c in range(self.in_channels):
                for leaf in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                    own[f'scalar.{c}.{leaf}'].copy_(ck[f'bn0.{leaf}'])
        else:
            ck = {k.replace('net.', '').replace('_orig_mod.', ''): v for k, v in ck.items()}
            for key in own:
                if key.startswith(('fc.', 'head.', 'tscam_conv.')) or key not in ck:
                    continue
                own[key].copy_(ck[key])
        self.shadow_trusted = False

    # -- the two halves the autograd node and the fused step share ------------------------------------------------
    def _forward_impl(self, x, training):
        B = x.shape[0]
        dt = self.compute_dtype
        box = []
        ops.stage('front')
        mean_rstd, scale_shift = self._bn_front(x, training, overlap=lambda: box.append(self._drop_scales(B, self.enc, x.device, training)))
        drop = box[0]
        tok, s_patch = self.enc.forward_patch(x, scale_shift, dt)
        s_layers = []
        for li in range(self.enc.nl):
            ops.stage(f'stage{li}')
            tok, s = self.enc.forward_layer(li, tok, B, drop)
            s_layers.append(s)
        ops.stage('head+loss')
        xn, s_fin = self.enc.forward_final(tok)
        y, s_head = self.head.forward(xn, B)
        return y, dict(feat=x, mean_rstd=mean_rstd, patch=s_patch, layers=s_layers, fin=s_fin, head=s_head, B=B)

    def _backward_impl(self, saved, douts, on_range_done=None):
        """Hand-written backward. `on_range_done(a, b)` is told each arena range [a, b) whose gradients are final,
        back to front, so the data-parallel loop can all-reduce it while earlier layers are still running."""
        dy = douts[0] if isinstance(douts, (tuple, list)) else douts
        B, dt = saved['B'], self.compute_dtype
        a = self.arena
        ops.defer_stage_joins(on_range_done is None)     # no gradient range leaves before the end: one join of the weight-gradient stream, below
        try:
            ops.stage('head+loss')
            dxn = self.head.backward(dy, saved['head'], B, dt)
            dx = self.enc.backward_final(dxn, saved['fin'])
            hi = a.size
            for li in reversed(range(self.enc.nl)):
                ops.stage(f'stage{li}')
                dx = self.enc.backward_layer(li, dx, saved['layers'][li], B)
                if on_range_done is not None and li in (3, 2):     # buckets: {stage3+norm+head}, {stage2}, {rest}
                    lo = a.offsets[self.enc.first_param_of_layer(li)][0]
                    on_range_done(lo, hi)
                    hi = lo
            ops.stage('front')
            dw, db = self._bn_grads()
            self.enc.backward_patch(dx, saved['patch'], saved['feat'], saved['mean_rstd'], dw, db, accumulate_bn=False)
            ops.join_wgrads(dx.device)
        finally:
            ops.defer_stage_joins(False)
        if on_range_done is not None:
            on_range_done(0, hi)

    def forward(self, x):
        """
        x: (batch_size, num_channels, time_frames, mel_bins) features of 10-second chunks
        """
        return {self.out_key: self._run(x)}


def copy_passt_entry(dst, src, ck, in_channels):
    """accdoa.py:273-300 for one encoder entry `src` of an AudioSet PaSST checkpoint `ck`."""
    if src == 'patch_embed.proj.weight':
        dst.copy_(ck[src].repeat(1, in_channels, 1, 1) / in_channels)
    elif src in ('time_new_pos_embed', 'freq_new_pos_embed'):
        axis = -1 if src.startswith('time') else -2
        have, want = ck[src].shape[axis], dst.shape[axis]
        if have >= want:
            dst.copy_(ck[src].narrow(axis, int((have - want) / 2), want))
        else:
            dst.copy_(torch.nn.functional.interpolate(ck[src], size=(1, want), mode='bilinear'))
    elif 'head' in src:
        if src in ('head.0.weight', 'head.0.bias'):
            dst.copy_(ck[src])
    else:
        dst.copy_(ck[src])


class PASST(HTSATNetBase):
    """models/accdoa.py:249-329: scalar BatchNorms -> PaSST -> Linear(E, 3*C) -> tanh."""
    out_key = 'accdoa'
    tracks_axes = 3

    def __init__(self, cfg, num_classes, in_channels=7, pretrained_path='ckpts/passt-s-f128-p16-s10-ap.476-swa.pt', **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self._init_common(cfg, in_channels)
        self.enc = PasstEncoder(self.arena, 'encoder.', in_channels, mel_bins=self.mel_bins, **kwargs)
        self.head = FcTanhHead(self.arena, 'fc.', self.enc.num_features, num_classes * self.tracks_axes)
        self._finish_init()
        if pretrained_path:
            self.load_ckpts(pretrained_path)

    def load_ckpts(self, pretrained_path, audioset_pretrain=True):
        """accdoa.py:270-309: AudioSet PaSST checkpoints (1-channel patch-embed replicated / in_channels, the time /
        frequency positional embeddings centre-cropped — or bilinearly stretched — to the grid in use, only head.0 of
        the classifier kept) or PSELDNets checkpoints (fc skipped)."""
        own = self.state_dict()
        if audioset_pretrain:
            ck = torch.load(pretrained_path, map_location='cpu')
            for key in own:
                if key.startswith('encoder.'):
                    copy_passt_entry(own[key], key[len('encoder.'):], ck, self.in_channels)
        else:
            ck = torch.load(pretrained_path, map_location='cpu')['state_dict']
            ck = {k.replace('net.', '').replace('_orig_mod.', ''): v for k, v in ck.items()}
            for key in own:
                if not key.startswith('fc.'):
                    own[key].copy_(ck[key])
        self.shadow_trusted = False

    def _forward_impl(self, x, training):
        B, dt = x.shape[0], self.compute_dtype
        box = []
        mean_rstd, scale_shift = self._bn_front(x, training, overlap=lambda: box.append(self._drop_scales(B, self.enc, x.device, training)))
        drop = box[0]
        tok, s_front = self.enc.forward_front(x, scale_shift, dt, training)
        s_blocks = []
        for i in range(self.enc.depth):
            tok, s = self.enc.forward_block(i, tok, B, drop)
            s_blocks.append(s)
        fmap, s_back = self.enc.forward_back(tok, B, s_front['maps'])
        y, s_head = self.head.forward(fmap, B)
        return y, dict(feat=x, mean_rstd=mean_rstd, front=s_front, blocks=s_blocks, back=s_back, head=s_head, B=B)

    def _backward_impl(self, saved, douts, on_range_done=None):
        dy = douts[0] if isinstance(douts, (tuple, list)) else douts
        B, dt, a = saved['B'], self.compute_dtype, self.arena
        dfmap = self.head.backward(dy, saved['head'], dt)
        dx = self.enc.backward_back(dfmap, saved['back'], B)
        hi = a.size
        cuts = {self.enc.depth - 2, self.enc.depth - 4}          # all-reduce buckets: last 2 blocks + tail, 2 more, rest
        for i in reversed(range(self.enc.depth)):
            dx = self.enc.backward_block(i, dx, saved['blocks'][i], B)
            if on_range_done is not None and i in cuts and i > 0:
                lo = a.offsets[self.enc.first_param_of_block(i)][0]
                ops.join_wgrads(dx.device)             # the blocks' weight gradients ran on the second stream (ops.linear_wgrad_side)
                on_range_done(lo, hi)
                hi = lo
        dw, db = self._bn_grads()
        self.enc.backward_front(dx, saved['front'], saved['feat'], saved['mean_rstd'], dw, db, B)
        ops.join_wgrads(dx.device)
        if on_range_done is not None:
            on_range_done(0, hi)

    def forward(self, x):
        """
        x: (batch_size, num_channels, time_frames, mel_bins) features of 10-second chunks
        """
        return {self.out_key: self._run(x)}


def decoder_config(cfg):
    """(cfg.model.decoder, cfg.model.num_decoder_layers) of an attribute- or dict-style config."""
    model = cfg.model if hasattr(cfg, 'model') else cfg.get('model', {})
    if model is None:
        return None, 2
    decoder = model.decoder if hasattr(model, 'decoder') else model.get('decoder')
    n_layers = model.num_decoder_layers if hasattr(model, 'num_decoder_layers') else model.get('num_decoder_layers', 2)
    return decoder, n_layers


def make_decoder(arena, prefix, decoder, num_feats, n_layers):
    """model_utilities.py:245-263 `Decoder`: the block stack behind `prefix` (= '<name>.decoder.'), None for nn.Identity."""
    if decoder == 'conformer':
        return ConformerDecoder(arena, prefix, num_feats, n_layers)
    if decoder == 'gru':
        return GRUDecoder(arena, prefix, num_feats, n_layers)
    if decoder == 'transformer':
        return TransformerDecoder(arena, prefix, num_feats, n_layers)
    if decoder is None:
        return None
    raise NotImplementedError(f"{decoder} is not implemented")          # model_utilities.py:262-263


class StaticBufferMixin:
    """Moves the non-arena buffers of the encoders / decoders (BatchNorm running statistics, positional tables) onto the
    device when the arena is materialised and keeps them addressable by their reference names in `self._bn_bufs`."""
    _bn_bufs = None

    def _materialize(self, device):
        fresh = self._materialized_on != device
        super()._materialize(device)
        if fresh:
            from .components.seld_net import _get
            self._bn_bufs = {}
            names = [n for e in self._encoders() for n in e.static_buffers()]
            for name in names:
                node = _get(self, name.rsplit('.', 1)[0])
                leaf = name.rsplit('.', 1)[1]
                t = node._buffers[leaf].detach().to(device)
                t = (t.float() if t.is_floating_point() else t.long()).contiguous()
                node._buffers[leaf] = t
                self._bn_bufs[name] = t


class CRNN(StaticBufferMixin, HTSATNetBase):
    """models/accdoa.py:12-95: scalar BatchNorms -> CNN8 / CNN12 (the PANNs CNN14 conv stack) -> frequency mean ->
    decoder -> 'repeat' x8 interpolation + 10-frame mean -> Linear -> tanh. Built on the MI355X path with
    `cfg.model.decoder` = 'conformer' (configs/model/crnn.yaml:5; ConformerBlocks, model_utilities.py:254-255) or None
    (nn.Identity, :260-261), 'gru' or 'transformer'."""
    out_key = 'accdoa'
    tracks_axes = 3
    decoder_prefix = 'decoder.decoder.'          # Decoder(...).decoder = ConformerBlocks
    forced_decoder_layers = None

    def __init__(self, cfg, num_classes, in_channels=7, encoder='CNN8', pretrained_path=None, audioset_pretrain=True,
                 num_features=[32, 64, 128, 256]):
        super().__init__()
        decoder, n_layers = decoder_config(cfg)
        if self.forced_decoder_layers is not None:
            decoder, n_layers = 'conformer', self.forced_decoder_layers
        if decoder not in (None, 'conformer', 'gru', 'transformer'):
            raise NotImplementedError(f"{decoder} is not implemented")          # model_utilities.py:262-263
        self.num_classes = num_classes
        self.interpolate_time_ratio = 2 ** 3
        self._init_common(cfg, in_channels)
        self.conv_enc = ConvEncoder(self.arena, 'convs.', in_channels, encoder, list(num_features))
        self.num_features = list(num_features)
        self.dec_blocks = make_decoder(self.arena, self.decoder_prefix, decoder, self.num_features[-1], n_layers)
        self.head = FcTanhHead(self.arena, 'fc.', self.num_features[-1], num_classes * self.tracks_axes)
        self._finish_init()
        self._taps = None
        self._bn_bufs = None
        if pretrained_path:
            self.load_ckpts(pretrained_path, audioset_pretrain)

    def load_ckpts(self, pretrained_path, audioset_pretrain=True):
        """accdoa.py:44-63: PANNs CNN14 checkpoints (first conv replicated / in_channels, bn0 copied into every scalar)
        or PSELDNets checkpoints (fc skipped)."""
        own = self.state_dict()
        if audioset_pretrain:
            ck = torch.load(pretrained_path, map_location='cpu')['model']
            for key in own:
                if not key.startswith('convs.'):
                    continue
                src = key[len('convs.'):]
                if src == 'conv_block1.conv1.weight':
                    own[key].copy_(ck[src].repeat(1, self.in_channels, 1, 1) / self.in_channels)
                else:
                    own[key].copy_(ck[src])
            for c in range(self.in_channels):
                for leaf in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'):
                    own[f'scalar.{c}.{leaf}'].copy_(ck[f'bn0.{leaf}'])
        else:
            ck = torch.load(pretrained_path, map_location='cpu')['state_dict']
            ck = {k.replace('net.', '').replace('_orig_mod.', ''): v for k, v in ck.items()}
            for key in own:
                if not key.startswith('fc.'):
                    own[key].copy_(ck[key])
        self.shadow_trusted = False

    def _encoders(self):
        return [self.conv_enc] + ([self.dec_blocks] if self.dec_blocks is not None else [])

    def _pool(self, device, n_in):
        if self._taps is None or self._taps['i0'].device != device or self._taps['n_in'] != n_in:
            taps = ops.pool_taps(n_in=n_in, ratio=self.interpolate_time_ratio, n_keep=self.tgt_output_frames * self.pred_res,
                                 group=self.pred_res, method='repeat')
            self._taps = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in taps.items()}
        return self._taps

    def _forward_impl(self, x, training):
        B, _, T, F = x.shape
        dt = self.compute_dtype
        mean_rstd, scale_shift = self._bn_front(x, training)
        x0 = ops.cnn_input(x, scale_shift, dt, self.conv_enc.cin_p)
        enc, s_enc = self.conv_enc.forward(x0, B, T, F, dt, training, self._bn_bufs)
        n_in = s_enc['T_out']
        s_dec = None
        if self.dec_blocks is not None:
            enc, s_dec = self.dec_blocks.forward(enc, B, n_in, training, self._bn_bufs)
        if n_in * self.interpolate_time_ratio != self.tgt_output_frames * self.pred_res:
            raise NotImplementedError(f"{n_in} encoder frames x {self.interpolate_time_ratio} do not cover "
                                      f"{self.tgt_output_frames} x {self.pred_res} output frames")
        taps = self._pool(x.device, n_in)
        pooled = ops.rows_pool_fwd(enc, taps, B)
        y, s_head = self.head.forward(pooled, B)
        return y, dict(feat=x, mean_rstd=mean_rstd, enc=s_enc, dec=s_dec, head=s_head, taps=taps, B=B)

    def _backward_impl(self, saved, douts, on_range_done=None):
        dy = douts[0] if isinstance(douts, (tuple, list)) else douts
        B, dt = saved['B'], self.compute_dtype
        dpooled = self.head.backward(dy, saved['head'], dt)
        denc = ops.rows_pool_bwd(dpooled, saved['taps'], B)
        if saved['dec'] is not None:
            denc = self.dec_blocks.backward(denc, saved['dec'], B)
        dx0 = self.conv_enc.backward(denc, saved['enc'], B, dt)
        dw, db = self._bn_grads()
        ops.cnn_input_bwd(saved['feat'], saved['mean_rstd'], dx0, dw, db)
        if on_range_done is not None:
            on_range_done(0, self.arena.size)

    def forward(self, x):
        """
        x: waveform features, (batch_size, num_channels, time_frames, mel_bins)
        """
        return {self.out_key: self._run(x)}


class _NotBuilt:
    """Registry entries the reference itself cannot construct (einv2.ConvConformer: its super().__init__ call passes the arguments of
    an older signature and fails on `cfg.data`, einv2.py:177-180)."""

    def __init__(self, *a, **k):
        raise NotImplementedError("einv2.ConvConformer cannot be constructed in the reference either (einv2.py:177-180); "
                                  "use einv2.CRNN with model.decoder=conformer")


class ConvConformer(CRNN):
    """models/accdoa.py:98-104: CRNN whose decoder is replaced by ConformerBlocks(num_layers=2) (keys `decoder.layers.*`)."""
    decoder_prefix = 'decoder.'
    forced_decoder_layers = 2
