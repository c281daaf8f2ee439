"""Shared machinery of the HTS-AT SELD networks: reference-named module tree over a flat parameter arena, the
per-input-channel "scalar" BatchNorm front (models/accdoa.py:223-227), autograd glue, and the fused
clip+AdamW step. Concrete networks (models/accdoa.py, multi_accdoa.py, einv2.py mirrors) only declare their
encoders and heads."""
import torch
import torch.nn as nn

from ... import _lib, ops
from .arena import ParamArena
from .htsat import default_init


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state-dict keys."""


def _attach(root, dotted, tensor, is_param):
    parts = dotted.split('.')
    node = root
    for part in parts[:-1]:
        if not hasattr(node, part):
            node.add_module(part, _Node())
        node = getattr(node, part)
    if is_param:
        node.register_parameter(parts[-1], nn.Parameter(tensor))
    else:
        node.register_buffer(parts[-1], tensor)


def _get(root, dotted):
    obj = root
    for part in dotted.split('.'):
        obj = getattr(obj, part)
    return obj


class _NetFn(torch.autograd.Function):
    """Whole-network autograd node: forward runs the HIP forward, backward the hand-written HIP backward, which
    deposits every parameter gradient in the arena; the per-parameter views are handed back to autograd."""

    @staticmethod
    def forward(ctx, net, x, *params):
        outs, saved = net._forward_impl(x, net.training)
        ctx.net, ctx.saved_state = net, saved
        return outs

    @staticmethod
    def backward(ctx, *douts):
        net = ctx.net
        net._backward_impl(ctx.saved_state, douts)
        grads = tuple(net.arena.g(n).clone() for n in net.arena.entries)
        return (None, None) + grads


class HTSATNetBase(nn.Module):
    compute_dtype = torch.float32       # float32 = parity mode; bfloat16 = throughput mode

    def _init_common(self, cfg, in_channels):
        data = cfg.data if hasattr(cfg, 'data') else cfg['data']
        get = (lambda k: getattr(data, k)) if not isinstance(data, dict) else (lambda k: data[k])
        self.mel_bins = get('n_mels')
        self.label_res = 0.1
        self.output_frames = None
        self.tgt_output_frames = int(10 / 0.1)
        self.pred_res = int(get('sample_rate') / get('hoplen') * self.label_res)
        self.in_channels = in_channels
        self.arena = ParamArena()
        for c in range(in_channels):
            self.arena.add(f'scalar.{c}.weight', (self.mel_bins,))
        for c in range(in_channels):
            self.arena.add(f'scalar.{c}.bias', (self.mel_bins,))
        self._materialized_on = None
        self.sync_bn_group = None         # set to a torch.distributed group for sync-BN (configs/trainer/gpu.yaml:9)
        self.shadow_trusted = False
        self.bn_momentum, self.bn_eps = 0.1, 1e-5

    def _finish_init(self):
        """Create the reference-named parameter/buffer tree (CPU tensors until the first forward on a GPU)."""
        for name in self.arena.entries:
            _attach(self, name, default_init(name, self.arena.offsets[name][1]), True)
        for enc in self._encoders():
            for name, t in enc.static_buffers().items():
                _attach(self, name, t, False)
        C = self.in_channels
        self._rm = torch.zeros(C, self.mel_bins)
        self._rv = torch.ones(C, self.mel_bins)
        self._nbt = torch.zeros(C, dtype=torch.long)
        for c in range(C):
            _attach(self, f'scalar.{c}.running_mean', self._rm[c], False)
            _attach(self, f'scalar.{c}.running_var', self._rv[c], False)
            _attach(self, f'scalar.{c}.num_batches_tracked', self._nbt[c], False)

    def _encoders(self):
        return [v for v in vars(self).values() if hasattr(v, 'static_buffers')]

    # -- device placement ------------------------------------------------------------------------------------
    def _materialize(self, device):
        if self._materialized_on == device:
            self._check_master_version()
            return
        _lib.require_gpu()
        init = {n: _get(self, n).detach().to(device=device, dtype=torch.float32) for n in self.arena.entries}
        self.arena.materialize(device, init)
        for n in self.arena.entries:
            _get(self, n).data = self.arena.p(n)
        C = self.in_channels
        rm = torch.stack([_get(self, f'scalar.{c}.running_mean').detach().float() for c in range(C)]).to(device)
        rv = torch.stack([_get(self, f'scalar.{c}.running_var').detach().float() for c in range(C)]).to(device)
        nbt = torch.stack([_get(self, f'scalar.{c}.num_batches_tracked').detach() for c in range(C)]).to(device)
        self._rm, self._rv, self._nbt = rm.contiguous(), rv.contiguous(), nbt.contiguous()
        for c in range(C):
            node = _get(self, f'scalar.{c}')
            node._buffers['running_mean'] = self._rm[c]
            node._buffers['running_var'] = self._rv[c]
            node._buffers['num_batches_tracked'] = self._nbt[c]
        self._materialized_on = device
        self.shadow_trusted = False
        self._params = [_get(self, n) for n in self.arena.entries]
        self._master_sig = None
        self._check_master_version()

    def _master_signature(self):
        # every in-place write torch knows about bumps a version counter: the arena's own (ops on arena.flat) or a
        # parameter's (load_state_dict, torch optimizers, EMA, user edits). The fused AdamW kernel writes master and bf16
        # shadow together through raw pointers and bumps neither.
        return self.arena.flat._version + sum(p._version for p in self._params)

    def _check_master_version(self):
        """The bf16 shadow (and its transposed copies) are valid only for the fp32 master values they were cast from:
        any in-place change since then invalidates them, whoever made it (a sticky 'trusted' flag kept serving
        stale weights after load_state_dict / a torch optimizer step that followed a fused step)."""
        sig = self._master_signature()
        if sig != self._master_sig:
            self.arena.shadow_valid = False
            self.arena.shadow_t_valid = False
            self._master_sig = sig

    def _apply(self, fn, *a, **k):
        # .to()/.cuda() re-create tensors: drop the arena binding, it is rebuilt on the next forward
        self._materialized_on = None
        return super()._apply(fn, *a, **k)

    # -- the BN front shared by every HTS-AT variant -----------------------------------------------------------
    def _bn_front(self, feat, training, overlap=None):
        """Scalar-BatchNorm statistics -> (mean, rstd) and the folded (scale, shift). With a sync-BN group the 2 x 7 x 64 sums are
        all-reduced ASYNCHRONOUSLY (RCCL runs on its own stream) right behind the statistics kernel; `overlap()` - host work and
        launches that do not need the statistics, e.g. drawing the DropPath factors - runs while the collective is on the wire,
        and the compute stream only waits for it in front of the finalize kernel (configs/trainer/gpu.yaml:9)."""
        a, C = self.arena, self.in_channels
        n = C * self.mel_bins
        w = a.flat[a.offsets['scalar.0.weight'][0]: a.offsets['scalar.0.weight'][0] + n]
        b = a.flat[a.offsets['scalar.0.bias'][0]: a.offsets['scalar.0.bias'][0] + n]
        B, _, T, _ = feat.shape
        count = float(B * T)
        sums, centered, work = None, False, None
        if training:
            group = self.sync_bn_group
            centered = group is None
            ops.host_mark('bn: in')
            import os as _os
            if _os.environ.get('PSELD_BN_DROP_FIRST') == '1' and overlap is not None:     # (diagnostic A/B: which call stalls the host)
                overlap(); overlap = None
                ops.host_mark('bn: DropPath factors drawn FIRST')
            sums = ops.bn_scalar_stats(feat, centered=centered)
            ops.host_mark('bn: statistics launched')
            if group is not None:
                import torch.distributed as dist
                work = dist.all_reduce(sums[:2 * n], group=group, async_op=True)
                count *= dist.get_world_size(group)
                ops.host_mark('bn: all-reduce issued')
        if overlap is not None:
            overlap()
            ops.host_mark('bn: DropPath factors drawn')
        if work is not None:
            diag = getattr(self, 'comm_diag', None)               # trainer.comm_diag: events around the wait = the time the compute stream stalled
            if diag is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                work.wait()
                e1.record()
                diag['sync_bn'].append((e0, e1))
            else:
                work.wait()
            ops.host_mark('bn: waited')
        mean_rstd, scale_shift = ops.bn_scalar_finalize(sums, count, centered, w, b, self._rm.view(-1), self._rv.view(-1),
                                                       self._nbt, training, self.bn_momentum, self.bn_eps)
        return mean_rstd, scale_shift

    def _bn_grads(self):
        a, n = self.arena, self.in_channels * self.mel_bins
        ow, ob = a.offsets['scalar.0.weight'][0], a.offsets['scalar.0.bias'][0]
        return a.grad[ow:ow + n], a.grad[ob:ob + n]

    def _check_input(self, x):
        if x.ndim != 4:
            raise ValueError("x shape must be (batch_size, num_channels, time_frames, mel_bins)")
        if not x.is_cuda:
            raise _lib.PseldError("network input must live on the MI355X (no CPU fallback)")
        B, C, T, F = x.shape
        if self.output_frames is None:
            self.output_frames = int(T // self.pred_res)
        if self.output_frames < self.tgt_output_frames:
            raise NotImplementedError('5-second clip pairing (models/accdoa.py:214-219) is not built on the MI355X path')
        elif self.output_frames > self.tgt_output_frames:
            raise NotImplementedError('output_frames > tgt_output_frames is not implemented')
        if C != self.in_channels or F != self.mel_bins:
            raise ValueError(f"expected [B, {self.in_channels}, T, {self.mel_bins}] features, got {tuple(x.shape)}")

    def _drop_scales(self, B, enc, device, training):
        """Per-sample DropPath factors (model_utilities.py:216-232): 0 or 1/keep for every (block, branch)."""
        if not training or enc.cfg['drop_path_rate'] <= 0:
            return None
        keep = getattr(enc, '_keep', None)
        if keep is None or keep.device != device:      # built once: a per-step torch.tensor(...) is a blocking H2D copy
            keep = enc._keep = 1.0 - torch.tensor(enc.rates, device=device, dtype=torch.float32).view(-1, 1, 1)
        u = torch.rand(len(enc.rates), 2, B, device=device)
        return (torch.floor(keep + u) / keep).contiguous()

    def _run(self, x):
        self._check_input(x)
        self._materialize(x.device)           # (re-checks the master version: the shadow follows any in-place weight change)
        x = x.contiguous().float()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            params = [_get(self, n) for n in self.arena.entries]
            return _NetFn.apply(self, x, *params)
        from .htsat import inference_mode
        with inference_mode():                # no backward will follow: the fused blocks skip what they would save for it
            outs, _ = self._forward_impl(x, self.training)
        return outs

    # -- fused optimiser over the arena ------------------------------------------------------------------------
    def zero_grad_arena(self):
        self.arena.grad.zero_()

    def fused_adamw_step(self, lr, max_norm=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, grad_scale=1.0,
                         grad_norm=None, hyper=None):
        """clip_grad_norm_(max_norm) + AdamW over the whole arena in two kernels; refreshes the bf16 shadow.
        hyper: device tensor {lr, bias corrections} of a hipGraph-captured step (trainer.py); the caller then owns arena.step."""
        a = self.arena
        a.ensure_opt_state()
        if hyper is None:
            a.step += 1
        frozen = self._frozen_state()
        if frozen is not None:                 # adapter fine-tuning: frozen parameters take no part in the clip norm or the update
            a.grad.copy_(ops.mul(a.grad, frozen['mask']))
            grad_norm = None                   # a norm taken before the masking would count the frozen entries
        if max_norm and grad_norm is None:
            grad_norm = ops.grad_norm(a.grad)
        shadow = None
        if self.compute_dtype == torch.bfloat16:
            if a.shadow is None:
                a.shadow = torch.empty(a.size, dtype=torch.bfloat16, device=a.flat.device)
            shadow = a.shadow
        ops.adamw_step(a.flat, a.grad, a.m, a.v, a.step, lr, grad_norm_t=grad_norm, max_norm=max_norm or 0.0,
                       grad_scale=grad_scale, betas=betas, eps=eps, weight_decay=weight_decay, shadow=shadow, hyper=hyper)
        if shadow is not None:
            a.shadow_valid = True
            a.shadow_t_valid = False       # the transposed copies follow lazily (arena.wt)
            self.shadow_trusted = True
        if frozen is not None:                 # undo the weight decay on the frozen entries: p = p * mask + p_frozen * (1 - mask)
            a.flat.copy_(ops.axpby(ops.mul(a.flat, frozen['mask']), frozen['keep'], 1.0, 1.0))
            a.shadow_valid = False
        self._master_sig = self._master_signature()
        return grad_norm

    def _frozen_state(self):
        """None when every parameter trains; else {'mask': 1 for trainable arena elements, 'keep': the frozen values * (1 - mask)}.
        Built once per materialisation from the parameters' requires_grad flags (accdoa.py:148-170 freeze_layers_if_needed)."""
        st = getattr(self, '_frozen_cache', None)
        flags = {n: _get(self, n).requires_grad for n in self.arena.entries}
        sig = tuple(flags.values())            # re-built when a caller toggles requires_grad between steps
        if st is not None and st['flat'] is self.arena.flat and st['sig'] == sig:
            return st['state']
        state = None
        if not all(flags.values()):
            mask = torch.zeros_like(self.arena.flat)
            for n, f in flags.items():
                if f:
                    self.arena.view(mask, n, padded=True).fill_(1.0)
            state = {'mask': mask, 'keep': self.arena.flat * (1.0 - mask)}
        self._frozen_cache = {'flat': self.arena.flat, 'sig': sig, 'state': state}
        return state
