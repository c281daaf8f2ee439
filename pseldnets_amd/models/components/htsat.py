"""HTS-AT (Swin) encoder and SELD heads driven on MI355X kernels — forward AND hand-written backward.

Host-side mirror of the reference's `models/components/htsat.py` (HTSAT_Swin_Transformer :385-568,
SwinTransformerBlock :152-268, WindowAttention :53-148, PatchMerging :272-314, BasicLayer :317-381),
`models/components/model_utilities.py` (PatchEmbed :174-213, Mlp :129-171, DropPath :216-242, CrossStitch :35-54)
and the head part of `models/accdoa.py:204-246`. No tensor arithmetic happens in this file: every step is one call
into the C ABI (pseldnets_amd/ops.py). Differences from the reference that are deliberate and invisible in results:
  * tokens stay in natural [B*L, C] order for the whole network; roll / window partition / reverse / patch-merge
    gather / time-frequency fold / token->map reshapes are address maps inside the kernels, never copies;
  * fc1's GEMM epilogue emits gelu(u) AND gelu'(u) (the pre-activation itself is never stored), so erf is evaluated
    once per element for forward + backward; attention probabilities are recomputed in backward instead of stored;
  * eval-mode attention-map averaging (htsat.py:371-377) is skipped — no SELD head consumes it.
"""
import math
import os

import torch

from ... import ops

GELU_DUAL = os.environ.get('PSELD_GELU_DUAL', '1') != '0'
# Fused MLP forward of the C = 192 / 384 blocks (csrc/mlp8f.hip, one launch, the bits of the two launches): PSELD_MLP_PANEL = 0 (off) | 192 |
# 384 | 1 (both). Read once at import (no getenv in a launch path). Round 6, measured: C = 192 (eight waves x 32 rows) 225 against 261-273 us
# per block isolated and 17.62 against 17.72 ms in the step (three alternating same-box runs each, profiles/r06_mlp_panel_ab.txt): ON;
# C = 384 (four waves, 512 registers) 204 against 176 us - one wave per SIMD serialises matrix part, GELU and LDS-DMA issue - and +0.3 ms
# in the step: OFF. docs/EXPERIMENTS.md, round 6.
_MLP_PANEL = os.environ.get('PSELD_MLP_PANEL', '192')


_MLP_PANEL_MB = int(os.environ.get('PSELD_MLP_PANEL_MB', '0'))      # (A/B aid: panel geometry, pseld_mlp_panel_force)
_mlp_panel_forced = [False]


def _mlp_panel(C, M):
    if _MLP_PANEL_MB and not _mlp_panel_forced[0]:
        from ... import _lib
        _lib.lib().pseld_mlp_panel_force(_MLP_PANEL_MB)
        _mlp_panel_forced[0] = True
    return M >= 36864 and (_MLP_PANEL == '1' or _MLP_PANEL == str(C))
# Fused MLP blocks (csrc/mlp.hip: LN2 -> fc1 -> GELU -> fc2 -> DropPath + residual in one kernel, backward by recompute).
# PSELD_FUSED_MLP: comma-separated channel widths that take the fused kernels ('' = none). Default: C = 96 (stage 0 of HTS-AT),
# where the layer-wise chain is HBM-bound and the fused one measures faster; at C = 192 the recompute costs more than the traffic
# it saves (tools/mlp_bench.py), C >= 384 is MFMA-bound and was never a candidate.
FUSED_MLP_WIDTHS = tuple(int(v) for v in os.environ.get('PSELD_FUSED_MLP', '96').split(',') if v.strip())
# Fused attention half of a block (csrc/swin.hip: norm1 -> qkv -> window attention -> proj -> DropPath + shortcut in one kernel; stage 0,
# bf16). PSELD_FUSED_ATTN=0: layer-wise.
FUSED_ATTN = os.environ.get('PSELD_FUSED_ATTN', '1') != '0'
# Input-gradient GEMMs that end in a LayerNorm backward (norm1 <- attn.qkv, norm2 <- mlp.fc1) take pseld_gemm_dgrad_lnbwd where a tile spans the
# row (C = 96 / 192, bf16): one launch instead of GEMM + LayerNorm backward. PSELD_FUSED_LNBWD=0: two launches.
FUSED_LNBWD = os.environ.get('PSELD_FUSED_LNBWD', '1') != '0'
FUSED_ATTN_TAIL = os.environ.get('PSELD_FUSED_ATTN', '1') != 'front'      # 'front': stop in front of proj (A/B of the fused tail)
# A/B knobs of the backward pass, read once at import (nothing in a step reads the environment):
LN_DEFER = os.environ.get('PSELD_LN_DEFER', '1') == '1'            # one reduction per stage for the LayerNorms' d(gamma) / d(beta) partials
# PSELD_WGRAD_GROUP=<blocks per launch> (0: every weight gradient its own launch; 99: the whole stage at once; unset: by the batch, below),
# PSELD_WGRAD_GROUP_MATS=<n>: flush every n matrices
WGRAD_GROUP_BLOCKS = int(os.environ.get('PSELD_WGRAD_GROUP', '-1'))
# Unset: a stage's weight gradients run as ONE launch when the step holds at most this many chunks. The split-K slabs of a single launch are
# 256 workgroups x one fp32 tile whatever the batch: at the reference's 32 chunks (synth_maccdoa.yaml:8) they are 3x the operand bytes, and the
# grouped launch (all tiles of the stage, no token split, no slabs, no reduction launches) wins - 6.22 -> 5.70 ms per step; 64 chunks 7.80 ->
# 7.68; from 96 chunks on the deferral costs more than the slabs (10.31 -> 10.35; 192 chunks: docs/EXPERIMENTS.md)
WGRAD_GROUP_AUTO_MAX_CHUNKS = 64
WGRAD_GROUP_MATS = int(os.environ.get('PSELD_WGRAD_GROUP_MATS', '0')) or (1 << 30)
WGRAD_GROUP_STAGES = tuple(int(v) for v in os.environ.get('PSELD_WGRAD_GROUP_STAGES', '0,1,2,3').split(',') if v.strip())   # stages that group
MLP_DW_FIRST = os.environ.get('PSELD_MLP_DW_FIRST', '1') == '1'
_inference = [False]      # set by the no-grad forward (seld_net._run): nothing is saved for a backward pass


class inference_mode:
    def __enter__(self):
        self.prev, _inference[0] = _inference[0], True

    def __exit__(self, *exc):
        _inference[0] = self.prev

DEFAULTS = dict(spec_size=256, patch_size=4, patch_stride=(4, 4), embed_dim=96, depths=(2, 2, 6, 2),
                num_heads=(4, 8, 16, 32), window_size=8, mlp_ratio=4.0, qkv_bias=True, drop_rate=0.0,
                attn_drop_rate=0.0, drop_path_rate=0.1, ape=False, patch_norm=True, norm_before_mlp='ln')


class SwinEncoder:
    """One HTSAT_Swin_Transformer whose parameters live in `arena` under `prefix` (reference key names)."""

    def __init__(self, arena, prefix, in_chans, mel_bins=64, cfg_adapt=None, **kw):
        cfg = dict(DEFAULTS)
        cfg.update({k: v for k, v in kw.items() if k in DEFAULTS})
        if cfg['window_size'] != 8 or cfg['patch_size'] != 4 or tuple(cfg['patch_stride']) != (4, 4) \
                or cfg['spec_size'] != 256 or mel_bins != 64 or cfg['ape'] or not cfg['patch_norm'] \
                or cfg['norm_before_mlp'] != 'ln' or cfg['drop_rate'] or cfg['attn_drop_rate'] or not cfg['qkv_bias']:
            raise NotImplementedError("the MI355X path is built for the reference's HTS-AT geometry: spec 256, mel 64, "
                                      "patch 4/4, window 8, LayerNorm blocks, no dropout/ape")
        self.cfg, self.arena, self.prefix, self.in_chans = cfg, arena, prefix, in_chans
        self.E = cfg['embed_dim']
        self.depths, self.heads = list(cfg['depths']), list(cfg['num_heads'])
        self.nl = len(self.depths)
        if self.nl != 4:
            raise NotImplementedError("head geometry (8x8 final grid -> [C, 2, 32]) needs 4 stages")
        self.num_features = self.E * 2 ** (self.nl - 1)
        self.time_res = 4 * 2 ** (self.nl - 1)
        self.SF = 2
        self.grid = 64
        self.rates = [v.item() for v in torch.linspace(0, cfg['drop_path_rate'], sum(self.depths))]
        # Adapter fine-tuning (configs/adapt/adapter.yaml; htsat.py:105-110,142-143, model_utilities.py:149-168,
        # model_utilities_adapt.py:7-44): bottleneck MLPs behind attention.proj ('SpatialAdapter') and beside the block MLP
        # ('MlpAdapter'): x -> fc2(gelu(fc1(x))) * adapter_scalar
        ad = dict(cfg_adapt or {})
        method = str(ad.get('method', '') or '')
        # LoRA (configs/adapt/lora.yaml; model_utilities_adapt.py:47-158): every get_linear_layer / get_conv2d_layer site
        # (qkv, proj, fc1, fc2, reduction, patch-embed conv) gets rank-r factors, its base weight is frozen
        self.lora = 'lora' in method
        lk, ck = dict(ad.get('linear_kwargs', {}) or {}), dict(ad.get('conv_kwargs', {}) or {})
        self.lora_r, self.lora_rc = int(lk.get('r', 0)), int(ck.get('r', 0))
        if self.lora:
            if self.lora_r <= 0 or self.lora_rc <= 0 or self.lora_r % 8 or self.lora_rc % 8 or lk.get('lora_dropout', 0.) or lk.get('fan_in_fan_out', False):
                raise NotImplementedError("LoRA: r a positive multiple of 8 for linear and conv layers, no dropout, no fan_in_fan_out")
            self.lora_s, self.lora_sc = lk.get('lora_alpha', 1) / self.lora_r, ck.get('lora_alpha', 1) / self.lora_rc
        self._weff, self._lora_names = {}, []
        akw = dict(ad.get('adapt_kwargs', {}) or {})
        pos = akw.get('position', []) or []
        self.attn_adapter = 'adapter' in method and 'SpatialAdapter' in pos
        self.mlp_adapter = 'adapter' in method and 'MlpAdapter' in pos
        self.adapter_ratio = float(akw.get('mlp_ratio', 0.25))
        self.adapter_scale = akw.get('adapter_scalar', 1)
        self._ad_scale = None
        self.frozen_weights = False            # set by the network when freeze_layers_if_needed froze the backbone weights
        if self.attn_adapter or self.mlp_adapter:
            if akw.get('type', 'adapter') != 'adapter' or akw.get('act_layer', 'gelu') != 'gelu' or akw.get('new_adapter'):
                raise NotImplementedError("adapters: only type=adapter, act_layer=gelu, no new_adapter (configs/adapt/adapter.yaml)")
            if isinstance(self.adapter_scale, str) and self.adapter_scale != 'learnable_scalar':
                raise NotImplementedError(f"adapter_scalar={self.adapter_scale}: a number or 'learnable_scalar' (model_utilities_adapt.py:19-22)")
        self.learn_scale = self.adapter_scale == 'learnable_scalar'
        a, p, E = arena, prefix, self.E

        def lora_entries(base, out_f, in_f):
            if self.lora:
                a.add(base + 'lora_A', (self.lora_r, in_f)); a.add(base + 'lora_B', (out_f, self.lora_r))
                self._lora_names.append(base)
        a.add(p + 'patch_embed.proj.weight', (E, in_chans, 4, 4))
        a.add(p + 'patch_embed.proj.bias', (E,))
        if self.lora:
            a.add(p + 'patch_embed.proj.lora_A.weight', (self.lora_rc, in_chans, 4, 4))
            a.add(p + 'patch_embed.proj.lora_B.weight', (E, self.lora_rc, 1, 1))
        a.add(p + 'patch_embed.norm.weight', (E,))
        a.add(p + 'patch_embed.norm.bias', (E,))
        for li in range(self.nl):
            C, h = E * 2 ** li, self.heads[li]
            hid = int(C * cfg['mlp_ratio'])
            for bi in range(self.depths[li]):
                b = f'{p}layers.{li}.blocks.{bi}.'
                a.add(b + 'norm1.weight', (C,)); a.add(b + 'norm1.bias', (C,))
                a.add(b + 'attn.relative_position_bias_table', (225, h))
                a.add(b + 'attn.qkv.weight', (3 * C, C)); a.add(b + 'attn.qkv.bias', (3 * C,)); lora_entries(b + 'attn.qkv.', 3 * C, C)
                a.add(b + 'attn.proj.weight', (C, C)); a.add(b + 'attn.proj.bias', (C,)); lora_entries(b + 'attn.proj.', C, C)
                ah = int(C * self.adapter_ratio)
                if self.attn_adapter:
                    if self.learn_scale:
                        a.add(b + 'attn.adapter.scale', (1,))
                    a.add(b + 'attn.adapter.fc1.weight', (ah, C)); a.add(b + 'attn.adapter.fc1.bias', (ah,))
                    a.add(b + 'attn.adapter.fc2.weight', (C, ah)); a.add(b + 'attn.adapter.fc2.bias', (C,))
                a.add(b + 'norm2.weight', (C,)); a.add(b + 'norm2.bias', (C,))
                a.add(b + 'mlp.fc1.weight', (hid, C)); a.add(b + 'mlp.fc1.bias', (hid,)); lora_entries(b + 'mlp.fc1.', hid, C)
                a.add(b + 'mlp.fc2.weight', (C, hid)); a.add(b + 'mlp.fc2.bias', (C,)); lora_entries(b + 'mlp.fc2.', C, hid)
                if self.mlp_adapter:
                    if self.learn_scale:
                        a.add(b + 'mlp.adapter.scale', (1,))
                    a.add(b + 'mlp.adapter.fc1.weight', (ah, C)); a.add(b + 'mlp.adapter.fc1.bias', (ah,))
                    a.add(b + 'mlp.adapter.fc2.weight', (C, ah)); a.add(b + 'mlp.adapter.fc2.bias', (C,))
            if li < self.nl - 1:
                d = f'{p}layers.{li}.downsample.'
                a.add(d + 'reduction.weight', (2 * C, 4 * C)); lora_entries(d + 'reduction.', 2 * C, 4 * C)
                a.add(d + 'norm.weight', (4 * C,)); a.add(d + 'norm.bias', (4 * C,))
        a.add(p + 'norm.weight', (self.num_features,)); a.add(p + 'norm.bias', (self.num_features,))

    def static_buffers(self):
        """The reference's non-trainable state-dict entries (htsat.py:79-90 relative_position_index, :203-226
        attn_mask). The kernels recompute both from coordinates; they exist for checkpoint-key compatibility."""
        q = torch.arange(64)
        qy, qx = q // 8, q % 8
        rel = (qy[:, None] - qy[None, :] + 7) * 15 + (qx[:, None] - qx[None, :] + 7)
        out = {}
        for li in range(self.nl):
            _, _, res = self.stage_dims(li)
            for bi in range(self.depths[li]):
                b = f'{self.prefix}layers.{li}.blocks.{bi}.'
                out[b + 'attn.relative_position_index'] = rel.clone()
                if bi % 2 == 1 and res > 8:
                    pos = torch.arange(res)
                    lab1 = (pos >= res - 8).long() + (pos >= res - 4).long()
                    lab = 3 * lab1[:, None] + lab1[None, :]                      # region label of the shifted image
                    win = lab.view(res // 8, 8, res // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
                    diff = win[:, None, :] - win[:, :, None]
                    out[b + 'attn_mask'] = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))
        return out

    # -- helpers ------------------------------------------------------------------------------------------
    def stage_dims(self, li):
        return self.E * 2 ** li, self.heads[li], self.grid // 2 ** li

    def block_index(self, li, bi):
        return sum(self.depths[:li]) + bi

    def first_param_of_layer(self, li):
        return f'{self.prefix}layers.{li}.blocks.0.norm1.weight'

    # -- forward -------------------------------------------------------------------------------------------
    def forward_patch(self, feat, scale_shift, dtype, c_first=0):
        """model_utilities.py:205-213 on bn->pad->fold->patches (htsat.py:547-553 forward_patch)."""
        a, p = self.arena, self.prefix
        self.lora_refresh(dtype)
        A0 = ops.bn_fold_patchify(feat, scale_shift, dtype, c_first, self.in_chans)
        W = self._w(p + 'patch_embed.proj.weight', dtype).view(self.E, self.in_chans * 16)
        P0 = ops.linear_fwd(A0, W, a.p(p + 'patch_embed.proj.bias'))
        x = ops.layernorm_fwd(P0, a.p(p + 'patch_embed.norm.weight'), a.p(p + 'patch_embed.norm.bias'))
        return x, dict(A0=A0, P0=P0, c_first=c_first)

    def backward_patch(self, dx, saved, feat, mean_rstd, bn_dw, bn_db, accumulate_bn):
        a, p = self.arena, self.prefix
        dtype = dx.dtype
        dP0 = ops.layernorm_bwd(dx, saved['P0'], a.p(p + 'patch_embed.norm.weight'), a.g(p + 'patch_embed.norm.weight'),
                                a.g(p + 'patch_embed.norm.bias'))
        if self.frozen_weights:
            ops.colsum(dP0, a.g(p + 'patch_embed.proj.bias'))
        else:
            ops.linear_wgrad(dP0, saved['A0'], a.g(p + 'patch_embed.proj.weight').view(self.E, self.in_chans * 16),
                             dbias=a.g(p + 'patch_embed.proj.bias'))
            if self.lora:
                self._lora_grads(p + 'patch_embed.proj.weight')
        W = self._w(p + 'patch_embed.proj.weight', dtype).view(self.E, self.in_chans * 16)
        dA0 = ops.linear_dgrad(dP0, W)
        ops.bn_scalar_bwd(feat, mean_rstd, dA0, bn_dw, bn_db, saved['c_first'], accumulate=accumulate_bn)

    # -- LoRA: the GEMMs run on the effective weight W + s * B A; its gradient is folded back onto the factors ----------------
    def _w(self, name, dtype):
        return self._weff[name] if self.lora else self.arena.w(name, dtype)

    def _wt(self, name, dtype):
        return None if self.lora else self.arena.wt(name, dtype)

    def _lora_factors(self, wname):
        a = self.arena
        if wname.endswith('patch_embed.proj.weight'):
            base = wname[:-len('weight')]
            K = self.in_chans * 16
            return (a.p(base + 'lora_A.weight').view(self.lora_rc, K), a.p(base + 'lora_B.weight').view(self.E, self.lora_rc),
                    a.g(base + 'lora_A.weight').view(self.lora_rc, K), a.g(base + 'lora_B.weight').view(self.E, self.lora_rc), self.lora_sc, K)
        base = wname[:-len('weight')]
        return a.p(base + 'lora_A'), a.p(base + 'lora_B'), a.g(base + 'lora_A'), a.g(base + 'lora_B'), self.lora_s, a.p(base + 'lora_A').shape[1]

    def lora_refresh(self, dtype):
        """W_eff = W + s * B @ A for every LoRA site (fp32 product, one GEMM with the base weight as its residual), in the
        compute dtype. Equals the reference's unmerged train-mode sum x W^T + s (x A^T) B^T and its merged eval-mode weight."""
        if not self.lora:
            return
        a, p = self.arena, self.prefix
        dev = a.flat.device
        for wname in [p + 'patch_embed.proj.weight'] + [b + 'weight' for b in self._lora_names]:
            A, Bm, _, _, sc, K = self._lora_factors(wname)
            W = a.p(wname).view(Bm.shape[0], K)
            sv = torch.full((1,), float(sc), dtype=torch.float32, device=dev)
            weff = ops.linear_dgrad(Bm.contiguous(), A.contiguous(), rowscale=sv, rows_per_scale=Bm.shape[0], resid=W)
            if dtype == torch.bfloat16:
                wb = torch.empty(weff.numel(), dtype=torch.bfloat16, device=dev)
                ops.cast_bf16(weff.view(-1), wb)
                weff = wb.view(weff.shape)
            self._weff[wname] = weff.view(a.p(wname).shape) if not wname.endswith('patch_embed.proj.weight') else weff

    def _lora_grads(self, wname):
        """dW_eff sits in the (frozen) base weight's gradient slot: dA = s * B^T dW_eff, dB = s * dW_eff A^T."""
        A, Bm, dA, dB, sc, K = self._lora_factors(wname)
        G = self.arena.g(wname).view(Bm.shape[0], K)
        sv = torch.full((1,), float(sc), dtype=torch.float32, device=G.device)
        ops.linear_wgrad(Bm.contiguous(), G, dA, rowscale=sv, rows_per_scale=Bm.shape[0])
        dB.copy_(ops.linear_fwd(G, A.contiguous(), rowscale=sv, rows_per_scale=G.shape[0]))

    def _wgrad(self, dy, x, wname, bname=None, gelu_on_x=False, rowscale=None, rows_per_scale=1, per_scale_elems=0):
        """dW (+ dbias). With the backbone frozen (adapter fine-tuning) only the bias gradient is formed: a column sum of the
        (DropPath-scaled) output gradient instead of the split-K weight-gradient GEMM."""
        a = self.arena
        if not self.frozen_weights:
            grp = getattr(self, '_wgroup', None)
            if grp is not None and not self.lora and not gelu_on_x and dy.dtype == torch.bfloat16:
                # deferred: the stage's weight gradients run as ONE persistent launch at the end of the stage (backward_layer) - together
                # their output tiles fill the chip without splitting the tokens, so no fp32 slabs are written or reduced
                grp.append((dy, x, a.g(wname), a.g(bname) if bname else None, rowscale, rows_per_scale))
                if len(grp) >= getattr(self, '_wgroup_mats', 1 << 30):         # (PSELD_WGRAD_GROUP_MATS: flush every n matrices)
                    self._flush_wgroup()
                return
            if not self.lora and getattr(self, '_side_ok', False):
                # on the second stream, beside the input-gradient / attention / LayerNorm chain (ops.linear_wgrad_side; joined at
                # the end of the stage in backward_layer)
                ops.linear_wgrad_side(dy, x, a.g(wname), dbias=a.g(bname) if bname else None, gelu_on_x=gelu_on_x, rowscale=rowscale,
                                      rows_per_scale=rows_per_scale)
                return
            ops.linear_wgrad(dy, x, a.g(wname), dbias=a.g(bname) if bname else None, gelu_on_x=gelu_on_x, rowscale=rowscale,
                             rows_per_scale=rows_per_scale)
            if self.lora:
                self._lora_grads(wname)
        elif bname is not None:
            ops.colsum(dy if rowscale is None else ops.rowscale(dy, rowscale, per_scale_elems), a.g(bname))

    # -- Adapter (model_utilities_adapt.py:7-44): fc2(gelu(fc1(x))) * scale (+ resid) --------------------------------------
    def _scale_vec(self, device, pre=None):
        if self.learn_scale:
            return self.arena.p(pre + 'scale')                 # nn.Parameter(torch.ones(1)) of this adapter
        if self._ad_scale is None or self._ad_scale.device != device:
            self._ad_scale = torch.full((1,), float(self.adapter_scale), dtype=torch.float32, device=device)
        return self._ad_scale

    def _adapter_fwd(self, x, pre, resid=None):
        a, dtype, M = self.arena, x.dtype, x.shape[0]
        h, g = ops.linear_fwd(x, a.w(pre + 'fc1.weight', dtype), a.p(pre + 'fc1.bias'), gelu_dual=True)
        y = ops.linear_fwd(h, a.w(pre + 'fc2.weight', dtype), a.p(pre + 'fc2.bias'), resid=resid, rowscale=self._scale_vec(x.device, pre),
                           rows_per_scale=M)
        return y, dict(h=h, g=g)

    def _adapter_bwd(self, dy, x, sv, pre, dresid=None):
        """dy = gradient of the adapter output; returns the gradient wrt its input (+ dresid for the 'adapter(x) + x' form)."""
        a, dtype, M = self.arena, dy.dtype, dy.shape[0]
        sc = self._scale_vec(dy.device, pre)
        ops.linear_wgrad(dy, sv['h'], a.g(pre + 'fc2.weight'), dbias=a.g(pre + 'fc2.bias'), rowscale=sc, rows_per_scale=M)
        if self.learn_scale:
            # d/ds of (fc2(h) * s): <dy, fc2(h)> = (<dW2, W2> + <db2, b2>) / s with the s-scaled gradients just written
            ops.dot_div(a.g(pre + 'fc2.weight'), a.p(pre + 'fc2.weight'), sc, a.g(pre + 'scale'))
            ops.dot_div(a.g(pre + 'fc2.bias'), a.p(pre + 'fc2.bias'), sc, a.g(pre + 'scale'), accumulate=True)
        dh = ops.linear_dgrad(dy, a.w(pre + 'fc2.weight', dtype), wt=a.wt(pre + 'fc2.weight', dtype), mul=sv['g'], rowscale=sc, rows_per_scale=M)
        ops.linear_wgrad(dh, x, a.g(pre + 'fc1.weight'), dbias=a.g(pre + 'fc1.bias'))
        return ops.linear_dgrad(dh, a.w(pre + 'fc1.weight', dtype), wt=a.wt(pre + 'fc1.weight', dtype), resid=dresid)

    def _mlp_fused(self, x, L):
        """The block MLP runs on the fused kernels: full fine-tuning (no adapters / LoRA / frozen weights), a width they were built
        and enabled for, whole 32-token tiles per sample."""
        return (x.shape[1] in FUSED_MLP_WIDTHS and not (self.mlp_adapter or self.lora or self.frozen_weights)
                and ops.mlp_fused_supported(x, L))

    def forward_layer(self, li, x, B, drop_scale=None):
        """BasicLayer li (htsat.py:364-378): its blocks, then PatchMerging. drop_scale: f32[n_blocks_total, 2, B]."""
        a, p, dtype = self.arena, self.prefix, x.dtype
        C, heads, res = self.stage_dims(li)
        L = res * res
        saved_blocks = []
        for bi in range(self.depths[li]):
            b = f'{p}layers.{li}.blocks.{bi}.'
            gi = self.block_index(li, bi)
            shift = 0 if (bi % 2 == 0 or res <= 8) else 4
            s1 = s2 = None
            if drop_scale is not None and self.rates[gi] > 0:
                s1, s2 = drop_scale[gi, 0], drop_scale[gi, 1]
            x_mid = None
            if FUSED_ATTN and FUSED_ATTN_TAIL and not self.attn_adapter and ops.swin_attn_fused_supported(x, res, heads):
                # the whole attention half - norm1 -> qkv -> window attention -> proj -> DropPath + shortcut - in ONE kernel (csrc/swin.hip); it
                # leaves the same saved operands as the four launches (nothing but x_mid in a no-grad forward)
                x_mid, ao, qkv, xh1, lse = ops.swin_block_attn_fwd(x, a.p(b + 'norm1.weight'), a.p(b + 'norm1.bias'), self._w(b + 'attn.qkv.weight', dtype),
                                                                   a.p(b + 'attn.qkv.bias'), a.p(b + 'attn.relative_position_bias_table'),
                                                                   self._w(b + 'attn.proj.weight', dtype), a.p(b + 'attn.proj.bias'), B, res, heads, shift,
                                                                   rowscale=s1, need_saved=not _inference[0])
            elif FUSED_ATTN and ops.swin_attn_fused_supported(x, res, heads):
                # norm1 -> qkv -> window attention in ONE kernel; the adapter branch follows layer-wise
                ao, qkv, xh1, lse = ops.swin_attn_fwd(x, a.p(b + 'norm1.weight'), a.p(b + 'norm1.bias'), self._w(b + 'attn.qkv.weight', dtype),
                                                      a.p(b + 'attn.qkv.bias'), a.p(b + 'attn.relative_position_bias_table'), B, res, heads,
                                                      shift, need_saved=not _inference[0])
            else:
                xh1 = ops.layernorm_fwd(x, a.p(b + 'norm1.weight'), a.p(b + 'norm1.bias'))
                qkv = ops.linear_fwd(xh1, self._w(b + 'attn.qkv.weight', dtype), a.p(b + 'attn.qkv.bias'))
                ao, lse = ops.window_attn_fwd(qkv, a.p(b + 'attn.relative_position_bias_table'), B, res, heads, shift)
            ad = {}
            if x_mid is not None:
                pass
            elif self.attn_adapter:
                # x = adapter(proj(attn)) + proj(attn) (htsat.py:141-143), then the block's DropPath + residual
                a0 = ops.linear_fwd(ao, self._w(b + 'attn.proj.weight', dtype), a.p(b + 'attn.proj.bias'))
                a1, ad['attn'] = self._adapter_fwd(a0, b + 'attn.adapter.', resid=a0)
                ad['a0'] = a0
                x_mid = ops.add(x, ops.rowscale(a1, s1, L * C)) if s1 is not None else ops.add(x, a1)
            else:
                x_mid = ops.linear_fwd(ao, self._w(b + 'attn.proj.weight', dtype), a.p(b + 'attn.proj.bias'), resid=x,
                                       rowscale=s1, rows_per_scale=L)
            if self._mlp_fused(x_mid, L):
                # x_out = x_mid + s2 * (fc2(gelu(fc1(LN2(x_mid)))) + b2) in ONE kernel; the backward recomputes the hidden activations
                # from xh2 = LN2(x_mid), the only activation saved besides the residual stream
                x_out, xh2 = ops.mlp_fwd(x_mid, a.p(b + 'norm2.weight'), a.p(b + 'norm2.bias'), self._w(b + 'mlp.fc1.weight', dtype),
                                         a.p(b + 'mlp.fc1.bias'), self._w(b + 'mlp.fc2.weight', dtype), a.p(b + 'mlp.fc2.bias'),
                                         rowscale=s2, rows_per_scale=L, need_xh=not _inference[0])
                saved_blocks.append(dict(x_in=x, xh1=xh1, qkv=qkv, ao=ao, lse=lse, x_mid=x_mid, xh2=xh2, fused_mlp=True, s1=s1, s2=s2,
                                         shift=shift, ad=ad))
                x = x_out
                continue
            xh2 = ops.layernorm_fwd(x_mid, a.p(b + 'norm2.weight'), a.p(b + 'norm2.bias'))
            if self.mlp_adapter:
                xs, ad['mlp'] = self._adapter_fwd(xh2, b + 'mlp.adapter.')          # xs = adapter(x) (model_utilities.py:160-170)
            if GELU_DUAL and _mlp_panel(C, x_mid.shape[0]) and ops.mlp_panel_fwd_supported(xh2, 4 * C) and not self.lora:
                # fc1 -> GELU pair -> fc2 -> DropPath + shortcut in ONE launch (csrc/mlp8f.hip): the bits of the two launches below, h never re-read
                x_out, hact, gact = ops.mlp_panel_fwd(xh2, self._w(b + 'mlp.fc1.weight', dtype), a.p(b + 'mlp.fc1.bias'), self._w(b + 'mlp.fc2.weight', dtype),
                                                      a.p(b + 'mlp.fc2.bias'), x_mid, rowscale=s2, rows_per_scale=L)
                if self.mlp_adapter:
                    x_out = ops.add(x_out, ops.rowscale(xs, s2, L * C) if s2 is not None else xs)
                saved_blocks.append(dict(x_in=x, xh1=xh1, qkv=qkv, ao=ao, lse=lse, x_mid=x_mid, xh2=xh2, h=hact, g=gact, s1=s1,
                                         s2=s2, shift=shift, ad=ad))
            elif GELU_DUAL:
                # fc1 epilogue emits h = gelu(u) and g = gelu'(u): erf is evaluated once per element, not in fc2/dW2/dU
                hact, gact = ops.linear_fwd(xh2, self._w(b + 'mlp.fc1.weight', dtype), a.p(b + 'mlp.fc1.bias'), gelu_dual=True)
                x_out = ops.linear_fwd(hact, self._w(b + 'mlp.fc2.weight', dtype), a.p(b + 'mlp.fc2.bias'), resid=x_mid,
                                       rowscale=s2, rows_per_scale=L)
                if self.mlp_adapter:
                    x_out = ops.add(x_out, ops.rowscale(xs, s2, L * C) if s2 is not None else xs)
                saved_blocks.append(dict(x_in=x, xh1=xh1, qkv=qkv, ao=ao, lse=lse, x_mid=x_mid, xh2=xh2, h=hact, g=gact, s1=s1,
                                         s2=s2, shift=shift, ad=ad))
            else:
                u = ops.linear_fwd(xh2, self._w(b + 'mlp.fc1.weight', dtype), a.p(b + 'mlp.fc1.bias'))
                x_out = ops.linear_fwd(u, self._w(b + 'mlp.fc2.weight', dtype), a.p(b + 'mlp.fc2.bias'), resid=x_mid,
                                       rowscale=s2, rows_per_scale=L, gelu_in=True)
                if self.mlp_adapter:
                    x_out = ops.add(x_out, ops.rowscale(xs, s2, L * C) if s2 is not None else xs)
                saved_blocks.append(dict(x_in=x, xh1=xh1, qkv=qkv, ao=ao, lse=lse, x_mid=x_mid, xh2=xh2, u=u, s1=s1, s2=s2,
                                         shift=shift, ad=ad))
            x = x_out
        saved = dict(blocks=saved_blocks)
        if li < self.nl - 1:
            d = f'{p}layers.{li}.downsample.'
            xm = ops.layernorm_fwd(x, a.p(d + 'norm.weight'), a.p(d + 'norm.bias'), merge_res=res)
            saved.update(x_pre=x, xm=xm)
            x = ops.linear_fwd(xm, self._w(d + 'reduction.weight', dtype))
        return x, saved

    def backward_layer(self, li, dx, saved, B):
        # the second stream pays from ~64 chunks per step on: at the reference's batch of 32 the ~50 forks per step cost more host
        # time than the overlap returns (7.26 against 7.55 ms per step measured)
        self._side_ok = ops.wgrad_side_enabled(dx.device, B)
        rpb = self._rpb_state(dx.device)
        if li == self.nl - 1:
            rpb['acc'].zero_()                # the backward starts at the last stage: every block's accumulator, one launch
        if not LN_DEFER:
            self._defer = None                                       # (A/B knob: every LayerNorm reduces its own partials)
        elif getattr(self, '_defer', None) is None or self._defer.buf.device != dx.device:
            self._defer = ops.DeferredReductions(dx.device)      # d(gamma) / d(beta) partials of the stage's LayerNorms: one reduction
        # WGRAD_GROUP_BLOCKS (PSELD_WGRAD_GROUP). The grouped launch is 16-29 % faster than its members one by one in isolation (tools/wgrad8_check.py group: no token split, no
        # fp32 slabs) and SLOWER inside the 192-chunk step (same box, 30 steps: 18.74 ms ungrouped, 19.25 / 19.00 / 19.06 ms with 1 / 2 blocks / the
        # whole stage per launch): a deferred weight gradient no longer runs beside the input-gradient kernel that reads the same dY, and
        # one chip-wide persistent launch leaves the second stream nothing to interleave. Small batches are the other way round (top of the file)
        self._wgroup_blocks = WGRAD_GROUP_BLOCKS if WGRAD_GROUP_BLOCKS >= 0 else (99 if B <= WGRAD_GROUP_AUTO_MAX_CHUNKS else 0)
        self._wgroup_mats = WGRAD_GROUP_MATS
        self._wgroup = [] if (dx.dtype == torch.bfloat16 and self._wgroup_blocks > 0 and li in WGRAD_GROUP_STAGES) else None
        self._wgroup_n = 0
        dx = self._backward_layer(li, dx, saved, B)
        self._flush_wgroup()
        self._wgroup = None
        if self._defer is not None:
            self._defer.flush()
        ops.bias_table_grad_batched(rpb['acc'], self.arena.grad, rpb['desc'][li], self.depths[li], self.stage_dims(li)[1])
        ops.join_wgrads(dx.device, final=False)      # the stage's weight gradients are complete before its gradient range is all-reduced (skipped without a group)
        return dx

    def _flush_wgroup(self):
        """The deferred weight gradients (see _wgrad) as one persistent launch, on the second stream when that is on."""
        if self._wgroup:
            if self._side_ok:
                ops.linear_wgrad_group_side(self._wgroup)
            else:
                ops.linear_wgrad_group(self._wgroup)
            self._wgroup = []
        self._wgroup_n = 0

    def _rpb_state(self, device):
        """Accumulators of d(relative_position_bias_table) for every block (fp32 [heads][64 keys][64 queries] each) and, per stage, the
        descriptors pseld_bias_table_grad_batched reads: {accumulator offset, offset of the table's gradient in the arena, heads}."""
        st = getattr(self, '_rpb', None)
        if st is not None and st['acc'].device == device and st['grad'] is self.arena.grad:
            return st
        off, desc, total = {}, [], 0
        for li in range(self.nl):
            heads = self.stage_dims(li)[1]
            rows = []
            for bi in range(self.depths[li]):
                off[(li, bi)] = total
                name = f'{self.prefix}layers.{li}.blocks.{bi}.attn.relative_position_bias_table'
                rows += [total, self.arena.offsets[name][0], heads]
                total += heads * 4096
            desc.append(torch.tensor(rows, dtype=torch.long, device=device))
        self._rpb = dict(acc=torch.zeros(total, dtype=torch.float32, device=device), desc=desc, off=off, grad=self.arena.grad)
        return self._rpb

    def _backward_layer(self, li, dx, saved, B):
        a, p, dtype = self.arena, self.prefix, dx.dtype
        C, heads, res = self.stage_dims(li)
        L = res * res
        if li < self.nl - 1:
            d = f'{p}layers.{li}.downsample.'
            self._wgrad(dx, saved['xm'], d + 'reduction.weight')
            dxm = ops.linear_dgrad(dx, self._w(d + 'reduction.weight', dtype), wt=self._wt(d + 'reduction.weight', dtype))
            dx = ops.layernorm_bwd(dxm, saved['x_pre'], a.p(d + 'norm.weight'), a.g(d + 'norm.weight'),
                                   a.g(d + 'norm.bias'), merge_res=res, defer=self._defer)
        for bi in reversed(range(self.depths[li])):
            b = f'{p}layers.{li}.blocks.{bi}.'
            s = saved['blocks'][bi]
            # ---- MLP branch:  x_out = x_mid + s2 * (fc2(gelu(u)) + b2) ------------------------------------
            if s.get('fused_mlp'):
                w1, w2 = self._w(b + 'mlp.fc1.weight', dtype), self._w(b + 'mlp.fc2.weight', dtype)
                w1t, w2t = self._wt(b + 'mlp.fc1.weight', dtype), self._wt(b + 'mlp.fc2.weight', dtype)
                if w1t is None:                                   # f32 (parity) mode keeps no transposed copies
                    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
                dwargs = (s['xh2'], dx, w1, a.p(b + 'mlp.fc1.bias'), w2t, a.g(b + 'mlp.fc1.weight'), a.g(b + 'mlp.fc1.bias'),
                          a.g(b + 'mlp.fc2.weight'), a.g(b + 'mlp.fc2.bias'))
                # the weight-gradient kernel is forked onto the second stream BEFORE dx is launched: the earlier the side chain of the
                # block starts the better (A/B PSELD_MLP_DW_FIRST=0 - fork behind dx, so that dw runs beside the HBM-bound kernels that
                # follow instead of beside the equally VALU-bound dx - measured 21.33 against 20.99 ms per step)
                dw_first = MLP_DW_FIRST
                if not dw_first:
                    dxh2 = ops.mlp_bwd_dx(s['xh2'], dx, w1, a.p(b + 'mlp.fc1.bias'), w2t, w1t, rowscale=s['s2'], rows_per_scale=L)
                if getattr(self, '_side_ok', False):
                    ops.mlp_bwd_dw_side(*dwargs, rowscale=s['s2'], rows_per_scale=L)     # the four parameter gradients, second stream
                else:
                    ops.mlp_bwd_dw(*dwargs, rowscale=s['s2'], rows_per_scale=L)
                if dw_first:
                    dxh2 = ops.mlp_bwd_dx(s['xh2'], dx, w1, a.p(b + 'mlp.fc1.bias'), w2t, w1t, rowscale=s['s2'], rows_per_scale=L)
            elif 'h' in s:
                self._wgrad(dx, s['h'], b + 'mlp.fc2.weight', b + 'mlp.fc2.bias', rowscale=s['s2'], rows_per_scale=L, per_scale_elems=L * C)
                du = ops.linear_dgrad(dx, self._w(b + 'mlp.fc2.weight', dtype), wt=self._wt(b + 'mlp.fc2.weight', dtype), mul=s['g'], rowscale=s['s2'], rows_per_scale=L)
            else:
                self._wgrad(dx, s['u'], b + 'mlp.fc2.weight', b + 'mlp.fc2.bias', gelu_on_x=True, rowscale=s['s2'], rows_per_scale=L, per_scale_elems=L * C)
                du = ops.linear_dgrad(dx, self._w(b + 'mlp.fc2.weight', dtype), wt=self._wt(b + 'mlp.fc2.weight', dtype), gelu_grad_of=s['u'], rowscale=s['s2'], rows_per_scale=L)
            if not s.get('fused_mlp'):
                self._wgrad(du, s['xh2'], b + 'mlp.fc1.weight', b + 'mlp.fc1.bias')
                dxh2_ad = None
                if self.mlp_adapter:            # the adapter branch sees the same DropPath-scaled gradient
                    dxs = ops.rowscale(dx, s['s2'], L * C) if s['s2'] is not None else dx
                    dxh2_ad = self._adapter_bwd(dxs, s['xh2'], s['ad']['mlp'], b + 'mlp.adapter.')
                w1t = self._wt(b + 'mlp.fc1.weight', dtype)
                if FUSED_LNBWD and dxh2_ad is None and w1t is not None and ops.dgrad_lnbwd_supported(du, C):
                    dxh2 = None               # fc1's input gradient and norm2's backward in one launch
                    dx_mid = ops.linear_dgrad_lnbwd(du, w1t, s['x_mid'], a.p(b + 'norm2.weight'), a.g(b + 'norm2.weight'), a.g(b + 'norm2.bias'),
                                                    dres=dx, defer=self._defer)
                else:
                    dxh2 = ops.linear_dgrad(du, self._w(b + 'mlp.fc1.weight', dtype), wt=w1t, resid=dxh2_ad)
            if dxh2 is not None:
                dx_mid = ops.layernorm_bwd(dxh2, s['x_mid'], a.p(b + 'norm2.weight'), a.g(b + 'norm2.weight'),
                                           a.g(b + 'norm2.bias'), dres=dx, defer=self._defer)
            # ---- attention branch:  x_mid = x_in + s1 * (proj(attn(qkv)) + bp) ---------------------------------
            if self.attn_adapter:
                da1 = ops.rowscale(dx_mid, s['s1'], L * C) if s['s1'] is not None else dx_mid
                da0 = self._adapter_bwd(da1, s['ad']['a0'], s['ad']['attn'], b + 'attn.adapter.', dresid=da1)
                self._wgrad(da0, s['ao'], b + 'attn.proj.weight', b + 'attn.proj.bias')
                dao = ops.linear_dgrad(da0, self._w(b + 'attn.proj.weight', dtype), wt=self._wt(b + 'attn.proj.weight', dtype))
            else:
                self._wgrad(dx_mid, s['ao'], b + 'attn.proj.weight', b + 'attn.proj.bias', rowscale=s['s1'], rows_per_scale=L, per_scale_elems=L * C)
                dao = None
                wpt = self._wt(b + 'attn.proj.weight', dtype)
                if not (FUSED_ATTN and FUSED_ATTN_TAIL and wpt is not None and ops.swin_block_attn_bwd_supported(s['qkv'], res, heads)):
                    dao = ops.linear_dgrad(dx_mid, self._w(b + 'attn.proj.weight', dtype), wt=self._wt(b + 'attn.proj.weight', dtype), rowscale=s['s1'], rows_per_scale=L)
            # d(relative_position_bias_table): the block leaves its [heads][64][64] sums in its own accumulator; one launch per stage
            # turns them into the table gradients (backward_layer)
            o = self._rpb['off'][(li, bi)]
            if dao is None:
                # stage 0: the projection's input gradient is formed inside the attention backward (csrc/attn.hip: attn_bwd24_kernel<true>)
                dqkv = ops.swin_block_attn_bwd(s['qkv'], a.p(b + 'attn.relative_position_bias_table'), s['ao'], s['lse'], dx_mid,
                                               wpt, None, B, res, heads, s['shift'], rowscale=s['s1'],
                                               acc=self._rpb['acc'][o:o + heads * 4096])
            else:
                dqkv = ops.window_attn_bwd(s['qkv'], a.p(b + 'attn.relative_position_bias_table'), s['ao'], s['lse'], dao,
                                           None, B, res, heads, s['shift'], acc=self._rpb['acc'][o:o + heads * 4096])
            self._wgrad(dqkv, s['xh1'], b + 'attn.qkv.weight', b + 'attn.qkv.bias')
            wqt = self._wt(b + 'attn.qkv.weight', dtype)
            if FUSED_LNBWD and wqt is not None and ops.dgrad_lnbwd_supported(dqkv, C):
                # qkv's input gradient and norm1's backward (+ the shortcut's gradient) in one launch
                dx = ops.linear_dgrad_lnbwd(dqkv, wqt, s['x_in'], a.p(b + 'norm1.weight'), a.g(b + 'norm1.weight'), a.g(b + 'norm1.bias'),
                                            dres=dx_mid, defer=self._defer)
            else:
                dxh1 = ops.linear_dgrad(dqkv, self._w(b + 'attn.qkv.weight', dtype), wt=wqt)
                dx = ops.layernorm_bwd(dxh1, s['x_in'], a.p(b + 'norm1.weight'), a.g(b + 'norm1.weight'),
                                       a.g(b + 'norm1.bias'), dres=dx_mid, defer=self._defer)
            if getattr(self, '_wgroup', None) is not None:
                self._wgroup_n += 1
                if self._wgroup_n >= self._wgroup_blocks:
                    self._flush_wgroup()
        return dx

    def forward_final(self, x):
        """htsat.py:523-525 final LayerNorm (the token->map reshape is folded into the head's im2col)."""
        a, p = self.arena, self.prefix
        xn = ops.layernorm_fwd(x, a.p(p + 'norm.weight'), a.p(p + 'norm.bias'))
        return xn, dict(x_last=x)

    def backward_final(self, dxn, saved):
        a, p = self.arena, self.prefix
        return ops.layernorm_bwd(dxn, saved['x_last'], a.p(p + 'norm.weight'), a.g(p + 'norm.weight'), a.g(p + 'norm.bias'))


class TscamHead:
    """accdoa.py:230-242: Conv2d(C, D, (SF, 3), pad (0,1)) + interpolate/crop/mean (+ tanh) on the final tokens."""

    def __init__(self, arena, prefix, in_features, out_dim, act_tanh):
        self.arena, self.prefix, self.C, self.D, self.act = arena, prefix, in_features, out_dim, act_tanh
        self.Dp = (out_dim + 7) // 8 * 8
        arena.add(prefix + 'weight', (out_dim, in_features, 2, 3), pad_rows=self.Dp)
        arena.add(prefix + 'bias', (out_dim,), pad_rows=self.Dp)
        self.taps = None

    def _taps(self, device):
        if self.taps is None or self.taps['i0'].device != device:
            self.taps = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in ops.pool_taps().items()}
        return self.taps

    def forward(self, xn, B):
        a, p, dtype = self.arena, self.prefix, xn.dtype
        A = ops.head_im2col(xn, B)
        W = a.w(p + 'weight', dtype, padded=True).view(self.Dp, self.C * 6)
        z = ops.linear_fwd(A, W, a.p(p + 'bias', padded=True))
        y = ops.head_pool_fwd(z, self._taps(xn.device), B, self.D, self.act)
        return y, dict(A=A, y=y)

    def backward(self, dy, saved, B, dtype, accumulate_into=None):
        a, p = self.arena, self.prefix
        dz = ops.head_pool_bwd(dy.contiguous(), saved['y'], self._taps(dy.device), B, self.D, self.Dp, dtype, self.act)
        ops.linear_wgrad(dz, saved['A'], a.g(p + 'weight', padded=True).view(self.Dp, self.C * 6),
                         dbias=a.g(p + 'bias', padded=True))
        W = a.w(p + 'weight', dtype, padded=True).view(self.Dp, self.C * 6)
        dA = ops.linear_dgrad(dz, W, wt=a.wt(p + 'weight', dtype, padded=True))
        return ops.head_col2im(dA, B)


def default_init(name, shape):
    """Initial values distributed like the reference constructors' defaults (nn.Linear / nn.Conv2d kaiming-uniform,
    LayerNorm/BatchNorm ones/zeros, trunc_normal(0.02) bias table, CrossStitch U(0.1, 0.9))."""
    t = torch.empty(shape)
    leaf = name.rsplit('.', 1)[-1]
    if 'relative_position_bias_table' in name:
        torch.nn.init.trunc_normal_(t, std=.02)
    elif leaf == 'in_proj_weight':                                       # nn.MultiheadAttention: xavier_uniform
        torch.nn.init.xavier_uniform_(t)
    elif leaf == 'in_proj_bias' or name.endswith('out_proj.bias'):
        t.zero_()
    elif leaf.startswith(('weight_ih_l', 'weight_hh_l', 'bias_ih_l', 'bias_hh_l')):   # nn.GRU: U(-1/sqrt(H), 1/sqrt(H))
        k = 1.0 / math.sqrt(shape[0] / 3)
        t.uniform_(-k, k)
    elif 'lora_B' in name:                                               # model_utilities_adapt.py:99-100,147-148: LoRA starts as identity
        t.zero_()
    elif 'lora_A' in name:
        torch.nn.init.kaiming_uniform_(t.view(shape[0], -1), a=math.sqrt(5))
    elif name.endswith('.adapter.scale'):                               # model_utilities_adapt.py:20: nn.Parameter(torch.ones(1))
        t.fill_(1.0)
    elif '.adapter.fc2.' in name:                                       # model_utilities_adapt.py:26-30: the adapter starts as identity
        t.zero_()
    elif name.startswith('stitch'):
        t.uniform_(0.1, 0.9)
    elif '.sequential.' in name and 'layers.' in name:                 # Conformer decoder (conformer/modules.py:41-47, attention.py:69-70)
        if len(shape) == 1 and '.conv.bias' in name:
            fan_in = shape[0] // 2 if '.2.conv.' in name else shape[0]
            t.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        elif len(shape) == 1:
            t.fill_(1.0 if leaf == 'weight' else 0.0)
        elif '.conv.weight' in name:
            torch.nn.init.kaiming_uniform_(t, a=math.sqrt(5))
        else:
            torch.nn.init.xavier_uniform_(t)
    elif 'token' in name or 'pos_embed' in name:                       # passt.py:139-149,201-207
        torch.nn.init.trunc_normal_(t, std=.02)
    elif 'norm' in name or name.startswith('scalar') or '.head.0.' in name or '.bn1.' in name or '.bn2.' in name:
        t.fill_(1.0 if leaf == 'weight' else 0.0)
    elif leaf == 'weight':
        torch.nn.init.kaiming_uniform_(t.view(shape[0], -1), a=math.sqrt(5))
    else:  # bias of a Linear/Conv: U(-1/sqrt(fan_in), 1/sqrt(fan_in)); fan_in is not known here -> small uniform
        t.uniform_(-0.02, 0.02)
    return t
