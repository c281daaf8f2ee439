"""Convolutional encoder of the CRNN networks driven on MI355X kernels — forward AND hand-written backward.

Host-side mirror of the reference's `models/components/backbone.py` (CNN8 :6-31, CNN12 :33-60 = the conv stack of PANNs
CNN14) and `models/components/model_utilities.py` ConvBlock :92-126. Activations are NHWC rows [B*T*F, C]; every 3x3
convolution is im2col (tap-major columns, 16-byte copies) -> MFMA GEMM against a tap-major copy of the reference's
[Cout, Cin, 3, 3] weight, its input gradient GEMM -> col2im, its weight gradient the split-K GEMM over the im2col
matrix kept from the forward pass (HBM is large: 0.2 GB per chunk), permuted back into the reference layout. The
im2col matrix is built per slice of the batch. BatchNorm2d (train-mode batch statistics) + ReLU and the
average pools are per-column / per-pixel kernels (csrc/cnn.hip). No tensor arithmetic happens in this file.
"""
import os

import torch

from ... import ops

CONV_SLICE_ELEMS = 1 << 28      # im2col elements per batch slice (512 MB in bf16)


class ConvEncoder:
    """CNN8 / CNN12 whose parameters live in `arena` under `prefix` (reference key names)."""

    def __init__(self, arena, prefix, in_chans, kind, num_features):
        if kind == 'CNN8':
            pools = [(2, 2), (2, 2), (2, 2), (1, 2)]
        elif kind == 'CNN12':
            pools = [(2, 2), (2, 2), (2, 2), (1, 2), (1, 2), (1, 2)]
        else:
            raise NotImplementedError(f'encoder {kind} is not implemented')
        if len(num_features) != len(pools):
            raise ValueError(f'{kind} needs {len(pools)} feature widths')
        self.arena, self.prefix, self.in_chans, self.pools = arena, prefix, in_chans, pools
        self.cin_p = (in_chans + 7) // 8 * 8                   # input channels padded so that 9*Cin is a multiple of 8
        self.widths = list(num_features)
        self.num_features = self.widths[-1]
        self.bn_buffers = {}
        self.keep_im2col = True            # training: keep the forward im2col matrices for the weight-gradient GEMMs
        # layers with at most this many input channels run as implicit convolutions (no im2col / col2im at all): they are
        # the HBM-bound ones; the deep layers keep the explicit matrix and the LDS-DMA GEMM feeder
        self.implicit_max_channels = int(os.environ.get('PSELD_CONV_IMPLICIT_MAXC', 256))
        cin = in_chans
        for i, cout in enumerate(self.widths):
            b = f'{prefix}conv_block{i + 1}.'
            arena.add(b + 'conv1.weight', (cout, cin, 3, 3))
            arena.add(b + 'bn1.weight', (cout,)); arena.add(b + 'bn1.bias', (cout,))
            arena.add(b + 'conv2.weight', (cout, cout, 3, 3))
            arena.add(b + 'bn2.weight', (cout,)); arena.add(b + 'bn2.bias', (cout,))
            cin = cout

    def static_buffers(self):
        out = {}
        for i, cout in enumerate(self.widths):
            for j in (1, 2):
                b = f'{self.prefix}conv_block{i + 1}.bn{j}.'
                out[b + 'running_mean'] = torch.zeros(cout)
                out[b + 'running_var'] = torch.ones(cout)
                out[b + 'num_batches_tracked'] = torch.zeros((), dtype=torch.long)
        return out

    # -- one convolution: y[B*T*F, Cout] = im2col(x) @ Wp^T, in batch slices ---------------------------------------
    def _weight(self, name, dtype, cin_p):
        """Tap-major [Cout, 9*cin_p] copy of the conv weight in the compute dtype (zero rows for the padded channels)."""
        return ops.conv_weight_to_tap(self.arena.w(name, dtype), cin_p)

    @staticmethod
    def _slices(B, rows_per_sample, K):
        per = max(1, CONV_SLICE_ELEMS // max(1, rows_per_sample * K))
        return [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]

    def _implicit(self, cin_p):
        return cin_p <= self.implicit_max_channels

    @staticmethod
    def _row_slices(B, rows_per_sample):
        per = max(1, ((1 << 24) - 1) // rows_per_sample)           # the implicit loaders decode rows below 2^24
        return [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]

    def _conv_fwd(self, x, W, B, T, F, keep):
        """Returns (y, the im2col slices when `keep`: the weight-gradient GEMM of the backward reads them again, and
        288 GB of HBM hold them easily — 0.2 GB per ten-second chunk at the crnn.yaml widths)."""
        rows, K = T * F, W.shape[1]
        y = torch.empty((B * rows, W.shape[0]), dtype=x.dtype, device=x.device)
        if self._implicit(x.shape[1]):
            for b0, b1 in self._row_slices(B, rows):
                ops.conv3x3_fwd(x[b0 * rows:b1 * rows], W, b1 - b0, T, F, out=y[b0 * rows:b1 * rows])
            return y, None
        kept = [] if keep else None
        for b0, b1 in self._slices(B, rows, K):
            A = ops.im2col3x3(x[b0 * rows:b1 * rows], b1 - b0, T, F)
            ops.linear_fwd(A, W, out=y[b0 * rows:b1 * rows])
            if keep:
                kept.append(A)
        return y, kept

    def _conv_bwd(self, dy, x, W, dW, B, T, F, cin_p, kept=None, name=None):
        """dW (reference layout, overwritten) and dx = col2im(dy @ Wp) — or, for the implicit layers, dx = the 3x3
        convolution of dy with the flipped, transposed weight."""
        rows, K = T * F, W.shape[1]
        dx = torch.empty((B * rows, cin_p), dtype=x.dtype, device=x.device)
        dWp = torch.empty((W.shape[0], K), dtype=torch.float32, device=x.device)
        if self._implicit(cin_p):
            Wd = ops.conv_weight_to_tap_t(self.arena.w(name, x.dtype), cin_p)
            for n, (b0, b1) in enumerate(self._row_slices(B, rows)):
                ops.conv3x3_wgrad(dy[b0 * rows:b1 * rows], x[b0 * rows:b1 * rows], dWp, b1 - b0, T, F, accumulate=n > 0)
                ops.conv3x3_fwd(dy[b0 * rows:b1 * rows], Wd, b1 - b0, T, F, out=dx[b0 * rows:b1 * rows])
            ops.conv_wgrad_from_tap(dWp, dW, cin_p)
            return dx
        for n, (b0, b1) in enumerate(self._slices(B, rows, K)):
            A = kept[n] if kept is not None else ops.im2col3x3(x[b0 * rows:b1 * rows], b1 - b0, T, F)
            ops.linear_wgrad(dy[b0 * rows:b1 * rows], A, dWp, accumulate=n > 0)
            dA = ops.linear_dgrad(dy[b0 * rows:b1 * rows], W)
            ops.col2im3x3(dA, b1 - b0, T, F, cin_p, out=dx[b0 * rows:b1 * rows])
        ops.conv_wgrad_from_tap(dWp, dW, cin_p)
        return dx

    def _bn(self, b, j, y, training, buffers):
        a = self.arena
        rm, rv, nbt = buffers[b + f'bn{j}.running_mean'], buffers[b + f'bn{j}.running_var'], buffers[b + f'bn{j}.num_batches_tracked']
        sums = ops.bn2d_stats(y) if training else None
        mean_rstd, scale_shift = ops.bn2d_finalize(sums, y.shape[0], a.p(b + f'bn{j}.weight'), a.p(b + f'bn{j}.bias'), rm, rv, nbt,
                                                   training)
        return ops.bn_relu_fwd(y, scale_shift), mean_rstd

    # -- forward / backward, one ConvBlock at a time (the EINV2 CRNN cross-stitches between blocks) -------------------
    def forward_block(self, i, x, B, T, F, dtype, training, buffers):
        """ConvBlock i + its average pool (backbone.py:17-28 / :45-57). x: NHWC rows [B*T*F, cin_p]. Returns
        (rows [B*T'*F', C_i], saved, T', F')."""
        cout = self.widths[i]
        cin_p = self.cin_p if i == 0 else self.widths[i - 1]
        b = f'{self.prefix}conv_block{i + 1}.'
        W1 = self._weight(b + 'conv1.weight', dtype, cin_p)
        keep = training and self.keep_im2col
        y1, A1 = self._conv_fwd(x, W1, B, T, F, keep)
        z1, mr1 = self._bn(b, 1, y1, training, buffers)
        W2 = self._weight(b + 'conv2.weight', dtype, cout)
        y2, A2 = self._conv_fwd(z1, W2, B, T, F, keep)
        z2, mr2 = self._bn(b, 2, y2, training, buffers)
        pt, pf = self.pools[i]
        s = dict(x=x, y1=y1, z1=z1, mr1=mr1, y2=y2, z2=z2, mr2=mr2, T=T, F=F, cin_p=cin_p, A1=A1, A2=A2)
        return ops.avgpool_fwd(z2, B, T, F, pt, pf), s, T // pt, F // pf

    def backward_block(self, i, dx, s, B, dtype):
        a, cout = self.arena, self.widths[i]
        b = f'{self.prefix}conv_block{i + 1}.'
        pt, pf = self.pools[i]
        dz2 = ops.avgpool_bwd(dx, B, s['T'], s['F'], pt, pf)
        dy2 = ops.bn_relu_bwd(s['y2'], s['z2'], dz2, s['mr2'], a.p(b + 'bn2.weight'), a.g(b + 'bn2.weight'), a.g(b + 'bn2.bias'))
        W2 = self._weight(b + 'conv2.weight', dtype, cout)
        dz1 = self._conv_bwd(dy2, s['z1'], W2, a.g(b + 'conv2.weight'), B, s['T'], s['F'], cout, s['A2'], b + 'conv2.weight')
        s['A2'] = None
        dy1 = ops.bn_relu_bwd(s['y1'], s['z1'], dz1, s['mr1'], a.p(b + 'bn1.weight'), a.g(b + 'bn1.weight'), a.g(b + 'bn1.bias'))
        W1 = self._weight(b + 'conv1.weight', dtype, s['cin_p'])
        dx = self._conv_bwd(dy1, s['x'], W1, a.g(b + 'conv1.weight'), B, s['T'], s['F'], s['cin_p'], s['A1'], b + 'conv1.weight')
        s['A1'] = None
        return dx

    def forward_tail(self, x, B, T, F):
        """x.mean(dim=3) (accdoa.py:81): [B*T*F, C] -> [B*T, C]."""
        return ops.avgpool_fwd(x, B, T, F, 1, F) if F > 1 else x

    def backward_tail(self, dx, B, T, F):
        return ops.avgpool_bwd(dx, B, T, F, 1, F) if F > 1 else dx

    # -- forward / backward over the whole stack ----------------------------------------------------------------
    def forward(self, x, B, T, F, dtype, training, buffers):
        """x: NHWC rows [B*T*F, cin_p] (already normalised). Returns ([B*T', C_last] after the frequency mean, saved)."""
        saved = []
        for i in range(len(self.widths)):
            x, s, T, F = self.forward_block(i, x, B, T, F, dtype, training, buffers)
            saved.append(s)
        return self.forward_tail(x, B, T, F), dict(blocks=saved, T_out=T, F_last=F)

    def backward(self, dx, saved, B, dtype):
        dx = self.backward_tail(dx, B, saved['T_out'], saved['F_last'])
        for i in reversed(range(len(self.widths))):
            dx = self.backward_block(i, dx, saved['blocks'][i], B, dtype)
        return dx
