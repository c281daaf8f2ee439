"""GRU decoder of the CRNN networks driven on MI355X kernels — forward AND hand-written backward (BPTT).

Host-side mirror of `Decoder('gru')` (reference models/components/model_utilities.py:249-252: nn.GRU(num_feats, num_feats // 2,
num_layers, bidirectional, batch_first); configs/model/default.yaml). Parameters keep nn.GRU's state-dict names under `prefix`
(`weight_ih_l0`, `weight_hh_l0_reverse`, ...; gate order r | z | n). The input projections of all timesteps, every weight
gradient and the input gradient are single GEMMs over [B*T, .]; the recurrence itself is one B-row GEMM (h_{t-1} W_hh^T) and
one gate kernel per timestep and direction (125 steps here), and the same pair in reverse for the backward, all launched by
ONE C-ABI call per layer and direction (pseld_gru_seq_fwd / _bwd); in bf16 the B-row GEMMs take the skinny kernel of gemm.hip
(32 output columns per workgroup, K split over the waves).
"""
import torch

from ... import ops


class GRUDecoder:
    def __init__(self, arena, prefix, num_feats, num_layers=2):
        if num_feats % 16:
            raise ValueError("GRU decoder: num_feats must be a multiple of 16")
        self.arena, self.prefix, self.I, self.H, self.L = arena, prefix, num_feats, num_feats // 2, num_layers
        for layer in range(num_layers):
            for sfx in ('', '_reverse'):
                n_in = num_feats if layer == 0 else 2 * self.H
                arena.add(f'{prefix}weight_ih_l{layer}{sfx}', (3 * self.H, n_in))
                arena.add(f'{prefix}weight_hh_l{layer}{sfx}', (3 * self.H, self.H))
                arena.add(f'{prefix}bias_ih_l{layer}{sfx}', (3 * self.H,))
                arena.add(f'{prefix}bias_hh_l{layer}{sfx}', (3 * self.H,))

    def static_buffers(self):
        return {}

    def forward(self, x, B, T, training=None, buffers=None):
        """x [B*T, I] (row b*T + t) -> [B*T, 2H] (forward | reverse hidden states)."""
        a, H, dt = self.arena, self.H, x.dtype
        saved = []
        for layer in range(self.L):
            out = torch.empty((B * T, 2 * H), dtype=dt, device=x.device)
            o3 = out.view(B, T, 2 * H)
            per_dir = []
            for d, sfx in enumerate(('', '_reverse')):
                p = f'{self.prefix}%s_l{layer}{sfx}'
                gi = ops.linear_fwd(x, a.w(p % 'weight_ih', dt), a.p(p % 'bias_ih')).view(B, T, 3 * H)
                gates = torch.empty((T, B, 4 * H), dtype=dt, device=x.device)
                ops.gru_seq_fwd(gi, a.w(p % 'weight_hh', dt), a.p(p % 'bias_hh'), o3[:, :, d * H:(d + 1) * H], gates, reverse=d == 1)
                per_dir.append(gates)
            saved.append(dict(x=x, out=out, gates=per_dir))
            x = out
        return x, dict(layers=saved, T=T)

    def backward(self, dout, saved, B):
        """dout [B*T, 2H] -> dx [B*T, I]; parameter gradients land in the arena."""
        a, H, T, dt = self.arena, self.H, saved['T'], dout.dtype
        for layer in reversed(range(self.L)):
            sv = saved['layers'][layer]
            x, o3, d3 = sv['x'], sv['out'].view(B, T, 2 * H), dout.view(B, T, 2 * H)
            dx = None
            for d, sfx in enumerate(('', '_reverse')):
                p = f'{self.prefix}%s_l{layer}{sfx}'
                w_hh, w_hh_t = a.w(p % 'weight_hh', dt), a.wt(p % 'weight_hh', dt)
                gates = sv['gates'][d]
                dgi, dgh, hprev_all = ops.gru_seq_bwd(d3[:, :, d * H:(d + 1) * H], o3[:, :, d * H:(d + 1) * H], gates, w_hh, w_hh_t,
                                                      reverse=d == 1)
                dgi2 = dgi.view(B * T, 3 * H)
                ops.linear_wgrad(dgh.view(T * B, 3 * H), hprev_all.view(T * B, H), a.g(p % 'weight_hh'), dbias=a.g(p % 'bias_hh'))
                ops.linear_wgrad(dgi2, x, a.g(p % 'weight_ih'), dbias=a.g(p % 'bias_ih'))
                dx = ops.linear_dgrad(dgi2, a.w(p % 'weight_ih', dt), wt=a.wt(p % 'weight_ih', dt), resid=dx)
            dout = dx
        return dout
