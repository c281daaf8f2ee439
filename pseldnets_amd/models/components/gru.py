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
        outs, saved = self.forward_many([self], [x], B, T)
        return outs[0], saved[0]

    def backward(self, dout, saved, B):
        """dout [B*T, 2H] -> dx [B*T, I]; parameter gradients land in the arena."""
        return self.backward_many([self], [dout], [saved], B)[0]

    @staticmethod
    def forward_many(decs, xs, B, T):
        """Several GRU stacks of one geometry (the six decoders of the EINV2 tail; a single stack is the two directions of each
        layer) advanced together: per layer ONE launch per timestep covers every stack and direction."""
        L, H = decs[0].L, decs[0].H
        assert all(d.L == L and d.H == H for d in decs)
        dt = xs[0].dtype
        saved = [[] for _ in decs]
        xs = list(xs)
        for layer in range(L):
            outs = [torch.empty((B * T, 2 * H), dtype=dt, device=xs[0].device) for _ in decs]
            gis, whh, bhh, seqs, gates, rev = [], [], [], [], [], []
            for dec, x, out in zip(decs, xs, outs):
                a, o3 = dec.arena, out.view(B, T, 2 * H)
                for d, sfx in enumerate(('', '_reverse')):
                    p = f'{dec.prefix}%s_l{layer}{sfx}'
                    gis.append(ops.linear_fwd(x, a.w(p % 'weight_ih', dt), a.p(p % 'bias_ih')).view(B, T, 3 * H))
                    whh.append(a.w(p % 'weight_hh', dt)); bhh.append(a.p(p % 'bias_hh'))
                    seqs.append(o3[:, :, d * H:(d + 1) * H])
                    gates.append(torch.empty((T, B, 4 * H), dtype=dt, device=x.device))
                    rev.append(d == 1)
            ops.gru_multi_fwd(gis, whh, bhh, seqs, gates, rev)
            for i, (x, out) in enumerate(zip(xs, outs)):
                saved[i].append(dict(x=x, out=out, gates=gates[2 * i:2 * i + 2]))
            xs = outs
        return xs, [dict(layers=s, T=T) for s in saved]

    @staticmethod
    def backward_many(decs, douts, saveds, B):
        L, H, T = decs[0].L, decs[0].H, saveds[0]['T']
        dt = douts[0].dtype
        douts = list(douts)
        for layer in reversed(range(L)):
            dseqs, seqs, gates, whh, whht, rev = [], [], [], [], [], []
            for dec, dout, sv in zip(decs, douts, saveds):
                a, s = dec.arena, sv['layers'][layer]
                o3, d3 = s['out'].view(B, T, 2 * H), dout.view(B, T, 2 * H)
                for d, sfx in enumerate(('', '_reverse')):
                    p = f'{dec.prefix}%s_l{layer}{sfx}'
                    dseqs.append(d3[:, :, d * H:(d + 1) * H]); seqs.append(o3[:, :, d * H:(d + 1) * H]); gates.append(s['gates'][d])
                    whh.append(a.w(p % 'weight_hh', dt)); whht.append(a.wt(p % 'weight_hh', dt)); rev.append(d == 1)
            dgis, dghs, hprevs = ops.gru_multi_bwd(dseqs, seqs, gates, whh, whht, rev)
            new = []
            for i, (dec, sv) in enumerate(zip(decs, saveds)):
                a, x = dec.arena, sv['layers'][layer]['x']
                dx = None
                for d, sfx in enumerate(('', '_reverse')):
                    p = f'{dec.prefix}%s_l{layer}{sfx}'
                    dgi2 = dgis[2 * i + d].view(B * T, 3 * H)
                    ops.linear_wgrad(dghs[2 * i + d].view(T * B, 3 * H), hprevs[2 * i + d].view(T * B, H), a.g(p % 'weight_hh'),
                                     dbias=a.g(p % 'bias_hh'))
                    ops.linear_wgrad(dgi2, x, a.g(p % 'weight_ih'), dbias=a.g(p % 'bias_ih'))
                    dx = ops.linear_dgrad(dgi2, a.w(p % 'weight_ih', dt), wt=a.wt(p % 'weight_ih', dt), resid=dx)
                new.append(dx)
            douts = new
        return douts
