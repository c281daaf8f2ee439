"""Debug: graph-mode trainer and eager trainer in lockstep; per-step state differences."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import htsat as oh
from oracle import synth
from pseldnets_amd.models import multi_accdoa
from pseldnets_amd.trainer import FusedTrainer
dev = torch.device('cuda:0')
TINY = dict(embed_dim=48, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], drop_path_rate=0.0)


class A(dict):
    __getattr__ = dict.__getitem__


CFG = A(data=A(n_mels=64, sample_rate=24000, hoplen=240), adapt=A())
dtype = torch.bfloat16 if os.environ.get('DT', 'bf16') == 'bf16' else torch.float32


def mk(use_graph):
    net = multi_accdoa.HTSAT(CFG, 3, 7, pretrained_path=None, **TINY)
    net.load_state_dict(oh.formula_state('multi_accdoa', 3, 7, dict(TINY, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16))), strict=False)
    net.compute_dtype = dtype
    net = net.to(dev)
    return net, FusedTrainer(net, None, 'adpit', lr=2e-5, max_norm=1.0, use_graph=use_graph, graph_warmup=3)


ne, te = mk(False)
ng, tg = mk(True)
lab = synth.formula_adpit_label(2, 100, 3).to(dev)
x = oh.formula_features(2).to(dev)
rd = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()
order = os.environ.get('ORDER', 'graph_first')
ne._materialize(dev); ng._materialize(dev)
for i in range(7):
    # the eager trainer starts every step from the graph trainer's state
    with torch.no_grad():
        ne.arena.flat.copy_(ng.arena.flat); ne._rm.copy_(ng._rm)
        if hasattr(ne, '_rv'): ne._rv.copy_(ng._rv)
        if ng.arena.m is not None:
            ne.arena.ensure_opt_state(); ne.arena.m.copy_(ng.arena.m); ne.arena.v.copy_(ng.arena.v)
        ne.arena.step = ng.arena.step
    if order == 'graph_first':
        lg = tg.training_step(x, {'adpit_label': lab})['loss_all'].item()
        le = te.training_step(x, {'adpit_label': lab})['loss_all'].item()
    else:
        le = te.training_step(x, {'adpit_label': lab})['loss_all'].item()
        lg = tg.training_step(x, {'adpit_label': lab})['loss_all'].item()
    ae, ag = ne.arena, ng.arena
    print(i, f'loss {le:.7f} {lg:.7f}', 'flat', rd(ag.flat, ae.flat), 'grad', rd(ag.grad, ae.grad), 'm', rd(ag.m, ae.m), 'v', rd(ag.v, ae.v))
    worst = sorted(((rd(ag.g(n), ae.g(n)), n) for n in ae.entries), reverse=True)[:4]
    if worst[0][0] > 1e-6:
        print('    worst grads:', [(f'{a:.2e}', n) for a, n in worst])
