"""Race screen at the bench workload: full HTS-AT mACCDOA bf16, 192 chunks, train-mode forward + backward REPEATED on the same weights and
inputs. Every kernel of the step runs at its production geometry (persistent GEMMs with hand-counted vmcnt, LDS-DMA pipelines, the weight
gradients on their side stream); everything they produce is deterministic by construction except the relative-position bias-table
gradients (fp32 atomics), so the network output and every other parameter gradient must be bit-identical run after run.
python tools/step_determinism.py [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from pseldnets_amd import ops
from pseldnets_amd.models import multi_accdoa
import test_htsat_gpu as T

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
torch.manual_seed(11)
net = multi_accdoa.HTSAT(T.CFG, 170, 7, pretrained_path=None, **T.kw(dict(T.FULL, drop_path_rate=0.1)))
net.compute_dtype = torch.bfloat16
net.to(dev)
Bt = 192
g = torch.Generator().manual_seed(5)
x = torch.randn(Bt, 7, 1001, 64, generator=g).to(dev)
act = (torch.rand(Bt, 100, 170, generator=g) < 0.02).float()
lab = torch.zeros(Bt, 100, 6, 4, 170)
lab[:, :, 0, 0] = act
lab[:, :, 0, 1:] = torch.nn.functional.normalize(torch.randn(Bt, 100, 3, 170, generator=g), dim=2) * act.unsqueeze(2)
lab = lab.to(dev)
net._materialize(dev)
atomics = [n for n in net.arena.entries if 'relative_position_bias_table' in n]
ref_y = ref_g = None
bad_y = bad_g = 0
worst = 0.0
for r in range(reps):
    torch.manual_seed(123)                       # the same DropPath masks every repeat
    y, saved = net._forward_impl(x, True)
    _, dpred = ops.adpit_loss(y, lab)
    net.zero_grad_arena()
    net._backward_impl(saved, (dpred,))
    torch.cuda.synchronize()
    gr = net.arena.grad.clone()
    for n in atomics:
        net.arena.view(gr, n).zero_()
    if ref_y is None:
        ref_y, ref_g = y.clone(), gr
        continue
    if not torch.equal(y, ref_y):
        bad_y += 1
    if not torch.equal(gr, ref_g):
        bad_g += 1
        worst = max(worst, ((gr - ref_g).norm() / ref_g.norm()).item())
    del saved
print(f"{reps} train-mode forward + backward passes at 192 chunks (bf16, drop_path 0.1): outputs differing from the first pass: {bad_y}; "
      f"gradient arenas (bias tables excluded: fp32 atomics) differing: {bad_g} (worst rel-L2 {worst:.2e})")
sys.exit(1 if bad_y or bad_g else 0)
