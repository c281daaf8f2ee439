"""The RCCL layer (include/pseld_comm.h, pseldnets_amd/comm.py) on the ONE GPU of the test box: communicator life cycle at world size 1
(RCCL refuses two ranks on one device, profiles/r02_rccl_two_ranks_one_gpu.log, so no multi-rank run exists here - the transfer plan is
tested as arithmetic in tests/test_abi.py), the fixed-order sum kernel of the direct algorithm against numpy, and a fused training
step through `comm='rccl_direct'` equal to the step without a group."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module', autouse=True)
def _leave_no_group_behind():
    """The module's gloo group is torn down at its end: the single-rank RCCL tests of test_htsat_gpu.py need to create their own."""
    yield
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def _group():
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:29641', rank=0, world_size=1)
    return dist.group.WORLD


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("world,me", [(8, 0), (8, 3), (8, 7), (2, 1), (4, 2)])
def test_fixed_order_sum_kernel(dev, dtype, world, me):
    from pseldnets_amd import comm
    L = comm.lib()
    n, stride = 8 * 12345, 8 * 12352                       # chunk shorter than the stride (a bucket's last chunk)
    g = torch.Generator().manual_seed(world * 10 + me)
    copies = [torch.randn(n, generator=g).to(dtype) for _ in range(world)]
    own = copies[me].clone().to(dev)
    scratch = torch.zeros((world - 1) * stride, dtype=dtype, device=dev)
    for r in range(world):
        if r != me:
            slot = r if r < me else r - 1
            scratch[slot * stride: slot * stride + n] = copies[r].to(dev)
    rc = L.pseld_comm_sum_in_rank_order(own.data_ptr(), scratch.data_ptr(), n, stride, me, world, 0 if dtype == torch.float32 else 1,
                                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.pseld_comm_last_error()
    acc = copies[0].float().numpy().copy()
    for r in range(1, world):
        acc = acc + copies[r].float().numpy()                  # fp32, rank order: the kernel's order
    want = torch.from_numpy(acc).to(dtype)
    assert torch.equal(own.cpu(), want), (own.cpu().float() - want.float()).abs().max()


def test_single_rank_communicator_and_trainer_step(dev):
    from oracle import htsat as oh, synth
    from pseldnets_amd import comm
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.trainer import FusedTrainer
    group = _group()
    c = comm.RcclComm(group, dev, 'rccl_direct')
    assert (c.rank, c.world) == (0, 1) and comm.lib().pseld_comm_world(c.handle) == 1
    t = torch.arange(4096, dtype=torch.float32, device=dev)
    w = c.allreduce_(t)
    w.wait()
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(4096, dtype=torch.float32))       # one rank: the identity, through the whole call path
    c.close()

    class A(dict):
        __getattr__ = dict.__getitem__
    tiny = dict(embed_dim=48, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], drop_path_rate=0.0)
    cfg = A(data=A(n_mels=64, sample_rate=24000, hoplen=240), adapt=A())

    def run(group, comm_kind):
        net = multi_accdoa.HTSAT(cfg, 3, 7, pretrained_path=None, **tiny)
        net.load_state_dict(oh.formula_state('multi_accdoa', 3, 7, tiny), strict=False)
        net.to(dev)
        tr = FusedTrainer(net, None, 'adpit', lr=1e-4, max_norm=1.0, process_group=group, comm=comm_kind)
        x, lab = oh.formula_features(2).to(dev), synth.formula_adpit_label(2, 100, 3).to(dev)
        losses = [tr.training_step(x, {'adpit_label': lab})['loss_all'].item() for _ in range(2)]
        torch.cuda.synchronize()
        if tr._rccl is not None:
            tr._rccl.close()
        return losses, net.arena.flat.detach().cpu().clone()
    l0, f0 = run(None, 'torch')
    # every run differs from the next by the fp32 atomics of the bias-table gradients (and the grouped path takes the clipping norm from
    # per-bucket parts): parameters after two steps agree to that noise (AdamW normalises gradients: a near-zero gradient moves a weight by
    # up to lr), whatever the transport - the same bound tests/test_htsat_gpu.py::test_fused_step_through_a_single_rank_rccl_group uses
    for kind in ('torch', 'rccl', 'rccl_direct'):
        l1, f1 = run(group, kind)
        assert all(abs(a - b) < 1e-5 * abs(a) for a, b in zip(l0, l1)), (kind, l0, l1)
        assert (f0 - f1).abs().max().item() <= 2.5e-4 and ((f0 - f1).norm() / f0.norm()).item() < 1e-5, kind
