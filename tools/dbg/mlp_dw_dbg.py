import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import test_mlp_gpu as t
dev = torch.device('cuda')
for dtype in (torch.float32, torch.bfloat16):
    for C, M, rps, drop in ((96, 512, 256, False), (96, 1024, 256, True), (192, 512, 256, False)):
        c = t._case(C, M, rps, dtype, dev, seed=1, drop=drop)
        ref = t._reference(c, rps)
        got = t._run_fused(c, rps)
        H = 4 * C
        print(dtype, C, M, 'drop', drop, {k: round(t._rel(got[k], ref[k]), 6) for k in ('y', 'dxh', 'dw1', 'db1', 'dw2', 'db2')})
        e = (got['db2'].double() - ref['db2']).abs().cpu()
        print('  db2 abs err by channel block of 8:', [round(e[i:i + 8].max().item(), 4) for i in range(0, C, 8)], 'ref max', ref['db2'].abs().max().item())
        e = (got['db1'].double() - ref['db1']).abs().cpu()
        print('  db1 err by 32-block:', [round(e[i:i + 32].max().item(), 4) for i in range(0, H, 32)], 'ref max', ref['db1'].abs().max().item())
        e = (got['dw1'].double() - ref['dw1']).abs().cpu()
        print('  dw1 err rows by 32-block:', [round(e[i:i + 32].max().item(), 4) for i in range(0, H, 32)])
        print('  dw1 err cols by 8-block:', [round(e[:, i:i + 8].max().item(), 4) for i in range(0, C, 8)])
        print('  ratio got/ref db2[:8]', (got['db2'][:8].double() / ref['db2'][:8]).cpu().numpy().round(3))
        print('  ratio got/ref db1[:8]', (got['db1'][:8].double() / ref['db1'][:8]).cpu().numpy().round(3))
