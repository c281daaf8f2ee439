import sys, ctypes, torch
sys.path.insert(0, '.')
from pseldnets_amd import ops, _lib
L = _lib.lib()
raw = L
dev = torch.device('cuda:0'); dt = torch.bfloat16
for (name, M, N, K, kind) in [('s0 fc1 fwd (out-heavy)', 786432, 384, 96, 'fwd'), ('s0 fc2 fwd (in-heavy)', 786432, 96, 384, 'fwd'), ('s0 fc2 dgrad (out-heavy)', 786432, 96, 384, 'dgrad'), ('s0 fc1 dgrad (in-heavy)', 786432, 384, 96, 'dgrad'), ('s2 fc1 fwd', 49152, 1536, 384, 'fwd'), ('s3 fc1 fwd', 12288, 3072, 768, 'fwd'), ('s0 qkv wgrad', 786432, 288, 96, 'wgrad'), ('s1 fc1 wgrad', 196608, 768, 192, 'wgrad'), ('s2 fc1 wgrad', 49152, 1536, 384, 'wgrad'), ('s2 proj wgrad', 49152, 384, 384, 'wgrad')]:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev).to(dt)
    dbg = torch.zeros(60000 * 6, dtype=torch.int64, device=dev)
    dwb = torch.empty(N * K + N, device=dev); dw = dwb[:N * K].view(N, K); db = dwb[N * K:]
    fn = (lambda: ops.linear_fwd(x, w, b)) if kind == 'fwd' else ((lambda: ops.linear_dgrad(dy, w)) if kind == 'dgrad' else (lambda: ops.linear_wgrad(dy, x, dw, dbias=db)))
    fn(); fn(); torch.cuda.synchronize()
    raw.pseld_gemm_set_debug_buffer(dbg.data_ptr())
    fn(); torch.cuda.synchronize()
    raw.pseld_gemm_set_debug_buffer(None)
    d = dbg.view(-1, 6).cpu()
    d = d[d[:, 0] > 0].double()
    n = d.shape[0]
    t0 = d[:, 0].min()
    load = (d[:, 1] - d[:, 0]).median().item(); kloop = (d[:, 2] - d[:, 1]).median().item(); epi = (d[:, 3] - d[:, 2]).median().item()
    total = (d[:, 3].max() - t0).item()
    cs = (d[:, 4] - d[:, 2]).median().item(); issue = (d[:, 5] - d[:, 4]).median().item(); drain = (d[:, 3] - d[:, 5]).median().item()
    print(f"   epilogue split: acc->LDS tile {cs:.0f}, read tile + issue stores {issue:.0f}, drain stores {drain:.0f}")
    print(f"{name}: blocks {n}; median cycles: first-load+store_lds {load:.0f}, rest of K loop {kloop:.0f}, epilogue+drain {epi:.0f}; kernel span {total:.0f} ticks (100MHz ticks = {total/100:.1f} us)")
