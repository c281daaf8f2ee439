"""s_memtime phase stamps of the weight-gradient kernels: python tools/wgrad_stamps.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
L = _lib.lib()
dev = torch.device('cuda:0'); dt = torch.bfloat16
for M, N, K in [(49152, 1536, 384), (49152, 384, 384), (12288, 2304, 768)]:
    dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
    dwb = torch.empty(N * K + N, device=dev); dw = dwb[:N * K].view(N, K); db = dwb[N * K:]
    fn = lambda: ops.linear_wgrad(dy, x, dw, dbias=db)
    fn(); fn(); torch.cuda.synchronize()
    dbg = torch.zeros(60000 * 6, dtype=torch.int64, device=dev)
    L.pseld_gemm_set_debug_buffer(dbg.data_ptr())
    fn(); torch.cuda.synchronize()
    L.pseld_gemm_set_debug_buffer(None)
    d = dbg.view(-1, 6).cpu()
    d = d[d[:, 0] > 0].double()
    med = lambda t: t.median().item()
    print(f"M={M} N={N} K={K}: workgroups {d.shape[0]}; first slice {med(d[:, 1] - d[:, 0]):.0f}; slices 1-8 {med(d[:, 4] - d[:, 1]) / 8:.0f}/slice; "
          f"whole loop {med(d[:, 2] - d[:, 0]):.0f}; epilogue issue {med(d[:, 5] - d[:, 2]):.0f}; store drain {med(d[:, 3] - d[:, 5]):.0f}; "
          f"start skew {(d[:, 0].max() - d[:, 0].min()).item():.0f}; span {(d[:, 3].max() - d[:, 0].min()).item():.0f}")
