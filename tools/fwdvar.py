import sys, torch
sys.path.insert(0, '.')
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M, N, K = 786432, 96, 96
x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); b = torch.randn(N, device=dev)
wt = w.t().contiguous()
y = torch.empty(M, N, device=dev, dtype=dt)
print('fwd NT with bias   ', timeit(lambda: ops.linear_fwd(x, w, b, out=y)))
print('fwd NT without bias', timeit(lambda: ops.linear_fwd(x, w, None, out=y)))
print('dgrad NN (x @ wt)  ', timeit(lambda: ops.linear_dgrad(x, wt, out=y)))
x0 = torch.zeros_like(x)
print('fwd NT zeros input ', timeit(lambda: ops.linear_fwd(x0, w, None, out=y)))
print('dgrad NN zeros     ', timeit(lambda: ops.linear_dgrad(x0, wt, out=y)))
