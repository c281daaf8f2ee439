import torch
dev = torch.device('cuda:0')
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (151, 453, 1200):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device=dev); b = torch.empty_like(a)
    tf = t(lambda: a.zero_()); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.float().sum() if False else torch.sum(a.view(torch.int16)[: n // 1]))
    print(f"{mb} MB: fill {mb/1e6/tf:.2f} TB/s write | copy {2*mb/1e6/tc:.2f} TB/s (read + write)")
