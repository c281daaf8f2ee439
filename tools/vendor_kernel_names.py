import torch
dev = torch.device('cuda:0')
for M, N, K in [(8192, 8192, 8192), (12288, 3072, 768), (12288, 768, 3072), (49152, 1536, 384)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16()
    for _ in range(3):
        y = x @ w.t()
    torch.cuda.synchronize()
