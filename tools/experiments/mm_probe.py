import torch
dev = torch.device('cuda:0')
W = torch.randn(512, 3072, device=dev); X = torch.randn(2, 3072, 245760, device=dev)
for _ in range(3): Y = torch.matmul(W, X)
torch.cuda.synchronize()
