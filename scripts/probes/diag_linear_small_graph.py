import sys, os, torch
sys.path[:0]=['/root/repo','/root/repo/open-genie_amd','/root/repo/tests']
from genie import functional as GF
torch.manual_seed(0)
w = torch.randn(128, 10, device='cuda'); b = torch.randn(128, device='cuda')
# 1. plain eager, first call in a fresh process
x = torch.randn(2, 10, 4, 4, 4, device='cuda')
c = x.float().reshape(2, 10, -1).mean(-1)
y = GF.linear(c, w, b)
ref = torch.nn.functional.linear(c, w, b)
print('eager first call max diff', (y - ref).abs().max().item())
# 2. under a graph with changing input
side = torch.cuda.Stream()
xs = torch.randn(2, 10, 4, 4, 4, device='cuda')
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        c = xs.float().reshape(2, 10, -1).mean(-1); y = GF.linear(c, w, b)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    c = xs.float().reshape(2, 10, -1).mean(-1)
    yg = GF.linear(c, w, b)
    rg = torch.nn.functional.linear(c, w, b)
for i in range(3):
    xs.copy_(torch.randn(2, 10, 4, 4, 4, device='cuda') * (i + 1))
    g.replay(); torch.cuda.synchronize()
    print('replay', i, 'max diff hip vs torch-in-graph', (yg - rg).abs().max().item(), 'ref norm', rg.norm().item())
