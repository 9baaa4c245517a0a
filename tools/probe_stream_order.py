"""Does a kernel see the data written by the previous kernel of the same stream when three streams run concurrently?  (torch ops only)"""
import torch
ss = [torch.cuda.Stream() for _ in range(3)]
n = 20 * 1024 * 1024 // 4
bad = 0
bufs = [[torch.empty(n, device='cuda') for _ in range(4)] for _ in ss]
junk = [torch.randn(64 * 1024 * 1024 // 4, device='cuda') for _ in ss]
torch.cuda.synchronize()
res = []
for it in range(200):
    for k, s in enumerate(ss):
        with torch.cuda.stream(s):
            b = bufs[k][it % 4]
            b.fill_(float(it))                      # writer kernel
            res.append((b != float(it)).sum())      # reader kernel right behind it
            junk[k].mul_(1.0000001)                 # traffic that churns the caches
torch.cuda.synchronize()
print("mismatching elements seen by readers:", int(torch.stack(res).sum()))
