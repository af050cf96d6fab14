import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mtp_amd.ops_dcnv3 import dcnv3_backward
dev = "cuda"
bf = torch.bfloat16
mode = sys.argv[1]
for (N, HW, M) in [(8, 16, 96), (8, 32, 48)]:
    sets = []
    nset = 1 if mode == "hot" else 24
    for i in range(nset):
        x = torch.randn(N, HW, HW, M * 16, device=dev).to(bf)
        off = torch.zeros(N, HW, HW, M * 18, device=dev, dtype=bf)
        if mode == "tiny":
            off = (torch.randn(N, HW, HW, M * 18, device=dev) * 0.01).to(bf)
        m = torch.softmax(torch.randn(N, HW, HW, M, 9, device=dev), -1).reshape(N, HW, HW, M * 9).to(bf)
        G = (torch.randn(N, HW, HW, M * 16, device=dev) * (1e-6 if mode == "smallgrad" else 1.0)).to(bf)
        sets.append((x, off, m, G))
    a = (3, 3, 1, 1, 1, 1, 1, 1, M, 16, 2.0)
    big = torch.empty(512 << 20, device=dev, dtype=torch.uint8)
    for it in range(24):
        x, off, m, G = sets[it % nset]
        if mode != "hot":
            big.fill_(it)          # push everything out of L2 / the infinity cache
        dcnv3_backward(x, off, m, *a, G, 256, 0)
    torch.cuda.synchronize()
