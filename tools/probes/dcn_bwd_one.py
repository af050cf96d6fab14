"""the DCNv3 forward + backward at InternImage-XL's four level geometries (512^2, batch 8, bf16, offsets drawn like bench.py's heads), 10 calls each -- for rocprofv3"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from mtp_amd.ops_dcnv3 import functions as F

dev, dt = "cuda", torch.bfloat16
for (N, HW, M) in [(8, 128, 12), (8, 64, 24), (8, 32, 48), (8, 16, 96)]:
    x = torch.randn(N, HW, HW, M * 16, device=dev).to(dt)
    m = torch.softmax(torch.randn(N, HW, HW, M, 9, device=dev), -1).reshape(N, HW, HW, M * 9).to(dt)
    off = (torch.randn(N, HW, HW, M * 18, device=dev) * 0.02 * (M * 16) ** 0.5).to(dt)
    G = torch.randn(N, HW, HW, M * 16, device=dev).to(dt)
    a = (3, 3, 1, 1, 1, 1, 1, 1, M, 16, 2.0)
    for _ in range(10):
        F.dcnv3_forward(x, off, m, *a, 256)
        F.dcnv3_backward_act(x, off, m, *a, G, 256, ((M * 18 + 7) // 8) * 8)
torch.cuda.synchronize()
