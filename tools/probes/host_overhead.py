"""host enqueue time vs GPU time of one training step (is the step launch-bound?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import mtp_amd
from mtp_amd.parallel import DataParallelTrainer

class A: image_size = 224; use_ckpt = "False"; precision = "bf16"
torch.manual_seed(0)
net = mtp_amd.vit_l_rvsa(A).cuda().train()
tr = DataParallelTrainer(net, total_steps=100, feature_dtype=torch.bfloat16)
img = torch.randn(64, 3, 224, 224, device="cuda")
def lg(feats):
    leaves = [f.detach().requires_grad_(True) for f in feats]
    loss = sum(f.float().mean() for f in leaves); loss.backward()
    return loss.detach(), [f.grad for f in leaves]
for _ in range(2): tr.step(img, lg)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); tr.step(img, lg); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host enqueue %.1f ms, total %.1f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tr.step(img, lg); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
