import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mtp_amd
from mtp_amd.parallel import DataParallelTrainer
from torch.profiler import profile, ProfilerActivity
class A: image_size=224; use_ckpt="False"; precision="bf16"
torch.manual_seed(0)
net = mtp_amd.vit_l_rvsa(A).cuda().train()
tr = DataParallelTrainer(net, total_steps=100, feature_dtype=torch.bfloat16)
img = torch.randn(64,3,224,224,device="cuda")
def lg(feats):
    loss = sum(f.sum(dtype=torch.float32)/f.numel() for f in feats)
    return loss, [torch.full_like(f, 1.0/f.numel()) for f in feats]
for _ in range(2): tr.step(img, lg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(img, lg)
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=6)
rows=[e for e in ev if ("Memcpy" in e.key or "Memset" in e.key or "copy_" in e.key or "aten::fill_" in e.key or "aten::zero_" in e.key)]
rows.sort(key=lambda e:-e.count)
for e in rows[:14]:
    print(e.key, e.count, [s for s in e.stack if "mtp_amd" in s or "bench" in s or "prof_step" in s][:3])
