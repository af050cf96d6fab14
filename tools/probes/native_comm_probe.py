"""Why does the step slow down when the gradient exchange goes through the C-ABI RCCL communicator (world size 1, MTP_FORCE_COMM=1)?
host enqueue vs total time of the step, thread count; exchange on / off.   MTP_NATIVE_COMM=0|1 python tools/probes/native_comm_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1", MTP_FORCE_COMM="1")
import torch
import torch.distributed as dist
import mtp_amd
from mtp_amd.parallel import DataParallelTrainer


class A: image_size = 224; use_ckpt = "False"; precision = "bf16"


def lg(feats):
    loss = sum(f.sum(dtype=torch.float32) / f.numel() for f in feats)
    return loss, [torch.full_like(f, 1.0 / f.numel()) for f in feats]


def measure(tr, img, tag):
    for _ in range(3):
        tr.step(img, lg)
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(5):
        t0 = time.perf_counter(); tr.step(img, lg); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
    print("%-60s host enqueue %.1f ms, step %.1f ms | %d threads" % (tag, min(hs), min(ts), len(os.listdir("/proc/self/task"))), flush=True)


torch.manual_seed(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
net = mtp_amd.vit_l_rvsa(A).cuda().train()
tr = DataParallelTrainer(net, total_steps=100, feature_dtype=torch.bfloat16)
img = torch.randn(64, 3, 224, 224, device="cuda")
red = tr.reducer
print("native:", red.native is not None, "mode", red.mode)
red.active = False
measure(tr, img, "exchange off (before any collective)")
red.active = True
measure(tr, img, "exchange on")
red.active = False
measure(tr, img, "exchange off again")
if red.native is not None and "--stream" in sys.argv:
    # the same collectives on another stream object: is it the stream?
    red.stream = torch.cuda.Stream()
    red.active = True
    measure(tr, img, "exchange on, fresh side stream")
if "--bucket" in sys.argv:
    red.active = True
    red.bucket_bytes = 1 << 40
    measure(tr, img, "exchange on, ONE collective per step")
dist.destroy_process_group()
