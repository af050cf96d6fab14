#!/usr/bin/env python
"""Headline benchmark: images/sec of one pretrain step of the ViT-L + RVSA backbone (224^2, bf16) on N MI355X.

A "step" = backbone forward + backward of the MTP shared encoder on one synthetic batch (B=64 images per GPU),
bucketed RCCL gradient all-reduce (N > 1, overlapped with the backward on a side stream), clip_grad_norm_(5) and AdamW --
the reference's step recipe (main_pretrain.py:721-788) with `loss = sum(mean(f))` over the 4 feature maps standing in
for the three task decoders (they live in un-vendored mmseg/mmdet/mmrotate).  Inputs are resident in HBM.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 ...          # no launcher environment: starts its own 8 ranks (torch.distributed.run on 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...

Prints ONE JSON line (rank 0) with `roofline` (dominant GEMM kernel family, HIP-event timed on extra steps run right behind the timed
region, so that `value` times the step alone) and, at N=1, `cpu_baseline` (the oracle -- a CPU port of the reference path -- timed on the
host cores on bounded samples: BASELINE configs[0] on all threads and on one, the headline model's fwd+bwd on all threads).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md
FWD_GF_PER_IMAGE = {"vit_l": 129.97, "vit_b": 39.78}   # BASELINE.md section 4 (dense contractions, forward)


class GemmTimer:
    """HIP-event timing of every GEMM launch inside the timed region (events on the launch stream = torch's current stream)."""

    def __init__(self, ops):
        self.ops, self.on = ops, False
        self.recs = ([], [])      # [0]: instrumented steps run on ONE stream (a launch's duration is the kernel's own); [1]: as the step runs (side stream)
        self.rec = self.recs[0]
        self._nt, self._tn = ops.gemm_nt, ops.gemm_tn

    def install(self):
        def nt(a, w, out, *args, **kw):
            if not self.on:
                return self._nt(a, w, out, *args, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self._nt(a, w, out, *args, **kw)
            e.record()
            self.rec.append(("gemm_nt", 2.0 * a.shape[0] * w.shape[0] * a.shape[1], s, e, (a.shape[0], w.shape[0], a.shape[1], str(out.dtype)[6:], kw.get("epi", 0))))
            return r

        def tn(a, b, out, *args, **kw):
            if not self.on:
                return self._tn(a, b, out, *args, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self._tn(a, b, out, *args, **kw)
            e.record()
            self.rec.append(("gemm_tn", 2.0 * a.shape[0] * a.shape[1] * b.shape[1], s, e, (a.shape[1], b.shape[1], a.shape[0], "split-K", 0)))
            return r
        self.ops.gemm_nt, self.ops.gemm_tn = nt, tn
        flush0 = self.ops.WgradQueue._launch
        timer = self

        def flush(q):      # the grouped weight-gradient launches (the bulk of the TN family)
            if not timer.on or not q.jobs:
                return flush0(q)
            fl = sum(2.0 * j[0].shape[0] * j[0].shape[1] * j[1].shape[1] for j in q.jobs)
            shape = ("grouped", len(q.jobs), q.tiles, q.jobs[0][0].shape[0], 0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = flush0(q)
            e.record()
            timer.rec.append(("gemm_tn", fl, s, e, shape))
            return r
        self.ops.WgradQueue._launch = flush     # (_launch runs under the queue's launch stream: the events land there)

    def use(self, k):
        self.rec = self.recs[k]

    def summary(self, k=0):
        fam = {}
        for name, fl, s, e, _ in self.recs[k]:
            d = fam.setdefault(name, [0.0, 0.0, 0])
            d[0] += fl
            d[1] += s.elapsed_time(e) * 1e-3
            d[2] += 1
        return {k: dict(flops=v[0], seconds=v[1], launches=v[2]) for k, v in fam.items()}


    def shapes(self):
        """per-(family, shape) table of the instrumented launches (--gemm-shapes): where a model's GEMM time goes"""
        t = {}
        for name, fl, s, e, shape in self.recs[0]:
            d = t.setdefault((name,) + tuple(shape), [0.0, 0.0, 0])
            d[0] += fl
            d[1] += s.elapsed_time(e) * 1e-3
            d[2] += 1
        rows = sorted(t.items(), key=lambda kv: -kv[1][1])
        return ["%-60s %5d launches %9.1f us avg %8.3f ms total %7.1f TF/s" % (str(k), v[2], v[1] / v[2] * 1e6, v[1] * 1e3, v[0] / v[1] / 1e12) for k, v in rows]


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(model, seconds_budget=38.0):
    """The oracle (CPU port of the reference path, pinned to the reference's golden vectors) timed on the host cores, the legs BASELINE.md section 5 names,
    each a bounded sample: ViT-B forward at batch 2 (BASELINE configs[0]) on all host threads and on ONE thread, and the headline model's fwd+bwd at a
    small batch on all threads.  Top-level value / cores / sample = the fwd+bwd leg (the metric's own workload); `samples` holds every leg with its
    configuration, thread count and the CPU model.  Baseline, not target."""
    import statistics
    import recipe
    from oracle import vit_rvsa_oracle as O
    cfgs = dict(vit_l=(1024, 24, 16, 6, [7, 11, 15, 23]), vit_b=(768, 12, 12, 3, [3, 5, 7, 11]))
    cores = os.cpu_count() or 1
    many = min(cores, 64)
    cpu = _cpu_model()
    t_start = time.time()

    def leg(name, threads, B, backward, budget, max_iters, warm_iters):
        cfg = cfgs[name]
        torch.set_num_threads(threads)
        p = {k: v.requires_grad_(backward) for k, v in recipe.make_params(recipe.state_shapes(cfg[0], cfg[1], cfg[2], cfg[3])).items()}
        img = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(2023))

        def one():
            if backward:
                feats = O.backbone_forward(img, p, cfg[1], cfg[2], cfg[3], cfg[4])
                sum(f.mean() for f in feats).backward()
                for v in p.values():
                    v.grad = None
            else:
                with torch.no_grad():
                    O.backbone_forward(img, p, cfg[1], cfg[2], cfg[3], cfg[4])
        t0 = time.time()
        for _ in range(warm_iters):
            one()
        warm = time.time() - t0
        ts = []
        while len(ts) < 1 or (sum(ts) + warm + max(ts) <= budget and len(ts) < max_iters):      # (another iteration only when it still fits the budget)
            t1 = time.time()
            one()
            ts.append(time.time() - t1)
        dt = statistics.median(ts)
        return dict(config="%s %s, batch %d, 224x224, fp32" % ({"vit_b": "ViT-B + RVSA", "vit_l": "ViT-L + RVSA"}[name], "fwd+bwd" if backward else "forward", B),
                    value=round(B / dt, 3), unit="images/sec", seconds_per_iter=round(dt, 4), threads=threads, host_threads=cores, cpu_model=cpu,
                    iters=len(ts), warmup_iters=warm_iters, statistic="median")
    samples = [leg("vit_b", many, 2, False, 4.0, 5, 2),          # BASELINE configs[0]: "ViT-B/16 backbone forward only, batch=2 ... CPU PyTorch reference"
               leg("vit_b", 1, 2, False, 10.0, 3, 1),            # the same on one thread
               leg(model, many, 4, True, max(8.0, seconds_budget - (time.time() - t_start) - 1.0), 5, 1)]
    torch.set_num_threads(many)
    head = samples[-1]
    return dict(value=head["value"], unit="images/sec", cores=head["threads"], kind="port", cpu_model=cpu,
                sample="oracle (torch CPU fp32 restatement of the reference path) %s, median of %d timed iters after %d warm-up, %d of %d host threads; "
                       "`samples` adds BASELINE configs[0] (ViT-B forward, batch 2) on all threads and on one" % (head["config"], head["iters"], head["warmup_iters"], head["threads"], cores),
                samples=samples, seconds_spent=round(time.time() - t_start, 1))


def parse_rccl_log(path):
    """what RCCL reported about itself in its NCCL_DEBUG=INFO log (rank 0): version, channel count, the algorithm / protocol names it mentions for the
    collectives of the run, the transports of its rings.  A tolerant line scan -- the log format is RCCL's, not ours; absent keys stay None."""
    import re
    info = dict(version=None, channels=None, algorithms=[], protocols=[], transports=[], collective_lines=0, log_lines=0, tuning_lines=[])
    try:
        lines = open(path, errors="replace").read().splitlines()
    except OSError:
        return None
    keep = os.environ.get("MTP_RCCL_LOG_COPY")      # keep the raw log next to the line (the temporary file goes with the box)
    if keep:
        try:
            with open(keep, "w") as f:
                f.write("\n".join(lines[:4000]) + "\n")
        except OSError:
            pass
    info["log_lines"] = len(lines)
    chans = set()
    for ln in lines:
        m = re.search(r"(?:RCCL|NCCL) version\s*:?\s*([\w.+:-]+)", ln)
        if m and not info["version"]:
            info["version"] = m.group(1)
        m = re.search(r"Bytes -> Algo (\d+) proto (\d+)", ln)      # NCCL's TUNING line: the algorithm / protocol by number
        if m:
            a, pr = int(m.group(1)), int(m.group(2))
            a = ("Tree", "Ring", "CollNetDirect", "CollNetChain", "NVLS", "NVLSTree", "PAT")[a] if a < 7 else str(a)
            pr = ("LL", "LL128", "Simple")[pr] if pr < 3 else str(pr)
            if a not in info["algorithms"]:
                info["algorithms"].append(a)
            if pr not in info["protocols"]:
                info["protocols"].append(pr)
            if len(info["tuning_lines"]) < 4:
                info["tuning_lines"].append(ln[-200:])
        m = re.search(r"Channel (\d+)[/ :]", ln)
        if m:
            chans.add(int(m.group(1)))
        m = re.search(r"(\d+) coll channels", ln)
        if m:
            info["channels"] = int(m.group(1))
        m = re.search(r"nChannels (\d+)", ln)
        if m and info["channels"] is None:
            info["channels"] = int(m.group(1))
        for a in ("Ring", "Tree", "CollNet", "NVLS", "PAT"):
            if re.search(r"\b(?:Algo|algo|algorithm)\b.*\b%s\b" % a, ln) or re.search(r"\b%s\b.*\b(?:LL128|LL|Simple)\b" % a, ln):
                if a not in info["algorithms"]:
                    info["algorithms"].append(a)
        for pr in ("LL128", "LL", "Simple"):
            if (re.search(r"(?:Proto|proto|protocol)\W+%s\b" % pr, ln) or re.search(r"\b(?:Ring|Tree|CollNet|NVLS|PAT)\b\W+%s\b" % pr, ln)) and pr not in info["protocols"]:
                info["protocols"].append(pr)
        m = re.search(r" via (\S+)", ln)                          # "Channel 00 : 0[0] -> 1[1] via P2P/IPC": the transport of a ring / tree edge
        if m and m.group(1) not in info["transports"]:
            info["transports"].append(m.group(1))
        m = re.search(r"\bnranks (\d+)", ln)
        if m:
            info["nranks"] = int(m.group(1))
        if "AllReduce" in ln or "ReduceScatter" in ln or "AllGather" in ln:
            info["collective_lines"] += 1
    if info["channels"] is None and chans:
        info["channels"] = max(chans) + 1
    if info.get("nranks") == 1:
        info["note"] = "one rank: RCCL copies instead of choosing an algorithm / protocol (no TUNING lines, no ring edges)"
    return info


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _error_line(args, msg):
    """the JSON line of a run that could not be measured (one line, last on stdout, like a result): value null + `error`"""
    return json.dumps({"metric": "images/sec pretrain step (ViT-L+RVSA, %d^2, bf16)" % args.image_size, "value": None, "unit": "images/sec",
                       "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "error": msg})


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (WORLD_SIZE unset): start the N ranks here, the way the
    reference starts its own (torch.distributed launch, main_pretrain.py:121-140, 508-518): one process per GPU, rendezvous on
    127.0.0.1.  The children's output is passed through; the ONE JSON line of rank 0 is re-printed last.  Returns the exit code."""
    import subprocess
    if not args.cpu_standin:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            print(_error_line(args, "--gpus %d but only %d GPU(s) visible to this process" % (args.gpus, have)), flush=True)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("GPU_MAX_HW_QUEUES", "8")              # compute, weight-gradient, exchange and RCCL streams on separate hardware queues (mtp_amd/__init__.py)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = proc.stdout.splitlines()
    result = None
    for i in range(len(lines) - 1, -1, -1):
        ln = lines[i].strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                if "metric" in json.loads(ln):
                    result = lines.pop(i)
                    break
            except ValueError:
                pass
    for ln in lines:
        print(ln)
    if result is None:
        result = _error_line(args, "the %d-rank launch exited with code %d without printing a result line" % (args.gpus, proc.returncode))
    print(result, flush=True)
    return proc.returncode if proc.returncode else (0 if "\"error\"" not in result else 3)


def run_cpu_standin(args, world, rank):
    """TEST STAND-IN, not a measurement (tests/test_bench_launch.py): the same launch / rendezvous / report plumbing with gloo ranks
    on CPU.  The HIP engine cannot run here, so a step = filling the flat gradient buffer of a small backbone and driving GradReducer
    with the engine's completion order (FPN tail, bursts of blocks, embeddings) -- exactly what DataParallelTrainer.step does around
    the kernels.  Prints the bench JSON schema with `standin: true` and a `comm` object."""
    import torch.distributed as dist
    import mtp_amd
    from mtp_amd.parallel import FlatParams, GradReducer
    dist.init_process_group("gloo")
    torch.manual_seed(2023)
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(embed_dim=128, depth=6, num_heads=2, interval=3, qkv_bias=True, use_abs_pos_emb=True, out_indices=[1, 2, 3, 5])
    flat = FlatParams(net, unused=net._unused_params)
    red = GradReducer(flat, bucket_bytes=1 << 20, mode=args.comm_mode, bf16=args.comm_bf16)
    gen = torch.Generator().manual_seed(100 + rank)
    bursts = [[6], [5, 4, 3, 2], [1], [0], [-1]]     # the engine's reports: FPN tail, a burst of blocks, split_last's block 1 / block 0, embeddings

    def step():
        flat.grad.copy_(torch.randn(flat.total, generator=gen))
        red.begin_step()
        for burst in bursts:
            for g in burst:
                red.on_block_done(g)
        red.finish()
    for _ in range(args.warmup):
        step()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dist.barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    # every rank holds the same reduced gradients
    chk = flat.grad[:flat.reduced].double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    out = {"metric": "images/sec pretrain step (ViT-L+RVSA, %d^2, bf16)" % args.image_size, "value": round(world * args.batch * args.steps / dt, 2),
           "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "standin": True,
           "config": {"workload": "CPU / gloo STAND-IN of the launch + gradient-exchange plumbing (no kernels, not a measurement)",
                      "global_batch": world * args.batch, "parallelism": "dp%d" % world},
           "comm": dict(ranks=dist.get_world_size(), backend=dist.get_backend(), mode=red.mode, bf16=bool(red.bf16), collectives_per_step=red.collectives,
                        bytes_per_step=int(red.bytes_reduced), replicas_identical=bool(lo.item() == hi.item()))}
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


class ClockSampler:
    """sclk / socket power of the GPU while the timed region runs (verdict r02 #2: "record sclk + power ... so the power-limited
    argument is evidence"): a thread reads the amdgpu hwmon files of the device (matched by PCI address) every `period` seconds --
    two small sysfs reads, no GPU work, no process spawn."""

    def __init__(self, device_index=0, period=0.05):
        import glob
        import threading
        self.dir, self.period, self.f, self.p, self.cap = None, period, [], [], None
        self._stop, self._th = threading.Event(), None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if not os.path.exists(os.path.join(h, "freq1_input")):
                continue
            pci = os.path.basename(os.path.realpath(os.path.join(h, "..", "..")))       # .../0000:bb:dd.f
            if want is None or pci.lower().startswith(want):
                self.dir = h
                break
        if self.dir:
            try:
                self.cap = int(open(os.path.join(self.dir, "power1_cap")).read()) / 1e6
            except Exception:
                self.cap = None

    def _read(self, name):
        try:
            return int(open(os.path.join(self.dir, name)).read())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            f, p = self._read("freq1_input"), self._read("power1_input")
            if f is not None:
                self.f.append(f / 1e6)
            if p is not None:
                self.p.append(p / 1e6)
            self._stop.wait(self.period)

    def start(self):
        import threading
        if self.dir:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def stop(self):
        if self._th is None:
            return None
        self._stop.set()
        self._th.join(timeout=1.0)
        if not self.f:
            return None
        out = dict(sclk_mhz=dict(avg=round(sum(self.f) / len(self.f)), min=round(min(self.f)), max=round(max(self.f))), samples=len(self.f),
                   source=self.dir + "/{freq1_input,power1_input}, every %d ms over the timed region" % int(self.period * 1e3))
        if self.p:
            out["power_w"] = dict(avg=round(sum(self.p) / len(self.p)), max=round(max(self.p)), cap=self.cap)
        return out


def _traffic_from_profiles(dom):
    """HBM bytes per launch of the dominant kernel family from the committed PMC passes -- only when that file was taken at sources
    identical to what is running: its `_csrc_sha` must equal the hash of mtp_amd/csrc now (verdict r02 #7: refuse otherwise)."""
    import glob
    from tools.pmc_hbm import csrc_sha
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    if not files:
        return None, None
    path = files[-1]
    pmc = json.load(open(path))
    sha = csrc_sha()
    rel = os.path.relpath(path, ROOT)
    if pmc.get("_csrc_sha") != sha:
        return None, "%s was taken at other kernel sources (csrc hash %s, now %s): not quoted" % (rel, pmc.get("_csrc_sha", "none"), sha)
    return pmc[dom + "_kernel"]["hbm_bytes_per_launch"], "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, csrc hash %s, commit %s)" % (rel, sha, pmc.get("_commit", "?"))


def make_loss_and_grads(heads, chans, device="cuda"):
    """the loss that stands in for the task decoders and the cotangents of the four maps it hands to the backbone's backward (`--heads`).
    Returns f(feats) -> (loss, [d loss / d f_i]); tests/test_hip_backbone.py checks every variant's cotangents against torch autograd on the same maps."""
    if heads == "standin3":
        # Three STAND-IN task heads (semantic segmentation / instance segmentation / rotated detection of models.py:112-179, 329-335,
        # which call mmseg / mmdet / mmrotate decoders that are not vendored): head t scores every pixel of every map with its own
        # 1x1 projection w[t][i] (C -> 1) and averages -- three consumers of the four maps with their own parameters and a
        # per-channel cotangent, nothing more.  Labelled as stand-ins in `config`.
        gh = torch.Generator(device=device).manual_seed(7)
        head_w = [[torch.randn(c, device=device, generator=gh) / c ** 0.5 for c in chans] for _ in range(3)]

        def loss_and_grads(feats):
            loss, grads = 0.0, []
            for i, f in enumerate(feats):
                wsum = head_w[0][i] + head_w[1][i] + head_w[2][i]
                pix = f.shape[0] * f.shape[2] * f.shape[3]
                loss = loss + (f.sum(dim=(0, 2, 3), dtype=torch.float32) * wsum).sum() / pix
                grads.append((wsum / pix).to(f.dtype).view(1, -1, 1, 1).expand_as(f).contiguous())
            return loss, grads
        loss_and_grads.head_w = head_w
        return loss_and_grads
    if heads == "standin_seg":
        # ONE stand-in segmentation head (the UperNet decoder of FT/Semantic_Segmentation/configs/mtp/loveda/*.py lives in un-vendored mmseg): every map gets
        # its own 1x1 projection to 7 classes (LoveDA) and a per-pixel cross-entropy against fixed random labels at its own resolution; the cotangents
        # of the maps come from torch autograd, so -- unlike `mean` -- they differ from pixel to pixel.  Labelled as a stand-in in `config`.
        gh = torch.Generator(device=device).manual_seed(11)
        seg = {}

        def loss_and_grads(feats):
            loss, leaves = 0.0, []
            for i, f in enumerate(feats):
                if i not in seg:
                    c = f.shape[1]
                    seg[i] = (torch.randn(7, c, device=device, generator=gh) / c ** 0.5,
                              torch.randint(0, 7, (f.shape[0], f.shape[2], f.shape[3]), device=device, generator=gh))
                w, y = seg[i]
                fl = f.detach().requires_grad_(True)
                logits = torch.einsum("bchw,kc->bkhw", fl.float(), w)
                loss = loss + torch.nn.functional.cross_entropy(logits, y)
                leaves.append(fl)
            grads = torch.autograd.grad(loss, leaves)
            return loss.detach(), [g.to(f.dtype).contiguous() for g, f in zip(grads, feats)]
        loss_and_grads.seg = seg
        return loss_and_grads

    def loss_and_grads(feats):
        # stand-in for the three task decoders: loss = sum_i mean(f_i), d loss / d f_i = 1 / numel(f_i), written out by hand
        # (one f32-accumulating reduction + one fill per map, every step) instead of through autograd's f32 copies of the maps
        loss = sum(f.sum(dtype=torch.float32) / f.numel() for f in feats)
        return loss, [torch.full_like(f, 1.0 / f.numel()) for f in feats]
    return loss_and_grads


def _hbm_fractions_from_profiles():
    """per-kernel rate against the 8 TB/s HBM peak for the HBM-bound kernels of this workload, from the committed table of
    tools/hbm_fractions.py (rocprofv3 single-stream kernel statistics joined with the PMC passes) -- quoted only when it was taken at the kernel sources
    that are running, like `traffic`"""
    import glob
    from tools.pmc_hbm import csrc_sha
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_fractions.json")))
    if not files:
        return None
    tab = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if tab.get("_csrc_sha") != csrc_sha():
        return "%s was taken at other kernel sources (csrc hash %s): not quoted" % (rel, tab.get("_csrc_sha", "none"))
    return dict(source=rel, peak_TBps=tab["peak_TBps"], unit="TB/s on the algorithmic bytes of one launch",
                kernels={k: dict(us=v["us_per_launch"], TBps=v["TBps"], frac=v["frac"], counter_over_algorithmic=(round(v["counter_bytes"] / v["algorithmic_bytes"], 2) if v.get("counter_bytes") else None))
                         for k, v in tab["kernels"].items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="vit_l", choices=["vit_l", "vit_b", "internimage_xl"],
                    help="internimage_xl = BASELINE configs[4]'s backbone (use --image-size 512 --batch 8 --no-cpu-baseline)")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-timer", action="store_true")
    ap.add_argument("--no-forward-only", dest="forward_only", action="store_false",
                    help="skip the extra pass that times the forward alone (inference mode, no saved activations; SURVEY 8d asks for forward-only "
                         "numbers next to the step).  It runs AFTER the timed region and is reported as `forward_only`, never as `value`")
    ap.add_argument("--forward-only", dest="forward_only", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.set_defaults(forward_only=True)
    ap.add_argument("--use-ckpt", action="store_true", help="activation checkpointing (`use_ckpt='True'`, VIT:799-800): every block's forward is recomputed in the "
                    "backward -- the recipe MTP pretrains with at 448^2 (Readme.md:233-240).  The recomputation is NOT counted in the step's flops")
    ap.add_argument("--gemm-shapes", action="store_true", help="print the per-shape table of the instrumented GEMM launches to stderr")
    ap.add_argument("--timer-every", type=int, default=0,
                    help="0 (default since round 6): no step of the timed region carries per-launch HIP events -- `value` times the step alone; the events behind "
                         "`roofline` (one stream) and `roofline.concurrent` (as the step runs) are recorded on extra steps right AFTER the timed region.  N > 0: "
                         "additionally instrument every N-th timed step the old way (two events per GEMM launch cost ~1.3 ms per instrumented step)")
    ap.add_argument("--host-input", action="store_true", help="additionally time the step fed from HOST uint8 batches (pinned staging, side-stream H2D, "
                                                              "fused preprocess): reported as `host_input`, never as `value`")
    ap.add_argument("--heads", default="mean", choices=["mean", "standin3", "standin_seg"],
                    help="what consumes the four feature maps: `mean` = sum_i mean(f_i) (the headline line); `standin3` = three labelled stand-in task "
                         "heads (BASELINE configs[1]: 'ViT-B/16 + 3 MTP decoder heads'; the real decoders live in un-vendored mmseg / mmdet / mmrotate); `standin_seg` = one "
                         "labelled stand-in segmentation head (BASELINE configs[4]: 'InternImage-XL ... segmentation decoder'): per-map 1x1 class projection + "
                         "cross-entropy against fixed random labels, through torch autograd")
    ap.add_argument("--image-size", type=int, default=224, help="224 = the headline metric; 448 = what MTP actually pretrains at (use --batch 16)")
    ap.add_argument("--comm-mode", default=os.environ.get("MTP_COMM_MODE", "allreduce"), choices=["allreduce", "rs_ag"],
                    help="gradient exchange per bucket: one all-reduce, or reduce-scatter + all-gather (mtp_amd.parallel.GradReducer)")
    ap.add_argument("--comm-bf16", action="store_true", default=os.environ.get("MTP_COMM_BF16") == "1",
                    help="exchange the gradient buckets as bf16 (cast on the side stream, f32 again before the optimizer): half the xGMI bytes")
    ap.add_argument("--wgrad-side-stream", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="A/B: grouped weight-gradient launches on a side stream (1) or the main stream (0); -1 = the engines' own defaults")
    ap.add_argument("--wgrad-keep", type=int, default=-1, help="A/B: weight-gradient bursts that may stay in flight on the side stream (InternImage)")
    ap.add_argument("--wgrad-max-jobs", type=int, default=-1)
    ap.add_argument("--cpu-standin", action="store_true",
                    help="TEST ONLY: gloo ranks on CPU exercising the launcher / rendezvous / comm-report plumbing without kernels (prints `standin: true`)")
    args = ap.parse_args()

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver (read when HIP initialises)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")              # see mtp_amd/__init__.py
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us (the driver runs `python bench.py --gpus N ...`): start the ranks ourselves
        raise SystemExit(self_launch(args, sys.argv[1:]))
    if world != args.gpus:
        if rank == 0:
            print(_error_line(args, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)), flush=True)
        raise SystemExit(2)
    if rank != 0:
        # stdout carries rank 0's ONE result line: whatever the other ranks (or the libraries under them -- RCCL prints a
        # version banner through C stdio, flushed when the process exits) write to fd 1 goes to stderr instead
        sys.stdout.flush()
        os.dup2(2, 1)
    if args.cpu_standin:
        return run_cpu_standin(args, world, rank)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local:
        if rank == 0:
            print(_error_line(args, "rank %d has no GPU (%d visible)" % (local, torch.cuda.device_count() if torch.cuda.is_available() else 0)), flush=True)
        raise SystemExit(2)
    torch.cuda.set_device(local)
    if args.wgrad_side_stream >= 0:
        from mtp_amd.engine import BackboneEngine
        from mtp_amd.engine_intern import InternEngine
        BackboneEngine.wgrad_side_stream = InternEngine.wgrad_side_stream = args.wgrad_side_stream
    if args.wgrad_keep >= 0:
        from mtp_amd.engine import BackboneEngine
        from mtp_amd.engine_intern import InternEngine
        BackboneEngine.wgrad_keep = InternEngine.wgrad_keep = args.wgrad_keep
    if args.wgrad_max_jobs >= 0:
        from mtp_amd.engine import BackboneEngine
        from mtp_amd.engine_intern import InternEngine
        BackboneEngine.wgrad_max_jobs = InternEngine.wgrad_max_jobs = args.wgrad_max_jobs
    import torch.distributed as dist
    force_comm = os.environ.get("MTP_FORCE_COMM") == "1"     # debugging aid: run the RCCL path on a single GPU
    rccl_log = None
    if world > 1 or force_comm:
        # what RCCL chose (algorithm / protocol / channels) goes into `comm.rccl`: rank 0's INFO log, written to a file and parsed after the run.  The image
        # exports NCCL_DEBUG=VERSION (round 5: that is why `rccl` was null in the first forced-comm lines); a level the user picked for a real log stays.
        if os.environ.get("NCCL_DEBUG", "").upper() in ("", "VERSION", "WARN") and os.environ.get("MTP_RCCL_LOG", "1") != "0":
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), "mtp_rccl_%d.%d.log" % (os.getpid(), rank))
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,TUNING"      # (no COLL: one formatted line per collective would land inside the timed region)
            os.environ["NCCL_DEBUG_FILE"] = rccl_log
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import mtp_amd
    from mtp_amd import ops
    from mtp_amd.parallel import DataParallelTrainer
    ops.lib()

    class A:
        image_size = args.image_size
        use_ckpt = "True" if args.use_ckpt else "False"
        precision = args.precision
    torch.manual_seed(2023)    # identical initial replicas (main_pretrain.py:107)
    if args.model == "internimage_xl":
        # models.py:92-104 (drop_path 0.2).  The reference's factory checkpoints every layer (with_cp=True) to fit its 16-GB devices; like the ViT lines, the
        # benchmark keeps the activations (288 GB of HBM) unless --use-ckpt asks for the reference's memory recipe -- `config.activation_checkpointing` says which
        net = mtp_amd.internimage_xl(precision=args.precision, with_cp=bool(args.use_ckpt))
        with torch.no_grad():      # the zero-initialised offset / mask heads re-drawn so the sampling really deforms (as fixture f12 does)
            for n, p in net.named_parameters():
                if ".dcn.offset.weight" in n or ".dcn.mask.weight" in n:
                    p.normal_(0, 0.02)
    else:
        net = (mtp_amd.vit_l_rvsa if args.model == "vit_l" else mtp_amd.vit_b_rvsa)(A)
    with torch.no_grad():      # zero-initialised tables re-drawn N(0, 0.02^2) so no branch is trivially zero
        for n, p in net.named_parameters():
            if "rel_pos" in n:
                p.normal_(0, 0.02)
    net = net.cuda().train()
    fdt = torch.bfloat16 if args.precision == "bf16" else torch.float32
    trainer = DataParallelTrainer(net, lr=6e-5, weight_decay=0.05, max_norm=5.0, total_steps=1000, feature_dtype=fdt,
                                  comm_mode=args.comm_mode, comm_bf16=args.comm_bf16)
    torch.manual_seed(2023 + rank)   # per-rank data / drop-path streams (main_pretrain.py:517)
    B = args.batch
    img = torch.randn(B, 3, args.image_size, args.image_size, device="cuda")

    loss_and_grads = make_loss_and_grads(args.heads, [net.embed_dim] * 4 if hasattr(net, "embed_dim") else list(net.out_channels))      # (InternImage: 192 / 384 / 768 / 1536)

    timer = GemmTimer(ops)
    if not args.no_gemm_timer:
        timer.install()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(img, loss_and_grads)
    sync()
    sampler = ClockSampler(torch.cuda.current_device()) if rank == 0 else None
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    timed_steps = 0
    # Instrumented steps INSIDE the timed region (every --timer-every-th) run exactly as every other step does -- the weight-gradient bursts on their side
    # stream, where a data-gradient GEMM shares the CUs with weight-gradient tiles and its events measure the pair: `roofline.concurrent`.  The steps behind
    # `roofline` proper run every launch on the one compute stream, so that the events around a GEMM launch time that kernel alone; they change how the step
    # runs and therefore come AFTER the timed region (the timed loop used to mix the two and flip a class attribute mid-run).
    eng_cls = type(trainer.engine)
    side_default = getattr(eng_cls, "wgrad_side_stream", False)
    conc_steps = 0
    for i in range(args.steps):
        timer.on = (not args.no_gemm_timer) and args.timer_every > 0 and i % args.timer_every == 0
        if timer.on:
            timer.use(1 if side_default else 0)
            if side_default:
                conc_steps += 1
            else:
                timed_steps += 1
        loss = trainer.step(img, loss_and_grads)
    sync()
    dt = time.perf_counter() - t0
    clocks = sampler.stop() if sampler is not None else None
    timer.on = False
    if not args.no_gemm_timer:
        # the instrumented steps, right behind the timed region (same weights, same buffers, same clocks): first as the timed steps ran (`concurrent`,
        # engines with a weight-gradient side stream only), then with every launch on the one compute stream (`roofline` proper)
        n_inst = max(2, min(4, args.steps // 5))
        if side_default:
            timer.on = True
            timer.use(1)
            for _ in range(n_inst):
                trainer.step(img, loss_and_grads)
            sync()
            conc_steps += n_inst
            eng_cls.wgrad_side_stream = False
        try:
            timer.on = True
            timer.use(0)
            for _ in range(n_inst):
                trainer.step(img, loss_and_grads)
            sync()
            timed_steps += n_inst
        finally:
            eng_cls.wgrad_side_stream = side_default
            timer.on = False
    tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt

    comm = None
    if trainer.reducer.active:
        # communication report (N > 1, or MTP_FORCE_COMM=1): the same steps once more with HIP events around every collective on the
        # side stream, and once with the collectives switched off -- the difference is what the overlap does not hide
        red = trainer.reducer
        red.timing, red.timed = True, []
        for _ in range(args.steps):
            trainer.step(img, loss_and_grads)
        sync()
        red.timing = False
        nbytes = sum(b for b, _, _ in red.timed) / args.steps
        secs = sum(a.elapsed_time(bb) for _, a, bb in red.timed) * 1e-3 / args.steps
        ncoll = red.collectives          # collectives issued in the last step (2 per bucket in rs_ag mode)
        wire = red.wire_bytes
        red.active = False
        t2 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(img, loss_and_grads)
        sync()
        tnc = torch.tensor([time.perf_counter() - t2], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tnc, op=dist.ReduceOp.MAX)
        red.active = True
        ms_nocomm = float(tnc.item()) / args.steps * 1e3
        busf = 2.0 * (world - 1) / world if world > 1 else 1.0
        rccl = parse_rccl_log(rccl_log) if (rank == 0 and rccl_log) else None
        if rccl_log:
            try:
                os.unlink(rccl_log)
            except OSError:
                pass
        ex = red.describe()
        if isinstance(ex.get("communicator"), dict):
            seen = ex["communicator"].get("nranks")       # what RCCL says about the communicator the collectives ran on (mtp_comm_info), on every rank
        else:
            # torch.distributed's collectives (MTP_NATIVE_COMM=0, or the agreed second choice when mtp_comm_init failed): every rank adds 1 through the
            # process group's RCCL communicator; rank 0 also has RCCL's own log line, which wins when the two disagree
            one = torch.ones(1, device="cuda")
            dist.all_reduce(one)
            seen = int(round(float(one.item())))
            if rank == 0 and rccl and rccl.get("nranks") not in (None, seen):
                seen = rccl.get("nranks")
        if world > 1 and seen != world:
            # the exchange did not run over the communicator the line would claim: no result line
            if rank == 0:
                print(_error_line(args, "--gpus %d but the RCCL communicator of the gradient exchange reports %r ranks" % (world, seen)), flush=True)
            raise SystemExit(4)
        comm = dict(ranks=dist.get_world_size(), rccl_nranks=seen, backend=dist.get_backend(), collectives_per_step=ncoll, bytes_per_step=int(nbytes), exposed_ms=round(ms - ms_nocomm, 3), rccl=rccl,
                    wire_bytes_per_step=int(wire), exchange=ex,
                    allreduce_ms_per_step=round(secs * 1e3, 3), bus_GBps=round(wire * busf / max(secs, 1e-9) / 1e9, 1),
                    xgmi_peak_GBps=7 * 153, ms_per_step_without_comm=round(ms_nocomm, 3), exposed_comm_ms=round(ms - ms_nocomm, 3),
                    note="all-reduce time = HIP events on the side stream around each collective (its own duration, overlapped with the "
                         "backward); exposed = step time with minus without collectives; bus GB/s = bytes x 2(N-1)/N / all-reduce time")

    forward_only = None
    if args.forward_only:
        eng = trainer.engine
        for _ in range(args.warmup):
            eng.forward(img, training=False, need_grad=False, feature_dtype=fdt)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng.forward(img, training=False, need_grad=False, feature_dtype=fdt)
        sync()
        tf = torch.tensor([time.perf_counter() - t1], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        dtf = float(tf.item())
        gff = FWD_GF_PER_IMAGE.get(args.model, 0.0)
        forward_only = dict(value=round(world * B * args.steps / dtf, 2), unit="images/sec", ms_per_pass=round(dtf / args.steps * 1e3, 3),
                            mfma_frac=(round(B * args.steps / dtf * gff * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4) if (args.image_size == 224 and gff) else None),
                            note="forward pass alone (inference mode: no activations saved, no drop path), same weights / input / feature dtype as the step")

    host_input = None
    if args.host_input:
        # PCIe-inclusive variant of the same step: raw uint8 HWC batches start in pageable HOST memory, are staged
        # through pinned buffers and copied on a side stream while the previous step computes; normalise / flip / pad / im2col
        # run on the device inside the patch-embed kernel.  Reported next to `value`, never instead of it.
        import itertools
        from mtp_amd.data import HostBatchPrefetcher
        net.set_data_preprocessor()
        pool = [torch.randint(0, 256, (B, args.image_size, args.image_size, 3), dtype=torch.uint8) for _ in range(4)]
        pf = HostBatchPrefetcher(itertools.islice(itertools.cycle(pool), args.warmup + args.steps), device="cuda", depth=2)
        for _ in range(args.warmup):
            trainer.step(next(pf), loss_and_grads)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(next(pf), loss_and_grads)
        sync()
        dth = time.perf_counter() - t1
        th = torch.tensor([dth], device="cuda", dtype=torch.float64)
        if world > 1:
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
        dth = float(th.item())
        host_input = dict(value=round(world * B * args.steps / dth, 2), unit="images/sec", ms_per_step=round(dth / args.steps * 1e3, 3),
                          h2d_bytes_per_step=B * args.image_size * args.image_size * 3,
                          note="same step fed from pageable host uint8 (B,H,W,3) batches: pinned staging + side-stream H2D + fused preprocess")

    if rank == 0:
        if args.gemm_shapes:
            print("\n".join(timer.shapes()), file=sys.stderr)
        fams = timer.summary()
        roof = None
        if fams:
            dom = max(fams, key=lambda k: fams[k]["seconds"])
            d = fams[dom]
            ach = d["flops"] / d["seconds"] / 1e12
            traffic, traffic_src = None, None   # HBM bytes per launch of the dominant family, from the committed PMC passes (separate runs)
            if args.model == "vit_l" and args.precision == "bf16" and B == 64 and args.image_size == 224:
                try:
                    traffic, traffic_src = _traffic_from_profiles(dom)
                except Exception as e:
                    traffic, traffic_src = None, "unreadable: %s" % e
            roof = dict(bound="mfma", kernel=dom, achieved=round(ach, 1), peak=PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3,
                        unit="TFLOP/s", frac=round(ach / (PEAK_BF16_TFLOPS if args.precision == "bf16" else 157.3), 4), traffic=traffic, traffic_source=traffic_src,
                        flops_per_launch=round(d["flops"] / d["launches"]),
                        avg_launch_us=round(d["seconds"] / d["launches"] * 1e6, 1), launches_per_step=d["launches"] // max(1, timed_steps),
                        instrumented_steps=timed_steps,
                        instrumented_where=("extra steps run right after the timed region (none of the timed steps carries events: `value` is the step alone)"
                                            + ("; every launch on ONE stream, so a launch's events time that kernel alone -- `concurrent` = the same events on steps run as "
                                               "the timed steps are (weight-gradient bursts on the side stream)" if side_default else "")
                                            + ("; plus every %d-th timed step (--timer-every)" % args.timer_every if args.timer_every > 0 else "")),
                        families={k: dict(tflops=round(v["flops"] / v["seconds"] / 1e12, 1), ms_per_step=round(v["seconds"] / max(1, timed_steps) * 1e3, 2))
                                  for k, v in fams.items()})
            if conc_steps:
                fc = timer.summary(1)
                roof["concurrent"] = dict(
                    note="the same events in a step run as the timed steps are: weight-gradient bursts on the side stream, so a launch shares the CUs with "
                         "the other family's tiles and its duration is no longer the kernel's own (`roofline` proper comes from instrumented steps run on one stream)",
                    instrumented_steps=conc_steps,
                    families={k: dict(tflops=round(v["flops"] / v["seconds"] / 1e12, 1), ms_per_step=round(v["seconds"] / conc_steps * 1e3, 2),
                                      avg_launch_us=round(v["seconds"] / v["launches"] * 1e6, 1)) for k, v in fc.items()})
        gf = FWD_GF_PER_IMAGE.get(args.model, 0.0) * 3.0
        step_frac = round(value / world * gf * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 4) if (args.image_size == 224 and gf and args.precision == "bf16") else None
        if roof is not None:
            # the number north_star targets (>= 0.70): the whole step's algorithmic flops (389.9 GF per image: forward + dgrad + wgrad of every dense contraction) over
            # the measured step time, against the dense bf16 MFMA peak -- `frac` above is the dominant kernel family alone
            roof["step_frac"] = step_frac
            roof["step_flops_per_image"] = gf * 1e9 if gf else None
            if args.model == "vit_l" and args.precision == "bf16" and B == 64 and args.image_size == 224:
                try:
                    roof["hbm_bound_kernels"] = _hbm_fractions_from_profiles()
                except Exception as e:
                    roof["hbm_bound_kernels"] = "unreadable: %s" % e
        label = {"vit_l": "ViT-L + RVSA", "vit_b": "ViT-B + RVSA", "internimage_xl": "InternImage-XL (DCNv3)"}[args.model]
        out = {
            "metric": ("images/sec pretrain step (ViT-L+RVSA, %d^2, bf16)" if args.model == "vit_l" else "images/sec pretrain step (ViT-B+RVSA, %d^2)" if args.model == "vit_b"
                       else "images/sec backbone train step (InternImage-XL, %d^2)") % args.image_size,
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "%s backbone fwd+bwd + grad all-reduce + clip + AdamW, %dx%d, batch %d per GPU%s"
                                   % (label, args.image_size, args.image_size, B,
                                      " (BASELINE configs[4]'s backbone and tile size; stand-in loss instead of the segmentation decoder)" if args.model == "internimage_xl" and args.image_size == 512
                                      else (" (BASELINE configs[2]/[3])" if args.model == "vit_l" and B == 64 else " (BASELINE configs[1])" if args.model == "vit_b" and B == 32 else "")
                                      if args.image_size == 224 else " (not the headline configuration)"),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "loss": float(loss),
                       "activation_checkpointing": bool(args.use_ckpt),
                       "heads": ("3 stand-in task heads (per-map 1x1 projection + mean each; the mm* decoders are not vendored)" if args.heads == "standin3"
                                 else "1 stand-in segmentation head (per-map 1x1 projection to 7 classes + per-pixel cross-entropy vs fixed random labels, torch autograd; "
                                      "mmseg's UperNet is not vendored)" if args.heads == "standin_seg" else "sum_i mean(f_i)")},
            "step_mfma_frac": step_frac,
            "roofline": roof,
        }
        if clocks is not None:
            out["clocks"] = clocks
            if roof is not None and args.precision == "bf16":
                # the same achieved rate against the MFMA peak at the clock the chip actually sustained (2.5 PF is quoted at 2.4 GHz)
                roof["frac_at_measured_sclk"] = round(roof["achieved"] / (PEAK_BF16_TFLOPS * clocks["sclk_mhz"]["avg"] / 2400.0), 4)
        if comm is not None:
            out["comm"] = comm
        if host_input is not None:
            out["host_input"] = host_input
        if forward_only is not None:
            out["forward_only"] = forward_only
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model)
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line must be the LAST line of stdout: RCCL prints a version banner through C stdio, which would otherwise be
        # flushed at process exit, after Python's print
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
