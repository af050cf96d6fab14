"""Which op is not run-to-run deterministic?  Wraps every function of mtp_amd.ops: after each call (device synchronised) a
checksum of every tensor argument (sum and sum of squares in f64) is logged.  The same step is run N times; the first log
entry that differs from run 0 names the op whose OUTPUT changed while all earlier entries (its inputs) were identical.
usage: python tools/probes/race_finder.py [runs]"""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import mtp_amd
from mtp_amd import ops

LOG = []


def _sig(t):
    return t.detach().float().clone()


def _wrap(name, fn):
    def inner(*a, **k):
        out = fn(*a, **k)
        torch.cuda.synchronize()
        ts = [x for x in list(a) + list(k.values()) if torch.is_tensor(x) and x.is_cuda and x.numel()]
        LOG.append((name, [tuple(t.shape) for t in ts], [_sig(t) for t in ts]))
        return out
    return inner


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for n in dir(ops):
        f = getattr(ops, n)
        if isinstance(f, types.FunctionType) and not n.startswith("_") and f.__module__ == ops.__name__:
            setattr(ops, n, _wrap(n, f))
    torch.manual_seed(0)
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, embed_dim=256, depth=8, num_heads=4, interval=4, qkv_bias=True, use_abs_pos_emb=True,
                                       out_indices=[1, 3, 5, 7], drop_path_rate=0.0, precision="bf16").cuda()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(7)).cuda()
    from mtp_amd.parallel import FlatParams
    eng = net._engine()
    flat = FlatParams(net, unused=net._unused_params)
    ref = None
    bad = {}
    for it in range(runs):
        LOG.clear()
        flat.grad.zero_()
        feats, ctx = eng.forward(x, training=True, need_grad=True)
        eng.backward(ctx, [torch.full_like(f, 1.0 / f.numel()) for f in feats], flat.G)
        torch.cuda.synchronize()
        if it == 0:
            continue        # builds the weight images
        if ref is None:
            ref = list(LOG)
            print("ops per step:", len(ref))
            continue
        assert len(LOG) == len(ref)
        level = 1e-9
        for i, (a, b) in enumerate(zip(LOG, ref)):
            for j, (t1, t0) in enumerate(zip(a[2], b[2])):
                d = float((t1 - t0).abs().max() / (t0.abs().max() + 1e-30))
                if d > 8 * level:       # report every op at which the run-to-run difference jumps by ~an order of magnitude
                    print("run %d op %3d %-22s arg %d %-18s rel diff %.2e" % (it, i, a[0], j, a[1][j], d))
                    level = d
                    bad[a[0]] = bad.get(a[0], 0) + 1
    print("divergent runs by op:", bad)


if __name__ == "__main__":
    main()
