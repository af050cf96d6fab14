"""Run-to-run determinism screen: the same step N times, max relative difference of every gradient against run 0.  f32 atomics
(RVSA scatter, colsum) reorder sums: ~1e-7.  Anything near 1e-4 is a race.  usage: python tools/probes/determinism.py [runs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import mtp_amd


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.manual_seed(0)
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=224, embed_dim=256, depth=8, num_heads=4, interval=4, qkv_bias=True, use_abs_pos_emb=True,
                                       out_indices=[1, 3, 5, 7], drop_path_rate=0.0, precision="bf16").cuda()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(7)).cuda()
    from mtp_amd.parallel import FlatParams
    eng = net._engine()
    flat = FlatParams(net, unused=net._unused_params)
    names = [n for n, _ in net.named_parameters()]
    ref = None
    worst = {}
    for it in range(runs):
        flat.grad.zero_()
        feats, ctx = eng.forward(x, training=True, need_grad=True)
        eng.backward(ctx, [torch.full_like(f, 1.0 / f.numel()) for f in feats], flat.G)
        torch.cuda.synchronize()
        g = {n: flat.G[n].clone() for n in flat.G}
        f0 = [f.clone() for f in feats]
        if ref is None:
            ref, fref = g, f0
            continue
        for a, b in zip(f0, fref):
            assert torch.equal(a, b), "forward not deterministic"
        for n in g:
            d = float((g[n] - ref[n]).abs().max() / (ref[n].abs().max() + 1e-30))
            worst[n] = max(worst.get(n, 0.0), d)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:12]
    for n, d in top:
        print("%-50s %.3e" % (n, d))


if __name__ == "__main__":
    main()
