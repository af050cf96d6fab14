"""outputs of the RVSA backward at the ViT-L geometry into a file (run once per kernel: MTP_RVSA_BWD=4 = the kernel of rounds 2-5), or compare two such files bit by bit.
   python tools/probes/rvsa_bwd_bits.py dump out.pt | python tools/probes/rvsa_bwd_bits.py cmp a.pt b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def dump(path):
    from mtp_amd import ops
    torch.manual_seed(0)
    B, Hp, Wp, H, C = 8, 14, 14, 16, 1024
    T = B * Hp * Wp
    dev = "cuda"
    bf = torch.bfloat16
    qkv = (torch.randn(T, 3 * C, device=dev) * 0.5).to(bf)
    samp = torch.randn(B * 4, 5 * H, device=dev) * 0.2
    o, lse = torch.empty(T, C, device=dev, dtype=bf), torch.empty(B * 4 * H * 49, device=dev)
    r13a, r13b, tab = torch.randn(13, 64, device=dev) * 0.1, torch.randn(13, 64, device=dev) * 0.1, torch.randn(169, H, device=dev) * 0.1
    ops.rvsa_attn_fwd(qkv, samp, o, lse, r13a, r13b, tab, B, Hp, Wp, H, 0.125)
    do = (torch.randn(T, C, device=dev) * 0.1).to(bf)
    dqkv, dsamp = torch.empty(T, 3 * C, device=dev, dtype=bf), torch.empty(B * 4, 5 * H, device=dev)
    ga, gb, gt = torch.zeros(13, 64, device=dev), torch.zeros(13, 64, device=dev), torch.zeros(169, H, device=dev)
    ops.rvsa_attn_bwd(qkv, samp, o, do, lse, dqkv, dsamp, r13a, r13b, tab, ga, gb, gt, B, Hp, Wp, H, 0.125)
    torch.cuda.synchronize()
    torch.save(dict(o=o.cpu(), lse=lse.cpu(), dqkv=dqkv.cpu(), dsamp=dsamp.cpu(), rel_h=ga.cpu(), rel_w=gb.cpu(), tab=gt.cpu()), path)


def cmp(a, b):
    x, y = torch.load(a), torch.load(b)
    for k in x:
        same = torch.equal(x[k], y[k])
        d = (x[k].float() - y[k].float()).abs().max().item() / (x[k].float().abs().max().item() + 1e-30)
        print("%-6s %s  max |diff| / max |x| = %.3g" % (k, "bit-identical" if same else "DIFFERENT", d))


if __name__ == "__main__":
    dump(sys.argv[2]) if sys.argv[1] == "dump" else cmp(sys.argv[2], sys.argv[3])
