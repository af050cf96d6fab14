"""Distance of the RVSA samples of a run to the nearest cell edge of the bilinear interpolation, measured through the ORACLE in float64.

F.grid_sample (VIT:397-404) is only piecewise differentiable: a sample whose pixel coordinate lies within f32 rounding of an integer takes one or the other
one-sided derivative depending on the last bit of the coordinate, and every gradient upstream of that block moves by up to 1e-3 of its maximum (round 2,
profiles/r02_parity_errors.json).  That is a property of the INPUT: parity tests that hold fp32 gradients to north_star's 1e-3 run on inputs
that have no such sample -- the same rule tests/golden/find_f13_seed.py applies to fixture f13 through the reference itself -- and keep the first input they
were written with as a second, looser "hard" case.

    python tests/golden/kinks.py 448 21 2 1 200      # image size, parameter seed, interval, first input seed, count   -> first seed with distance >= 1e-4 px
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import recipe  # noqa: E402
from oracle import vit_rvsa_oracle as O  # noqa: E402


def min_edge_distance(img, p, depth, heads, interval, out_indices):
    """min over every sample of every RVSA block of |pixel coordinate - nearest integer| (x and y), the whole forward evaluated in float64"""
    best = [1.0, 0]
    orig = O.rvsa_sample_coords

    def spy(samp, B, Hp, Wp, nheads):
        ix, iy = orig(samp, B, Hp, Wp, nheads)
        d = torch.minimum((ix - ix.round()).abs().min(), (iy - iy.round()).abs().min())
        best[0] = min(best[0], float(d))
        best[1] += 2 * ix.numel()
        return ix, iy
    O.rvsa_sample_coords = spy
    try:
        with torch.no_grad():
            O.backbone_forward(img.double(), {k: (v.double() if v.dtype.is_floating_point else v) for k, v in p.items()}, depth, heads, interval, out_indices)
    finally:
        O.rvsa_sample_coords = orig
    return best[0], best[1]


def small_model_params(size, seed, interval, embed_dim=128, depth=4, heads=2):
    import mtp_amd
    net = mtp_amd.ViT_Win_RVSA_V3_WSZ7(img_size=size, embed_dim=embed_dim, depth=depth, num_heads=heads, interval=interval, qkv_bias=True, use_abs_pos_emb=True,
                                       out_indices=[0, 1, 2, 3], precision="fp32", feature_dtype=torch.float32)
    return recipe.make_params({k: v.shape for k, v in net.state_dict().items() if v.dtype.is_floating_point}, seed=seed)


def main():
    size, pseed, interval = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    first = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    count = int(sys.argv[5]) if len(sys.argv) > 5 else 100
    thr = float(sys.argv[6]) if len(sys.argv) > 6 else 1e-4
    p = small_model_params(size, pseed, interval)
    for seed in range(first, first + count):
        img = recipe.make_input(1, size, size, seed=seed)
        d, n = min_edge_distance(img, p, 4, 2, interval, [0, 1, 2, 3])
        print("input seed %d: closest sample %.3e px from a cell edge (%d coordinates)" % (seed, d, n), flush=True)
        if d >= thr:
            print("CHOSEN %d %.6e" % (seed, d), flush=True)
            return


if __name__ == "__main__":
    main()
