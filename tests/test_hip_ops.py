"""GPU: every C-ABI op vs the oracle on seeded inputs.  f32 mode must meet north_star's 1e-3 (we assert 2e-4 relative to
the tensor's max); bf16 mode is checked against the oracle evaluated on the same bf16-rounded inputs."""
import os

import pytest
import torch

from conftest import ROOT, rel_err
from oracle import vit_rvsa_oracle as O

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-4, torch.bfloat16: 1.5e-2}


@pytest.fixture(scope="module")
def ops():
    from mtp_amd import ops as o
    o.lib()
    return o


def rnd(*shape, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    t = torch.randn(*shape, generator=g) * scale
    return t.to(dtype).float() if dtype == torch.bfloat16 else t    # values exactly representable in the op's dtype


def dev(t, dtype=None):
    return t.to("cuda", dtype=dtype or t.dtype).contiguous()


def e(*shape, dtype=torch.float32):
    return torch.empty(*shape, device="cuda", dtype=dtype)


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("variant", [0, 1, 32, 36])   # default / register-staged / 256x128 8-wave tile (+ grouped order)
@pytest.mark.parametrize("M,N,K", [(392, 384, 128), (300, 256, 192), (1024, 768, 768), (128, 128, 64)])
def test_gemm_nt_bias(ops, dtype, variant, M, N, K):
    a, w, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1), rnd(N, seed=2)
    out = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N, dtype=dtype), bias=dev(b), variant=variant)
    assert rel_err(out.float().cpu(), a @ w.t() + b) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_gemm_nt_asymmetric_identity(ops, dtype):
    """A = I with an asymmetric B catches transposed / permuted C fragments (guide rule 16)."""
    M = N = K = 128
    a = torch.eye(M)
    w = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 0.25 + 1).float()
    w = w.to(dtype).float()
    out = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N, dtype=torch.float32 if dtype == torch.float32 else dtype))
    assert torch.equal(out.float().cpu(), w.t().contiguous())


@pytest.mark.parametrize("dtype", DT)
def test_gemm_nt_epilogues(ops, dtype):
    M, N, K, rps = 392, 256, 128, 196
    a, w, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.2), rnd(N, seed=2)
    ref = a @ w.t() + b
    # GELU (+ pre-activation)
    u = e(M, N, dtype=dtype)
    h = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N, dtype=dtype), epi=ops.EPI_BIAS_GELU, bias=dev(b), aux=u)
    assert rel_err(u.float().cpu(), ref) < TOL[dtype] and rel_err(h.float().cpu(), O.gelu(ref)) < TOL[dtype]
    # residual + per-sample drop-path scale, f32 out
    res, rs = rnd(M, N, seed=3), torch.tensor([0.0, 1.0 / 0.9])
    out = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N), epi=ops.EPI_BIAS_RES, bias=dev(b), res=dev(res), rowscale=dev(rs), rows_per_sample=rps)
    assert rel_err(out.cpu(), res + rs.repeat_interleave(rps)[:, None] * ref) < TOL[dtype]
    # broadcast residual (pos_embed): res row = m % 196
    pos = rnd(rps, N, seed=4)
    out = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N), epi=ops.EPI_BIAS_RES, bias=dev(b), res=dev(pos), res_mod=rps)
    assert rel_err(out.cpu(), pos.repeat(2, 1) + ref) < TOL[dtype]
    # dGELU
    uu = rnd(M, N, dtype=dtype, seed=5)
    out = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N, dtype=dtype), epi=ops.EPI_DGELU, aux=dev(uu, dtype))
    assert rel_err(out.float().cpu(), (a @ w.t()) * O.dgelu(uu)) < TOL[dtype]
    # the pair the engine uses for fc1 and its backward: forward stores gelu'(u), backward multiplies by it
    dg = e(M, N, dtype=dtype)
    h2 = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N, dtype=dtype), epi=ops.EPI_BIAS_GELU_DG, bias=dev(b), aux=dg)
    assert rel_err(h2.float().cpu(), O.gelu(ref)) < TOL[dtype] and rel_err(dg.float().cpu(), O.dgelu(ref)) < TOL[dtype]
    fac = rnd(M, N, dtype=dtype, seed=7)
    out = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N, dtype=dtype), epi=ops.EPI_MUL, aux=dev(fac, dtype))
    assert rel_err(out.float().cpu(), (a @ w.t()) * fac) < TOL[dtype]
    # bias_mod (ConvTranspose2d bias repeated per tap), f32 out from ACT in
    b4 = rnd(N // 4, seed=6)
    out = ops.gemm_nt(dev(a, dtype), dev(w, dtype), e(M, N), bias=dev(b4), bias_mod=N // 4)
    assert rel_err(out.cpu(), a @ w.t() + b4.repeat(4)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_gemm_nt_tile_variants_bit_identical(ops, dtype):
    """the 256x128 / 8-wave kernel (variant 32; picked automatically for one-round problems such as M = 12544, N = 1024) and the
    tile orders accumulate in the same k order as the default kernel: every epilogue must come out bit-identical"""
    M, N, K, rps = 1024, 384, 256, 256   # complete 128-row tiles
    a, w, b = dev(rnd(M, K, dtype=dtype), dtype), dev(rnd(N, K, dtype=dtype, seed=1, scale=0.2), dtype), dev(rnd(N, seed=2))
    res, rs, uu = dev(rnd(M, N, seed=3)), dev(torch.tensor([0.0, 1.1, 0.9, 1.0])), dev(rnd(M, N, dtype=dtype, seed=5), dtype)
    outs = {}
    for v in (0, 32, 36, 4, 64):
        u = e(M, N, dtype=dtype)
        h = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_BIAS_GELU, bias=b, aux=u, variant=v)
        r = ops.gemm_nt(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=b, res=res, rowscale=rs, rows_per_sample=rps, variant=v)
        d = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_DGELU, aux=uu, variant=v)
        outs[v] = (u, h, r, d)
    for v in (32, 36, 4, 64):
        for x, y in zip(outs[0], outs[v]):
            assert torch.equal(x, y), v


# ---- the 8-wave pipelined kernel (gemm_p8.hip): variant 256 = tile height picked per problem, 512 = 224 x 256 tiles, 768 = 256 x 256 tiles
P8_SHAPES = [(256, 256, 128),      # one tile, the shortest pipeline (one K-tile pair: prologue + drain only)
             (512, 768, 256),      # 6 tiles, two pairs
             (392, 264, 384),      # ragged M and N edges (clamped DMA rows, predicated stores)
             (6272, 1024, 768),    # ViT-B token count (24.5 tile rows), patch-embed contraction
             (1568, 2304, 1024)]   # more tiles than a quick run has CUs busy: several rounds / persistent tile loop


PS = 32768       # variant bit 15: persistent tiles (the next tile's first K-tiles are issued before the epilogue of the current one)
PLAIN_TN = 1 << 19     # variant bit 19 (grouped TN): the plain phases of rounds 2-4 instead of the read-ahead phases (the next phase's fragments read under the current MFMAs)


@pytest.mark.parametrize("variant", [256, 512, 768, 514, 512 + PS, 768 + PS])
@pytest.mark.parametrize("M,N,K", P8_SHAPES)
def test_gemm_nt_p8_vs_oracle(ops, variant, M, N, K):
    dtype = torch.bfloat16
    a, w, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.1), rnd(N, seed=2)
    da, dw, out = dev(a, dtype), dev(w, dtype), e(M, N, dtype=dtype)
    assert ops.gemm_nt_tile(da, dw, out, bias=dev(b), variant=variant) == 256      # really the new kernel, not the fall-through
    ops.gemm_nt(da, dw, out, bias=dev(b), variant=variant)
    assert rel_err(out.float().cpu(), a @ w.t() + b) < TOL[dtype]


def test_gemm_nt_p8_asymmetric_identity(ops):
    """A = I against an asymmetric B: catches transposed / permuted fragments and a wrong half-tile <-> slot mapping"""
    M = N = K = 512
    w = ((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :]) % 251).float()     # exact in bf16, no two rows alike
    a = torch.eye(M)
    for variant in (512, 768):
        out = ops.gemm_nt(dev(a, torch.bfloat16), dev(w, torch.bfloat16), e(M, N), variant=variant)
        assert torch.equal(out.cpu(), w.t().contiguous()), variant


@pytest.mark.parametrize("M,N,K", P8_SHAPES)
def test_gemm_nt_p8_bit_identical_to_128_wide_kernels(ops, M, N, K):
    """same k order of accumulation as the 128-wide kernels -> every epilogue bit-identical (variant 1024 forbids the new kernel);
    repeated launches screen the counted-wait pipeline for races (a stale or half-landed LDS tile shows up as a mismatch)"""
    dtype, rps = torch.bfloat16, 196
    a, w, b = dev(rnd(M, K, dtype=dtype), dtype), dev(rnd(N, K, dtype=dtype, seed=1, scale=0.1), dtype), dev(rnd(N, seed=2))
    res, uu = dev(rnd(M, N, seed=3)), dev(rnd(M, N, dtype=dtype, seed=5), dtype)
    rs = dev(1.0 + 0.1 * rnd((M + rps - 1) // rps, seed=6))

    def run(v):
        u = e(M, N, dtype=dtype)
        h = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_BIAS_GELU, bias=b, aux=u, variant=v)
        r = ops.gemm_nt(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=b, res=res, rowscale=rs, rows_per_sample=rps, variant=v)
        d = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_DGELU, aux=uu, variant=v)
        f = ops.gemm_nt(a, w, e(M, N), bias=b, variant=v)
        dg = e(M, N, dtype=dtype)
        h2 = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_BIAS_GELU_DG, bias=b, aux=dg, variant=v)
        mu = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_MUL, aux=uu, variant=v)
        return u, h, r, d, f, dg, h2, mu
    ref = run(1024)
    assert ops.gemm_nt_tile(a, w, e(M, N, dtype=dtype), bias=b, variant=1024) == 128
    NT_, SC1, PLAIN = 1 << 20, 2 << 20, 3 << 20          # store policy of the epilogue: nt / sc1 (write-through) / plain stores (0 = picked per epilogue)
    for v in (512, 768, 512 + PS, 768 + PS, 512 + NT_, 512 + SC1, 512 + PLAIN, 512 + PS + NT_, 512 + PS + SC1, 512 + PS + PLAIN):
        for rep in range(4):
            for x, y in zip(ref, run(v)):
                assert torch.equal(x, y), (v, rep)


def test_gemm_nt_p8_vit_l_shapes_race_screen(ops):
    """the training shapes themselves (M = 64 x 196 tokens; qkv / proj / fc1 / fc2 / dqkv contractions), every launch compared
    bit for bit with the 128-wide kernels; under a full grid the DMA runs far ahead of / behind the readers"""
    T, C = 12544, 1024
    dtype = torch.bfloat16
    for (N, K) in [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C), (C, 3 * C)]:
        a, w, b = dev(rnd(T, K, dtype=dtype), dtype), dev(rnd(N, K, dtype=dtype, seed=1, scale=0.05), dtype), dev(rnd(N, seed=2))
        ref = ops.gemm_nt(a, w, e(T, N, dtype=dtype), bias=b, variant=1024)
        for v in (512, 768, 512 + PS, 768 + PS):
            out = e(T, N, dtype=dtype)
            for rep in range(6):
                out.zero_()
                ops.gemm_nt(a, w, out, bias=b, variant=v)
                assert torch.equal(out, ref), (N, K, v, rep)


# ---- the strip kernel (gemm_s8.hip, variant bit 17): 128 x 256 strips, two accumulator sets, epilogue slices under the next strip's K loop
S8 = 1 << 17
# M / N off the strip grid (clamped DMA rows, masked stores), one strip, several strips per workgroup (> 256 strips), K from the minimum (11 K-tiles) up
S8_SHAPES = [(128, 256, 704), (392, 520, 768), (1000, 264, 1024), (2048, 2304, 768), (12544, 1024, 1024), (12544, 3072, 1024), (6272, 768, 3072)]


@pytest.mark.parametrize("M,N,K", S8_SHAPES[:5])
def test_gemm_nt_s8_vs_oracle(ops, M, N, K):
    dtype = torch.bfloat16
    a, w, b = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.1), rnd(N, seed=2)
    da, dw, out = dev(a, dtype), dev(w, dtype), e(M, N, dtype=dtype)
    assert ops.gemm_nt_tile(da, dw, out, bias=dev(b), variant=S8) == 64      # really the strip kernel, not a fall-through
    ops.gemm_nt(da, dw, out, bias=dev(b), variant=S8)
    assert rel_err(out.float().cpu(), a @ w.t() + b) < TOL[dtype]


def test_gemm_nt_s8_asymmetric_identity(ops):
    """A = I against an asymmetric B: a transposed / permuted fragment, a wrong half-tile <-> slot mapping or a wrong lane after the
    permlane16 swaps of the epilogue shows up as a misplaced integer"""
    M = N = K = 1024
    w = ((torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :]) % 251).float()     # exact in bf16, no two rows alike
    a = torch.eye(M)
    for dt in (torch.float32, torch.bfloat16):
        out = ops.gemm_nt(dev(a, torch.bfloat16), dev(w, torch.bfloat16), e(M, N, dtype=dt), variant=S8)
        assert torch.equal(out.float().cpu(), w.t().contiguous()), dt


@pytest.mark.parametrize("M,N,K", S8_SHAPES)
def test_gemm_nt_s8_bit_identical_to_128_wide_kernels(ops, M, N, K):
    """same k order of accumulation and the same epilogue arithmetic as the 128-wide kernels -> every epilogue the strip kernel has must be
    bit-identical (variant 1024 forbids the pipelined kernels); repeated launches screen the counted-wait pipeline and the in-flight
    side loads for races (a stale LDS tile, a side register read before its load landed or a slice of the wrong strip is a mismatch)"""
    dtype, rps = torch.bfloat16, 196
    a, w, b = dev(rnd(M, K, dtype=dtype), dtype), dev(rnd(N, K, dtype=dtype, seed=1, scale=0.1), dtype), dev(rnd(N, seed=2))
    res, uu = dev(rnd(M, N, seed=3)), dev(rnd(M, N, dtype=dtype, seed=5), dtype)
    rs = dev(1.0 + 0.1 * rnd((M + rps - 1) // rps, seed=6))

    def run(v):
        r = ops.gemm_nt(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=b, res=res, rowscale=rs, rows_per_sample=rps, variant=v)
        r0 = ops.gemm_nt(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=b, res=res, variant=v)
        f = ops.gemm_nt(a, w, e(M, N), bias=b, variant=v)
        f0 = ops.gemm_nt(a, w, e(M, N), variant=v)
        y = ops.gemm_nt(a, w, e(M, N, dtype=dtype), bias=b, variant=v)
        y0 = ops.gemm_nt(a, w, e(M, N, dtype=dtype), variant=v)
        dg = e(M, N, dtype=dtype)
        h2 = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_BIAS_GELU_DG, bias=b, aux=dg, variant=v)
        mu = ops.gemm_nt(a, w, e(M, N, dtype=dtype), epi=ops.EPI_MUL, aux=uu, variant=v)
        rn = ops.gemm_nt(a, w, e(M, N), epi=ops.EPI_BIAS_RES, res=res, variant=v)      # residual without a bias: InternImage's data-gradient GEMMs (round 6)
        return r, r0, f, f0, y, y0, dg, h2, mu, rn
    ref = run(1024)
    assert ops.gemm_nt_tile(a, w, e(M, N), epi=ops.EPI_BIAS_RES, res=res, variant=S8) == 64
    assert ops.gemm_nt_tile(a, w, e(M, N, dtype=dtype), bias=b, variant=S8) == 64
    assert ops.gemm_nt_tile(a, w, e(M, N, dtype=dtype), epi=ops.EPI_MUL, aux=uu, variant=S8) == 64
    assert ops.gemm_nt_tile(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=b, res=res, rowscale=rs, rows_per_sample=rps, variant=S8) == 64
    names = ("res+rowscale", "res", "f32 bias", "f32", "bf16 bias", "bf16", "gelu' out", "gelu out", "mul", "res, no bias")
    for v in (S8, S8 + 2):
        for rep in range(3):
            for nm, x, y in zip(names, ref, run(v)):
                assert torch.equal(x, y), (nm, v, rep, (x.float() - y.float()).abs().max().item())


def test_gemm_nt_s8_bias_mod_and_fallbacks(ops):
    """ConvTranspose2d's repeated bias (bias_mod) through the strip kernel; what it does not take (K below 11 K-tiles, the broadcast
    residual of the patch embedding) falls through to the other families even when forced"""
    dtype = torch.bfloat16
    M, N, K, C = 512, 1024, 768, 256
    a, w, b = dev(rnd(M, K, dtype=dtype), dtype), dev(rnd(N, K, dtype=dtype, seed=1, scale=0.1), dtype), dev(rnd(C, seed=2))
    ref = ops.gemm_nt(a, w, e(M, N, dtype=dtype), bias=b, bias_mod=C, variant=1024)
    out = ops.gemm_nt(a, w, e(M, N, dtype=dtype), bias=b, bias_mod=C, variant=S8)
    assert ops.gemm_nt_tile(a, w, e(M, N, dtype=dtype), bias=b, bias_mod=C, variant=S8) == 64 and torch.equal(out, ref)
    a2 = dev(rnd(M, 512, dtype=dtype), dtype)
    w2 = dev(rnd(N, 512, dtype=dtype, seed=1), dtype)
    assert ops.gemm_nt_tile(a2, w2, e(M, N, dtype=dtype), variant=S8) != 64
    res = dev(rnd(128, N, seed=3))
    bn = dev(rnd(N, seed=4))
    assert ops.gemm_nt_tile(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=bn, res=res, res_mod=128, variant=S8) != 64
    r1 = ops.gemm_nt(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=bn, res=res, res_mod=128, variant=S8)
    r2 = ops.gemm_nt(a, w, e(M, N), epi=ops.EPI_BIAS_RES, bias=bn, res=res, res_mod=128, variant=1024)
    assert torch.equal(r1, r2)


def test_gemm_nt_s8_vit_l_shapes_race_screen(ops):
    """the training shapes with a 1024-long contraction (qkv / proj / fc1 and their data gradients) at M = 64 x 196 tokens: up to 1568 strips
    on 256 persistent workgroups, every launch compared bit for bit with the 128-wide kernels"""
    T, C = 12544, 1024
    dtype = torch.bfloat16
    for (N, K) in [(3 * C, C), (C, C), (4 * C, C), (C, 4 * C)]:
        a, w, b = dev(rnd(T, K, dtype=dtype), dtype), dev(rnd(N, K, dtype=dtype, seed=1, scale=0.05), dtype), dev(rnd(N, seed=2))
        ref = ops.gemm_nt(a, w, e(T, N, dtype=dtype), bias=b, variant=1024)
        out = e(T, N, dtype=dtype)
        for rep in range(6):
            out.zero_()
            ops.gemm_nt(a, w, out, bias=b, variant=S8)
            assert torch.equal(out, ref), (N, K, rep)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Kc,M,N,split", [(392, 384, 128, 1), (392, 256, 128, 3), (1000, 128, 768, 4), (64, 128, 128, 1), (12544, 256, 128, None)])
def test_gemm_tn(ops, dtype, Kc, M, N, split):
    a, b = rnd(Kc, M, dtype=dtype, scale=0.5), rnd(Kc, N, dtype=dtype, seed=1, scale=0.5)
    out = ops.gemm_tn(dev(a, dtype), dev(b, dtype), e(M, N), split_k=split)
    assert rel_err(out.cpu(), a.t() @ b) < 3e-4   # f32 accumulate/output in both modes; inputs are exact


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Kc,M,N,split", [(1024, 384, 256, 3), (12544, 256, 128, None), (392, 256, 128, 2), (128, 128, 384, 1),
                                          # edge tiles of the transpose-read kernel (InternImage's 192-channel level: 192 = 128 + 64,
                                          # 216-row offset heads, the 112-row padded mask head, the 96 x 32 stem convolution)
                                          (1024, 192, 192, None), (512, 216, 192, 2), (640, 112, 192, 1), (256, 96, 32, 1), (2048, 192, 768, None),
                                          (128, 8, 8, 1)])
def test_gemm_tn_bias_gradient_byproduct(ops, dtype, Kc, M, N, split):
    """colsum += dY.sum(0) out of the dW GEMM: fused in the transpose-read kernel (bf16, complete tiles), a separate pass
    otherwise; both ACCUMULATE (Linear bias gradient, VIT:50-52 backward)."""
    a, b = rnd(Kc, M, dtype=dtype, scale=0.5), rnd(Kc, N, dtype=dtype, seed=1, scale=0.5)
    cs0 = rnd(M, seed=2)
    cs = dev(cs0)
    out = ops.gemm_tn(dev(a, dtype), dev(b, dtype), e(M, N), split_k=split, colsum=cs)
    assert rel_err(out.cpu(), a.t() @ b) < 3e-4
    assert rel_err(cs.cpu(), cs0 + a.sum(0)) < 1e-4


# ---- grouped weight gradients on the 8-phase pipeline (gemm_tn_p8.hip)
def _wgrad_group(ops, shapes, seed=0):
    q, refs = ops.WgradQueue(), []
    for i, (Kc, M, N, with_cs) in enumerate(shapes):
        a = rnd(Kc, M, dtype=torch.bfloat16, scale=0.5, seed=seed + i)
        b = rnd(Kc, N, dtype=torch.bfloat16, seed=seed + 50 + i, scale=0.5)
        cs0 = rnd(M, seed=seed + 90 + i) if with_cs else None
        dw, cs = e(M, N), (dev(cs0) if with_cs else None)
        dw.fill_(float("nan"))                                   # the launch must overwrite every element
        assert q.add(dev(a, torch.bfloat16), dev(b, torch.bfloat16), dw, cs)
        refs.append((a, b, cs0, dw, cs))
    return q, refs


@pytest.mark.parametrize("shapes", [
    [(128, 256, 256, True)],                                                                  # one tile, one K-tile pair
    [(256, 512, 256, True), (384, 256, 768, False), (1024, 256, 256, True)],                   # different contractions in one launch
    [(1536, 768, 256, True), (1536, 256, 256, True), (1536, 1024, 256, False), (1536, 256, 1024, True)]])   # a block's four gradients
@pytest.mark.parametrize("variant", [0, PLAIN_TN])   # the 8-wave 8-phase kernel, read-ahead and plain phases (the 4-wave 32x32x16 form of round 3 lives in tools/ablation/)
def test_gemm_tn_grouped_vs_oracle(ops, shapes, variant):
    q, refs = _wgrad_group(ops, shapes)
    q.variant = variant
    q.flush()
    for a, b, cs0, dw, cs in refs:
        assert rel_err(dw.cpu(), a.t() @ b) < 3e-4
        if cs is not None:
            assert rel_err(cs.cpu(), cs0 + a.sum(0)) < 1e-4


@pytest.mark.parametrize("shapes", [
    [(256, 192, 192, True)],                                                                   # one edge tile (InternImage level 0: 192 channels)
    [(384, 216, 192, True), (384, 112, 192, True), (256, 768, 192, False), (256, 192, 768, True)],   # offset / mask heads, fc1 / fc2 of that level
    [(512, 96, 32, True), (256, 8, 8, True), (128, 264, 520, True)],                           # narrower than one half tile; 1-chunk problem; tiles 2 x 3 with both edges
    [(16384, 192, 192, True), (16384, 384, 216, True), (32768, 96, 32, False)],                # long contractions: cut into pieces inside the launch
    [(16384 + 128, 192, 384, True)]])                                                          # ... whose last piece is shorter
@pytest.mark.parametrize("variant", [0, PLAIN_TN])
def test_gemm_tn_grouped_edge_tiles_and_pieces(ops, shapes, variant):
    """sizes off the 256 grid (multiples of 8: the last tile row / column is clamped on the way in and masked on the way out) and few-tile
    problems with a long contraction (ops.grouped_splits: pieces of ~4096 rows, each an own workgroup, summed by one reduction launch)"""
    q, refs = _wgrad_group(ops, shapes)
    assert any(j[4] > 1 for j in q.jobs) == (shapes[0][0] >= 16384)
    q.variant = variant
    q.flush()
    for a, b, cs0, dw, cs in refs:
        assert rel_err(dw.cpu(), a.float().t() @ b.float()) < 3e-4
        if cs is not None:
            assert rel_err(cs.cpu(), cs0 + a.float().sum(0)) < 1e-4


@pytest.mark.parametrize("variant", [0, PLAIN_TN])
def test_gemm_tn_grouped_vit_l_block_repeatable(ops, variant):
    """the four weight gradients of a ViT-L block at the training size (T = 12544 tokens, 192 tiles, 98 K-tile pairs each) against
    the split-K kernels of gemm.hip, and launch-to-launch bit-identical (no atomics on dW; one atomic per bias-gradient entry):
    a stale or half-landed LDS tile in the counted-wait pipeline would show up as a difference"""
    T, C = 12544, 1024
    shapes = [(T, 3 * C, C, True), (T, C, C, True), (T, 4 * C, C, True), (T, C, 4 * C, True)]
    ins = [(dev(rnd(K, M, dtype=torch.bfloat16, scale=0.5, seed=i), torch.bfloat16), dev(rnd(K, N, dtype=torch.bfloat16, scale=0.5, seed=9 + i), torch.bfloat16))
           for i, (K, M, N, _) in enumerate(shapes)]
    first = None
    for rep in range(3):
        q = ops.WgradQueue(variant=variant)
        outs = []
        for (a, b), (K, M, N, _) in zip(ins, shapes):
            dw, cs = e(M, N), torch.zeros(M, device="cuda")
            assert q.add(a, b, dw, cs)
            outs.append((dw, cs))
        q.flush()
        torch.cuda.synchronize()
        if first is None:
            first = outs
            for (a, b), (dw, cs) in zip(ins, outs):
                ref, rcs = e(*dw.shape), torch.zeros_like(cs)
                ops.gemm_tn(a, b, ref, colsum=rcs)
                assert rel_err(dw, ref) < 2e-5 and rel_err(cs, rcs) < 2e-5
        else:
            for (x, xc), (y, yc) in zip(first, outs):
                assert torch.equal(x, y), rep
                if variant in (0, PLAIN_TN):
                    assert torch.equal(xc, yc), rep
                else:      # the 4-wave form spreads the bias gradient over the tile row: tiles_n f32 atomics per entry, order not fixed
                    assert rel_err(xc, yc) < 1e-6, rep


def test_wgrad_queue_falls_back_for_other_problems(ops):
    """f32 parity mode and sizes off the 8 / 128 grid do not queue: they run at once through mtp_gemm_tn"""
    q = ops.WgradQueue()
    a, b = rnd(392, 384, scale=0.5), rnd(392, 128, scale=0.5, seed=1)
    dw = e(384, 128)
    assert not q.add(dev(a), dev(b), dw) and not q.jobs
    assert rel_err(dw.cpu(), a.t() @ b) < 3e-4
    ab, bb = rnd(392, 256, dtype=torch.bfloat16), rnd(392, 256, dtype=torch.bfloat16, seed=1)   # contraction not a multiple of 128
    dw = e(256, 256)
    assert not q.add(dev(ab, torch.bfloat16), dev(bb, torch.bfloat16), dw)
    assert rel_err(dw.cpu(), ab.t() @ bb) < 3e-4


@pytest.mark.parametrize("dtype", DT)
def test_gemm_tn_deferred_partial_sums(ops, dtype):
    """split-K partial tiles left in the workspace (defer=...) and reduced later by one batched launch == the immediate path"""
    jobs, outs, refs = [], [], []
    for i, (Kc, M, N, split) in enumerate([(1024, 256, 128, 4), (2048, 128, 384, 2), (512, 128, 128, 1), (1536, 384, 256, 3)]):
        a, b = dev(rnd(Kc, M, dtype=dtype, scale=0.5, seed=i), dtype), dev(rnd(Kc, N, dtype=dtype, seed=10 + i, scale=0.5), dtype)
        refs.append(ops.gemm_tn(a, b, e(M, N), split_k=split))
        outs.append(ops.gemm_tn(a, b, e(M, N), split_k=split, defer=jobs))
    assert len(jobs) == 3            # the split = 1 GEMM writes its result directly
    ops.sum_partials(jobs)
    assert not jobs
    for o, r in zip(outs, refs):
        assert torch.equal(o, r)


@pytest.mark.parametrize("dtype", DT)
def test_weight_images_one_launch(ops, dtype):
    """every GEMM-side image (W, W^T in ACT; stacked f32 head weights / biases) out of one descriptor-table launch"""
    ws = [rnd(192, 128, seed=1), rnd(64, 256, seed=2), rnd(100, 36, seed=3), rnd(32, 128, seed=4), rnd(16, 128, seed=5), rnd(1, 32, seed=6)]
    dws = [dev(w) for w in ws]
    stack, bias = e(48, 128), e(32)
    imgs = [(e(192, 128, dtype=dtype), e(128, 192, dtype=dtype)), (None, e(256, 64, dtype=dtype)), (e(100, 36, dtype=dtype), e(36, 100, dtype=dtype))]
    entries = [(dws[0], imgs[0][0], imgs[0][1], False), (dws[1], None, imgs[1][1], False), (dws[2], imgs[2][0], imgs[2][1], False),
               (dws[3], stack[:32], None, True), (dws[4], stack[32:], None, True), (dws[5], bias, None, True)]
    odd, odd_w, odd_t = rnd(6, 2, seed=7), e(6, 2), e(2, 6, dtype=dtype)          # tiny odd sizes: element-wise path
    entries += [(dev(odd), odd_w, None, True), (dev(odd), None, odd_t, False)]
    # ragged tiles on the 16-byte path (round 5: 8 bf16 / 4 f32 per lane): rows and columns that are multiples of 8 but not of 64, and a 4-but-not-8 pair
    extra = [rnd(200, 72, seed=8), rnd(136, 1032, seed=9), rnd(108, 196, seed=10)]
    ximgs = [(e(*w.shape, dtype=dtype), e(w.shape[1], w.shape[0], dtype=dtype)) for w in extra]
    entries += [(dev(w), a, at, False) for w, (a, at) in zip(extra, ximgs)]
    tab = ops.WeightImages(entries, dtype)
    tab.refresh()
    assert torch.equal(odd_w.cpu(), odd) and torch.equal(odd_t.cpu(), odd.t().contiguous().to(dtype))
    for w, (a, at) in zip(extra, ximgs):
        assert torch.equal(a.cpu(), w.to(dtype)) and torch.equal(at.cpu(), w.t().contiguous().to(dtype))
    for w, (a, at) in zip(ws[:3], imgs):
        if a is not None:
            assert torch.equal(a.cpu(), w.to(dtype))
        assert torch.equal(at.cpu(), w.t().contiguous().to(dtype))
    assert torch.equal(stack.cpu(), torch.cat([ws[3], ws[4]])) and torch.equal(bias.cpu(), ws[5].view(-1))
    dws[1].mul_(2.0)   # parameters change in place (optimizer step): a refresh re-reads the same table
    tab.refresh()
    assert torch.equal(imgs[1][1].cpu(), (2 * ws[1]).t().contiguous().to(dtype))


# ------------------------------------------------------------------------------------------------ LayerNorm & reductions
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,C", [(392, 128), (50, 768), (7, 1024), (33, 96), (130, 384), (21, 260), (9, 1280)])      # (every per-width instantiation: 1, 3, 4, 1, 2, 2, 6 groups per lane)
def test_layernorm_fwd_bwd(ops, dtype, rows, C):
    x, g, b = rnd(rows, C, scale=2.0) + 0.5, 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    mean, rstd = e(rows), e(rows)
    y = ops.layernorm_fwd(dev(x), dev(g), dev(b), e(rows, C, dtype=dtype), mean, rstd)
    yr, mr, rr = O.layernorm_fwd(x, g, b)
    assert rel_err(y.float().cpu(), yr) < TOL[dtype] and rel_err(mean.cpu(), mr) < 1e-5 and rel_err(rstd.cpu(), rr) < 1e-5
    dy, dres, extra = rnd(rows, C, dtype=dtype, seed=3), rnd(rows, C, seed=4), rnd(rows, C, seed=5)
    rps = rows // 2 if rows % 2 == 0 else rows
    cs = torch.tensor([0.5, 2.0])[: rows // rps]
    dx, dxc, dg, db = e(rows, C), e(rows, C, dtype=dtype), e(C), e(C)
    ops.layernorm_bwd(dev(dy, dtype), dev(x), mean, rstd, dev(g), dx, dg, db, dres=dev(dres), extra=dev(extra), dx_copy=dxc,
                      copy_scale=dev(cs), rows_per_sample=rps)
    dxr, dgr, dbr = O.layernorm_bwd(dy, x, mr, rr, g)
    tot = dxr + dres + extra
    assert rel_err(dx.cpu(), tot) < 2e-4 and rel_err(dxc.float().cpu(), tot * cs.repeat_interleave(rps)[:, None]) < TOL[dtype]
    assert rel_err(dg.cpu(), dgr) < 2e-4 and rel_err(db.cpu(), dbr) < 2e-4
    # accumulating entry point (training engine): dgamma / dbeta += ..., no partial buffer
    dg2, db2, dx2 = dev(torch.ones(C)), dev(torch.full((C,), -2.0)), e(rows, C)
    ops.layernorm_bwd(dev(dy, dtype), dev(x), mean, rstd, dev(g), dx2, dg2, db2, dres=dev(dres), extra=dev(extra), accumulate=True)
    assert rel_err(dg2.cpu(), 1 + dgr) < 2e-4 and rel_err(db2.cpu(), dbr - 2) < 2e-4 and rel_err(dx2.cpu(), tot) < 2e-4


def test_layernorm_bwd_deferred_parameter_gradients(ops):
    """several LayerNorm backwards keep their partial rows; ONE reduction launch per shape finishes all the dgamma / dbeta pairs"""
    items, want, bufs = [], [], []
    for j, (rows, C) in enumerate([(392, 128), (392, 128), (50, 768), (392, 128)]):
        x, g = rnd(rows, C, scale=2.0, seed=10 * j) + 0.5, 1 + 0.1 * rnd(C, seed=10 * j + 1)
        _, mr, rr = O.layernorm_fwd(x, g, torch.zeros(C))
        dy = rnd(rows, C, seed=10 * j + 3)
        gb = dev(torch.full((2 * C,), float(j)))          # [dgamma | dbeta] adjacent, as in the flat gradient buffer
        ops.layernorm_bwd(dev(dy), dev(x), dev(mr), dev(rr), dev(g), e(rows, C), gb[:C], gb[C:], accumulate=True, defer=items)
        _, dgr, dbr = O.layernorm_bwd(dy, x, mr, rr, g)
        want.append(torch.cat([dgr, dbr]) + j)
        bufs.append(gb)
    assert len(items) == 4 and all(torch.equal(b.cpu(), torch.full_like(b.cpu(), float(j))) for j, b in enumerate(bufs))
    ops.reduce_rows_deferred(items)
    assert not items
    for b, w in zip(bufs, want):
        assert rel_err(b.cpu(), w) < 2e-4
    # gradients that are NOT adjacent fall back to the immediate reduction
    rows, C = 50, 768
    x, g, dy = rnd(rows, C, seed=90) + 0.5, 1 + 0.1 * rnd(C, seed=91), rnd(rows, C, seed=92)
    _, mr, rr = O.layernorm_fwd(x, g, torch.zeros(C))
    dg, db = e(C), e(C)
    dg.zero_(), db.zero_()
    ops.layernorm_bwd(dev(dy), dev(x), dev(mr), dev(rr), dev(g), e(rows, C), dg, db, accumulate=True, defer=items)
    _, dgr, dbr = O.layernorm_bwd(dy, x, mr, rr, g)
    assert not items and rel_err(dg.cpu(), dgr) < 2e-4 and rel_err(db.cpu(), dbr) < 2e-4


@pytest.mark.parametrize("dtype", DT)
def test_layernorm_gelu_act_io(ops, dtype):
    rows, C = 784, 128
    x, g, b = rnd(rows, C, dtype=dtype), 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
    mean, rstd = e(rows), e(rows)
    y = ops.layernorm_fwd(dev(x, dtype), dev(g), dev(b), e(rows, C, dtype=dtype), mean, rstd, gelu=True)
    z, mr, rr = O.layernorm_fwd(x, g, b)
    assert rel_err(y.float().cpu(), O.gelu(z)) < TOL[dtype]
    dy = rnd(rows, C, dtype=dtype, seed=3)
    dx, dg, db = e(rows, C, dtype=dtype), e(C), e(C)
    ops.layernorm_bwd(dev(dy, dtype), dev(x, dtype), mean, rstd, dev(g), dx, dg, db, beta=dev(b), gelu=True)
    dxr, dgr, dbr = O.layernorm_bwd(dy * O.dgelu(z), x, mr, rr, g)
    assert rel_err(dx.float().cpu(), dxr) < TOL[dtype] and rel_err(dg.cpu(), dgr) < 2e-4 and rel_err(db.cpu(), dbr) < 2e-4


@pytest.mark.parametrize("dtype", DT)
def test_colsum_reduce_rows_axpy_cast(ops, dtype):
    dy = rnd(3000, 384, dtype=dtype)
    assert rel_err(ops.colsum(dev(dy, dtype), e(384)).cpu(), dy.sum(0)) < 2e-4
    part = rnd(5000, 260)
    assert rel_err(ops.reduce_rows(dev(part), e(260)).cpu(), part.sum(0)) < 2e-4
    # column slice of a wider partial buffer, accumulating into a non-zero gradient (partials -> two parameters, no staging copy)
    dpart, g0 = dev(part), rnd(160, seed=3)
    acc = ops.reduce_rows(dpart[:, 100:], dev(g0), accumulate=True)
    assert rel_err(acc.cpu(), g0 + part[:, 100:].sum(0)) < 2e-4
    srcs = [rnd(n, seed=20 + i) for i, n in enumerate((32 * 128, 5, 16 * 128, 1, 700, 64))]
    dsts = [e(n) for n in (32 * 128, 5, 16 * 128, 1, 700, 64)]
    ops.copy_segments([dev(s) for s in srcs], dsts)
    assert all(torch.equal(d.cpu(), s) for s, d in zip(srcs, dsts))
    y, x = rnd(1001), rnd(1001, seed=9)
    assert rel_err(ops.axpy(dev(y), dev(x), 0.5).cpu(), y + 0.5 * x) < 1e-6
    src = rnd(1027)
    assert torch.equal(ops.cast(dev(src), e(1027, dtype=torch.bfloat16)).cpu(), src.to(torch.bfloat16))
    s = rnd(392, 128)
    out = ops.scale_rows_cast(dev(s), e(392, 128, dtype=dtype), dev(torch.tensor([2.0, 0.0])), 196)
    assert rel_err(out.float().cpu(), s * torch.tensor([2.0, 0.0]).repeat_interleave(196)[:, None]) < TOL[dtype]


# ------------------------------------------------------------------------------------------------ layout ops (bit-exact data movement)
@pytest.mark.parametrize("dtype", DT)
def test_patchify_roundtrip(ops, dtype):
    img = rnd(2, 3, 64, 48, dtype=dtype)
    cols = ops.patchify(dev(img), e(2 * 4 * 3, 768, dtype=dtype))
    ref, _ = O.patchify(img)
    assert torch.equal(cols.float().cpu(), ref)
    back = ops.unpatchify(cols, e(2, 3, 64, 48))
    assert torch.equal(back.cpu(), img)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("H,W,flip", [(224, 224, True), (200, 150, True), (64, 96, False), (33, 31, True)])
def test_preprocess_patchify_uint8(ops, dtype, H, W, flip):
    """uint8 HWC batch -> normalised patch rows in one kernel == oracle.preprocess (MTP_DataPreprocessor's image path: flip,
    (x - mean) / std, pad bottom/right to a multiple of 32) followed by oracle.patchify; exact in f32, exact after rounding in bf16"""
    mean, std = (123.675, 116.28, 103.53), (58.395, 57.12, 57.375)
    g = torch.Generator().manual_seed(H * 1000 + W)
    img = torch.randint(0, 256, (3, H, W, 3), generator=g, dtype=torch.uint8)
    ref_img = O.preprocess(img, mean, std, bgr_to_rgb=flip, pad_size_divisor=32, pad_value=0.0)
    ref, (Hp, Wp) = O.patchify(ref_img)
    assert (Hp, Wp) == ops.padded_grid(H, W, 16, 32)
    cols = ops.preprocess_patchify(img.cuda(), e(3 * Hp * Wp, 768, dtype=dtype), 16, mean, std, bgr_to_rgb=flip, pad_divisor=32, pad_value=0.0)
    assert torch.equal(cols.cpu(), ref.to(dtype))
    cols = ops.preprocess_patchify(img.cuda(), e(3 * Hp * Wp, 768, dtype=dtype), 16, mean, std, bgr_to_rgb=flip, pad_divisor=32, pad_value=-1.5)
    ref2, _ = O.patchify(O.preprocess(img, mean, std, bgr_to_rgb=flip, pad_size_divisor=32, pad_value=-1.5))
    assert torch.equal(cols.cpu(), ref2.to(dtype))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("L", [0, 1, 2])
def test_tokens_nchw_roundtrip(ops, dtype, L):
    B, Hp, Wp, C = 2, 14, 14, 128
    x = rnd(B * Hp * Wp * 4 ** L, C, dtype=dtype)
    f = ops.tokens_to_nchw(dev(x, dtype), e(B, C, Hp << L, Wp << L, dtype=dtype), B, Hp, Wp, L)
    assert torch.equal(f.float().cpu(), O.tokens_to_nchw(x, B, Hp, Wp, L))
    back = ops.nchw_to_tokens(f, e(x.shape[0], C, dtype=dtype), B, Hp, Wp, L)
    assert torch.equal(back.float().cpu(), x)
    x2 = rnd(3 * 5 * 4 ** L, 68)   # ragged tile edges, f32 -> f32
    f2 = ops.tokens_to_nchw(dev(x2), e(1, 68, 3 << L, 5 << L), 1, 3, 5, L)
    assert torch.equal(f2.cpu(), O.tokens_to_nchw(x2, 1, 3, 5, L))
    x3 = rnd(2 * 4 * 6 * 4 ** L, 72, dtype=dtype)   # positions and channels multiples of 8 but not of 64: ragged tiles of the 16-byte bf16 kernel (round 5)
    f3 = ops.tokens_to_nchw(dev(x3, dtype), e(2, 72, 4 << L, 6 << L, dtype=dtype), 2, 4, 6, L)
    assert torch.equal(f3.float().cpu(), O.tokens_to_nchw(x3, 2, 4, 6, L))
    assert torch.equal(ops.nchw_to_tokens(f3, e(x3.shape[0], 72, dtype=dtype), 2, 4, 6, L).float().cpu(), x3)
    if L == 0:      # the 7 x 7 tap behind the max-pool: 49 positions per plane (not a multiple of 4: the element-wise edge path), whole and ragged channel tiles, mixed dtypes
        for Cc in (128, 72):
            x4 = rnd(3 * 49, Cc, dtype=dtype)
            f4 = ops.tokens_to_nchw(dev(x4, dtype), e(3, Cc, 7, 7, dtype=dtype), 3, 7, 7, 0)
            assert torch.equal(f4.float().cpu(), O.tokens_to_nchw(x4, 3, 7, 7, 0))
            assert torch.equal(ops.nchw_to_tokens(f4, e(3 * 49, Cc, dtype=dtype), 3, 7, 7, 0).float().cpu(), x4)
            f5 = ops.tokens_to_nchw(dev(x4.float()), e(3, Cc, 7, 7, dtype=dtype), 3, 7, 7, 0)       # f32 tokens -> ACT map (what the FPN tail does)
            assert torch.equal(f5.float().cpu(), O.tokens_to_nchw(x4.float(), 3, 7, 7, 0).to(dtype).float())
            b5 = ops.nchw_to_tokens(f5, e(3 * 49, Cc), 3, 7, 7, 0)                                   # ACT map -> f32 tokens
            assert torch.equal(b5.cpu(), f5.float().cpu().permute(0, 2, 3, 1).reshape(3 * 49, Cc))


@pytest.mark.parametrize("dtype", DT)
def test_weight_packing(ops, dtype):
    w = rnd(384, 128)
    assert torch.equal(ops.transpose_cast(dev(w), e(128, 384, dtype=dtype)).cpu(), w.t().contiguous().to(dtype))
    # (64, 96), (136, 72), (256, 128): the tiled kernels (channels multiples of 8; ragged and whole 64 x 64 tiles); (10, 12): the element-wise ones
    for Cin, Cout in ((64, 96), (136, 72), (256, 128), (10, 12)):
        cw = rnd(Cin, Cout, 2, 2, seed=3)
        wg, wgT = e(4 * Cout, Cin, dtype=dtype), e(Cin, 4 * Cout, dtype=dtype)
        ops.convt_pack(dev(cw), wg, wgT)
        ref = O.convT_gemm_weight(cw)
        assert torch.equal(wg.cpu(), ref.to(dtype)) and torch.equal(wgT.cpu(), ref.t().contiguous().to(dtype)), (Cin, Cout)
        only_t = e(Cin, 4 * Cout, dtype=dtype)
        ops.convt_pack(dev(cw), None, only_t)
        assert torch.equal(only_t.cpu(), ref.t().contiguous().to(dtype))
        dwg = rnd(4 * Cout, Cin, seed=4)
        dw = ops.convt_unpack_grad(dev(dwg), e(Cin, Cout, 2, 2))
        wr = cw.clone().requires_grad_(True)
        (O.convT_gemm_weight(wr) * dwg).sum().backward()
        assert torch.equal(dw.cpu(), wr.grad), (Cin, Cout)


@pytest.mark.parametrize("dtype", DT)
def test_maxpool_tokens(ops, dtype):
    B, Hp, Wp, C = 2, 14, 14, 128
    x = rnd(B * Hp * Wp, C)
    x[5] = x[4]   # force ties: gradient goes to the first maximum
    y = ops.maxpool2_tokens_fwd(dev(x), e(B * 49, C, dtype=dtype), B, Hp, Wp)
    xr = x.clone().requires_grad_(True)
    ref = torch.nn.functional.max_pool2d(O.tokens_to_nchw(xr, B, Hp, Wp, 0), 2, 2)
    assert torch.equal(y.float().cpu(), O.nchw_to_tokens(ref, B, 7, 7, 0).detach().to(dtype).float())
    dy = rnd(B * 49, C, dtype=dtype, seed=2)
    ref.backward(O.tokens_to_nchw(dy, B, 7, 7, 0))
    dx = ops.maxpool2_tokens_bwd(dev(x), dev(dy, dtype), e(B * Hp * Wp, C), B, Hp, Wp)
    assert torch.equal(dx.cpu(), xr.grad)


# ------------------------------------------------------------------------------------------------ attention
def _attn_inputs(T, C, dtype, seed=0):
    return rnd(T, 3 * C, dtype=dtype, seed=seed)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Hp,Wp", [(14, 14), (9, 12), (28, 28), (17, 19), (32, 32)])   # 28x28 = 448^2 pretraining inputs, 32x32 = 512^2:
def test_full_attention_fwd_bwd(ops, dtype, Hp, Wp):                                    # beyond 256 tokens -> multi-workgroup kernels
    B, heads, hd = 2, 2, 64
    C, N = heads * hd, Hp * Wp
    T = B * N
    qkv = _attn_inputs(T, C, dtype)
    rh, rw = 0.3 * rnd(2 * Hp - 1, hd, seed=1), 0.3 * rnd(2 * Wp - 1, hd, seed=2)
    o, lse = e(T, C, dtype=dtype), e(B * heads * N)
    ops.full_attn_fwd(dev(qkv, dtype), o, lse, dev(rh), dev(rw), B, Hp, Wp, heads, hd ** -0.5)
    q = qkv.clone().requires_grad_(True)
    rhr, rwr = rh.clone().requires_grad_(True), rw.clone().requires_grad_(True)
    oref, lref = O.full_attn_fwd(q, B, Hp, Wp, heads, rhr, rwr)
    assert rel_err(o.float().cpu(), oref) < TOL[dtype] and rel_err(lse.cpu().reshape(lref.shape), lref) < (1e-4 if dtype == torch.float32 else 5e-3)
    do = rnd(T, C, dtype=dtype, seed=3)
    gq, gh, gw = torch.autograd.grad(oref, (q, rhr, rwr), do)
    dqkv, drh, drw = e(T, 3 * C, dtype=dtype), e(*rh.shape), e(*rw.shape)
    ops.full_attn_bwd(dev(qkv, dtype), o, dev(do, dtype), lse, dqkv, dev(rh), dev(rw), drh, drw, B, Hp, Wp, heads, hd ** -0.5)
    assert rel_err(dqkv.float().cpu(), gq) < TOL[dtype]
    assert rel_err(drh.cpu(), gh) < 10 * TOL[dtype] and rel_err(drw.cpu(), gw) < 10 * TOL[dtype]


@pytest.mark.parametrize("Hp,Wp,B", [(16, 16, 2), (15, 13, 2), (7, 16, 3), (16, 5, 2), (3, 3, 2), (1, 1, 2), (13, 14, 1), (2, 9, 4)])
def test_full_attention_row_aligned_kernels_small_grids(ops, Hp, Wp, B):
    """token grids of at most 16 x 16 run on the row-aligned kernels (attn_full_v3.hip: one MFMA tile = one image row, the
    relative-position logits as one-hot k-slots of the same contraction, bias rows in two bf16 halves): full width / height (no padding
    column, no -30000 slots), odd Hp (a key-tile pair with a missing tile), narrow and single-token grids -- forward, dq / dk / dv and both
    table gradients vs the oracle's autograd"""
    dtype = torch.bfloat16
    heads, hd = 3, 64
    C, N = heads * hd, Hp * Wp
    T = B * N
    qkv = _attn_inputs(T, C, dtype, seed=7)
    rh, rw = 0.3 * rnd(2 * Hp - 1, hd, seed=1), 0.3 * rnd(2 * Wp - 1, hd, seed=2)
    o, lse = e(T, C, dtype=dtype), e(B * heads * N)
    ops.full_attn_fwd(dev(qkv, dtype), o, lse, dev(rh), dev(rw), B, Hp, Wp, heads, hd ** -0.5)
    q = qkv.clone().requires_grad_(True)
    rhr, rwr = rh.clone().requires_grad_(True), rw.clone().requires_grad_(True)
    oref, lref = O.full_attn_fwd(q, B, Hp, Wp, heads, rhr, rwr)
    assert rel_err(o.float().cpu(), oref) < TOL[dtype] and rel_err(lse.cpu().reshape(lref.shape), lref) < 5e-3
    do = rnd(T, C, dtype=dtype, seed=3)
    gq, gh, gw = torch.autograd.grad(oref, (q, rhr, rwr), do)
    dqkv, drh, drw = e(T, 3 * C, dtype=dtype), e(*rh.shape), e(*rw.shape)
    ops.full_attn_bwd(dev(qkv, dtype), o, dev(do, dtype), lse, dqkv, dev(rh), dev(rw), drh, drw, B, Hp, Wp, heads, hd ** -0.5)
    for name, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        if N == 1 and name != "dv":      # one key: softmax = 1, the exact dq / dk are zero (the kernel's are bf16 rounding noise of dP - delta)
            assert float(dqkv[:, sl].float().abs().max()) < 1e-6, name
            continue
        assert rel_err(dqkv[:, sl].float().cpu(), gq[:, sl]) < TOL[dtype], name
    if N > 1:
        assert rel_err(drh.cpu(), gh) < 10 * TOL[dtype] and rel_err(drw.cpu(), gw) < 10 * TOL[dtype]


@pytest.mark.parametrize("Hp,Wp,B", [(64, 64, 1), (28, 28, 2), (40, 25, 2), (33, 40, 1), (20, 50, 2), (40, 8, 2), (26, 10, 2)])
def test_full_attention_flash_large_grids(ops, Hp, Wp, B):
    """bf16 flash forward + flash MFMA backward (attn_full_flash_bwd.hip) beyond 256 tokens: 64 x 64 = the 1024^2 detection
    fine-tunes (4096 tokens, 127-row tables), non-square grids whose key blocks start mid-row, Wp = 10 (the narrowest grid the
    flash backward takes) and Wp = 8 (falls through to the three-pass f32-math kernels) -- vs the oracle's autograd"""
    dtype = torch.bfloat16
    heads, hd = 2, 64
    C, N = heads * hd, Hp * Wp
    T = B * N
    qkv = _attn_inputs(T, C, dtype, seed=5)
    rh, rw = 0.3 * rnd(2 * Hp - 1, hd, seed=1), 0.3 * rnd(2 * Wp - 1, hd, seed=2)
    o, lse = e(T, C, dtype=dtype), e(B * heads * N)
    ops.full_attn_fwd(dev(qkv, dtype), o, lse, dev(rh), dev(rw), B, Hp, Wp, heads, hd ** -0.5)
    q = qkv.clone().requires_grad_(True)
    rhr, rwr = rh.clone().requires_grad_(True), rw.clone().requires_grad_(True)
    oref, lref = O.full_attn_fwd(q, B, Hp, Wp, heads, rhr, rwr)
    assert rel_err(o.float().cpu(), oref) < TOL[dtype] and rel_err(lse.cpu().reshape(lref.shape), lref) < 5e-3
    do = rnd(T, C, dtype=dtype, seed=3)
    gq, gh, gw = torch.autograd.grad(oref, (q, rhr, rwr), do)
    dqkv, drh, drw = e(T, 3 * C, dtype=dtype), e(*rh.shape), e(*rw.shape)
    ops.full_attn_bwd(dev(qkv, dtype), o, dev(do, dtype), lse, dqkv, dev(rh), dev(rw), drh, drw, B, Hp, Wp, heads, hd ** -0.5)
    for name, sl in (("dq", slice(0, C)), ("dk", slice(C, 2 * C)), ("dv", slice(2 * C, 3 * C))):
        assert rel_err(dqkv[:, sl].float().cpu(), gq[:, sl]) < TOL[dtype], name
    assert rel_err(drh.cpu(), gh) < 10 * TOL[dtype] and rel_err(drw.cpu(), gw) < 10 * TOL[dtype]
    # measured errors of the flash kernels on inputs that are EXACT in bf16, against the oracle in f32 on the same numbers (VERDICT r04 #4c asked for an
    # fp32-mode run of these kernels at 1e-3: they have no f32 form -- P and dS are bf16 MFMA operands, 2^-9 relative per element -- so what the kernels
    # lose is recorded here per grid as relative L2, and bounded: forward 3e-3, gradients 6e-3)
    from conftest import record_parity
    l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    grp = "flash_bf16_exact_inputs_%dx%d" % (Hp, Wp)
    vals = dict(o=l2(o.float().cpu(), oref.detach()), dq=l2(dqkv[:, :C].float().cpu(), gq[:, :C]), dk=l2(dqkv[:, C:2 * C].float().cpu(), gq[:, C:2 * C]),
                dv=l2(dqkv[:, 2 * C:].float().cpu(), gq[:, 2 * C:]), drel_h=l2(drh.cpu(), gh), drel_w=l2(drw.cpu(), gw))
    for k, v in vals.items():
        record_parity(grp, k, v)
    assert vals["o"] < 3e-3 and max(vals["dq"], vals["dk"], vals["dv"]) < 6e-3, vals


@pytest.mark.parametrize("C,heads", [(1024, 16), (768, 12), (1280, 3)])
def test_rvsa_sampling_fwd_at_model_widths(ops, C, heads):
    """the fused sampling-head forward at ViT-L / ViT-B widths: 5 * heads outputs split over ceil(N / 16) workgroups per window, a pass's weight loads in flight
    together (round 6), C not a multiple of 1024 (the masked tail of the 4-step k loop)"""
    B, Hp, Wp = 3, 14, 14
    x = rnd(B * Hp * Wp, C, dtype=torch.bfloat16)
    w, b = rnd(5 * heads, C, seed=1, scale=0.1), rnd(5 * heads, seed=2)
    R = B * 4
    avg, pooled, y = e(R, C), e(R, C), e(R, 5 * heads)
    ops.rvsa_sampling_fwd(dev(x, torch.bfloat16), dev(w), dev(b), avg, pooled, y, B, Hp, Wp)
    ar, pr = O.rvsa_pool_fwd(x, B, Hp, Wp)
    assert rel_err(avg.cpu(), ar) < 1e-5 and rel_err(pooled.cpu(), pr) < 1e-5 and rel_err(y.cpu(), pr @ w.t() + b) < 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Hp,Wp", [(14, 14), (16, 12)])
def test_rvsa_pool_and_small_linear(ops, dtype, Hp, Wp):
    B, C, heads = 2, 128, 2
    T = B * Hp * Wp
    x = rnd(T, C, dtype=dtype)
    nh, nw = ops.rvsa_windows(Hp, Wp)
    R = B * nh * nw
    avg, pooled = e(R, C), e(R, C)
    ops.rvsa_pool_fwd(dev(x, dtype), avg, pooled, B, Hp, Wp)
    ar, pr = O.rvsa_pool_fwd(x, B, Hp, Wp)
    assert rel_err(avg.cpu(), ar) < 1e-5 and rel_err(pooled.cpu(), pr) < 1e-5
    w, b = rnd(5 * heads, C, seed=1, scale=0.1), rnd(5 * heads, seed=2)
    y = ops.small_linear_fwd(pooled, dev(w), dev(b), e(R, 5 * heads))
    assert rel_err(y.cpu(), pr @ w.t() + b) < 1e-5
    dy = rnd(R, 5 * heads, seed=3)
    dx, dw, db = e(R, C), e(5 * heads, C), e(5 * heads)
    ops.small_linear_bwd(pooled, dev(w), dev(dy), dx, dw, db)
    assert rel_err(dx.cpu(), dy @ w) < 1e-5 and rel_err(dw.cpu(), dy.t() @ pr) < 1e-5 and rel_err(db.cpu(), dy.sum(0)) < 1e-5
    base = rnd(T, C, dtype=dtype, seed=4)
    acc = dev(base, dtype)
    ops.rvsa_pool_bwd(dx, avg, acc, B, Hp, Wp, accumulate=True)
    assert rel_err(acc.float().cpu(), base + O.rvsa_pool_bwd(dx.cpu(), ar, B, Hp, Wp)) < TOL[dtype]
    # the fused forms (one launch each way) give the same numbers
    avg2, pooled2, y2 = e(R, C), e(R, C), e(R, 5 * heads)
    ops.rvsa_sampling_fwd(dev(x, dtype), dev(w), dev(b), avg2, pooled2, y2, B, Hp, Wp)
    assert rel_err(avg2.cpu(), ar) < 1e-5 and rel_err(pooled2.cpu(), pr) < 1e-5 and rel_err(y2.cpu(), pr @ w.t() + b) < 1e-5
    acc2 = dev(base, dtype)
    ops.rvsa_sampling_bwd(dev(dy), dev(w), avg, acc2, B, Hp, Wp)
    assert rel_err(acc2.float().cpu(), base + O.rvsa_pool_bwd(dy @ w, ar, B, Hp, Wp)) < TOL[dtype]
    dw2, db2 = e(5 * heads, C), e(5 * heads)
    ops.small_linear_bwd(pooled, dev(w), dev(dy), None, dw2, db2)
    assert rel_err(dw2.cpu(), dy.t() @ pr) < 1e-5 and rel_err(db2.cpu(), dy.sum(0)) < 1e-5
    # ... and the stacked heads' gradients accumulated straight into three separate parameters (2H | 2H | H rows)
    rows = [2 * heads, 2 * heads, heads]
    base_w = [rnd(r, C, seed=20 + i) for i, r in enumerate(rows)]
    base_b = [rnd(r, seed=30 + i) for i, r in enumerate(rows)]
    gw, gb = [dev(t) for t in base_w], [dev(t) for t in base_b]
    ops.small_linear_dw_segments(pooled, dev(dy), gw, gb)
    full_w, full_b, r0 = dy.t() @ pr, dy.sum(0), 0
    for i, r in enumerate(rows):
        assert rel_err(gw[i].cpu(), base_w[i] + full_w[r0:r0 + r]) < 1e-5 and rel_err(gb[i].cpu(), base_b[i] + full_b[r0:r0 + r]) < 1e-5
        r0 += r


def test_batched_small_linear_wgrad_and_deferred_reductions_equal_their_unbatched_forms(ops):
    """the batched kernels of round 4 against the one-problem launches they replace -- mtp_small_linear_dw_segments_batched (the stacked
    sampling heads of a burst of blocks; more jobs than SL_BATCH so the flush is cut into several launches), mtp_reduce_rows_batched_f32 and the
    column-major mtp_reduce_rows_t_batched_f32 (more buffers than REDUCE_BATCH_MAX), each with and without accumulation"""
    R, K, heads = 256, 256, 4
    rows = [2 * heads, 2 * heads, heads]
    N = sum(rows)
    njobs = ops.SL_BATCH + 3
    refs = []
    for j in range(njobs):
        x, dy = dev(rnd(R, K, seed=100 + j)), dev(rnd(R, N, seed=200 + j))
        rw = [dev(rnd(r, K, seed=300 + 7 * j + i)) for i, r in enumerate(rows)]
        rb = [dev(rnd(r, seed=400 + 7 * j + i)) for i, r in enumerate(rows)]
        ops.small_linear_dw_segments(x, dy, rw, rb)            # the unbatched form, accumulating into the same starting values
        refs.append((rw, rb))
    kept = []
    jobs = []
    for j in range(njobs):
        x, dy = dev(rnd(R, K, seed=100 + j)), dev(rnd(R, N, seed=200 + j))
        gw = [dev(rnd(r, K, seed=300 + 7 * j + i)) for i, r in enumerate(rows)]
        gb = [dev(rnd(r, seed=400 + 7 * j + i)) for i, r in enumerate(rows)]
        jobs.append((x, dy, gw, gb))
        kept.append((gw, gb))
    ops.small_linear_dw_segments_flush(jobs)
    for (gw, gb), (rw, rb) in zip(kept, refs):
        for a, b in zip(gw + gb, rw + rb):
            assert rel_err(a, b) < 1e-6
    # deferred row reductions: row-major partial buffers -> contiguous outputs
    nbuf, prow, cols = ops.REDUCE_BATCH_MAX + 5, 37, 768
    for acc in (False, True):
        items, outs, refs = [], [], []
        for i in range(nbuf):
            part = dev(rnd(prow, cols, seed=500 + i))
            base = rnd(cols, seed=600 + i)
            o, r = dev(base), dev(base)
            ops.reduce_rows(part, r, accumulate=acc)
            items.append((part, o, acc))
            outs.append(o)
            refs.append(r)
        ops.reduce_rows_deferred(items)
        assert not items
        for o, r in zip(outs, refs):
            assert rel_err(o, r) < 1e-6           # (the batched kernel walks the rows in a different order: f32 rounding only)
    # ... and the transposed form: column a * C + b of the partial rows -> out[b * R + a] (the attention kernels' per-workgroup table partials)
    Rt, Ct = 26, 64
    for acc in (False, True):
        items, outs, refs = [], [], []
        for i in range(nbuf):
            part = rnd(prow, Rt * Ct, seed=700 + i)
            base = rnd(Ct * Rt, seed=800 + i)
            o = dev(base)
            items.append((dev(part), o, acc, (Rt, Ct)))
            outs.append(o)
            ref = part.sum(0).view(Rt, Ct).t().contiguous().view(-1)
            refs.append(ref + base if acc else ref)
        ops.reduce_rows_deferred(items)
        for o, r in zip(outs, refs):
            assert rel_err(o.cpu(), r) < 1e-6


@pytest.mark.parametrize("R,N,K", [(1024, 80, 1024), (37, 10, 128), (130, 5, 1100 * 4), (6, 83, 768), (67, 12, 1536), (300, 80, 256)])
def test_small_linear_shapes(ops, R, N, K):
    """ragged rows / outputs, the ViT-L head shape, and K above the register path (generic forward kernel)"""
    x, w, b, dy = rnd(R, K), rnd(N, K, seed=1, scale=0.1), rnd(N, seed=2), rnd(R, N, seed=3)
    y = ops.small_linear_fwd(dev(x), dev(w), dev(b), e(R, N))
    assert rel_err(y.cpu(), x @ w.t() + b) < 1e-5
    dx, dw, db = e(R, K), e(N, K), e(N)
    ops.small_linear_bwd(dev(x), dev(w), dev(dy), dx, dw, db)
    assert rel_err(dx.cpu(), dy @ w) < 1e-5 and rel_err(dw.cpu(), dy.t() @ x) < 1e-5 and rel_err(db.cpu(), dy.sum(0)) < 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Hp,Wp", [(14, 14), (16, 12), (32, 32)])
def test_layernorm_bwd_with_window_addend(ops, dtype, Hp, Wp):
    """mtp_rvsa_sampling_bwd_win + mtp_layernorm_bwd_win (round 4: norm1's backward adds the sampling heads' input gradient per 7 x 7 window
    while it reads the row) against torch: dy_eff = dy + broadcast((dsamp . w) * leaky'(avg) / 49) through LayerNorm's backward -- and against
    the two-pass form (mtp_rvsa_sampling_bwd into dy, then mtp_layernorm_bwd).  32 x 32 is the padded case (35 x 35, 25 windows)."""
    B, C, N5 = 3, 256, 20
    T = B * Hp * Wp
    nh, nw = ops.rvsa_windows(Hp, Wp)
    R = B * nh * nw
    pt, pl = ((7 - Hp % 7) % 7) // 2, ((7 - Wp % 7) % 7) // 2
    x, dy, gamma, beta = rnd(T, C), rnd(T, C, dtype=dtype, seed=1), 1.0 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    dsamp, w, avg, dres = rnd(R, N5, seed=4), 0.1 * rnd(N5, C, seed=5), rnd(R, C, seed=6), rnd(T, C, seed=7)
    # torch: the per-window factor, spread over the tokens
    g_ref = (dsamp @ w) * torch.where(avg > 0, torch.tensor(1.0), torch.tensor(0.01)) / 49.0
    ys, xs = torch.arange(Hp), torch.arange(Wp)
    win = (((ys + pt) // 7)[:, None] * nw + ((xs + pl) // 7)[None, :]).reshape(-1)                    # (Hp * Wp,) window of each token
    win_all = (torch.arange(B)[:, None] * (nh * nw) + win[None, :]).reshape(-1)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (C,), gr, br, 1e-6)
    yr.backward(dy.float() + g_ref[win_all])
    mean, rstd = e(T), e(T)
    ops.layernorm_fwd(dev(x), dev(gamma), dev(beta), e(T, C, dtype=dtype), mean, rstd)
    g = ops.rvsa_sampling_bwd_win(dev(dsamp), dev(w), dev(avg), e(R, C))
    assert rel_err(g.cpu(), g_ref) < 1e-5
    dx, dgm, dbt = e(T, C), e(C), e(C)
    ops.layernorm_bwd(dev(dy, dtype), dev(x), mean, rstd, dev(gamma), dx, dgm, dbt, dres=dev(dres), win_add=g, grid=(B, Hp, Wp))
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert rel_err(dx.cpu(), xr.grad + dres) < tol and rel_err(dgm.cpu(), gr.grad) < tol and rel_err(dbt.cpu(), br.grad) < tol
    # the two-pass form it replaces (the bf16 form rounds dy + g to bf16 before the LayerNorm backward; the fused one adds in f32)
    dy2 = dev(dy, dtype).clone()
    ops.rvsa_sampling_bwd(dev(dsamp), dev(w), dev(avg), dy2, B, Hp, Wp)
    dx2, dgm2, dbt2 = e(T, C), e(C), e(C)
    ops.layernorm_bwd(dy2, dev(x), mean, rstd, dev(gamma), dx2, dgm2, dbt2, dres=dev(dres))
    assert rel_err(dx.cpu(), dx2.cpu()) < tol and rel_err(dgm.cpu(), dgm2.cpu()) < tol


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Hp,Wp,sscale", [(14, 14, 0.3), (16, 12, 0.3), (32, 32, 0.2), (14, 14, 0.0)])
def test_rvsa_attention_fwd_bwd(ops, dtype, Hp, Wp, sscale):
    """sscale = 0: identity sampling (samples on exact pixel centres; value path only).  32x32 -> padded 35x35, 25 windows."""
    B, heads, hd = 2, 2, 64
    C, T = heads * hd, B * Hp * Wp
    nh, nw = ops.rvsa_windows(Hp, Wp)
    R = B * nh * nw
    qkv = _attn_inputs(T, C, dtype, seed=7)
    samp = sscale * rnd(R, 5 * heads, seed=8)
    rh, rw, tab = 0.3 * rnd(13, hd, seed=1), 0.3 * rnd(13, hd, seed=2), 0.3 * rnd(169, heads, seed=3)
    o, lse = e(T, C, dtype=dtype), e(R * heads * 49)
    ops.rvsa_attn_fwd(dev(qkv, dtype), dev(samp), o, lse, dev(rh), dev(rw), dev(tab), B, Hp, Wp, heads, hd ** -0.5)
    q, sp = qkv.clone().requires_grad_(True), samp.clone().requires_grad_(True)
    rhr, rwr, tr = rh.clone().requires_grad_(True), rw.clone().requires_grad_(True), tab.clone().requires_grad_(True)
    oref, lref = O.rvsa_attn_fwd(q, sp, B, Hp, Wp, heads, rhr, rwr, tr)
    assert rel_err(o.float().cpu(), oref) < TOL[dtype]
    assert rel_err(lse.cpu().reshape(B, nh, nw, heads, 49).permute(0, 3, 1, 2, 4), lref) < (1e-4 if dtype == torch.float32 else 5e-3)
    do = rnd(T, C, dtype=dtype, seed=4)
    gq, gs, gh, gw, gt = torch.autograd.grad(oref, (q, sp, rhr, rwr, tr), do)
    dqkv, dsamp = e(T, 3 * C, dtype=dtype), e(R, 5 * heads)
    drh, drw, dtab = e(13, hd), e(13, hd), e(169, heads)
    ops.rvsa_attn_bwd(dev(qkv, dtype), dev(samp), o, dev(do, dtype), lse, dqkv, dsamp, dev(rh), dev(rw), dev(tab), drh, drw, dtab,
                      B, Hp, Wp, heads, hd ** -0.5)
    assert rel_err(dqkv.float().cpu(), gq) < TOL[dtype]
    assert rel_err(drh.cpu(), gh) < 10 * TOL[dtype] and rel_err(drw.cpu(), gw) < 10 * TOL[dtype] and rel_err(dtab.cpu(), gt) < 10 * TOL[dtype]
    if sscale > 0:   # at sscale == 0 every sample sits on a bilinear kink: d/d(coord) is one-sided (see test_oracle_golden)
        assert rel_err(dsamp.cpu(), gs) < 10 * TOL[dtype]


def test_rvsa_backward_atomic_scatter_form_in_a_subprocess(ops, tmp_path):
    """the backward's in-kernel f32-atomic scatter (grids beyond what the dense-product kernel takes; MTP_RVSA_SCATTER=dense forces it -- read once
    per process, hence the subprocess) against the default dense-product form on the same inputs: same bf16 W / dK_sel operands, f32 accumulation in
    a different order"""
    import subprocess
    import sys
    code = """
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from mtp_amd import ops
B, heads, hd, Hp, Wp = 2, 2, 64, 14, 14
C, T = heads * hd, B * Hp * Wp
g = torch.Generator().manual_seed(5)
qkv = (0.5 * torch.randn(T, 3 * C, generator=g)).cuda().bfloat16()
samp = (0.3 * torch.randn(B * 4, 5 * heads, generator=g)).cuda()
rh, rw, tab = (0.3 * torch.randn(13, hd, generator=g)).cuda(), (0.3 * torch.randn(13, hd, generator=g)).cuda(), (0.3 * torch.randn(169, heads, generator=g)).cuda()
do = torch.randn(T, C, generator=g).cuda().bfloat16()
o, lse = torch.empty(T, C, device="cuda", dtype=torch.bfloat16), torch.empty(B * 4 * heads * 49, device="cuda")
ops.rvsa_attn_fwd(qkv, samp, o, lse, rh, rw, tab, B, Hp, Wp, heads, hd ** -0.5)
dqkv, dsamp = torch.empty(T, 3 * C, device="cuda", dtype=torch.bfloat16), torch.empty(B * 4, 5 * heads, device="cuda")
d1, d2, d3 = torch.zeros(13, hd, device="cuda"), torch.zeros(13, hd, device="cuda"), torch.zeros(169, heads, device="cuda")
ops.rvsa_attn_bwd(qkv, samp, o, do, lse, dqkv, dsamp, rh, rw, tab, d1, d2, d3, B, Hp, Wp, heads, hd ** -0.5)
torch.cuda.synchronize()
torch.save({"dqkv": dqkv.float().cpu(), "dsamp": dsamp.cpu()}, sys.argv[1])
""" % (ROOT, os.path.join(ROOT, "tests"))
    outs = {}
    for mode in ("gemm", "dense"):
        f = str(tmp_path / (mode + ".pt"))
        env = dict(os.environ, MTP_RVSA_SCATTER=mode)
        r = subprocess.run([sys.executable, "-c", code, f], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = torch.load(f)
    assert rel_err(outs["dense"]["dqkv"], outs["gemm"]["dqkv"]) < 1e-2 and rel_err(outs["dense"]["dsamp"], outs["gemm"]["dsamp"]) < 1e-3


# ---- the attention kernels at the launch geometry of the headline benchmark: 64 images x 16 heads x 64 dims (ViT-L, B = 64)
@pytest.mark.parametrize("dtype", DT)
def test_full_attention_at_vit_l_b64_geometry(ops, dtype):
    """1024 (image, head) problems of 196 tokens, C = 1024 -- every workgroup of the real launch -- vs the oracle's autograd"""
    B, heads, hd, Hp, Wp = 64, 16, 64, 14, 14
    C, N = heads * hd, Hp * Wp
    T = B * N
    qkv = _attn_inputs(T, C, dtype, seed=11)
    rh, rw = 0.3 * rnd(2 * Hp - 1, hd, seed=1), 0.3 * rnd(2 * Wp - 1, hd, seed=2)
    o, lse = e(T, C, dtype=dtype), e(B * heads * N)
    ops.full_attn_fwd(dev(qkv, dtype), o, lse, dev(rh), dev(rw), B, Hp, Wp, heads, hd ** -0.5)
    q = qkv.clone().requires_grad_(True)
    rhr, rwr = rh.clone().requires_grad_(True), rw.clone().requires_grad_(True)
    oref, lref = O.full_attn_fwd(q, B, Hp, Wp, heads, rhr, rwr)
    assert rel_err(o.float().cpu(), oref) < TOL[dtype] and rel_err(lse.cpu().reshape(lref.shape), lref) < (1e-4 if dtype == torch.float32 else 5e-3)
    do = rnd(T, C, dtype=dtype, seed=3)
    gq, gh, gw = torch.autograd.grad(oref, (q, rhr, rwr), do)
    dqkv, drh, drw = e(T, 3 * C, dtype=dtype), e(*rh.shape), e(*rw.shape)
    ops.full_attn_bwd(dev(qkv, dtype), o, dev(do, dtype), lse, dqkv, dev(rh), dev(rw), drh, drw, B, Hp, Wp, heads, hd ** -0.5)
    assert rel_err(dqkv.float().cpu(), gq) < TOL[dtype]
    # the table gradients sum 64 x 16 x 196 x 196 terms: relative to their own size the bf16 rounding noise averages out
    assert rel_err(drh.cpu(), gh) < 10 * TOL[dtype] and rel_err(drw.cpu(), gw) < 10 * TOL[dtype]


@pytest.mark.parametrize("dtype", DT)
def test_rvsa_attention_at_vit_l_b64_geometry(ops, dtype):
    """16384 (image, window, head) problems -- the real launch of a ViT-L RVSA block at batch 64 -- forward and every gradient"""
    B, heads, hd, Hp, Wp = 64, 16, 64, 14, 14
    C, T = heads * hd, B * Hp * Wp
    nh, nw = ops.rvsa_windows(Hp, Wp)
    R = B * nh * nw
    qkv = _attn_inputs(T, C, dtype, seed=17)
    samp = 0.3 * rnd(R, 5 * heads, seed=8)
    rh, rw, tab = 0.3 * rnd(13, hd, seed=1), 0.3 * rnd(13, hd, seed=2), 0.3 * rnd(169, heads, seed=3)
    o, lse = e(T, C, dtype=dtype), e(R * heads * 49)
    ops.rvsa_attn_fwd(dev(qkv, dtype), dev(samp), o, lse, dev(rh), dev(rw), dev(tab), B, Hp, Wp, heads, hd ** -0.5)
    q, sp = qkv.clone().requires_grad_(True), samp.clone().requires_grad_(True)
    rhr, rwr, tr = rh.clone().requires_grad_(True), rw.clone().requires_grad_(True), tab.clone().requires_grad_(True)
    oref, lref = O.rvsa_attn_fwd(q, sp, B, Hp, Wp, heads, rhr, rwr, tr)
    assert rel_err(o.float().cpu(), oref) < TOL[dtype]
    do = rnd(T, C, dtype=dtype, seed=4)
    gq, gs, gh, gw, gt = torch.autograd.grad(oref, (q, sp, rhr, rwr, tr), do)
    dqkv, dsamp = e(T, 3 * C, dtype=dtype), e(R, 5 * heads)
    drh, drw, dtab = e(13, hd), e(13, hd), e(169, heads)
    ops.rvsa_attn_bwd(dev(qkv, dtype), dev(samp), o, dev(do, dtype), lse, dqkv, dsamp, dev(rh), dev(rw), dev(tab), drh, drw, dtab,
                      B, Hp, Wp, heads, hd ** -0.5)
    assert rel_err(dqkv.float().cpu(), gq) < TOL[dtype]
    assert rel_err(drh.cpu(), gh) < 10 * TOL[dtype] and rel_err(drw.cpu(), gw) < 10 * TOL[dtype] and rel_err(dtab.cpu(), gt) < 10 * TOL[dtype]
    assert rel_err(dsamp.cpu(), gs) < 10 * TOL[dtype]


# ------------------------------------------------------------------------------------------------ optimizer
def test_adamw_flat_and_sqnorm(ops):
    n = 4096 + 512
    p, g = rnd(n), rnd(n, seed=1)
    seg = torch.tensor([0, 1024, 4096], dtype=torch.int64)
    wd = torch.tensor([0.05, 0.0, 0.05])
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    ref_p = [p[:1024].clone().requires_grad_(True), p[1024:4096].clone().requires_grad_(True), p[4096:].clone().requires_grad_(True)]
    opt = torch.optim.AdamW([{"params": [ref_p[0], ref_p[2]], "weight_decay": 0.05}, {"params": [ref_p[1]], "weight_decay": 0.0}], lr=lr, betas=(b1, b2), eps=eps)
    dp, dg, dm, dv = dev(p), dev(g), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    sq = torch.zeros(1, device="cuda")
    for step in range(1, 4):
        for rp, sl in zip(ref_p, (slice(0, 1024), slice(1024, 4096), slice(4096, n))):
            rp.grad = g[sl].clone()
        torch.nn.utils.clip_grad_norm_(ref_p, 5.0)
        opt.step()
        sq.zero_()
        ops.sqnorm(dg, sq)
        assert abs(sq.item() - float((g ** 2).sum())) < 1e-3 * float((g ** 2).sum())
        hyper = torch.tensor([lr, b1, b2, eps, 1 - b1 ** step, 1 - b2 ** step], device="cuda")
        ops.adamw_flat(dp, dg, dm, dv, dev(seg), dev(wd), hyper, sq, max_norm=5.0)
    assert rel_err(dp.cpu(), torch.cat([r.detach() for r in ref_p])) < 1e-5


def test_cu_mask_stream_entry_points(ops):
    """mtp_stream_create_cu_mask / mtp_probe_placement (round 6): a launch on the masked stream touches exactly the CUs of the mask (16 of every XCC for a
    half), the GEMM dispatch sized by the stream's CU count gives the same bits as on the whole chip, masks that leave an XCC empty are refused."""
    import ctypes as C
    from mtp_amd import _lib
    lib = _lib.load()
    for kind in ("interleaved", "blocked"):
        seen = []
        for part in range(2):
            st = ops.cu_mask_stream("cuda", ops.cu_mask_words(kind, part))
            rec = torch.full((2048, 2), -1, device="cuda", dtype=torch.int32)
            torch.cuda.synchronize()
            assert lib.mtp_probe_placement(C.c_void_p(rec.data_ptr()), 2048, 40000, C.c_void_p(st.cuda_stream)) == 0
            st.synchronize()
            r = rec.cpu()
            xcc, hw = r[:, 0] & 15, r[:, 1]
            slots = set((int(x) << 12) | ((int(h) >> 8) & 0xff) for x, h in zip(xcc, hw))     # (XCC, se | sh | cu)
            assert len(slots) == 128, "%s half %d ran on %d CUs" % (kind, part, len(slots))
            assert all(int((xcc == x).sum()) == 256 for x in range(8))        # every XCC still gets one workgroup in eight
            seen.append(slots)
        assert not (seen[0] & seen[1])
    # the same GEMM on the whole chip and on a 128-CU stream (other tile height / kernel family by CU count): bit-identical
    a = torch.randn(6272, 1024, device="cuda").bfloat16()
    w = torch.randn(1024, 1024, device="cuda").bfloat16()
    y0 = ops.gemm_nt(a, w, torch.empty(6272, 1024, device="cuda", dtype=torch.bfloat16))
    st = ops.cu_mask_stream("cuda", ops.cu_mask_words("interleaved", 0))
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        y1 = ops.gemm_nt(a, w, torch.empty(6272, 1024, device="cuda", dtype=torch.bfloat16))
    st.synchronize()
    assert torch.equal(y0, y1)
    h = C.c_void_p()
    bad = (C.c_uint32 * 8)(*([0x7f7f7f7f] * 8))      # XCC 7 without any CU
    assert lib.mtp_stream_create_cu_mask(bad, 8, C.byref(h)) == -1
    assert lib.mtp_stream_create_cu_mask(None, 8, C.byref(h)) == -1


def test_low_priority_stream_entry_points(ops):
    """mtp_stream_create_low_priority / mtp_stream_destroy (round 4): the handle is a real HIP stream -- kernels launched on it through the C ABI run and
    are ordered by events against the compute stream -- and ops.low_priority_stream wraps one per device for torch (the weight-gradient side stream)."""
    import ctypes as C
    from mtp_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.mtp_stream_create_low_priority(C.byref(h)) == 0 and h.value
    st = torch.cuda.ExternalStream(h.value)
    x = torch.randn(4096, 256, device="cuda")
    out = torch.zeros(256, device="cuda")
    ready = torch.cuda.Event()
    ready.record()
    with torch.cuda.stream(st):
        st.wait_event(ready)
        ops.reduce_rows(x, out)
    st.synchronize()
    assert rel_err(out, x.sum(0)) < 1e-5
    assert lib.mtp_stream_destroy(h) == 0
    assert lib.mtp_stream_create_low_priority(None) != 0 and lib.mtp_stream_destroy(None) != 0
    a, b = ops.low_priority_stream("cuda"), ops.low_priority_stream(torch.device("cuda", torch.cuda.current_device()))
    assert a is b and a.cuda_stream != torch.cuda.current_stream().cuda_stream
