"""Per-kernel parity: every C-ABI entry point against fp64 / oracle math on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from drakegpt_amd import ops
    return ops


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def maxabs(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(256, 96, 32), (300, 80, 384), (1000, 1152, 384), (128, 128, 64), (2048, 384, 1536), (77, 50257 // 16, 64)])
def test_gemm_nt_plain(dev, dtype, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    B = torch.randn(N, K, generator=g).to(dtype)
    ref = A.double() @ B.double().T
    out = ops.gemm_nt(A.to(dev), B.to(dev), torch.float32)
    torch.cuda.synchronize()
    assert rel(out, ref) < 2e-6, rel(out, ref)


def test_gemm_nt_asymmetric_identity(dev):
    """A = I against an asymmetric B catches a transposed C/D map (cdna guide section 3)."""
    ops = _ops()
    K = 128
    A = torch.eye(K)
    B = torch.arange(96 * K, dtype=torch.float32).reshape(96, K) / 7.0
    for dt in (torch.float32, torch.bfloat16):
        out = ops.gemm_nt(A.to(dt).to(dev), B.to(dt).to(dev), torch.float32)
        assert torch.equal(out.cpu(), B.to(dt).float().T.contiguous()), dt


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(320, 200, 128), (300, 384, 128), (1100, 1152, 192), (256, 1536, 128), (40000, 192, 128)])
def test_gemm_nt_epilogue(dev, dtype, M, N, K):
    """every epilogue option on: square tiles (N = 200), 128 x 192 tiles (N % 192 == 0) with ragged M, several tiles
    per workgroup (40000 x 192: 313 tiles), and -- bf16 output with a 16-byte-aligned mask -- the mask-prefetch variant"""
    from oracle import rng_ref
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).to(dtype)
    B = torch.randn(N, K, generator=g).to(dtype)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    mask = torch.randn(M, N, generator=g).to(dtype)
    acc = A.double() @ B.double().T + bias.double()
    # bias + relu, activation-dtype output
    out = ops.gemm_nt(A.to(dev), B.to(dev), dtype, bias=bias.to(dev), relu=True)
    ref = acc.clamp_min(0)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel(out, ref) < tol
    # relu-mask (backward of ReLU fused into the dX GEMM)
    out = ops.gemm_nt(A.to(dev), B.to(dev), torch.float32, relu_mask=mask.to(dev))
    ref = (A.double() @ B.double().T) * (mask.double() > 0)
    assert rel(out, ref) < 2e-6
    out = ops.gemm_nt(A.to(dev), B.to(dev), dtype, relu_mask=mask.to(dev), bias=bias.to(dev))
    assert rel(out, acc * (mask.double() > 0)) < tol
    # the same mask as one bit per element: emitted by the Linear+ReLU GEMM, consumed by the dX GEMM
    if ops.gemm_nt_sign_bits_supported(dtype, N, K):
        bits = ops.new_sign_bits(M, N, dev)
        out = ops.gemm_nt(A.to(dev), B.to(dev), dtype, bias=bias.to(dev), relu=True, sign_bits_out=bits)
        assert rel(out, acc.clamp_min(0)) < tol
        pos = (out.float().cpu() > 0)
        out2 = ops.gemm_nt(A.to(dev), B.to(dev), dtype, bias=bias.to(dev), sign_bits=bits)
        assert rel(out2, acc * pos.double()) < tol
        out3 = ops.gemm_nt(A.to(dev), B.to(dev), torch.float32, sign_bits=bits)
        assert rel(out3, (A.double() @ B.double().T) * pos.double()) < 2e-6
    else:
        assert dtype == torch.float32 or N % 8
        with pytest.raises(RuntimeError):
            ops.gemm_nt(A.to(dev), B.to(dev), dtype, sign_bits=torch.zeros(M * N, dtype=torch.uint8, device=dev))
    # bias + dropout + residual, fp32 out; mask recomputed on the host from the same hash
    seed, step, site, p = 1234, 5, 9, 0.25
    rng = ops.new_rng_state(seed, dev, step)
    out = ops.gemm_nt(A.to(dev), B.to(dev), torch.float32, bias=bias.to(dev), dropout_p=p, rng_state=rng, site=site, residual=resid.to(dev))
    keep = torch.from_numpy(rng_ref.keep_mask(seed, step, site, p, M * N).reshape(M, N)).double()
    ref = acc * keep / (1 - p) + resid.double()
    assert rel(out, ref) < 2e-6
    frac = keep.mean().item()
    assert abs(frac - (1 - p)) < 0.01, frac


@pytest.mark.parametrize("M,N,K,ld", [(256, 1536, 384, 1536), (16384, 1536, 128, 1536), (1024, 384, 128, 400), (4096, 128, 128, 128), (41088, 192, 128, 192)])
def test_gemm_nt_colsum_epilogue(dev, M, N, K, ld):
    """the sign-bit-masked dX GEMM also writes partial rows of the column sums of its output (FeedForward's first bias
    gradient): wide and square tiles, a strided partial buffer, one partial row per 32 rows (16 tiles on 16 workgroups) and
    one per workgroup and wave row (1024 / 321 tiles, several per workgroup)"""
    ops = _ops()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).to(bf)
    B = torch.randn(N, K, generator=g).to(bf)
    W = torch.randn(N, K, generator=g).to(bf)
    rows = ops.gemm_nt_colsum_rows(bf, M, N, K)
    assert 0 < rows <= M // 32
    bits = ops.new_sign_bits(M, N, dev)
    f = ops.gemm_nt(A.to(dev), W.to(dev), bf, bias=torch.zeros(N, device=dev), relu=True, sign_bits_out=bits)
    pos = (f.float().cpu() > 0).double()
    part = torch.full((rows + 3, ld), 7.0, device=dev)
    out = ops.gemm_nt(A.to(dev), B.to(dev), bf, sign_bits=bits, colsum_part=part[:rows, :N])
    ref = (A.double() @ B.double().T) * pos
    assert rel(out, ref) < 6e-3
    got = part.cpu().double()
    assert torch.all(got[rows:] == 7.0) and torch.all(got[:, N:] == 7.0)          # nothing outside [rows, N] is touched
    assert not torch.any(got[:rows, :N] == 7.0)                                   # every partial row is written
    assert rel(got[:rows, :N].sum(0), ref.sum(0)) < 1e-5
    total = torch.empty(N, device=dev)
    ops.reduce_partials(part, ld, rows, total, N)
    assert rel(total, ref.sum(0)) < 1e-5
    # not offered where a tile would be partial, or for fp32 operands; too few partial rows are refused
    assert ops.gemm_nt_colsum_rows(bf, M + 32, N, K) == 0
    assert ops.gemm_nt_colsum_rows(bf, M, N + 8, K) == 0
    assert ops.gemm_nt_colsum_rows(torch.float32, M, N, K) == 0
    with pytest.raises(RuntimeError):
        ops.gemm_nt(A.to(dev), B.to(dev), bf, sign_bits=bits, colsum_part=part[:rows - 1, :N])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,P,Q,S", [(256, 96, 32, 1), (4096, 384, 384, 2), (1000, 80, 384, 3), (640, 1152, 384, 1), (16384, 128, 128, 8), (130, 200, 72, 2)])
def test_gemm_tn(dev, dtype, R, P, Q, S):
    ops = _ops()
    g = torch.Generator().manual_seed(R + P + Q)
    A = torch.randn(R, P, generator=g).to(dtype)
    B = torch.randn(R, Q, generator=g).to(dtype)
    ref = A.double().T @ B.double()
    part = torch.full((S, P, Q), float("nan"), device=dev)
    ops.gemm_tn(A.to(dev), B.to(dev), part, P * Q, S, P, Q)
    out = torch.empty(P, Q, device=dev)
    ops.reduce_partials(part, P * Q, S, out, P * Q)
    assert rel(out, ref) < 3e-6, rel(out, ref)


@pytest.mark.parametrize("with_short", [True, False])
def test_gemm_tn_grouped(dev, with_short):
    """one launch for many dW problems (different shapes and contraction lengths, ragged P/Q, column-sliced operands);
    more tiles than CUs so workgroups walk several tiles of different problems with one stage pipeline.  Without the
    one-K-step problem the workspace run splits every tile into two K halves chained through flags."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    shapes = [(1024, 384, 384), (1024, 1152, 384), (1024, 384, 1536), (1024, 1536, 384), (1024, 80, 384), (64 if with_short else 320, 200, 72),
              (2048, 1536, 768), (512, 128, 128), (192, 264, 520)] * 2
    shapes += [(256, 2048, 2048)]                       # 256 tiles on its own
    probs, refs, outs = [], [], []
    for i, (R, P, Q) in enumerate(shapes):
        lda = (P + 7) // 8 * 8 + (8 if i % 3 == 0 else 0)            # padded / sliced leading dimensions
        ldb = (Q + 7) // 8 * 8
        A = torch.zeros(R, lda, dtype=torch.bfloat16)
        A[:, :P] = torch.randn(R, P, generator=g).to(torch.bfloat16)
        B = torch.zeros(R, ldb, dtype=torch.bfloat16)
        B[:, :Q] = torch.randn(R, Q, generator=g).to(torch.bfloat16)
        refs.append(A[:, :P].double().T @ B[:, :Q].double())
        out = torch.full((P * Q,), float("nan"), device=dev)
        outs.append(out)
        probs.append((A.to(dev)[:, :P] if lda != P else A.to(dev), B.to(dev)[:, :Q] if ldb != Q else B.to(dev), out, P, Q))
    assert sum(((P + 255) // 256) * ((Q + 127) // 128) for _, P, Q in shapes) > 256
    ws = ops.gemm_tn_grouped_workspace(probs, dev)
    first = None
    for workspace in (None, ws, ws):                    # whole tiles, then K halves twice (the flags must reset themselves)
        for out in outs:
            out.fill_(float("nan"))
        ops.gemm_tn_grouped(probs, workspace)
        torch.cuda.synchronize()
        for out, ref, (R, P, Q) in zip(outs, refs, shapes):
            assert rel(out.view(P, Q), ref) < 3e-6, (workspace is not None, R, P, Q, rel(out.view(P, Q), ref))
        if workspace is not None:
            if first is None:
                first = [o.clone() for o in outs]
            else:
                assert all(torch.equal(a, b) for a, b in zip(first, outs))     # run-to-run identical
    with pytest.raises(RuntimeError):                   # contraction length must be a multiple of 64
        ops.gemm_tn_grouped([(torch.zeros(100, 64, dtype=torch.bfloat16, device=dev), torch.zeros(100, 64, dtype=torch.bfloat16, device=dev),
                              torch.zeros(64 * 64, device=dev), 64, 64)])


def _tn_status(ws):
    return int(ws[-16:].view(torch.int32)[0].item())


@pytest.mark.parametrize("R", [128, 704])
def test_gemm_tn_grouped_every_tile_split_single_round(dev, R):
    """one layer's problem set (66 tiles): every tile is cut into two K halves, ONE item per workgroup (144 <= #CUs), the second
    half on workgroup 72 + t waiting for workgroup t.  Same result as the unsplit launch; the bounded wait never runs out."""
    ops = _ops()
    g = torch.Generator().manual_seed(R)
    C, V = 384, 80
    shapes = [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C), (V, C)]
    assert sum(((P + 255) // 256) * ((Q + 127) // 128) for P, Q in shapes) == 66
    probs, outs = [], []
    for P, Q in shapes:
        A = torch.randn(R, (P + 7) // 8 * 8, generator=g).to(torch.bfloat16).to(dev)
        B = torch.randn(R, Q, generator=g).to(torch.bfloat16).to(dev)
        outs.append(torch.full((P * Q,), float("nan"), device=dev))
        probs.append((A[:, :P], B, outs[-1], P, Q))
    ws = ops.gemm_tn_grouped_workspace(probs, dev)
    ops.gemm_tn_grouped(probs, None)
    whole = [o.clone() for o in outs]
    for _ in range(2):
        for o in outs:
            o.fill_(float("nan"))
        ops.gemm_tn_grouped(probs, ws)
        for (A, B, _, P, Q), o, w in zip(probs, outs, whole):
            assert rel(o.view(P, Q), A.double().T @ B.double()) < 3e-6 and rel(o, w) < 1e-6
    assert _tn_status(ws) == 0


@pytest.mark.parametrize("R", [192, 1024])
def test_gemm_tn_grouped_leftover_split(dev, R):
    """the training step's own problem set (6 layers x 4 Linears + lm_head + token table = 384 tiles of 256 x 128): with the
    workspace one whole tile per workgroup plus one half of a leftover tile (R = 192: an odd number of K steps, halves of 1
    and 2); same results as without the workspace, bit-identical from run to run, flags reset for the next launch"""
    ops = _ops()
    g = torch.Generator().manual_seed(R)
    C, V = 384, 80
    shapes = [(V, C)] + [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)] * 6 + [(V, C)]
    assert sum(((P + 255) // 256) * ((Q + 127) // 128) for P, Q in shapes) == 384
    probs, outs = [], []
    for P, Q in shapes:
        A = torch.randn(R, (P + 7) // 8 * 8, generator=g).to(torch.bfloat16).to(dev)
        B = torch.randn(R, Q, generator=g).to(torch.bfloat16).to(dev)
        out = torch.full((P * Q,), float("nan"), device=dev)
        outs.append(out)
        probs.append((A[:, :P], B, out, P, Q))
    ws = ops.gemm_tn_grouped_workspace(probs, dev)
    ops.gemm_tn_grouped(probs, None)
    whole = [o.clone() for o in outs]
    for (A, B, _, P, Q), o in zip(probs, whole):
        assert rel(o.view(P, Q), A.double().T @ B.double()) < 3e-6
    runs = []
    for _ in range(2):
        for o in outs:
            o.fill_(float("nan"))
        ops.gemm_tn_grouped(probs, ws)
        runs.append([o.clone() for o in outs])
    for a, b, w, (P, Q) in zip(runs[0], runs[1], whole, shapes):
        assert torch.equal(a, b)
        assert rel(a, w) < 1e-6, (P, Q)
    assert _tn_status(ws) == 0


def test_gemm_tn_grouped_leftover_switch_off_never_splits_multi_round(dev):
    """DG_TN_LEFTOVER=0 (A/B switch, read once per process: child process): with 384 tiles on 256 CUs "cut every tile" would put
    second halves on lower-numbered workgroups than their producers (a deadlock when part of the grid is not resident -- ADVICE
    r1); the launch rules now refuse that and run whole tiles: correct results, error word clear."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, torch
sys.path.insert(0, %r)
from drakegpt_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
C, V, R = 384, 80, 256
shapes = [(V, C)] + [(C, 4 * C), (4 * C, C), (C, C), (3 * C, C)] * 6 + [(V, C)]
probs = []
for P, Q in shapes:
    A = torch.randn(R, (P + 7) // 8 * 8, generator=g).to(torch.bfloat16).to(dev)
    B = torch.randn(R, Q, generator=g).to(torch.bfloat16).to(dev)
    probs.append((A[:, :P], B, torch.full((P * Q,), float("nan"), device=dev), P, Q))
ws = ops.gemm_tn_grouped_workspace(probs, dev)
ops.gemm_tn_grouped(probs, ws)
torch.cuda.synchronize()
for A, B, o, P, Q in probs:
    ref = A.double().T @ B.double()
    assert ((o.view(P, Q).double() - ref).norm() / ref.norm()).item() < 3e-6
assert int(ws[-16:].view(torch.int32)[0].item()) == 0
print("leftover-off ok")
""" % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DG_TN_LEFTOVER="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "leftover-off ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("G,n,stride", [(256, 23104, 23104), (256, 1000, 1024), (37, 4096, 4100), (32, 2050, 2052), (8, 77, 80), (300, 64, 64)])
def test_reduce_partials(dev, G, n, stride):
    """both kernels (one output per thread / the tall many-partials form), padded strides, run-to-run identical"""
    ops = _ops()
    g = torch.Generator().manual_seed(G + n)
    part = torch.randn(G, stride, generator=g)
    ref = part[:, :n].double().sum(0)
    pd = part.to(dev)
    out = torch.full((n,), float("nan"), device=dev)
    ops.reduce_partials(pd, stride, G, out, n)
    assert rel(out, ref) < 2e-6, rel(out, ref)
    out2 = torch.empty_like(out)
    ops.reduce_partials(pd, stride, G, out2, n)
    assert torch.equal(out, out2)


def test_gemm_tn_asymmetric(dev):
    """exact integer data through the transposed LDS reads (bf16) -- every element must match."""
    ops = _ops()
    R, P, Q = 128, 48, 40
    A = ((torch.arange(R * P).reshape(R, P) % 7) - 3).float()
    B = ((torch.arange(R * Q).reshape(R, Q) % 5) - 2).float()
    ref = A.T @ B
    for dt in (torch.bfloat16, torch.float32):
        part = torch.empty((1, P, Q), device=dev)
        ops.gemm_tn(A.to(dt).to(dev), B.to(dt).to(dev), part, P * Q, 1, P, Q)
        assert torch.equal(part[0].cpu(), ref), dt


@pytest.mark.parametrize("C", [32, 384, 768, 1024, 100, 2048])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_layernorm(dev, C, out_dtype):
    ops = _ops()
    M = 517
    g = torch.Generator().manual_seed(C)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    w = torch.randn(C, generator=g)
    b = torch.randn(C, generator=g)
    dy = torch.randn(M, C, generator=g)
    dres = torch.randn(M, C, generator=g)
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xd, (C,), wd, bd, 1e-5)
    ref.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), out_dtype)
    assert rel(y, ref.detach()) < (1e-6 if out_dtype == torch.float32 else 4e-3)
    G = 16
    pg = torch.empty(G, C, device=dev)
    pb = torch.empty(G, C, device=dev)
    dx = ops.layernorm_bwd(dy.to(dev), x.to(dev), w.to(dev), mean, rstd, dres.to(dev), pg, pb, C, G)
    assert rel(dx, xd.grad + dres.double()) < 2e-6
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ops.reduce_partials(pg, C, G, dg, C)
    ops.reduce_partials(pb, C, G, db, C)
    assert rel(dg, wd.grad) < 2e-6 and rel(db, bd.grad) < 2e-6
    if C % 4 == 0:
        # dy handed over in bf16 (what a bf16-output dX GEMM produces): exact for the rounded dy
        dyb = dy.bfloat16()
        xd2 = x.double().requires_grad_(True)
        wd2, bd2 = w.double().requires_grad_(True), b.double().requires_grad_(True)
        torch.nn.functional.layer_norm(xd2, (C,), wd2, bd2, 1e-5).backward(dyb.double())
        dx = ops.layernorm_bwd(dyb.to(dev), x.to(dev), w.to(dev), mean, rstd, dres.to(dev), pg, pb, C, G)
        assert rel(dx, xd2.grad + dres.double()) < 2e-6
        ops.reduce_partials(pg, C, G, dg, C)
        ops.reduce_partials(pb, C, G, db, C)
        assert rel(dg, wd2.grad) < 2e-6 and rel(db, bd2.grad) < 2e-6
    else:
        with pytest.raises(RuntimeError):
            ops.layernorm_bwd(dy.bfloat16().to(dev), x.to(dev), w.to(dev), mean, rstd, dres.to(dev), pg, pb, C, G)


@pytest.mark.parametrize("C", [32, 384, 768, 1024])
@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("dy_dtype,g_dtype", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16)])
def test_layernorm_bwd_fused(dev, C, p, dy_dtype, g_dtype):
    """dg_layernorm_bwd_fused against fp64: dx = dresid + LN'(dy) (ref: nn.LayerNorm at src/model_component.py:488-489 under the
    residual add of :505-506), g = dropout_bwd(dx) with the keep-mask of `site` (ref: backward of nn.Dropout at :454 / :324, mask
    from oracle/rng_ref.py), the bias partial rows = column sums of the unrounded g, and the dgamma / dbeta partials."""
    from oracle import rng_ref
    ops = _ops()
    M, G, site, seed, step = 1000, 24, 9, 4321, 3           # ragged row chunks: 1000 = 23 * 42 + 34
    gen = torch.Generator().manual_seed(C + int(100 * p))
    x = torch.randn(M, C, generator=gen) * 1.5 - 0.3
    w = torch.randn(C, generator=gen)
    b = torch.randn(C, generator=gen)
    dy = torch.randn(M, C, generator=gen).to(dy_dtype)
    dres = torch.randn(M, C, generator=gen)
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd, (C,), wd, bd, 1e-5).backward(dy.double())
    dx_ref = xd.grad + dres.double()
    keep = torch.from_numpy(rng_ref.keep_mask(seed, step, site, p, M * C).reshape(M, C)).double() if p > 0 else torch.ones(M, C).double()
    g_ref = dx_ref * keep / (1.0 - p)
    _, mean, rstd = ops.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), torch.float32)
    pg, pb, pq = (torch.full((G, C + 8), float("nan"), device=dev) for _ in range(3))
    rng = ops.new_rng_state(seed, dev, step)
    dx, gq = ops.layernorm_bwd_fused(dy.to(dev), x.to(dev), w.to(dev), mean, rstd, dres.to(dev), pg, pb, C + 8, G, g_dtype, p, rng, site, pq)
    torch.cuda.synchronize()
    assert gq.dtype == g_dtype and gq.shape == (M, C)
    assert rel(dx, dx_ref) < 2e-6
    if g_dtype == torch.float32:
        assert rel(gq, g_ref) < 2e-6
    else:
        # exactly the bf16 rounding of the kernel's own fp32 g: dropped elements are exact zeros, kept ones within half an ulp
        assert torch.all(gq.float().cpu()[keep == 0] == 0)
        assert rel(gq, g_ref) < 3e-3 and maxabs(gq, g_ref.float().bfloat16()) <= 2 ** -7 * g_ref.abs().max().item()
    out = torch.empty(3, C, device=dev)
    for i, part in enumerate((pg, pb, pq)):
        assert torch.isfinite(part[:, :C]).all()            # every partial row written
        ops.reduce_partials(part, C + 8, G, out[i], C)
    assert rel(out[0], wd.grad) < 3e-6 and rel(out[1], bd.grad) < 3e-6
    assert rel(out[2], g_ref.sum(0)) < 3e-6                 # column sums of g BEFORE the bf16 rounding
    # without a bias behind it (first block: g is the X operand of the token-table problem): gbias_part = NULL
    dx2, g2 = ops.layernorm_bwd_fused(dy.to(dev), x.to(dev), w.to(dev), mean, rstd, None, pg, pb, C + 8, G, g_dtype, 0.0, None, 0, None)
    assert rel(dx2, xd.grad) < 2e-6 and rel(g2, xd.grad) < (2e-6 if g_dtype == torch.float32 else 3e-3)


@pytest.mark.parametrize("C", [384, 768])
def test_layernorm_bwd_fused_bf16_gradient_stream(dev, C):
    """the engine's bf16 / fp8 modes: dresid arrives and dx leaves in bf16 (stream_dtype): dx is the bf16 rounding of the fp32
    result on the bf16 dresid, g and the partial rows are computed from the UNROUNDED dx"""
    from oracle import rng_ref
    ops = _ops()
    M, G, site, seed, step, p = 1000, 24, 5, 11, 2, 0.2
    gen = torch.Generator().manual_seed(C)
    x = torch.randn(M, C, generator=gen)
    w, b = torch.randn(C, generator=gen), torch.randn(C, generator=gen)
    dy = torch.randn(M, C, generator=gen).bfloat16()
    dres = torch.randn(M, C, generator=gen).bfloat16()
    xd = x.double().requires_grad_(True)
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xd, (C,), wd, bd, 1e-5).backward(dy.double())
    dx_ref = xd.grad + dres.double()
    keep = torch.from_numpy(rng_ref.keep_mask(seed, step, site, p, M * C).reshape(M, C)).double()
    g_ref = dx_ref * keep / (1.0 - p)
    _, mean, rstd = ops.layernorm_fwd(x.to(dev), w.to(dev), b.to(dev), torch.float32)
    pg, pb, pq = (torch.full((G, C), float("nan"), device=dev) for _ in range(3))
    rng = ops.new_rng_state(seed, dev, step)
    dx, gq = ops.layernorm_bwd_fused(dy.to(dev), x.to(dev), w.to(dev), mean, rstd, dres.to(dev), pg, pb, C, G, torch.bfloat16, p, rng, site, pq,
                                     stream_dtype=torch.bfloat16)
    assert dx.dtype == torch.bfloat16 and gq.dtype == torch.bfloat16
    assert rel(dx, dx_ref) < 3e-3 and maxabs(dx, dx_ref.float().bfloat16()) <= 2 ** -7 * dx_ref.abs().max().item()
    assert rel(gq, g_ref) < 3e-3 and torch.all(gq.float().cpu()[keep == 0] == 0)
    out = torch.empty(3, C, device=dev)
    for i, part in enumerate((pg, pb, pq)):
        ops.reduce_partials(part, C, G, out[i], C)
    assert rel(out[0], wd.grad) < 3e-6 and rel(out[1], bd.grad) < 3e-6 and rel(out[2], g_ref.sum(0)) < 3e-6
    with pytest.raises(RuntimeError):          # the bf16 stream exists for the bf16 operand form only
        ops.layernorm_bwd_fused(dy.float().to(dev), x.to(dev), w.to(dev), mean, rstd, dres.to(dev), pg, pb, C, G, torch.float32, 0.0, None, 0, None,
                                stream_dtype=torch.bfloat16)
    # consumers of the stream's two ends: dropout-backward of a bf16 gradient (top of the stack) and the embedding backward
    part = torch.empty(G, C, device=dev)
    g2 = ops.dropout_bwd_cast(dres.to(dev), torch.bfloat16, p, rng, site, colsum_part=part, part_stride=C, n_partials=G)
    want = dres.double() * keep / (1.0 - p)
    assert rel(g2, want) < 3e-3 and rel(part.sum(0), want.sum(0)) < 1e-5
    B, T = 8, 125
    idx = torch.randint(0, 50, (B, T), generator=gen).to(dev)
    d3 = dres.to(dev).view(B, T, C)
    dtok, dpos = torch.empty(50, C, device=dev), torch.empty(T, C, device=dev)
    ops.embed_bwd(idx, d3, dtok, dpos)
    ref_tok = torch.zeros(50, C, dtype=torch.float64).index_add_(0, idx.cpu().view(-1), dres.double())
    assert rel(dpos, dres.double().view(B, T, C).sum(0)) < 1e-6 and rel(dtok, ref_tok) < 1e-5


def test_layernorm_bwd_fused_rejects_unsupported_width(dev):
    ops = _ops()
    M, C = 64, 1028 * 2
    assert not ops.layernorm_bwd_fused_supported(C) and not ops.layernorm_bwd_fused_supported(30)
    z = torch.zeros(M, C, device=dev)
    v = torch.zeros(C, device=dev)
    r = torch.zeros(M, device=dev)
    part = torch.zeros(4, C, device=dev)
    with pytest.raises(RuntimeError):
        ops.layernorm_bwd_fused(z, z, v, r, r, None, part, part, C, 4, torch.float32, 0.0, None, 0, None)


def _attn_ref(qkv, B, T, NH, H, keep=None, p=0.0):
    """fp64 reference of Head2.forward over packed qkv [B*T, 3*NH*H]; returns out, and is differentiable"""
    C = NH * H
    q, k, v = qkv.view(B, T, 3, NH, H).permute(2, 0, 3, 1, 4)          # each (B,NH,T,H)
    w = q @ k.transpose(-2, -1) * H ** -0.5
    tril = torch.tril(torch.ones(T, T, dtype=torch.bool))
    w = w.masked_fill(~tril, float("-inf")).softmax(-1)
    if keep is not None:
        w = w * keep / (1 - p)
    o = w @ v                                                           # (B,NH,T,H)
    return o.permute(0, 2, 1, 3).reshape(B * T, C)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,NH,H,p", [(2, 8, 4, 8, 0.0), (3, 37, 2, 16, 0.1), (2, 256, 3, 64, 0.2), (1, 1, 1, 32, 0.0), (2, 130, 2, 64, 0.0),
                                        (3, 100, 2, 64, 0.3)])
def test_attention(dev, dtype, B, T, NH, H, p):
    from oracle import rng_ref
    ops = _ops()
    g = torch.Generator().manual_seed(T * 7 + H)
    C = NH * H
    qkv = (torch.randn(B * T, 3 * C, generator=g)).to(dtype)
    dout = torch.randn(B * T, C, generator=g).to(dtype)
    seed, step, site = 77, 3, 4
    keep = None
    rng = None
    if p > 0:
        keep = torch.from_numpy(rng_ref.keep_mask(seed, step, site, p, B * NH * T * T).reshape(B, NH, T, T)).double()
        rng = ops.new_rng_state(seed, dev, step)
    qd = qkv.double().requires_grad_(True)
    ref = _attn_ref(qd, B, T, NH, H, keep, p)
    ref.backward(dout.double())
    out, lse = ops.attn_fwd(qkv.to(dev), B, T, NH, H, H ** -0.5, p, rng, site)
    tol = 2e-5 if dtype == torch.float32 else 8e-3
    assert rel(out, ref.detach()) < tol, rel(out, ref.detach())
    dqkv = ops.attn_bwd(qkv.to(dev), out, dout.to(dev), lse, B, T, NH, H, H ** -0.5, p, rng, site)
    # backward consumes the kernel's own (rounded) forward output: compare against fp64 grads
    tolg = 3e-5 if dtype == torch.float32 else 2e-2
    assert rel(dqkv, qd.grad) < tolg, rel(dqkv, qd.grad)
    # round 3: the forward pass can leave its keep decisions as wave masks and the dQ pass then selects with them instead of hashing
    # again -- the same decisions, hence the same output and the same gradient bit for bit (shapes on the generic kernels: no masks)
    out_k, lse_k = ops.attn_fwd(qkv.to(dev), B, T, NH, H, H ** -0.5, p, rng, site, keep=True)
    has = getattr(out_k, "dg_keep", None) is not None
    assert has == (p > 0 and dtype == torch.bfloat16 and H == 64)
    # (the two template instances contract their fp32 expressions differently: an ulp of lse, never a decision)
    assert torch.equal(out_k, out) and rel(lse_k, lse) < 1e-6
    if has:
        assert out_k.dg_keep.numel() == B * NH * (((T + 31) // 32) * ((T + 31) // 32 + 1) // 2) * 128
        dqkv_k = ops.attn_bwd(qkv.to(dev), out, dout.to(dev), lse, B, T, NH, H, H ** -0.5, p, rng, site, keep_bits=out_k.dg_keep)
        a, b = dqkv_k.float(), dqkv.float()
        # a wrong keep decision moves an element by many bf16 ulps; contraction differences by one at most
        assert bool(((a - b).abs() <= 2.0 ** -6 * torch.maximum(a.abs(), b.abs()) + 1e-6 * b.abs().max()).all())
        assert rel(a, b) < 2e-4, rel(a, b)
        with pytest.raises(RuntimeError):                      # a record buffer that is too small is rejected before the launch
            ops.attn_bwd(qkv.to(dev), out, dout.to(dev), lse, B, T, NH, H, H ** -0.5, p, rng, site, keep_bits=out_k.dg_keep[:-128])


def test_cross_entropy_and_reduce(dev):
    ops = _ops()
    for M, V in ((256, 80), (64, 50257), (33, 7)):
        g = torch.Generator().manual_seed(V)
        logits = torch.randn(M, V, generator=g) * 3
        tgt = torch.randint(0, V, (M,), generator=g)
        ld = logits.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(ld, tgt)
        ref.backward()
        Vp = (V + 7) // 8 * 8
        for dt in (torch.float32, torch.bfloat16):
            dl = torch.full((M, Vp), float("nan"), dtype=dt, device=dev)
            rows = ops.cross_entropy(logits.to(dev), tgt.to(dev), V, dlogits=dl, grad_scale=1.0 / M)
            loss = ops.reduce_sum(rows, 1.0 / M)
            assert abs(loss.item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
            assert rel(dl[:, :V], ld.grad) < (1e-6 if dt == torch.float32 else 5e-3)
            if Vp > V:
                assert torch.all(dl[:, V:] == 0)


@pytest.mark.parametrize("M,V,ldl,ldd,n", [(16384, 80, 128, 128, 256), (1000, 80, 80, 88, 7), (48, 7, 7, 8, 3), (256, 128, 128, 128, 2048)])
def test_cross_entropy_fused_loss_head(dev, M, V, ldl, ldd, n):
    """the one-launch loss head of the captured step (ref: F.cross_entropy at src/model.py:604-607 + the lm_head bias gradient):
    per-row losses, gradient rows, column-sum partials and the mean loss against fp64; the arrival counter is back at zero, so a
    second launch (new data) gives the new loss; launch after launch the loss is bit-identical"""
    ops = _ops()
    g = torch.Generator().manual_seed(M + V)
    for dt in (torch.bfloat16, torch.float32):
        scratch = torch.zeros(n + 1, device=dev)
        for trial in range(2):
            buf = torch.full((M, ldl), 99.0)
            buf[:, :V] = torch.randn(M, V, generator=g) * 3
            tgt = torch.randint(0, V, (M,), generator=g)
            x = buf[:, :V].double()
            ref_rows = torch.logsumexp(x, 1) - x[torch.arange(M), tgt]
            ref_grad = (torch.softmax(x, 1) - torch.nn.functional.one_hot(tgt, V)) / M
            dl = torch.full((M, ldd), float("nan"), dtype=dt, device=dev)
            part = torch.full((n, 200), float("nan"), device=dev)
            loss = torch.full((), float("nan"), device=dev)
            logits = buf.to(dev)[:, :V]
            assert ops.cross_entropy_fused_supported(logits, dl, n)
            rows = ops.cross_entropy_fused(logits, tgt.to(dev), V, dl, 1.0 / M, part[:, 3:], 200, n, scratch, loss, 1.0 / M)
            torch.cuda.synchronize()
            assert rel(rows, ref_rows) < 1e-6
            assert abs(loss.item() - ref_rows.mean().item()) < 1e-5 * max(1.0, ref_rows.mean().item())
            assert rel(dl[:, :V], ref_grad) < (1e-6 if dt == torch.float32 else 5e-3) and torch.all(dl[:, V:] == 0)
            assert rel(part[:, 3:3 + V].double().sum(0), ref_grad.sum(0)) < 1e-4 or ref_grad.sum(0).abs().max() < 1e-6
            assert torch.isnan(part[:, :3]).all() and torch.isnan(part[:, 3 + V:]).all()
            assert scratch[n].view(torch.int32).item() == 0
            loss2 = torch.zeros((), device=dev)
            ops.cross_entropy_fused(logits, tgt.to(dev), V, dl, 1.0 / M, None, 0, n, scratch, loss2, 1.0 / M)
            assert loss2.item() == loss.item()
    big = torch.zeros((8, 136), dtype=torch.bfloat16, device=dev)
    assert not ops.cross_entropy_fused_supported(torch.zeros((8, 130), device=dev), big, 4)
    with pytest.raises(RuntimeError):
        ops.cross_entropy_fused(torch.zeros((8, 130), device=dev), torch.zeros(8, dtype=torch.long, device=dev), 130, big, 1.0, None, 0, 4, None, None, 1.0)


def test_cross_entropy_bf16_logits_in_place(dev):
    """large-vocabulary form: bf16 logits [M, ld] overwritten in place by their bf16 gradient (the engine at V = 50257); loss
    and gradient against fp64 on the same bf16 logits, padding columns zeroed, fp32 logits still refuse to alias"""
    ops = _ops()
    M, V, ld = 64, 50257, 50304
    g = torch.Generator().manual_seed(8)
    buf = torch.zeros(M, ld, dtype=torch.bfloat16)
    buf[:, :V] = (torch.randn(M, V, generator=g) * 3).bfloat16()
    buf[:, V:] = 7.0                                               # stale padding must not survive
    tgt = torch.randint(0, V, (M,), generator=g)
    x = buf[:, :V].double()
    lse = torch.logsumexp(x, 1)
    ref_loss = lse - x[torch.arange(M), tgt]
    ref_grad = (torch.softmax(x, 1) - torch.nn.functional.one_hot(tgt, V)) / M
    d = buf.to(dev)
    rows = ops.cross_entropy(d[:, :V], tgt.to(dev), V, dlogits=d, grad_scale=1.0 / M)
    torch.cuda.synchronize()
    assert rel(rows, ref_loss) < 1e-6
    assert rel(d[:, :V], ref_grad) < 4e-3 and torch.all(d[:, V:] == 0)
    lf = torch.randn(8, V, generator=g).to(dev)
    with pytest.raises(RuntimeError):
        ops.cross_entropy(lf, tgt[:8].to(dev), V, dlogits=lf)


def test_embed(dev):
    ops = _ops()
    B, T, C, V = 5, 8, 32, 80
    g = torch.Generator().manual_seed(0)
    idx = torch.randint(0, V, (B, T), generator=g)
    tok = torch.randn(V, C, generator=g)
    pos = torch.randn(T + 3, C, generator=g)
    x = ops.embed_fwd(idx.to(dev), tok.to(dev), pos.to(dev))
    assert torch.equal(x.cpu(), tok[idx] + pos[:T])
    dx = torch.randn(B, T, C, generator=g)
    dtok = torch.empty(V, C, device=dev)
    dpos = torch.zeros(T + 3, C, device=dev)
    ops.embed_bwd(idx.to(dev), dx.to(dev), dtok, dpos[:T])
    ref_tok = torch.zeros(V, C, dtype=torch.float64).index_add_(0, idx.reshape(-1), dx.reshape(-1, C).double())
    assert rel(dtok, ref_tok) < 1e-6
    assert rel(dpos[:T], dx.double().sum(0)) < 1e-6 and torch.all(dpos[T:] == 0)
    with pytest.raises(IndexError):
        ops.embed_fwd(torch.zeros((1, T + 4), dtype=torch.long, device=dev), tok.to(dev), pos.to(dev))


def test_casts_and_colsum(dev):
    from oracle import rng_ref
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    W = torch.randn(80, 384, generator=g)
    assert torch.equal(ops.cast(W.to(dev), torch.bfloat16).cpu(), W.bfloat16())
    assert torch.equal(ops.cast(W.bfloat16().to(dev), torch.float32).cpu(), W.bfloat16().float())
    Wt = ops.transpose_cast(W.to(dev), torch.bfloat16)
    assert Wt.shape == (384, 80) and torch.equal(Wt.cpu(), W.T.bfloat16())
    W2 = torch.randn(50, 36, generator=g)
    Wt2 = ops.transpose_cast(W2.to(dev), torch.float32)       # R=50 -> ld 52, zero padded
    assert Wt2.shape == (36, 52) and torch.equal(Wt2[:, :50].cpu(), W2.T) and torch.all(Wt2[:, 50:] == 0)
    # dropout backward + cast + column sums
    M, N, p = 1000, 384, 0.2
    dy = torch.randn(M, N, generator=g)
    seed, step, site = 5, 0, 2
    rng = ops.new_rng_state(seed, dev, step)
    G = 31
    part = torch.empty(G, N, device=dev)
    gq = ops.dropout_bwd_cast(dy.to(dev), torch.float32, p, rng, site, colsum_part=part, part_stride=N, n_partials=G)
    keep = torch.from_numpy(rng_ref.keep_mask(seed, step, site, p, M * N).reshape(M, N))
    ref = dy * keep / (1 - p)
    assert rel(gq, ref) < 1e-6
    cs = torch.empty(N, device=dev)
    ops.reduce_partials(part, N, G, cs, N)
    assert rel(cs, ref.double().sum(0)) < 1e-5
    mask = torch.randn(M, N, generator=g)
    gm = ops.dropout_bwd_cast(dy.to(dev), torch.bfloat16, 0.0, None, 0, relu_mask=mask.to(dev))
    assert torch.equal(gm.cpu(), (dy * (mask > 0)).bfloat16())
    part2 = torch.empty(G, N, device=dev)
    ops.colsum(dy.bfloat16().to(dev), part2, N, G)
    ops.reduce_partials(part2, N, G, cs, N)
    assert rel(cs, dy.bfloat16().double().sum(0)) < 1e-5


@pytest.mark.parametrize("M,N,G,ld", [(16384, 1536, 256, 1536), (1000, 384, 31, 392), (517, 80, 16, 80), (64, 1032, 1, 1040), (300, 84, 7, 88), (256, 50257, 8, 50304), (100, 33, 3, 35)])
def test_colsum_bf16(dev, M, N, G, ld):
    """the 16-byte-load kernel (leading dimension a multiple of 8; N may be ragged: 84, 50257) and the generic one (ld = 35),
    ragged row chunks, padded leading dimension"""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    A = torch.zeros(M, ld, dtype=torch.bfloat16)
    A[:, :N] = torch.randn(M, N, generator=g).bfloat16()
    part = torch.full((G, N), float("nan"), device=dev)
    ops.colsum(A.to(dev)[:, :N], part, N, G)
    cs = torch.empty(N, device=dev)
    ops.reduce_partials(part, N, G, cs, N)
    assert rel(cs, A[:, :N].double().sum(0)) < 1e-5


@pytest.mark.parametrize("n,advance", [(10007, False), (10007, True), (3_000_001, True)])
def test_adamw_matches_oracle(dev, n, advance):
    """advance: the step word moves on inside the AdamW launch (last workgroup to arrive), as dg_state_advance does"""
    from oracle import drake_ref as R
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(n, generator=g)
    params = {"w": p0.clone()}
    opt = R.AdamWState(["w"], 1e-3, (0.9, 0.95))
    pd = p0.clone().to(dev)
    pad = (n + 3) // 4 * 4
    m = torch.zeros(pad, device=dev)[:n]
    v = torch.zeros(pad, device=dev)[:n]
    hyper = torch.tensor([1e-3, 0.9, 0.95, 1e-8, 1e-2], device=dev)
    st = ops.new_rng_state(0, dev, 0)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for it in range(4):
        gr = torch.randn(n, generator=g)
        opt.step(params, {"w": gr})
        ops.adamw_step(pd, gr.to(dev), m, v, hyper, st, shadow_bf16=shadow, advance=advance)
        if not advance:
            ops.state_advance(st)
        assert st.tolist()[2:] == [it + 1, 0]
    assert maxabs(pd, params["w"]) < 2e-6
    assert torch.equal(shadow.cpu(), pd.cpu().bfloat16())


def test_batch_gather_and_softmax(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    data = torch.randint(0, 80, (5000,), generator=g)
    off = torch.randint(5000 - 8, (32,), generator=g)
    x, y = ops.batch_gather(data.to(dev), off.to(dev), 8)
    assert torch.equal(x.cpu(), torch.stack([data[i:i + 8] for i in off]))
    assert torch.equal(y.cpu(), torch.stack([data[i + 1:i + 9] for i in off]))
    logits = torch.randn(3, 5, 80, generator=g).to(dev)
    pr = ops.softmax_rows(logits[:, -1, :])
    assert rel(pr, logits[:, -1, :].double().softmax(-1)) < 1e-6


def test_batch_embed_fwd_is_gather_plus_embedding(dev):
    """get_batch + embedding in one launch (ref: src/preprocessing.py:43-45, src/model.py:595-597): the row of the staged
    offset block comes from the step word, clamped to the block; ids, targets, stream and one-hot rows as the two kernels give them"""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B, T, C, V = 6, 8, 64, 80                       # (the one-hot rows are written by the C / 4 work items of a row, 8 columns each)
    data = torch.randint(0, V, (3000,), generator=g)
    block = torch.randint(3000 - T, (5, B), generator=g)
    tok, pos = torch.randn(V, C, generator=g), torch.randn(T, C, generator=g)
    for step, base, n_rows, want_row in ((7, 7, 5, 0), (9, 7, 5, 2), (11, 7, 5, 4), (40, 7, 5, 4), (3, 7, 5, 4), (9, 7, 2, 1), (None, None, 5, 0)):
        st = ops.new_rng_state(123, dev, step) if step is not None else None
        ctl = torch.tensor([base, n_rows], dtype=torch.int32, device=dev) if step is not None else None
        x_ids = torch.full((B, T), -1, dtype=torch.int64, device=dev)
        y_ids = torch.full((B, T), -1, dtype=torch.int64, device=dev)
        onehot = torch.full((B * T, 88), 7.0, dtype=torch.bfloat16, device=dev)
        out = ops.batch_embed_fwd(data.to(dev), block.to(dev), st, ctl, x_ids, y_ids, tok.to(dev), pos.to(dev), onehot=onehot)
        off = block[want_row]
        xs = torch.stack([data[i:i + T] for i in off]); ys = torch.stack([data[i + 1:i + T + 1] for i in off])
        assert torch.equal(x_ids.cpu(), xs) and torch.equal(y_ids.cpu(), ys), (step, base, n_rows)
        assert torch.equal(out.cpu(), tok[xs] + pos)
        oh = torch.zeros(B * T, 88); oh[torch.arange(B * T), xs.reshape(-1)] = 1.0
        assert torch.equal(onehot.float().cpu(), oh)
    with pytest.raises(ValueError):
        ops.batch_embed_fwd(data.to(dev), block[0].to(dev), None, None, x_ids, y_ids, tok.to(dev), pos.to(dev))


def test_transpose_cast_batched(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    shapes = [(80, 384), (1152, 384), (384, 1536), (50, 36)]
    for dt in (torch.bfloat16, torch.float32):
        gr = 8 if dt == torch.bfloat16 else 4
        pairs = []
        for R, C in shapes:
            W = torch.randn(R, C, generator=g).to(dev)
            Wt = torch.full((C, (R + gr - 1) // gr * gr), float("nan"), dtype=dt, device=dev)
            pairs.append((W, Wt))
        table = ops.make_transpose_table(pairs, dev)
        ops.transpose_cast_batched(*table, dt)
        for W, Wt in pairs:
            R = W.shape[0]
            assert torch.equal(Wt[:, :R].cpu(), W.T.to(dt).cpu())
            assert torch.all(Wt[:, R:] == 0)
    # bf16 -> bf16 (the engine's W^T refresh from the bf16 shadow): 16-byte pieces, ragged edges, an unaligned descriptor
    bf = torch.bfloat16
    pairs = []
    for R, C in shapes + [(200, 72), (64, 64)]:
        W = torch.randn(R, C, generator=g).to(bf).to(dev)
        Wt = torch.full((C, (R + 7) // 8 * 8), float("nan"), dtype=bf, device=dev)
        pairs.append((W, Wt))
    table = ops.make_transpose_table(pairs, dev)
    ops.transpose_cast_batched(*table, bf, in_dtype=bf)
    for W, Wt in pairs:
        R = W.shape[0]
        assert torch.equal(Wt[:, :R].cpu(), W.T.cpu())
        assert torch.all(Wt[:, R:] == 0)


def test_argument_validation_rejects_before_launch(dev):
    """bad shapes / alignment / dtypes come back as DG_ERR_* from the C ABI (no kernel is launched)."""
    import ctypes as C
    from drakegpt_amd import _lib
    ops = _ops()
    lib = _lib.lib
    x = torch.zeros(64, 40, device=dev)
    a = _lib.GemmNtArgs()
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = x.data_ptr(), 40, x.data_ptr(), 40, x.data_ptr(), 64
    a.M, a.N, a.K, a.in_dtype, a.out_dtype = 64, 64, 38, _lib.DG_F32, _lib.DG_F32
    assert lib.dg_gemm_nt(C.byref(a), None) == -2                    # K not a multiple of 4 floats: DG_ERR_ALIGN
    a.K, a.in_dtype = 40, 7
    assert lib.dg_gemm_nt(C.byref(a), None) == -3                    # DG_ERR_DTYPE
    a.in_dtype, a.M = _lib.DG_F32, 0
    assert lib.dg_gemm_nt(C.byref(a), None) == -1                    # DG_ERR_ARG
    assert lib.dg_attn_fwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 8, 1, 300, 1.0, 0.0, None, 0, _lib.DG_F32, None, 0, None) == -1
    assert lib.dg_layernorm_fwd(None, x.data_ptr(), x.data_ptr(), x.data_ptr(), 0, x.data_ptr(), x.data_ptr(), 4, 40, 1e-5, None) == -1
    with pytest.raises(RuntimeError, match="invalid argument"):
        _lib.check(-1, "probe")
    with pytest.raises(TypeError):
        ops.gemm_nt(x, x.bfloat16(), torch.float32)                  # operand dtypes differ
    with pytest.raises(RuntimeError, match="GPU"):
        ops.cast(torch.zeros(4), torch.bfloat16)                     # CPU tensor: no fallback


def test_engine_minimal_shapes(dev):
    """B = 1, T = 1 (softmax over one key, single-row GEMMs) through the graph-captured engine, both precisions."""
    import drakegpt_amd as D
    from drakegpt_amd.engine import TrainEngine
    for precision in ("fp32", "bf16"):
        torch.manual_seed(1)
        m = D.TransformerLM(80, 64, 8, 1, 1, 0.5, precision=precision).to(dev)
        eng = TrainEngine(m, 1, 1, lr=1e-3)
        eng.set_batch(torch.tensor([[3]], device=dev), torch.tensor([[5]], device=dev))
        l0 = eng.step().item()
        l1 = eng.step().item()
        assert l0 > 0 and l1 > 0 and l0 == l0 and l1 == l1


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("M,p", [(64, 0.0), (4160, 0.2), (16384, 0.2)])
def test_block_chain_kernel_equals_the_launches_it_replaces(dev, mode, M, p):
    """dg_block_chain_fwd (round 3): the row-local chain of a residual block -- proj + residual, LayerNorm 2, FFN1 (ReLU, sign
    bits), FFN2 + residual, the next block's LayerNorm 1 and packed QKV (ref: src/model_component.py:454,505-506,320-325,392-393,
    404) -- in one persistent launch (modes 0 / 1 / 2) or as the two row-complete GEMMs with a LayerNorm in the epilogue (modes
    3 / 4), against dg_gemm_nt + dg_layernorm_fwd on the same operands: the GEMM outputs, the dropout masks and the ReLU sign
    bits are IDENTICAL (same MFMA sequence per element, same hash), the row statistics differ by an fp32 rounding (four 96-column
    partials combined with Chan's formula instead of one wave-wide sum), hence an occasional bf16 ulp in the LayerNorm outputs
    and what follows them.  One block (M = 64), several blocks per workgroup (M = 16384 > 256 x 64 is not: 4160 = 65 blocks on
    256 CUs is), every mode."""
    ops = _ops()
    from drakegpt_amd import sublayers as S
    C = 384
    assert ops.block_chain_supported(M, C, torch.bfloat16) and not ops.block_chain_supported(M + 8, C, torch.bfloat16)
    assert not ops.block_chain_supported(M, 768, torch.bfloat16)
    g = torch.Generator().manual_seed(M + mode)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    bf = torch.bfloat16
    o = rn(M, C).to(bf).to(dev)
    x = rn(M, C, sc=2.0).to(dev)
    f_in = rn(M, 4 * C).clamp_min(0).to(bf).to(dev)
    W = {k: v.to(bf).to(dev) for k, v in dict(wproj=rn(C, C, sc=C ** -0.5), w1=rn(4 * C, C, sc=C ** -0.5), w2=rn(C, 4 * C, sc=(4 * C) ** -0.5),
                                           wqkv=rn(3 * C, C, sc=C ** -0.5)).items()}
    Vv = {k: v.to(dev) for k, v in dict(bproj=rn(C, sc=0.1), b1=rn(4 * C, sc=0.1), b2=rn(C, sc=0.1), ln2w=1 + rn(C, sc=0.1), ln2b=rn(C, sc=0.1),
                                        ln1w=1 + rn(C, sc=0.1), ln1b=rn(C, sc=0.1)).items()}
    rng = ops.new_rng_state(99, dev, 5) if p > 0 else None
    sp, sf = S.site_proj(2), S.site_ffn(2)
    Wp = {k: ops.pack_chain_weights(v) for k, v in W.items()}
    # packing is a permutation of 16-byte chunks: same multiset of values, different order
    assert torch.equal(Wp["w1"].view(-1).float().sort().values, W["w1"].view(-1).float().sort().values) and not torch.equal(Wp["w1"], W["w1"])
    r = {}
    if mode in (0, 1, 3):
        r["x1"] = ops.gemm_nt(o, W["wproj"], torch.float32, bias=Vv["bproj"], dropout_p=p, rng_state=rng, site=sp, residual=x)
        r["h2"], r["mean2"], r["rstd2"] = ops.layernorm_fwd(r["x1"], Vv["ln2w"], Vv["ln2b"], bf)
    if mode in (0, 1):
        r["bits"] = ops.new_sign_bits(M, 4 * C, dev)
        r["f"] = ops.gemm_nt(r["h2"], W["w1"], bf, bias=Vv["b1"], relu=True, sign_bits_out=r["bits"])
    if mode in (0, 1, 4):
        fin, x1in = (f_in, x) if mode == 4 else (r["f"], r["x1"])
        r["x2"] = ops.gemm_nt(fin, W["w2"], bf if mode == 1 else torch.float32, bias=Vv["b2"], dropout_p=p, rng_state=rng, site=sf, residual=x1in)
    if mode in (0, 2, 4):
        r["h1"], r["mean1"], r["rstd1"] = ops.layernorm_fwd(x if mode == 2 else r["x2"], Vv["ln1w"], Vv["ln1b"], bf)
    if mode in (0, 2):
        r["qkv"] = ops.gemm_nt(r["h1"], W["wqkv"], bf)
    if mode == 4:
        got = ops.block_chain_fwd(4, M, C, f=f_in, x1=x, dropout_p=p, rng_state=rng, site_proj=sp, site_ffn=sf, **Wp, **Vv)
    else:
        got = ops.block_chain_fwd(mode, M, C, o=o, x=x, dropout_p=p, rng_state=rng, site_proj=sp, site_ffn=sf, **Wp, **Vv)
    torch.cuda.synchronize()
    assert sorted(got) == sorted(r)
    exact = {"x1"} | ({"x2"} if mode == 4 else set())          # in front of every LayerNorm of the launch: bit for bit
    for k, ref in r.items():
        if k == "bits":
            # bytes of rows beyond M (the second half of a last, half-filled 128-row tile) are never written by either side
            idx = torch.arange(ref.numel(), device=dev)
            row = (idx // 3072 // 8) * 128 + ((idx // 384) % 8 >> 1) * 32 + ((idx // 64) % 2) * 16 + (idx % 16)
            ok = row < M
            assert int((got[k][ok] != ref[ok]).sum()) <= max(2, ref.numel() // 100000), k      # a bf16 ulp of h2 can move a pre-activation across zero
        elif k in exact:
            assert torch.equal(got[k], ref), k
        else:
            e = rel(got[k].float(), ref.float())
            # (mode 0: the second LayerNorm's input already carries the first one's bf16-ulp differences through FFN1 / FFN2)
            stat_tol = 1e-4 if (mode == 0 and k in ("mean1", "rstd1")) else 2e-6
            assert e < (stat_tol if k.startswith(("mean", "rstd")) else 3e-4), (k, e)
    with pytest.raises(RuntimeError):                          # operands are validated before the launch
        ops.block_chain_fwd(3, M, C, o=o, x=x[: M - 1], wproj=Wp["wproj"], bproj=Vv["bproj"], ln2w=Vv["ln2w"], ln2b=Vv["ln2b"])


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("M,p", [(64, 0.0), (4160, 0.2), (16384, 0.2)])
def test_block_chain_bwd_kernel_equals_the_launches_it_replaces(dev, mode, M, p):
    """dg_block_chain_bwd (round 3): the backward pass's row-local chain between two attention-backward calls -- dX of the packed
    q / k / v Linears + LayerNorm-1 backward + the dropout backward of the block below, dX of the second FFN Linear through the ReLU
    sign bits, dX of the first, LayerNorm-2 backward + proj's dropout backward, dX of proj (autograd through
    src/model_component.py:392-393,404,505,324,322-323,321,506,454) -- in one persistent launch, against dg_gemm_nt +
    dg_layernorm_bwd_fused on the same operands.  The kernel rounds the dX GEMM's accumulators to bf16 in front of the LayerNorm
    backward exactly as the separate launch does, so the statistics / dgamma / dbeta agree to summation order; x-hat crosses the
    row-sum exchange as bf16 (registers), which shows as an occasional bf16 ulp in dx / g.  Modes 0 (both halves), 1 (top of the
    stack), 2 (block 0: no dropout, no bias partial)."""
    ops = _ops()
    from drakegpt_amd import sublayers as S
    C = 384
    assert ops.block_chain_bwd_supported(M, C, torch.bfloat16) and not ops.block_chain_bwd_supported(M + 8, C, torch.bfloat16)
    assert not ops.block_chain_bwd_supported(M, 768, torch.bfloat16) and not ops.block_chain_bwd_supported(64 * 4096, C, torch.bfloat16)
    gen = torch.Generator().manual_seed(7 * M + mode)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=gen) * sc
    bf, f32 = torch.bfloat16, torch.float32
    has_q, has_2 = mode in (0, 2), mode in (0, 1)
    WT = {k: v.to(bf).to(dev) for k, v in dict(wqkvT=rn(C, 3 * C, sc=(3 * C) ** -0.5), w2T=rn(4 * C, C, sc=C ** -0.5),
                                            w1T=rn(C, 4 * C, sc=(4 * C) ** -0.5), wprojT=rn(C, C, sc=C ** -0.5)).items()}
    WTp = {k: ops.pack_chain_weights(v) for k, v in WT.items()}
    dqkv = rn(M, 3 * C).to(bf).to(dev)
    x, x1 = rn(M, C, sc=2.0).to(dev), rn(M, C, sc=2.0).to(dev)
    ln1w, ln2w = (1 + rn(C, sc=0.1)).to(dev), (1 + rn(C, sc=0.1)).to(dev)
    _, mean1, rstd1 = ops.layernorm_fwd(x, ln1w, torch.zeros_like(ln1w), bf)
    h2, mean2, rstd2 = ops.layernorm_fwd(x1, ln2w, torch.zeros_like(ln2w), bf)
    dresid, g_in = rn(M, C).to(bf).to(dev), rn(M, C).to(bf).to(dev)
    bits = ops.new_sign_bits(M, 4 * C, dev)
    ops.gemm_nt(h2, rn(4 * C, C, sc=C ** -0.5).to(bf).to(dev), bf, relu=True, sign_bits_out=bits)
    rng = ops.new_rng_state(4321, dev, 5) if p > 0 else None
    s_ffn, s_proj = S.site_ffn(2), S.site_proj(2)
    G = S.n_partials_for(M)
    rows_cs = ops.gemm_nt_colsum_rows(bf, M, 4 * C, C)
    stride = 12 * C
    Pr = torch.zeros((max(G, rows_cs, 1), stride), dtype=f32, device=dev)
    Pc = torch.zeros((2 * (M // 64), stride), dtype=f32, device=dev)
    r = {}
    if has_q:
        dh = ops.gemm_nt(dqkv, WT["wqkvT"], bf, K=3 * C)
        r["dx1"], r["g1"] = ops.layernorm_bwd_fused(dh, x, ln1w, mean1, rstd1, dresid, Pr[0, 0:C], Pr[0, C:2 * C], stride, G, bf, p if mode == 0 else 0.0,
                                                    rng, s_ffn, Pr[0, 2 * C:3 * C] if mode == 0 else None, stream_dtype=bf)
    if has_2:
        g = r["g1"] if mode == 0 else g_in
        if rows_cs:
            r["df"] = ops.gemm_nt(g, WT["w2T"], bf, K=C, sign_bits=bits, colsum_part=Pr[:rows_cs, 8 * C:12 * C])
        else:
            r["df"] = ops.gemm_nt(g, WT["w2T"], bf, K=C, sign_bits=bits)
            ops.colsum(r["df"], Pr[0, 8 * C:12 * C], stride, G)
        dh2 = ops.gemm_nt(r["df"], WT["w1T"], bf, K=4 * C)
        r["dx2"], r["g2"] = ops.layernorm_bwd_fused(dh2, x1, ln2w, mean2, rstd2, r["dx1"] if mode == 0 else dresid, Pr[0, 3 * C:4 * C], Pr[0, 4 * C:5 * C],
                                                    stride, G, bf, p, rng, s_proj, Pr[0, 5 * C:6 * C], stream_dtype=bf)
        r["dout"] = ops.gemm_nt(r["g2"], WT["wprojT"], bf, K=C)
    kw = dict(part_stride=stride, dropout_p=p, rng_state=rng, site_ffn_below=s_ffn, site_proj=s_proj)
    if has_q:
        kw.update(dqkv=dqkv, wqkvT=WTp["wqkvT"], x=x, mean1=mean1, rstd1=rstd1, ln1w=ln1w, dresid1=dresid, dln1w_part=Pc[0, 0:C], dln1b_part=Pc[0, C:2 * C],
                  gbias1_part=Pc[0, 2 * C:3 * C] if mode == 0 else None)
    if has_2:
        kw.update(w2T=WTp["w2T"], bits=bits, db1_part=Pc[0, 8 * C:12 * C], w1T=WTp["w1T"], x1=x1, mean2=mean2, rstd2=rstd2, ln2w=ln2w,
                  dln2w_part=Pc[0, 3 * C:4 * C], dln2b_part=Pc[0, 4 * C:5 * C], gbias2_part=Pc[0, 5 * C:6 * C], wprojT=WTp["wprojT"])
        if mode == 1:
            kw.update(g_in=g_in, dresid2=dresid)
    got = ops.block_chain_bwd(mode, M, C, **kw)
    torch.cuda.synchronize()
    assert sorted(got) == sorted(r)
    # measured (M = 16384, p = 0.2): first half 3e-4, second half of mode 0 (it inherits the first half's ulps through two GEMMs) 2e-3
    for k, ref in r.items():
        if mode == 1 and k == "df":
            assert torch.equal(got[k], ref), k                  # same operands, same MFMA sequence, same mask bits: bit for bit
            continue
        e = rel(got[k].float(), ref.float())
        assert e < (4e-3 if (mode == 0 and k in ("df", "dx2", "g2", "dout")) else 1.5e-3), (k, e)
    sr, sc = Pr.sum(0), Pc.sum(0)
    for i, nm in enumerate(["dln1w", "dln1b", "gbias1", "dln2w", "dln2b", "gbias2"]):
        a, b = sc[i * C:(i + 1) * C], sr[i * C:(i + 1) * C]
        if float(b.abs().sum()) == 0.0:
            assert float(a.abs().sum()) == 0.0, nm
            continue
        first = i < 3 or mode == 1
        assert rel(a, b) < (2e-4 if first else 6e-3), (nm, rel(a, b))
    if has_2:
        # (the GEMM epilogue sums the fp32 values, the stand-alone column-sum kernel of the small shapes the bf16-rounded ones)
        assert rel(sc[8 * C:12 * C], sr[8 * C:12 * C]) < (1e-5 if (mode == 1 and rows_cs) else 3e-3)
    with pytest.raises(RuntimeError):                          # operands are validated before the launch
        bad = dict(kw)
        if has_q:
            bad["x"] = x[: M - 1]
        else:
            bad["x1"] = x1[: M - 1]
        ops.block_chain_bwd(mode, M, C, **bad)
