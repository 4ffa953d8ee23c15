"""GPU parity of the individual C-ABI ops against the oracle / a torch fp32 reference."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import score_oracle as so
from score_amd import _lib

pytestmark = pytest.mark.gpu


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def dev(a, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_gather_bit_exact():
    # tf.nn.embedding_lookup (score.py:51-66): pure row copy -> bit-exact
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for N, D, n in ((1000, 16, 5000), (50000, 64, 70001), (300, 4, 17), (4096, 128, 999)):
        table = rng.standard_normal((N, D)).astype(np.float32)
        idx = rng.integers(0, N, n).astype(np.int32)
        idx[:3] = [0, N - 1, 0]
        t, i = dev(table), dev(idx)
        out = torch.empty((n, D), dtype=torch.float32, device="cuda")
        _lib.check(lib.score_gather_fwd(P(t), N, D, P(i), n, P(out), stream()), "gather")
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), table[idx])
    # empty input is a no-op
    assert lib.score_gather_fwd(P(t), N, D, P(i), 0, P(out), stream()) == 0
    # error behaviour: bad D, null pointer
    assert lib.score_gather_fwd(P(t), N, 6, P(i), 4, P(out), stream()) == -2
    assert lib.score_gather_fwd(C.c_void_p(0), N, 4, P(i), 4, P(out), stream()) == -1


@pytest.mark.parametrize("trans", [0, 1, 2])
@pytest.mark.parametrize("shape", [(64, 64, 16), (130, 70, 37), (2048, 384, 448), (5, 1, 80), (200, 80, 1000),
                                   (1184, 80, 4096),
                                   # the dense layers of the reference's own shapes (K = 80 .. 256 on small grids), odd K
                                   (256, 200, 80), (2560, 80, 168), (1800, 96, 112), (200, 176, 200), (300, 70, 255),
                                   (130, 130, 256), (77, 33, 129), (2000, 80, 208), (64, 64, 1)])
def test_gemm_fp32(trans, shape):
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K + trans)
    a = torch.randn((M, K), device="cuda", generator=g)
    b = torch.randn((K, N), device="cuda", generator=g)
    bias = torch.randn((N,), device="cuda", generator=g)
    A = a if trans != 2 else a.t().contiguous()          # stored [K,M]
    Bm = b if trans != 1 else b.t().contiguous()         # stored [N,K]
    lda = A.shape[1]
    ldb = Bm.shape[1]
    scratch = torch.empty((1 << 22,), device="cuda")
    ref = (a.double() @ b.double())
    tol = 2e-6 * (a.abs().double() @ b.abs().double()) + 1e-6
    for flags in (0, 1 | 2, 4):
        c0 = torch.randn((M, N + 3), device="cuda", generator=g)
        c = c0.clone()
        _lib.check(lib.score_gemm(trans, M, N, K, P(A), lda, P(Bm), ldb, P(c), N + 3, P(bias), flags, 1.0,
                                  C.c_void_p(0), 0, P(scratch), scratch.numel(), stream()), "gemm")
        torch.cuda.synchronize()
        want = ref.clone()
        if flags & 1:
            want = want + bias.double()
        if flags & 2:
            want = want.clamp_min(0)
        if flags & 4:
            want = want + c0[:, :N].double()
        assert torch.equal(c[:, N:], c0[:, N:]), "wrote outside the N columns"
        err = (c[:, :N].double() - want).abs()
        assert bool((err <= tol).all()), float((err / tol).max())


@pytest.mark.parametrize("trans", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 32), (200, 76, 100), (2048, 384, 448), (4096, 80, 1184), (448, 256, 4096), (20480, 384, 64),
                                   (64, 32, 36)])
def test_gemm_bf16x3_is_fp32_accurate(trans, shape):
    # "bf16x3": operands split exactly into three bf16, six bf16 MFMAs per k-step; must meet the SAME
    # error bound as the f32-MFMA kernel (relative to sum |a||b|)
    lib = _lib.load()
    M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(M + N + K + trans)
    a = torch.randn((M, K), device="cuda", generator=g) * torch.logspace(-3, 3, K, device="cuda")   # wide dynamic range
    b = torch.randn((K, N), device="cuda", generator=g)
    bias = torch.randn((N,), device="cuda", generator=g)
    A = a if trans != 2 else a.t().contiguous()
    Bm = b if trans != 1 else b.t().contiguous()
    scratch = torch.empty((1 << 22,), device="cuda")
    ref = a.double() @ b.double()
    tol = 2e-6 * (a.abs().double() @ b.abs().double()) + 1e-6
    for flags in (32, 32 | 1 | 2, 32 | 4):
        c0 = torch.randn((M, N + 4), device="cuda", generator=g)
        c = c0.clone()
        _lib.check(lib.score_gemm(trans, M, N, K, P(A), A.shape[1], P(Bm), Bm.shape[1], P(c), N + 4, P(bias), flags,
                                  1.0, C.c_void_p(0), 0, P(scratch), scratch.numel(), stream()), "gemm")
        torch.cuda.synchronize()
        want = ref.clone()
        if flags & 1:
            want = want + bias.double()
        if flags & 2:
            want = want.clamp_min(0)
        if flags & 4:
            want = want + c0[:, :N].double()
        assert torch.equal(c[:, N:], c0[:, N:])
        err = (c[:, :N].double() - want).abs()
        assert bool((err <= tol).all()), float((err / tol).max())


@pytest.mark.parametrize("trans_b,G,M,N,K,with_bias", [
    (0, 2, 18432, 384, 448, True),      # cfg-3's GRU input projections (18 active slices): 256 panels of 144 rows
    (1, 2, 18432, 448, 384, False),     # their input gradients: N = 448 leaves the eighth wave's columns as padding
    (0, 2, 20480, 384, 448, True),      # all 20 slices: 160-row panels
    (1, 2, 19000, 448, 384, False),     # a ragged last panel (rows past M are copies of row M - 1)
    (0, 1, 33000, 320, 64, True),       # one group, N not a multiple of 128, two k-tiles
    (1, 3, 12345, 512, 96, True),       # three groups, the widest N, odd number of k-tiles
])
def test_gemm_panel_products_match_fp64_within_the_fp32_bound(trans_b, G, M, N, K, with_bias):
    # csrc/gemm_panel.hip: whole-N output panels, weights as MFMA-fragment images; same bound as the tiled bf16x3 kernel
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K + trans_b)
    abuf = [torch.randn((M, K + 4), device="cuda", generator=g) for _ in range(G)]      # rows K + 4 apart: lda > K
    for x in abuf:
        x[:, :K] *= torch.logspace(-3, 3, K, device="cuda")                               # wide dynamic range
    a = [x[:, :K] for x in abuf]
    b = [torch.randn((K, N), device="cuda", generator=g) for _ in range(G)]
    bias = [torch.randn((N,), device="cuda", generator=g) for _ in range(G)]
    Bm = [x if trans_b == 0 else x.t().contiguous() for x in b]
    c0 = [torch.randn((M, N + 4), device="cuda", generator=g) for _ in range(G)]
    c = [x.clone() for x in c0]
    images = torch.empty((G * (K // 32) * 8 * ((N + 127) // 128) * 768,), device="cuda")
    arr = lambda ts: (C.c_void_p * G)(*[t.data_ptr() for t in ts])
    rc = lib.score_gemm_panel_products(trans_b, G, M, N, K, arr(a), K + 4, arr(Bm), Bm[0].shape[1], arr(c), N + 4,
                                       arr(bias) if with_bias else None, P(images), images.numel(), stream())
    _lib.check(rc, "gemm_panel_products")
    torch.cuda.synchronize()
    for i in range(G):
        want = a[i].double() @ b[i].double()
        if with_bias:
            want = want + bias[i].double()
        tol = 2e-6 * (a[i].abs().double() @ b[i].abs().double()) + 1e-6
        assert torch.equal(c[i][:, N:], c0[i][:, N:])          # nothing written past column N
        err = (c[i][:, :N].double() - want).abs()
        assert bool((err <= tol).all()), (i, float((err / tol).max()))


@pytest.mark.parametrize("trans_b,M,Nh,K,with_bias", [
    (0, 24576, 384, 384, True),       # cfg-5 Taobao-shaped projections at B = 512: 3H = 768 columns as two halves, I = 384
    (0, 24576, 384, 896, True),       # cfg-5 Tmall-shaped projections: I = 896 (28 k-tiles)
    (1, 24576, 448, 768, False),      # cfg-5 Tmall-shaped input gradients: I = 896 output columns as two halves of 448, K = 3H
    (1, 24000, 384, 768, True),       # ragged last panel, transposed weights inside a wider matrix
])
def test_gemm_panel_column_halves_of_one_matrix_match_fp64(trans_b, M, Nh, K, with_bias):
    """The H = 256 form of csrc/engine.hip: each side's weights are ONE matrix whose output columns are split into two panel
    groups -- B_h = a column range of a [K, 2Nh] matrix (trans_b 0: ldb = 2Nh > N) or a row range of a [2Nh, K] one inside a
    wider allocation (trans_b 1: ldb > K) -- and both halves write column ranges of ONE C (ldc = 2Nh + 4, column offset Nh),
    with the matching halves of one bias row.  Two sides = four groups, as the engine launches them."""
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + Nh + K + trans_b)
    G, N2 = 4, 2 * Nh
    a = [torch.randn((M, K), device="cuda", generator=g) * torch.logspace(-2, 2, K, device="cuda") for _ in range(2)]
    if trans_b == 0:
        wbuf = [torch.randn((K, N2), device="cuda", generator=g) for _ in range(2)]        # C = A . W
        Bh = [wbuf[s_][:, h * Nh:] for s_ in range(2) for h in range(2)]                   # views: column offset, ldb = N2
        ldb = N2
        wfull = wbuf
    else:
        wbuf = [torch.randn((N2, K + 8), device="cuda", generator=g) for _ in range(2)]    # C = A . W^T, rows K + 8 apart
        Bh = [wbuf[s_][h * Nh:, :K] for s_ in range(2) for h in range(2)]
        ldb = K + 8
        wfull = [x[:, :K].t() for x in wbuf]
    bias = [torch.randn((N2,), device="cuda", generator=g) for _ in range(2)]
    c0 = [torch.randn((M, N2 + 4), device="cuda", generator=g) for _ in range(2)]
    c = [x.clone() for x in c0]
    Ah = [a[s_] for s_ in range(2) for h in range(2)]
    Ch = [c[s_][:, h * Nh:] for s_ in range(2) for h in range(2)]
    bh = [bias[s_][h * Nh:] for s_ in range(2) for h in range(2)]
    images = torch.empty((G * (K // 32) * 8 * ((Nh + 127) // 128) * 768,), device="cuda")
    arr = lambda ts: (C.c_void_p * G)(*[t.data_ptr() for t in ts])
    rc = lib.score_gemm_panel_products(trans_b, G, M, Nh, K, arr(Ah), K, arr(Bh), ldb, arr(Ch), N2 + 4,
                                       arr(bh) if with_bias else None, P(images), images.numel(), stream())
    _lib.check(rc, "gemm_panel_products")
    torch.cuda.synchronize()
    for s_ in range(2):
        want = a[s_].double() @ wfull[s_].double()
        if with_bias:
            want = want + bias[s_].double()
        tol = 2e-6 * (a[s_].abs().double() @ wfull[s_].abs().double()) + 1e-6
        assert torch.equal(c[s_][:, N2:], c0[s_][:, N2:])          # nothing written past column 2 Nh
        err = (c[s_][:, :N2].double() - want).abs()
        assert bool((err <= tol).all()), (s_, float((err / tol).max()))


def test_gemm_panel_products_refuse_what_they_do_not_cover():
    lib = _lib.load()
    a = torch.zeros((64, 64), device="cuda"); b = torch.zeros((64, 384), device="cuda"); c = torch.zeros((64, 384), device="cuda")
    img = torch.zeros((1 << 20,), device="cuda")
    one = lambda t: (C.c_void_p * 1)(t.data_ptr())
    # too few rows to fill the chip, N too small, K not a multiple of 32: SCORE_E_SHAPE (-2), nothing launched
    assert lib.score_gemm_panel_products(0, 1, 64, 384, 64, one(a), 64, one(b), 384, one(c), 384, None, P(img), img.numel(), stream()) == -2
    big = torch.zeros((40000, 64), device="cuda"); cb = torch.zeros((40000, 384), device="cuda")
    assert lib.score_gemm_panel_products(0, 1, 40000, 128, 64, one(big), 64, one(b), 384, one(cb), 384, None, P(img), img.numel(), stream()) == -2
    assert lib.score_gemm_panel_products(0, 1, 40000, 384, 48, one(big), 64, one(b), 384, one(cb), 384, None, P(img), img.numel(), stream()) == -2
    # scratch for the images too small: SCORE_E_WORKSPACE
    assert lib.score_gemm_panel_products(0, 1, 40000, 384, 64, one(big), 64, one(b), 384, one(cb), 384, None, P(img), 100, stream()) == -3
    torch.cuda.synchronize()


@pytest.mark.parametrize("force", [0, 32])
def test_gemm_row_grouped_bias(force):
    # flags bits 16+: bias row group g -> output row r adds bias[r // g, :]  (folded attention layer)
    lib = _lib.load()
    M, N, K, grp = 20 * 37, 80, 72, 20
    g = torch.Generator(device="cuda").manual_seed(11)
    a, b = torch.randn((M, K), device="cuda", generator=g), torch.randn((K, N), device="cuda", generator=g)
    bias = torch.randn((M // grp, N), device="cuda", generator=g)
    c = torch.empty((M, N), device="cuda")
    scratch = torch.empty((1 << 20,), device="cuda")
    _lib.check(lib.score_gemm(0, M, N, K, P(a), K, P(b), N, P(c), N, P(bias), force | 1 | 2 | (grp << 16), 1.0,
                              C.c_void_p(0), 0, P(scratch), scratch.numel(), stream()), "gemm")
    want = torch.relu(a.double() @ b.double() + bias.double().repeat_interleave(grp, dim=0))
    assert float((c.double() - want).abs().max()) < 1e-4


def test_gemm_dropout_epilogue():
    lib = _lib.load()
    M, N, K = 128, 200, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    a, b = torch.randn((M, K), device="cuda", generator=g), torch.randn((K, N), device="cuda", generator=g)
    bias = torch.zeros((N,), device="cuda")
    mask = (torch.rand((M, N), device="cuda", generator=g) < 0.8).to(torch.uint8)
    c = torch.empty((M, N), device="cuda")
    _lib.check(lib.score_gemm(0, M, N, K, P(a), K, P(b), N, P(c), N, P(bias), 1 | 2 | 8, 0.8, P(mask), 0,
                              C.c_void_p(0), 0, stream()), "gemm")
    want = torch.relu(a @ b) / 0.8 * mask
    assert torch.allclose(c, want, rtol=1e-5, atol=1e-5)
    # hashed mask: keep fraction ~ keep_prob, kept values scaled by 1/keep
    c2 = torch.empty((M, N), device="cuda")
    _lib.check(lib.score_gemm(0, M, N, K, P(a), K, P(b), N, P(c2), N, P(bias), 1 | 8, 0.8, C.c_void_p(0), 1234,
                              C.c_void_p(0), 0, stream()), "gemm")
    full = a @ b
    kept = c2 != 0
    assert abs(float(kept.float().mean()) - 0.8) < 0.02
    assert torch.allclose(c2[kept], full[kept] / 0.8, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("D,F,K,B,T", [(4, 3, 2, 5, 3), (16, 4, 10, 9, 11), (16, 3, 10, 9, 11), (64, 4, 10, 6, 5),
                                       (64, 3, 5, 7, 4), (8, 1, 4, 6, 5), (128, 2, 20, 3, 4), (32, 5, 7, 4, 3)])
def test_coattn_fwd_bwd(D, F, K, B, T):
    # fused gather + co_attention (score.py:147-167) vs Oracle B's collapsed form + autograd
    lib = _lib.load()
    rng = np.random.default_rng(D + F + K)
    N = 500
    Dx = F * D
    table = rng.standard_normal((N, D)).astype(np.float32)
    table[0] = 0
    idx1 = rng.integers(0, N, (B, T, K, F)).astype(np.int32)
    idx2 = rng.integers(0, N, (B, T, K, F)).astype(np.int32)
    idx1[0, 0] = 0
    idx2[0, 0] = 0
    idx1[1, :, 2:] = idx1[1, :, :1]
    tgt_idx = rng.integers(1, N, (B, F))
    W = (rng.standard_normal((3 * Dx, 1)) * 0.2).astype(np.float32)
    bias = np.asarray([0.1], dtype=np.float32)
    tt = torch.tensor(table, requires_grad=True)
    Wt, bt = torch.tensor(W, requires_grad=True), torch.tensor(bias, requires_grad=True)
    s1 = tt[torch.as_tensor(idx1).long()].reshape(B, T, K, Dx)
    s2 = tt[torch.as_tensor(idx2).long()].reshape(B, T, K, Dx)
    tg = tt[torch.as_tensor(tgt_idx).long()].reshape(B, Dx).detach().requires_grad_(True)
    o1, o2, info = so._co_attention_collapsed(s1, s2, tg, Wt, bt)
    g1 = torch.tensor(rng.standard_normal((B, T, Dx)).astype(np.float32))
    g2 = torch.tensor(rng.standard_normal((B, T, Dx)).astype(np.float32))
    gi = torch.tensor(rng.standard_normal((B, T, 2 * K)).astype(np.float32))
    ((o1 * g1).sum() + (o2 * g2).sum() + (info * gi).sum()).backward()

    dt, di1, di2 = dev(table), dev(idx1), dev(idx2)
    dtg, dW, db = dev(tg.detach().numpy()), dev(W.reshape(-1)), dev(bias)
    out1 = torch.zeros((B * T, Dx + 4), device="cuda")
    out2 = torch.zeros((B * T, Dx), device="cuda")
    oinfo = torch.zeros((B * T, 2 * K + 1), device="cuda")
    rs = torch.zeros((B * T, K), device="cuda")
    _lib.check(lib.score_coattn_fwd(P(dt), N, D, F, K, B, T, P(di1), P(di2), P(dtg), P(dW), P(db), P(out1), Dx + 4,
                                    P(out2), Dx, P(oinfo), 2 * K + 1, P(rs), 0, stream()), "coattn_fwd")
    torch.cuda.synchronize()
    def relerr(got, want):       # max abs error relative to the largest reference magnitude
        want = want.detach().numpy().reshape(got.shape)
        return float(np.abs(got.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-30))
    assert relerr(out1[:, :Dx], o1) < 1e-5
    assert relerr(out2, o2) < 1e-5
    assert relerr(oinfo[:, :2 * K], info) < 1e-5
    assert float(out1[:, Dx:].abs().max()) == 0 and float(oinfo[:, 2 * K:].abs().max()) == 0

    gt = torch.zeros((N, D), device="cuda")
    dzs = torch.zeros((B * T,), device="cuda")
    gW = torch.zeros((3 * Dx,), device="cuda")
    scratch = torch.empty((1 << 21,), device="cuda")
    dg1, dg2, dgi = dev(g1), dev(g2), dev(gi)      # keep the device tensors alive across the call
    _lib.check(lib.score_coattn_bwd(P(dt), P(gt), N, D, F, K, B, T, P(di1), P(di2), P(dW), P(rs), P(dg1), Dx,
                                    P(dg2), Dx, P(dgi), 2 * K, P(dzs), P(gW), P(scratch), scratch.numel(),
                                    0, stream()), "coattn_bwd")
    torch.cuda.synchronize()
    # row gradients: autograd on `tt` also holds the target-row path, which this op does not own,
    # but tg was detached so tt.grad is exactly the neighbour-row scatter; row 0 is masked out.
    want = tt.grad.numpy().copy()
    want[0] = 0
    got = gt.cpu().numpy()
    assert np.abs(got[0]).max() == 0
    assert np.allclose(got, want, rtol=2e-4, atol=2e-5), np.abs(got - want).max()
    wg = Wt.grad.numpy().reshape(-1)
    assert np.allclose(gW[Dx:].cpu().numpy(), wg[Dx:], rtol=2e-4, atol=2e-5)
    # dzsum feeds w_t / bias / target grads: sum_bt dzsum == dbias
    assert abs(float(dzs.sum()) - float(bt.grad)) < 1e-4 * max(1.0, abs(float(bt.grad)))


def test_coattn_rca_sum_mode():
    lib = _lib.load()
    rng = np.random.default_rng(5)
    N, D, F, K, B, T = 200, 16, 3, 10, 4, 3
    table = rng.standard_normal((N, D)).astype(np.float32)
    table[0] = 0
    idx1 = rng.integers(0, N, (B, T, K, F)).astype(np.int32)
    idx2 = rng.integers(0, N, (B, T, K, F)).astype(np.int32)
    o1 = torch.zeros((B * T, F * D), device="cuda")
    o2 = torch.zeros((B * T, F * D), device="cuda")
    z = C.c_void_p(0)
    dt, d1, d2 = dev(table), dev(idx1), dev(idx2)
    _lib.check(lib.score_coattn_fwd(P(dt), N, D, F, K, B, T, P(d1), P(d2), z, z, z, P(o1),
                                    F * D, P(o2), F * D, z, 0, z, 1, stream()), "rca")
    w1 = table[idx1].reshape(B * T, K, F * D).sum(1)
    w2 = table[idx2].reshape(B * T, K, F * D).sum(1)
    assert np.allclose(o1.cpu().numpy(), w1, rtol=1e-5, atol=1e-5)
    assert np.allclose(o2.cpu().numpy(), w2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,T,H", [(5, 3, 8), (70, 11, 32), (33, 6, 128), (64, 9, 128), (9, 4, 20), (17, 5, 16), (40, 7, 64),
                                   (20, 1, 128), (16, 1, 32), (3, 2, 64), (35, 1, 256), (64, 2, 256)])
def test_gru_fwd_bwd(B, T, H):
    # dynamic_rnn(GRUCell) recurrence (score.py:205-208) vs the oracle's _gru + autograd
    lib = _lib.load()
    rng = np.random.default_rng(B + T + H)
    I = 12
    x = torch.tensor(rng.standard_normal((B, T, I)).astype(np.float32), requires_grad=True)
    Wg = torch.tensor((rng.standard_normal((I + H, 2 * H)) * 0.3).astype(np.float32), requires_grad=True)
    bg = torch.tensor(np.ones(2 * H, dtype=np.float32), requires_grad=True)
    Wc = torch.tensor((rng.standard_normal((I + H, H)) * 0.3).astype(np.float32), requires_grad=True)
    bc = torch.tensor((rng.standard_normal(H) * 0.1).astype(np.float32), requires_grad=True)
    length = rng.integers(1, T + 1, B)
    length[0] = T
    outs, hfin = so._gru(x, torch.as_tensor(length), Wg, bg, Wc, bc, H)
    go = torch.tensor(rng.standard_normal((B, T, H)).astype(np.float32))
    gf = torch.tensor(rng.standard_normal((B, H)).astype(np.float32))
    ((outs * go).sum() + (hfin * gf).sum()).backward()

    with torch.no_grad():
        xp = torch.cat([x.reshape(B * T, I) @ Wg[:I] + bg, x.reshape(B * T, I) @ Wc[:I] + bc], 1)
    dxp = dev(xp.numpy())
    dWg, dWc = dev(Wg.detach().numpy()), dev(Wc.detach().numpy())
    dl = dev(length.astype(np.int32))
    out = torch.zeros((B * T, H), device="cuda")
    gates = torch.zeros((B * T, 3 * H), device="cuda")
    fin = torch.zeros((B, H), device="cuda")
    wg_h = C.c_void_p(dWg.data_ptr() + I * 2 * H * 4)
    wc_h = C.c_void_p(dWc.data_ptr() + I * H * 4)
    _lib.check(lib.score_gru_fwd(B, T, H, P(dxp), wg_h, 2 * H, wc_h, H, P(dl), P(out), H, P(gates), P(fin),
                                 stream()), "gru_fwd")
    torch.cuda.synchronize()
    assert np.allclose(out.cpu().numpy(), outs.detach().numpy().reshape(B * T, H), rtol=1e-5, atol=2e-6)
    assert np.allclose(fin.cpu().numpy(), hfin.detach().numpy(), rtol=1e-5, atol=2e-6)

    dxproj = torch.zeros((B * T, 3 * H), device="cuda")
    rh = torch.zeros((B * T, H), device="cuda")
    hprev = torch.zeros((B * T * H + 3 * H * H,), device="cuda")
    dgo, dgf = dev(go), dev(gf)
    _lib.check(lib.score_gru_bwd(B, T, H, wg_h, 2 * H, wc_h, H, P(dl), P(out), H, P(gates), P(dgo), H,
                                 P(dgf), P(dxproj), P(rh), P(hprev), stream()), "gru_bwd")
    torch.cuda.synchronize()
    d = dxproj.cpu()
    # dx = dxproj . Wx^T ; dWx = x^T dxproj ; dWh = [hprev ; rh]^T dxproj ; db = colsum
    xf = x.detach().reshape(B * T, I)
    dx = d[:, :2 * H] @ Wg.detach()[:I].t() + d[:, 2 * H:] @ Wc.detach()[:I].t()
    assert np.allclose(dx.numpy(), x.grad.reshape(B * T, I).numpy(), rtol=2e-4, atol=2e-5)
    hp = hprev[:B * T * H].view(B * T, H).cpu()
    gWg = torch.cat([xf.t() @ d[:, :2 * H], hp.t() @ d[:, :2 * H]], 0)
    gWc = torch.cat([xf.t() @ d[:, 2 * H:], rh.cpu().t() @ d[:, 2 * H:]], 0)
    assert np.allclose(gWg.numpy(), Wg.grad.numpy(), rtol=2e-4, atol=5e-5)
    assert np.allclose(gWc.numpy(), Wc.grad.numpy(), rtol=2e-4, atol=5e-5)
    assert np.allclose(d[:, :2 * H].sum(0).numpy(), bg.grad.numpy(), rtol=2e-4, atol=5e-5)


def test_adam_matches_tf_form():
    lib = _lib.load()
    rng = np.random.default_rng(1)
    n, n_reg = 10007, 5000
    p = rng.standard_normal(n).astype(np.float32)
    pad = (-n) % 4
    params = {"p": p.copy()}
    opt = so.TFAdam(params)
    dp = dev(np.concatenate([p, np.zeros(pad, np.float32)]))
    dm, dv = torch.zeros_like(dp), torch.zeros_like(dp)
    lam = 1e-3
    for step in range(3):
        g = rng.standard_normal(n).astype(np.float32)
        g[::7] = 0
        geff = g.copy()
        geff[:n_reg] += np.float32(lam) * params["p"][:n_reg]
        a = float(opt.alpha(1e-3))
        opt.step(params, {"p": geff}, 1e-3)
        dg = dev(np.concatenate([g, np.zeros(pad, np.float32)]))
        _lib.check(lib.score_adam(P(dp), P(dm), P(dv), P(dg), n, n_reg, lam, a, 0.9, 0.999, 1e-8, None, stream()), "adam")
        torch.cuda.synchronize()
        assert np.allclose(dp.cpu().numpy()[:n], params["p"], rtol=1e-6, atol=1e-7)
        assert np.allclose(dm.cpu().numpy()[:n], opt.m["p"], rtol=1e-6, atol=1e-7)   # fma contraction: <= 1 ulp of the operands


@pytest.mark.parametrize("n,key_bits,dist", [
    (1, 5, "uniform"), (63, 11, "uniform"), (8192, 11, "uniform"), (8193, 21, "uniform"), (100_000, 18, "uniform"),
    (309_401, 21, "hot"),            # the reference's own batch (B = 200, T = 11, K = 10): where the plan uses this sort
    (2_874_369, 21, "hot"),          # cfg-3's occurrence count: two passes of 11 + 10 bits, hot categorical rows and the dummy row
    (1_000_003, 23, "hot"),          # 12 + 11 bits (the 5 M-row tables)
    (600_001, 32, "uniform"),        # three passes (an owner shard in front of the row)
    (300_000, 12, "uniform"),        # one pass of 12 bits
    (70_000, 21, "equal"),           # every key equal: the order of the values must be the input order
    (50_000, 1, "uniform"),
])
def test_sort_pairs_is_a_stable_sort(n, key_bits, dist):
    """csrc/sort.hip (score_sort_pairs): the occurrence sort of the index plan for small batches against torch's stable sort --
    keys ascending, equal keys in input order (what makes the pull scatter's sums reproducible)"""
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n + key_bits)
    hi = 1 << key_bits
    if dist == "uniform":
        keys = torch.randint(0, hi, (n,), device="cuda", generator=g, dtype=torch.int64)
    elif dist == "equal":
        keys = torch.full((n,), min(hi - 1, 12345), device="cuda", dtype=torch.int64)
    else:       # 30 % dummy row 0, 40 % twelve hot rows, the rest uniform
        u = torch.rand((n,), device="cuda", generator=g)
        keys = torch.randint(0, hi, (n,), device="cuda", generator=g, dtype=torch.int64)
        hot = hi - 1 - torch.randint(0, 12, (n,), device="cuda", generator=g, dtype=torch.int64)
        keys = torch.where(u < 0.3, torch.zeros_like(keys), torch.where(u < 0.7, hot, keys))
    vals = torch.arange(n, device="cuda", dtype=torch.int64)
    want_k, order = torch.sort(keys, stable=True)
    k0 = torch.where(keys >= (1 << 31), keys - (1 << 32), keys).to(torch.int32)         # the same 32 bits as signed words
    v0 = vals.to(torch.int32)
    k1, v1 = torch.empty_like(k0), torch.empty_like(v0)
    tb = int(lib.score_sort_pairs_temp_bytes(n))
    temp = torch.empty((tb,), dtype=torch.uint8, device="cuda")
    where = C.c_int32(-1)
    rc = lib.score_sort_pairs(P(k0), P(v0), P(k1), P(v1), n, key_bits, P(temp), tb, C.byref(where), stream())
    _lib.check(rc, "score_sort_pairs")
    torch.cuda.synchronize()
    ko, vo = (k1, v1) if where.value == 1 else (k0, v0)
    assert torch.equal(ko.to(torch.int64) & 0xFFFFFFFF, want_k)
    assert torch.equal(vo.to(torch.int64), order)
    # too little scratch / bad arguments
    assert lib.score_sort_pairs(P(k0), P(v0), P(k1), P(v1), n, key_bits, P(temp), 16, C.byref(where), stream()) == -3
    assert lib.score_sort_pairs(P(k0), P(v0), P(k1), P(v1), n, 33, P(temp), tb, C.byref(where), stream()) == -1


def test_rows_and_dense_in_one_launch_equal_the_two_calls():
    """score_adam_rows_and_dense (the small-table sweep + the dense variables, cfg-2's optimizer) against score_adam_rows followed by
    score_adam: bit for bit, states 0 / 1 / 2, the regularised range, a tail that is not a multiple of four -- and a set guard
    word: nothing applied by either half, ONE suppressed step counted"""
    lib = _lib.load()
    g_ = torch.Generator(device="cuda").manual_seed(3)
    R, D, n, n_reg = 1537, 16, 10243, 6001
    def mk():
        torch.manual_seed(5)
        t = lambda *s: torch.randn(*s, device="cuda")
        return dict(p=t(R, D), m=t(R, D) * 0.1, v=t(R, D).abs() * 0.01, g=t(R, D), wp=t(n + 1), wm=t(n + 1) * 0.1, wv=t(n + 1).abs() * 0.01,
                    wg=t(n + 1), flags=torch.randint(0, 3, (R,), dtype=torch.uint8, device="cuda", generator=g_.manual_seed(3)))
    a, b = mk(), mk()
    assert torch.equal(a["flags"], b["flags"]) and int((a["flags"] == 2).sum()) > 100 and int((a["flags"] == 0).sum()) > 100
    for step in range(3):
        al = 1e-3 * (1 + step)
        _lib.check(lib.score_adam_rows(P(a["p"]), P(a["m"]), P(a["v"]), P(a["g"]), R, D, P(a["flags"]), al, 0.9, 0.999, 1e-8, None, stream()), "rows")
        _lib.check(lib.score_adam(P(a["wp"]), P(a["wm"]), P(a["wv"]), P(a["wg"]), n, n_reg, 1e-3, al, 0.9, 0.999, 1e-8, None, stream()), "dense")
        _lib.check(lib.score_adam_rows_and_dense(P(b["p"]), P(b["m"]), P(b["v"]), P(b["g"]), R, D, P(b["flags"]), P(b["wp"]), P(b["wm"]),
                                                 P(b["wv"]), P(b["wg"]), n, n_reg, 1e-3, al, 0.9, 0.999, 1e-8, None, stream()), "both")
        torch.cuda.synchronize()
        for k in a:
            assert torch.equal(a[k], b[k]), (step, k)
        for x in (a, b):
            x["flags"][::3] = 2          # some rows get a gradient again
    word = torch.tensor([4, 0], dtype=torch.int32, device="cuda")
    guard = _lib.Guard(id_status=word.data_ptr(), skipped=word.data_ptr() + 4)
    snap = {k: v.clone() for k, v in b.items()}
    _lib.check(lib.score_adam_rows_and_dense(P(b["p"]), P(b["m"]), P(b["v"]), P(b["g"]), R, D, P(b["flags"]), P(b["wp"]), P(b["wm"]),
                                             P(b["wv"]), P(b["wg"]), n, n_reg, 1e-3, 1e-3, 0.9, 0.999, 1e-8, guard, stream()), "guarded")
    torch.cuda.synchronize()
    assert all(torch.equal(snap[k], b[k]) for k in b) and word.tolist() == [4, 1]
