import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, ctypes as C
from oracle import score_oracle as so
from score_amd import _lib
from test_gpu_ops import P, dev, stream
lib=_lib.load()
for (D,F,K,B,T) in [(4, 3, 2, 5, 3), (16, 4, 10, 9, 11), (16, 3, 10, 9, 11), (64, 4, 10, 6, 5),(64, 3, 5, 7, 4), (8, 1, 4, 6, 5), (128, 2, 20, 3, 4), (32, 5, 7, 4, 3)]:
    rng = np.random.default_rng(D + F + K)
    N = 500; Dx=F*D
    table = rng.standard_normal((N, D)).astype(np.float32); table[0]=0
    idx1 = rng.integers(0, N, (B, T, K, F)).astype(np.int32)
    idx2 = rng.integers(0, N, (B, T, K, F)).astype(np.int32)
    idx1[0, 0] = 0; idx2[0, 0] = 0
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
    out1 = torch.zeros((B * T, Dx), device="cuda"); out2 = torch.zeros((B * T, Dx), device="cuda")
    oinfo = torch.zeros((B * T, 2 * K), device="cuda"); rs = torch.zeros((B * T, K), device="cuda")
    _lib.check(lib.score_coattn_fwd(P(dt), N, D, F, K, B, T, P(di1), P(di2), P(dtg), P(dW), P(db), P(out1), Dx, P(out2), Dx, P(oinfo), 2 * K, P(rs), 0, stream()), "f")
    gt = torch.zeros((N, D), device="cuda"); dzs = torch.zeros((B * T,), device="cuda"); gW = torch.zeros((3 * Dx,), device="cuda")
    scratch = torch.empty((1 << 21,), device="cuda")
    a,b_,c = dev(g1),dev(g2),dev(gi)
    _lib.check(lib.score_coattn_bwd(P(dt), P(gt), N, D, F, K, B, T, P(di1), P(di2), P(dW), P(rs), P(a), Dx, P(b_), Dx, P(c), 2 * K, P(dzs), P(gW), P(scratch), scratch.numel(), 0, stream()), "b")
    torch.cuda.synchronize()
    want = tt.grad.numpy().copy(); want[0]=0
    got = gt.cpu().numpy()
    err = np.abs(got-want)
    bad = np.argwhere(err > 1e-4*np.abs(want).max())
    print((D,F,K,B,T), "max err", err.max(), "nbad", len(bad), "of", (want!=0).sum(), "dW err", np.abs(gW[Dx:].cpu().numpy()-Wt.grad.numpy().reshape(-1)[Dx:]).max(), "dzs", float(dzs.sum()), float(bt.grad))
    if len(bad):
        rows = np.unique(bad[:,0])[:5]
        for r in rows:
            print("  row", r, "got", got[r][:4], "want", want[r][:4], "occ1", np.argwhere(idx1==r)[:3].tolist(), "occ2", np.argwhere(idx2==r)[:3].tolist())
