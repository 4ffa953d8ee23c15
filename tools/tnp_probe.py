"""Round-3 probe: the recurrences' weight gradients [x ; h]^T . [dgates | dcand] of both sides on large output tiles
(tools/tnp/tnp_probe.hip) against the shipped tiled kernel (score_gemm trans = 2, one call per product; in the step the
eight products go out as ONE grouped launch of 100 us + 14 us of slab reduce).  Run on the GPU box: python tools/tnp_probe.py"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from score_amd import _lib
lib = _lib.load()
so = os.path.join(tempfile.mkdtemp(), "tnp.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-result"] +
                      [a for a in sys.argv[1:] if a.startswith("-D")] + [os.path.join(root, "tools", "tnp", "tnp2_probe.hip" if os.environ.get("TNP2") else "tnp_probe.hip"), "-o", so])
x = C.CDLL(so)
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3
stripped = any(a_.startswith("-DTNP_NO") for a_ in sys.argv[1:])
K, I, H = int(os.environ.get("TNP_K", 18432)), 448, 128
NS_G = int(os.environ.get("TNP_NS", 16 if os.environ.get("TNP2") else 28))
for sides in (2,):
    xs = [torch.randn((K, I), device="cuda") for _ in range(sides)]
    hp = [torch.randn((K, H), device="cuda") for _ in range(sides)]
    rh = [torch.randn((K, H), device="cuda") for _ in range(sides)]
    dy = [torch.randn((K, 3 * H), device="cuda") * 0.1 for _ in range(sides)]
    M = I + H
    ns_g, ns_c = NS_G, int(os.environ.get('TNP_NSC', (NS_G + 1) // 2))
    ch = lambda ns: ((K + ns - 1) // ns + 31) // 32 * 32
    ch_g, ch_c = ch(ns_g), ch(ns_c)
    ns_g, ns_c = (K + ch_g - 1) // ch_g, (K + ch_c - 1) // ch_c
    slab_g = [torch.empty((ns_g, M, 2 * H), device="cuda") for _ in range(sides)]
    slab_c = [torch.empty((ns_c, M, H), device="cuda") for _ in range(sides)]
    out_g = [torch.empty((M, 2 * H), device="cuda") for _ in range(sides)]
    out_c = [torch.empty((M, H), device="cuda") for _ in range(sides)]
    # jobs: gk of every side, then ck of every side
    X = xs + xs; Hm = hp + rh; Y = dy + dy; slab = slab_g + slab_c
    nj = len(X)
    ia = lambda v: (C.c_int * nj)(*v)
    pa = lambda ts: (C.c_void_p * nj)(*[t.data_ptr() for t in ts])
    args = (nj, pa(X), pa(Hm), pa(Y), pa(slab), ia([I] * nj), ia([H] * nj), ia([3 * H] * nj), ia([I] * nj), ia([H] * nj),
            ia([0] * sides + [2 * H] * sides), ia([2 * H] * sides + [H] * sides), ia([ns_g] * sides + [ns_c] * sides),
            ia([ch_g] * sides + [ch_c] * sides), K, 4)
    def run():
        rc = x.tnp_launch(*args, st())
        assert rc == 0, rc
    def reduce():
        for s_ in range(sides):
            x.tnp_reduce(P(slab_g[s_]), ns_g, C.c_int64(M * 2 * H), P(out_g[s_]), st())
            x.tnp_reduce(P(slab_c[s_]), ns_c, C.c_int64(M * H), P(out_c[s_]), st())
    run(); reduce(); torch.cuda.synchronize()
    err = 0.0
    if not stripped:
        for s_ in range(sides):
            a_g = torch.cat([xs[s_], hp[s_]], 1).double(); a_c = torch.cat([xs[s_], rh[s_]], 1).double()
            ref_g = a_g.t() @ dy[s_][:, :2 * H].double(); ref_c = a_c.t() @ dy[s_][:, 2 * H:].double()
            err = max(err, float((out_g[s_].double() - ref_g).abs().max() / ref_g.abs().max()),
                      float((out_c[s_].double() - ref_c).abs().max() / ref_c.abs().max()))
    # shipped: the eight products, one call each
    scr = torch.empty((1 << 23,), device="cuda")
    c2 = torch.empty((I, 2 * H), device="cuda")
    def shipped():
        for s_ in range(sides):
            for (A, B_, n0, n1) in ((xs[s_], dy[s_], 0, 2 * H), (xs[s_], dy[s_], 2 * H, 3 * H), (hp[s_], dy[s_], 0, 2 * H), (rh[s_], dy[s_], 2 * H, 3 * H)):
                Mo = A.shape[1]
                lib.score_gemm(2, Mo, n1 - n0, K, P(A), Mo, C.c_void_p(B_.data_ptr() + 4 * n0), 3 * H, P(c2), n1 - n0, None, 32, C.c_float(1.0),
                               None, C.c_uint64(0), P(scr), C.c_int64(scr.numel()), st())
    t_k = timeit(run); t_r = timeit(reduce); t_s = timeit(shipped, 5)
    mac = 2.0 * sides * M * 3 * H * K
    print("K=%d sides=%d ns=%d/%d chunk=%d/%d: large tiles %7.1f us (+ reduce %5.1f us) = %5.1f TF-eq, err %.1e   shipped, 8 separate calls %7.1f us"
          % (K, sides, ns_g, ns_c, ch_g, ch_c, t_k, t_r, mac / t_k / 1e6, err, t_s), flush=True)
