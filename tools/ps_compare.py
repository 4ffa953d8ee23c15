"""Region-by-region comparison of the per-sample whole-model kernels (csrc/persample.h) with the layer-by-layer pass on one
batch: prints the max |difference| / max |value| of every workspace region both forms write.  python tools/ps_compare.py [config]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from score_amd import _lib
from score_amd.model import SCORE
from score_amd.synth import make_world

FWD = [("xside", 2, "BT*I"), ("info", 1, "BT*4*K"), ("rsave", 2, "BT*K"), ("query", 1, "B*I"), ("head_inp", 1, "B*Dh"),
       ("gates", 2, "BT*3*H"), ("gru_out", 2, "BT*H"), ("q", 1, "B*Dk"), ("ainp", 1, "BT*2*Dk"), ("a1", 1, "BT*80"),
       ("a2", 1, "BT*40"), ("att_score", 1, "BT"), ("bn", 1, "B*Dh"), ("f1", 1, "B*200"), ("f2", 1, "B*80"), ("logit", 1, "B"),
       ("y_pred", 1, "B"), ("lossb", 1, "B"), ("dlogit", 1, "B"), ("dz2", 1, "B*80")]
BWD = [("dz1", 1, "B*200"), ("dbn", 1, "B*Dh"), ("dgstage", 1, "B*Dh"), ("ds", 1, "BT"), ("da2", 1, "BT*40"), ("da1", 1, "BT*80"),
       ("adzsum", 1, "B*80"), ("dq", 1, "B*Dk"), ("dxproj", 2, "BT*3*H"), ("rh", 2, "BT*H"), ("hprev", 2, "BT*H"),
       ("dxside", 2, "BT*I"), ("pcoef", 2, "BT*K"), ("dzcoef", 2, "BT*K"), ("dtgt", 1, "B*I"), ("S", 1, "2*B")]


def regions(m, B, A, names):
    cfg = m.cfg
    K, H = cfg.obj_per_time_slice, cfg.hidden_size
    Du, Di = cfg.user_fnum * cfg.eb_dim, cfg.item_fnum * cfg.eb_dim
    env = dict(B=B, BT=B * A, I=Du + Di, K=K, H=H, Dk=2 * H + 4 * K, Dh=2 * H + Du + Di)
    _, ws = m._workspace(B)
    out = {}
    for name, n, size in names:
        a, b = _lib.workspace_field(cfg, B, name)
        cnt = eval(size, {}, env)
        for i, off in enumerate((a, b)[:n]):
            out["%s[%d]" % (name, i)] = ws[off:off + cnt].clone()
    return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tmall_default"
    world, kw = make_world(name)
    B = kw.pop("batch")
    m = SCORE(**kw)
    batch = world.batch(B, 0)
    db = m.device_batch(batch)
    A = db.active_slices or kw["max_time_len"]
    res = {}
    for flags in (512, 0):
        m.debug_flags = flags
        m.forward_backward(db, 1e-4, 1.0)
        torch.cuda.synchronize()
        res[flags] = (regions(m, B, A, FWD + BWD), m.w_g.clone(), m.dense_table_grad().clone(),
                      float(m._workspace(B)[1][_lib.workspace_layout(m.cfg, B).loss].item()))
    ref, got = res[512], res[0]
    print("loss layered %.7f fused %.7f" % (ref[3], got[3]))
    for k in ref[0]:
        a, b = ref[0][k], got[0][k]
        print("%-12s max|ref| %.3e  max|diff| %.3e  nan %d" % (k, float(a.abs().max()), float((a - b).abs().max()),
                                                            int(torch.isnan(b).sum())))
    print("w_g        max|ref| %.3e  max|diff| %.3e" % (float(ref[1].abs().max()), float((ref[1] - got[1]).abs().max())))
    for e in m.entries:
        a, b = m._view(ref[1], e), m._view(got[1], e)
        print("  %-40s %.3e  %.3e" % (e[0], float(a.abs().max()), float((a - b).abs().max())))
    print("table_g    max|ref| %.3e  max|diff| %.3e" % (float(ref[2].abs().max()), float((ref[2] - got[2]).abs().max())))


if __name__ == "__main__":
    main()
