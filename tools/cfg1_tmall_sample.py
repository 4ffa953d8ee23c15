#!/usr/bin/env python3
"""BASELINE.json configs[0] end to end on the MI355X path: the reference's bundled Tmall sample
(tests/golden/tmall_sample_log.npz = score-data/Tmall/raw_data/user_log_format1.csv as integer columns)
-> id remap (feateng_tmall.py) -> in-memory temporal graph (graph_storage.py) -> target lines (gen_target.py)
-> DeviceGraphLoader -> SCORE.train / evaluate_device under the training loop's rules (train_score.py:165-275)
with the Tmall constants of train_score.py:339-364.

  python tools/cfg1_tmall_sample.py [--batch 8] [--lr 1e-3] [--reg 1e-4] [--model SCORE]

The sample has 62 users, so the reference's batch sizes (100 / 200) give eval_iter_num = 0 (a modulo by
zero in train_score.py:230 too); the default batch here is 8 lines x 2."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(batch=16, lr=1e-3, reg_lambda=1e-4, model_type="SCORE", epochs=6, log=print, seed=1111):
    from score_amd import dataprep as dp, harness
    from score_amd.graph import DeviceGraphLoader
    from score_amd.model import MODELS
    raw = np.load(os.path.join(ROOT, "tests", "golden", "tmall_sample_log.npz"))["log"]
    g, r, targets = dp.tmall_pipeline(raw)
    c = dp.TMALL
    T = c["time_slice_num"] - c["start_time"] - 1                     # train_score.py:362
    K = c["obj_per_time_slice"]
    model = MODELS[model_type](r["feature_size"], c["eb_dim"], c["hidden_size"], T, K, c["user_fnum"], c["item_fnum"],
                               seed=seed)
    g.to_device(model.device)
    mk_train = lambda: DeviceGraphLoader(g, batch, targets["train"], c["start_time"], c["pred_time_train"],
                                         c["train_neg"], T, K)
    mk_vali = lambda: DeviceGraphLoader(g, c["eval_batch_size"], targets["validation"], c["start_time"],
                                        c["pred_time_validation"], c["test_neg"], T, K)
    mk_test = lambda: DeviceGraphLoader(g, c["eval_batch_size"], targets["test"], c["start_time"],
                                        c["pred_time_test"], c["test_neg"], T, K)
    out = harness.train_loop(model, mk_train, mk_vali, lr, reg_lambda, batch, len(targets["train"]), epochs=epochs, log=log)
    test = harness.evaluate_device(model, mk_test(), reg_lambda)
    out["test"] = dict(zip(("logloss", "auc", "ndcg_5", "ndcg_10", "hr_1", "hr_5", "hr_10", "mrr", "loss"), test))
    out["lines"] = {k: len(v) for k, v in targets.items()}
    out["feature_size"] = r["feature_size"]
    return model, out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--reg", type=float, default=1e-4)
    ap.add_argument("--model", default="SCORE")
    a = ap.parse_args()
    _, out = run(a.batch, a.lr, a.reg, a.model)
    print(json.dumps({k: v for k, v in out.items()}, default=float))
