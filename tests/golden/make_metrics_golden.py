"""Golden vectors for the ranking metrics of code/score/train_score.py:94-142, produced by the
REFERENCE's own functions.  train_score.py cannot be imported (it imports tensorflow at the top),
so this script -- run in the build container only, where /root/reference is mounted -- parses the
file, extracts the five metric function definitions with `ast`, executes just those, and records
their outputs on seeded inputs.  Only inputs/outputs are committed (metrics_golden.npz)."""
import ast
import math
import os

import numpy as np

REF = "/root/reference/code/score/train_score.py"
HERE = os.path.dirname(os.path.abspath(__file__))
WANT = ("getNDCG_at_K", "getHR_at_K", "getMRR", "get_ranking_quality", "get_ndcg")


def load_reference_functions():
    tree = ast.parse(open(REF).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    ns = {"np": np, "math": math, "TEST_NEG_SAMPLE_NUM": 99}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


def main():
    ns = load_reference_functions()
    rng = np.random.Generator(np.random.PCG64(7))
    out = {}
    for case, n_lines in enumerate((1, 3, 17)):
        preds = rng.random(n_lines * 100)
        if case == 1:
            preds[0] = preds[1:100].max() + 1.0          # a perfect first line
            preds[100:200] = 0.5                          # all ties on the second
        iids = rng.integers(1000, 5000, n_lines * 100)
        res = ns["get_ranking_quality"](preds.tolist(), iids.tolist())
        out["c%d/preds" % case] = preds
        out["c%d/iids" % case] = iids
        out["c%d/quality" % case] = np.asarray(res, dtype=np.float64)
        out["c%d/ndcg5" % case] = np.float64(ns["get_ndcg"](preds.tolist(), iids.tolist()))
    # scalar helpers on hand-made rank lists
    ranklist = [5, 9, 2, 7, 1, 8, 3, 4, 6, 0]
    out["scalar/ranklist"] = np.asarray(ranklist)
    out["scalar/ndcg"] = np.asarray([[ns["getNDCG_at_K"](ranklist, t, k) for k in (1, 5, 10)] for t in range(10)])
    out["scalar/hr"] = np.asarray([[ns["getHR_at_K"](ranklist, t, k) for k in (1, 5, 10)] for t in range(10)])
    out["scalar/mrr"] = np.asarray([ns["getMRR"](ranklist, t) for t in range(11)])
    np.savez_compressed(os.path.join(HERE, "metrics_golden.npz"), **out)
    print("wrote metrics_golden.npz")


if __name__ == "__main__":
    main()
