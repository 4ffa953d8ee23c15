"""cfg-1 fixture: the reference's bundled Tmall sample (score-data/Tmall/raw_data/user_log_format1.csv,
9,999 behaviour rows) as integer columns.  DATA only: the seven CSV columns, row order kept, nothing of
the reference's code.  /root/reference does not exist on the GPU box, so the plumbing test reads this file.

Run (container only):  python tests/golden/make_tmall_sample.py"""
import os

import numpy as np

SRC = "/root/reference/score-data/Tmall/raw_data/user_log_format1.csv"
HERE = os.path.dirname(os.path.abspath(__file__))
COLUMNS = ("user_id", "item_id", "cat_id", "seller_id", "brand_id", "time_stamp", "action_type")


def main():
    with open(SRC) as f:
        header = f.readline().strip().split(",")
        assert tuple(header) == COLUMNS, header
        rows = [[int(x) if x else -1 for x in line.rstrip("\n").split(",")] for line in f if line.strip()]   # an empty field
        # (one brand_id) is its own vocabulary entry in the reference (a '' key): kept as -1
    a = np.asarray(rows, dtype=np.int32)            # time_stamp is MMDD (e.g. 0829 -> 829)
    out = os.path.join(HERE, "tmall_sample_log.npz")
    np.savez_compressed(out, log=a, columns=np.asarray(COLUMNS))
    print("wrote", out, a.shape, os.path.getsize(out))


if __name__ == "__main__":
    main()
