"""Helper of tests/test_dist_gloo.py::test_three_ranks_on_the_rccl_code_path (run as a script in a process of its own).

torch's multi-threaded process group ("threaded" backend, torch.testing._internal) gives W ranks as THREADS of one process with
Python implementations of the collectives -- including the list form of all_to_all, which gloo lacks.  To score_amd.dist's
TorchDistComm that backend is "not gloo", so it takes the branch RCCL takes: all_to_all_single for the counts,
all_to_all_remote = the own segment copied locally + ONE dist.all_to_all over per-peer views with EMPTY tensors in the own slot
(what a rank owns never goes through the collective), all_reduce.  No multi-GPU node has been available to any round; this at
least executes that branch's index arithmetic with three ranks, against one oracle model trained on the concatenated batch.
Prints one JSON line."""
import json
import os
import sys
import threading

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main(world=3, refuse=False, force=None):
    import torch.testing._internal.distributed.multi_threaded_pg as mt
    torch._C._distributed_c10d._set_thread_isolation_mode(True)
    mt._install_threaded_pg()
    from oracle import score_oracle as so
    from score_amd.dist import ShardedSCORE, TorchDistComm
    from cpu_backend import CpuBackend
    from helpers import random_batch, batch_tuple, NAMES
    if force:
        os.environ["SCORE_A2A"] = force           # (what bench.py --a2a sets)
    else:
        os.environ.pop("SCORE_A2A", None)
    cfg_args = (203, 4, 8, 3, 3, 3, 4)            # odd N: the last shards are padded
    cfg = so.Cfg(*cfg_args, model_type="SCORE")
    params = so.init_params(cfg, 5)
    store = dist.HashStore()
    out, errs = {}, []
    batches = {r: [random_batch(np.random.default_rng(500 + r), cfg, 4 + r) for _ in range(2)] for r in range(world)}   # unequal B

    def rank_main(r):
        try:
            dist.init_process_group(backend="threaded", rank=r, world_size=world, store=store)
            comm = TorchDistComm()
            assert not comm._gloo and comm.world == world
            seen = []
            orig = comm.dist.all_to_all

            def spy(outs, ins, group=None):          # the own slot of the collective is empty on every rank
                seen.append((int(outs[r].numel()), int(ins[r].numel()), len(outs)))
                if refuse:                           # a backend whose list form will not take empty slots (argument check:
                    raise RuntimeError("all_to_all: empty tensor in slot %d" % r)      # nothing enqueued, every rank alike)
                return orig(outs, ins, group=group)
            import types
            comm.dist = types.SimpleNamespace(**{k: getattr(comm.dist, k) for k in dir(comm.dist) if not k.startswith("__")})
            comm.dist.all_to_all = spy
            be = CpuBackend(r, world, "SCORE", cfg_args, params)
            model = ShardedSCORE(*cfg_args, comm=comm, backend=be)
            bts = [batch_tuple(b) for b in batches[r]]
            losses = [model.train(None, bts[0], 1e-3, 1e-3, keep_prob=1.0, next_batch=bts[1])]
            losses.append(model.train(None, bts[1], 1e-3, 1e-3, keep_prob=1.0))
            out[r] = dict(losses=losses, shard=be.full_table_part(), dense={k: v.copy() for k, v in be.dense.items()},
                          a2a=seen, last=comm.last, list_form=comm._list_form)
            dist.destroy_process_group()
        except Exception:
            import traceback
            errs.append(traceback.format_exc())
    ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join(timeout=240) for t in ths]
    if errs or len(out) != world:
        print(json.dumps({"ok": False, "errors": errs[:2]}))
        return 1
    ref = so.OracleModel(*cfg_args, model_type="SCORE", params=so.init_params(cfg, 5))
    worst = 0.0
    for i in range(2):
        cat = tuple(np.concatenate([batches[r][i][n] for r in range(world)]) for n in NAMES)
        lref = ref.train(None, cat, 1e-3, 1e-3, keep_prob=1.0)
        for r in range(world):
            worst = max(worst, abs(out[r]["losses"][i] - lref) / max(1.0, abs(lref)))
    N, D = cfg.N, cfg.D
    full = np.zeros((N, D), dtype=np.float32)
    for r in range(world):
        n_r = len(range(r, N, world))
        full[r::world] = out[r]["shard"][:n_r]
    want = ref.params["emb_mtx"].copy()
    want[0] = 0
    dense_same = all(np.array_equal(out[0]["dense"][k], out[r]["dense"][k]) for r in range(1, world) for k in out[0]["dense"])
    print(json.dumps({"ok": True, "loss_rel_err": worst, "table_max_err": float(np.abs(full - want).max()),
                      "dense_identical_across_ranks": bool(dense_same),
                      "own_slot_sizes": sorted({(a, b) for r in range(world) for a, b, _ in out[r]["a2a"]}),
                      "list_collectives_per_rank": [len(out[r]["a2a"]) for r in range(world)],
                      "list_form": [out[r]["list_form"] for r in range(world)],
                      "last": [list(out[r]["last"]) for r in range(world)]}))
    return 0


if __name__ == "__main__":
    mode = sys.argv[2] if len(sys.argv) > 2 else ""
    # modes: "" (nothing asked: the split form), "force_remote", "refuse_probe" (SCORE_A2A=probe, the list form raises),
    # "refuse" (nothing asked, the list form raises if entered), "refuse_force_split"
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 3, refuse=mode.startswith("refuse"),
                  force="split" if mode.endswith("force_split") else "remote" if mode.endswith("force_remote")
                  else "probe" if mode.endswith("probe") else None))
