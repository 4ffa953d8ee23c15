"""world_size-2 gloo (CPU) test of the sharded data-parallel path: score_amd/dist.py's request /
row / gradient routing, dense all-reduce and global-batch loss scaling, against ONE oracle model
trained on the concatenated batch.  Compute is the oracle-backed CpuBackend (tests/cpu_backend.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model_type, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from oracle import score_oracle as so
    from score_amd.dist import ShardedSCORE, TorchDistComm
    from cpu_backend import CpuBackend
    from helpers import random_batch, batch_tuple
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg_args = (203, 4, 8, 3, 3, 3, 4)            # odd N: the last shard is padded
    cfg = so.Cfg(*cfg_args, model_type=model_type)
    params = so.init_params(cfg, 5)
    rng = np.random.default_rng(100 + rank)
    batches = [random_batch(rng, cfg, 6) for _ in range(2)]
    be = CpuBackend(rank, world, model_type, cfg_args, params)
    model = ShardedSCORE(*cfg_args, comm=TorchDistComm(), backend=be, model_type=model_type)
    bts = [batch_tuple(b) for b in batches]
    # the next batch's index-only phase is run ahead, inside the current step
    losses = [model.train(None, bts[0], 1e-3, 1e-3, keep_prob=1.0, next_batch=bts[1])]
    losses.append(model.train(None, bts[1], 1e-3, 1e-3, keep_prob=1.0))
    pred, label, eloss = model.eval(None, batch_tuple(batches[0]), 1e-3)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), losses=np.asarray(losses), shard=be.full_table_part(),
             pred=np.asarray(pred), eloss=eloss, **{"dense/" + k: v for k, v in be.dense.items()},
             **{"b%d/%s" % (i, k): v for i, b in enumerate(batches) for k, v in b.items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("model_type", ["SCORE", "RCA"])
def test_two_rank_sharded_training_matches_single_model(tmp_path, model_type):
    from oracle import score_oracle as so
    from helpers import NAMES
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, model_type, str(tmp_path)), nprocs=world, join=True)
    z = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    cfg_args = (203, 4, 8, 3, 3, 3, 4)
    cfg = so.Cfg(*cfg_args, model_type=model_type)
    ref = so.OracleModel(*cfg_args, model_type=model_type, params=so.init_params(cfg, 5))
    for i in range(2):
        cat = tuple(np.concatenate([z[r]["b%d/%s" % (i, n)] for r in range(world)]) for n in NAMES)
        lref = ref.train(None, cat, 1e-3, 1e-3, keep_prob=1.0)
        for r in range(world):
            assert abs(z[r]["losses"][i] - lref) < 1e-5 * max(1.0, abs(lref)), (i, r)
    # replicated dense variables: identical on both ranks and equal to the single model's
    # (the bias of the last attention layer is softmax-shift-invariant: its true gradient is 0, so
    #  Adam amplifies rounding noise to +-lr there; it is only required to agree across ranks)
    shift_invariant = {"SCORE": "dense_5/bias", "RCA": "dense_3/bias"}[model_type]
    for k, v in ref.params.items():
        if k == "emb_mtx":
            continue
        assert np.array_equal(z[0]["dense/" + k], z[1]["dense/" + k]), k
        if k != shift_invariant:
            assert np.allclose(z[0]["dense/" + k], v, rtol=0, atol=2e-6), k
    # table: shard r holds rows r, r+G, ... ; padded tail rows stay zero
    N = cfg.N
    full = np.zeros((N, cfg.D), dtype=np.float32)
    for r in range(world):
        part = z[r]["shard"]
        n_r = len(range(r, N, world))
        full[r::world] = part[:n_r]
        assert not part[n_r:].any()
    want = ref.params["emb_mtx"].copy()
    want[0] = 0
    assert np.allclose(full, want, rtol=0, atol=2e-6)
    # eval on rank r's first batch == the single model evaluated on that batch
    for r in range(world):
        b = tuple(z[r]["b0/%s" % n] for n in NAMES)
        pref, _, lref = ref.eval(None, b, 1e-3)
        assert np.allclose(z[r]["pred"], pref, atol=1e-5)
        assert abs(float(z[r]["eloss"]) - lref) < 1e-5


# ---- the branch RCCL takes, with three and four ranks (threads of one process: torch's "threaded" process group) ---------
@pytest.mark.parametrize("world,mode", [(3, ""), (4, ""), (3, "force_remote"), (4, "force_remote")])
def test_ranks_on_the_rccl_code_path(world, mode):
    """score_amd/dist.py under a backend that is not gloo: all_to_all_single for the counts, the rows and the row gradients
    (nothing asked: the split form), or -- SCORE_A2A=remote -- all_to_all_remote = the own segment copied locally + ONE list
    all_to_all with empty own slots; all_reduce.  No multi-GPU node has been available to any round; tests/threaded_pg_worker.py
    runs that branch's index arithmetic with `world` ranks of unequal batch sizes (oracle backend) against one oracle model on
    the concatenated batch."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(HERE, "threaded_pg_worker.py"), str(world), mode], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["ok"] and j["loss_rel_err"] < 1e-5 and j["table_max_err"] < 2e-6 and j["dense_identical_across_ranks"]
    assert len({tuple(x) for x in j["last"]}) == 1               # every rank entered the same last collective (same sequence number)
    if mode == "force_remote":
        assert j["own_slot_sizes"] == [[0, 0]]                   # what a rank owns never went through the collective
        assert len(set(j["list_collectives_per_rank"])) == 1 and j["list_collectives_per_rank"][0] >= 6
        assert j["list_form"] == [True] * world and "'forced': 'remote'" in out.stderr and "'split'" not in out.stderr
    else:                                                        # nothing asked: the list form is never entered, not even to probe it
        assert j["list_form"] == [False] * world and j["list_collectives_per_rank"] == [0] * world
        assert "'form': 'split'" in out.stderr and "'forced': None" in out.stderr and "'remote'" not in out.stderr


def test_a_backend_that_refuses_empty_slots_gets_the_split_form():
    """TorchDistComm.probe_a2a with SCORE_A2A=probe (run by ShardedSCORE at set-up): both forms are tried; if the backend's list
    all_to_all will not take empty tensors, every rank settles on the split form -- the verdict is all-reduced, nothing is decided
    by catching an error inside a training step -- with one line on stderr from rank 0, and the results are the same."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(HERE, "threaded_pg_worker.py"), "3", "refuse_probe"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["ok"] and j["loss_rel_err"] < 1e-5 and j["table_max_err"] < 2e-6 and j["dense_identical_across_ranks"]
    assert j["list_form"] == [False] * 3 and j["list_collectives_per_rank"] == [1, 1, 1]      # tried once per rank: the set-up probe
    assert out.stderr.count("all-to-all probe over 3 ranks") == 1 and "'remote': False" in out.stderr and "'form': 'split'" in out.stderr


@pytest.mark.parametrize("mode", ["refuse", "refuse_force_split"])
def test_a_form_nobody_asked_for_is_never_entered(mode):
    """with nothing asked, or SCORE_A2A=split / bench.py --a2a split, the set-up probe runs all_to_all_single
    ONLY -- a list-form all_to_all that hangs or fails on one rank cannot take such a job down at set-up.  Here the list form
    raises whenever it is entered ("refuse") and the spy counts its calls: none, on any rank, and the run's results are the same."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(HERE, "threaded_pg_worker.py"), "3", mode],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j["ok"] and j["loss_rel_err"] < 1e-5 and j["table_max_err"] < 2e-6 and j["dense_identical_across_ranks"]
    assert j["list_form"] == [False] * 3 and j["list_collectives_per_rank"] == [0, 0, 0]
    assert ("'forced': 'split'" if mode.endswith("split") else "'forced': None") in out.stderr and "'remote'" not in out.stderr


# ---- an id outside the table on ONE rank: rejected on EVERY rank before the step starts (score.py:51-66) -------------
def _badid_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    from oracle import score_oracle as so
    from score_amd.dist import ShardedSCORE, TorchDistComm
    from cpu_backend import CpuBackend
    from helpers import random_batch, batch_tuple
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg_args = (203, 4, 8, 3, 3, 3, 4)
    cfg = so.Cfg(*cfg_args, model_type="SCORE")
    params = so.init_params(cfg, 5)
    rng = np.random.default_rng(300 + rank)
    good = [random_batch(rng, cfg, 6) for _ in range(3)]
    bad = {k: v.copy() for k, v in good[1].items()}
    if rank == 1:                                   # only rank 1 feeds it
        bad["item_1hop"][2, 1, 0, 1] = cfg_args[0] + 9
        bad["target_user"][0, 0] = -3

    def run(with_bad):
        be = CpuBackend(rank, world, "SCORE", cfg_args, params)
        model = ShardedSCORE(*cfg_args, comm=TorchDistComm(), backend=be)
        out, msg = [], ""
        # the bad batch is ALSO the look-ahead batch of the step before it: that step must run and report normally
        out.append(model.train(None, batch_tuple(good[0]), 1e-3, 1e-3, keep_prob=1.0,
                               next_batch=batch_tuple(bad) if with_bad else batch_tuple(good[2])))
        if with_bad:
            before = (be.table["t"].copy(), {k: v.copy() for k, v in be.dense.items()})
            try:
                model.train(None, batch_tuple(bad), 1e-3, 1e-3, keep_prob=1.0)
            except ValueError as e:
                msg = str(e)
            assert np.array_equal(before[0], be.table["t"]) and all(np.array_equal(before[1][k], be.dense[k]) for k in be.dense)
            try:
                model.eval(None, batch_tuple(bad), 1e-3)
                msg += " | eval: no error"
            except ValueError:
                pass
        out.append(model.train(None, batch_tuple(good[2]), 1e-3, 1e-3, keep_prob=1.0))
        return out, msg, be.table["t"].copy()
    l_bad, msg, t_bad = run(True)
    l_ref, _, t_ref = run(False)
    np.savez(os.path.join(out_dir, "bad%d.npz" % rank), l_bad=np.asarray(l_bad), l_ref=np.asarray(l_ref), msg=np.asarray(msg),
             same_table=bool(np.array_equal(t_bad, t_ref)))
    dist.destroy_process_group()


def test_bad_id_on_one_rank_is_rejected_on_every_rank_before_the_step(tmp_path):
    world = 2
    mp.spawn(_badid_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    z = [np.load(str(tmp_path / ("bad%d.npz" % r))) for r in range(world)]
    for r in range(world):
        msg = str(z[r]["msg"])
        assert "rank 1" in msg and "(item_1hop)" in msg and "(target_user)" in msg and "rank 0:" not in msg, (r, msg)
        assert "no error" not in msg
        # the run with the rejected batch in the middle == the run that never saw it (losses and shard, bit for bit)
        assert np.array_equal(z[r]["l_bad"], z[r]["l_ref"]) and bool(z[r]["same_table"])


# ---- harness.train_loop with several ranks: every rank must take the same branches -----------------
def _loop_worker(rank, world, port, out_dir, with_len):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from score_amd import harness as h
    from score_amd.dist import TorchDistComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Model(object):
        """train / eval are collectives (as ShardedSCORE's are): a rank that takes one step more than the others, or
        evaluates when they do not, would hang here (the process group times out) or pair up mismatched calls"""
        device = "cpu"

        def __init__(self):
            self.comm, self.n_train, self.n_eval, self.saved, self.pairs = TorchDistComm(), 0, 0, [], []

        def train(self, sess, batch_data, lr, reg_lambda, next_batch=None):
            t = torch.tensor([1.0 + self.n_train])
            self.comm.all_reduce_sum(t)
            assert float(t) == world * (1.0 + self.n_train)        # everybody is at the same step
            self.n_train += 1
            self.pairs.append((batch_data, next_batch))
            return 1.0 / self.n_train

        def save(self, sess, path):
            self.saved.append(self.n_train)

    m = Model()
    script = [[0.10, 0.30, 0.20, 0.50, 0.40, 0.45, 0.9], [0.30, 0.20, 0.50, 0.10, 0.70, 0.65, 0.1]][rank]

    def ev(model, batches, reg_lambda):
        t = torch.tensor([100.0 + model.n_eval])
        model.comm.all_reduce_sum(t)
        assert float(t) == world * (100.0 + model.n_eval)
        v = script[min(model.n_eval, len(script) - 1)]
        model.n_eval += 1
        return 0.0, 0.5, v, v, v, v, v, v, 0.25 + rank
    n_batches = 5 if rank == 0 else 3                          # rank 1's loader ends first: everybody stops there
    # with_len: loaders with a length (one agreement per EPOCH, no per-step collective); else plain iterators (one per step)
    mk = (lambda: list(range(10 * rank, 10 * rank + n_batches))) if with_len else \
         (lambda: iter(range(10 * rank, 10 * rank + n_batches)))
    n_coll = [0]
    orig = m.comm.all_reduce_sum

    def counted(t):
        n_coll[0] += 1
        return orig(t)
    m.comm.all_reduce_sum = counted
    out = h.train_loop(m, mk, lambda: [], 1e-3, 1e-4, 4, 18, epochs=2,
                       save_path="ckpt", evaluate_fn=ev, log=lambda s: None)
    # collectives of the loop itself: 6 train + 3 eval (the scripted model's own) + 3 metric means + the agreement
    assert n_coll[0] == 6 + 3 + 3 + (2 if with_len else 10), n_coll[0]      # (no length: the per-epoch probe + 4 per epoch)
    np.savez(os.path.join(out_dir, "loop%d.npz" % rank), steps=out["steps"], mrrs=np.asarray(out["vali_mrrs"]),
             vloss=np.asarray(out["vali_losses"]), saved=np.asarray(m.saved), n_eval=m.n_eval,
             last_next=np.asarray([-1 if p[1] is None else p[1] for p in m.pairs]))
    dist.destroy_process_group()


@pytest.mark.parametrize("with_len", [True, False])
def test_train_loop_two_ranks_take_the_same_branches(tmp_path, with_len):
    world = 2
    mp.spawn(_loop_worker, args=(world, _free_port(), str(tmp_path), with_len), nprocs=world, join=True)
    z = [np.load(str(tmp_path / ("loop%d.npz" % r))) for r in range(world)]
    # 3 steps per epoch (the shorter loader), eval_iter_num = 3: evaluations at steps 0, 3, 6
    assert int(z[0]["steps"]) == int(z[1]["steps"]) == 6 and int(z[0]["n_eval"]) == int(z[1]["n_eval"]) == 3
    for k in ("mrrs", "vloss", "saved"):
        assert np.array_equal(z[0][k], z[1][k]), k             # the rules saw the same numbers on both ranks
    assert np.allclose(z[0]["mrrs"], [0.2, 0.25, 0.35]) and np.allclose(z[0]["vloss"], [0.75] * 3)
    assert z[0]["saved"].tolist() == [3, 6]
    # nobody prefetches for a step that does not happen: the third step of an epoch has no look-ahead batch on either rank
    assert z[0]["last_next"].tolist() == [1, 2, -1, 1, 2, -1] and z[1]["last_next"].tolist() == [11, 12, -1, 11, 12, -1]
