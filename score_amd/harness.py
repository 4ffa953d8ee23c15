"""Host-side harness around the hot path: the evaluation loop and ranking metrics the reference's
driver wraps around model.eval (code/score/train_score.py:94-163).  Pure NumPy/Python, no TF."""
import math
import time

import numpy as np

TEST_NEG_SAMPLE_NUM = 99          # train_score.py:19: one positive + 99 negatives per target line


def getNDCG_at_K(ranklist, target_item, k):
    """log 2 / log(rank + 2) if the positive is ranked in the top k, else 0 (train_score.py:104-108)."""
    for i in range(min(k, len(ranklist))):
        if ranklist[i] == target_item:
            return math.log(2) / math.log(i + 2)
    return 0


def getHR_at_K(ranklist, target_item, k):
    return 1 if target_item in ranklist[:k] else 0


def getMRR(ranklist, target_item):
    for i, it in enumerate(ranklist):
        if it == target_item:
            return 1. / (i + 1)
    return 0


def _ranked(preds, target_iids, per_line):
    p = np.asarray(preds, dtype=np.float64).reshape(-1, per_line)
    ids = np.asarray(target_iids).reshape(-1, per_line)
    # descending score.  The reference calls np.argsort with its default kind, whose order among EQUAL scores
    # depends on NumPy's build (introsort / SIMD sort); a stable sort pins it (equal scores: larger index
    # first after the reversal) -- the rule score_ranking_quality implements on the device
    order = np.argsort(p, axis=1, kind="stable")[:, ::-1]
    ranked = np.take_along_axis(ids, order, axis=1)
    # 0-based rank of the first entry equal to the positive's id (column 0 of every line)
    hit = ranked == ids[:, :1]
    return hit.argmax(axis=1)


def get_ranking_quality(preds, target_iids, neg_sample_num=TEST_NEG_SAMPLE_NUM):
    """(NDCG@5, NDCG@10, HR@1, HR@5, HR@10, MRR) over lines of 1 + neg_sample_num candidates
    (train_score.py:122-142)."""
    r = _ranked(preds, target_iids, neg_sample_num + 1).astype(np.float64)
    gain = math.log(2) / np.log(r + 2)
    return (float(np.mean(np.where(r < 5, gain, 0.0))), float(np.mean(np.where(r < 10, gain, 0.0))),
            float(np.mean(r < 1)), float(np.mean(r < 5)), float(np.mean(r < 10)), float(np.mean(1.0 / (r + 1))))


def get_ndcg(preds, target_iids, neg_sample_num=TEST_NEG_SAMPLE_NUM):
    return get_ranking_quality(preds, target_iids, neg_sample_num)[0]


def evaluate(model, batches, reg_lambda, sess=None, neg_sample_num=TEST_NEG_SAMPLE_NUM, verbose=False):
    """train_score.py:144-163: run model.eval over the batches of a target file and return
    (logloss, auc, ndcg_5, ndcg_10, hr_1, hr_5, hr_10, mrr, mean batch loss)."""
    from sklearn.metrics import log_loss, roc_auc_score
    preds, labels, target_iids, losses = [], [], [], []
    t = time.time()
    for batch_data in batches:
        pred, label, loss = model.eval(sess, batch_data, reg_lambda)
        preds += pred
        labels += label
        losses.append(loss)
        ids = batch_data[5]
        if hasattr(ids, "cpu"):                   # device batches (DeviceGraphLoader) hand tensors over
            ids = ids.cpu().numpy()
        target_iids += np.array(ids)[:, 0].tolist()
    logloss = log_loss(labels, preds)
    auc = roc_auc_score(labels, preds)
    loss = sum(losses) / len(losses)
    ndcg_5, ndcg_10, hr_1, hr_5, hr_10, mrr = get_ranking_quality(preds, target_iids, neg_sample_num)
    if verbose:
        print("EVAL TIME: %.4fs" % (time.time() - t))
    return logloss, auc, ndcg_5, ndcg_10, hr_1, hr_5, hr_10, mrr, loss


TRAIN_NEG_SAMPLE_NUM = 1          # train_score.py:18


def train_loop(model, train_batches, vali_batches, lr, reg_lambda, train_batch_size, dataset_size, sess=None,
               epochs=6, save_path=None, evaluate_fn=None, neg_sample_num=TEST_NEG_SAMPLE_NUM, log=print, feed_ahead=True):
    """The training loop train_score.py:165-275 wraps around model.train / model.eval, rule for rule:

      * one evaluation before the first step (:205);
      * ``eval_iter_num = (dataset_size // 3) // (train_batch_size / (1 + TRAIN_NEG_SAMPLE_NUM))`` (:217, a float
        floor: three validation passes per epoch) -- every eval_iter_num-th step the mean training loss since the
        last evaluation is recorded and the validation set is evaluated (:230-245);
      * the model is saved whenever the new validation MRR beats every earlier one (:246-252);
      * early stop, only after the first epoch and with more than two evaluations (:254-258): MRR fell twice in
        a row, or improved by <= 0.001 twice in a row;
      * at most ``epochs`` (6, :219) passes over the training targets, a fresh loader per epoch (:222).

    ``train_batches`` / ``vali_batches``: zero-argument callables returning a fresh iterable of batches (the
    reference constructs a new GraphLoader each time).  ``evaluate_fn(model, batches, reg_lambda)`` defaults to
    evaluate_device when the model has eval_async, else evaluate.  Returns a dict with the curves the reference
    pickles (:264-266), the index of the best validation MRR and ``best_mrr`` (its return value, :275)."""
    if evaluate_fn is None:
        if hasattr(model, "eval_async"):
            evaluate_fn = lambda m, b, r: evaluate_device(m, b, r, neg_sample_num)
        else:
            evaluate_fn = lambda m, b, r: evaluate(m, b, r, sess, neg_sample_num)
    curves = dict(train_losses=[], vali_losses=[], vali_ndcgs_5=[], vali_ndcgs_10=[], vali_hrs_1=[], vali_hrs_5=[],
                  vali_hrs_10=[], vali_mrrs=[])

    # A model sharded over several ranks (score_amd.dist.ShardedSCORE): every train / eval call is a collective, so all
    # ranks must take the same branches.  The validation metrics every rule below looks at are therefore the MEAN over
    # the ranks (each rank evaluates its own validation shard), and an epoch ends for everybody as soon as any rank's
    # loader runs out (ranks may hold different numbers of batches); checkpoints are then all from the same step.
    comm = getattr(model, "comm", None)
    multi = comm is not None and getattr(comm, "world", 1) > 1

    def across_ranks(values, op="mean"):
        if not multi:
            return list(values)
        import torch
        dev = getattr(model, "device", "cpu")
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        comm.all_reduce_sum(t)
        out = t.cpu().tolist()
        return [v / comm.world for v in out] if op == "mean" else out

    def validate():
        _, _, n5, n10, h1, h5, h10, mrr, loss = evaluate_fn(model, vali_batches(), reg_lambda)
        n5, n10, h1, h5, h10, mrr, loss = across_ranks((n5, n10, h1, h5, h10, mrr, loss))
        for k, v in (("vali_ndcgs_5", n5), ("vali_ndcgs_10", n10), ("vali_hrs_1", h1), ("vali_hrs_5", h5),
                     ("vali_hrs_10", h10), ("vali_mrrs", mrr), ("vali_losses", loss)):
            curves[k].append(v)
        return n5, n10, h1, h5, h10, mrr, loss

    step, saves = 0, []
    n5, n10, h1, h5, h10, mrr, vloss = validate()
    log("STEP %d  LOSS TRAIN: NULL  LOSS VALI: %.4f  NDCG@5 VALI: %.4f  NDCG@10 VALI: %.4f  HR@1 VALI: %.4f  "
        "HR@5 VALI: %.4f  HR@10 VALI: %.4f  MRR VALI: %.4f" % (step, vloss, n5, n10, h1, h5, h10, mrr))
    early_stop = False
    eval_iter_num = (dataset_size // 3) // (train_batch_size / (1 + TRAIN_NEG_SAMPLE_NUM))
    if eval_iter_num < 1:
        raise ValueError("dataset_size %d too small for batch size %d: eval_iter_num = %r" %
                         (dataset_size, train_batch_size, eval_iter_num))
    losses_step = []
    vali_mrrs = curves["vali_mrrs"]
    # a model whose train() takes next_batch= (the row-sharded ShardedSCORE) is fed one batch ahead: its index plan and
    # row requests for batch t+1 then run under step t instead of on its critical path (two host read-backs per step)
    import inspect
    try:
        ahead = "next_batch" in inspect.signature(model.train).parameters
    except (TypeError, ValueError):
        ahead = False

    use_async = not ahead and hasattr(model, "train_async") and hasattr(model, "device")
    # ... and the single-device model's train_async takes the upcoming batch as a hint (SCOREBASE.apply_adam(next_batch=): with
    # the time-tiled table optimizer its rows are brought up to date beside this step's tail instead of in front of the next
    # forward pass).  Only device-resident batches qualify -- what model.feed() yields
    try:
        async_hint = use_async and "next_batch" in inspect.signature(model.train_async).parameters
    except (TypeError, ValueError):
        async_hint = False

    def with_next(it):
        it = iter(it)
        try:
            cur = next(it)
        except StopIteration:
            return
        for nxt in it:
            yield cur, nxt
            cur = nxt
        yield cur, None

    def steps_agreed(source):
        """several ranks, once per EPOCH: if every rank's loader has a length, the ranks agree on min(length) with one small
        all-reduce (each rank writes its length into its own slot) and the epoch then needs no per-step agreement -- an
        all-reduce plus a host read-back ahead of every step drains the stream each time, and the host could no longer run
        ahead of the GPU (which is what train_async, feed() and next_batch= are for).  None: some loader has no length"""
        if not multi:
            return None
        n = len(source) if hasattr(source, "__len__") else -1
        slots = [0.0] * (2 * comm.world)
        slots[comm.rank] = 1.0 if n >= 0 else 0.0
        slots[comm.world + comm.rank] = float(max(n, 0))
        got = across_ranks(slots, op="sum")
        if sum(got[:comm.world]) < comm.world:
            return None
        return int(min(got[comm.world:]))

    def in_step(pairs, n_steps=None):
        """several ranks: stop together.  n_steps (steps_agreed): exactly that many steps, the look-ahead batch of the last
        one dropped, no collective.  Otherwise a step is taken only if EVERY rank still has a batch for it (one small
        all-reduce per step, ahead of the step's own collectives); the look-ahead batch is dropped when some rank has
        none, so nobody prefetches for a step that will not happen"""
        if not multi:
            for p in pairs:
                yield p
            return
        it = iter(pairs)
        if n_steps is not None:
            for i in range(n_steps):
                p = next(it, None)
                if p is None:        # a loader whose __len__ overstates what it yields: say so here, on this rank, instead of
                    raise RuntimeError("train_loop: the loader of rank %d ended after %d of the %d batches its len() promised "
                                       "(the other ranks are inside that step's collectives: they time out)" % (comm.rank, i, n_steps))
                yield (p[0], p[1] if i + 1 < n_steps else None)
            return
        while True:
            p = next(it, None)
            have = across_ranks((0.0 if p is None else 1.0, 0.0 if (p is None or p[1] is None) else 1.0), op="sum")
            if have[0] < comm.world:
                return
            yield (p[0], p[1] if have[1] == comm.world else None)
    for epoch in range(epochs):
        if early_stop:
            break
        # host feed tuples (nested lists, as GraphLoader yields them) are converted and uploaded a batch or two ahead on a
        # worker thread (SCOREBASE.feed), under the step that is running
        source = train_batches()
        n_agreed = steps_agreed(source)
        if hasattr(model, "feed") and feed_ahead:
            source = model.feed(source)
        for batch_data, next_data in in_step(with_next(source), n_agreed):
            if early_stop:
                break
            if ahead:
                loss = model.train(sess, batch_data, lr, reg_lambda, next_batch=next_data)
            elif use_async:
                # the loss stays on the device until the next evaluation needs the mean: the host does not wait for the
                # GPU every step, so preparing batch t+1 (list flattening, loader work) overlaps step t
                if async_hint and next_data is not None and hasattr(next_data, "flat"):
                    loss = model.train_async(batch_data, lr, reg_lambda, next_batch=next_data).clone()
                else:
                    loss = model.train_async(batch_data, lr, reg_lambda).clone()
            else:
                loss = model.train(sess, batch_data, lr, reg_lambda)
            step += 1
            losses_step.append(loss)
            if step % eval_iter_num == 0:
                if use_async:
                    import torch
                    losses_step = torch.stack(losses_step).cpu().tolist()
                    if any(v != v for v in losses_step) and hasattr(model, "check_ids"):
                        model.check_ids()       # a NaN loss is how an out-of-range feature id shows (SCOREBASE.check_ids)
                train_loss = sum(losses_step) / len(losses_step)
                curves["train_losses"].append(train_loss)
                losses_step = []
                n5, n10, h1, h5, h10, mrr, vloss = validate()
                log("STEP %d  LOSS TRAIN: %.4f  LOSS VALI: %.4f  NDCG@5 VALI: %.4f  NDCG@10 VALI: %.4f  HR@1 VALI: %.4f  "
                    "HR@5 VALI: %.4f  HR@10 VALI: %.4f  MRR VALI: %.4f" % (step, train_loss, vloss, n5, n10, h1, h5, h10, mrr))
                if vali_mrrs[-1] > max(vali_mrrs[:-1]):
                    saves.append(step)
                    if save_path is not None:
                        model.save(sess, save_path)
                if len(vali_mrrs) > 2 and epoch > 0:
                    if vali_mrrs[-1] < vali_mrrs[-2] and vali_mrrs[-2] < vali_mrrs[-3]:
                        early_stop = True
                    if (vali_mrrs[-1] - vali_mrrs[-2]) <= 0.001 and (vali_mrrs[-2] - vali_mrrs[-3]) <= 0.001:
                        early_stop = True
        # the tail of an epoch after its last evaluation was never looked at: a batch with an id outside the table there would
        # leave the sticky status word set and every later optimizer step suppressed without anybody hearing of it
        if use_async and hasattr(model, "check_ids"):
            model.check_ids()
    if hasattr(model, "check_ids"):
        model.check_ids()
    index = int(np.argmax(vali_mrrs))
    curves.update(best_index=index, best_mrr=vali_mrrs[index], steps=step, saved_at_steps=saves,
                  eval_iter_num=eval_iter_num, early_stopped=early_stop)
    return curves


def ranking_quality_device(preds, target_iids, neg_sample_num=TEST_NEG_SAMPLE_NUM, return_ranks=False):
    """get_ranking_quality on the device (score_ranking_quality, include/score_hip.h): preds float32 and
    target_iids int32 device tensors of n_lines * (1 + neg_sample_num) entries.  One 6-float read-back."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    per = neg_sample_num + 1
    preds = preds.reshape(-1).contiguous().float()
    ids = target_iids.reshape(-1).contiguous().to(torch.int32)
    n_lines = preds.numel() // per
    if n_lines * per != preds.numel() or ids.numel() != preds.numel():
        raise ValueError("preds / target_iids must hold n_lines * (1 + neg_sample_num) entries")
    out = torch.empty((6,), dtype=torch.float32, device=preds.device)
    ranks = torch.empty((n_lines,), dtype=torch.int32, device=preds.device)
    scratch = torch.empty((6 * n_lines,), dtype=torch.float32, device=preds.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.score_ranking_quality(p(preds), p(ids), n_lines, per, p(out), p(ranks), p(scratch), scratch.numel(),
                                         C.c_void_p(torch.cuda.current_stream(preds.device).cuda_stream)),
               "score_ranking_quality")
    res = tuple(float(x) for x in out.cpu().tolist())
    return (res, ranks) if return_ranks else res


def auc_logloss_device(preds, labels):
    """(roc_auc_score, log_loss) of device tensors preds float32 [n], labels int32 [n] (score_auc_logloss)."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    preds = preds.reshape(-1).contiguous().float()
    labels = labels.reshape(-1).contiguous().to(torch.int32)
    n = preds.numel()
    if labels.numel() != n or n == 0:
        raise ValueError("preds and labels must be non-empty and of equal length")
    out = torch.empty((2,), dtype=torch.float64, device=preds.device)
    scratch = torch.empty((int(lib.score_auc_scratch_bytes(n)),), dtype=torch.uint8, device=preds.device)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.score_auc_logloss(p(preds), p(labels), n, p(out), p(scratch), scratch.numel(),
                                     C.c_void_p(torch.cuda.current_stream(preds.device).cuda_stream)),
               "score_auc_logloss")
    auc, ll = out.cpu().tolist()
    return auc, ll


def evaluate_device(model, batches, reg_lambda, neg_sample_num=TEST_NEG_SAMPLE_NUM):
    """evaluate() with predictions, ids and labels kept on the device: one forward per batch
    (model.eval_async), ranking metrics by score_ranking_quality, AUC / log-loss by score_auc_logloss; two
    small read-backs at the end.  Returns the same 9-tuple as evaluate()."""
    import torch
    preds, labels, iids, losses = [], [], [], []
    for batch_data in batches:
        db = model.device_batch(batch_data)
        pred, label, loss = model.eval_async(db, reg_lambda)
        preds.append(pred.clone()); labels.append(label); losses.append(loss.clone())
        iids.append(db.tensors[5][:, 0])
    preds, labels, iids = torch.cat(preds), torch.cat(labels), torch.cat(iids)
    ndcg_5, ndcg_10, hr_1, hr_5, hr_10, mrr = ranking_quality_device(preds, iids, neg_sample_num)
    auc, logloss = auc_logloss_device(preds, labels)
    loss = float(torch.stack(losses).mean().item())
    if loss != loss and hasattr(model, "check_ids"):
        model.check_ids()       # (a NaN loss is how an id outside the table shows: raised HERE, not blamed on the next train step)
    return (logloss, auc, ndcg_5, ndcg_10, hr_1, hr_5, hr_10, mrr, loss)
