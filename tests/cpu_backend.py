"""CPU stand-in for score_amd.dist.HipBackend, built on the oracle.  TEST ONLY: lets the
world_size-2 gloo tests drive score_amd/dist.py's routing (requests, rows, gradients,
dense all-reduce, loss scaling) without a GPU.  The product never imports this."""
import numpy as np
import torch

from oracle import score_oracle as so

NAMES = ("user_1hop", "user_2hop", "item_1hop", "item_2hop", "target_user", "target_item", "label", "length")
PLAN_ORDER = ("user_1hop", "item_2hop", "user_2hop", "item_1hop", "target_user", "target_item")


class CpuBackend(object):
    def __init__(self, rank, world, model_type, cfg_args, params):
        self.rank, self.world = rank, world
        self.device = torch.device("cpu")
        self.cfg = so.Cfg(*cfg_args, model_type=model_type)
        self.D = self.cfg.D
        emb = np.asarray(params["emb_mtx"], dtype=np.float32).copy()
        emb[0] = 0
        rows_local = (self.cfg.N + world - 1) // world
        shard = np.zeros((rows_local, self.D), dtype=np.float32)
        part = emb[rank::world]
        shard[:part.shape[0]] = part
        self.table = {"t": shard}
        self.table_opt = so.TFAdam(self.table)
        self.dense = {k: np.asarray(v, dtype=np.float32).copy() for k, v in params.items() if k != "emb_mtx"}
        self.dense_opt = so.TFAdam(self.dense)
        self.reg = {n: r for n, _, _, r in so.param_spec(self.cfg)}
        self.global_batch = 0
        self._wg = None

    def plan(self, batch_data, slot=0):
        b = dict(so.batch_to_arrays(batch_data))
        # what score_index_plan's occurrence fill does (score_state_t.id_status): an id outside [0, N) is reported -- bit i =
        # position of the tensor in the feed tuple -- and read as the dummy row 0
        bad_ids = 0
        for i, n in enumerate(NAMES[:6]):
            out = (b[n] < 0) | (b[n] >= self.cfg.N)
            if out.any():
                bad_ids |= 1 << i
                b[n] = np.where(out, 0, b[n]).astype(b[n].dtype)
        G = self.world
        rows_local = (self.cfg.N + G - 1) // G
        shift = 1
        while (1 << shift) < rows_local:
            shift += 1
        flat = np.concatenate([b[n].ravel() for n in PLAN_ORDER] + [np.zeros(1, np.int32)]).astype(np.int64)
        keys = ((flat % G) << shift) | (flat // G) if G > 1 else flat
        uniq, inv = np.unique(keys, return_inverse=True)
        U = len(uniq)
        offs = [int(np.searchsorted(uniq, o << shift)) for o in range(G)] + [U]
        remapped = {}
        pos = 0
        for n in PLAN_ORDER:
            sz = b[n].size
            remapped[n] = inv[pos:pos + sz].reshape(b[n].shape).astype(np.int32)
            pos += sz
        remapped["label"], remapped["length"] = b["label"], b["length"]
        return dict(B=b["label"].shape[0], U=U, offsets=offs, remapped=remapped,
                    unique_rows=torch.from_numpy((uniq & ((1 << shift) - 1)).astype(np.int32)), batch=b, bad_ids=bad_ids)

    def gather(self, req_rows):
        return torch.from_numpy(self.table["t"][req_rows.numpy().astype(np.int64)])

    def set_global_batch(self, n):
        self.global_batch = int(n)

    def forward(self, plan, mini, reg_lambda, keep_prob, masks):
        cfgU = so.Cfg(plan["U"], self.cfg.D, self.cfg.H, self.cfg.T, self.cfg.K, self.cfg.Fu, self.cfg.Fi,
                      self.cfg.model_type)
        P = so.to_torch_params(self.dense, torch.float32, requires_grad=True)
        P["emb_mtx"] = mini.clone().requires_grad_(True)
        if masks is not None:
            masks = [torch.as_tensor(np.asarray(m)) for m in masks]
        out = so.forward(cfgU, P, so.to_torch_batch(plan["remapped"]), keep_prob, masks, 0.0)
        share = out["log_loss"] * (plan["B"] / float(self.global_batch))
        loss = torch.stack([share.detach() + reg_lambda * out["l2"].detach(), share.detach(), out["l2"].detach()])
        return dict(P=P, share=share, y_pred=out["y_pred"].detach(), loss=loss)

    def backward(self, plan, mini, fw, keep_prob):
        fw["share"].backward()
        P = fw["P"]
        self._wg = torch.cat([(P[k].grad if P[k].grad is not None else torch.zeros_like(P[k])).reshape(-1)
                              for k in self.dense])
        return P["emb_mtx"].grad.clone()

    def dense_grad(self):
        return self._wg

    def accumulate(self, req_rows, grads_in, counts=None):
        g = torch.zeros(self.table["t"].shape, dtype=torch.float32)
        if req_rows.numel():
            g.index_add_(0, req_rows.long(), grads_in)
        self._tg = g.numpy()

    def adam(self, lr, reg_lambda):
        self.table_opt.step(self.table, {"t": self._tg}, lr)
        grads, pos = {}, 0
        for k, v in self.dense.items():
            g = self._wg[pos:pos + v.size].numpy().reshape(v.shape)
            pos += v.size
            grads[k] = g + np.float32(reg_lambda) * v if self.reg[k] else g
        self.dense_opt.step(self.dense, grads, lr)
        self._lam = reg_lambda

    def labels(self, plan):
        return torch.from_numpy(plan["batch"]["label"])

    def full_table_part(self):
        return self.table["t"]
