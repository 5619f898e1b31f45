"""Multi-GPU layer of the localisation path: one process per GPU, query images sharded over ranks, ONE all-gather
of fixed-size per-query pose-candidate records at shard end (RCCL over xGMI through torch.distributed backend
"nccl"; "gloo" on CPU in the tests).  The reference evaluates on a single GPU only (nerf_evaluator.py:153,
nerfmatch_evaluator.py:70), so this is new behaviour: there is no data-path collective,
each rank holds full replicas of the (small) NeRF and matcher weights."""
import torch
import torch.distributed as dist

RECORD_FLOATS = 20  # [query_idx, c2w_est (16, row-major; NaN when PnP failed / not run), R_err, t_err, num_matches]


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n, rank=None, world_size=None):
    """Round-robin shard of range(n): rank r owns r, r+W, r+2W, ... (no shuffle, like DistributedSampler(shuffle=False))."""
    if rank is None or world_size is None:
        rank, world_size = world()
    return list(range(rank, n, world_size))


def make_record(query_idx, c2w_est, r_err, t_err, num_matches, device="cpu"):
    rec = torch.full((RECORD_FLOATS,), float("nan"), dtype=torch.float32)
    rec[0] = float(query_idx)
    if c2w_est is not None:
        rec[1:17] = torch.as_tensor(c2w_est, dtype=torch.float32).reshape(-1)
    rec[17], rec[18], rec[19] = float(r_err), float(t_err), float(num_matches)
    return rec.to(device)


def gather_records(records, n_total, device):
    """records: (k_local, RECORD_FLOATS) of this rank's shard -> (n, RECORD_FLOATS) ordered by query index on every rank.
    With `n_total` given the shards differ in length by at most one row and are padded to ceil(n/W); with n_total None
    (batched evaluation: shards differ by up to a batch) the padded length is the maximum over ranks (one extra 8-byte
    all-reduce).  One all_gather moves W * per * 80 bytes (latency-bound; per-link xGMI bandwidth is irrelevant here)."""
    rank, W = world()
    if W > 1 and dist.get_backend() == "gloo":
        device = "cpu"  # gloo gathers host tensors only (its GPU support ends at broadcast / all_reduce): the 80-byte records are staged through the host
    records = records.to(device=device, dtype=torch.float32)
    if W == 1:
        out = records
    else:
        if n_total is None:
            k = torch.tensor([records.shape[0]], device=device, dtype=torch.int64)
            dist.all_reduce(k, op=dist.ReduceOp.MAX)
            per = int(k.item())
        else:
            per = (n_total + W - 1) // W
        pad = torch.full((per, RECORD_FLOATS), float("nan"), device=device, dtype=torch.float32)
        pad[:, 0] = -1.0
        pad[: records.shape[0]] = records
        buf = [torch.empty_like(pad) for _ in range(W)]
        dist.all_gather(buf, pad)
        out = torch.cat(buf)
        out = out[out[:, 0] >= 0]
    order = torch.argsort(out[:, 0])
    return out[order]


def agree_calibration(renderer, device):
    """fp16x3 operand scales across ranks.  They are chosen on a seeded probe bundle (NerfRenderer.calibrate), i.e. they depend on the
    replicated parameters only and every rank computes the same ones; this makes it a CHECKED property: one 24-int MIN all-reduce,
    every rank adopts the element-wise minimum (the safe side, should a rank have re-calibrated after a saturation event) and re-packs if
    its own differed.  Returns True when all ranks had agreed already.  World size 1 / other precisions: no collective, True."""
    rank, W = world()
    scales = renderer.calibrate(device) if renderer is not None and hasattr(renderer, "calibrate") else None
    if W == 1 or scales is None:
        return True
    mine = torch.tensor(scales["coarse"] + scales["fine"], dtype=torch.int32, device=device)
    low = mine.clone()
    dist.all_reduce(low, op=dist.ReduceOp.MIN)
    same = torch.tensor([int(torch.equal(low, mine))], dtype=torch.int32, device=device)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if not torch.equal(low, mine):
        vals = low.tolist()
        renderer.set_calibration(device, dict(coarse=vals[:12], fine=vals[12:]))
    return bool(int(same.item()))


# ----------------------------------------------------------------------------- data-parallel training
def require_initialized():
    """One process per GPU: a launcher that set WORLD_SIZE > 1 must have called init_process_group before any of the
    data-parallel helpers are built -- otherwise they would silently run un-synchronised."""
    import os

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("WORLD_SIZE > 1 but torch.distributed is not initialised: call init_process_group first")


def broadcast_module(module, src=0):
    """Every parameter and buffer of `module` takes rank `src`'s value (no-op at world size 1)."""
    require_initialized()
    if world()[1] == 1:
        return
    seen = set()
    for t in list(module.parameters()) + list(module.buffers()):
        if id(t) in seen:  # shared modules (im_sa is pt_sa) register the same tensor twice
            continue
        seen.add(id(t))
        dist.broadcast(t.data, src=src)



class GradBuckets:
    """Bucketed gradient all-reduce for data-parallel training of the matcher (one process per GPU, replicas of the
    weights, a different batch per rank) -- the role Lightning's DDP plugin plays in the reference
    (nerfmatch_c2f_trainer.py:793-860, `gpu_num` > 1).

    Parameters are packed, in reverse registration order (the order the backward pass finishes them), into flat buckets
    of about `bucket_mb` MiB.  A post-accumulate-grad hook per parameter counts its bucket down; when the last gradient of
    a bucket has been produced the bucket is copied into its flat buffer and an asynchronous all-reduce is launched
    in bucket order (every rank issues the same sequence of collectives; RCCL: ring over xGMI, per-link bound, so few large buckets -- the default 64 MiB keeps each ring step
    >= 8 MiB on 8 GPUs, well past the latency-dominated regime -- while still overlapping with the rest of the backward pass).
    `finish()` waits for the outstanding buckets, averages and scatters the result back into `.grad`; parameters that
    produced no gradient on this rank (e.g. the fine stage when no match survived) contribute zeros, so every rank
    issues the same collectives.  World size 1: every method is a no-op."""

    def __init__(self, params, bucket_mb=64, average=True):
        require_initialized()
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.rank, self.world = world()
        self.buckets = []   # list of lists of parameter indices
        self._owner = {}
        cap = int(bucket_mb * (1 << 20))
        cur, cur_bytes = [], 0
        for idx in reversed(range(len(self.params))):
            nbytes = self.params[idx].numel() * self.params[idx].element_size()
            if cur and cur_bytes + nbytes > cap:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(idx)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        for b, idxs in enumerate(self.buckets):
            for i in idxs:
                self._owner[i] = b
        self._flat = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._hooks = []
        if self.world > 1:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.start()

    def _make_hook(self, i):
        def hook(_param):
            b = self._owner[i]
            self._pending[b] -= 1
            # every rank must issue the collectives in the same order: bucket b goes out once buckets 0..b are complete
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1
        return hook

    def start(self):
        """Call before each backward pass (after zero_grad)."""
        if world()[1] != self.world:
            raise RuntimeError("GradBuckets was built under a different world size; build it after init_process_group")
        self._pending = [len(idxs) for idxs in self.buckets]
        self._work = [None] * len(self.buckets)
        self._next = 0

    def _launch(self, b):
        idxs = self.buckets[b]
        ps = [self.params[i] for i in idxs]
        total = sum(p.numel() for p in ps)
        flat = self._flat[b]
        if flat is None or flat.device != ps[0].device:
            flat = self._flat[b] = torch.empty(total, device=ps[0].device, dtype=ps[0].dtype)
        off = 0
        for p in ps:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self._work[b] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """Call after backward(): launches buckets whose parameters produced no gradient, waits, writes `.grad` back."""
        if self.world == 1:
            return
        while self._next < len(self.buckets):  # buckets held back by a parameter without gradient on this rank
            self._launch(self._next)
            self._next += 1
        for b, idxs in enumerate(self.buckets):
            self._work[b].wait()
            flat = self._flat[b]
            if self.average:
                flat.div_(self.world)
            off = 0
            for i in idxs:
                p = self.params[i]
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        self.start()

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
