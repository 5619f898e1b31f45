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
    """records: (k_local, RECORD_FLOATS) of this rank's shard -> (n_total, RECORD_FLOATS) ordered by query index on
    every rank.  Shards differ in length by at most one row, so rows are padded to ceil(n/W) and one all_gather moves
    W * ceil(n/W) * 80 bytes (latency-bound; per-link xGMI bandwidth is irrelevant here)."""
    rank, W = world()
    records = records.to(device=device, dtype=torch.float32)
    if W == 1:
        out = records
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
