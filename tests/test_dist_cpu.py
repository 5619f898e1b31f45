"""CPU, world_size 2 over gloo: query sharding + the single all-gather of pose-candidate records (nerfmatch_amd/dist.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerfmatch_amd import dist as nmdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = nmdist.shard_indices(n)
        recs = [nmdist.make_record(i, torch.eye(4) * (i + 1) if i % 3 else None, 0.5 * i, 0.25 * i, 100 + i) for i in mine]
        local = torch.stack(recs) if recs else torch.empty(0, nmdist.RECORD_FLOATS)
        out = nmdist.gather_records(local, n, "cpu")
        q.put((rank, mine, out.tolist()))  # plain lists: tensors in an mp.Queue outlive-race the worker
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8, 1])
def test_shard_and_gather_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = {r: mine for r, mine, _ in res}
    assert sorted(shards[0] + shards[1]) == list(range(n)) and not set(shards[0]) & set(shards[1])
    for _, _, out in res:  # every rank ends up with every query, ordered
        out = torch.tensor(out).reshape(-1, nmdist.RECORD_FLOATS)
        assert out.shape == (n, nmdist.RECORD_FLOATS)
        assert out[:, 0].tolist() == list(range(n))
        for i in range(n):
            assert out[i, 19] == 100 + i and out[i, 17] == 0.5 * i
            if i % 3:
                assert torch.equal(out[i, 1:17].reshape(4, 4), torch.eye(4) * (i + 1))
            else:
                assert torch.isnan(out[i, 1:17]).all()


def test_single_process_passthrough():
    recs = torch.stack([nmdist.make_record(i, None, 1.0, 2.0, 3) for i in (2, 0, 1)])
    out = nmdist.gather_records(recs, 3, "cpu")
    assert out[:, 0].tolist() == [0.0, 1.0, 2.0]
    assert nmdist.shard_indices(5) == [0, 1, 2, 3, 4]


def _train_worker(rank, world, port, q):
    """Two ranks, different batches: after GradBuckets the gradients equal those of the concatenated batch (mean loss)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(rank)  # every rank initialises differently: broadcast_module makes them rank 0's replicas
        model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
        unused = torch.nn.Linear(3, 3)  # never reached on rank 1: contributes zeros there
        nmdist.broadcast_module(torch.nn.ModuleList([model, unused]), src=0)
        params = list(model.parameters()) + list(unused.parameters())
        buckets = nmdist.GradBuckets(params, bucket_mb=0.0001)  # ~100 bytes: several buckets
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 6, 8, generator=g)
        for step in range(2):  # two steps: the bucket state is re-armed by finish()
            for p in params:
                p.grad = None
            loss = model(x[rank]).pow(2).mean()
            if rank == 0:
                loss = loss + unused(torch.ones(3)).sum()
            loss.backward()
            buckets.finish()
        q.put((rank, len(buckets.buckets), [p.grad.flatten().tolist() for p in params]))
    finally:
        dist.destroy_process_group()


def test_grad_buckets_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] > 2
    # single-process reference: mean of the two ranks' losses
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 4))
    unused = torch.nn.Linear(3, 3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 8, generator=g)
    with torch.enable_grad():
        loss = 0.5 * (model(x[0]).pow(2).mean() + model(x[1]).pow(2).mean()) + 0.5 * unused(torch.ones(3)).sum()
        loss.backward()
    ref = [p.grad.flatten() for p in list(model.parameters()) + list(unused.parameters())]
    for rank, _, grads in res:
        for mine, r in zip(grads, ref):
            assert torch.allclose(torch.tensor(mine), r, atol=1e-6), rank


def test_uninitialised_world_is_rejected(monkeypatch):
    """WORLD_SIZE > 1 without init_process_group: the data-parallel helpers refuse instead of running un-synchronised."""
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="not initialised"):
        nmdist.GradBuckets([torch.nn.Parameter(torch.zeros(3))])
    with pytest.raises(RuntimeError, match="not initialised"):
        nmdist.broadcast_module(torch.nn.Linear(2, 2))


def _eval_worker(rank, world, port, q):
    """NeRFMatchEvaluator.eval_data_loader under gloo, world size 2, with the GPU work stubbed out: batches of 2 queries are
    dealt round-robin over the ranks (ragged shards: rank 0 gets 3 batches / 5 queries, rank 1 gets 2 / 4) and every rank
    ends up with the records of all 9 queries in query order."""
    from argparse import Namespace

    from nerfmatch_amd import synth
    from nerfmatch_amd.nerfmatch_evaluator import NeRFMatchEvaluator

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ev = NeRFMatchEvaluator(Namespace(model=synth.matcher_config("c2f"), exp=Namespace(seed=0), data=Namespace()))
        seen = []

        def begin(batch, renderer, o):
            seen.append(batch["first"])
            return dict(Q=batch["image"].shape[0], batch=batch)

        def finish(st):
            b = st["batch"]
            Q = st["Q"]
            return dict(R_err=[0.5 * (b["first"] + j) for j in range(Q)], t_err=[0.25 * (b["first"] + j) for j in range(Q)],
                        num_matches=[100 + b["first"] + j for j in range(Q)], c2w_ests=[torch.eye(4) * (b["first"] + j + 1) for j in range(Q)],
                        iter_t_errs=[], iter_R_errs=[])

        ev._localize_begin, ev._localize_finish = begin, finish
        sizes = [2, 2, 2, 2, 1]
        loader = [dict(image=torch.zeros(n, 1), first=sum(sizes[:i])) for i, n in enumerate(sizes)]
        out = ev.eval_data_loader(data_loader=loader, solver="none")
        #  as many batches as ranks and the LAST one short: rank 1's only batch is the short one, which must not
        # define the batch size (indices 0, 1, 2 -- not 0, 1, 1); and batches that carry their own `idx`, of any sizes
        short = [dict(image=torch.zeros(2, 1), first=0), dict(image=torch.zeros(1, 1), first=2)]
        out2 = ev.eval_data_loader(data_loader=short, solver="none")
        own = [dict(image=torch.zeros(1, 1), first=0, idx=torch.tensor([0])), dict(image=torch.zeros(3, 1), first=1, idx=torch.tensor([1, 2, 3])),
               dict(image=torch.zeros(2, 1), first=4, idx=torch.tensor([4, 5]))]
        out3 = ev.eval_data_loader(data_loader=own, solver="none")
        assert out2["query_idx"].tolist() == [0, 1, 2] and out2["num_matches"].tolist() == [100.0, 101.0, 102.0], out2
        assert out3["query_idx"].tolist() == list(range(6)) and out3["num_matches"].tolist() == [100.0 + i for i in range(6)], out3
        q.put((rank, seen[:3 if rank == 0 else 2], out["query_idx"].tolist(), out["num_matches"].tolist(), out["R_err"].tolist(), out["c2w_est"][:, 0, 0].tolist()))
    finally:
        dist.destroy_process_group()


def test_evaluator_shards_batches_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 4, 8] and res[1][1] == [2, 6]  # first query of the batches each rank localised
    for _, _, idx, nm, rerr, diag in res:
        assert idx == list(range(9)) and nm == [100.0 + i for i in range(9)]
        assert rerr == [0.5 * i for i in range(9)] and diag == [float(i + 1) for i in range(9)]
