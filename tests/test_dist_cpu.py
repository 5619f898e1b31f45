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
