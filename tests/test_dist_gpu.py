"""GPU: the sharded localisation loop under real process isolation.  No multi-GPU node is available to the build, so
the N > 1 path is exercised by TWO processes on the ONE GPU: gloo process group (RCCL refuses two ranks on one device), the per-query
records staged through the host for the all-gather, everything else -- local_device(), per-process blob caches and operand-scale
calibration, the CU-partitioned two-stream loop, the shard-end collectives -- exactly what an 8-GPU run executes per rank."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
WORKER = Path(__file__).resolve().parent / "two_proc_worker.py"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_processes_on_one_gpu(gpu, built_lib, tmp_path):
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env0["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    # the single-process run of the same six-query stream (a fresh child as well: nothing of this pytest process's caches is involved)
    single = tmp_path / "single.pt"
    r = subprocess.run([sys.executable, str(WORKER), str(single)], env=dict(env0, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    port = str(_free_port())
    outs = [tmp_path / f"rank{k}.pt" for k in range(2)]
    procs = [subprocess.Popen([sys.executable, str(WORKER), str(outs[k])], env=dict(env0, WORLD_SIZE="2", RANK=str(k), LOCAL_RANK="0",
                                                                                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for k in range(2)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    ref = torch.load(single, weights_only=False)
    got = [torch.load(o, weights_only=False) for o in outs]
    assert ref["world"] == 1 and [g["world"] for g in got] == [2, 2] and sorted(g["rank"] for g in got) == [0, 1]
    assert all(g["agreed"] for g in got), "agree_calibration: the ranks' operand scales differed"
    assert got[0]["scales"] == got[1]["scales"] == ref["scales"]
    for g in got:  # every rank holds the records of ALL queries, ordered by query index, equal to the single-process run bit for bit
        assert g["query_idx"].tolist() == ref["query_idx"].tolist() == list(range(6))
        for k in ("num_matches", "R_err", "t_err", "c2w_est"):
            assert np.array_equal(g[k], ref[k], equal_nan=True), k
    assert int(ref["num_matches"].sum()) > 0
    # the shards: rank r localised queries r, r + 2, r + 4 -- and what it left in those batches is what the single process computed
    seen = set()
    for g in got:
        assert set(g["mine"]) == set(range(g["rank"], 6, 2))
        for q, d in g["mine"].items():
            seen.add(q)
            for k, v in d.items():
                assert torch.equal(v, ref["mine"][q][k]), (g["rank"], q, k)
    assert seen == set(range(6))
