"""N>1 path on CPU: world_size-2 gloo processes, windows sharded round-robin, poses
all-gathered (stands in for RCCL over xGMI; the oracle stands in for the GPU solve)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_windows, q):
    sys.path.insert(0, os.path.join(ROOT, "mov-slam_amd")); sys.path.insert(0, ROOT)
    from movba import shard, synth
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.windows_for_rank(n_windows, rank, world)
    local = []
    for wid in mine:
        w = synth.make_window(3, 1, 30, seed=shard.window_seed(wid), run_lo=2, run_hi=3)
        local.append(torch.from_numpy(oracle.solve(w)["poses"]))
    g = shard.gather_poses(torch.stack(local))
    dist.barrier()
    q.put((rank, mine, g.numpy()))
    dist.destroy_process_group()


def test_round_robin_assignment():
    from movba import shard
    assert shard.windows_for_rank(8, 0, 8) == [0] and shard.windows_for_rank(8, 7, 8) == [7]
    assert shard.windows_for_rank(8, 1, 2) == [1, 3, 5, 7]
    all_ids = sorted(sum((shard.windows_for_rank(8, r, 4) for r in range(4)), []))
    assert all_ids == list(range(8))
    assert shard.window_seed(3) == 2003


def test_two_rank_gloo_gather_matches_single_process(oracle_mod):
    from movba import shard, synth
    world, n_windows = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_windows, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    expect = {wid: oracle_mod.solve(synth.make_window(3, 1, 30, seed=shard.window_seed(wid), run_lo=2, run_hi=3))["poses"]
              for wid in range(n_windows)}
    for rank, mine, g in got:
        assert g.shape == (world, n_windows // world, 4, 7)
        for r in range(world):
            for k, wid in enumerate(shard.windows_for_rank(n_windows, r, world)):
                np.testing.assert_array_equal(g[r, k], expect[wid])     # every rank holds every window's poses
