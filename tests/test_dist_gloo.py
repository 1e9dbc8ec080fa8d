"""world_size-2 gloo test of the multi-GPU layer (fixca.dist): parameter
broadcast from rank 0, image sharding i mod N, status all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fixca
from fixca import dist as fdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2,
                                 x_blue=0.25) if rank == 0 else None
        got = fdist.broadcast_params(mine)
        idx = fdist.shard_indices(11, rank, world)
        ok_all = fdist.all_ranks_ok(True)
        ok_one_bad = fdist.all_ranks_ok(rank != 1)
        q.put((rank, got.as_dict(), idx, ok_all, ok_one_bad))
    finally:
        dist.destroy_process_group()


def test_broadcast_shard_status_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = fixca.make_params(blue=2.0, red=-2.0, lens_x=658, lens_y=1280, interpolation=2, x_blue=0.25).as_dict()
    assert res[0][1] == want and res[1][1] == want
    assert res[0][2] == [0, 2, 4, 6, 8, 10] and res[1][2] == [1, 3, 5, 7, 9]
    assert sorted(res[0][2] + res[1][2]) == list(range(11))            # every image exactly once
    assert res[0][3] and res[1][3]
    assert not res[0][4] and not res[1][4]                              # one failing rank fails the batch


def test_params_tensor_roundtrip():
    p = fixca.make_params(blue=-3.3, red=4.7, lens_x=-1, lens_y=-1, interpolation=1, y_red=-3.0)
    t = fdist.params_to_tensor(p)
    assert t.dtype == torch.uint8 and t.numel() == 80
    assert fdist.tensor_to_params(t).as_dict() == p.as_dict()
    assert fdist.broadcast_params(p).as_dict() == p.as_dict()           # no process group: identity
    assert fdist.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
