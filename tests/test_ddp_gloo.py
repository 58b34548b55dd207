"""CPU, world_size 2 over gloo: the N>1 path of the sampler.  Clouds shard over ranks; the only
exchange on the data path is the all-reduce/world_size of the nb-1 boundary quantiles
(reference utils/ops.py:191-199) followed by the in-place momentum blend (201-233)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import torch_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from samble_amd import ops, synth
    nb, B, N = 6, 4, 256
    state = None
    log = []
    for call in range(3):
        # each rank owns clouds rank*B .. rank*B+B-1 of the global batch
        z = O.zscore(torch.from_numpy(np.abs(synth.features(B, 1, N, 50 + call, first_cloud=rank * B))))
        local_q = O.batch_quantiles(z.reshape(B, 1, N, 1), nb)
        q = ops.world_average(local_q.clone())
        state = ops.blend_boundaries(state, q, nb, 0.99)
        log.append((local_q.clone(), q.clone(), state[0].clone(), state[1].clone()))
    out[rank] = log
    dist.barrier()
    dist.destroy_process_group()


def test_boundary_exchange_two_ranks():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    state = None
    for call in range(3):
        lq0, q0, up0, lo0 = r0[call]
        lq1, q1, up1, lo1 = r1[call]
        assert not torch.equal(lq0, lq1), "ranks see different shards"
        # what the reference computes: SUM all-reduce then / world_size
        expect = (lq0 + lq1) / world
        assert torch.equal(q0, expect) and torch.equal(q1, expect)
        state = O.blend_boundaries(state, expect.clone(), 6, 0.99)
        for up, lo in ((up0, lo0), (up1, lo1)):
            assert torch.equal(up, state[0]) and torch.equal(lo, state[1]), "ranks hold identical boundaries"


def test_shards_tile_the_global_batch():
    from samble_amd import synth
    full = synth.features(8, 16, 64, 9)
    a = synth.features(4, 16, 64, 9, first_cloud=0)
    b = synth.features(4, 16, 64, 9, first_cloud=4)
    assert np.array_equal(np.concatenate([a, b]), full)


def test_single_process_is_identity():
    from samble_amd import ops
    q = torch.tensor([0.3, 0.1, -0.2])
    assert torch.equal(ops.world_average(q.clone()), q)
