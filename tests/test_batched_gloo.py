"""CPU test of the N>1 path: world size 2 over gloo.  The batch is sharded with no data-path collective;
per-matrix records gathered at the end must be identical to the single-process run (the per-matrix work
is done by the CPU oracle here -- on the GPU box the same driver calls the HIP path)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _work(i):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import matgen
    from oracle import oracle as O
    d, idx, p = matgen.poisson3d(6 + (i % 3))
    d = d + np.where(d > 0, 0.01 * i, 0.0)          # distinct diagonal shift per batch member
    L, U = O.orc().ilu0((d, idx, p, True))
    x = O.orc().apply_lu(L, U, np.ones(p.shape[0] - 1), O.ID)
    return (int(L[2][-1]), int(U[2][-1]), float(x.sum()))


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from ilupp_amd.batched import run_batch
    recs = run_batch(8, _work)
    dist.barrier()
    if rank == 0:
        q.put(recs)
    dist.destroy_process_group()


def test_shard_is_a_partition():
    sys.path.insert(0, ROOT)
    from ilupp_amd.batched import shard
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard(8, r, world))
        assert seen == list(range(8))


@pytest.mark.timeout(300)
def test_batched_world2_matches_single_process():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [_work(i) for i in range(8)]
    assert got == want
