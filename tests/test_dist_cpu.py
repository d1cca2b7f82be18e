"""world_size-2 gloo test of the utterance sharding used by bench.py --gpus N and by the multi-GPU sweep."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from uvad_amd import dist as udist
    import numpy as np
    from uvad_amd.synth import synth_pcm
    r, lr, w = udist.init(backend="gloo")
    idx = udist.shard_indices(n_total, r, w)
    assert len(idx) == udist.shard_count(n_total, r, w)
    # each rank regenerates exactly its utterances; "result" = a per-utterance statistic over 3 frames
    local = torch.from_numpy(np.stack([np.abs(synth_pcm(1, 480, seed=1000, first=i)[0]).reshape(3, 160).mean(1) for i in idx])) \
        if idx else torch.zeros(0, 3)
    # root-resident mode: scatter from rank 0 must hand every rank exactly the rows it would have generated
    root = torch.from_numpy(np.stack([np.abs(synth_pcm(1, 480, seed=1000, first=i)[0]).reshape(3, 160).mean(1) for i in range(n_total)])) if r == 0 else None
    got = udist.scatter_rows(root, n_total, r, w, like=torch.zeros(1, 3, dtype=torch.float32))
    assert got.shape[0] == len(idx) and (len(idx) == 0 or torch.equal(got.to(local.dtype), local))
    # the per-step form bench.py --scatter uses: the root re-orders the corpus ONCE, then scatters views (uneven shards included)
    pre = udist.preshard_rows(root, n_total, w) if r == 0 else None
    if r == 0:
        per = (n_total + w - 1) // w
        assert tuple(pre.shape) == (w, per, 3)
        for rr in range(w):
            assert torch.equal(pre[rr, : udist.shard_count(n_total, rr, w)], root[udist.shard_indices(n_total, rr, w)])
    buf = torch.full(((n_total + w - 1) // w, 3), -1.0, dtype=torch.float32)
    for _ in range(2):   # twice into the same receive buffer, as the double-buffered loop does
        got2 = udist.scatter_rows(pre, n_total, r, w, like=torch.zeros(1, 3, dtype=torch.float32), presharded=True, out=buf)
        assert got2.shape[0] == len(idx) and (len(idx) == 0 or (got2.data_ptr() == buf.data_ptr() and torch.equal(got2.to(local.dtype), local)))
    full = udist.gather_rows(local, n_total, r, w)
    t = udist.max_over_ranks(float(r + 1))
    udist.barrier()
    if r == 0:
        q.put((full.numpy(), t))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8, 1])
def test_shard_gather_world2(n_total):
    from uvad_amd.synth import synth_pcm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, tmax = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.stack([np.abs(synth_pcm(1, 480, seed=1000, first=i)[0]).reshape(3, 160).mean(1) for i in range(n_total)])
    assert np.array_equal(full, want.astype(np.float32))      # 2-way result == 1-way result, bit for bit
    assert tmax == 2.0


def test_shard_indices_partition():
    from uvad_amd.dist import shard_count, shard_indices
    for n in (0, 1, 5, 256, 4097):
        for w in (1, 2, 4, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert [len(p) for p in parts] == [shard_count(n, r, w) for r in range(w)]
