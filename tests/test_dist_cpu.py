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
    # bench.py's per-rank report: every rank sees every rank's row, in rank order
    seen = udist.all_gather_floats([float(r), float(len(idx)), 0.5 * r])
    assert seen == [[float(k), float(udist.shard_count(n_total, k, w)), 0.5 * k] for k in range(w)], seen
    assert udist.backend_name() == "gloo"
    udist.barrier()
    if r == 0:
        q.put((full.numpy(), t))
    dist.destroy_process_group()


def _run_world(world, n_total):
    from uvad_amd.synth import synth_pcm
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, tmax = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = np.stack([np.abs(synth_pcm(1, 480, seed=1000, first=i)[0]).reshape(3, 160).mean(1) for i in range(n_total)])
    assert np.array_equal(full, want.astype(np.float32))      # n-way result == 1-way result, bit for bit
    assert tmax == float(world)


@pytest.mark.parametrize("n_total", [7, 8, 1])
def test_shard_gather_world2(n_total):
    _run_world(2, n_total)


def test_shard_scatter_gather_world8_uneven_4099_rows():
    """The driver's scaling run is N = 8: eight gloo ranks here (RCCL needs the node), cfg 4's batch of 4 096 utterances plus 3 so that
    the shards are uneven (three ranks own 513 rows, five 512): scatter from the root (both forms, twice into one receive buffer),
    gather back, the per-rank all-gather of bench.py's `ranks` object, max over ranks -- every row bit-identical to the unsharded run."""
    _run_world(8, 4096 + 3)


def test_shard_indices_partition():
    from uvad_amd.dist import shard_count, shard_indices
    for n in (0, 1, 5, 256, 4097):
        for w in (1, 2, 4, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert [len(p) for p in parts] == [shard_count(n, r, w) for r in range(w)]


def test_bench_gpus_n_outside_torchrun_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment must start its ranks itself (a child
    `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...` on 127.0.0.1) instead of dying on WORLD_SIZE != --gpus,
    and relay the children's exit code.  Here there is no GPU, so the ranks stop at "bench.py needs a GPU" -- what is checked is
    that BOTH ranks were started by the self-launch and that their failure comes back as a non-zero exit code (the successful path,
    with the JSON line, runs in the -m gpu rehearsal)."""
    import subprocess, sys
    if torch.cuda.is_available():
        pytest.skip("covered by tests/test_gpu_scale.py::test_bench_two_ranks_on_one_gpu_gloo_rehearsal on a GPU box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["UVAD_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert "without a torchrun environment" in p.stderr and "--nproc-per-node 2" in p.stderr, p.stderr[-1500:]
    assert p.stderr.count("bench.py needs a GPU") >= 2, p.stderr[-1500:]          # both ranks ran bench.py's main()
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
