#!/usr/bin/env python3
"""BASELINE cfg 4: corpus sweep with the utterance batch sharded over the GPUs of one node (rank r owns
utterances i = r mod n; PCM is synthesised on the device per shard, never stored; no data-path collective;
the per-frame outputs of the LAST batch are gathered to rank 0 with RCCL to exercise the gather).
    python tools/run_cfg4.py --hours 10                                  (1 GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/run_cfg4.py --hours 10000
Full 10 k hours = 3.6e9 frames; pass a smaller --hours for a sample, the rate is what is reported."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import uvad_amd
from uvad_amd import dist as udist
from uvad_amd.synth import seed_weights

ap = argparse.ArgumentParser()
ap.add_argument("--hours", type=float, default=10.0)
ap.add_argument("--batch", type=int, default=4096, help="utterances per global batch (10 s each)")
ap.add_argument("--reproducible", action="store_true",
                help="pin every rank to the recurrent form the GLOBAL batch would use on one GPU: bit-identical logits for any number of GPUs")
args = ap.parse_args()
rank, local_rank, world = udist.init()
dev = torch.device("cuda", local_rank); torch.cuda.set_device(dev)
m = uvad_amd.PyanNet2(encoding_dim=64); m.build(); seed_weights(m, 1234, 4.0)
m.attach_fbank(uvad_amd.FbankConfig(num_filters=64, window_type="hamming")); m = m.to(dev).eval()
rt = m.runtime(dev)
if args.reproducible:
    rt.set_recurrent_tile(rt.recurrent_tile_for(args.batch))
S = 160000
n_utt = int(args.hours * 3600 / 10)
n_batches = max(1, n_utt // args.batch)
local = udist.shard_count(args.batch, rank, world)
g = torch.Generator(device=dev)
def shard_pcm(batch_idx):
    g.manual_seed(42 * 1_000_003 + batch_idx * world + rank)      # Philox stream keyed by (seed, batch, rank)
    return 0.1 * torch.randn(local, S, generator=g, device=dev)
pcm = shard_pcm(0); rt.forward(pcm, want_probs=False); torch.cuda.synchronize(dev); udist.barrier()
t0 = time.perf_counter()
for b in range(n_batches):
    pcm = shard_pcm(b)                                                # generation is inside the timed region
    logits, _ = rt.forward(pcm, want_probs=False)
torch.cuda.synchronize(dev); udist.barrier()
dt = udist.max_over_ranks(time.perf_counter() - t0, device=dev if world > 1 else None)
full = udist.gather_rows(logits, args.batch, rank, world)
if rank == 0:
    frames = n_batches * args.batch * 1000
    print(json.dumps({"config": f"{args.hours} h sweep, global batch {args.batch} x 10 s, {world} GPU(s), utterance i -> rank i mod n",
                      "frames_per_s": frames / dt, "hours_of_audio_per_s": frames / 100 / 3600 / dt, "seconds": dt,
                      "gathered_shape": list(full.shape)}))
