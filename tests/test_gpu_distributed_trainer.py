"""The real GAN training step across two ranks (train.py:75-85,133-184): two processes share the
one test GPU, each runs `HotPathTrainer.step` twice on its half of a batch, gradients all-reduced (mean) per optimizer
before clipping.  With t and every noise draw pinned per sample, the reduced gradients of both steps and the weights
after them must equal a single process stepping on the concatenated batch.  Step 2 is the interesting one: its D
gradient contains the reference's D-gradient leak (the G-phase backward of step 1 deposits into D's .grad, train.py
never clears it before the next D phase), which stays local on each rank until the step-2 D all-reduce sums it.
gloo carries the all-reduce here (RCCL needs one GPU per rank; the driver's multi-GPU runs use nccl) -- through the same
chunked path: the k=3 weight gradients enter the collective behind an event while the backward is still running
(GradBucket.all_reduce_chunk_async), the rest follows in all_reduce_mean.  A second test runs one trainer step with the
"nccl" backend (RCCL) on a world of one, collectives forced: the flat bucket, the side stream and the event hand-off are
legal for RCCL, and the result equals the step without any process group."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import golden, hot_path_configs, write_stats, load_seeded, Tape

pytestmark = pytest.mark.gpu
B, L, STEPS = 4, 64, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    gen = torch.Generator().manual_seed(77)
    mel = torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5
    cond = torch.randn(B, L, 256, generator=gen)
    spk = torch.randn(B, 256, generator=gen)
    # per step: D-phase forward (t, 3 noises) then G-phase forward (t, 3 noises)  (train.py:133,153)
    ts = [torch.randint(0, 4, (B,), generator=gen) for _ in range(2 * STEPS)]
    noises = [torch.randn(B, 1, 80, L, generator=gen) for _ in range(6 * STEPS)]
    return mel, cond, spk, ts, noises


def _run(mg, manifest, stats_dir, lo, hi, always_exchange=False, chunks=None):
    """STEPS trainer steps on samples [lo, hi); returns reduced gradients per update and the final weights."""
    args, pre, mc, tr = hot_path_configs("naive", 4, multi_speaker=True, stats_dir=stats_dir)
    G = mg.GaussianDiffusion(args, pre, mc, tr)
    D = mg.JCUDiscriminator(pre, mc, tr)
    load_seeded(G, manifest, "diffusion_naive_ms1", 32)
    load_seeded(D, manifest, "jcu_ms1", 42)
    with torch.no_grad():   # the fixture recipe leaves output_projection at its zero init: make the path live
        G.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
    G, D = G.cuda(), D.cuda()
    mel, cond, spk, ts, noises = _data()
    G.t_fn = Tape([t[lo:hi].numpy() for t in ts])
    G.noise_fn = Tape([n[lo:hi].numpy() for n in noises])
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    trainer.bucketG.always_exchange = trainer.bucketD.always_exchange = always_exchange
    if chunks is not None:      # count the early chunks that went out
        inner = trainer.bucketG.all_reduce_chunk_async
        trainer.bucketG.all_reduce_chunk_async = lambda *a, **k: chunks.append(a[:2]) or inner(*a, **k)
    seen = []
    trainer.grad_hook = lambda name, bucket: seen.append((name, bucket.flat.detach().cpu().clone()))
    pad = torch.zeros(hi - lo, L, dtype=torch.bool, device="cuda")
    for _ in range(STEPS):
        out = trainer.step(mel[lo:hi].cuda(), cond[lo:hi].cuda(), spk[lo:hi].cuda(), pad)
        assert all(torch.isfinite(v).all() for v in out.values())
    assert G.t_fn.i == 2 * STEPS and G.noise_fn.i == 6 * STEPS
    weights = {"G." + k: v.detach().cpu() for k, v in G.named_parameters()}
    weights.update({"D." + k: v.detach().cpu() for k, v in D.named_parameters()})
    return seen, weights


def _worker(rank, world, port, stats_dir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        import mixgan_tts_amd as mg
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest.json")) as f:
            manifest = json.load(f)
        per = B // world
        chunks = []
        seen, weights = _run(mg, manifest, stats_dir, rank * per, (rank + 1) * per, chunks=chunks)
        assert len(chunks) == STEPS and all(hi - lo > 7_000_000 for lo, hi in chunks), chunks   # one early chunk per G update
        # numpy arrays are pickled by value (shared-memory tensors would need this process to outlive the receive)
        q.put((rank, [(n, f.numpy()) for n, f in seen], {k: v.numpy() for k, v in weights.items()}))
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_steps_equal_single_process(manifest, tmp_path):
    import mixgan_tts_amd as mg
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"], n_speakers=5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, stats, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t_: t_[0])
    got = [(r, [(n, torch.from_numpy(f)) for n, f in seen], {k: torch.from_numpy(v) for k, v in w.items()})
           for r, seen, w in got]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ref_seen, ref_w = _run(mg, manifest, stats, 0, B)          # single process, whole batch
    assert [n for n, _ in ref_seen] == ["D", "G"] * STEPS
    for r in range(2):
        seen, weights = got[r][1], got[r][2]
        assert [n for n, _ in seen] == ["D", "G"] * STEPS
        for i, ((name, flat), (_, ref)) in enumerate(zip(seen, ref_seen)):
            # step 1 is pure summation-order noise; step 2 starts from weights that already differ in the last bits
            tol = 2e-4 if i < 2 else 2e-3
            err = (flat - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
            assert err <= tol, "rank %d update %d (%s): reduced gradient differs by %.2e" % (r, i, name, err)
        # Adam turns a gradient into +-lr however small it is, so compare the UPDATE in the L2 sense
        num, den = _update_distance(weights, ref_w, _initial_weights(mg, manifest, stats))
        assert den > 0 and (num / den) ** 0.5 <= 2e-2, "rank %d weight update differs: %.3e" % (r, (num / den) ** 0.5)
    # both ranks hold identical weights (same reduced gradients, same optimizer state)
    for k in got[0][2]:
        assert torch.equal(got[0][2][k], got[1][2][k]), k


def _initial_weights(mg, manifest, stats_dir):
    args, pre, mc, tr = hot_path_configs("naive", 4, multi_speaker=True, stats_dir=stats_dir)
    G = mg.GaussianDiffusion(args, pre, mc, tr)
    D = mg.JCUDiscriminator(pre, mc, tr)
    load_seeded(G, manifest, "diffusion_naive_ms1", 32)
    load_seeded(D, manifest, "jcu_ms1", 42)
    with torch.no_grad():
        G.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
    w = {"G." + k: v.detach().clone() for k, v in G.named_parameters()}
    w.update({"D." + k: v.detach().clone() for k, v in D.named_parameters()})
    return w


def _update_distance(weights, ref_w, init):
    num = den = 0.0
    for k, w in weights.items():
        d_ref = (ref_w[k] - init[k]).double()
        d_got = (w - init[k]).double()
        num += float((d_got - d_ref).pow(2).sum())
        den += float(d_ref.pow(2).sum())
    return num, den


def _rccl_worker(port, stats_dir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)      # "nccl" IS RCCL on ROCm
    try:
        import json
        import mixgan_tts_amd as mg
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest.json")) as f:
            manifest = json.load(f)
        assert dist.get_backend() == "nccl"
        chunks = []
        seen, weights = _run(mg, manifest, stats_dir, 0, B, always_exchange=True, chunks=chunks)
        torch.cuda.synchronize()
        q.put(("ok", len(chunks), [(n, f.numpy()) for n, f in seen], {k: v.numpy() for k, v in weights.items()}))
    except Exception as exc:      # report instead of hanging the parent on q.get
        q.put(("error", repr(exc), None, None))
        raise
    finally:
        dist.destroy_process_group()


def test_rccl_world_of_one_runs_the_bucketed_exchange(manifest, tmp_path):
    """init_process_group("nccl") + HotPathTrainer.step x2 with the collectives forced on a world of one: RCCL loads and
    accepts the flat bucket slices, the side-stream launch behind the backward's event and the stream-level waits; the
    reduced gradients and the weights equal the same steps without a process group (sum over one rank, divided by 1)."""
    import mixgan_tts_amd as mg
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"], n_speakers=5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), stats, q))
    p.start()
    status, n_chunks, seen, weights = q.get(timeout=600)
    p.join(120)
    assert status == "ok", n_chunks
    assert p.exitcode == 0
    assert n_chunks == STEPS
    ref_seen, ref_w = _run(mg, manifest, stats, 0, B)
    assert [n for n, _ in seen] == [n for n, _ in ref_seen] == ["D", "G"] * STEPS
    for i, ((name, flat), (_, ref)) in enumerate(zip(seen, ref_seen)):
        tol = 2e-5 if i < 2 else 2e-3      # step 2 starts from weights that already differ in the last bits
        err = (torch.from_numpy(flat) - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert err <= tol, "update %d (%s): %.2e" % (i, name, err)
    num, den = _update_distance({k: torch.from_numpy(v) for k, v in weights.items()}, ref_w,
                                _initial_weights(mg, manifest, stats))
    assert den > 0 and (num / den) ** 0.5 <= 2e-2
