"""GPU parity of the losses and of the GAN training step (train.py:91-184 restricted to the hot
path): gradients of both phases against torch.autograd on the CPU oracle with identical injected
t / noise, plus the reference's D-gradient leak and optimizer wiring."""
import numpy as np
import pytest
import torch

from helpers import golden, T, seeded, assert_close, hot_path_configs, write_stats, load_seeded, Tape
from oracle import refmath as R, schedule as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    m.lib()
    return m


def test_losses_golden(mg):
    g = golden("mel_loss")
    pred = T(g["pred"]).cuda().requires_grad_()
    v = mg.losses.get_mel_loss(pred, T(g["targ"]).cuda(), T(g["pad"]).cuda())
    assert abs(v.item() - float(g["loss"])) < 2e-6
    v.backward()
    pr = T(g["pred"]).requires_grad_()
    R.mel_l1(pr, T(g["targ"]), T(g["pad"])).backward()
    assert_close(pred.grad.cpu(), pr.grad, 1e-6, "mel L1 gradient")
    # LSGAN / FM scalars + gradients vs the oracle on the fixture's feature maps
    j = golden("jcu_ms0_L37")
    maps = {k: [T(j["%s%d" % (k, i)]) for i in range(5)] for k in ("fc", "fu", "rc", "ru")}
    dm = {k: [m.cuda().requires_grad_() for m in v] for k, v in maps.items()}
    cm = {k: [m.clone().requires_grad_() for m in v] for k, v in maps.items()}
    d_fn, g_fn = mg.losses.get_adversarial_losses_fn("lsgan")
    r, f = d_fn(dm["rc"][-1], dm["ru"][-1], dm["fc"][-1], dm["fu"][-1])
    adv = g_fn(dm["fc"][-1], dm["fu"][-1])
    fm = mg.losses.get_fm_loss(dm["rc"], dm["ru"], dm["fc"], dm["fu"])
    for a, k in ((r, "r_loss"), (f, "f_loss"), (adv, "adv"), (fm, "fm")):
        assert abs(a.item() - float(j[k])) <= 2e-6 * max(1.0, abs(float(j[k]))), k
    (r + f + adv + 10.0 * fm).backward()
    rr, ff = R.d_loss(cm["rc"][-1], cm["ru"][-1], cm["fc"][-1], cm["fu"][-1])
    (rr + ff + R.g_loss(cm["fc"][-1], cm["fu"][-1]) + 10.0 * R.fm_loss(cm["rc"], cm["ru"], cm["fc"], cm["fu"])).backward()
    for k in ("fc", "fu"):
        for i in range(5):
            assert_close(dm[k][i].grad.cpu(), cm[k][i].grad, 1e-6, "%s%d grad" % (k, i))
    # the fused sums the trainer uses (losses.weighted_means: one launch each way) against the same fixture
    fm_ = {k: [m.cuda().requires_grad_() for m in v] for k, v in maps.items()}
    d_tot, d_r, d_f = mg.losses.d_loss_total(fm_["rc"][-1], fm_["ru"][-1], fm_["fc"][-1], fm_["fu"][-1])
    g_tot, adv2, fm2 = mg.losses.g_adv_fm_total(fm_["rc"], fm_["ru"], fm_["fc"], fm_["fu"], 10.0)
    for a, ref in ((d_r, float(j["r_loss"])), (d_f, float(j["f_loss"])), (adv2, float(j["adv"])), (fm2, 10.0 * float(j["fm"])),
                   (d_tot, float(j["r_loss"]) + float(j["f_loss"])), (g_tot, float(j["adv"]) + 10.0 * float(j["fm"]))):
        assert abs(a.item() - ref) <= 2e-6 * max(1.0, abs(ref))
    (d_tot + g_tot).backward()
    for k in ("fc", "fu"):
        for i in range(5):
            assert_close(fm_[k][i].grad.cpu(), cm[k][i].grad, 1e-6, "fused %s%d grad" % (k, i))
    assert all(m.grad is None for m in fm_["rc"][:-1] + fm_["ru"][:-1]), "real maps are targets of the FM term: no gradient"
    again = mg.losses.g_adv_fm_total(fm_["rc"], fm_["ru"], fm_["fc"], fm_["fu"], 10.0)[0]
    assert torch.equal(again, g_tot), "fixed summation order"
    # ... and on the WHOLE maps of one discriminator pass over [fake; real] (what the trainer feeds them): same scalars,
    # the fake rows' gradients as above, zero (G phase) / the real-logit gradients (D phase) on the real rows
    Bf = maps["fc"][0].shape[0]
    whole = {k: [torch.cat([f, r]).cuda().requires_grad_() for f, r in zip(maps["f" + k], maps["r" + k])] for k in ("c", "u")}
    d2, d2r, d2f = mg.losses.d_loss_total_2b(whole["c"][-1], whole["u"][-1], Bf)
    g2, adv3, fm3 = mg.losses.g_adv_fm_total_2b(whole["c"], whole["u"], Bf, 10.0)
    for a, ref in ((d2, d_tot), (d2r, d_r), (d2f, d_f), (g2, g_tot), (adv3, adv2), (fm3, fm2)):
        assert abs(a.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    g2.backward()
    for k in ("c", "u"):
        for i in range(5):
            assert_close(whole[k][i].grad[:Bf].cpu(), cm["f" + k][i].grad if i < 4 else _adv_only(cm, k), 1e-6, "2B fake rows %s%d" % (k, i))
            assert float(whole[k][i].grad[Bf:].abs().sum()) == 0.0, "real rows are targets"
    for k in ("c", "u"):
        for m in whole[k]:
            m.grad = None
    d2 = mg.losses.d_loss_total_2b(whole["c"][-1], whole["u"][-1], Bf)[0]
    d2.backward()
    dr = {k: maps[k][-1].clone().requires_grad_() for k in ("fc", "fu", "rc", "ru")}
    sum(R.d_loss(dr["rc"], dr["ru"], dr["fc"], dr["fu"])).backward()
    for k in ("c", "u"):
        assert_close(whole[k][-1].grad[:Bf].cpu(), dr["f" + k].grad, 1e-6, "D phase, fake logits")
        assert_close(whole[k][-1].grad[Bf:].cpu(), dr["r" + k].grad, 1e-6, "D phase, real logits")
        assert all(m.grad is None for m in whole[k][:-1])


def _adv_only(cm, k):
    """Gradient of the last (logit) map from the combined oracle loss of test_losses_golden: r + f + adv + 10 fm -- the
    2B generator loss holds only adv; recompute that part."""
    x = cm["f" + k][4].detach().clone().requires_grad_()
    other = cm["f" + ("u" if k == "c" else "c")][4].detach()
    (R.g_loss(x, other) if k == "c" else R.g_loss(other, x)).backward()
    return x.grad


def _setup(mg, manifest, tmp_path, B=3, L=40):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
    G = mg.GaussianDiffusion(args, pre, mc, tr)
    load_seeded(G, manifest, "diffusion_naive_ms0", 61)
    D = mg.JCUDiscriminator(pre, mc, tr)
    load_seeded(D, manifest, "jcu_ms0", 62)
    WG, _ = seeded(manifest, "diffusion_naive_ms0", 61, requires_grad=True)
    WD, _ = seeded(manifest, "jcu_ms0", 62, requires_grad=True)
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    buf["spec_min"], buf["spec_max"] = T(e["spec_min"])[None, None], T(e["spec_max"])[None, None]
    gen = torch.Generator().manual_seed(7)
    lens = torch.tensor([L, L - 9, L - 3])[:B]
    pad = torch.arange(L)[None, :] >= lens[:, None]
    mel = (torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5).masked_fill(pad.unsqueeze(-1), 0.0)
    cond = torch.randn(B, L, 256, generator=gen)
    tapes = [[torch.tensor([2, 0, 3])[:B]] + [torch.randn(B, 1, 80, L, generator=gen) for _ in range(3)] for _ in range(2)]
    return G.cuda(), D.cuda(), WG, WD, buf, mel, cond, pad, tapes, tr, mc


def test_training_step_gradients_vs_oracle(mg, manifest, tmp_path):
    G, D, WG, WD, buf, mel, cond, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path)
    lam = tr["loss"]["lambda_fm"]
    # ---------------- oracle: D phase then G phase with torch.autograd on CPU
    def oracle_phase(tape_items, d_phase):
        tape = R.NoiseTape(tape_items)
        c = cond.clone().requires_grad_()
        x0, x_t, x_prev, x_pp, t = R.diffusion_forward(WG, buf, "naive", 4, mel, c, None, pad, None, tape)
        if d_phase:
            fc, fu = R.jcu_forward(WD, x_t.detach(), x_pp.detach(), None, t)
            rc, ru = R.jcu_forward(WD, x_t.detach(), x_prev.detach(), None, t)
            r, f = R.d_loss(rc[-1], ru[-1], fc[-1], fu[-1])
            return r + f, c
        fc, fu = R.jcu_forward(WD, x_t, x_pp, None, t)
        rc, ru = R.jcu_forward(WD, x_t, x_prev, None, t)
        loss = (R.g_loss(fc[-1], fu[-1]) + R.mel_l1(R.denorm_spec(x0, buf["spec_min"], buf["spec_max"]), mel, pad)
                + lam * R.fm_loss(rc, ru, fc, fu))
        return loss, c

    ld, _ = oracle_phase(tapes[0], True)
    ld.backward()
    ref_d = {k: v.grad.clone() for k, v in WD.items()}
    for v in list(WD.values()) + list(WG.values()):
        v.grad = None
    lg, c_ref = oracle_phase(tapes[1], False)
    lg.backward()
    ref_g = {k: v.grad.clone() for k, v in WG.items()}
    ref_leak = {k: v.grad.clone() for k, v in WD.items()}

    # ---------------- product: the same two phases, gradients inspected before the optimizers run
    d_fn, g_fn = mg.losses.get_adversarial_losses_fn("lsgan")
    melc, condc, padc = mel.cuda(), cond.cuda(), pad.cuda()
    G.t_fn, G.noise_fn = Tape([tapes[0][0].numpy()]), Tape([a.numpy() for a in tapes[0][1:]])
    x0, x_t, x_prev, x_pp, t = G(melc, condc.clone().requires_grad_(), None, padc)
    fc, fu = D(x_t.detach(), x_pp.detach(), None, t)
    rc, ru = D(x_t.detach(), x_prev.detach(), None, t)
    r, f = d_fn(rc[-1], ru[-1], fc[-1], fu[-1])
    assert abs((r + f).item() - ld.item()) < 1e-5 * max(1, abs(ld.item()))
    (r + f).backward()
    for k, p in D.named_parameters():
        assert_close(p.grad.cpu(), ref_d[k], 1e-4, "D-phase grad " + k)
    D.zero_grad()
    G.zero_grad()
    G.t_fn, G.noise_fn = Tape([tapes[1][0].numpy()]), Tape([a.numpy() for a in tapes[1][1:]])
    cg = condc.clone().requires_grad_()
    x0, x_t, x_prev, x_pp, t = G(melc, cg, None, padc)
    fc, fu = D(x_t, x_pp, None, t)
    rc, ru = D(x_t, x_prev, None, t)
    loss = g_fn(fc[-1], fu[-1]) + mg.losses.get_mel_loss(G.denorm_spec(x0), melc, padc) + lam * mg.losses.get_fm_loss(rc, ru, fc, fu)
    assert abs(loss.item() - lg.item()) < 1e-5 * max(1, abs(lg.item()))
    loss.backward()
    assert_close(cg.grad.cpu(), c_ref.grad, 1e-4, "G-phase d_cond")
    for k, p in G.named_parameters():
        assert_close(p.grad.cpu(), ref_g[k], 2e-4, "G-phase grad " + k)
    for k, p in D.named_parameters():
        assert_close(p.grad.cpu(), ref_leak[k], 2e-4, "leaked D grad " + k)


def test_trainer_step_runs_and_leaks_d_grads(mg, manifest, tmp_path):
    G, D, WG, WD, buf, mel, cond, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path)
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    before_g = [p.detach().clone() for p in G.parameters()]
    before_d = [p.detach().clone() for p in D.parameters()]
    out = trainer.step(mel.cuda(), cond.cuda(), None, pad.cuda())
    assert all(torch.isfinite(v).all() for v in out.values())
    assert any(not torch.equal(a, b) for a, b in zip(before_g, G.parameters()))
    assert any(not torch.equal(a, b) for a, b in zip(before_d, D.parameters()))
    # train.py:84-85 + :159-160: zero_grad runs after step, the G-phase backward then leaves gradients on D
    assert all(p.grad is None for p in G.parameters())
    assert all(p.grad is not None for p in D.parameters())
    out2 = trainer.step(mel.cuda(), cond.cuda(), None, pad.cuda())
    assert all(torch.isfinite(v).all() for v in out2.values())
    trainer.end_epoch()
    assert abs(trainer.optG.param_groups[0]["lr"] - 1e-4 * 0.999) < 1e-12


def test_trainer_fused_and_per_term_losses_agree(mg, manifest, tmp_path):
    """HotPathTrainer.fused_losses (one launch per loss sum) against the per-term path through d_loss_fn / g_loss_fn /
    get_fm_loss: same logged losses, same gradients reaching the optimizers (the t / noise draws pinned by the seed)."""
    seen = {}
    for fused in (True, False):
        G, D, WG, WD, buf, mel, cond, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path)
        trainer = mg.HotPathTrainer(G, D, tr, mc)
        trainer.fused_losses = fused
        grads = {}
        trainer.grad_hook = lambda name, bucket: grads.__setitem__(name, bucket.flat.clone())
        torch.manual_seed(5)
        out = trainer.step(mel.cuda(), cond.cuda(), None, pad.cuda())
        seen[fused] = (out, grads)
    for k in ("d_loss", "adv_loss", "mel_loss", "fm_loss"):
        a, b = seen[True][0][k].item(), seen[False][0][k].item()
        assert abs(a - b) <= 2e-6 * max(1.0, abs(b)), k
    for name in ("D", "G"):
        assert_close(seen[True][1][name].cpu(), seen[False][1][name].cpu(), 2e-5, "bucket " + name)


def test_trainer_paired_and_separate_forwards_agree(mg, manifest, tmp_path):
    """HotPathTrainer.pair_forwards: the D-phase and G-phase generator forwards in one launch (the second one's t /
    noise drawn early, in the same order) against two launches -- same logged losses, same gradients reaching the
    optimizers, over two steps (the second step reuses both workspaces)."""
    seen = {}
    for paired in (True, False):
        G, D, WG, WD, buf, mel, cond, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path)
        trainer = mg.HotPathTrainer(G, D, tr, mc)
        trainer.pair_forwards = paired
        grads = []
        trainer.grad_hook = lambda name, bucket: grads.append((name, bucket.flat.clone()))
        torch.manual_seed(5)
        outs = [trainer.step(mel.cuda(), cond.cuda(), None, pad.cuda()) for _ in range(2)]
        seen[paired] = (outs, grads)
        assert G._pair_stash is None                     # consumed by the G-phase forward
    for o1, o2 in zip(seen[True][0], seen[False][0]):
        for k in ("d_loss", "adv_loss", "mel_loss", "fm_loss"):
            a, b = o1[k].item(), o2[k].item()
            assert abs(a - b) <= 5e-5 * max(1.0, abs(b)), k
    for (n1, g1), (n2, g2) in zip(seen[True][1], seen[False][1]):
        assert n1 == n2
        assert_close(g1.cpu(), g2.cpu(), 1e-4, "bucket " + n1)


def test_training_learns_a_solvable_task(mg, tmp_path):
    """End to end through every kernel of the step (both forwards, data and weight gradients, fused losses, flat clip +
    Adam, weight repack): with the normalised target in the conditioner's first 80 channels the denoiser can read x0
    off it, and 200 steps must take the mel L1 well below where it started (tools/dbg/train_learns.py: 1.97 -> 0.76
    in 300 steps)."""
    e = golden("elementwise")
    stats = write_stats(tmp_path, [-11.5] * 80, [2.0] * 80)
    args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
    tr = dict(tr)
    tr["optimizer"] = dict(tr["optimizer"], init_lr_G=1e-3)
    torch.manual_seed(0)
    G = mg.GaussianDiffusion(args, pre, mc, tr).cuda()
    D = mg.JCUDiscriminator(pre, mc, tr).cuda()
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    gen = torch.Generator(device="cuda").manual_seed(0)
    B, L = 8, 256
    losses_seen = []
    for step in range(200):
        z = torch.randn(B, L // 8 + 1, 80, device="cuda", generator=gen)
        mel = torch.nn.functional.interpolate(z.transpose(1, 2), size=L, mode="linear").transpose(1, 2) * 3.0 - 5.0
        mel = mel.clamp(-11.5, 2.0).contiguous()
        cond = torch.zeros(B, L, 256, device="cuda")
        cond[:, :, :80] = (mel + 11.5) / 13.5 * 2 - 1
        out = trainer.step(mel, cond, None, torch.zeros(B, L, dtype=torch.bool, device="cuda"))
        losses_seen.append(out["mel_loss"])
    vals = torch.stack(losses_seen).cpu()
    assert torch.isfinite(vals).all()
    start, end = vals[:5].mean().item(), vals[-5:].mean().item()
    assert end < 0.75 * start, (start, end)


def test_checkpoint_resume_through_the_trainer(mg, manifest, tmp_path):
    """utils/model.py:12-53 + train.py:252-267 around the HIP trainer: get_model() -> HotPathTrainer(resume=...) -> two
    steps + an epoch -> save_checkpoint with the trainer's optimizers -> get_model(restore_step) hands back stock Adam
    objects whose state a second trainer adopts; both trainers then take the same third step."""
    import types
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
    tr = dict(tr)
    tr["path"] = {"ckpt_path": str(tmp_path / "ckpt")}
    tr["step"] = {"total_step_aux": 7}
    tr["optimizer_fs2"] = {"betas": [0.9, 0.98], "eps": 1e-9, "weight_decay": 0.0, "warm_up_step": 4000,
                           "anneal_steps": [300000], "anneal_rate": 0.3}
    configs = (pre, mc, tr)
    gen = torch.Generator().manual_seed(3)
    B, L = 2, 48
    mel = (torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5).cuda()
    cond = torch.randn(B, L, 256, generator=gen).cuda()
    pad = torch.zeros(B, L, dtype=torch.bool).cuda()

    def trainer_for(restore_step):
        a = types.SimpleNamespace(model="naive", restore_step=restore_step)
        model, D, f, g, d, sg, sd, epoch = mg.get_model(a, configs, "cuda", train=True)
        t = mg.HotPathTrainer(model.diffusion, D, tr, mc, g_param_order=list(model.parameters()), resume=(g, d, sg, sd))
        return model, D, f, t, epoch
    model, D, f, t1, _ = trainer_for(0)
    with torch.no_grad():   # the output projection starts at zero (model/modules.py:418): give the gradients something to do
        model.diffusion.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=None)
    for s_ in range(2):
        torch.manual_seed(10 + s_)
        t1.step(mel, cond, None, pad)
    t1.end_epoch()
    mg.save_checkpoint(tr, 2, 3, model, D, f, t1.optG, t1.optD, t1.sdlG, t1.sdlD)
    m2, D2, f2, t2, epoch = trainer_for(2)
    assert epoch == 3
    assert t2.optG.param_groups[0]["lr"] == t1.optG.param_groups[0]["lr"] == pytest.approx(1e-4 * 0.999)
    assert t2.sdlG.last_epoch == t1.sdlG.last_epoch == 1
    for a_, b_ in ((t1.optG, t2.optG), (t1.optD, t2.optD)):
        assert float(b_._steps) == float(a_._steps) == 2
        assert torch.equal(a_.flat_m, b_.flat_m) and torch.equal(a_.flat_v, b_.flat_v) and torch.equal(a_.flat_p, b_.flat_p)
    # a restarted process has no leaked D gradients from the previous G phase (they are not checkpointed, in the
    # reference either): drop them on the running side so that both take the same step
    for p in D.parameters():
        p.grad = None
    for tt in (t1, t2):
        torch.manual_seed(99)
        tt.step(mel, cond, None, pad)
    # not bitwise: a few reductions on the path combine partial sums with atomics (mel L1, the small per-sample GEMMs)
    for (k, p), q in zip(model.named_parameters(), m2.parameters()):
        assert_close(p.detach().cpu(), q.detach().cpu(), 1e-5, k)
    for (k, p), q in zip(D.named_parameters(), D2.parameters()):
        assert_close(p.detach().cpu(), q.detach().cpu(), 1e-5, k)


def test_batch_shard_gradients_average_to_the_full_batch(mg, manifest, tmp_path):
    """What the multi-GPU design rests on: with equal shard sizes and equal frame counts, the
    mean over ranks of the per-shard generator gradients equals the single-process gradient of the whole batch --
    checked on the real kernels with the t / noise draws of each sample pinned (two 'ranks' = two halves of B=4)."""
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
    G = mg.GaussianDiffusion(args, pre, mc, tr)
    load_seeded(G, manifest, "diffusion_naive_ms0", 61)
    with torch.no_grad():   # the fixture recipe leaves output_projection at its zero init: make the path live
        G.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
    G = G.cuda()
    B, L = 4, 96
    gen = torch.Generator().manual_seed(21)
    mel = torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5
    cond = torch.randn(B, L, 256, generator=gen)
    pad = torch.zeros(B, L, dtype=torch.bool)
    t = torch.tensor([3, 0, 2, 1])
    noises = [torch.randn(B, 1, 80, L, generator=gen) for _ in range(3)]

    def grads(lo, hi):
        G.zero_grad(set_to_none=True)
        G.t_fn = Tape([t[lo:hi].numpy()])
        G.noise_fn = Tape([n[lo:hi].numpy() for n in noises])
        c = cond[lo:hi].cuda().requires_grad_()
        x0, *_ = G(mel[lo:hi].cuda(), c, None, pad[lo:hi].cuda())
        loss = mg.losses.get_mel_loss(G.denorm_spec(x0), mel[lo:hi].cuda(), pad[lo:hi].cuda())
        loss.backward()
        return {k: p.grad.detach().clone() for k, p in G.named_parameters() if p.grad is not None}, c.grad.clone()

    full, dc_full = grads(0, B)
    h0, dc0 = grads(0, B // 2)
    h1, dc1 = grads(B // 2, B)
    assert len(full) > 100
    for k, g in full.items():
        avg = 0.5 * (h0[k] + h1[k])
        scale = g.abs().max().item() + 1e-12
        assert (avg - g).abs().max().item() <= 2e-4 * scale + 1e-9, k
    # the conditioner gradient of a sample only depends on its own shard (scaled by the shard's share of the mean)
    assert_close(torch.cat([dc0, dc1]).cpu() * 0.5, dc_full.cpu(), 2e-4, "d_cond across shards")
