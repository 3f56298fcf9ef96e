"""The trainer's boundary against train.py:75-85,131-184 and evaluate.py:70-120: the two generator forwards of a step
see DIFFERENT conditioners (the reference calls `model(*(batch[2:]))` once per phase, so a train-mode linguistic encoder
runs twice), `step_from_model` performs those two calls around the HIP step, `grad_acc_step > 1` accumulates like
`model_update`, `evaluate_step` is the same two phases without an update, and the default-constructed trainer's
optimizer state is addressed in `parameters()` order."""
import numpy as np
import pytest
import torch
from torch import nn

from helpers import golden, T, seeded, assert_close, hot_path_configs, write_stats, load_seeded, Tape
from oracle import refmath as R, schedule as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    m.lib()
    return m


def _setup(mg, manifest, tmp_path, B=3, L=40, grad_acc=1):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
    tr = dict(tr)
    tr["optimizer"] = dict(tr["optimizer"], grad_acc_step=grad_acc)
    G = mg.GaussianDiffusion(args, pre, mc, tr)
    load_seeded(G, manifest, "diffusion_naive_ms0", 61)
    D = mg.JCUDiscriminator(pre, mc, tr)
    load_seeded(D, manifest, "jcu_ms0", 62)
    with torch.no_grad():   # the fixture recipe leaves output_projection at its zero init: make the path live
        G.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
    WG = {k: v.detach().clone().requires_grad_() for k, v in G.state_dict().items() if v.dtype == torch.float32}
    WD = {k: v.detach().clone().requires_grad_() for k, v in D.state_dict().items()}
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    buf["spec_min"], buf["spec_max"] = T(e["spec_min"])[None, None], T(e["spec_max"])[None, None]
    gen = torch.Generator().manual_seed(7)
    lens = torch.tensor([L, L - 9, L - 3])[:B]
    pad = torch.arange(L)[None, :] >= lens[:, None]
    mel = (torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5).masked_fill(pad.unsqueeze(-1), 0.0)
    conds = [torch.randn(B, L, 256, generator=gen) for _ in range(2)]       # D-phase, G-phase
    tapes = [[torch.tensor([2, 0, 3])[:B]] + [torch.randn(B, 1, 80, L, generator=gen) for _ in range(3)] for _ in range(4)]
    return G.cuda(), D.cuda(), WG, WD, buf, mel, conds, pad, tapes, tr, mc


def _oracle_phase(WG, WD, buf, mel, cond, pad, tape_items, d_phase, lam):
    tape = R.NoiseTape(tape_items)
    c = cond.clone().requires_grad_()
    x0, x_t, x_prev, x_pp, t = R.diffusion_forward(WG, buf, "naive", 4, mel, c, None, pad, None, tape)
    if d_phase:
        fc, fu = R.jcu_forward(WD, x_t.detach(), x_pp.detach(), None, t)
        rc, ru = R.jcu_forward(WD, x_t.detach(), x_prev.detach(), None, t)
        r, f = R.d_loss(rc[-1], ru[-1], fc[-1], fu[-1])
        return r + f, c, {}
    fc, fu = R.jcu_forward(WD, x_t, x_pp, None, t)
    rc, ru = R.jcu_forward(WD, x_t, x_prev, None, t)
    adv = R.g_loss(fc[-1], fu[-1])
    mel_l = R.mel_l1(R.denorm_spec(x0, buf["spec_min"], buf["spec_max"]), mel, pad)
    fm = lam * R.fm_loss(rc, ru, fc, fu)
    return adv + mel_l + fm, c, {"adv_loss": adv, "mel_loss": mel_l, "fm_loss": fm}


def _tape_trainer(G, tapes):
    G.t_fn = Tape([tp[0].numpy() for tp in tapes])
    G.noise_fn = Tape([a.numpy() for tp in tapes for a in tp[1:]])


def _recording_hook(seen, G_like, D):
    """grad_hook: the reduced-gradient buckets per parameter name, and -- when the G update is about to happen -- the
    discriminator's weights as the G phase saw them (already stepped by this call's D phase)."""
    def hook(name, bucket):
        seen[name] = _bucket_grads(bucket, (G_like if name == "G" else D).named_parameters())
        if name == "G":
            seen["D after its update"] = {k: v.detach().cpu().clone().requires_grad_() for k, v in D.state_dict().items()}
    return hook


def _bucket_grads(bucket, named):
    return {k: bucket.flat[bucket.offsets[id(p)]:bucket.offsets[id(p)] + p.numel()].view_as(p).detach().cpu().clone()
            for k, p in named if id(p) in bucket.offsets}


@pytest.mark.parametrize("paired", [True, False])
def test_step_with_two_conditioners_vs_oracle(mg, manifest, tmp_path, paired):
    """D phase on cond_d, G phase on cond: d_loss and the D bucket must come from the first conditioner, the generator
    losses, the G bucket and d(cond) from the second -- against torch.autograd on the CPU oracle with the same injected
    t / noise, with both forwards in one launch (second conditioner pointer of mg_denoiser_fwd_pair) and in two."""
    G, D, WG, WD, buf, mel, conds, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path)
    lam = tr["loss"]["lambda_fm"]
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    trainer.pair_forwards = paired
    seen = {}
    trainer.grad_hook = _recording_hook(seen, G, D)
    _tape_trainer(G, tapes[:2])
    cg = conds[1].cuda().requires_grad_()
    out = trainer.step(mel.cuda(), cg, None, pad.cuda(), cond_d=conds[0].cuda())
    assert G._pair_stash is None
    assert G.noise_fn.i == 6 and G.t_fn.i == 2
    # the oracle: D phase on the first conditioner; the G phase sees the discriminator AFTER its update (train.py:146,156)
    ld, _, _ = _oracle_phase(WG, WD, buf, mel, conds[0], pad, tapes[0], True, lam)
    ld.backward()
    ref_d = {k: v.grad.clone() for k, v in WD.items()}
    for v in WG.values():
        v.grad = None
    lg, c_ref, parts = _oracle_phase(WG, seen["D after its update"], buf, mel, conds[1], pad, tapes[1], False, lam)
    lg.backward()
    ref_g = {k: v.grad.clone() for k, v in WG.items() if v.grad is not None}
    assert abs(out["d_loss"].item() - ld.item()) < 1e-5 * max(1.0, abs(ld.item()))
    for k, v in parts.items():
        assert abs(out[k].item() - v.item()) < 2e-5 * max(1.0, abs(v.item())), k
    for k, g in ref_d.items():
        assert_close(seen["D"][k], g, 1e-4, "D bucket " + k)
    assert len(ref_g) > 100
    for k, g in ref_g.items():
        assert_close(seen["G"][k], g, 2e-4, "G bucket " + k)
    assert_close(cg.grad.cpu(), c_ref.grad, 1e-4, "d cond (G phase)")
    # the same conditioner for both phases is what cond_d=None means
    same = mg.HotPathTrainer(*_setup(mg, manifest, tmp_path)[:2], tr, mc)
    G2 = same.G
    _tape_trainer(G2, tapes[:2])
    o_none = same.step(mel.cuda(), conds[1].cuda(), None, pad.cuda())
    ld_same, _, _ = _oracle_phase(WG, WD, buf, mel, conds[1], pad, tapes[0], True, lam)
    assert abs(o_none["d_loss"].item() - ld_same.item()) < 1e-5 * max(1.0, abs(ld_same.item()))
    # (the D loss depends on the conditioner only through x_{t-1}'s prediction: a small but resolvable difference)
    assert abs(ld_same.item() - ld.item()) > 2e-5, "the two conditioners must give different D losses for this test to bite"


class DropoutEncoder(nn.Module):
    """Stands in for the (out-of-scope) linguistic encoder: a learned frame embedding with dropout, so that two calls
    in train mode return different conditioners, as the reference's encoder does (model/linguistic_encoder.py)."""

    def __init__(self, L, H=256):
        super().__init__()
        self.table = nn.Parameter(torch.randn(L, H, generator=torch.Generator().manual_seed(1)))
        self.drop = nn.Dropout(0.3)
        self.seen = []

    def forward(self, texts, src_lens, word_boundaries, src_masks, src_w_lens, src_w_masks, mel_masks, max_mel_len,
                attn_priors, p_targets, e_targets, d_targets, p_control, d_control):
        B = texts.shape[0]
        out = self.drop(self.table[None, :max_mel_len].expand(B, -1, -1)) * mel_masks.unsqueeze(-1)
        self.seen.append(out)
        mel_lens = mel_masks.sum(1)
        return (out, None, None, torch.zeros(B, 3, device=out.device), torch.zeros(B, 3, device=out.device), mel_lens,
                mel_masks, None, None)


def _model_and_batch(mg, manifest, tmp_path, B=2, L=48):
    e = golden("elementwise")
    stats = write_stats(tmp_path, e["spec_min"], e["spec_max"])
    args, pre, mc, tr = hot_path_configs("naive", 4, stats_dir=stats)
    enc = DropoutEncoder(L)
    model = mg.MixGANTTS(args, pre, mc, tr, linguistic_encoder=enc)
    load_seeded(model.diffusion, manifest, "diffusion_naive_ms0", 61)
    with torch.no_grad():
        model.diffusion.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
    D = mg.JCUDiscriminator(pre, mc, tr)
    load_seeded(D, manifest, "jcu_ms0", 62)
    model, D = model.cuda().train(), D.cuda()
    gen = torch.Generator().manual_seed(12)
    mel_lens = torch.tensor([L, L - 8])[:B]
    mels = (torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5) * (torch.arange(L)[None, :] < mel_lens[:, None]).unsqueeze(-1)
    cu = lambda a: a.cuda()  # noqa: E731
    batch = [["id%d" % i for i in range(B)], ["txt"] * B, cu(torch.zeros(B, dtype=torch.long)),
             cu(torch.ones(B, 5, dtype=torch.long)), cu(torch.full((B,), 5)), 5, cu(torch.ones(B, 3, dtype=torch.long)),
             cu(torch.full((B,), 3)), 3, None, None, cu(mels), cu(mel_lens), L, cu(torch.zeros(B, 5)),
             cu(torch.zeros(B, 5)), cu(torch.ones(B, 3, dtype=torch.long))]
    tapes = [[torch.tensor([3, 1])[:B]] + [torch.randn(B, 1, 80, L, generator=gen) for _ in range(3)] for _ in range(2)]
    return model, D, enc, batch, tapes, tr, mc, mels, mel_lens


@pytest.mark.parametrize("pair", [False, True])
def test_step_from_model_is_train_py_around_the_hip_step(mg, manifest, tmp_path, pair):
    """train.py:131-184: two model calls per step.  The stand-in encoder's two train-mode outputs differ; the step must
    use the first for the D phase and the second for the G phase (checked against the oracle chain on the two recorded
    conditioners), write p_targets to batch[9], train the encoder through d(cond), and -- with pair=True, both encoder
    passes up front and one launch for both generator forwards -- give the same step."""
    model, D, enc, batch, tapes, tr, mc, mels, mel_lens = _model_and_batch(mg, manifest, tmp_path)
    G = model.diffusion
    lam = tr["loss"]["lambda_fm"]
    trainer = mg.HotPathTrainer(G, D, tr, mc, extra_g_params=list(enc.parameters()), g_param_order=list(model.parameters()))
    seen = {}
    trainer.grad_hook = _recording_hook(seen, model, D)
    WG = {k: v.detach().cpu().clone().requires_grad_() for k, v in G.state_dict().items() if v.dtype == torch.float32}
    WD = {k: v.detach().cpu().clone().requires_grad_() for k, v in D.state_dict().items()}
    table0 = enc.table.detach().clone()
    _tape_trainer(G, tapes)
    torch.manual_seed(77)                    # the encoder's dropout draws
    out = trainer.step_from_model(model, batch, pair=pair)
    assert len(enc.seen) == 2 and not torch.equal(enc.seen[0], enc.seen[1])
    assert batch[9] is batch[14]             # train.py:155 `batch[9] = p_targets` (the model hands p_targets back)
    assert G.noise_fn.i == 6 and G.t_fn.i == 2
    e = golden("elementwise")
    buf = {k: T(v) for k, v in S.diffusion_buffers(S.beta_schedule("vpsde", 4, 0.1, 40, 0.008)).items()}
    buf["spec_min"], buf["spec_max"] = T(e["spec_min"])[None, None], T(e["spec_max"])[None, None]
    pad = torch.arange(mels.shape[1])[None, :] >= mel_lens[:, None]
    c_d, c_g = enc.seen[0].detach().cpu(), enc.seen[1].detach().cpu()
    ld, _, _ = _oracle_phase(WG, WD, buf, mels, c_d, pad, tapes[0], True, lam)
    ld.backward()
    ref_d = {k: v.grad.clone() for k, v in WD.items()}
    for v in WG.values():
        v.grad = None
    lg, c_ref, parts = _oracle_phase(WG, seen["D after its update"], buf, mels, c_g, pad, tapes[1], False, lam)
    lg.backward()
    assert abs(out["d_loss"].item() - ld.item()) < 1e-5 * max(1.0, abs(ld.item()))
    for k, v in parts.items():
        assert abs(out[k].item() - v.item()) < 2e-5 * max(1.0, abs(v.item())), k
    for k, g in ref_d.items():
        assert_close(seen["D"][k], g, 1e-4, "D bucket " + k)
    for k, v in WG.items():
        if v.grad is not None:
            assert_close(seen["G"]["diffusion." + k], v.grad, 2e-4, "G bucket " + k)
    # the encoder is trained by the G phase only, through d(cond) of the SECOND conditioner: d table = sum_b mask * d cond
    keep = (enc.seen[1].detach() != 0).float().cpu() / 0.7          # dropout's keep / (1 - p) factor
    d_table = (c_ref.grad * keep).sum(0)
    assert_close(seen["G"]["linguistic_encoder.table"][:d_table.shape[0]], d_table, 2e-4, "encoder table gradient")
    assert not torch.equal(enc.table.detach(), table0), "the injected encoder is stepped by optG"


def test_evaluate_step_is_the_step_without_an_update(mg, manifest, tmp_path):
    """evaluate.py:70-120: same two forwards and four discriminator passes under no_grad, losses only."""
    G, D, WG, WD, buf, mel, conds, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path)
    lam = tr["loss"]["lambda_fm"]
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    before = [p.detach().clone() for p in list(G.parameters()) + list(D.parameters())]
    _tape_trainer(G, tapes[:2])
    ev = trainer.evaluate_step(mel.cuda(), conds[1].cuda(), None, pad.cuda(), cond_d=conds[0].cuda())
    assert all(torch.equal(a, b) for a, b in zip(before, list(G.parameters()) + list(D.parameters())))
    assert all(p.grad is None for p in list(G.parameters()) + list(D.parameters()))
    with torch.no_grad():
        ld, _, _ = _oracle_phase(WG, WD, buf, mel, conds[0], pad, tapes[0], True, lam)
        lg, _, parts = _oracle_phase(WG, WD, buf, mel, conds[1], pad, tapes[1], False, lam)
    assert abs(ev["d_loss"].item() - ld.item()) < 1e-5 * max(1.0, abs(ld.item()))
    for k, v in parts.items():
        assert abs(ev[k].item() - v.item()) < 2e-5 * max(1.0, abs(v.item())), k
    vals = trainer.log_scalars(ev)
    assert all(isinstance(v, float) and np.isfinite(v) for v in vals.values())


def test_grad_acc_step_accumulates_like_model_update(mg, manifest, tmp_path):
    """train.py:75-85 with grad_acc_step = 2: loss / 2, backward every call, clip + step + zero_grad on even steps only.
    Two calls must leave the weights where ONE update with the summed (halved) gradients of both calls leaves them --
    reproduced here by a second trainer whose optimizers are driven by hand from the buckets captured on the first."""
    G, D, WG, WD, buf, mel, conds, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path, grad_acc=2)
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    assert trainer.grad_acc == 2
    p0 = [p.detach().clone() for p in list(G.parameters()) + list(D.parameters())]
    hooks, snap = [], {}

    def hook(name, bucket):
        hooks.append((name, bucket.flat.clone()))
        if name == "G":      # the discriminator as the second G phase saw it: stepped by this call's D phase
            snap["D"] = {k: v.detach().cpu().clone().requires_grad_() for k, v in D.state_dict().items()}
    trainer.grad_hook = hook
    _tape_trainer(G, tapes)
    trainer.step(mel.cuda(), conds[0].cuda(), None, pad.cuda())
    assert not hooks, "step 1 of 2: no update"
    assert all(torch.equal(a, b) for a, b in zip(p0, list(G.parameters()) + list(D.parameters())))
    assert all(p.grad is not None for p in D.parameters()) and any(p.grad is not None for p in G.parameters())
    trainer.step(mel.cuda(), conds[1].cuda(), None, pad.cuda())
    assert [n for n, _ in hooks] == ["D", "G"]
    assert any(not torch.equal(a, b) for a, b in zip(p0, list(G.parameters()) + list(D.parameters())))
    assert all(p.grad is None for p in G.parameters())
    # reference gradients: the four phases (D1, G1, D2, G2) by torch.autograd on the oracle, each scaled by 1/2;
    # the D update at step 2 sees D1 + leak(G1) + D2 (the G-phase backward of step 1 deposited into D and nothing
    # cleared it, train.py:84-85), the G update sees G1 + G2
    lam = tr["loss"]["lambda_fm"]
    accD = {k: torch.zeros_like(v) for k, v in WD.items()}
    accG = {k: torch.zeros_like(v) for k, v in WG.items()}
    for i, (cond, dphase) in enumerate([(conds[0], True), (conds[0], False), (conds[1], True), (conds[1], False)]):
        for v in list(WD.values()) + list(WG.values()):
            v.grad = None
        loss, _, _ = _oracle_phase(WG, WD if i < 3 else snap["D"], buf, mel, cond, pad, tapes[i], dphase, lam)
        (loss / 2).backward()
        if i < 3:               # G2's deposit into D comes after D's update
            for k, v in WD.items():
                if v.grad is not None:
                    accD[k] += v.grad
        if not dphase:
            for k, v in WG.items():
                if v.grad is not None:
                    accG[k] += v.grad
    gotD = dict(zip([k for k, _ in D.named_parameters()],
                    [hooks[0][1][trainer.bucketD.offsets[id(p)]:trainer.bucketD.offsets[id(p)] + p.numel()].view_as(p).cpu()
                     for _, p in D.named_parameters()]))
    for k, g in accD.items():
        assert_close(gotD[k], g, 2e-4, "accumulated D gradient " + k)
    for k, p in G.named_parameters():
        off = trainer.bucketG.offsets[id(p)]
        assert_close(hooks[1][1][off:off + p.numel()].view_as(p).cpu(), accG[k], 3e-4, "accumulated G gradient " + k)


def test_default_trainer_state_dict_is_in_parameters_order(mg, manifest, tmp_path):
    """with g_param_order=None the G bucket is laid out in Denoiser.grad_order() (per-layer kinds
    grouped), but optG.state_dict() must still index parameters like torch.optim.Adam(list(G.parameters())) does --
    a stock Adam that loads it must find every moment on the right parameter."""
    G, D, WG, WD, buf, mel, conds, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path)
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    torch.manual_seed(3)
    trainer.step(mel.cuda(), conds[0].cuda(), None, pad.cuda())
    params = [p for p in G.parameters() if p.requires_grad]
    assert [id(p) for p in trainer.optG.param_groups[0]["params"]] == [id(p) for p in params]
    assert [id(p) for p in trainer.bucketG.params] != [id(p) for p in params], "the bucket keeps its own (grad_order) layout"
    stock = torch.optim.Adam(params, lr=1e-4)
    stock.load_state_dict(trainer.optG.state_dict())
    for p in params:
        mine, theirs = trainer.optG.state[p], stock.state[p]
        assert theirs["exp_avg"].shape == p.shape
        assert torch.equal(mine["exp_avg"], theirs["exp_avg"]) and torch.equal(mine["exp_avg_sq"], theirs["exp_avg_sq"])
        assert float(theirs["step"]) == 1.0
    # moments are genuinely per parameter: two same-shaped tensors of different layers must not have been swapped
    a = G.denoise_fn.residual_layers[0].conv_layer.conv.weight
    b = G.denoise_fn.residual_layers[1].conv_layer.conv.weight
    assert not torch.equal(stock.state[a]["exp_avg"], stock.state[b]["exp_avg"])


@pytest.mark.parametrize("kind,ms", [("shallow", False), ("naive", True)])
def test_step_from_model_paired_equals_unpaired(mg, manifest, tmp_path, kind, ms):
    """step_from_model through the model variants the first test does not reach.
    shallow: the decoder / PostNet run in train mode inside both model calls (their dropout masks differ between the calls,
    so the D phase and the G phase see different coarse mels), slot 15 keeps the decoder's graph (postnet_loss trains
    decoder / PostNet / encoder, model/loss.py:165-167), the mel loss targets the detached coarse mel (:168-170),
    lambda_fm_shallow weighs the FM term.  multi-speaker naive: the speaker embedding is a trained table whose rows reach the
    denoiser and the discriminator of both phases.  With every random source pinned in call order, one launch for both
    generator forwards (pair=True) must give the step of two launches; `upstream_loss` reaches the encoder;
    evaluate_from_model returns the same loss names without touching a weight."""
    from oracle import weights as WR
    seen = {}
    for pair in (False, True):
        e = golden("elementwise")
        stats = write_stats(tmp_path, e["spec_min"], e["spec_max"], n_speakers=5 if ms else 0)
        args, pre, mc, tr = hot_path_configs(kind, 4, multi_speaker=ms, stats_dir=stats)
        B, L = 2, 48
        enc = DropoutEncoder(L)
        enc.drop.p = 0.0
        model = mg.MixGANTTS(args, pre, mc, tr, linguistic_encoder=enc)
        w = WR.draw(manifest["mixgantts_%s_ms%d" % (kind, ms)]["seeded"], 61)
        sd = model.state_dict()
        for k, a in w.items():
            if k in sd and tuple(sd[k].shape) == a.shape:
                sd[k] = torch.from_numpy(a)
        model.load_state_dict(sd)
        with torch.no_grad():
            model.diffusion.denoise_fn.output_projection.conv.weight.normal_(0, 0.05, generator=torch.Generator().manual_seed(3))
        D = mg.JCUDiscriminator(pre, mc, tr)
        load_seeded(D, manifest, "jcu_ms%d" % ms, 62)
        model, D = model.cuda().train(), D.cuda()
        gen = torch.Generator().manual_seed(12)
        mel_lens = torch.tensor([L, L - 8])
        mels = (torch.rand(B, L, 80, generator=gen) * 13.5 - 11.5) * (torch.arange(L)[None, :] < mel_lens[:, None]).unsqueeze(-1)
        cu = lambda a: a.cuda()  # noqa: E731
        batch = [["a", "b"], ["t"] * B, cu(torch.tensor([1, 3])), cu(torch.ones(B, 5, dtype=torch.long)),
                 cu(torch.full((B,), 5)), 5, cu(torch.ones(B, 3, dtype=torch.long)), cu(torch.full((B,), 3)), 3, None, None,
                 cu(mels), cu(mel_lens), L, cu(torch.zeros(B, 5)), cu(torch.zeros(B, 5)), cu(torch.ones(B, 3, dtype=torch.long))]
        tapes = [[torch.tensor([3, 1])] + [torch.randn(B, 1, 80, L, generator=gen) for _ in range(3)] for _ in range(4)]
        _tape_trainer(model.diffusion, tapes)
        mgen = torch.Generator().manual_seed(99)      # the transformer's dropout keep-masks, in call order
        mg.transformer.DROPOUT_FN = lambda shape, p, device: (torch.rand(shape, generator=mgen) >= p).to(device)
        try:
            others = [p for n, p in model.named_parameters() if not n.startswith("diffusion.")]
            trainer = mg.HotPathTrainer(model.diffusion, D, tr, mc, extra_g_params=others, g_param_order=list(model.parameters()))
            weights0 = [p.detach().clone() for p in list(model.parameters()) + list(D.parameters())]
            ev = trainer.log_scalars(trainer.evaluate_from_model(model, list(batch)))
            assert all(torch.equal(a, b) for a, b in zip(weights0, list(model.parameters()) + list(D.parameters())))
            assert all(p.grad is None for p in model.parameters())
            got = {}
            trainer.grad_hook = _recording_hook(got, model, D)
            upstream = lambda batch_, output, step: 0.25 * (enc.table ** 2).mean()  # noqa: E731
            out = trainer.step_from_model(model, batch, upstream_loss=upstream, pair=pair)
        finally:
            mg.transformer.DROPOUT_FN = None
        assert model.diffusion.noise_fn.i == 12 and model.diffusion.t_fn.i == 4
        vals = trainer.log_scalars(out)
        want = {"d_loss", "adv_loss", "mel_loss", "fm_loss"} | ({"postnet_loss"} if kind == "shallow" else set())
        assert set(vals) >= want and set(ev) >= want
        assert all(np.isfinite(v) for v in list(vals.values()) + list(ev.values()))
        assert got["G"]["linguistic_encoder.table"].abs().sum() > 0
        if kind == "shallow":
            assert vals["postnet_loss"] > 0 and got["G"]["decoder.layer_stack.0.slf_attn.w_qs.weight"].abs().sum() > 0
        if ms:
            assert got["G"]["speaker_emb.weight"][[1, 3]].abs().sum() > 0          # the two speakers of the batch ...
            assert float(got["G"]["speaker_emb.weight"][[0, 2, 4]].abs().sum()) == 0.0   # ... and nobody else
        seen[pair] = (vals, got)
    for k, v in seen[False][0].items():
        assert abs(seen[True][0][k] - v) <= 5e-5 * max(1.0, abs(v)), k
    for name in ("D", "G"):
        for k, g in seen[False][1][name].items():
            assert_close(seen[True][1][name][k], g, 2e-4, "%s bucket %s, paired vs unpaired" % (name, k))


def test_captured_step_replays_with_live_hyperparameters(mg, manifest, tmp_path):
    """HotPathTrainer.capture: the whole GAN step as one hipGraph.  Replays must train (weights move, losses finite), count
    optimizer steps like eager steps do, take the learning rate of the moment from device memory (lr = 0: a replay leaves
    every weight where it was; an lr-scheduler epoch step shows in the next update), draw fresh t / noise per replay, leave
    eager users of the modules with current weights (version bump -> repack), and -- with identical weights, data and lr = 0
    on both sides -- compute the losses an eager step computes in distribution (same data, other random draws: compared
    through the deterministic part, the discriminator loss at fixed inputs)."""
    G, D, WG, WD, buf, mel, conds, pad, tapes, tr, mc = _setup(mg, manifest, tmp_path, B=3, L=64)
    trainer = mg.HotPathTrainer(G, D, tr, mc)
    melc, condc, padc = mel.cuda(), conds[0].cuda(), pad.cuda()
    step = trainer.capture(melc, condc, None, padc)
    n0 = float(trainer.optG._steps)
    w0 = [p.detach().clone() for p in list(G.parameters()) + list(D.parameters())]
    outs = []
    for _ in range(3):
        out = step(melc, condc, None, padc)
        outs.append({k: float(v) for k, v in out.items()})
    assert all(np.isfinite(v) for o in outs for v in o.values())
    assert outs[0]["mel_loss"] != outs[1]["mel_loss"], "fresh t / noise per replay"
    assert float(trainer.optG._steps) == n0 + 3 and float(trainer.optD._steps) == n0 + 3 and trainer.step_no == int(n0) + 4
    w1 = [p.detach().clone() for p in list(G.parameters()) + list(D.parameters())]
    assert sum(int(not torch.equal(a, b)) for a, b in zip(w0, w1)) > 100
    # optimizer state reads like a stock Adam's after the same number of steps
    sd = trainer.optG.state_dict()
    assert float(sd["state"][0]["step"]) == n0 + 3
    # eager users see the replayed weights: the module's own forward equals a fresh module loaded with its state_dict
    x = torch.randn(2, 1, 80, 64, device="cuda")
    tt = torch.tensor([1, 3], device="cuda")
    cc = torch.randn(2, 256, 64, device="cuda")
    with torch.no_grad():
        y = G.denoise_fn(x, tt, cc, None)
    args, pre, mc2, tr2 = hot_path_configs("naive", 4, stats_dir=str(tmp_path))
    fresh = mg.GaussianDiffusion(args, pre, mc2, tr2)
    fresh.load_state_dict({k: v.detach().cpu() for k, v in G.state_dict().items()})
    with torch.no_grad():
        y2 = fresh.cuda().denoise_fn(x, tt, cc, None)
    assert torch.equal(y, y2), "packed-weight cache of the eager path is stale after replays"
    # lr = 0 through the ordinary param_groups: a replay must not move a weight
    for opt in (trainer.optG, trainer.optD):
        opt.param_groups[0]["lr"] = 0.0
    step(melc, condc, None, padc)
    w2 = [p.detach().clone() for p in list(G.parameters()) + list(D.parameters())]
    assert all(torch.equal(a, b) for a, b in zip(w1, w2)), "the learning rate was baked into the graph"
    # ... and a nonzero one moves them again, by an update proportional to it
    for opt, lr in ((trainer.optG, 1e-4), (trainer.optD, 2e-4)):
        opt.param_groups[0]["lr"] = lr
    step(melc, condc, None, padc)
    w3 = [p.detach() for p in list(G.parameters()) + list(D.parameters())]
    moved = max(float((a - b).abs().max()) for a, b in zip(w2, w3))
    assert 0 < moved <= 2e-3, moved            # an Adam update is of the order of lr (|m| / sqrt(v) can exceed 1 by a few x)
    trainer.check()
    with pytest.raises(ValueError):
        step(melc[:2], condc[:2], None, padc[:2])
