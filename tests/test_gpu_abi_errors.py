"""Error behaviour of the C ABI on a live GPU (return codes, never exit()),
and degenerate sizes through the Python mirrors."""
import ctypes

import pytest
import torch

from helpers import hot_path_configs, load_seeded, seeded, assert_close
from oracle import refmath as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import mixgan_tts_amd as m
    assert torch.cuda.is_available()
    m.lib()
    return m


def test_workspace_and_shape_errors(mg, manifest, tmp_path):
    from mixgan_tts_amd import _lib
    L = mg.lib()
    _, pre, mc, _ = hot_path_configs(stats_dir=str(tmp_path))
    den = mg.Denoiser(pre, mc).cuda()
    packed = den.packed_weights()
    B, Lf = 2, 40
    x = torch.randn(B, 80, Lf, device="cuda")
    t = torch.zeros(B, dtype=torch.long, device="cuda")
    cond = torch.randn(B, 256, Lf, device="cuda")
    out = torch.empty_like(x)
    need = L.mg_denoiser_workspace_floats(ctypes.byref(den._dims), B, Lf, 0)
    ws = torch.empty(need - 1, device="cuda")
    args = (ctypes.byref(den._dims), _lib.fptr(packed), _lib.fptr(x), _lib.iptr(t, torch.int64), _lib.fptr(cond), None,
            _lib.fptr(out), _lib.fptr(ws))
    assert L.mg_denoiser_fwd(*args, ws.numel(), B, Lf, 0, None) == -3          # MG_ERR_WORKSPACE
    assert L.mg_denoiser_fwd(*args, need, 0, Lf, 0, None) == -2                 # MG_ERR_SHAPE (B = 0)
    assert L.mg_denoiser_fwd(*args, need, B, Lf, 3, None) == -1                 # save + split is not a valid mode
    assert L.mg_attention_fwd(_lib.fptr(x), None, _lib.fptr(out), 1, 8, 2, 64, 0.1, None) == -2   # d_head != 128
    with pytest.raises(mg.MixganHipError):
        _lib.check(-3)
    # multi-speaker denoiser without a speaker embedding is an argument error, not a crash
    _, pre2, mc2, _ = hot_path_configs(multi_speaker=True, stats_dir=str(tmp_path))
    den2 = mg.Denoiser(pre2, mc2).cuda()
    with pytest.raises(mg.MixganHipError):
        den2.run(x, t, cond, None)
    torch.cuda.synchronize()


def test_degenerate_sizes(mg, manifest, tmp_path):
    """L = 1 and B = 1 through every stage; a fully padded tail; non-contiguous inputs to the mirrors."""
    _, pre, mc, _ = hot_path_configs(stats_dir=str(tmp_path))
    den = mg.Denoiser(pre, mc)
    load_seeded(den, manifest, "denoiser_ms0", 5)
    W, _ = seeded(manifest, "denoiser_ms0", 5)
    den = den.cuda()
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(1, 1, 80, 1, generator=gen)
    cond = torch.randn(1, 256, 1, generator=gen)
    t = torch.tensor([2])
    with torch.no_grad():
        assert_close(den(x.cuda(), t.cuda(), cond.cuda(), None).cpu(), R.denoiser_forward(W, "", x, t, cond, None), 2e-5, "L=1")
    # non-contiguous conditioner view ([B,L,H].transpose) is accepted (made contiguous by the mirror)
    c2 = torch.randn(2, 33, 256, generator=gen)
    x2 = torch.randn(2, 1, 80, 33, generator=gen)
    t2 = torch.tensor([0, 3])
    with torch.no_grad():
        got = den(x2.cuda(), t2.cuda(), c2.cuda().transpose(1, 2), None)
    assert_close(got.cpu(), R.denoiser_forward(W, "", x2, t2, c2.transpose(1, 2), None), 2e-5, "non-contiguous cond")
