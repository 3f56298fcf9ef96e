"""mg_conv1d_fwd_split: the reduction of a convolution dealt to several workgroups per output tile (csrc/conv_mfma.h)
against the unsplit kernel and against torch's fp32 conv1d on the CPU -- the JCU discriminator's tail shapes
(model/mixgantts.py:219-248: 512 -> 128 channels, k = 5, at L/4 frames) and their data gradients, ragged sizes, the
fused input vector / bias / leaky ReLU, and repeated launches on one scratch."""
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,Ci,Co,K,L,stride,pad", [
    (16, 512, 128, 5, 250, 1, 2),      # the tail's first convolution at the training shard size (2B = 16, L/4)
    (16, 128, 512, 5, 250, 1, 2),      # its data gradient: rows = 512, reduction 128 x 5
    (3, 512, 128, 5, 37, 1, 2),        # ragged
    (2, 128, 64, 5, 1000, 1, 2),       # <= 64 rows: the one-block-pair tiling
    (5, 512, 128, 5, 131, 1, 2),
    (4, 256, 256, 3, 100, 1, 1),
    (1, 512, 128, 5, 9, 1, 2),
])
def test_split_reduction_equals_unsplit_and_torch(B, Ci, Co, K, L, stride, pad):
    import mixgan_tts_amd as mg
    ops = mg.ops
    gen = torch.Generator().manual_seed(B * 1000 + L)
    x = torch.randn(B, Ci, L, generator=gen)
    w = torch.randn(Co, Ci, K, generator=gen) * (Ci * K) ** -0.5
    bias = torch.randn(Co, generator=gen)
    vec = torch.randn(B, Ci, generator=gen)
    xp = F.pad(x + vec[:, :, None], (pad, pad))                      # the vector is added to in-range samples only
    ref = F.leaky_relu(F.conv1d(xp, w, bias, stride=stride), 0.2)
    xd, wd, bd, vd = x.cuda(), w.cuda(), bias.cuda(), vec.cuda()
    packed = ops.pack_conv_weight(wd)
    plain = ops.conv1d_packed(xd, packed, bd, Co, K, stride, pad, "lrelu", in_vec=vd)
    outs = [ops.conv1d_packed(xd, packed, bd, Co, K, stride, pad, "lrelu", in_vec=vd, split=True) for _ in range(3)]
    torch.cuda.synchronize()
    assert_close(plain.cpu(), ref, 2e-5, "unsplit vs torch")
    assert_close(outs[0].cpu(), ref, 2e-5, "split vs torch")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), "fixed summation order"
    # accumulate / add epilogue through the split path
    base = torch.randn(B, Co, ref.shape[2], generator=gen).cuda()
    acc = base.clone()
    ops.conv1d_packed(xd, packed, None, Co, K, stride, pad, None, alpha=0.5, out=acc, accumulate=True, split=True)
    ref2 = base.cpu() + 0.5 * F.conv1d(F.pad(x, (pad, pad)), w, None, stride=stride)
    assert_close(acc.cpu(), ref2, 2e-5, "split + accumulate")


def test_split_scratch_is_per_stream():
    import mixgan_tts_amd as mg
    ops = mg.ops
    dev = torch.device("cuda", 0)
    a = ops.split_scratch(dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        b = ops.split_scratch(dev)
    assert a.data_ptr() != b.data_ptr() and ops.split_scratch(dev) is a
    assert a.numel() >= 512 * 16384
