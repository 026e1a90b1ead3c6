"""The small deep stride-1 / stride-2 cost-regularisation layers (conv3 .. conv6, cost_reg_net.py:13-24,58-62) through
enerf_conv3d_layer against torch's Conv3d (what the reference's ConvBnReLU3D wraps, utils.py:22-33): the block-shared-weight kernel
(conv3d_wl.hip, the default for < 1024 wave tiles) and the round-2 .. 5 forms it replaces, on shapes that exercise its edge logic —
volumes 1 and 2 planes thick (whole kd taps skipped), ragged last tiles, rows that are no multiple of 16, a batch boundary inside
a block.  CPU: the kernel source under the lane emulator.  GPU (`-m gpu`): the same cases on the gfx950 build."""
import pytest
import torch
import torch.nn.functional as F

from enerf_amd.lib import Options

S1, S2 = 0, 1
SHAPES = [  # kind, cin, cout, (B, D, h, w)
    (S1, 32, 32, (1, 2, 8, 10)),      # level-1 conv4: two planes, every voxel has one padding kd
    (S1, 64, 64, (1, 1, 8, 5)),       # level-1 conv6: one plane -> 9 of 27 taps; 40 voxels: ragged last tile
    (S2, 32, 64, (1, 2, 8, 10)),      # level-1 conv5: 2 -> 1 planes, kd = 0 is padding for the whole layer
    (S2, 16, 32, (1, 4, 16, 20)),     # level-1 conv3
    (S1, 32, 32, (1, 12, 4, 5)),      # level-0 conv4: rows of 5 voxels (tiles wrap rows and planes)
    (S2, 16, 32, (1, 7, 9, 11)),      # odd extents: (D - 1) / 2 + 1 outputs, last input plane unused
    (S1, 16, 16, (2, 3, 4, 6)),       # batch boundary inside a block (72 voxels per sample)
    (S1, 32, 48, (1, 3, 5, 7)),       # three row tiles
]


def _run(lib, dev, kind, cin, cout, shape, opt):
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + kind)
    B, D, h, w = shape
    wt = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.1)
    x = torch.randn((B, D, h, w, cin), generator=g)
    ref = F.conv3d(x.permute(0, 4, 1, 2, 3).double(), wt.double(), stride=2 if kind == S2 else 1, padding=1).permute(0, 2, 3, 4, 1)
    packed = lib.conv3d_layer_pack(wt.to(dev), cin, cout, kind)
    out = lib.conv3d_layer(packed, cin, cout, kind, x.to(dev).contiguous(), None, opt)
    if dev != "cpu":
        torch.cuda.synchronize()
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert out.shape == ref.shape and err < 2e-5, (kind, cin, cout, shape, err)


@pytest.mark.parametrize("variant", [0, 1, 4])
@pytest.mark.parametrize("case", range(len(SHAPES)))
def test_small_layers_emulated(case, variant):
    from emu_lib import emu_lib
    kind, cin, cout, shape = SHAPES[case]
    _run(emu_lib(), "cpu", kind, cin, cout, shape, Options(conv3d_small_variant=variant))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 4])
def test_small_layers_gpu(variant):
    from enerf_amd.lib import get_lib
    for kind, cin, cout, shape in SHAPES + [(S1, 64, 64, (1, 1, 32, 40)), (S1, 32, 32, (1, 2, 64, 80)), (S2, 16, 32, (1, 24, 32, 40)),
                                            (S1, 64, 64, (1, 2, 16, 20))]:
        _run(get_lib(), "cuda:0", kind, cin, cout, shape, Options(conv3d_small_variant=variant))
