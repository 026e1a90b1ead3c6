"""Error behaviour of the C ABI (include/enerf_hip.h): every entry point validates its arguments before touching memory,
returns ENERF_EINVAL and leaves a message in enerf_last_error() — no launch, no crash.  Runs on the CPU lane-emulator
build of the same sources (tests/emu), i.e. the same validation code the GPU library has."""
import ctypes as C

import pytest
import torch

from emu_lib import emu_lib
from enerf_amd.lib import EnerfError, FrameArgs, GatherArgs, MlpBwdArgs

EINVAL = 1


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _msg(lib):
    return lib.dll.enerf_last_error().decode()


def test_forward_rejects_bad_frames(lib):
    assert lib.dll.enerf_forward(None, None) != 0 and "null args" in _msg(lib)
    a = FrameArgs()
    a.cas.num = 0
    assert lib.dll.enerf_forward(C.byref(a), None) != 0 and "cas_config.num" in _msg(lib)
    a.cas.num = 2
    a.B, a.S, a.H, a.W = 1, 5, 64, 64                                   # S outside 2..4
    assert lib.dll.enerf_forward(C.byref(a), None) != 0
    a.S, a.H = 3, 66                                                    # H not a multiple of 4
    assert lib.dll.enerf_forward(C.byref(a), None) != 0
    a.H = 64
    assert lib.dll.enerf_forward(C.byref(a), None) != 0 and "null batch tensor" in _msg(lib)
    assert lib.dll.enerf_forward_workspace_bytes(None) <= 0             # a size query on a bad frame reports failure too


def test_mask_compact_rejects_bad_arguments(lib):
    m = torch.ones(16, dtype=torch.uint8)
    idx, cnt = torch.zeros(16, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
    ws = torch.zeros(1024, dtype=torch.uint8)
    f = lib.dll.enerf_mask_compact
    assert f(m.data_ptr(), 3, 16, idx.data_ptr(), cnt.data_ptr(), ws.data_ptr(), 1024, None) != 0 and "elem_bytes=3" in _msg(lib)
    assert f(m.data_ptr(), 1, 0, idx.data_ptr(), cnt.data_ptr(), ws.data_ptr(), 1024, None) != 0
    assert f(None, 1, 16, idx.data_ptr(), cnt.data_ptr(), ws.data_ptr(), 1024, None) != 0
    assert f(m.data_ptr(), 1, 16, idx.data_ptr(), cnt.data_ptr(), ws.data_ptr(), 1, None) != 0      # workspace too small


def test_training_entries_reject_bad_arguments(lib):
    x = torch.zeros(1, 4, 4, 4, 8)
    with pytest.raises(EnerfError, match="unsupported"):                # only 3x3x3, 1x3x3, 1x5x5, 1x1x1 weight gradients
        gw = torch.empty(8, 8, 5, 5, 5)
        lib._check(lib.dll.enerf_conv_wgrad(x.data_ptr(), x.data_ptr(), 1, 4, 4, 4, 8, 4, 4, 4, 8, 5, 5, 5, 1, 2, 2, 2,
                                            gw.data_ptr(), None, 0, None), "conv_wgrad")
    with pytest.raises(EnerfError, match="null pointer"):
        lib._check(lib.dll.enerf_conv_wgrad(None, x.data_ptr(), 1, 4, 4, 4, 8, 4, 4, 4, 8, 3, 3, 3, 1, 1, 1, 1, x.data_ptr(), None, 0, None),
                   "conv_wgrad")
    a, b, gw = torch.zeros(10, 8), torch.zeros(10, 4), torch.zeros(8, 4)
    with pytest.raises(EnerfError, match="bad arguments"):              # row stride smaller than the used columns
        lib._check(lib.dll.enerf_gemm_wgrad(a.data_ptr(), 4, 8, b.data_ptr(), 4, 4, 10, gw.data_ptr(), None, None, 0, None), "gemm_wgrad")
    with pytest.raises(EnerfError, match="P out of range"):
        lib._check(lib.dll.enerf_gemm_wgrad(a.data_ptr(), 8, 8, b.data_ptr(), 4, 4, 0, gw.data_ptr(), None, None, 0, None), "gemm_wgrad")
    with pytest.raises(EnerfError, match="bad arguments"):              # channel count must be a multiple of 4
        s = torch.zeros(2, 6, dtype=torch.float64)
        t = torch.zeros(5, 6)
        lib._check(lib.dll.enerf_channel_sums(t.data_ptr(), t.data_ptr(), None, None, None, 5, 6, s.data_ptr(), None), "channel_sums")
    # ABI v9: the two halves of the 5x5 stride-2 input gradient, and the 64-source image gather
    w, pk, dz = torch.zeros(16, 8, 5, 5), torch.zeros(lib.dll.enerf_conv2d_s2k5_dgrad_packed_floats(8, 16)), torch.zeros(1, 4, 4, 16)
    w3 = torch.zeros(4 * 8 * 16 * 9)
    with pytest.raises(EnerfError, match="null pointer"):
        lib._check(lib.dll.enerf_conv2d_s2k5_dgrad_pack(w.data_ptr(), 8, 16, None, pk.data_ptr(), None), "s2k5_pack")
    with pytest.raises(EnerfError, match="8 -> 16 and 16 -> 32"):
        lib._check(lib.dll.enerf_conv2d_s2k5_dgrad_pack(w.data_ptr(), 8, 8, w3.data_ptr(), pk.data_ptr(), None), "s2k5_pack")
    with pytest.raises(EnerfError, match="workspace too small"):
        gx = torch.zeros(1, 8, 8, 8)
        ws = torch.zeros(4)
        lib._check(lib.dll.enerf_conv2d_s2k5_dgrad_packed(pk.data_ptr(), 8, 16, dz.data_ptr(), None, gx.data_ptr(), 1, 4, 4, ws.data_ptr(), 16, None),
                   "s2k5_packed")
    with pytest.raises(EnerfError, match="bad arguments"):              # 65 sources
        srcs = (C.c_void_p * 65)(*([w.data_ptr()] * 65))
        idx = torch.zeros(4, dtype=torch.int32)
        out = torch.zeros(4)
        lib._check(lib.dll.enerf_gather_images(C.cast(srcs, C.c_void_p), 65, idx.data_ptr(), idx.data_ptr(), 4, out.data_ptr(), None), "gather_images")
    g = GatherArgs()
    assert lib.dll.enerf_gather_fwd(None, None) != 0 and "null args" in _msg(lib)
    assert lib.dll.enerf_gather_fwd(C.byref(g), None) != 0 and "null input" in _msg(lib)
    assert lib.dll.enerf_gather_bwd(C.byref(g), None) != 0
    m = MlpBwdArgs()
    assert lib.dll.enerf_nerf_mlp_bwd(C.byref(m), None) != 0
    assert lib.dll.enerf_nerf_mlp_bwd(None, None) != 0
    v = torch.zeros(4, 8)
    assert lib.dll.enerf_nerf_mlp_fwd(v.data_ptr(), None, None, 4, 3, 11, v.data_ptr(), None) != 0
    assert lib.dll.enerf_composite(None, None, 4, 2, 0, None, None, None, None) != 0
    assert lib.dll.enerf_depth_regression_bwd(None, None, None, None, 1, 1, 1, 8, 1, None, None, None) != 0


def test_io_entries_reject_bad_arguments(lib):
    assert lib.dll.enerf_gen_rays_at(None, None, None, 1, 4, 1.0, None, None) != 0
    assert lib.dll.enerf_rays_bbox_mask(None, None, 4, None, None) != 0
    with pytest.raises(EnerfError):
        lib.gen_rays_at(torch.eye(4)[None], torch.eye(3)[None], torch.zeros(1, 4, 2), 1.0)       # xy must be int32
    with pytest.raises(EnerfError, match="contiguous float32"):
        lib.gather_fwd(torch.zeros(1, 4, 3, dtype=torch.float64), torch.zeros(1, 4), torch.zeros(1, 4, 2),
                       torch.zeros(1, 2, 4, 4, 11), torch.zeros(1, 2, 4, 4, 8), torch.zeros(1, 2, 16), torch.zeros(1, 4))


def test_feature_volume_rejects_shapes_beyond_its_grid_decomposition(lib):
    """the warp kernel carries (b, d) in gridDim.z and forms voxel indices with 24-bit multiplies; B*D > 65535 planes
    (or B*D*h >= 2^23 rows) must be refused by the C entry, not launched."""
    f = torch.zeros(1, 2, 4, 4, 8)
    proj, dv, vol = torch.zeros(1, 2, 3, 4), torch.zeros(4), torch.zeros(4)
    rc = lib.dll.enerf_build_feature_volume(f.data_ptr(), proj.data_ptr(), dv.data_ptr(), 1, 2, 8, 4, 4, 70000, 1, 1,
                                            vol.data_ptr(), None)
    assert rc != 0 and "grid-carried" in _msg(lib)
