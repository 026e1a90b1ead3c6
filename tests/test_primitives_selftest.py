"""enerf_selftest_primitives (csrc/selftest.hip): every gfx950-only helper of csrc/common.h — and its ENERF_EMU twin, which is what
the CPU suite's kernel tests execute — against one memory-only specification.  CPU: the emulator build (the twins).  `-m gpu`: the
gfx950 build (the intrinsics).  Zero disagreeing lanes in every check on both is what ties the emulated kernel tests to the hardware
code path."""
import pytest
import torch


def test_emulator_twins_match_the_specification():
    from emu_lib import emu_lib
    got = emu_lib().selftest_primitives("cpu", blocks=9)
    assert len(got) == 20 and all(v == 0 for v in got.values()), got


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_gfx950_intrinsics_match_the_specification():
    from enerf_amd.lib import get_lib
    for blocks in (1, 24, 509):                           # 509: a grid that is no multiple of the 8 XCDs
        got = get_lib().selftest_primitives("cuda:0", blocks=blocks)
        torch.cuda.synchronize()
        assert len(got) == 20 and all(v == 0 for v in got.values()), (blocks, got)
