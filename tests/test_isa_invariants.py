"""Invariants of the gfx950 code objects that neither the CPU lane emulator nor a parity test can see.

`vmem_wait_pending<41>()` in k_smooth1_fused (conv2d.hip) stands for "pass 1's LDS-DMA patch copy has landed" only while hipcc
emits exactly 41 vector-memory loads between that copy and the wait; the emulator compiles both to no-ops.  The check compiles the
source for gfx950 (no GPU needed) and counts them in the ISA."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_counted_vmcnt_waits_match_the_loads_behind_the_lds_dma_copy():
    from isa_vmcnt_check import check
    waits = check("conv2d.hip", "k_smooth1_fused")
    assert waits, "k_smooth1_fused: no counted s_waitcnt vmcnt(N) found — the kernel changed; update this test with it"
    for n, loads in waits:
        assert loads is not None, f"vmcnt({n}) without an LDS-DMA copy in front of it"
        assert n == loads, f"k_smooth1_fused waits vmcnt({n}) but {loads} vector loads follow the LDS-DMA copy: the wait no longer means 'copy landed'"
