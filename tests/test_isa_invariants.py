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


def test_only_the_known_kernels_spill_vector_registers_or_use_scratch(tmp_path):
    """Scratch traffic in a hot kernel is invisible to every parity test and to the emulator.  The built library's kernel
    metadata must show vector-register spills / a private segment ONLY for the kernels known to have them: the level-0 render
    kernel (F = 35: 35-channel records, DESIGN.md §8 lead 5), the opt-in bf16 render variants and the 4-wave render variant of
    standalone enerf_render_rays calls.  Every default-path kernel of the dtu frame and of the training step is spill-free."""
    import re
    import subprocess
    from enerf_amd.lib import LIB_PATH
    tools = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(f"{tools}/llvm-objdump") or not os.path.exists(LIB_PATH):
        pytest.skip("llvm tools or library not available")
    lib = shutil.copy(LIB_PATH, tmp_path / "lib.so")
    subprocess.run([f"{tools}/llvm-objdump", "--offloading", str(lib)], capture_output=True, text=True, cwd=tmp_path)
    allowed = re.compile(r"k_render_raysILi9E|k_render_raysILi3ELi[234]ELi12ELi3ELb0ELb1ELi[36]E|k_render_raysILi3ELi4ELi4ELi2ELb1ELb0ELi0E")
    seen, offenders = 0, []
    for co in sorted(p for p in os.listdir(tmp_path) if "gfx950" in p):
        notes = subprocess.run([f"{tools}/llvm-readelf", "--notes", str(tmp_path / co)], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            vspill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
            seen += 1
            if (scratch or vspill) and not allowed.search(name):
                offenders.append((name, scratch, vspill))
    assert seen > 250, seen                                # 289 kernels in round 6
    assert not offenders, offenders
