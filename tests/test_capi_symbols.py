"""The C-ABI library loads without a GPU and exports every symbol include/enerf_hip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "enerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(enerf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_bound():
    from enerf_amd.lib import EXPORTED_SYMBOLS
    assert sorted(EXPORTED_SYMBOLS) == _declared()


def test_hip_library_builds_and_exports_all_symbols():
    import __graft_entry__ as g
    g.build()                                             # hipcc cross-compiles gfx950 without a GPU
    from enerf_amd.lib import LIB_PATH, EnerfLib
    out = subprocess.run(["nm", "-D", "--defined-only", LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (enerf_[a-z0-9_]+)", out))
    assert set(_declared()) <= exported, set(_declared()) - exported
    lib = EnerfLib(LIB_PATH)                              # dlopen + ABI version; no compute calls here
    from enerf_amd.lib import ABI_VERSION
    assert lib.dll.enerf_abi_version() == ABI_VERSION == 11
    assert lib.dll.enerf_nerf_packed_floats(11) > 0 and lib.dll.enerf_cost_reg_packed_floats(16, 1) > 0


def test_gfx950_code_object_contains_mfma(tmp_path):
    """The product kernels really are CDNA4 matrix-core code (not a generic fallback): unbundle the gfx950 code
    objects into a temp dir, disassemble, and count the two fp32 MFMA instructions the kernels are written around."""
    import shutil
    from enerf_amd.lib import LIB_PATH
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump) or not os.path.exists(LIB_PATH):
        pytest.skip("llvm-objdump or library not available")
    lib = shutil.copy(LIB_PATH, tmp_path / "lib.so")          # --offloading extracts next to its input
    out = subprocess.run([objdump, "--offloading", str(lib)], capture_output=True, text=True, cwd=tmp_path).stdout
    assert "gfx950" in out
    cos = sorted(p for p in os.listdir(tmp_path) if "gfx950" in p)
    assert cos, "no gfx950 code object in the library"
    n_mfma, n_b4, other, bf16_in = 0, 0, set(), set()
    for co in cos:
        dis = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(tmp_path / co)], capture_output=True, text=True).stdout
        sym = ""
        for line in dis.splitlines():
            ms = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if ms:
                sym = ms.group(1)
                continue
            for m in re.findall(r"\bv_mfma_[a-z0-9_]+", line):
                if m == "v_mfma_f32_16x16x4_f32":
                    n_mfma += 1
                elif m == "v_mfma_f32_4x4x1_16b_f32":        # batched 4x4 fp32 shape of conv3d_b4.hip (Cout = 8 layers)
                    n_b4 += 1
                elif m == "v_mfma_f32_16x16x32_bf16":        # the OPT-IN bf16x3 / bf16x6 render variants (enerf_options_t.render_precision)
                    bf16_in.add(sym)
                else:
                    other.add(m)
    assert n_mfma > 5000, n_mfma                             # render + conv2d + conv3d kernels (17 k in round 1)
    assert n_b4 > 5000, n_b4                                 # 27 taps x Cin x 2 halves x voxels-per-lane per b4 kernel
    assert not other, other                                  # no other MFMA shape anywhere
    # the default path is exact fp32: bf16 MFMAs live ONLY in the render kernels instantiated with BX = 3 / 6
    assert bf16_in and all(re.search(r"k_render_raysILi3ELi[234]ELi12ELi3ELb0ELb1ELi[36]E", f) for f in bf16_in), bf16_in
    assert not [p for p in os.listdir(os.path.dirname(LIB_PATH)) if "hipv4" in p], "code objects leaked into the package"


def test_missing_library_fails_loudly(tmp_path):
    from enerf_amd.lib import EnerfError, EnerfLib
    with pytest.raises(EnerfError, match="no fallback"):
        EnerfLib(str(tmp_path / "nope.so"))
