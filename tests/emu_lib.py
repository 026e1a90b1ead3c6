"""Test helper: the CPU-emulated twin of libenerf_hip.so (same kernel sources, g++ + tests/emu/hip_emu.h)."""
import functools

from enerf_amd.lib import EnerfLib
from emu.build_emu import build


@functools.lru_cache(maxsize=1)
def emu_lib() -> EnerfLib:
    return EnerfLib(build())
