"""Import the UNMODIFIED reference network modules from /root/reference (this container only).

TEST INFRASTRUCTURE ONLY — never imported by the product path.  /root/reference does not exist on
the GPU box; everything that needs the reference at run time lives in ``oracle/make_golden.py`` and
writes fixtures into ``tests/golden/``.

Shims: a ``kornia.utils.create_meshgrid`` stand-in on sys.path, env ``workspace``,
and an argv that makes ``lib/config/config.py:191-201`` parse the requested yaml at import time.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import tempfile

REFERENCE_ROOT = "/root/reference"
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kornia_shim")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lib", "networks", "enerf"))


def load_reference(cfg_file: str = "configs/enerf/dtu_pretrain.yaml", opts: list[str] | None = None):
    """Return ``(cfg, network_module)`` of the reference.  One configuration per process (the
    reference's cfg is a module-global built at import time)."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    if "lib.config" in sys.modules:
        raise RuntimeError("reference already imported in this process; cfg is import-time global")
    os.environ.setdefault("workspace", tempfile.mkdtemp(prefix="enerf_ws_"))
    sys.dont_write_bytecode = True
    for p in (_SHIM, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = ["oracle", "--cfg_file", cfg_file] + list(opts or [])
    os.chdir(REFERENCE_ROOT)                     # yaml parent_cfg paths are cwd-relative
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            from lib.config import cfg            # noqa: E402
            from lib.networks.enerf import network as ref_network  # noqa: E402
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
    return cfg, ref_network
