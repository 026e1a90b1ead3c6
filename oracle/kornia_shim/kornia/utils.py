"""Stand-in for the one kornia function the reference uses (lib/networks/enerf/utils.py:4,65).

TEST INFRASTRUCTURE ONLY: lets ``oracle/ref_loader.py`` import the *unmodified* reference modules in
a container without kornia.  ``create_meshgrid(H, W, normalized_coordinates=False)`` returns a
``(1, H, W, 2)`` tensor whose last axis is ``(x, y)`` pixel coordinates (kornia's documented layout).
"""
import torch


def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    if normalized_coordinates:
        xs = (xs / (width - 1) - 0.5) * 2
        ys = (ys / (height - 1) - 0.5) * 2
    gx, gy = torch.meshgrid(xs, ys, indexing="ij")            # (W, H) each
    return torch.stack([gx, gy], -1).permute(1, 0, 2).unsqueeze(0)
