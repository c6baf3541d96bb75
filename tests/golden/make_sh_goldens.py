"""Golden vectors for the real spherical-harmonics basis (bands 0-4) from the REFERENCE's own table
``/root/reference/src/misc/sht.py::rsh_cart_4`` (build container only; Ynm at index n(n+1)+m -- the ordering and
signs the 3DGS family uses for bands 0-3, and the table that pins band 4 here).
Writes tests/golden/sh_basis_goldens.pt.   python tests/golden/make_sh_goldens.py
"""
import importlib.util
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent


def main():
    spec = importlib.util.spec_from_file_location("ref_sht", "/root/reference/src/misc/sht.py")
    sht = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sht)
    gen = torch.Generator().manual_seed(44)
    d = torch.randn(257, 3, generator=gen, dtype=torch.float64)
    d = torch.cat([d, torch.eye(3, dtype=torch.float64), -torch.eye(3, dtype=torch.float64)])
    d = d / d.norm(dim=-1, keepdim=True)
    torch.save({"directions": d, "rsh_cart_4": sht.rsh_cart_4(d)}, HERE / "sh_basis_goldens.pt")
    print("wrote", HERE / "sh_basis_goldens.pt", tuple(d.shape))


if __name__ == "__main__":
    main()
