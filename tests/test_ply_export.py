"""export_ply against the vertex table the reference's own export_ply builds (tests/golden/make_ply_goldens.py)."""
import numpy as np
import torch

from spfsplatv2_amd import ply_export


def test_matches_reference_table_and_roundtrips(golden_dir, tmp_path):
    g = torch.load(golden_dir / "ply_golden.pt")
    assert g["names"] == ply_export.ATTRIBUTES
    path = tmp_path / "sub" / "scene.ply"
    ply_export.export_ply(g["extrinsics"], g["means"], g["scales"], g["rotations"], g["harmonics"], g["opacities"], path)
    got = ply_export.read_ply(path)
    want = g["table"].numpy()
    assert got.shape == want.shape == (300, 17)
    np.testing.assert_allclose(got[:, :13], want[:, :13], rtol=1e-5, atol=1e-5)     # xyz, normals, dc, opacity, scales
    # quaternions (w,x,y,z): q and -q are the same rotation
    dots = np.abs((got[:, 13:] * want[:, 13:]).sum(axis=1))
    assert float(np.abs(dots - 1).max()) < 1e-5
    assert np.all(got[:, 3:6] == 0)
    head = path.read_bytes()[:64]
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 300\n")
