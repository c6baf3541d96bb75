"""Host logic of the adapter's in-place row view (no GPU): which `raw_gaussians` tensors are read where they lie."""
import torch

from spfsplatv2_amd.rasterizer import raw_rows


def test_rows_of_a_head_output_view_are_read_in_place():
    head = torch.randn(2, 3, 100, 83)                              # [b, v, r, 1 + 82]: density first (encoder_spfsplatv2.py:261-268)
    view = head[..., 1:]
    for t in (view, view.reshape(2, 3, 100, 1, 1, 82)):            # "b v r srf c -> b v r srf () c" keeps it a view
        rows = raw_rows(t, 82)
        assert rows.data_ptr() == view.data_ptr() and tuple(rows.stride()) == (83, 1) and tuple(rows.shape) == (600, 82)
        assert torch.equal(rows, view.reshape(-1, 82))
    c = torch.randn(5, 7, 82)
    assert raw_rows(c, 82).data_ptr() == c.data_ptr() and tuple(raw_rows(c, 82).stride()) == (82, 1)
    assert raw_rows(torch.randn(82), 82).shape == (1, 82)


def test_rows_that_are_not_one_stride_are_copied_once():
    t = torch.randn(5, 82, 7).transpose(1, 2)                      # channels not contiguous
    r = raw_rows(t, 82)
    assert r.data_ptr() != t.data_ptr() and r.is_contiguous() and torch.equal(r, t.reshape(-1, 82))
    skip = torch.randn(2, 4, 100, 83)[:, ::2, :, 1:]               # two different strides over the leading dimensions
    r = raw_rows(skip, 82)
    assert r.is_contiguous() and torch.equal(r, skip.reshape(-1, 82))
    h = torch.randn(3, 10, 82, dtype=torch.float64)
    assert raw_rows(h, 82).dtype == torch.float32
