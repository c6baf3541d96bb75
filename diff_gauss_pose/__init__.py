"""Drop-in module name for the reference's rasterizer dependency.

SPFSplatV2 does ``from diff_gauss_pose import GaussianRasterizationSettings, GaussianRasterizer``
(/root/reference/src/model/decoder/cuda_splatting.py:5; package pinned at requirements.txt:88).  With this
repository on ``sys.path`` that import resolves to the MI355X HIP rasterizer without touching the reference.
"""
from spfsplatv2_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
