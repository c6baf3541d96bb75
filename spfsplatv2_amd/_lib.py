"""ctypes binding of libspfsplat_hip.so (C ABI: include/spfsplat_hip.h).

The HIP library is the only compute path of this package: if it is missing or does not export the
ABI declared in the header, importing/using the package fails loudly -- there is no CPU or
PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# SPF_LIB_DIR: development only -- a profiling/experimental build kept next to the regular one (see build.py)
LIB_PATH = Path(__file__).resolve().parent / os.environ.get("SPF_LIB_DIR", "_C") / "libspfsplat_hip.so"
ABI_VERSION = 6

STAGE_NAMES = ("project_fwd", "tile_scan", "bin_pairs", "tile_sort", "render_fwd", "render_bwd",
               "project_bwd", "rope2d")
STAGE_COUNT = len(STAGE_NAMES)


class SpfDims(C.Structure):
    _fields_ = [("S", C.c_int32), ("V", C.c_int32), ("G", C.c_int32), ("K", C.c_int32),
                ("sh_degree", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("scale_modifier", C.c_float), ("sh_layout", C.c_int32), ("sh_band4", C.c_int32),
                ("bin_cap", C.c_int32), ("pair_capacity", C.c_int64), ("raw_stride", C.c_int64), ("adapter_eps", C.c_float)]


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(f, C.c_void_p) for f in fields]})


SpfInputs = _ptr_struct("SpfInputs", ["means3D", "scales", "rotations", "opacities", "shs", "colors",
                                      "viewmatrix", "projmatrix", "tanfov", "bg", "view_scale", "viewmatrix64",
                                      "shs_high", "raw", "sh_mask"])
SpfState = _ptr_struct("SpfState", ["rec", "radii", "rect", "zkey", "tile_count", "tile_start", "tile_fill",
                                    "tile_flags", "counters", "pairs", "pair_off", "blk_total", "blk_base", "final_T",
                                    "n_contrib", "pair_cursor", "sh_clamp", "verdict_host"])
SpfOutputs = _ptr_struct("SpfOutputs", ["image", "depth", "alpha"])
SpfGrads = _ptr_struct("SpfGrads", ["dL_dimage", "dL_ddepth", "dL_dalpha", "gpair", "vpartial",
                                    "dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacities",
                                    "dL_dshs", "dL_dcolors", "dL_dviewmatrix", "dL_dmeans2D", "dL_dshs_high",
                                    "dL_draw"])



class SpfCamera(C.Structure):
    _fields_ = [(f, C.c_void_p) for f in ("extrinsics", "intrinsics", "near", "far", "viewmatrix", "projmatrix",
                                          "tanfov", "view_scale")] + [("R", C.c_int32), ("scale_invariant", C.c_int32),
                                                                    ("viewmatrix64", C.c_void_p)]


# Every symbol include/spfsplat_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "spf_abi_version": (C.c_int, []),
    "spf_last_error": (C.c_char_p, []),
    "spf_raster_num_tiles": (C.c_int, [C.c_int32, C.c_int32]),
    "spf_raster_view_partial_blocks": (C.c_int, [C.c_int32]),
    "spf_raster_launch_slot_tile": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "spf_raster_chunks": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "spf_raster_pair_shards": (C.c_int, [C.c_int32, C.c_int32]),
    "spf_raster_max_lds_tiles": (C.c_int, []),
    "spf_camera_forward": (C.c_int, [C.POINTER(SpfCamera), C.c_void_p]),
    "spf_camera_backward": (C.c_int, [C.POINTER(SpfCamera), C.c_void_p, C.c_void_p, C.c_void_p]),
    "spf_raster_forward_project": (C.c_int, [C.POINTER(SpfDims), C.POINTER(SpfInputs), C.POINTER(SpfState),
                                             C.c_void_p]),
    "spf_raster_forward_render": (C.c_int, [C.POINTER(SpfDims), C.POINTER(SpfInputs), C.POINTER(SpfState),
                                            C.POINTER(SpfOutputs), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]),
    "spf_raster_backward": (C.c_int, [C.POINTER(SpfDims), C.POINTER(SpfInputs), C.POINTER(SpfState),
                                      C.POINTER(SpfGrads), C.c_uint64, C.c_uint32, C.c_void_p]),
    "spf_decoder_prepare": (C.c_int, [C.POINTER(SpfCamera), C.c_void_p, C.c_uint64, C.c_void_p]),
    "spf_raster_forward_project_prepared": (C.c_int, [C.POINTER(SpfDims), C.POINTER(SpfInputs), C.POINTER(SpfState),
                                                      C.c_uint64, C.c_void_p]),
    "spf_camera_backward_partials": (C.c_int, [C.POINTER(SpfCamera), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "spf_mse_partial_blocks": (C.c_int, []),
    "spf_mse_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spf_mse_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spf_mse_forward_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "spf_mse_scale_grad": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "spf_adapter_forward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "spf_adapter_backward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "spf_rope2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                             C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float,
                             C.c_void_p]),
    "spf_rope2d_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                  C.c_void_p]),
    "spf_stage_timing_enable": (C.c_int, [C.c_int32]),
    "spf_stage_timing_sample_every": (C.c_int, [C.c_int32]),
    "spf_stage_times_ms": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "spf_stage_kernel_name": (C.c_char_p, [C.c_int32]),
}

_lib = None


class SpfError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m spfsplatv2_amd.build` "
            "(hipcc, gfx950).  spfsplatv2_amd has no fallback path.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.spf_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {got}, expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


_fast = False          # False: not tried yet; None: not available


def fast():
    """The compiled host binding (csrc/torch_binding.cpp -> _spf_torch.so next to the HIP library), or None when it has
    not been built or SPF_NO_FAST=1.  It drives the same C ABI of the same library -- only the interpreter time of a
    call differs (rasterizer.py keeps the ctypes path for that case)."""
    global _fast
    if _fast is False:
        _fast = None
        path = LIB_PATH.parent / "_spf_torch.so"
        if os.environ.get("SPF_NO_FAST", "0") != "1" and path.exists():
            load()                                   # the HIP library first (and its ABI check)
            import importlib.util
            try:
                spec = importlib.util.spec_from_file_location("_spf_torch", str(path))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if mod.abi_version() == ABI_VERSION:
                    _fast = mod
                else:                                # a stale build left next to a newer library: say so, once
                    import warnings
                    warnings.warn(f"spfsplatv2_amd: {path} was built for ABI {mod.abi_version()}, the library is ABI "
                                  f"{ABI_VERSION}; using the (slower) ctypes binding -- rebuild with "
                                  "`python -m spfsplatv2_amd.build`")
            except (ImportError, OSError) as e:      # e.g. built against another torch: fall back, but say so once
                import warnings
                warnings.warn(f"spfsplatv2_amd: {path} could not be loaded ({e}); using the ctypes binding")
    return _fast


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().spf_last_error().decode(errors="replace")
        raise SpfError(f"{what} failed (code {rc}): {msg}")


def stage_timing_enable(stages=True) -> None:
    """True = all stages, False = off, or an iterable of stage names (see STAGE_NAMES)."""
    if stages is True:
        mask = -1
    elif not stages:
        mask = 0
    else:
        mask = 0
        for s in stages:
            mask |= 1 << STAGE_NAMES.index(s)
    check(load().spf_stage_timing_enable(mask), "spf_stage_timing_enable")


def stage_timing_sample_every(n: int) -> None:
    """Record only every n-th launch of an enabled stage (an event pair costs ~11 us of idle GPU per launch)."""
    check(load().spf_stage_timing_sample_every(int(n)), "spf_stage_timing_sample_every")


def stage_times() -> dict[str, tuple[float, int]]:
    """{stage: (total device ms, launches)} since the last stage_timing_enable(True)."""
    ms = (C.c_float * STAGE_COUNT)()
    cnt = (C.c_int32 * STAGE_COUNT)()
    check(load().spf_stage_times_ms(ms, cnt), "spf_stage_times_ms")
    return {STAGE_NAMES[i]: (float(ms[i]), int(cnt[i])) for i in range(STAGE_COUNT)}


def stage_kernel_name(stage: str) -> str:
    return load().spf_stage_kernel_name(STAGE_NAMES.index(stage)).decode()
