"""ctypes binding of libmi_ipp.so (the C ABI declared in include/*.h).

This is the only door between the Python host code and the HIP kernels.  There is no CPU fallback:
if the shared library is missing or a symbol is absent, importing ``lib()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_IPP_PROBES=1 (profiles/ only): the build with the experiment switches compiled in (make -C csrc probes)
LIB_PATH = os.path.join(_HERE, "libmi_ipp_probes.so" if os.environ.get("MI_IPP_PROBES") == "1" else "libmi_ipp.so")

MI_OK = 0
IPC_HANDLE_BYTES = 64  # MI_IPC_HANDLE_BYTES
# mi_boundary / mi_engine (include/mi_lsdeconv.h)
BOUNDARY_ZERO, BOUNDARY_REPLICATE, BOUNDARY_CIRCULAR = 0, 1, 2
ENGINE_AUTO, ENGINE_DIRECT, ENGINE_FFT = 0, 1, 2
NORTH_SOUTH, WEST_EAST = 0, 1


MI_ERR_INVALID, MI_ERR_HIP, MI_ERR_FFT, MI_ERR_NOMEM, MI_ERR_UNSUPPORTED = -1, -2, -3, -4, -5  # include/mi_common.h


class MiError(RuntimeError):
    """A C-ABI call returned a negative mi_status; the message is mi_last_error()."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[mi_status {code}] {message}")
        self.code = code


class RlOptions(C.Structure):
    """mi_rl_options (include/mi_lsdeconv.h)."""
    _fields_ = [("niter", C.c_int), ("lambda_", C.c_float), ("stop_criterion", C.c_float),
                ("regularize_interval", C.c_int), ("engine", C.c_int), ("skip_edgetaper", C.c_int),
                ("gauss_taps", C.c_int), ("psf_grid", C.c_int * 3)]


class NccParams(C.Structure):
    """mi_ncc_params == NCC_parms_t without the enhance tables (CrossMIPs.h:65-86)."""
    _fields_ = [("enhance", C.c_int), ("maxIter", C.c_int), ("maxThr", C.c_float), ("widthThr", C.c_float),
                ("wRangeThr_i", C.c_int), ("wRangeThr_j", C.c_int), ("wRangeThr_k", C.c_int),
                ("minPoints", C.c_int), ("minDim_NCCsrc", C.c_int), ("minDim_NCCmap", C.c_int),
                ("UNR_NCC", C.c_float), ("INF_W", C.c_int), ("INV_COORD", C.c_int)]


class NccDescr(C.Structure):
    """mi_ncc_descr == NCC_descr_t (CrossMIPs.h:58-62)."""
    _fields_ = [("coord", C.c_int * 3), ("NCC_maxs", C.c_float * 3), ("NCC_widths", C.c_int * 3)]


_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol declared in include/*.h
SIGNATURES = {
    # mi_common.h
    "mi_last_error": (C.c_char_p, []),
    "mi_device_count": (_i, []),
    "mi_abi_version": (_i, []),
    "mi_stream_synchronize": (_i, [_i, _vp]),
    # mi_lsdeconv.h
    "mi_conv3d_replicate": (_i, [_i, _vp, _vp, _vp, _vp] + [_i] * 6),
    "mi_conv3d": (_i, [_i, _vp, _vp, _vp, _vp] + [_i] * 8),
    "mi_gauss3d_inplace": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, C.POINTER(C.c_float), _ip]),
    "mi_edgetaper3d": (_i, [_i, _vp, _vp, _vp, _vp] + [_i] * 6),
    "mi_otf": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _f]),
    "mi_u16_to_f32": (_i, [_i, _vp, _vp, _vp, _sz, _f]),
    "mi_subtract_dark": (_i, [_i, _vp, _vp, _vp, _sz, _f]),
    "mi_norm2": (_i, [_i, _vp, _vp, _sz, C.POINTER(C.c_double)]),
    "mi_release_cached_memory": (_sz, [_i]),
    "mi_cached_memory_bytes": (_sz, []),
    "mi_rl_fuses": (_i, [_vp]),
    "mi_rl_otf_is_real": (_i, [_vp]),
    "mi_rl_sharded_begin": (_i, [_vp, _vp, _vp]),
    "mi_rl_sharded_ratio": (_i, [_vp, _vp, _vp, _i, C.POINTER(C.c_int)]),
    "mi_rl_sharded_update": (_i, [_vp, _vp, _vp, _i, _i, C.POINTER(C.c_int)]),
    "mi_rl_spectrum_rows": (_i, [_vp, _vp, _i, _i, _vp, _i]),
    "mi_rl_spectrum_row_floats": (_sz, [_vp]),
    "mi_rl_sharded_stage": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(C.c_int)]),
    "mi_rl_spectrum_rows_z": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "mi_rl_z_granule": (_i, [_vp]),
    "mi_rl_set_overlap": (_i, [_vp, _i, _i]),
    "mi_rl_overlap_probe": (_i, [_vp, _vp, _vp, _ip, _i, _f, _i, C.POINTER(C.c_float)]),
    "mi_prctile": (_i, [_i, _vp, _vp, _sz, C.POINTER(C.c_double), _i, C.POINTER(C.c_float)]),
    "mi_rescale_block": (_i, [_i, _vp, _vp, _vp, _sz, _i, _f, _f, _f, _f]),
    "mi_pad_center": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i]),
    "mi_crop_center": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i]),
    "mi_rl_create": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "mi_rl_create_ex": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _ip, _ip, _i, C.POINTER(_vp)]),
    "mi_rl_destroy": (_i, [_vp]),
    "mi_rl_engine": (_i, [_vp]),
    "mi_rl_separable": (_i, [_vp]),
    "mi_rl_pair_layout": (_i, [_vp]),
    "mi_rl_device_bytes": (_sz, [_vp]),
    "mi_rl_forward_ratio": (_i, [_vp, _vp, _vp, _vp]),
    "mi_rl_adjoint_update": (_i, [_vp, _vp, _vp, _vp, _f, _vp]),
    "mi_rl_iterate": (_i, [_vp, _vp, _vp, _vp, _i]),
    "mi_rl_time_pass": (_i, [_vp, _vp, _i, _vp, _i, C.POINTER(C.c_float)]),
    "mi_rl_fft_spectrum_bytes": (_sz, [_vp]),
    "mi_rl_time_between": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, C.POINTER(C.c_float)]),
    "mi_rl_fft_placement": (_i, [_vp, C.POINTER(C.c_float), _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mi_rl_reg_term": (_i, [_i, _vp, _vp, _vp, _i, _i, _i]),
    "mi_rl_spatial": (_i, [_i, _vp, _vp, _vp, _vp] + [_i] * 6 + [C.POINTER(RlOptions), _ip]),
    "mi_rl_fft": (_i, [_i, _vp, _vp, _vp] + [_i] * 9 + [C.POINTER(RlOptions), _ip]),
    "mi_rl_fft_wiener": (_i, [_i, _vp, _vp, _vp] + [_i] * 9 + [C.POINTER(RlOptions), _ip]),
    "mi_decon_plan_create": (_i, [_i, C.POINTER(_vp)]),
    "mi_decon_plan_run": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 6 + [C.POINTER(RlOptions), _i, _ip, _i, _ip]),
    "mi_decon_plan_destroy": (_i, [_vp]),
    "mi_load_block": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp] + [_i] * 6),
    "mi_destripe_z": (_i, [_i, _vp, _vp, _i, _i, _i, _f, _i]),
    "mi_destripe_max_levels": (_i, [_i, _i]),
    "mi_decon": (_i, [_i, _vp, _vp, _vp, _vp] + [_i] * 6 + [C.POINTER(RlOptions), _i, _ip, _i, _ip]),
    "mi_engine_select": (_i, [_i] * 7),
    "mi_next_fast_len": (_i, [_i]),
    "mi_fft_good_size": (_i, [_i, _i]),
    "mi_pack_rows": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mi_unpack_rows": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mi_tiff_codec": (C.c_char_p, []),
    "mi_tiff_info": (_i, [C.c_char_p, _ip, _ip, _ip, _ip]),
    "mi_tiff_read_box": (_i, [C.POINTER(C.c_char_p), _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "mi_tiff_write_series": (_i, [C.POINTER(C.c_char_p), _i, _vp, _i, _i, _i, _i, _i, _i, _ip]),
    "mi_tiff_write_series_device": (_i, [_i, _vp, C.POINTER(C.c_char_p), _i, _vp, _i, _i, _i, _i, _ip]),
    "mi_peer_link_create": (_i, [_i, _sz, C.POINTER(_vp), C.c_char_p, C.c_char_p]),
    "mi_peer_link_connect": (_i, [_vp, _i, C.c_char_p, C.c_char_p, _i]),
    "mi_peer_link_begin": (_i, [_vp, _vp, C.c_uint, _i]),
    "mi_peer_link_send": (_i, [_vp, _vp, C.c_uint, _i, _i, _sz, _sz, _vp, _vp]),
    "mi_peer_link_recv": (_i, [_vp, _vp, C.c_uint, _i, _i, _i, C.POINTER(_vp)]),
    "mi_peer_exchange": (_i, [_vp, _vp, C.c_uint, _i, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "mi_peer_link_status": (_i, [_vp, C.POINTER(_i)]),
    "mi_peer_link_disconnect": (_i, [_vp]),
    "mi_peer_link_destroy": (_i, [_vp]),
    # mi_crossmips.h
    "mi_ncc_default_params": (None, [_i, _i, _i, C.POINTER(NccParams)]),
    "mi_ncc_mips": (_i, [_i, _vp, _vp, _vp] + [_i] * 10 + [C.POINTER(NccParams), C.POINTER(NccDescr)]),
    "mi_ncc_mips_host": (_i, [_i, _vp, _vp, _vp] + [_i] * 10 + [C.POINTER(NccParams), C.POINTER(NccDescr)]),
    "mi_ncc_mips_batch_begin": (_i, [_i, _vp, _i, C.POINTER(_vp), _i, _f, _ip, _ip, _i, _i, _i, _ip, _ip, _i, _i, _i, _ip, C.POINTER(NccParams),
                                     C.POINTER(_vp)]),
    "mi_ncc_mips_batch_end": (_i, [_vp, C.POINTER(NccParams), C.POINTER(NccDescr)]),
    "mi_ncc_mips_batch": (_i, [_i, _vp, _i, C.POINTER(_vp), _ip, _ip, _i, _i, _i, _ip, _ip, _i, _i, _i, _ip,
                               C.POINTER(NccParams), C.POINTER(NccDescr)]),
    "mi_ncc_mips_batch_u16": (_i, [_i, _vp, _i, C.POINTER(_vp), C.c_float, _ip, _ip, _i, _i, _i, _ip, _ip, _i, _i, _i, _ip,
                                   C.POINTER(NccParams), C.POINTER(NccDescr)]),
    "mi_ncc_mips_batch_u8": (_i, [_i, _vp, _i, C.POINTER(_vp), C.c_float, _ip, _ip, _i, _i, _i, _ip, _ip, _i, _i, _i, _ip,
                                  C.POINTER(NccParams), C.POINTER(NccDescr)]),
    "mi_ncc_time_mips": (_i, [_i, _vp, _i, C.POINTER(_vp), _ip, _ip, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float)]),
    "mi_ncc_time_mips_u8": (_i, [_i, _vp, _i, C.POINTER(_vp), C.c_float, _ip, _ip, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float)]),
    "mi_ncc_time_mips_u16": (_i, [_i, _vp, _i, C.POINTER(_vp), C.c_float, _ip, _ip, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float)]),
    "mi_ncc_stats": (None, [C.POINTER(C.c_longlong), _i]),
    "mi_ncc_compute_mips": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i] + [_vp] * 6),
    "mi_ncc_compute_map": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mi_ncc_compute_map_lag": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads libmi_ipp.so once and binds every declared symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            f"(python -c 'import __graft_entry__ as g; g.build()' or make -C {os.path.join(_HERE, 'csrc')}). "
            "There is no CPU fallback.")
    # One HIP / HSA runtime per process: PyTorch ships its own copies of libamdhip64 / libhsa-runtime64.  When they are loaded
    # first, this library binds to them (same sonames); the other way round the process ends up with the system's HIP runtime
    # under this library and torch's HSA runtime under torch, and torch then finds "No HIP GPUs".  The host layer uses torch
    # for device memory and streams anyway, so it goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    handle = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return handle


def last_error() -> str:
    return lib().mi_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    if rc != MI_OK:
        raise MiError(rc, last_error())


def require_gpu() -> None:
    """Fail loudly when no HIP device is visible (the product path never computes on the CPU)."""
    n = lib().mi_device_count()
    if n <= 0:
        raise RuntimeError(f"no HIP device available (mi_device_count() = {n}: {last_error()}); "
                           "the MI355X path has no CPU fallback")


def release_cached_memory(device=None) -> int:
    """Gives the device memory the library keeps for reuse back to the driver (``mi_release_cached_memory``); returns bytes."""
    return int(lib().mi_release_cached_memory(-1 if device is None else int(device)))


def current_stream_ptr(device) -> int:
    """hipStream_t of torch's current stream on ``device`` as an integer for the void* parameter."""
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)
