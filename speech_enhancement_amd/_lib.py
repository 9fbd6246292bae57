"""ctypes binding of libsea_mi355x.so (include/sea_mi355x.h).

The library is the product; this module only loads it and declares the prototypes.  There is no
fallback of any kind: if the shared object is missing (``python -c "import __graft_entry__ as g;
g.build()"`` builds it) or no gfx950 device is usable, calls raise.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEA_MI355X_LIB selects another build of the SAME library (timing-only diagnostic variants)
LIB_PATH = os.environ.get("SEA_MI355X_LIB") or os.path.join(_HERE, "libsea_mi355x.so")

_c = ctypes
_vp, _i, _ll, _l = _c.c_void_p, _c.c_int, _c.c_longlong, _c.c_long

# name -> (restype, argtypes); every function include/sea_mi355x.h declares
PROTOTYPES = {
    "etsi_denoise": (_i, [_vp, _vp, _l]),
    "etsi_denoise_synchronization": (_i, [_vp, _vp, _l]),
    "etsi_denoise_16k": (_i, [_vp, _vp, _l]),
    "etsi_denoise_16k_synchronization": (_i, [_vp, _vp, _l]),
    "rfft": (None, [_vp, _i, _i]),
    "sea_init": (_i, [_i]),
    "sea_device_count": (_i, []),
    "sea_last_error": (_c.c_char_p, []),
    "sea_version": (_c.c_char_p, []),
    "sea_tables_host": (_i, [_vp] * 11),
    "sea_gammatone_channels": (_i, [_vp, _vp, _vp]),
    "sea_ns_denoise_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sea_rfft256_batch": (_i, [_vp, _vp, _ll, _vp]),
    "sea_rfft_batch": (_i, [_vp, _i, _i, _ll, _vp]),
    "sea_compceps_frames": (_i, [_vp, _vp, _ll, _vp]),
    "sea_compceps_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _i, _vp]),
    "sea_ns_kernel_form": (_i, [_i]),
    "sea_ns_denoise_batch_fd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sea_afe_features_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sea_resynth64_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sea_resynth_scratch_bytes": (_ll, [_ll, _i]),
    "sea_denoise_utterances": (_i, [_vp, _vp, _vp, _i]),
    "sea_host_threads": (_i, []),
    "sea_packed_create": (_vp, []),
    "sea_packed_destroy": (None, [_vp]),
    "sea_packed_plan": (_i, [_vp, _vp, _i]),
    "sea_packed_slices": (_i, [_vp]),
    "sea_packed_segments": (_i, [_vp, _i, _vp, _vp, _vp, _i]),
    "sea_packed_denoise": (_i, [_vp]),
    "sea_denoise_ceps_utterances": (_i, [_vp, _vp, _vp, _vp, _vp, _i]),
    "sea_compceps_frame": (_i, [_vp, _vp]),
    "sea_resynth64": (_i, [_vp, _l, _vp, _i, _i, _vp]),
    "sea_resynth_utterances": (_i, [_vp, _vp, _vp, _i, _vp, _i]),
    "sea_subband64": (_i, [_vp, _l, _vp]),
    "sea_subband64_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sea_irm_target": (_i, [_vp, _vp, _l, _i, _vp]),
    "sea_irm_target_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sea_gammatone_filter": (_i, [_vp, _vp, _i, _l]),
    "sea_ns_stream_alloc": (_vp, []),
    "sea_ns_stream_init": (None, [_vp]),
    "sea_ns_stream_push": (_i, [_vp, _vp, _vp]),
    "sea_ns_stream_delete": (None, [_vp]),
    "sea_ns_streams_push": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sea_ns_denoise_batch_slice": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sea_ns_slice_state_floats": (_i, []),
    "sea_ns_streams_push_fd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sea_ns_state_floats": (_i, []),
    "sea_ns16k_streams_push": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sea_ns16k_state_floats": (_i, []),
    "sea_ns16k_kernel_form": (_i, [_i]),
    "sea_ns16k_tables_host": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "sea_ns16k_fft_host": (None, [_vp]),
    "etsi_denoise_mapping_global_init": (_i, [_vp, _vp]),
    "etsi_denoise_mapping_thread_init": (_i, [_vp, _vp]),
    "etsi_denoise_mapping_func_Wiener": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "etsi_denoise_mapping_func": (_i, [_vp, _vp, _vp, _vp]),
    "etsi_denoise_mapping_thread_release": (None, [_vp]),
    "etsi_denoise_mapping_global_release": (None, [_vp]),
    "sea_selftest_pi4": (_i, [_vp]),
    "sea_selftest_div": (_i, [_vp]),
    "sea_selftest_nsdiv": (_i, [_vp]),
    "sea_selftest_dc": (_i, [_vp, _vp, _vp, _vp, _i]),
    "sea_selftest_log": (_i, [_vp, _vp, _i]),
    "sea_selftest_log_dd": (_i, [_vp, _vp, _vp, _i]),
    "sea_selftest_log_sites": (_i, [_vp, _vp, _vp, _i]),
    "sea_selftest_log_guard": (_i, [_i, _vp, _vp, _i]),
    "sea_selftest_hostpipe_fault": (_i, [ctypes.c_longlong]),
    "sea_selftest_ns16k_pieces": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
}

_lib = None


class SeaError(RuntimeError):
    pass


def load():
    """Load the shared object once.  torch (when importable) is imported FIRST so that the HIP
    runtime the library binds to is the one torch already mapped: both carry the soname
    libamdhip64.so.7 and a process must not hold two HIP runtimes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SeaError(
            f"{LIB_PATH} is missing: build it with `make -C speech_enhancement_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().sea_last_error().decode(errors="replace")
        raise SeaError(f"{what} failed: {msg}")
