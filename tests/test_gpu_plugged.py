"""The engine running INSIDE the reference's own driver (VERDICT r02 item 5 ii).

oracle/_ref/libetsi_ref_plugged.so (oracle/Makefile, target `plugged`; built in the container that has
/root/reference, shipped to the GPU box as a built file like oracle/_ref/libetsi_ref.so) = the reference's eleven
etsi/cpp/*.c, unmodified, + SeaPlugin.c extracted from INTEGRATION.md section 2 + oracle/plug_driver.c, linked against
libsea_mi355x.so.  plug_trace() runs the reference's DoAdvProcess (etsi/cpp/ParmInterface.c:208-330) once per frame
with the FEParamsX NoiseSup / CompCeps slots (ParmInterface.h:120-178, wired ParmInterface.c:58-82) either left as
AdvProcessAlloc set them (the reference's own C) or overwritten by SeaAdvProcessAlloc (the engine).
"""
import ctypes
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGGED = os.path.join(ROOT, "oracle", "_ref", "libetsi_ref_plugged.so")
GOLD = os.path.join(ROOT, "tests", "golden")


def _lib():
    if not os.path.exists(PLUGGED):
        pytest.skip("oracle/_ref/libetsi_ref_plugged.so not built (needs /root/reference: `make -C oracle plugged`)")
    import speech_enhancement_amd as sea
    sea.load()  # torch's HIP runtime first, then the product library the plugged build links against
    lib = ctypes.CDLL(PLUGGED)
    lib.plug_trace.restype = ctypes.c_long
    lib.plug_trace.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return lib


def _run(lib, x, plugged):
    x = np.ascontiguousarray(x, dtype=np.int16)
    nfr = len(x) // 80
    out = np.zeros(nfr * 80, np.int16)
    ceps = np.zeros((max(nfr, 1), 14), np.float32)
    counts = (ctypes.c_long * 2)()
    assert lib.plug_trace(x.ctypes.data, len(x), int(plugged), out.ctypes.data, ceps.ctypes.data, counts) == nfr
    return out, ceps[:counts[1]].copy(), int(counts[0])


def test_unplugged_driver_reproduces_the_reference_fixtures():
    """CPU only: with the slots untouched, plug_trace is the reference -- its int16 output and cepstra equal the
    committed golden vectors (tests/golden/ns_golden.npz, written from oracle/_ref/libetsi_ref.so)."""
    lib = _lib()
    g = np.load(os.path.join(GOLD, "ns_golden.npz"))
    for name in ("leading_zeros", "ragged", "gap"):
        x = g[f"{name}/in"]
        out, ceps, nout = _run(lib, x, plugged=False)
        full = len(x) // 80 * 80
        assert np.array_equal(out, g[f"{name}/etsi_denoise"][:full]), name
        assert np.array_equal(ceps.view(np.uint32), g[f"{name}/ceps"].view(np.uint32)), name
        assert nout * 80 == g[f"{name}/den_f32"].size, name


@pytest.mark.gpu
def test_plugged_reference_driver_is_bit_identical():
    """DoAdvProcess through SeaAdvProcessAlloc (NoiseSup and CompCeps on the MI355X, one launch per frame) against
    the same driver with the reference's own slots: int16 output, number of NoiseSup outputs and all 14 cepstral
    coefficients of every frame identical, bit for bit."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    lib = _lib()
    g = np.load(os.path.join(GOLD, "ns_golden.npz"))
    worst = 0.0
    for name in ("plain_1s", "leading_zeros", "loud", "gap"):
        x = g[f"{name}/in"]
        ref_out, ref_ceps, ref_n = _run(lib, x, plugged=False)
        got_out, got_ceps, got_n = _run(lib, x, plugged=True)
        assert got_n == ref_n, f"{name}: {got_n} NoiseSup outputs, the reference's slots give {ref_n}"
        assert np.array_equal(got_out, ref_out), f"{name}: {np.count_nonzero(got_out != ref_out)} int16 samples differ"
        assert got_ceps.shape == ref_ceps.shape, name
        worst = max(worst, float(np.abs(got_ceps - ref_ceps).max()))
        assert np.array_equal(got_ceps.view(np.uint32), ref_ceps.view(np.uint32)), \
            f"{name}: cepstra differ, max |delta| {np.abs(got_ceps - ref_ceps).max()}"
    print("plugged vs unplugged cepstra worst |delta| =", worst)
