"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end for the two checker libraries:

  * ``libsea_oracle.so``       the CPU restatement (oracle/ns_oracle.c, oracle/resynth_oracle.c)
  * ``_ref/libetsi_ref.so``    the reference's own C (etsi/cpp/*.c) compiled where it lies under
                               /root/reference, plus oracle/ref_driver*.c (only where it was built)
  * ``_ref/libaurora_ref.so``  the two files of the 16 k-native variant that compile on their own
                               (aurora_etsi/rfft.cpp, MelProc.cpp) + oracle/ref_driver_aurora.cpp

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this
module.  The product package ``speech_enhancement_amd`` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libsea_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libetsi_ref.so")
AURORA_SO = os.path.join(_HERE, "_ref", "libaurora_ref.so")
NSCAL = 16

_c_void = ctypes.c_void_p


def build(ref=True):
    """Compile the restatement (always) and the reference build (when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"] + (["ref"] if ref else []))


def _ptr(a):
    return a.ctypes.data_as(_c_void) if a is not None else None


class _Lib:
    """Common wrapper: the restatement and the reference driver export the same entry points
    under the prefixes ``ora_`` and ``ref_``."""

    def __init__(self, path, prefix):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = ctypes.CDLL(path)
        self.p = prefix

    def _f(self, name):
        return getattr(self.lib, self.p + name)

    def etsi_denoise(self, x, fill=0):
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.full(x.shape, fill, dtype=np.int16)
        self._f("etsi_denoise")(_ptr(x), _ptr(out), ctypes.c_long(x.size))
        return out

    def rfft(self, x, m=None):
        """rfft (x, n, m), etsi/cpp/rfft.c:45-180; m defaults to log2 n (any 2^m <= n is accepted, as by the reference)"""
        y = np.array(x, dtype=np.float32, copy=True)
        n = y.size
        if m is None:
            m = int(np.log2(n))
        assert 1 << int(np.log2(n)) == n and (1 << m) <= n
        self._f("rfft")(_ptr(y), ctypes.c_int(n), ctypes.c_int(m))
        return y

    def ns_trace(self, x, want_state=True):
        """Returns dict(out_i16, den_f32[nout*80], ceps[nceps,14], scal[nfr,16], spec[nfr,4,65])."""
        x = np.ascontiguousarray(x, dtype=np.int16)
        nfr = x.size // 80
        out = np.zeros(x.size, np.int16)
        den = np.zeros(max(nfr, 1) * 80, np.float32)
        ceps = np.zeros((max(nfr, 1), 14), np.float32)
        scal = np.zeros((max(nfr, 1), NSCAL), np.float32) if want_state else None
        spec = np.zeros((max(nfr, 1), 4, 65), np.float32) if want_state else None
        counts = np.zeros(2, np.int64)
        self._f("ns_trace")(_ptr(x), ctypes.c_long(x.size), _ptr(out), _ptr(den), _ptr(ceps),
                            _ptr(scal), _ptr(spec), _ptr(counts))
        nout, nceps = int(counts[0]), int(counts[1])
        return dict(out_i16=out, den_f32=den[: nout * 80], ceps=ceps[:nceps],
                    scal=None if scal is None else scal[:nfr],
                    spec=None if spec is None else spec[:nfr], nout=nout, nceps=nceps)

    def ns_stream_f32(self, frames):
        """DoNoiseSup on float frames [n, 80] (no zero-frame gate): (out[nout*80] float32, produced[n])."""
        x = np.ascontiguousarray(frames, dtype=np.float32).reshape(-1, 80)
        out = np.zeros(x.size, np.float32)
        prod = np.zeros(x.shape[0], np.int32)
        fn = self._f("ns_stream_f32")
        fn.restype = ctypes.c_long
        nout = fn(_ptr(x), ctypes.c_long(x.shape[0]), _ptr(out), _ptr(prod))
        return out[: nout * 80], prod

    def afe_trace(self, x):
        """The full per-frame chain: WaveProc -> CompCeps -> PostProc -> VAD + flush.
        Returns dict(flags[nfr,5] = SpeechFoundVar/Spec/Mel/VADNS + FrameCounter, feat_cc[nceps,14],
        feat_pp[nceps,14], vad_out[nvad,15] = emitted feature frames + VAD flag, nout, nceps, nvad)."""
        x = np.ascontiguousarray(x, dtype=np.int16)
        nfr = x.size // 80
        cap = max(nfr, 1) + 16
        flags = np.zeros((max(nfr, 1), 5), np.int32)
        fcc = np.zeros((cap, 14), np.float32)
        fpp = np.zeros((cap, 14), np.float32)
        vad = np.zeros((cap, 15), np.float32)
        counts = np.zeros(3, np.int64)
        self._f("afe_trace")(_ptr(x), ctypes.c_long(x.size), _ptr(flags), _ptr(fcc), _ptr(fpp), _ptr(vad),
                             _ptr(counts))
        nout, nceps, nvad = (int(c) for c in counts)
        return dict(flags=flags[:nfr], feat_cc=fcc[:nceps], feat_pp=fpp[:nceps], vad_out=vad[:nvad],
                    nout=nout, nceps=nceps, nvad=nvad)

    def compceps_frame(self, data201):
        d = np.ascontiguousarray(data201, dtype=np.float32)
        assert d.size == 201
        c = np.zeros(14, np.float32)
        self._f("compceps_frame")(_ptr(d), _ptr(c))
        return c

    def ns_tables(self):
        t = dict(sigWindow=np.zeros(200, np.float32), irWindow=np.zeros(17, np.float32),
                 idct=np.zeros((25, 25), np.float32), melStart=np.zeros(25, np.int32),
                 melLen=np.zeros(25, np.int32), melData=np.zeros((25, 16), np.float32))
        self._f("ns_tables")(*[_ptr(t[k]) for k in
                               ("sigWindow", "irWindow", "idct", "melStart", "melLen", "melData")])
        return t

    def cc_tables(self):
        t = dict(hamming=np.zeros(100, np.float32), dct=np.zeros((12, 23), np.float32),
                 melStart=np.zeros(23, np.int32), melLen=np.zeros(23, np.int32),
                 melData=np.zeros((23, 32), np.float32))
        self._f("cc_tables")(*[_ptr(t[k]) for k in ("hamming", "dct", "melStart", "melLen", "melData")])
        return t


class Oracle(_Lib):
    def __init__(self, path=None):
        """path: another build of the SAME restatement (bench.py's -O3 -march=native timing build)."""
        if path is None and not os.path.exists(ORACLE_SO):
            build(ref=False)
        super().__init__(path or ORACLE_SO, "ora_")
        self.lib.ora_resynth64.restype = ctypes.c_int

    def resynth64(self, x, mask, binary=False, frames_l_over_160=False):
        x = np.ascontiguousarray(x, dtype=np.int16)
        mask = np.ascontiguousarray(mask, dtype=np.float32)
        out = np.zeros(x.size, np.int16)
        mode = int(bool(binary)) | (2 if frames_l_over_160 else 0)
        rc = self.lib.ora_resynth64(_ptr(x), ctypes.c_long(x.size), _ptr(mask),
                                    ctypes.c_int(mask.shape[0]), ctypes.c_int(mode), _ptr(out))
        if rc:
            raise ValueError("ora_resynth64: bad L/F")
        return out

    def gammatone(self, x, cf, bw, mid_ear):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x)
        self.lib.ora_gammatone(_ptr(x), _ptr(y), ctypes.c_float(cf), ctypes.c_float(bw),
                               ctypes.c_float(mid_ear), ctypes.c_long(x.size))
        return y

    def haircell(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.zeros_like(x)
        self.lib.ora_haircell(_ptr(x), _ptr(y), ctypes.c_long(x.size))
        return y

    def subband64(self, x):
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros((64, x.size), np.int16)
        rc = self.lib.ora_subband64(_ptr(x), ctypes.c_long(x.size), _ptr(out))
        if rc:
            raise ValueError("ora_subband64: bad L")
        return out

    def irm_target(self, pure64, noise64, window=1):
        """make_single_IBM's IRM target from two [64][L] int16 subband blocks -> float32 [F][64]."""
        pure64 = np.ascontiguousarray(pure64, dtype=np.int16)
        noise64 = np.ascontiguousarray(noise64, dtype=np.int16)
        assert pure64.shape == noise64.shape and pure64.shape[0] == 64
        L = pure64.shape[1]
        F = (L - 320) // 160 + 1
        out = np.zeros((max(F, 1), 64), np.float32)
        self.lib.ora_irm_target.restype = ctypes.c_int
        rc = self.lib.ora_irm_target(_ptr(pure64), _ptr(noise64), ctypes.c_long(L), ctypes.c_long(L),
                                     ctypes.c_int(window), _ptr(out))
        if rc:
            raise ValueError("ora_irm_target: bad L")
        return out[:F]

    def resynth_channels(self):
        cf, bw, me = (np.zeros(64, np.float32) for _ in range(3))
        self.lib.ora_resynth_channels(_ptr(cf), _ptr(bw), _ptr(me))
        return cf, bw, me

    # ---- the 16 k-native NoiseSup variant (ns16k_oracle.c) ----
    def ns16k_new(self):
        return Ns16k(self.lib)

    def ns16k_rfft(self, x, n=512, m=8):
        y = np.array(x, dtype=np.float32, copy=True)
        self.lib.ora16_rfft(_ptr(y), ctypes.c_int(n), ctypes.c_int(m))
        return y

    def ns16k_tables(self):
        sw, iw, gs = np.zeros(480, np.float32), np.zeros(17, np.float32), np.zeros(25, np.int32)
        g, d = np.zeros((25, 128), np.float32), np.zeros((25, 25), np.float32)
        self.lib.ora16_tables(_ptr(sw), _ptr(iw), _ptr(gs), _ptr(g), _ptr(d))
        return dict(sigWindow=sw, irWindow=iw, gammaStart=gs, gamma=g, idct=d)

    def ns16k_do_gamma(self, W):
        y = np.array(W, dtype=np.float32, copy=True)
        assert y.size >= 128
        self.lib.ora16_do_gamma(_ptr(y))
        return y[:25]

    def ns16k_idct(self, W):
        y = np.array(W, dtype=np.float32, copy=True)
        assert y.size >= 25
        self.lib.ora16_idct(_ptr(y))
        return y[:25]


class Ns16k:
    """One thread instance of the 16 k-native variant (etsi_denoise_mapping_thread_init .. thread_release)."""

    UNTOUCHED = -7

    def __init__(self, lib):
        self.lib = lib
        lib.ora16_new.restype = ctypes.c_void_p
        lib.ora16_push.restype = ctypes.c_long
        self.h = ctypes.c_void_p(lib.ora16_new())

    def push(self, x):
        """func_Wiener on float samples x: dict(out[n*160] float32, var, spec, mel, vadns, counter [n] int32 -- entries
        the reference leaves untouched hold UNTOUCHED --, wiener [rows, 25])."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size // 160
        out = np.full(max(n, 1) * 160, float(self.UNTOUCHED), np.float32)
        arrs = [np.full(max(n, 1), self.UNTOUCHED, np.int32) for _ in range(5)]
        w = np.zeros((max(n, 1), 25), np.float32)
        rows = ctypes.c_long(0)
        self.lib.ora16_push(self.h, _ptr(x), ctypes.c_long(x.size), _ptr(out), *[_ptr(a) for a in arrs], _ptr(w), ctypes.byref(rows))
        return dict(out=out[: n * 160], var=arrs[0][:n], spec=arrs[1][:n], mel=arrs[2][:n], vadns=arrs[3][:n], counter=arrs[4][:n],
                    wiener=w[: rows.value])

    def __del__(self):
        if self.h:
            self.lib.ora16_free(self.h)
            self.h = None


class AuroraReference:
    """rfft.cpp + MelProc.cpp of the 16 k-native variant, compiled where they lie (present only where it was built)."""

    def __init__(self):
        if not os.path.exists(AURORA_SO):
            raise FileNotFoundError(AURORA_SO)
        self.lib = ctypes.CDLL(AURORA_SO)

    def rfft(self, x, n=512, m=8):
        y = np.array(x, dtype=np.float32, copy=True)
        self.lib.ref16_rfft(_ptr(y), ctypes.c_int(n), ctypes.c_int(m))
        return y

    def tables(self):
        gs, gl = np.zeros(25, np.int32), np.zeros(25, np.int32)
        g, d = np.zeros((25, 128), np.float32), np.zeros((25, 25), np.float32)
        self.lib.ref16_tables(_ptr(gs), _ptr(gl), _ptr(g), _ptr(d))
        return dict(gammaStart=gs, gammaLen=gl, gamma=g, idct=d)

    def do_gamma(self, W):
        y = np.array(W, dtype=np.float32, copy=True)
        self.lib.ref16_do_gamma(_ptr(y))
        return y[:25]

    def idct(self, W):
        y = np.array(W, dtype=np.float32, copy=True)
        self.lib.ref16_idct(_ptr(y))
        return y[:25]


def have_aurora_reference():
    return os.path.exists(AURORA_SO)


class Reference(_Lib):
    """The reference C itself (present only where oracle/_ref was built)."""

    def __init__(self):
        super().__init__(REF_SO, "ref_")


def have_reference():
    return os.path.exists(REF_SO)


# ---------------------------------------------------------------------------------------------
# Seeded signals of SURVEY.md 8(c) (the known-answer inputs)
# ---------------------------------------------------------------------------------------------
def _lcg_stream(seed, n):
    """s <- s*1664525 + 1013904223 (uint32), n successive values AFTER each update."""
    out = np.empty(n, np.uint32)
    s = seed & 0xFFFFFFFF
    for i in range(n):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = s
    return out


def kat_ns_signal(L=160000, seed=12345):
    """x[i] = (short)(4000 sin(2 pi 440 i/8000) [(i mod 8000) < 4000] + ((int)(s>>16) % 2001 - 1000))"""
    s = _lcg_stream(seed, L)
    i = np.arange(L)
    tone = 4000.0 * np.sin(2 * np.pi * 440 * i / 8000.0) * ((i % 8000) < 4000)
    v = tone + ((s >> 16).astype(np.int64) % 2001 - 1000)
    return np.trunc(v).astype(np.int16)


def kat_resynth_case(L=48000, seed=1):
    """in[i] = (short)(3000 sin(2 pi 500 i/16000) + ((s>>16) % 1001 - 500)); then the SAME LCG
    continues: m[f][c] = ((s>>16) % 1000)/1000.0f, F = (L-320)/160+1 rows of 64."""
    F = (L - 320) // 160 + 1
    s = _lcg_stream(seed, L + F * 64)
    i = np.arange(L)
    v = 3000.0 * np.sin(2 * np.pi * 500 * i / 16000.0) + ((s[:L] >> 16).astype(np.int64) % 1001 - 500)
    x = np.trunc(v).astype(np.int16)
    m = (((s[L:] >> 16) % 1000).astype(np.float32) / np.float32(1000.0)).reshape(F, 64)
    return x, m


def weighted_checksum(out):
    out = np.asarray(out).astype(np.int64)
    return int(np.sum(out * ((np.arange(out.size) % 97) + 1)))
