"""Host-side mirror of the reference's interface for the hot path, over libsea_mi355x.so.

Names follow the reference:
  etsi_denoise(x)              etsi/cpp/AdvFrontEnd.c:125       (host arrays, the C drop-in itself)
  rfft(x)                      etsi/cpp/rfft.c:45
  DoCompCeps(data201)          etsi/cpp/CompCeps.c:309
  NoiseSup                     DoNoiseSupAlloc/Init/DoNoiseSup/Delete (etsi/cpp/NoiseSup.c:859-1440)
  resynth(x, mask, binary)     resyth_64sub_{ori,IBM}/cpp/extractwav.cpp:9
and the batched, HBM-resident forms the GPU wants (PackedBatch + *_batch).

torch is used for device memory and streams only (plumbing); all compute is in the HIP library.
"""
import ctypes

import numpy as np

from . import _lib


def _np_ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------------------------------------
# drop-ins on host arrays (these call the C symbols with the reference's own signatures)
# ------------------------------------------------------------------------------------------------
def etsi_denoise(x, fill=0):
    """int etsi_denoise(short*, short*, long).  Samples beyond the last full 80-sample frame are
    not written (they keep ``fill``), exactly like the reference."""
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.int16)
    out = np.full(x.shape, fill, dtype=np.int16)
    rc = lib.etsi_denoise(_np_ptr(x), _np_ptr(out), x.size)
    _lib.check(rc, "etsi_denoise")
    return out


def rfft(x, m=None):
    """void rfft(float*, n, m) (etsi/cpp/rfft.h:19): returns Re(0..n/2), Im(n/2-1..1); m defaults to log2 n."""
    lib = _lib.load()
    y = np.array(x, dtype=np.float32, copy=True)
    n = int(y.size)
    if n < 2 or n & (n - 1):
        raise ValueError("rfft: n must be a power of two")
    lib.rfft(_np_ptr(y), n, int(m) if m is not None else n.bit_length() - 1)
    return y


def DoCompCeps(data201):
    """DoCompCeps(Data, Coef, This) with data201[0] = Data[-1]; returns c1..c12, c0, logE."""
    lib = _lib.load()
    d = np.ascontiguousarray(data201, dtype=np.float32)
    if d.size != 201:
        raise ValueError("DoCompCeps needs Data[-1..199] (201 floats)")
    coef = np.zeros(14, np.float32)
    ptr = ctypes.c_void_p(d.ctypes.data + 4)
    _lib.check(lib.sea_compceps_frame(ptr, _np_ptr(coef)), "DoCompCeps")
    return coef


def resynth(x, mask, binary=False, frames_l_over_160=False):
    """resynth(): x int16[L], mask float32[F][64] with F=(L-320)/160+1 (or L/160 with
    frames_l_over_160, the frame count of 1dnn_resynth/extractwav.cpp:67) -> int16[L]."""
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.int16)
    mask = np.ascontiguousarray(mask, dtype=np.float32)
    out = np.zeros(x.size, np.int16)
    mode = int(bool(binary)) | (2 if frames_l_over_160 else 0)
    rc = lib.sea_resynth64(_np_ptr(x), x.size, _np_ptr(mask), int(mask.shape[0]), mode, _np_ptr(out))
    _lib.check(rc, "resynth")
    return out


def subbband(x):
    """subbband(): x int16[L] -> int16[64][L] (gammatone + hair cell per channel)."""
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.int16)
    out = np.zeros((64, x.size), np.int16)
    _lib.check(lib.sea_subband64(_np_ptr(x), x.size, _np_ptr(out)), "subbband")
    return out


def irm_target(pure64, noise64, window=1):
    """make_single_IBM's IRM target (enhancement_extract_test/cpp/show_IBM.cpp:105-169) from the [64][L] int16
    subband blocks of the clean and the noise signal (two subbband() outputs) -> float32 [F][64], the mask matrix
    resynth() takes.  window: 0 rectangular, 1 Hamming, 2 Hanning (asdk::SpecInfo's is unknown: parity unpinned)."""
    lib = _lib.load()
    pure64 = np.ascontiguousarray(pure64, dtype=np.int16)
    noise64 = np.ascontiguousarray(noise64, dtype=np.int16)
    if pure64.shape != noise64.shape or pure64.ndim != 2 or pure64.shape[0] != 64:
        raise ValueError("irm_target needs two [64][L] int16 blocks")
    L = pure64.shape[1]
    out = np.zeros((max((L - 320) // 160 + 1, 1), 64), np.float32)
    _lib.check(lib.sea_irm_target(_np_ptr(pure64), _np_ptr(noise64), L, int(window), _np_ptr(out)), "irm_target")
    return out


def gammaToneFilter(x, chan):
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x)
    _lib.check(lib.sea_gammatone_filter(_np_ptr(x), _np_ptr(y), int(chan), x.size), "gammaToneFilter")
    return y


class NoiseSup:
    """The FEParamsX NoiseSup slot (DoNoiseSupAlloc / Init / DoNoiseSup / Delete) on the GPU: one
    stream whose recursive state stays in HBM between 80-sample pushes."""

    def __init__(self):
        self._lib = _lib.load()
        self._h = self._lib.sea_ns_stream_alloc()
        if not self._h:
            raise _lib.SeaError("DoNoiseSupAlloc failed: " + self._lib.sea_last_error().decode())
        self._lib.sea_ns_stream_init(self._h)

    def init(self):
        self._lib.sea_ns_stream_init(self._h)

    def DoNoiseSup(self, in80):
        x = np.ascontiguousarray(in80, dtype=np.float32)
        assert x.size == 80
        out = np.zeros(80, np.float32)
        produced = self._lib.sea_ns_stream_push(self._h, _np_ptr(x), _np_ptr(out))
        return bool(produced), out

    def close(self):
        if self._h:
            self._lib.sea_ns_stream_delete(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tables():
    """The constant tables the library computed (for tests against the oracle's)."""
    lib = _lib.load()
    t = dict(sigWindow=np.zeros(200, np.float32), irWindow=np.zeros(17, np.float32),
             idct=np.zeros((25, 25), np.float32), melStart=np.zeros(25, np.int32),
             melLen=np.zeros(25, np.int32), melData=np.zeros((25, 16), np.float32),
             hamming=np.zeros(100, np.float32), dct=np.zeros((12, 23), np.float32),
             ccStart=np.zeros(23, np.int32), ccLen=np.zeros(23, np.int32),
             ccData=np.zeros((23, 32), np.float32))
    keys = ("sigWindow", "irWindow", "idct", "melStart", "melLen", "melData", "hamming", "dct",
            "ccStart", "ccLen", "ccData")
    _lib.check(lib.sea_tables_host(*[_np_ptr(t[k]) for k in keys]), "sea_tables_host")
    cf, bw, me = (np.zeros(64, np.float32) for _ in range(3))
    _lib.check(lib.sea_gammatone_channels(_np_ptr(cf), _np_ptr(bw), _np_ptr(me)), "sea_gammatone_channels")
    t.update(cf=cf, bw=bw, midEar=me)
    return t


# ------------------------------------------------------------------------------------------------
# batched, HBM-resident forms
# ------------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _stream_ptr():
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def launch_order(lengths, n_cu=256):
    """Launch order of the utterance-per-workgroup kernels: longest first (the dispatcher serves the
    oldest waves first, so the critical path -- the longest utterance -- starts at once and keeps
    priority), in rows of n_cu workgroups with every other row reversed.  Workgroups b, b+n_cu,
    b+2 n_cu, ... of a launch share a CU (tools/hwid_probe.hip), so the serpentine gives each CU a
    long, a short, a medium-long and a medium-short utterance instead of the four longest of their
    rows (measured on the bench corpus: 3.78 -> 3.67 ms, tools/order_sweep.py)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable").astype(np.int32)
    for k, start in enumerate(range(0, len(order), n_cu)):
        if k & 1:
            order[start:start + n_cu] = order[start:start + n_cu][::-1].copy()
    return order


class PackedBatch:
    """A batch of utterances packed back to back in one int16 tensor resident in HBM.

    data     int16 [total]     utterance u at data[offsets[u] : offsets[u]+lengths[u]]
    offsets  int64 [n]         multiples of 8 samples (16-byte aligned rows)
    lengths  int64 [n]
    order    int32 [n]         launch order (see launch_order)
    """

    def __init__(self, data, offsets, lengths, order, host_offsets, host_lengths):
        self.data, self.offsets, self.lengths, self.order = data, offsets, lengths, order
        self.host_offsets, self.host_lengths = host_offsets, host_lengths

    @property
    def n_utt(self):
        return len(self.host_lengths)

    @property
    def total(self):
        return int(self.data.numel())

    @property
    def n_frames(self):
        """NoiseSup frames (80 samples) in the batch."""
        return int(sum(int(l) // 80 for l in self.host_lengths))

    @staticmethod
    def layout(lengths):
        lengths = np.asarray(lengths, dtype=np.int64)
        padded = (lengths + 7) // 8 * 8
        offsets = np.concatenate(([0], np.cumsum(padded)[:-1])).astype(np.int64) if len(lengths) else np.zeros(0, np.int64)
        total = int(padded.sum())
        return offsets, total, launch_order(lengths)

    @classmethod
    def from_arrays(cls, utterances, device="cuda"):
        torch = _torch()
        lengths = np.array([len(u) for u in utterances], dtype=np.int64)
        offsets, total, order = cls.layout(lengths)
        host = np.zeros(max(total, 8), np.int16)
        for u, off in zip(utterances, offsets):
            host[off:off + len(u)] = u
        return cls(torch.from_numpy(host).to(device), torch.from_numpy(offsets).to(device),
                   torch.from_numpy(lengths).to(device), torch.from_numpy(order).to(device), offsets, lengths)

    def like(self, fill=0, dtype=None):
        torch = _torch()
        return torch.full_like(self.data, fill, dtype=dtype or self.data.dtype)

    def split(self, tensor, full_frames_only=False, hop=80):
        """Cut a packed result tensor back into per-utterance numpy arrays."""
        host = tensor.detach().cpu().numpy()
        out = []
        for off, L in zip(self.host_offsets, self.host_lengths):
            n = int(L) // hop * hop if full_frames_only else int(L)
            out.append(host[off:off + n].copy())
        return out


def ns_denoise_batch(batch, out=None, want_f32=False, use_order=True):
    """etsi_denoise semantics for every utterance of the batch, one launch.  Returns
    (out_int16, out_f32 or None, first_out int32[n]).  Asynchronous on the current stream."""
    torch = _torch()
    lib = _lib.load()
    if out is None:
        out = torch.zeros_like(batch.data)
    f32 = torch.zeros(batch.total, dtype=torch.float32, device=batch.data.device) if want_f32 else None
    first = torch.full((batch.n_utt,), -1, dtype=torch.int32, device=batch.data.device)
    rc = lib.sea_ns_denoise_batch(_dptr(batch.data), _dptr(out), _dptr(f32), _dptr(batch.offsets),
                                  _dptr(batch.lengths), _dptr(batch.order) if use_order else None,
                                  _dptr(first), batch.n_utt, _stream_ptr())
    _lib.check(rc, "sea_ns_denoise_batch")
    return out, f32, first


def compceps_batch(batch, den_f32, first_out):
    """CompCeps on the float NoiseSup stream.  Returns (ceps float32 [total,14], ceps_cum int64
    [n+1] on host, n_ceps int32[n] tensor)."""
    torch = _torch()
    lib = _lib.load()
    cap = np.maximum(np.asarray(batch.host_lengths) // 80 - 6, 0).astype(np.int64)
    cum = np.concatenate(([0], np.cumsum(cap))).astype(np.int64)
    total = int(cum[-1])
    dev = batch.data.device
    ceps = torch.zeros((max(total, 1), 14), dtype=torch.float32, device=dev)
    n_ceps = torch.zeros(batch.n_utt, dtype=torch.int32, device=dev)
    d_cum = torch.from_numpy(cum).to(dev)
    rc = lib.sea_compceps_batch(_dptr(den_f32), _dptr(batch.offsets), _dptr(batch.lengths), _dptr(first_out),
                                _dptr(d_cum), total, _dptr(ceps), _dptr(n_ceps), batch.n_utt, _stream_ptr())
    _lib.check(rc, "sea_compceps_batch")
    return ceps, cum, n_ceps


def afe_features_batch(batch, want_intermediates=False):
    """The full ETSI AFE feature chain of the reference's DoAdvProcess as it was before the author
    commented it out (etsi/cpp/ParmInterface.c:274-311): NoiseSup -> WaveProc -> CompCeps ->
    PostProc -> frame-dropping VAD, with FlushAdvProcess at the end.

    Returns a dict: feats (list of float32 [n_u,15] host arrays: c1..c12, c0, logE, VAD flag per
    emitted frame), out (int16 denoised audio tensor, as ns_denoise_batch), and with
    want_intermediates also flags (per output frame speech bits), feat_cc / feat_pp (per cepstral
    frame, after CompCeps / PostProc), ceps_cum, n_ceps, first_out, onset."""
    torch = _torch()
    lib = _lib.load()
    dev = batch.data.device
    n = batch.n_utt
    out = torch.zeros_like(batch.data)
    f32 = torch.zeros(batch.total, dtype=torch.float32, device=dev)
    first = torch.full((n,), -1, dtype=torch.int32, device=dev)
    onset = torch.zeros(n, dtype=torch.int32, device=dev)
    flags = torch.zeros(max(batch.total // 8, 1), dtype=torch.uint8, device=dev)
    _lib.check(lib.sea_ns_denoise_batch_fd(_dptr(batch.data), _dptr(out), _dptr(f32), _dptr(batch.offsets),
                                           _dptr(batch.lengths), _dptr(batch.order), _dptr(first), _dptr(flags),
                                           _dptr(onset), n, _stream_ptr()), "sea_ns_denoise_batch_fd")
    nfr = np.asarray(batch.host_lengths) // 80
    ccap = np.maximum(nfr - 6, 0).astype(np.int64)
    ccum = np.concatenate(([0], np.cumsum(ccap))).astype(np.int64)
    fcap = (nfr + 6).astype(np.int64)
    fcum = np.concatenate(([0], np.cumsum(fcap))).astype(np.int64)
    tc, tf = int(ccum[-1]), int(fcum[-1])
    feat_cc = torch.zeros((max(tc, 1), 14), dtype=torch.float32, device=dev)
    feat_pp = torch.zeros((max(tc, 1), 14), dtype=torch.float32, device=dev) if want_intermediates else None
    feat15 = torch.zeros((max(tf, 1), 15), dtype=torch.float32, device=dev)
    n_feat = torch.zeros(n, dtype=torch.int32, device=dev)
    n_ceps = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ccum, d_fcum = torch.from_numpy(ccum).to(dev), torch.from_numpy(fcum).to(dev)
    _lib.check(lib.sea_afe_features_batch(_dptr(f32), _dptr(flags), _dptr(batch.offsets), _dptr(batch.lengths),
                                          _dptr(first), _dptr(onset), _dptr(d_ccum), tc, _dptr(feat_cc),
                                          _dptr(feat_pp) if feat_pp is not None else None, _dptr(d_fcum),
                                          _dptr(feat15), _dptr(n_feat), _dptr(n_ceps), n, _stream_ptr()),
               "sea_afe_features_batch")
    torch.cuda.synchronize()
    host15, nf = feat15.cpu().numpy(), n_feat.cpu().numpy()
    res = dict(feats=[host15[fcum[u]:fcum[u] + int(nf[u])] for u in range(n)], out=out)
    if want_intermediates:
        res.update(flags=flags, feat_cc=feat_cc, feat_pp=feat_pp, ceps_cum=ccum, n_ceps=n_ceps, first_out=first,
                   onset=onset, den_f32=f32)
    return res


def rfft_batch(frames):
    """frames: float32 tensor [n,256] on the GPU -> rfft of every row."""
    torch = _torch()
    lib = _lib.load()
    frames = frames.contiguous()
    out = torch.empty_like(frames)
    _lib.check(lib.sea_rfft256_batch(_dptr(frames), _dptr(out), frames.shape[0], _stream_ptr()), "sea_rfft256_batch")
    return out


def rfft_any_batch(frames, m=None):
    """frames: float32 tensor [k, n] on the GPU -> rfft (x, n, m) of every row, any size the reference's routine takes
    (etsi/cpp/rfft.c:45-180: n a power of two, 2^m <= n; m defaults to log2 n)."""
    lib = _lib.load()
    out = frames.contiguous().clone()
    n = int(out.shape[1])
    if m is None:
        m = n.bit_length() - 1
    _lib.check(lib.sea_rfft_batch(_dptr(out), n, int(m), out.shape[0], _stream_ptr()), "sea_rfft_batch")
    return out


def compceps_frames(frames201):
    """frames201: float32 tensor [n,201] on the GPU -> [n,14]."""
    torch = _torch()
    lib = _lib.load()
    frames201 = frames201.contiguous()
    out = torch.empty((frames201.shape[0], 14), dtype=torch.float32, device=frames201.device)
    _lib.check(lib.sea_compceps_frames(_dptr(frames201), _dptr(out), frames201.shape[0], _stream_ptr()),
               "sea_compceps_frames")
    return out


class MaskBatch:
    """Per-utterance mask matrices [F_u][64] packed row-wise, F_u = (L_u-320)/160+1."""

    def __init__(self, data, row_offsets, host_row_offsets, host_rows):
        self.data, self.row_offsets = data, row_offsets
        self.host_row_offsets, self.host_rows = host_row_offsets, host_rows

    @classmethod
    def from_arrays(cls, masks, device="cuda"):
        torch = _torch()
        rows = np.array([m.shape[0] for m in masks], dtype=np.int64)
        offs = np.concatenate(([0], np.cumsum(rows)[:-1])).astype(np.int64)
        host = np.concatenate([np.ascontiguousarray(m, dtype=np.float32) for m in masks], axis=0)
        return cls(torch.from_numpy(host).to(device), torch.from_numpy(offs).to(device), offs, rows)


def resynth_scratch_elems(batch):
    """float32 elements of the HBM intermediate of resynth_batch (sea_resynth_scratch_bytes / 4)."""
    return int(_lib.load().sea_resynth_scratch_bytes(int(batch.total), int(batch.n_utt))) // 4


def resynth_batch(batch, masks, binary=False, out=None, scratch=None, use_order=True, frames_l_over_160=False):
    """64-band gammatone resynthesis of every utterance of the batch (two launches on the current
    stream).  ``scratch`` (float32, resynth_scratch_elems(batch) elements) may be passed to reuse the HBM-resident
    analysis intermediate between calls."""
    torch = _torch()
    lib = _lib.load()
    if np.any(np.asarray(batch.host_lengths) < (160 if frames_l_over_160 else 320)):
        raise ValueError("resynth needs utterances of at least one mask frame")
    if out is None:
        out = torch.zeros_like(batch.data)
    if scratch is None:
        scratch = torch.empty(resynth_scratch_elems(batch), dtype=torch.float32, device=batch.data.device)
    rc = lib.sea_resynth64_batch(_dptr(batch.data), _dptr(out), _dptr(batch.offsets), _dptr(batch.lengths),
                                 _dptr(masks.data), _dptr(masks.row_offsets), _dptr(scratch),
                                 _dptr(batch.order) if use_order else None, batch.n_utt,
                                 int(bool(binary)) | (2 if frames_l_over_160 else 0), _stream_ptr())
    _lib.check(rc, "sea_resynth64_batch")
    return out, scratch


def subband_batch(batch, out=None, use_order=True):
    """subbband() for every utterance of the batch.  Returns an int16 tensor of 64x the packed
    size: utterance u's [64][pitch] block starts at offsets[u]*64, pitch = length rounded up to 8."""
    torch = _torch()
    lib = _lib.load()
    if out is None:
        out = torch.zeros(batch.total * 64, dtype=torch.int16, device=batch.data.device)
    rc = lib.sea_subband64_batch(_dptr(batch.data), _dptr(out), _dptr(batch.offsets), _dptr(batch.lengths),
                                 _dptr(batch.order) if use_order else None, batch.n_utt, _stream_ptr())
    _lib.check(rc, "sea_subband64_batch")
    return out


def irm_target_batch(batch, pure_sub, noise_sub, window=1):
    """IRM target of every utterance of the batch from two subband_batch() outputs (clean, noise).  Returns a
    MaskBatch (rows (L_u - 320) / 160 + 1 per utterance) that resynth_batch() takes as it is."""
    torch = _torch()
    lib = _lib.load()
    if np.any(np.asarray(batch.host_lengths) < 320):
        raise ValueError("irm_target needs utterances of at least one 320-sample frame")
    rows = (np.asarray(batch.host_lengths) - 320) // 160 + 1
    offs = np.concatenate(([0], np.cumsum(rows)[:-1])).astype(np.int64)
    dev = batch.data.device
    irm = torch.zeros((int(rows.sum()), 64), dtype=torch.float32, device=dev)
    d_offs = torch.from_numpy(offs).to(dev)
    _lib.check(lib.sea_irm_target_batch(_dptr(pure_sub), _dptr(noise_sub), _dptr(batch.offsets), _dptr(batch.lengths),
                                        _dptr(d_offs), _dptr(irm), int(window), batch.n_utt, _stream_ptr()),
               "sea_irm_target_batch")
    return MaskBatch(irm, d_offs, offs, rows)


def ns16k_streams_push(frames, state=None, reset=None):
    """The 16 k-native NoiseSup variant behind the reference's batch plug-in symbols (aurora_etsi/NoiseSup.cpp:1140-1407;
    SURVEY 8(f) #4): frames float32 [B, nframes, 160] on the GPU, func_Wiener's frame gate applied inside.  Returns
    dict(out [B, nframes, 160], produced int32 [B, nframes], flags uint8 [B, nframes] (bit 0 SpeechFoundVar, 1 Spec, 2 Mel,
    3 VADNS), counter int32 [B, nframes] (0: the first stage did not run), wiener float32 [B, nframes, 25] (rows where
    produced), state); pass ``state`` back in to continue the same streams."""
    torch = _torch()
    lib = _lib.load()
    frames = frames.contiguous()
    B, nfr, hop = frames.shape
    assert hop == 160 and frames.dtype == torch.float32
    if state is None:
        state = torch.zeros((B, lib.sea_ns16k_state_floats()), dtype=torch.float32, device=frames.device)
        reset = True if reset is None else reset
    out = torch.zeros_like(frames)
    produced = torch.zeros((B, nfr), dtype=torch.int32, device=frames.device)
    flags = torch.zeros((B, nfr), dtype=torch.uint8, device=frames.device)
    counter = torch.zeros((B, nfr), dtype=torch.int32, device=frames.device)
    wiener = torch.zeros((B, nfr, 25), dtype=torch.float32, device=frames.device)
    rc = lib.sea_ns16k_streams_push(_dptr(frames), _dptr(out), _dptr(produced), _dptr(flags), _dptr(counter), _dptr(wiener),
                                    _dptr(state), B, nfr, int(bool(reset)), _stream_ptr())
    _lib.check(rc, "sea_ns16k_streams_push")
    return dict(out=out, produced=produced, flags=flags, counter=counter, wiener=wiener, state=state)


def ns16k_tables():
    """Host-side tables of the 16 k-native variant (no GPU needed), laid out as the reference's init code builds them."""
    lib = _lib.load()
    sw, iw, gs = np.zeros(480, np.float32), np.zeros(17, np.float32), np.zeros(25, np.int32)
    g, d = np.zeros((25, 128), np.float32), np.zeros((25, 25), np.float32)
    _lib.check(lib.sea_ns16k_tables_host(*[a.ctypes.data for a in (sw, iw, gs, g, d)]), "sea_ns16k_tables_host")
    return dict(sigWindow=sw, irWindow=iw, gammaStart=gs, gamma=g, idct=d)


def ns_streams_push(frames, state=None, reset=None, want_flags=False):
    """Batched DoNoiseSup: frames float32 [B, nframes, 80] on the GPU.  Returns (out, produced, state);
    pass ``state`` back in to continue the same streams.  want_flags: also return (flags uint8
    [B, nframes] with bit 0 SpeechFoundVar, 1 Spec, 2 Mel, 3 VADNS, frame_counter int32 [B, nframes]),
    the per-frame outputs of the reference's batch plug-in shape (NoiseSupExports.h:19-27)."""
    torch = _torch()
    lib = _lib.load()
    frames = frames.contiguous()
    B, nfr, hop = frames.shape
    assert hop == 80
    if state is None:
        state = torch.zeros((B, lib.sea_ns_state_floats()), dtype=torch.float32, device=frames.device)
        reset = True if reset is None else reset
    out = torch.zeros_like(frames)
    produced = torch.zeros((B, nfr), dtype=torch.int32, device=frames.device)
    if want_flags:
        flags = torch.zeros((B, nfr), dtype=torch.uint8, device=frames.device)
        counter = torch.zeros((B, nfr), dtype=torch.int32, device=frames.device)
        rc = lib.sea_ns_streams_push_fd(_dptr(frames), _dptr(out), _dptr(produced), _dptr(flags), _dptr(counter),
                                        _dptr(state), B, nfr, int(bool(reset)), _stream_ptr())
        _lib.check(rc, "sea_ns_streams_push_fd")
        return out, produced, state, flags, counter
    rc = lib.sea_ns_streams_push(_dptr(frames), _dptr(out), _dptr(produced), _dptr(state), B, nfr,
                                 int(bool(reset)), _stream_ptr())
    _lib.check(rc, "sea_ns_streams_push")
    return out, produced, state
