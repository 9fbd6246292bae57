"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Run on an MI355X with ``pytest -m gpu``.

Stated tolerances:
  * int16 audio (NoiseSup, resynth): max |delta| <= 2 LSB and >= 99.9 % of samples exact; which
    samples are zero / untouched and the frame count are exact (frame indexing bit-exact).
  * rfft: bit-exact (same butterflies, same order, no FMA).
  * float NoiseSup stream / Wiener internals: |delta| <= 1e-4 * max(1, |ref|).
  * CompCeps: |delta| <= 1e-3 absolute per coefficient.
The kernels are written to be bit-identical: the two double-precision logarithms per frame are guarded
(a result within the error bound of a float rounding boundary is recomputed in double-double arithmetic,
ns_core.h; swept over every float argument by test_selftest_log_guard_sweep), so the stated tolerances are
the contract and the known-answer / full-size tests assert equality; the tests print how exact the match was.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

INT16_MAX_LSB = 2
INT16_EXACT_FRACTION = 0.999


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return torch


def _assert_int16_close(got, want, what):
    got = np.asarray(got).astype(np.int64)
    want = np.asarray(want).astype(np.int64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    if got.size == 0:
        return
    d = np.abs(got - want)
    exact = float(np.mean(d == 0))
    print(f"{what}: n={got.size} exact={exact * 100:.4f}% max|d|={int(d.max())}")
    assert d.max() <= INT16_MAX_LSB, f"{what}: max |delta| {int(d.max())} LSB at {int(d.argmax())}"
    assert exact >= INT16_EXACT_FRACTION, f"{what}: only {exact * 100:.3f}% exact"
    # frame indexing: the structurally silent whole frames (gate, 4-frame latency) are silent here too
    nz = np.nonzero(want)[0]
    lead = (int(nz[0]) if nz.size else want.size) // 80 * 80
    assert not np.any(got[:lead]), f"{what}: output before the reference's first output frame"


def _mixed_corpus():
    """Seeded utterances covering the edge cases: leading zeros (every 5th), ragged lengths (not a
    multiple of 80), shorter than one frame, empty, all-zero, loud (clipping-range) input."""
    from speech_enhancement_amd import corpus
    utts = corpus.synth_corpus(10, max_len=16000)
    utts[1] = utts[1][:8000 + 37]          # ragged tail
    utts[2] = utts[2][:79]                 # shorter than one frame
    utts[3] = np.zeros(0, np.int16)        # empty
    utts[4] = np.zeros(2400, np.int16)     # all zero: gate never opens
    z = corpus.synth_utterance(6, 6400).astype(np.int32) * 6
    utts[6] = np.clip(z, -32768, 32767).astype(np.int16)  # loud
    utts[7] = np.concatenate([np.zeros(1000, np.int16), utts[7][:5000], np.zeros(1600, np.int16), utts[7][5000:8000]])
    return utts


def test_library_reports_gfx950():
    import speech_enhancement_amd as sea
    _torch()
    lib = sea.load()
    assert lib.sea_init(-1) == 0, lib.sea_last_error()
    assert b"gfx950" in lib.sea_version()


def test_rfft_bit_exact(oracle):
    import speech_enhancement_amd as sea
    torch = _torch()
    rng = np.random.default_rng(1)
    frames = (rng.standard_normal((257, 256)) * rng.uniform(0.1, 3000.0, (257, 1))).astype(np.float32)
    frames[0] = 0.0
    frames[1, :] = 0.0
    frames[1, 3] = 1.0
    frames[2] = 32767.0
    want = np.stack([oracle.rfft(f) for f in frames])
    got = sea.rfft_batch(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
        f"rfft not bit-exact: {np.sum(got.view(np.uint32) != want.view(np.uint32))} words differ"
    # the drop-in with the reference's signature
    one = sea.rfft(frames[5])
    assert np.array_equal(one.view(np.uint32), want[5].view(np.uint32))


def test_etsi_denoise_known_answer(oracle):
    """first non-zero index 320, out[320..335], weighted checksum 91888."""
    import speech_enhancement_amd as sea
    from oracle import oracle as O
    _torch()
    x = O.kat_ns_signal()
    got = sea.etsi_denoise(x, fill=-7777)
    want = oracle.etsi_denoise(x, fill=-7777)
    _assert_int16_close(got, want, "etsi_denoise KAT")
    assert int(np.nonzero(got)[0][0]) == 320
    assert list(got[320:336]) == [6, 27, 59, 76, 86, 82, 56, 67, 28, 7, -31, -52, -59, -78, -60, -84]
    cs = O.weighted_checksum(got)
    print("checksum", cs)
    assert cs == 91888
    assert np.array_equal(got, want)


def test_etsi_denoise_tail_untouched(oracle):
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    x = corpus.synth_utterance(3, 4000 + 53)
    got = sea.etsi_denoise(x, fill=-7777)
    want = oracle.etsi_denoise(x, fill=-7777)
    assert np.all(got[4000:] == -7777), "trailing partial frame must not be written (SURVEY F7)"
    _assert_int16_close(got, want, "etsi_denoise ragged")


def test_ns_batch_vs_oracle(oracle):
    import speech_enhancement_amd as sea
    _torch()
    utts = _mixed_corpus()
    batch = sea.PackedBatch.from_arrays(utts)
    out, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
    outs = batch.split(out, full_frames_only=True)
    f32s = batch.split(f32, full_frames_only=True)
    first = first.cpu().numpy()
    n_exact = n_total = 0
    for u, x in enumerate(utts):
        tr = oracle.ns_trace(x, want_state=False)
        want = tr["out_i16"][: len(x) // 80 * 80]
        _assert_int16_close(outs[u], want, f"utt {u} (L={len(x)})")
        n_exact += int(np.sum(outs[u] == want))
        n_total += want.size
        # frame indexing: first output frame and number of outputs
        nfr = len(x) // 80
        exp_first = nfr - tr["nout"] if tr["nout"] else -1
        assert int(first[u]) == exp_first, f"utt {u}: first_out {first[u]} vs {exp_first}"
        if tr["nout"]:
            got_f = f32s[u][exp_first * 80:]
            ref_f = tr["den_f32"]
            tol = 1e-4 * np.maximum(1.0, np.abs(ref_f))
            assert np.all(np.abs(got_f - ref_f) <= tol), f"utt {u}: float stream off by {np.abs(got_f - ref_f).max()}"
            print(f"utt {u}: float stream bit-exact words {np.mean(got_f.view(np.uint32) == ref_f.view(np.uint32)) * 100:.4f}%")
    print(f"batch: {n_exact}/{n_total} int16 samples exact")


def test_ns_batch_order_invariant():
    import speech_enhancement_amd as sea
    torch = _torch()
    utts = _mixed_corpus()
    batch = sea.PackedBatch.from_arrays(utts)
    a, _, _ = sea.ns_denoise_batch(batch, use_order=True)
    b, _, _ = sea.ns_denoise_batch(batch, use_order=False)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_ns_four_per_cu_any_launch_order(oracle):
    """1000 short utterances = four per CU: the dense six-wave form with all four workgroups of a CU resident and the issue
    priority by remaining frames switched on.  The priority rule takes the batch's longest utterance from the FIRST entry of the
    launch order; a caller may pass any permutation (then the rule is only less sharp): with the library's order, with a shuffled
    one and with none the outputs are the same words, and a spread of utterances equals the oracle."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    rng = np.random.default_rng(7)
    utts = [corpus.synth_utterance(300 + k, 80 * int(n) + int(r)) for k, (n, r) in enumerate(zip(rng.integers(6, 60, 1000), rng.integers(0, 80, 1000)))]
    batch = sea.PackedBatch.from_arrays(utts)
    a, _, fa = sea.ns_denoise_batch(batch, use_order=True)
    keep = batch.order.clone()
    batch.order = keep[torch.randperm(len(utts), generator=torch.Generator().manual_seed(3)).to(keep.device)]
    b, _, fb = sea.ns_denoise_batch(batch, use_order=True)
    batch.order = keep
    c, _, fc = sea.ns_denoise_batch(batch, use_order=False)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(fa, fb) and torch.equal(fa, fc)
    got = batch.split(a, full_frames_only=True)
    for u in range(0, len(utts), 37):
        tr = oracle.ns_trace(utts[u], want_state=False)
        assert np.array_equal(got[u], tr["out_i16"][: (len(utts[u]) // 80) * 80]), f"utterance {u} (L={len(utts[u])})"


def test_compceps_vs_oracle(oracle):
    import speech_enhancement_amd as sea
    torch = _torch()
    utts = _mixed_corpus()
    batch = sea.PackedBatch.from_arrays(utts)
    out, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
    ceps, cum, n_ceps = sea.compceps_batch(batch, f32, first)
    torch.cuda.synchronize()
    ceps, n_ceps = ceps.cpu().numpy(), n_ceps.cpu().numpy()
    worst = 0.0
    for u, x in enumerate(utts):
        tr = oracle.ns_trace(x, want_state=False)
        assert int(n_ceps[u]) == tr["nceps"], f"utt {u}: {n_ceps[u]} cepstral frames vs {tr['nceps']}"
        if tr["nceps"]:
            got = ceps[cum[u]:cum[u] + tr["nceps"]]
            d = np.abs(got - tr["ceps"]).max()
            worst = max(worst, float(d))
            assert d <= 1e-3, f"utt {u}: cepstra off by {d}"
    print("CompCeps worst |delta| =", worst)
    # DoCompCeps-shaped single-frame call, on an arbitrary frame (not from NoiseSup)
    rng = np.random.default_rng(2)
    data = (rng.standard_normal(201) * 500).astype(np.float32)
    got = sea.DoCompCeps(data)
    want = oracle.compceps_frame(data)
    assert np.abs(got - want).max() <= 1e-3
    # silent frame hits both floors (e^-50 and e^-10)
    z = np.zeros(201, np.float32)
    assert np.abs(sea.DoCompCeps(z) - oracle.compceps_frame(z)).max() <= 1e-3


def test_compceps_frames_amplitudes_and_ragged_tiles(oracle):
    """sea_compceps_frames (the tiled kernel: 16 frames per wave, two per transform, the mel pass's lanes dealt to
    (frame, band) items) on arbitrary 201-float frames: every count 1..35 (partial tiles, odd pairs) and amplitudes from the
    floors (e^-50, e^-10: CompCeps.c:405-406) up to 1e14 (power spectrum ~1e33: still finite in float).  Bit-identical to
    the restatement frame by frame (tolerance of north_star: 1e-3 relative to the log domain; measured 0.0)."""
    import speech_enhancement_amd as sea
    torch = _torch()
    rng = np.random.default_rng(77)
    base = rng.standard_normal((35, 201)).astype(np.float32)
    for n in (1, 2, 3, 15, 16, 17, 31, 32, 33, 35):
        got = sea.compceps_frames(torch.from_numpy(base[:n] * np.float32(300.0)).cuda()).cpu().numpy()
        want = np.stack([oracle.compceps_frame(base[i] * np.float32(300.0)) for i in range(n)])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"{n} frames"
    for sc in (1e-30, 1e-12, 1e-6, 1e-3, 1.0, 1e4, 1e9, 1e14):
        data = base * np.float32(sc)
        data[5] = 0.0  # a silent frame inside the tile
        got = sea.compceps_frames(torch.from_numpy(data).cuda()).cpu().numpy()
        want = np.stack([oracle.compceps_frame(data[i]) for i in range(len(data))])
        assert np.isfinite(got).all(), f"scale {sc}"
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"scale {sc}: max |d| {np.abs(got - want).max()}"


def test_ns_stream_plugin_vs_oracle(oracle):
    """DoNoiseSup-shaped streaming (state in HBM between calls) == the batch path == the oracle."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    x = corpus.synth_utterance(1, 80 * 40)
    tr = oracle.ns_trace(x, want_state=False)
    ns = sea.NoiseSup()
    outs = []
    for f in range(40):
        ok, y = ns.DoNoiseSup(x[f * 80:(f + 1) * 80].astype(np.float32))
        assert ok == (f >= 4)
        if ok:
            outs.append(y)
    ns.close()
    got = np.concatenate(outs)
    tol = 1e-4 * np.maximum(1.0, np.abs(tr["den_f32"]))
    assert np.all(np.abs(got - tr["den_f32"]) <= tol)
    # batched streams in two chunks, state carried in HBM
    frames = torch.from_numpy(x.astype(np.float32).reshape(1, 40, 80)).cuda().repeat(3, 1, 1)
    o1, p1, st = sea.ns_streams_push(frames[:, :17].contiguous())
    o2, p2, st = sea.ns_streams_push(frames[:, 17:].contiguous(), state=st, reset=False)
    o = torch.cat([o1, o2], 1).cpu().numpy()
    p = torch.cat([p1, p2], 1).cpu().numpy()
    assert np.array_equal(p[0], (np.arange(40) >= 4).astype(np.int32))
    for b in range(3):
        assert np.array_equal(o[b, 4:].reshape(-1).view(np.uint32), got.view(np.uint32))


@pytest.mark.parametrize("binary", [False, True])
def test_resynth_vs_oracle(oracle, binary):
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    lens = [4800, 3200 + 77, 320, 479, 8000]
    utts = [corpus.synth_utterance(20 + i, L) for i, L in enumerate(lens)]
    masks = [corpus.synth_mask(20 + i, L) for i, L in enumerate(lens)]
    masks[0][3:6, :] = 0.0          # frames the soft path must skip
    masks[0][10, ::2] = 0.5         # exactly at the IBM threshold: skipped by '> 0.5'
    batch = sea.PackedBatch.from_arrays(utts)
    mb = sea.MaskBatch.from_arrays(masks)
    out, _ = sea.resynth_batch(batch, mb, binary=binary)
    torch.cuda.synchronize()
    outs = batch.split(out)
    for u, (x, m) in enumerate(zip(utts, masks)):
        want = oracle.resynth64(x, m, binary=binary)
        _assert_int16_close(outs[u], want, f"resynth{'_IBM' if binary else ''} utt {u} (L={len(x)})")


def test_resynth_known_answer(oracle):
    """SURVEY 8(c) reference output: out[8000..8009] and checksum -2456454 (soft mask)."""
    import speech_enhancement_amd as sea
    from oracle import oracle as O
    _torch()
    x, m = O.kat_resynth_case()
    got = sea.resynth(x, m, binary=False)
    want = oracle.resynth64(x, m)
    _assert_int16_close(got, want, "resynth KAT")
    print("checksum", O.weighted_checksum(got), list(got[8000:8010]))
    assert np.array_equal(got, want)
    assert O.weighted_checksum(got) == -2456454
    assert list(got[8000:8010]) == [88, 528, 249, 1331, 1271, 1638, 2190, 1682, 1642, 2182]


def test_resynth_utterances_chunked_by_scratch_budget(oracle, monkeypatch):
    """sea_resynth_utterances (the host-buffer entry the file driver uses) cuts the list into sub-batches whose
    256-B-per-sample HBM intermediate fits a budget (60 % of the free HBM; SEA_RESYNTH_SCRATCH_MB overrides it) and
    reuses ONE scratch allocation for all of them.  A 3 MB budget forces seven utterances of 1.2-4.1 MB each into
    several sub-batches, the largest alone; every utterance must equal the oracle whatever the cut."""
    import ctypes
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    lib = sea.load()
    lens = [4800, 16000, 8000, 4800, 12000, 6400, 16000]
    utts = [corpus.synth_utterance(200 + i, L) for i, L in enumerate(lens)]
    masks = [corpus.synth_mask(200 + i, L) for i, L in enumerate(lens)]
    want = [oracle.resynth64(x, m) for x, m in zip(utts, masks)]

    def run():
        outs = [np.zeros_like(x) for x in utts]
        n = len(utts)
        pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in utts])
        pm = (ctypes.c_void_p * n)(*[m.ctypes.data for m in masks])
        po = (ctypes.c_void_p * n)(*[y.ctypes.data for y in outs])
        pl = (ctypes.c_long * n)(*lens)
        assert lib.sea_resynth_utterances(pin, pl, pm, 0, po, n) == 0, lib.sea_last_error()
        return outs
    monkeypatch.setenv("SEA_RESYNTH_SCRATCH_MB", "3")
    for got, w in zip(run(), want):
        assert np.array_equal(got, w)
    monkeypatch.delenv("SEA_RESYNTH_SCRATCH_MB")
    for got, w in zip(run(), want):
        assert np.array_equal(got, w)


def test_denoise_utterances_pipeline_any_chunking(oracle, monkeypatch):
    """sea_denoise_utterances (the host-buffer entry the file driver uses, etsi/cpp/main.cpp:43-67 for a list) is a
    copy / compute pipeline (csrc/hostpipe.hip): by default over TIME SLICES of the whole list (one launch per slice,
    the recursion carried in a state blob per utterance), with SEA_HOST_MODE=chunks over chunks of whole utterances.
    Whatever the number of slices / the chunk size, every utterance must equal the oracle and the samples etsi_denoise
    never writes (the trailing partial frame) must keep the caller's fill."""
    import ctypes
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    lib = sea.load()
    assert lib.sea_host_threads() >= 1
    rng = np.random.default_rng(11)
    lens = [int(v) for v in rng.integers(0, 40000, 90)] + [0, 79, 80, 81, 24000, 24000, 24037]
    utts = [corpus.synth_utterance(300 + i, L) for i, L in enumerate(lens)]
    utts[5] = np.zeros(lens[5], np.int16)  # gate never opens
    want = [oracle.etsi_denoise(x) for x in utts]

    def run():
        outs = [np.full(x.shape, 77, np.int16) for x in utts]
        n = len(utts)
        pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in utts])
        po = (ctypes.c_void_p * n)(*[y.ctypes.data for y in outs])
        pl = (ctypes.c_long * n)(*lens)
        assert lib.sea_denoise_utterances(pin, po, pl, n) == 0, lib.sea_last_error()
        return outs

    def check(what):
        for u, (got, w, L) in enumerate(zip(run(), want, lens)):
            full = L // 80 * 80
            assert np.array_equal(got[:full], w[:full]), f"{what}: utterance {u} (L={L}) differs"
            assert np.all(got[full:] == 77), f"{what}: utterance {u}: the trailing partial frame was written"

    for k in (None, "1", "2", "7", "40"):                 # default (4), one launch, and cuts the lengths do not align with
        if k is None:
            monkeypatch.delenv("SEA_HOST_SLICES", raising=False)
        else:
            monkeypatch.setenv("SEA_HOST_SLICES", k)
        check(f"{k or 'default'} time slices")
    monkeypatch.delenv("SEA_HOST_SLICES", raising=False)
    monkeypatch.setenv("SEA_HOST_MODE", "chunks")
    for mb in ("1", "3", None):
        if mb is None:
            monkeypatch.delenv("SEA_HOST_CHUNK_MB")
        else:
            monkeypatch.setenv("SEA_HOST_CHUNK_MB", mb)
        check(f"chunks of {mb} MB")


def test_packed_pinned_entry_points(oracle, monkeypatch):
    """sea_packed_*: the caller writes its samples into the pinned staging the library laid out (pieces in time order, one
    per time slice the utterance reaches) and reads the results from the matching output pieces -- no pack / unpack copies
    (what host/etsi_denoise_main.c's reader and writer threads do).  Every whole frame must equal the oracle's etsi_denoise,
    for one slice and for several, with empty / sub-frame / all-zero utterances in the list, and a set must be reusable."""
    import ctypes
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    lib = sea.load()
    rng = np.random.default_rng(23)
    lens = [int(v) for v in rng.integers(0, 48000, 60)] + [0, 79, 80, 161, 32000, 32000, 31999]
    utts = [corpus.synth_utterance(500 + i, L) for i, L in enumerate(lens)]
    utts[7] = np.zeros(lens[7], np.int16)
    want = [oracle.etsi_denoise(x) for x in utts]
    n = len(utts)
    p = lib.sea_packed_create()
    assert p
    try:
        for slices in ("1", "3", None, "40"):
            if slices is None:
                monkeypatch.delenv("SEA_HOST_SLICES", raising=False)
            else:
                monkeypatch.setenv("SEA_HOST_SLICES", slices)
            for rep in range(2):                                           # the same set planned and run twice
                pl = (ctypes.c_long * n)(*lens)
                assert lib.sea_packed_plan(p, pl, n) == 0, lib.sea_last_error()
                K = lib.sea_packed_slices(p)
                assert K >= 1
                segs = []
                for u, x in enumerate(utts):
                    pin, pout, cnt = (ctypes.c_void_p * K)(), (ctypes.c_void_p * K)(), (ctypes.c_long * K)()
                    k = lib.sea_packed_segments(p, u, pin, pout, cnt, K)
                    assert sum(cnt[i] for i in range(k)) == lens[u] // 80 * 80, f"utterance {u}: pieces do not cover its whole frames"
                    pos = 0
                    for i in range(k):
                        dst = np.ctypeslib.as_array(ctypes.cast(pin[i], ctypes.POINTER(ctypes.c_short)), shape=(cnt[i],))
                        dst[:] = x[pos:pos + cnt[i]]
                        pos += cnt[i]
                    segs.append((k, pout, cnt))
                assert lib.sea_packed_denoise(p) == 0, lib.sea_last_error()
                for u, (k, pout, cnt) in enumerate(segs):
                    got = np.concatenate([np.ctypeslib.as_array(ctypes.cast(pout[i], ctypes.POINTER(ctypes.c_short)), shape=(cnt[i],))
                                          for i in range(k)] or [np.zeros(0, np.int16)])
                    full = lens[u] // 80 * 80
                    assert np.array_equal(got, want[u][:full]), f"{slices or 'default'} slices, run {rep}: utterance {u} (L={lens[u]}) differs"
    finally:
        lib.sea_packed_destroy(p)


def test_host_pipeline_returns_fault_on_event_error(oracle, monkeypatch):
    """The three event-driven pipelines of csrc/hostpipe.hip poll hipEventQuery; any answer other than "done" / "not
    ready" must end the call with the reference's fault code 1 (etsi/cpp/AdvFrontEnd.c:205-209) -- not be polled for
    ever -- with the pool idle and the streams drained, and the next call must work.  sea_selftest_hostpipe_fault makes
    the nth query of the process report a device fault."""
    import ctypes
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    lib = sea.load()
    lens = [24000 + 160 * i for i in range(48)]            # 2.3 MB of int16: the pipeline, not the inline path
    utts = [corpus.synth_utterance(900 + i, L) for i, L in enumerate(lens)]
    masks = [corpus.synth_mask(900 + i, L) for i, L in enumerate(lens)]
    n = len(utts)
    pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in utts])
    pl = (ctypes.c_long * n)(*lens)
    pm = (ctypes.c_void_p * n)(*[m.ctypes.data for m in masks])

    def denoise():
        outs = [np.zeros_like(x) for x in utts]
        po = (ctypes.c_void_p * n)(*[y.ctypes.data for y in outs])
        return lib.sea_denoise_utterances(pin, po, pl, n), outs

    def resynth():
        outs = [np.zeros_like(x) for x in utts]
        po = (ctypes.c_void_p * n)(*[y.ctypes.data for y in outs])
        return lib.sea_resynth_utterances(pin, pl, pm, 0, po, n), outs

    rc, good = denoise()
    assert rc == 0, lib.sea_last_error()
    rc, good_rs = resynth()
    assert rc == 0, lib.sea_last_error()
    try:
        for mode in (None, "chunks"):
            if mode:
                monkeypatch.setenv("SEA_HOST_MODE", mode)
            for nth in (1, 3, 9):
                lib.sea_selftest_hostpipe_fault(nth)
                rc, _ = denoise()
                assert rc == 1, f"{mode or 'slices'}: fault at query {nth} was not reported"
                assert b"event" in lib.sea_last_error()
                lib.sea_selftest_hostpipe_fault(0)
                rc, outs = denoise()
                assert rc == 0, lib.sea_last_error()
                assert all(np.array_equal(a, b) for a, b in zip(outs, good)), "the call after a fault differs"
        monkeypatch.delenv("SEA_HOST_MODE", raising=False)
        lib.sea_selftest_hostpipe_fault(2)
        rc, _ = resynth()
        assert rc == 1
        lib.sea_selftest_hostpipe_fault(0)
        rc, outs = resynth()
        assert rc == 0, lib.sea_last_error()
        assert all(np.array_equal(a, b) for a, b in zip(outs, good_rs))
    finally:
        lib.sea_selftest_hostpipe_fault(0)


def test_gammatone_filter_vs_oracle(oracle):
    import speech_enhancement_amd as sea
    _torch()
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(2000) * 1000).astype(np.float32)
    cf, bw, me = oracle.resynth_channels()
    for chan in (0, 17, 63):
        got = sea.gammaToneFilter(x, chan)
        want = oracle.gammatone(x, cf[chan], bw[chan], me[chan])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"channel {chan}"
    # a burst followed by silence: the filter state decays through the denormal range, which the
    # packed-pair arithmetic of the kernels must follow bit for bit (no flush to zero)
    y = np.zeros(24000, np.float32)
    y[:400] = x[:400]
    for chan in (0, 40):
        want = oracle.gammatone(y, cf[chan], bw[chan], me[chan])
        tiny = np.abs(want[want != 0])
        assert tiny.min() < 1.17e-38, "case must reach denormals to mean anything"
        got = sea.gammaToneFilter(y, chan)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"denormal tail, channel {chan}"


def test_selftest_pi4_identity_all_floats():
    """The FFT's pi/4 butterfly multiplies by 1/sqrt2 in double instead of dividing by sqrt2: the
    two give the same float for every one of the 2^32 float inputs (checked exhaustively here)."""
    import ctypes
    import speech_enhancement_amd as sea
    _torch()
    n = ctypes.c_ulonglong(123)
    assert sea.load().sea_selftest_pi4(ctypes.byref(n)) == 0
    assert n.value == 0, f"{n.value} floats differ"


def test_selftest_dc_filter_fast_and_exact_paths():
    """DC-offset recurrence: the float-FMA fast path must equal the reference's double arithmetic,
    and frames whose operands could break that equivalence must take the exact path."""
    import speech_enhancement_amd as sea
    _torch()
    rng = np.random.default_rng(7)
    n = 64
    dif = (rng.standard_normal((n, 80)) * rng.uniform(1e-3, 3e3, (n, 1))).astype(np.float32)
    y0 = (rng.standard_normal(n) * 100).astype(np.float32)
    dif[1] = 0.0                                     # all-zero differences (exact by construction)
    dif[2, 40] = np.float32(1e-30)                   # tiny d against a large y: must fall back
    y0[2] = np.float32(5e4)
    dif[3, 0] = np.float32(3e9)                      # huge d against tiny y: must fall back
    y0[3] = np.float32(1e-3)
    y0[4] = 0.0
    out = np.zeros_like(dif)
    fb = np.zeros(n, np.int32)
    lib = sea.load()
    rc = lib.sea_selftest_dc(dif.ctypes.data, y0.ctypes.data, out.ctypes.data, fb.ctypes.data, n)
    assert rc == 0
    want = np.zeros_like(dif)
    for c in range(n):
        y = np.float32(y0[c])
        for i in range(80):
            y = np.float32(np.float64(dif[c, i]) + np.float64(0.9990234375) * np.float64(y))
            want[c, i] = y
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert fb[2] == 1 and fb[3] == 1 and fb[1] == 0
    print("dc fallbacks:", int(fb.sum()), "of", n)


def test_ns_pipeline_fill_and_drain(oracle):
    """Utterances of 0..12 frames (plus ragged tails): the 5-deep wave pipeline must fill and drain
    exactly like the serial reference (outputs start at frame 4, nothing before)."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    utts = [corpus.synth_utterance(50 + n, 80 * n + (n * 13) % 80) for n in range(13)]
    utts += [np.concatenate([np.zeros(80 * k, np.int16), corpus.synth_utterance(70 + k, 80 * 7)]) for k in (1, 2, 5)]
    batch = sea.PackedBatch.from_arrays(utts)
    out, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
    outs = batch.split(out, full_frames_only=True)
    first = first.cpu().numpy()
    for u, x in enumerate(utts):
        tr = oracle.ns_trace(x, want_state=False)
        want = tr["out_i16"][: len(x) // 80 * 80]
        assert np.array_equal(outs[u], want), f"utt {u} (L={len(x)})"
        assert int(first[u]) == (len(x) // 80 - tr["nout"] if tr["nout"] else -1)


def test_selftest_log_accuracy():
    """The kernels' own double log (two scalar sites per frame) against 40-digit references: within
    2 ulp, and the two float results the hot path derives from it equal the ones libm's log gives on
    every sampled argument."""
    import math
    from decimal import Decimal, getcontext
    import speech_enhancement_amd as sea
    _torch()
    getcontext().prec = 40
    rng = np.random.default_rng(11)
    x = np.concatenate([
        np.float32(64.0) + rng.uniform(0, 1, 4000).astype(np.float32) * np.float32(2.0) ** rng.integers(0, 36, 4000),
        (10.0 ** rng.uniform(-5, 12, 4000)).astype(np.float32),
        np.array([64.0, 1.0, 2.0, 0.5, 1.4142135, 1.4142137, 1e-5, 3.4e38, 1.1754944e-38], np.float32)])
    x = x.astype(np.float32)
    out = np.zeros(x.size, np.float64)
    assert sea.load().sea_selftest_log(x.ctypes.data, out.ctypes.data, x.size) == 0
    worst = 0.0
    for xi, yi in zip(x[::7], out[::7]):
        true = Decimal(float(xi)).ln()
        ulp = Decimal(math.ulp(float(true))) if true != 0 else Decimal(5e-324)
        worst = max(worst, float(abs(Decimal(float(yi)) - true) / ulp))
    print("ns_ln worst error (ulp):", worst)
    assert worst <= 2.0
    # the two complete call sites (fast log + guard + slow path) against this host's libm, exactly
    s1, s2 = np.zeros(x.size, np.float32), np.zeros(x.size, np.float32)
    assert sea.load().sea_selftest_log_sites(x.ctypes.data, s1.ctypes.data, s2.ctypes.data, x.size) == 0
    big = x >= 64.0
    ref1 = np.array([np.float32(0.5 + (math.log(float(v) / 64.0) / math.log(2.0)) * 16.0) for v in x[big]])
    print("VAD-energy site float mismatches:", int(np.sum(s1[big] != ref1)), "of", ref1.size)
    assert np.array_equal(s1[big], ref1) and np.all(np.isnan(s1[~big]))
    small = x.astype(np.float64) > 1e-5
    ref2 = np.array([np.float32((20 * math.log10(float(v))) / 3.0) for v in x[small]])
    print("log10 site float mismatches:", int(np.sum(s2[small] != ref2)), "of", ref2.size)
    assert np.array_equal(s2[small], ref2) and np.all(np.isnan(s2[~small]))


def _cr_ln(v):
    """ln(v) correctly rounded to double, from a 60-digit evaluation."""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    return float(Decimal(v).ln())


def _log10_fdlibm(x, ln):
    """glibc's log10 (sysdeps/ieee754/dbl-64/e_log10.c, the fdlibm formula) over a given natural log."""
    import struct
    bits = struct.unpack("<q", struct.pack("<d", x))[0]
    k = ((bits >> 52) & 0x7ff) - 1023
    i = 1 if k < 0 else 0
    xr = struct.unpack("<d", struct.pack("<q", (bits & 0x000fffffffffffff) | ((0x3ff - i) << 52)))[0]
    y = float(k + i)
    z = y * 3.69423907715893078616e-13 + 4.34294481903251816668e-01 * ln(xr)
    return z + y * 3.01029995663611771306e-01


def test_selftest_log_double_double():
    """The slow path's logarithm: hi + lo within 2^-78 (relative) of a 60-digit reference, hi correctly rounded."""
    from decimal import Decimal, getcontext
    import speech_enhancement_amd as sea
    _torch()
    getcontext().prec = 60
    rng = np.random.default_rng(5)
    x = np.concatenate([np.float32(10.0 ** rng.uniform(-5, 38, 3000)).astype(np.float64),
                        1.0 + rng.uniform(-0.3, 0.42, 2000), rng.uniform(0.5, 2.0, 2000),
                        np.array([1.0, 2.0, 0.5, 1.4142135623730951, 1.4142135623730954, 0.7071067811865476, 64.0, 1e-5])])
    hi, lo = np.zeros(x.size), np.zeros(x.size)
    assert sea.load().sea_selftest_log_dd(x.ctypes.data, hi.ctypes.data, lo.ctypes.data, x.size) == 0
    worst, wrong = 0.0, 0
    for xi, h, l in zip(x, hi, lo):
        true = Decimal(float(xi)).ln()
        if true == 0:
            assert h == 0.0 and l == 0.0
            continue
        worst = max(worst, float(abs((Decimal(float(h)) + Decimal(float(l)) - true) / true)))
        wrong += int(h + l != float(true))
    print(f"ns_ln_dd worst relative error 2^{np.log2(max(worst, 1e-40)):.1f}; not correctly rounded: {wrong} of {x.size}")
    assert worst < 2.0 ** -78 and wrong == 0


@pytest.mark.parametrize("site", [1, 2])
def test_selftest_log_guard_sweep(site):
    """EVERY float argument a log site can see goes through the site on the device (site 1: every finite float
    frameSum >= 64, 1.0e9 floats -- the int16 path stays below 2^37, the float streaming API does not; site 2: every
    float above 1e-5, 1.2e9 floats).  The guard fires on a few dozen of
    them; for each recorded hit the float the device returned must be the one a correctly rounded log gives
    (site 2: inside glibc's own log10 formula), and the float of the unguarded fast log is compared with this
    host's libm to show what the guard is for."""
    import math
    import speech_enhancement_amd as sea
    _torch()
    cap = 4096
    stats = np.zeros(8, np.uint64)
    hits = np.zeros((cap, 3), np.float32)
    assert sea.load().sea_selftest_log_guard(site, stats.ctypes.data, hits.ctypes.data, cap) == 0
    n, nhit, nflip, nrec, nmiss = (int(v) for v in stats[:5])
    print(f"site {site}: {n} arguments, {nhit} guard hits ({nhit / n:.2e} per call), slow path changed {nflip} floats, "
          f"fast != slow outside the guard window: {nmiss}")
    assert n == (0x7F7FFFFF - 0x42800000 + 1 if site == 1 else 0x7F7FFFFF - 0x3727C5AD + 1)
    assert nmiss == 0, "the guard window does not cover the fast form's error"
    window = 21 if site == 1 else 41
    assert 0 < nhit < 4 * window * n / 2 ** 29 + 20 and nrec == nhit and nhit <= cap
    bad_cr = libm_differs_fast = libm_differs_final = 0
    for x, fast, got in hits[:nhit]:
        x = float(x)
        if site == 1:
            want = np.float32(0.5 + (_cr_ln(x / 64.0) / math.log(2.0)) * 16.0)
            libm = np.float32(0.5 + (math.log(x / 64.0) / math.log(2.0)) * 16.0)
        else:
            want = np.float32((20 * _log10_fdlibm(x, _cr_ln)) / 3.0)
            libm = np.float32((20 * math.log10(x)) / 3.0)
        bad_cr += int(got != want)
        libm_differs_fast += int(fast != libm)
        libm_differs_final += int(got != libm)
    print(f"site {site}: hits where the unguarded float differs from this host's libm: {libm_differs_fast}; "
          f"after the guard: {libm_differs_final}")
    assert bad_cr == 0
    assert libm_differs_final <= max(1, nhit // 10)  # libm's log is correctly rounded in all but a few % of cases


def test_subband_vs_oracle(oracle):
    """subbband(): gammatone + Meddis hair cell + (short) cast -> 64 int16 streams.
    The oracle for this function is parity-unpinned (no reference build, no recorded output); the
    GPU must equal it bit for bit."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    lens = [4800, 1600 + 77, 1, 17, 16, 8000]
    utts = [corpus.synth_utterance(80 + i, L) for i, L in enumerate(lens)]
    batch = sea.PackedBatch.from_arrays(utts)
    out = sea.subband_batch(batch)
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for u, x in enumerate(utts):
        L = len(x)
        pitch = (L + 7) // 8 * 8
        blk = host[batch.host_offsets[u] * 64: batch.host_offsets[u] * 64 + 64 * pitch].reshape(64, pitch)
        want = oracle.subband64(x)
        assert np.array_equal(blk[:, :L], want), f"utt {u} (L={L})"
    one = sea.subbband(utts[0])
    assert np.array_equal(one, oracle.subband64(utts[0]))
    assert one.min() >= 0 and one.max() > 100      # hair-cell output is a non-negative firing rate


def test_irm_target_vs_oracle(oracle):
    """the IRM target of make_single_IBM on the subband streams of a clean and a noise signal
    (both produced by subbband() on the GPU).  Parity unpinned (asdk::SpecInfo is absent); the GPU's float
    transform against the oracle's double-precision DFT of the same definition: |delta| <= 1e-4, values in [0, 1],
    identical streams -> 0.5 exactly, host and batch entry points agree, the result drives resynth()."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    lens = [4800, 320, 320 + 160 * 3 + 11, 160 * 150 + 5]   # 149 frames: ten 16-frame tiles over the four waves of a workgroup
    clean = [corpus.synth_utterance(120 + i, L) for i, L in enumerate(lens)]
    noise = [(corpus.synth_utterance(140 + i, L).astype(np.int32) // 3).astype(np.int16) for i, L in enumerate(lens)]
    cb, nb = sea.PackedBatch.from_arrays(clean), sea.PackedBatch.from_arrays(noise)
    cs, ns = sea.subband_batch(cb), sea.subband_batch(nb)
    worst = 0.0
    for window in (1, 0, 2):
        mb = sea.irm_target_batch(cb, cs, ns, window)
        torch.cuda.synchronize()
        got_all = mb.data.cpu().numpy()
        for u, L in enumerate(lens):
            want = oracle.irm_target(oracle.subband64(clean[u]), oracle.subband64(noise[u]), window)
            got = got_all[mb.host_row_offsets[u]: mb.host_row_offsets[u] + mb.host_rows[u]]
            assert got.shape == want.shape == ((L - 320) // 160 + 1, 64)
            ok = ~np.isnan(want)
            assert np.array_equal(np.isnan(got), ~ok)
            worst = max(worst, float(np.abs(got[ok] - want[ok]).max()))
            assert got[ok].min() >= 0.0 and got[ok].max() <= 1.0
    print("IRM target max |delta| vs the oracle:", worst)
    assert worst <= 1e-4
    one = sea.irm_target(oracle.subband64(clean[0]), oracle.subband64(noise[0]))
    assert np.abs(one - oracle.irm_target(oracle.subband64(clean[0]), oracle.subband64(noise[0]))).max() <= 1e-4
    same = sea.irm_target(oracle.subband64(clean[0]), oracle.subband64(clean[0]))
    assert np.all(same[~np.isnan(same)] == np.float32(0.5))
    # the mask drives the resynthesis of the noisy mixture
    mix = (clean[0].astype(np.int32) + noise[0]).clip(-32768, 32767).astype(np.int16)
    y = sea.resynth(mix, one)
    assert np.array_equal(y, oracle.resynth64(mix, one))


def _float_stream_case(seed, nfr=60):
    rng = np.random.default_rng(seed)
    t = np.arange(nfr * 80)
    x = rng.standard_normal(nfr * 80) * 800 + 2000 * np.sin(t * 0.1) * ((t // 1600) % 2)
    return x.astype(np.float32).reshape(nfr, 80)


def test_ns_stream_float_amplitude_extremes(oracle):
    """DoNoiseSup takes floats of any magnitude.  The back half's guarded division (ns_div) only runs while
    the PSD is 0 or within [2^-40, 2^48] and the noise estimate within [2^-15, 2^28]; everything else takes
    the plain-division path.  Streams scaled from 1e-21 (PSD in the denormals) to 1e18, streams sitting on
    the domain's edges, and streams that jump between scales mid-way (the one-frame hand-over between the
    two paths) must all be bit-identical to the oracle."""
    import speech_enhancement_amd as sea
    torch = _torch()
    base = _float_stream_case(11)
    cases = [base * np.float32(sc) for sc in (1.0, 1e-3, 3e-6, 1e-7, 3e-9, 1e-12, 1e-19, 1e-21, 30.0, 1e3, 1e5, 1e9, 1e18)]
    mixed = _float_stream_case(12, 90)
    mixed[20:35] *= np.float32(1e-9)
    mixed[35:50] *= np.float32(1e7)
    mixed[50:52] = 0.0
    mixed[60:75] *= np.float32(1e-20)
    cases.append(mixed)
    cases.append(np.concatenate([base * np.float32(1e6), base, base * np.float32(2e-8)]))
    for k, x in enumerate(cases):
        want, wprod = oracle.ns_stream_f32(x)
        o, p, _ = sea.ns_streams_push(torch.from_numpy(x[None]).cuda().contiguous())
        p = p[0].cpu().numpy()
        assert np.array_equal(p, wprod), f"case {k}: produced flags"
        got = o[0].cpu().numpy()[p != 0].reshape(-1)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
            f"case {k}: {np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))} of {got.size} floats differ"


def test_selftest_nsdiv_guarded_division():
    """The NoiseSup back half divides with the compiler's own IEEE sequence minus v_div_scale / v_div_fixup
    (identities inside the per-frame guarded operand domain) and one reciprocal per denominator: bit-equal
    to plain division on 2^30 random + edge-mantissa pairs spanning that domain; same for the double
    reciprocal in the second-stage noise update."""
    import ctypes
    import speech_enhancement_amd as sea
    _torch()
    out = (ctypes.c_ulonglong * 4)(0, 99, 99, 99)
    assert sea.load().sea_selftest_nsdiv(out) == 0
    assert out[0] == 4096 * 256 * 1024
    assert out[1] == 0 and out[2] == 0, f"{out[1]} float / {out[2]} double quotients differ of {out[0]}"
    assert out[3] == 0, f"{out[3]} square roots differ (every float in [2^-96, 2^126] and 0 is compared)"


def test_selftest_div_by_middle_ear_gain():
    """The resynthesis kernels divide by the per-channel middle-ear gain with q = a*y, r = a - q*d,
    q' = q + r*y (y = 1/d): exhaustively equal to the IEEE quotient for every float inside the
    domain the kernels use it in (2^-100 <= |a| <= 2^100), for all 64 divisors."""
    import ctypes
    import speech_enhancement_amd as sea
    _torch()
    out = (ctypes.c_ulonglong * 2)()
    assert sea.load().sea_selftest_div(out) == 0
    assert out[1] == 2 * (200 * (1 << 23) + 1)           # both signs: exponents -100..99 in full, plus |a| = 2^100
    assert out[0] == 0, f"{out[0]} mismatching quotients"


def test_afe_feature_chain_vs_oracle(oracle):
    """NoiseSup -> WaveProc -> CompCeps -> PostProc -> VAD (+ flush), the chain the
    reference author commented out (ParmInterface.c:274-311).  The oracle for it is pinned bit-exact
    against the reference's own functions (tests/test_oracle.py); the GPU must reproduce the per-frame
    speech flags and VAD decisions exactly and the features within the CompCeps tolerance (1e-3)."""
    import speech_enhancement_amd as sea
    _torch()
    utts = _mixed_corpus()
    utts.append(np.zeros(480, np.int16))
    from speech_enhancement_amd import corpus
    utts.append(corpus.synth_utterance(31, 32000))
    batch = sea.PackedBatch.from_arrays(utts)
    res = sea.afe_features_batch(batch, want_intermediates=True)
    flags = res["flags"].cpu().numpy()
    fcc, fpp = res["feat_cc"].cpu().numpy(), res["feat_pp"].cpu().numpy()
    n_ceps, first = res["n_ceps"].cpu().numpy(), res["first_out"].cpu().numpy()
    worst = 0.0
    for u, x in enumerate(utts):
        tr = oracle.afe_trace(x)
        assert int(n_ceps[u]) == tr["nceps"], f"utt {u}"
        # speech flags per NoiseSup output frame
        nfr = len(x) // 80
        if tr["nout"]:
            f0 = int(first[u])
            got = flags[batch.host_offsets[u] // 8 + 10 * np.arange(f0, nfr)]
            want = tr["flags"][f0:nfr, :4] @ np.array([1, 2, 4, 8])
            assert np.array_equal(got, want), f"utt {u}: speech flags differ at {np.nonzero(got != want)[0][:5]}"
        c0 = res["ceps_cum"][u]
        if tr["nceps"]:
            for name, g, w in (("feat_cc", fcc, tr["feat_cc"]), ("feat_pp", fpp, tr["feat_pp"])):
                d = float(np.abs(g[c0:c0 + tr["nceps"]] - w).max())
                worst = max(worst, d)
                assert d <= 1e-3, f"utt {u} {name}: off by {d}"
        got15 = res["feats"][u]
        assert got15.shape == tr["vad_out"].shape, f"utt {u}: {got15.shape} vs {tr['vad_out'].shape}"
        if len(got15):
            assert np.array_equal(got15[:, 14], tr["vad_out"][:, 14]), f"utt {u}: VAD flags differ"
            d = float(np.abs(got15[:, :14] - tr["vad_out"][:, :14]).max())
            worst = max(worst, d)
            assert d <= 1e-3, f"utt {u} emitted features: off by {d}"
    print("AFE chain worst |delta| =", worst)
    # the audio is the same as the plain NoiseSup kernel's
    plain, _, _ = sea.ns_denoise_batch(batch)
    assert _torch().equal(plain, res["out"])


def test_etsi_denoise_from_concurrent_host_threads(oracle):
    """The reference's batch tool calls etsi_denoise() from N host threads
    (function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:111-121):
    the drop-in keeps a workspace and a stream per calling thread, results stay exact."""
    import threading
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    utts = [corpus.synth_utterance(60 + k, 2400 + 800 * (k % 5)) for k in range(16)]
    want = [oracle.etsi_denoise(x, fill=-7777) for x in utts]
    got = [None] * len(utts)
    errs = []

    def work(tid):
        try:
            for rep in range(3):
                for k in range(tid, len(utts), 4):
                    got[k] = sea.etsi_denoise(utts[k], fill=-7777)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for k in range(len(utts)):
        assert np.array_equal(got[k], want[k]), f"utterance {k}"


def test_frame_counter_int16_wrap(oracle):
    """An utterance longer than 32767 frames: the reference's int16 narrowing of the frame counter in
    FilterCalc is reproduced (the oracle is checked against the reference on the same
    input in tests/test_oracle.py)."""
    import speech_enhancement_amd as sea
    from tests.test_oracle import _long_utterance
    _torch()
    x = _long_utterance()
    got = sea.etsi_denoise(x)
    want = oracle.etsi_denoise(x)
    _assert_int16_close(got, want, "33100-frame utterance")
    assert np.array_equal(got, want)


@pytest.mark.parametrize("binary", [False, True])
def test_resynth_l_over_160_frame_count(oracle, binary):
    """Mode bit 1 of sea_resynth64_batch: lengths/160 mask rows per utterance (1dnn_resynth/extractwav.cpp:67)."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    lens = [4800, 3200 + 77, 200, 160, 8000]
    utts = [corpus.synth_utterance(70 + i, L) for i, L in enumerate(lens)]
    rng = np.random.default_rng(12)
    masks = [rng.random((L // 160, 64)).astype(np.float32) for L in lens]
    batch = sea.PackedBatch.from_arrays(utts)
    mb = sea.MaskBatch.from_arrays(masks)
    out, _ = sea.resynth_batch(batch, mb, binary=binary, frames_l_over_160=True)
    torch.cuda.synchronize()
    for u, (x, m, y) in enumerate(zip(utts, masks, batch.split(out))):
        want = oracle.resynth64(x, m, binary=binary, frames_l_over_160=True)
        assert np.array_equal(y, want), f"utt {u} (L={len(x)})"
    one = sea.resynth(utts[0], masks[0], binary=binary, frames_l_over_160=True)
    assert np.array_equal(one, oracle.resynth64(utts[0], masks[0], binary=binary, frames_l_over_160=True))


def test_ns_stream_plugin_flags_vs_oracle(oracle):
    """The streaming plug-in with the per-frame outputs of the reference's batch plug-in shape
    (SpeechFoundVar/Spec/Mel/VADNS + FrameCounter, NoiseSupExports.h:19-27), pushed in two calls so that
    the frame-dropping state crosses a state-blob save / load."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    x = corpus.synth_utterance(21, 80 * 120)          # no leading zero frames: the stream never gates
    tr = oracle.afe_trace(x)
    ns = oracle.ns_trace(x, want_state=False)
    frames = torch.from_numpy(x.astype(np.float32).reshape(1, -1, 80)).cuda().repeat(2, 1, 1)
    o1, p1, st, f1, c1 = sea.ns_streams_push(frames[:, :37].contiguous(), want_flags=True)
    o2, p2, st, f2, c2 = sea.ns_streams_push(frames[:, 37:].contiguous(), state=st, reset=False, want_flags=True)
    flags = torch.cat([f1, f2], 1).cpu().numpy()
    counter = torch.cat([c1, c2], 1).cpu().numpy()
    out = torch.cat([o1, o2], 1).cpu().numpy()
    want_flags = tr["flags"][:, :4] @ np.array([1, 2, 4, 8])
    for b in range(2):
        assert np.array_equal(flags[b], want_flags), np.nonzero(flags[b] != want_flags)[0][:5]
        assert np.array_equal(counter[b], tr["flags"][:, 4])
        assert np.array_equal(out[b, 4:].reshape(-1).view(np.uint32), ns["den_f32"].view(np.uint32))


def test_etsi_denoise_mapping_symbols(oracle, monkeypatch):
    """The reference's batch plug-in symbols (function/20141106_speech_enhancement/aurora_etsi/NoiseSupExports.h:35-42)
    in their 8 kHz-mode (etsi/ arithmetic) extension -- opt-in through SEA_MAPPING_8K=1 only: sm_glb_res is ignored as the
    reference ignores it (aurora_etsi/NoiseSup.cpp:913-922); the default, the 16 k-native variant, is
    tests/test_gpu_ns16k.py -- through the C ABI: global / thread instances, two
    func_Wiener calls on one thread instance (the state crosses the calls), zero frames skipped without touching
    their output entries (aurora_etsi/NoiseSup.cpp:1160-1171), a second thread instance starting afresh."""
    import ctypes
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    lib = sea.load()

    class In(ctypes.Structure):
        _fields_ = [("inData", ctypes.c_void_p), ("dataNum", ctypes.c_int)]

    class Out(ctypes.Structure):
        _fields_ = [("outData", ctypes.c_void_p), ("pSpeechFoundVar", ctypes.c_void_p), ("pSpeechFoundSpec", ctypes.c_void_p),
                    ("pSpeechFoundMel", ctypes.c_void_p), ("pSpeechFoundVADNS", ctypes.c_void_p), ("pFrameCounter", ctypes.c_void_p)]

    x = corpus.synth_utterance(5, 80 * 150)           # utterance 5: its first 400 samples (5 frames) are zero
    x[80 * 60:80 * 63] = 0                            # and three zero frames in the middle: skipped, no state change
    nfr = len(x) // 80
    tr = oracle.afe_trace(np.concatenate([x[:80 * 60], x[80 * 63:]]))   # what the processed frames must give
    ns = oracle.ns_trace(np.concatenate([x[:80 * 60], x[80 * 63:]]), want_state=False)
    keep = np.array([n for n in range(nfr) if np.any(x[80 * n:80 * n + 80])])
    assert len(keep) == nfr - 5 - 3
    glb, thd = ctypes.c_void_p(), ctypes.c_void_p()
    monkeypatch.setenv("SEA_MAPPING_8K", "1")         # read by global_init
    assert lib.etsi_denoise_mapping_global_init(ctypes.byref(glb), None) == 1
    monkeypatch.delenv("SEA_MAPPING_8K")
    for attempt in range(2):                          # the second pass: a fresh thread instance gives the same again
        assert lib.etsi_denoise_mapping_thread_init(ctypes.byref(thd), glb) == 1
        xf = x.astype(np.float32)
        out = np.full(nfr * 80, -7.0, np.float32)
        arrs = [np.full(nfr, -7, np.int32) for _ in range(5)]
        for a, b, fn in ((0, 37, lib.etsi_denoise_mapping_func_Wiener), (37, nfr, lib.etsi_denoise_mapping_func)):
            i = In(xf[80 * a:].ctypes.data, 80 * (b - a) + 13)          # 13 samples beyond the last whole frame: ignored
            o = Out(out[80 * a:].ctypes.data, *[v[a:].ctypes.data for v in arrs])
            args = (glb, thd, ctypes.byref(i), ctypes.byref(o)) + ((None,) if fn is lib.etsi_denoise_mapping_func_Wiener else ())
            assert fn(*args) == 0, lib.sea_last_error()
        lib.etsi_denoise_mapping_thread_release(ctypes.byref(thd))
        assert not thd.value
        var, spec, mel, vadns, cnt = arrs
        skipped = np.setdiff1d(np.arange(nfr), keep)
        assert np.all(out.reshape(nfr, 80)[skipped] == -7.0) and all(np.all(v[skipped] == -7) for v in arrs)
        # processed frame k of the compacted utterance = input frame keep[k]; the reference's trace counts from its first
        # non-zero frame too (DoAdvProcess gates the same way), 5 leading zero frames there as here
        tflags = tr["flags"][5:]                                           # per processed frame
        ran0 = tflags[:, 4] >= 1
        assert np.array_equal(cnt[keep][ran0], tflags[ran0, 4])
        assert np.all(cnt[keep][~ran0] == -7)
        for col, v in enumerate((var, spec, mel, vadns)):
            assert np.array_equal(v[keep][ran0], tflags[ran0, col]), f"flag {col}"
        produced = out.reshape(nfr, 80)[keep][4:]                           # outputs from the 5th processed frame on
        assert np.all(out.reshape(nfr, 80)[keep][:4] == -7.0)
        assert np.array_equal(produced.reshape(-1).view(np.uint32), ns["den_f32"].view(np.uint32))
    lib.etsi_denoise_mapping_global_release(ctypes.byref(glb))
    assert not glb.value


def test_ns_large_batch_kernel_form(oracle):
    """More than four utterances per CU selects the lower-register form of the pipelined kernel
    (transform address tables in LDS): same results."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    n = 4 * n_cu + 8
    base = [corpus.synth_utterance(90 + k, 800 + 80 * (k % 7)) for k in range(16)]
    utts = [base[k % 16] for k in range(n)]
    batch = sea.PackedBatch.from_arrays(utts)
    out, _, _ = sea.ns_denoise_batch(batch)
    torch.cuda.synchronize()
    got = batch.split(out, full_frames_only=True)
    want = [oracle.etsi_denoise(x)[: len(x) // 80 * 80] for x in base]
    for k in range(n):
        assert np.array_equal(got[k], want[k % 16]), f"utterance {k}"


def test_ns_all_kernel_forms_agree(oracle):
    """sea_ns_denoise_batch chooses among the forms of the same arithmetic by batch size (six waves per
    utterance, compiled for six or for seven waves per SIMD / four waves / four waves with less register use / two utterances
    per workgroup / one wave per utterance, streaming kernel and role-sequence kernel):
    forced one by one on the mixed corpus
    AND on the fill / drain corpus (0..12 frames, odd and even counts, 1..5 leading zero frames), every form
    matches the oracle bit for bit -- int16 audio, float stream and the index of the first output frame."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    lib = sea.load()
    utts = _mixed_corpus()
    utts += [corpus.synth_utterance(50 + n, 80 * n + (n * 13) % 80) for n in range(13)]
    utts += [np.concatenate([np.zeros(80 * k, np.int16), corpus.synth_utterance(70 + k, 80 * (7 + k % 2))]) for k in (1, 2, 3, 4, 5)]
    utts += [corpus.synth_utterance(90, 80 * 41), corpus.synth_utterance(91, 80 * 40 + 5)]
    batch = sea.PackedBatch.from_arrays(utts)
    traces = [oracle.ns_trace(x, want_state=False) for x in utts]
    prev = lib.sea_ns_kernel_form(0)
    try:
        for form in (1, 2, 3, 4, 5, 6, 7):
            lib.sea_ns_kernel_form(form)
            out, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
            torch.cuda.synchronize()
            got = batch.split(out, full_frames_only=True)
            gotf = batch.split(f32, full_frames_only=True)
            first_h = first.cpu().numpy()
            for u, (x, tr) in enumerate(zip(utts, traces)):
                nfr = len(x) // 80
                assert np.array_equal(got[u], tr["out_i16"][: nfr * 80]), f"form {form}, utterance {u} (L={len(x)})"
                assert int(first_h[u]) == (nfr - tr["nout"] if tr["nout"] else -1), f"form {form}, utterance {u}: first output"
                if tr["nout"]:
                    f0 = nfr - tr["nout"]
                    assert np.array_equal(gotf[u][f0 * 80: nfr * 80].view(np.uint32), tr["den_f32"].view(np.uint32)), \
                        f"form {form}, utterance {u}: float stream"
    finally:
        lib.sea_ns_kernel_form(prev)


def test_ns_time_slices_equal_one_launch(oracle):
    """sea_ns_denoise_batch_slice: the batch cut along the TIME axis, one launch per slice with the recursion carried in
    a state blob per utterance (what sea_denoise_utterances pipelines against its PCIe copies).  Cut points that fall
    inside leading-zero runs, inside the pipeline's four frames of latency, slices of a single frame, utterances that
    end before later slices start: int16 output, float stream and the index of the first output frame equal the oracle's
    (= the one-launch result) bit for bit, for both kernel forms the slices use."""
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    torch = _torch()
    lib = sea.load()
    utts = _mixed_corpus()
    utts += [corpus.synth_utterance(50 + n, 80 * n + (n * 13) % 80) for n in range(13)]
    utts += [np.concatenate([np.zeros(80 * k, np.int16), corpus.synth_utterance(70 + k, 80 * (7 + k % 2))]) for k in (1, 2, 3, 4, 5)]
    utts += [corpus.synth_utterance(90, 80 * 41), corpus.synth_utterance(91, 80 * 40 + 5)]
    utts.sort(key=len, reverse=True)                       # slices keep a PREFIX of the list: longest first
    traces = [oracle.ns_trace(x, want_state=False) for x in utts]
    nfr = np.array([len(x) // 80 for x in utts])
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    for bounds, rep in (((0, 1, 2, 3, 5, 9, 10, 40, 41, 97, int(nfr.max())), 1),       # the four-wave form
                        ((0, 7, 64, int(nfr.max())), 4 * n_cu // len(utts) + 1)):      # > 4 utterances per CU: the lower-register form
        pairs = sorted(list(zip(utts, traces)) * rep, key=lambda p: len(p[0]), reverse=True)
        ulist, tlist = [p[0] for p in pairs], [p[1] for p in pairs]
        n = len(ulist)
        nf = np.array([len(x) // 80 for x in ulist])
        state = torch.zeros((n, lib.sea_ns_slice_state_floats()), dtype=torch.float32, device="cuda")
        first = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        outs, outf = [[] for _ in range(n)], [[] for _ in range(n)]
        for k in range(len(bounds) - 1):
            b0, b1 = bounds[k], bounds[k + 1]
            act = int(np.sum(nf > b0))                      # a prefix, by the sort
            if act == 0:
                break
            parts = []
            for u in range(act):
                last = b1 >= nf[u]
                parts.append(ulist[u][80 * b0: (len(ulist[u]) if last else 80 * b1)])   # the trailing partial frame rides with the last slice
            sl = sea.PackedBatch.from_arrays(parts)
            o = torch.full_like(sl.data, -5)
            f32 = torch.zeros(sl.total, dtype=torch.float32, device="cuda")
            rc = lib.sea_ns_denoise_batch_slice(sl.data.data_ptr(), o.data_ptr(), f32.data_ptr(), sl.offsets.data_ptr(), sl.lengths.data_ptr(),
                                                None, first.data_ptr(), state.data_ptr(), act, b0, int(k > 0), None)
            assert rc == 0, lib.sea_last_error()
            torch.cuda.synchronize()
            for u, (a, b) in enumerate(zip(sl.split(o, full_frames_only=True), sl.split(f32, full_frames_only=True))):
                outs[u].append(a)
                outf[u].append(b)
        first_h = first.cpu().numpy()
        for u in range(n):
            tr, x = tlist[u], ulist[u]
            got = np.concatenate(outs[u]) if outs[u] else np.zeros(0, np.int16)
            assert np.array_equal(got, tr["out_i16"][: nf[u] * 80]), f"{len(bounds) - 1} slices, utterance {u} (L={len(x)})"
            if nf[u]:
                assert int(first_h[u]) == (nf[u] - tr["nout"] if tr["nout"] else -1), f"utterance {u}: first output"
            if tr["nout"]:
                f0 = nf[u] - tr["nout"]
                assert np.array_equal(np.concatenate(outf[u])[f0 * 80: nf[u] * 80].view(np.uint32), tr["den_f32"].view(np.uint32)), f"utterance {u}: float stream"


def test_denoise_utterances_one_long_utterance_in_slices(oracle):
    """A list of ONE two-minute utterance through sea_denoise_utterances: the time-slice pipeline cuts it into eight
    launches of one workgroup each (a chunk pipeline could not cut it at all); and a list whose short members leave the
    later slices (720 samples, 79, 0)."""
    import ctypes
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    _torch()
    lib = sea.load()
    for lens in ([16000 * 120 + 37], [16000 * 45, 16000 * 44 + 3, 80 * 9, 79, 0]):
        utts = [corpus.synth_utterance(400 + i, L) for i, L in enumerate(lens)]
        outs = [np.full(x.shape, 77, np.int16) for x in utts]
        n = len(utts)
        pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in utts])
        po = (ctypes.c_void_p * n)(*[y.ctypes.data for y in outs])
        pl = (ctypes.c_long * n)(*lens)
        assert lib.sea_denoise_utterances(pin, po, pl, n) == 0, lib.sea_last_error()
        for x, y, L in zip(utts, outs, lens):
            full = L // 80 * 80
            assert np.array_equal(y[:full], oracle.etsi_denoise(x)[:full]), f"L={L}"
            assert np.all(y[full:] == 77), f"L={L}: the trailing partial frame was written"
