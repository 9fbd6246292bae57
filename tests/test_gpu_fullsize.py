"""Full-size GPU parity (BASELINE configs[1..3] sizes): the whole 1024-utterance synthetic corpus
through the batch entry points, checked against the oracle on EVERY utterance (NoiseSup; the oracle
runs on all host threads) or on a spread sample plus size-independent properties (resynth)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def shard():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible")
    import bench
    return bench.build_shard(1024, 0, torch.device("cuda", 0))


def _threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def test_noisesup_1024_utterances_every_sample(shard, oracle):
    """configs[1]: 817 680 frames.  Every utterance bit-for-bit against the oracle; the pipelined
    and the one-wave kernels agree; the 320-sample latency / zero-frame gate hold everywhere."""
    import torch
    import speech_enhancement_amd as sea
    out, _, first = sea.ns_denoise_batch(shard)
    torch.cuda.synchronize()
    got = shard.split(out, full_frames_only=True)
    host = shard.data.cpu().numpy()
    utts = [host[o:o + l] for o, l in zip(shard.host_offsets, shard.host_lengths)]
    oracle.etsi_denoise(utts[0][:800])
    with ThreadPoolExecutor(_threads()) as ex:
        want = list(ex.map(oracle.etsi_denoise, utts))
    bad = [u for u in range(len(utts)) if not np.array_equal(got[u], want[u][: len(got[u])])]
    assert not bad, f"{len(bad)} utterances differ, first {bad[:5]}"
    first = first.cpu().numpy()
    for u, x in enumerate(utts):
        onset = 5 if u % 5 == 0 else 0           # 400 leading zero samples = 5 silent frames
        assert first[u] == onset + 4
        assert not np.any(got[u][: (onset + 4) * 80])
    assert shard.n_frames == 817680


def test_noisesup_configs4_shard_12500_utterances(oracle):
    """configs[4]: one of the eight LPT shards of the 100 000-utterance corpus -- 12 500 utterances, ~10 M
    frames, in ONE launch (48 utterances per CU: the large-batch kernel form).  A spread sample of 256
    utterances bit-for-bit against the oracle; on ALL of them the frame indexing properties: first output
    frame = onset + 4, nothing before it, and the shard's load within 0.1 % of the corpus mean."""
    import torch
    import bench
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    ids = bench.corpus_shard_ids(100000, 8, 0)
    assert len(ids) == 12500
    all_len = np.array([corpus.utterance_length(u) for u in range(100000)], dtype=np.int64)
    assert abs(all_len[ids].sum() / (all_len.sum() / 8) - 1) < 1e-3
    batch = bench.build_shard_ids(ids, torch.device("cuda", 0))
    assert batch.n_frames == int(np.sum(all_len[ids] // 80))
    out, _, first = sea.ns_denoise_batch(batch)
    torch.cuda.synchronize()
    host_in = batch.data.cpu().numpy()
    host_out = out.cpu().numpy()
    first = first.cpu().numpy()
    pick = list(range(0, 12500, 49))[:256]
    assert len(pick) == 256
    utts = [host_in[batch.host_offsets[k]: batch.host_offsets[k] + batch.host_lengths[k]] for k in pick]
    oracle.etsi_denoise(utts[0][:800])
    with ThreadPoolExecutor(_threads()) as ex:
        want = list(ex.map(oracle.etsi_denoise, utts))
    bad = [k for k, w in zip(pick, want)
           if not np.array_equal(host_out[batch.host_offsets[k]: batch.host_offsets[k] + batch.host_lengths[k]], w)]
    assert not bad, f"{len(bad)} of 256 sampled utterances differ, first {bad[:5]}"
    for k, u in enumerate(ids):
        onset = 5 if u % 5 == 0 else 0
        o = int(batch.host_offsets[k])
        assert first[k] == onset + 4, (k, u, first[k])
        assert not np.any(host_out[o: o + (onset + 4) * 80])
        assert np.any(host_out[o + (onset + 4) * 80: o + (onset + 5) * 80])
    del out, batch
    torch.cuda.empty_cache()


def test_noisesup_batch_composition_independence(shard):
    """An utterance's result does not depend on what else is in the batch or on launch order."""
    import torch
    import speech_enhancement_amd as sea
    full, _, _ = sea.ns_denoise_batch(shard)
    host = shard.data.cpu().numpy()
    pick = [3, 500, 1023]
    small = sea.PackedBatch.from_arrays([host[shard.host_offsets[u]: shard.host_offsets[u] + shard.host_lengths[u]]
                                         for u in pick])
    part, _, _ = sea.ns_denoise_batch(small, use_order=False)
    torch.cuda.synchronize()
    a, b = shard.split(full), small.split(part)
    for k, u in enumerate(pick):
        assert np.array_equal(a[u], b[k])


@pytest.mark.parametrize("binary", [False, True])
def test_resynth_1024_utterances(shard, oracle, binary):
    """configs[2]/[3]: 64-band resynthesis of the whole corpus (~17 GB intermediate).  A spread sample
    of utterances against the oracle, plus properties on all of them."""
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    masks_host = [corpus.synth_mask(u, int(L)) for u, L in enumerate(shard.host_lengths)]
    masks = sea.MaskBatch.from_arrays(masks_host)
    out, scratch = sea.resynth_batch(shard, masks, binary=binary)
    torch.cuda.synchronize()
    got = shard.split(out)
    host = shard.data.cpu().numpy()
    sample = list(range(0, 1024, 64))

    def ref(u):
        x = host[shard.host_offsets[u]: shard.host_offsets[u] + shard.host_lengths[u]]
        return oracle.resynth64(x, masks_host[u], binary=binary)
    with ThreadPoolExecutor(_threads()) as ex:
        want = list(ex.map(ref, sample))
    for u, w in zip(sample, want):
        assert np.array_equal(got[u], w), f"utt {u}"
    # beyond the last mask frame's window the output is silent (no frame covers it)
    for u, L in enumerate(shard.host_lengths):
        F = (int(L) - 320) // 160 + 1
        assert not np.any(got[u][F * 160:])
    # an all-zero mask gives silence for the whole batch (reuses the scratch buffer)
    zero = sea.MaskBatch.from_arrays([np.zeros_like(m) for m in masks_host])
    out0, _ = sea.resynth_batch(shard, zero, binary=binary, scratch=scratch)
    torch.cuda.synchronize()
    assert int(torch.count_nonzero(out0)) == 0


def test_afe_feature_chain_1024_utterances(shard, oracle):
    """SURVEY 8(f) #3 at configs[1] size: every utterance's emitted feature frames (c1..c12, c0, logE,
    VAD flag) against the reference-pinned oracle -- exact VAD decisions, features within 1e-3."""
    import speech_enhancement_amd as sea
    res = sea.afe_features_batch(shard)
    host = shard.data.cpu().numpy()
    utts = [host[o:o + l] for o, l in zip(shard.host_offsets, shard.host_lengths)]

    def ref(x):
        return oracle.afe_trace(x)["vad_out"]
    ref(utts[0][:1600])
    with ThreadPoolExecutor(_threads()) as ex:
        want = list(ex.map(ref, utts))
    worst, frames, speech = 0.0, 0, 0
    for u, (g, w) in enumerate(zip(res["feats"], want)):
        assert g.shape == w.shape, f"utt {u}: {g.shape} vs {w.shape}"
        assert np.array_equal(g[:, 14], w[:, 14]), f"utt {u}: VAD decisions differ"
        worst = max(worst, float(np.abs(g[:, :14] - w[:, :14]).max()))
        frames += len(g)
        speech += int(g[:, 14].sum())
    print(f"AFE chain: {frames} feature frames, {speech} flagged speech, worst |delta| = {worst}")
    assert worst <= 1e-3
    assert 0 < speech < frames
