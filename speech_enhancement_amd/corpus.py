"""Deterministic synthetic 16 kHz corpus.

Utterance u (0-based):
  * length L_u = 16000 * (2 + frac(u * 2654435761 / 2^32) * 4), rounded down to a multiple of 160
    (2-6 s, mean 4 s, all < 160 s so the reference's int16 frame counter never wraps: SURVEY F9)
  * speech-like part  sum_{h=1..8} (1/h) sin(2 pi h f0 i / 16000) * A(i),  f0 = 110 + (u mod 97),
    A(i) = 3000 * [(i mod 6400) < 3200]   (on/off at 2.5 Hz: exercises VAD and noise tracker)
  * plus uniform noise in [-700, 700] from the LCG  s <- s*1664525 + 1013904223 (uint32),
    seeded 12345+u, one step per sample, value ((s >> 16) % 1401) - 700
  * truncated to int16; the first 400 samples of every 5th utterance are zero (zero-frame gate,
    etsi/cpp/ParmInterface.c:250)
Masks for the resynthesis: m[f][c] = ((s >> 16) % 1000) / 1000 from an LCG seeded 777+u.
"""
import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


def lcg_stream(seed, n):
    """n successive LCG states after the update (vectorised: powers and the geometric series of
    the multiplier are taken modulo 2^32 by uint32 wrap-around)."""
    if n <= 0:
        return np.zeros(0, np.uint32)
    with np.errstate(over="ignore"):
        apow = np.cumprod(np.full(n, _A, dtype=np.uint32), dtype=np.uint32)         # a^1..a^n
        geo = np.cumsum(np.concatenate(([np.uint32(1)], apow[:-1])), dtype=np.uint32)  # sum_{k<i} a^k
        return apow * np.uint32(seed & 0xFFFFFFFF) + _C * geo


def utterance_length(u):
    frac = ((u * 2654435761) & 0xFFFFFFFF) / 2.0 ** 32
    return int(16000 * (2 + frac * 4)) // 160 * 160


def synth_utterance(u, length=None):
    L = utterance_length(u) if length is None else int(length)
    i = np.arange(L, dtype=np.float64)
    f0 = 110 + (u % 97)
    speech = np.zeros(L, np.float64)
    for h in range(1, 9):
        speech += np.sin(2 * np.pi * h * f0 * i / 16000.0) / h
    speech *= 3000.0 * ((np.arange(L) % 6400) < 3200)
    s = lcg_stream(12345 + u, L)
    noise = ((s >> np.uint32(16)) % np.uint32(1401)).astype(np.int64) - 700
    x = np.trunc(speech + noise).astype(np.int16)
    if u % 5 == 0:
        x[:400] = 0
    return x


def synth_mask(u, length):
    F = (int(length) - 320) // 160 + 1
    s = lcg_stream(777 + u, F * 64)
    return (((s >> np.uint32(16)) % np.uint32(1000)).astype(np.float32) / np.float32(1000.0)).reshape(F, 64)


def synth_corpus(n_utt, first=0, max_len=None):
    """List of int16 arrays for utterances first .. first+n_utt-1 (optionally length-capped)."""
    out = []
    for u in range(first, first + n_utt):
        L = utterance_length(u)
        if max_len is not None:
            L = min(L, int(max_len))
        out.append(synth_utterance(u, L))
    return out


def write_mask_text(f, utt_id, mask):
    """One Kaldi-style 64-column text matrix, byte for byte what the reference's make_single_IBM prints
    (enhancement_extract_test/cpp/show_IBM.cpp:194-208) and what its resynth driver parses
    (resyth_64sub_ori/cpp/main.cpp:84-145): SURVEY 8(f) #2."""
    mask = np.asarray(mask, dtype=np.float32)
    assert mask.ndim == 2 and mask.shape[1] == 64 and len(mask) >= 1
    f.write(f"{utt_id} [\n")
    for row in mask[:-1]:
        f.write("".join("%.7f " % v for v in row) + "\n ")
    f.write("".join("%.7f " % v for v in mask[-1]) + "]\n")


def read_mask_text(f):
    """Generator of (utt_id, float32 [rows,64]) from a text file of such matrices."""
    utt_id, rows = None, []
    for line in f:
        if "[" in line:
            if utt_id is not None:
                yield utt_id, np.asarray(rows, dtype=np.float32).reshape(-1, 64)
            utt_id, rows = line.split("[")[0].strip(), []
        elif utt_id is not None:
            vals = []
            for tok in line.split():
                try:
                    vals.append(float(tok))
                except ValueError:
                    break
            if vals:
                rows.append((vals + [0.0] * 64)[:64])
    if utt_id is not None:
        yield utt_id, np.asarray(rows, dtype=np.float32).reshape(-1, 64)
