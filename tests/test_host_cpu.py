"""CPU tests of the host side: the C ABI exports what include/sea_mi355x.h declares, host tables equal
the oracle's, the product fails loudly without a GPU (no CPU fallback), batch packing, the
synthetic corpus, and the multi-process sharding (gloo, world_size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sea_mi355x.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_header_declares_the_reference_entry_points():
    names = _declared_functions()
    for must in ("etsi_denoise", "etsi_denoise_synchronization", "etsi_denoise_16k",
                 "etsi_denoise_16k_synchronization", "rfft", "sea_ns_stream_alloc", "sea_ns_stream_init",
                 "sea_ns_stream_push", "sea_ns_stream_delete", "sea_compceps_frame", "sea_resynth64",
                 "sea_gammatone_filter", "sea_ns_denoise_batch", "sea_compceps_batch", "sea_resynth64_batch"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import _lib
    assert os.path.exists(sea.LIB_PATH), "build with make -C speech_enhancement_amd/csrc"
    lib = ctypes.CDLL(sea.LIB_PATH)
    declared = _declared_functions()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sea_mi355x.h but not exported"
    assert set(declared) == set(_lib.PROTOTYPES), set(declared) ^ set(_lib.PROTOTYPES)
    assert b"gfx950" in sea.load().sea_version()


def test_library_contains_gfx950_code_object():
    import speech_enhancement_amd as sea
    blob = open(sea.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"ns_denoise_kernel" in blob


def test_host_tables_match_oracle(oracle):
    """sea_tables_host needs no GPU: the product's own table builder (csrc/sea_tables.c) against the
    oracle's, entry by entry, bit for bit."""
    import speech_enhancement_amd as sea
    t = sea.tables()
    a, c = oracle.ns_tables(), oracle.cc_tables()
    for k in a:
        assert np.array_equal(t[k].view(np.uint32), a[k].view(np.uint32)), k
    for mine, theirs in (("hamming", "hamming"), ("dct", "dct"), ("ccStart", "melStart"), ("ccLen", "melLen"),
                         ("ccData", "melData")):
        assert np.array_equal(t[mine].view(np.uint32), c[theirs].view(np.uint32)), mine
    cf, bw, me = oracle.resynth_channels()
    assert np.array_equal(t["cf"], cf) and np.array_equal(t["bw"], bw) and np.array_equal(t["midEar"], me)


def test_compceps_mel_lane_map_is_complete_and_conflict_free():
    """No GPU needed: the lane map of the tiled CompCeps kernels' mel pass (csrc/sea_tables.c: cc_mel_lanes).  Every (frame of the
    pair, band) item sits in exactly one lane; a lane's 24 taps, read as 12 aligned pairs from an even bin, carry its band's
    triangle weights at the band's bins and zeros elsewhere; and no two lanes of a 32-lane group start on the same pair of the
    64 LDS banks (ds_read_b64), which holds for every tap pair since all lanes advance together."""
    import ctypes
    import speech_enhancement_amd as sea
    lib = sea.load()
    base, fb = np.zeros(64, np.int32), np.zeros(64, np.int32)
    w = np.zeros((24, 64), np.float32)
    assert lib.sea_debug_cc_mel_lanes(base.ctypes.data_as(ctypes.c_void_p), fb.ctypes.data_as(ctypes.c_void_p),
                                      w.ctypes.data_as(ctypes.c_void_p)) == 0
    t = sea.tables()
    start, length, data = t["ccStart"], t["ccLen"], t["ccData"].reshape(23, -1)
    row = 152  # SEA_CC_PWROW: words per power row
    seen = set()
    for lane in range(64):
        if fb[lane] < 0:
            assert not w[:, lane].any()
            continue
        h, band = divmod(int(fb[lane]), 24)
        assert h in (0, 1) and 0 <= band < 23 and (h, band) not in seen
        seen.add((h, band))
        first = int(base[lane]) - row * h  # the lane's first bin
        assert first % 2 == 0 and int(base[lane]) % 2 == 0 and 0 <= first <= start[band]
        assert first + 24 >= start[band] + length[band] and first + 24 <= row
        want = np.zeros(24, np.float32)
        want[start[band] - first: start[band] - first + length[band]] = data[band, :length[band]]
        assert np.array_equal(w[:, lane].view(np.uint32), want.view(np.uint32)), (lane, band)
    assert len(seen) == 46
    for g in (0, 1):
        lanes = [l for l in range(32 * g, 32 * g + 32) if fb[l] >= 0]
        banks = [(int(base[l]) // 2) % 32 for l in lanes]
        assert len(set(banks)) == len(banks), f"group {g}: two lanes on one pair of banks"


def test_ns16k_host_tables_and_schedule_match_oracle(oracle):
    """SURVEY 8(f) #4, no GPU needed: the product's tables of the 16 k-native variant (csrc/sea_tables.c) against the
    oracle's, bit for bit, and its table-driven transform schedule (digit-reversal places, butterflies per pass,
    twiddles: what ns16k_kernel.hip walks) run on the host against the oracle's rfft (x, 512, 8)."""
    import ctypes
    import speech_enhancement_amd as sea
    t, a = sea.ns16k_tables(), oracle.ns16k_tables()
    for k in a:
        assert np.array_equal(t[k].view(np.uint32), a[k].view(np.uint32)), k
    lib = sea.load()
    rng = np.random.default_rng(5)
    for _ in range(12):
        x = (rng.standard_normal(512) * 10 ** rng.uniform(-3, 4)).astype(np.float32)
        y = x.copy()
        lib.sea_ns16k_fft_host(y.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(y.view(np.uint32), oracle.ns16k_rfft(x).view(np.uint32))
        z = x.copy()   # the pipelined kernel's tables: register-resident start, one item per lane and level, swizzled work area
        lib.sea_ns16k_pipe_fft_host(z.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(z.view(np.uint32), y.view(np.uint32))
    # the twiddles are generated constants (csrc/ns16k_twiddles.inc), not the host libm's cosf / sinf at initialisation: both
    # table walks against what the REFERENCE's own rfft.cpp (compiled here) made of the fixture's frames
    g = np.load(os.path.join(ROOT, "tests", "golden", "aurora_golden.npz"))
    for x, want in zip(g["frames"], g["rfft512_8"]):
        for fn in (lib.sea_ns16k_fft_host, lib.sea_ns16k_pipe_fft_host):
            y = np.array(x, dtype=np.float32, copy=True)
            fn(y.ctypes.data_as(ctypes.c_void_p))
            assert np.array_equal(y.view(np.uint32), want.view(np.uint32))


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product must fail loudly, never compute on the CPU."""
    if _have_gpu():
        pytest.skip("GPU present")
    import speech_enhancement_amd as sea
    x = np.ones(800, np.int16)
    with pytest.raises(sea.SeaError):
        sea.etsi_denoise(x)
    with pytest.raises(sea.SeaError):
        sea.DoCompCeps(np.zeros(201, np.float32))
    with pytest.raises(sea.SeaError):
        sea.resynth(np.zeros(640, np.int16), np.zeros((3, 64), np.float32))
    lib = sea.load()
    assert lib.sea_init(-1) != 0 and lib.sea_last_error()


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under speech_enhancement_amd/ may reference it."""
    pkg = os.path.join(ROOT, "speech_enhancement_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hip", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "libsea_oracle" not in text and "from oracle" not in text and "import oracle" not in text, fn
                assert not re.search(r"\b(ora|ref)_[a-z0-9_]+\s*\(", text), fn


def test_packed_batch_layout():
    import speech_enhancement_amd as sea
    lengths = [1000, 0, 79, 4001, 8]
    offs, total, order = sea.PackedBatch.layout(lengths)
    assert list(offs) == [0, 1000, 1000, 1080, 5088] and total == 5096
    assert all(o % 8 == 0 for o in offs)
    assert list(order) == [3, 0, 2, 4, 1]                      # longest first, stable
    b = sea.PackedBatch.from_arrays([np.arange(n, dtype=np.int16) for n in lengths], device="cpu")
    assert b.n_utt == 5 and b.n_frames == 12 + 0 + 0 + 50 + 0
    parts = b.split(b.data)
    assert [len(p) for p in parts] == lengths and np.array_equal(parts[3], np.arange(4001, dtype=np.int16))
    assert [len(p) for p in b.split(b.data, full_frames_only=True)] == [960, 0, 0, 4000, 0]


def test_corpus_is_deterministic_and_matches_scalar_lcg():
    from speech_enhancement_amd import corpus
    s, ref = 12345, []
    for _ in range(50):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        ref.append(s)
    assert list(corpus.lcg_stream(12345, 50)) == ref
    a, b = corpus.synth_utterance(5), corpus.synth_utterance(5)
    assert np.array_equal(a, b) and a.dtype == np.int16
    assert len(a) == corpus.utterance_length(5) and len(a) % 160 == 0
    assert not np.any(a[:400]) and np.any(a[400:480])           # every 5th utterance starts silent
    lens = [corpus.utterance_length(u) for u in range(2000)]
    assert 32000 <= min(lens) and max(lens) <= 96000 and abs(np.mean(lens) - 64000) < 2000
    m = corpus.synth_mask(3, 4800)
    assert m.shape == ((4800 - 320) // 160 + 1, 64) and 0 <= m.min() and m.max() < 1


def test_lpt_and_block_sharding():
    from speech_enhancement_amd import corpus
    from speech_enhancement_amd.shard import block_shard, lpt_shards
    lengths = [corpus.utterance_length(u) for u in range(1000)]
    shards = lpt_shards(lengths, 8)
    allidx = np.concatenate(shards)
    assert sorted(allidx) == list(range(1000))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert (max(loads) - min(loads)) / np.mean(loads) < 0.01
    covered = [i for r in range(8) for i in block_shard(1003, 8, r)]
    assert covered == list(range(1003))


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from speech_enhancement_amd import corpus
from speech_enhancement_amd.shard import lpt_shards, block_shard, reduce_job
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
lengths = [corpus.utterance_length(u) for u in range(64)]
mine = lpt_shards(lengths, world)[rank]
frames = int(sum(lengths[i] // 80 for i in mine))
gathered = [None] * world
dist.all_gather_object(gathered, [int(i) for i in mine])
flat = sorted(i for g in gathered for i in g)
assert flat == list(range(64)), flat
total, tmax = reduce_job(frames, 1.0 + rank, dist, torch.device("cpu"))
assert total == sum(l // 80 for l in lengths) and tmax == float(world)
blk = list(block_shard(64, world, rank))
dist.all_gather_object(gathered, blk)
assert sorted(i for g in gathered for i in g) == list(range(64))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharding_world_size_2_gloo(tmp_path):
    """The N>1 path without GPUs: two gloo ranks shard a corpus, cover it exactly once, and reduce
    the job totals the way bench.py does (frames: sum, seconds: max)."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher: the parent (which never touches torch or the GPU) starts two
    rank processes, they meet over gloo and rank 0 prints ONE JSON line with n_gpus = 2 (--rehearse-cpu skips
    the GPU work).  configs[4] mode: rank r takes LPT shard r of the 100 000-utterance corpus."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--utts", "16",
                        "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    from speech_enhancement_amd import corpus
    assert j["n_gpus"] == 2 and j["seconds_max"] == 2.0
    assert j["total_frames"] == 3 * sum(corpus.utterance_length(u) // 80 for u in range(32))
    # beside the weak-scaled headline every --gpus N run carries BASELINE configs[4]: the WHOLE 100 000-utterance corpus
    # cut into N shards, one per rank (strong scaling) -- what a driver-run SCALE measures for N = 1, 2, 4, 8
    c4 = j["configs4"]
    assert c4["n_gpus"] == 2 and c4["scaling"] == "strong" and c4["utterances_rank0"] == 50000 and c4["seconds_max"] == 3.0
    assert c4["total_frames"] == 2 * sum(corpus.utterance_length(u) // 80 for u in range(100000))
    assert abs(c4["frames_rank0"] * 2 * 2 / c4["total_frames"] - 1) < 1e-3   # the two shards are balanced by samples
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--corpus-utts",
                        "100000", "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["utterances_rank0"] == 12500 and j["n_gpus"] == 2
    # the parent process of a real run must not have imported torch before spawning
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def run_rank")]
    assert "\nimport torch" not in head.replace("    import torch", "")


def test_bench_launcher_ends_the_job_when_a_rank_dies():
    """A rank that exits after the rendezvous would leave the others at the next barrier for gloo's timeout: the
    launcher ends them (the processes it started) and reports the failure at once."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SEA_BENCH_REHEARSE_FAIL_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--utts", "16",
                        "--rehearse-cpu"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert time.time() - t0 < 120
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "no GPU visible" in r.stderr and not r.stdout.strip()


def test_integration_shim_compiles_against_the_reference_header(tmp_path):
    """INTEGRATION.md section 2 shows the file a maintainer adds to the reference tree (etsi/cpp/SeaPlugin.c) to
    install the engine into the FEParamsX vtable (etsi/cpp/ParmInterface.h:120-178).  Keep it compiling against
    the reference's own header (this container only: /root/reference does not travel)."""
    ref = "/root/reference/etsi/cpp"
    if not os.path.exists(os.path.join(ref, "ParmInterface.h")):
        pytest.skip("/root/reference absent")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```c\n(/\* etsi/cpp/SeaPlugin\.c.*?)```", text, flags=re.S)
    assert m, "SeaPlugin.c block not found in INTEGRATION.md"
    src = tmp_path / "SeaPlugin.c"
    src.write_text(m.group(1))
    r = subprocess.run(["gcc", "-std=gnu99", "-Wall", "-Werror=implicit-function-declaration",
                        "-Werror=incompatible-pointer-types", "-Werror=int-conversion", "-c", "-I", ref, "-I",
                        os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "SeaPlugin.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", str(tmp_path / "SeaPlugin.o")], capture_output=True, text=True).stdout
    for sym in ("sea_ns_stream_alloc", "sea_ns_stream_init", "sea_ns_stream_push", "sea_ns_stream_delete",
                "sea_compceps_frame", "AdvProcessAlloc"):
        assert re.search(r"\bU " + sym + r"\b", nm), sym
    assert re.search(r"\bT SeaAdvProcessAlloc\b", nm)


def test_mask_text_format_roundtrip(tmp_path):
    """the Kaldi-style text matrix between feature extraction, the DNN and resynth
    (writer enhancement_extract_test/cpp/show_IBM.cpp:194-208, reader resyth_64sub_ori/cpp/main.cpp:
    84-145).  The C writer of the host library and the Python mirror emit the same bytes; the C
    reader gets the %.7f-rounded values back, matrix after matrix."""
    import subprocess
    from speech_enhancement_amd import corpus
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "speech_enhancement_amd", "host")
    drv = tmp_path / "drv.c"
    drv.write_text(r'''
#include "sea_host.h"
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv)
{   /* drv <bin-in> <txt-out> <bin-back>: n matrices of [rows][64] floats with ids u0, u1, ... */
    FILE *bi = fopen(argv[1], "rb"), *tx = fopen(argv[2], "w"), *bo;
    int n, u; long rows[16]; float *m[16]; char id[SEA_FILE_LEN];
    fread(&n, sizeof n, 1, bi);
    for (u = 0; u < n; u++) {
        fread(&rows[u], sizeof rows[u], 1, bi);
        m[u] = malloc(rows[u] * 64 * sizeof(float));
        fread(m[u], sizeof(float), rows[u] * 64, bi);
        sprintf(id, "u%d", u);
        if (sea_mask_text_write(tx, id, m[u], rows[u])) return 1;
    }
    fclose(tx);
    tx = fopen(argv[2], "r");
    bo = fopen(argv[3], "wb");
    for (u = 0; u < n; u++) {
        float *back = calloc(rows[u] * 64, sizeof(float));
        long got = sea_mask_text_read(tx, id, back, rows[u]);
        char want[16]; sprintf(want, "u%d", u);
        if (got != rows[u] || strcmp(id, want)) return 2;
        fwrite(back, sizeof(float), rows[u] * 64, bo);
    }
    if (sea_mask_text_read(tx, id, m[0], 1) != -1) return 3;
    return 0;
}
''')
    exe = tmp_path / "drv"
    subprocess.run(["gcc", "-O1", "-std=gnu99", "-I", host, "-o", str(exe), str(drv), os.path.join(host, "sea_host.c")],
                   check=True)
    rng = np.random.default_rng(4)
    mats = [rng.random((r, 64)).astype(np.float32) for r in (1, 2, 9)]
    mats[1][0, :3] = [0.0, 1.0, 0.5]
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(np.int32(len(mats)).tobytes())
        for m in mats:
            f.write(np.int64(len(m)).tobytes())
            f.write(m.tobytes())
    subprocess.run([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "m.txt"), str(tmp_path / "back.bin")], check=True)
    with open(tmp_path / "py.txt", "w") as f:
        for u, m in enumerate(mats):
            corpus.write_mask_text(f, f"u{u}", m)
    assert (tmp_path / "m.txt").read_bytes() == (tmp_path / "py.txt").read_bytes()
    text = (tmp_path / "m.txt").read_text()
    assert text.startswith("u0 [\n") and text.endswith(" ]\n") and "\n u" not in text
    back = np.fromfile(tmp_path / "back.bin", dtype=np.float32)
    want = np.concatenate([np.array([[float("%.7f" % v) for v in row] for row in m], np.float32).ravel() for m in mats])
    assert np.array_equal(back, want)
    with open(tmp_path / "m.txt") as f:
        got = list(corpus.read_mask_text(f))
    assert [g[0] for g in got] == ["u0", "u1", "u2"]
    assert all(np.array_equal(g[1].ravel(), want[o:o + m.size]) for g, m, o in
               zip(got, mats, np.cumsum([0] + [m.size for m in mats[:-1]])))
