#!/usr/bin/env python3
"""bench.py -- headline measurement of the NoiseSup hot path on MI355X.

One "step" = one pass of the hot path over one batch: etsi_denoise semantics (two-stage Wiener
NoiseSup incl. both 256-point rffts, int16 in -> int16 out) for every utterance of the rank's shard,
resident in HBM when the timed region starts.  Unit of throughput: NoiseSup frames (80 samples) per
second.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads
  default               BASELINE configs[1]: every rank owns its own 1024-utterance shard of the
                        SURVEY 8(d) corpus (weak scaling)
  --corpus-utts M       BASELINE configs[4]: the M-utterance corpus (100000) cut into --shards S
                        (default 8) shards balanced by samples (longest-processing-time greedy,
                        shard.py::lpt_shards); rank r runs shard r, so `--gpus 8 --corpus-utts 100000`
                        is the whole corpus at 12 500 utterances per GPU and `--gpus 1 --corpus-utts
                        100000` is one such shard on one GPU.

Multi-GPU: utterances are independent, so ranks shard the corpus with NO data-path collective (the
reference's own parallel harness is a shared-counter thread pool over files,
function/20141106_speech_enhancement/aurora_speech_enhancement/aurora_speech_enhancement.cpp:111-121,
311-327).  torch.distributed (gloo, host side) carries only the barrier and the final (sum of frames,
max of seconds).  When launched WITHOUT a launcher (`WORLD_SIZE` unset) and --gpus N > 1, this
process starts the N ranks itself, as child processes, BEFORE anything touches the GPU (torch is not
even imported in the parent), relays rank 0's JSON line and exits with the worst child status.

Rank 0 prints ONE JSON line.  Besides the headline it carries `roofline`, `cpu_baseline` (N = 1) and
`also`: the other hot-path kernels (resynth soft / IBM = configs[2] / [3], CompCeps, rfft256), a few
steps each, timed in the same run.
"""
import argparse
import contextlib
import datetime
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NS_BYTES_PER_FRAME = 320          # 160 B int16 read + 160 B int16 written
RS_BYTES_PER_HOP = 896            # 320 B in + 320 B out + 256 B mask row
CC_BYTES_PER_FRAME = 320 + 56     # 80 new floats read + 14 floats written
FFT_BYTES_PER_FRAME = 2048        # 256 floats in + 256 floats out
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
UTTS_PER_GPU = 1024
CSRC = os.path.join(ROOT, "speech_enhancement_amd", "csrc")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=UTTS_PER_GPU, help="utterances per GPU (default: configs[1])")
    ap.add_argument("--corpus-utts", type=int, default=0,
                    help="configs[4]: size of the whole corpus (100000); rank r runs LPT shard r of --shards")
    ap.add_argument("--shards", type=int, default=8, help="number of LPT shards the --corpus-utts corpus is cut into")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the resynth / CompCeps / rfft256 lines")
    ap.add_argument("--also-steps", type=int, default=5)
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the PCIe-inclusive host-buffer timings")
    ap.add_argument("--no-configs4", action="store_true",
                    help="skip the configs[4] lines (the one-shard line of the also-array and the whole-corpus `configs4` object)")
    ap.add_argument("--configs4-utts", type=int, default=100000,
                    help="size of the corpus of the `configs4` object (BASELINE configs[4]: 100000), cut into one LPT shard per rank")
    ap.add_argument("--cpu-utts", type=int, default=1024, help="utterances in the bounded CPU sample (1024 = ~10 core-seconds)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads for the CPU baseline (default 0 = every CPU this process may run on; 16 is reported beside)")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="no GPU work: launcher, gloo rendezvous, shard assignment and the job reduction only (CPU tests)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# rank launcher (parent process: no torch, no HIP)
# ------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """Start n rank processes of this script (env: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR/PORT), wait
    for all, return the worst exit status.  Rank 0 inherits stdout (its one JSON line is the result)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SEA_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # a rank that dies would leave the others waiting at the next barrier: when one exits non-zero, end the rest
    # (exactly the processes started here).  The same when this launcher is interrupted or told to stop (SIGTERM
    # from a harness timeout): no rank may be left behind holding a GPU at a gloo barrier.
    import signal

    def _stop(signum, frame):
        raise KeyboardInterrupt(f"signal {signum}")
    old = signal.signal(signal.SIGTERM, _stop)
    rc = 0
    live = list(procs)
    try:
        while live:
            time.sleep(0.2)
            for p in list(live):
                if p.poll() is None:
                    continue
                live.remove(p)
                rc = max(rc, abs(p.returncode))
                if p.returncode != 0:
                    for q in live:
                        q.terminate()
    except KeyboardInterrupt:
        rc = max(rc, 130)
    finally:
        signal.signal(signal.SIGTERM, old)
        for q in procs:
            if q.poll() is None:
                q.terminate()
        deadline = time.time() + 10.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()
    return rc


# ------------------------------------------------------------------------------------------------
# synthetic corpus in HBM
# ------------------------------------------------------------------------------------------------
_LCG_CACHE = {}


def _lcg_tables(n, device):
    """a^k and c * sum_{j<k} a^j (mod 2^32) for k = 1..n as int64 tensors: state k of the LCG seeded s
    is (apow[k] * s + cgeo[k]) mod 2^32 (speech_enhancement_amd/corpus.py::lcg_stream)."""
    import torch
    key = str(device)
    if key not in _LCG_CACHE or len(_LCG_CACHE[key][0]) < n:
        from speech_enhancement_amd import corpus
        one = corpus.lcg_stream(1, n).astype(np.int64)   # apow + cgeo
        zero = corpus.lcg_stream(0, n).astype(np.int64)  # cgeo
        apow = (one - zero) & 0xFFFFFFFF
        _LCG_CACHE[key] = (torch.from_numpy(apow).to(device), torch.from_numpy(zero).to(device))
    return _LCG_CACHE[key]


def build_shard_ids(ids, device, chunk=128):
    """The SURVEY 8(d) corpus utterances `ids`, packed into HBM.  Evaluated on the device in chunks of
    `chunk` utterances (float64 harmonics, int64 LCG); formula identical, element by element, to
    speech_enhancement_amd.corpus.synth_utterance."""
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    ids = [int(u) for u in ids]
    lengths = np.array([corpus.utterance_length(u) for u in ids], dtype=np.int64)
    offsets, total, order = sea.PackedBatch.layout(lengths)
    data = torch.zeros(max(total, 8), dtype=torch.int16, device=device)
    for c0 in range(0, len(ids), chunk):
        cid = ids[c0:c0 + chunk]
        Lc = int(lengths[c0:c0 + chunk].max())
        apow, cgeo = _lcg_tables(96000, device)
        i = torch.arange(Lc, dtype=torch.float64, device=device)
        speech = torch.zeros((len(cid), Lc), dtype=torch.float64, device=device)
        for h in range(1, 9):
            w = torch.tensor([2 * np.pi * h * (110 + (u % 97)) for u in cid], dtype=torch.float64, device=device)
            speech += torch.sin(w[:, None] * i[None, :] / 16000.0) / h
        speech *= (3000.0 * ((torch.arange(Lc, device=device) % 6400) < 3200))[None, :]
        seeds = torch.tensor([(12345 + u) & 0xFFFFFFFF for u in cid], dtype=torch.int64, device=device)
        s = (apow[None, :Lc] * seeds[:, None] + cgeo[None, :Lc]) & 0xFFFFFFFF
        noise = ((s >> 16) % 1401) - 700
        x = torch.trunc(speech + noise.to(torch.float64)).to(torch.int16)
        for k, u in enumerate(cid):
            if u % 5 == 0:
                x[k, :400] = 0
            L, o = int(lengths[c0 + k]), int(offsets[c0 + k])
            data[o:o + L] = x[k, :L]
        del speech, s, noise, x
    return sea.PackedBatch(data, torch.from_numpy(offsets).to(device), torch.from_numpy(lengths).to(device),
                           torch.from_numpy(order).to(device), offsets, lengths)


def build_shard(n_utt, first, device):
    """Utterances first .. first+n_utt-1 (the configs[1] shard of a rank)."""
    return build_shard_ids(range(first, first + n_utt), device)


def corpus_shard_ids(corpus_utts, shards, rank):
    """configs[4]: utterance ids of LPT shard `rank` of the corpus_utts-utterance corpus."""
    from speech_enhancement_amd import corpus
    from speech_enhancement_amd.shard import lpt_shards
    lengths = [corpus.utterance_length(u) for u in range(corpus_utts)]
    return lpt_shards(lengths, shards)[rank]


def build_masks(batch, ids, device):
    """Ratio masks of SURVEY 8(d) for the utterances of `batch` (LCG seeded 777 + u), packed row-wise."""
    import torch
    import speech_enhancement_amd as sea
    rows = (np.asarray(batch.host_lengths) - 320) // 160 + 1
    offs = np.concatenate(([0], np.cumsum(rows)[:-1])).astype(np.int64)
    total = int(rows.sum())
    data = torch.empty((total, 64), dtype=torch.float32, device=device)
    apow, cgeo = _lcg_tables(96000, device)
    for k, u in enumerate(ids):
        n = int(rows[k]) * 64
        s = (apow[:n] * ((777 + int(u)) & 0xFFFFFFFF) + cgeo[:n]) & 0xFFFFFFFF
        data[int(offs[k]):int(offs[k]) + int(rows[k])] = (((s >> 16) % 1000).to(torch.float32) / 1000.0).view(-1, 64)
    return sea.MaskBatch(data, torch.from_numpy(offs).to(device), offs, rows)


# ------------------------------------------------------------------------------------------------
# CPU baseline (the oracle as the thing timed -- allowed here and only here)
# ------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def quiet_stderr():
    """The reference prints 'NO SPEECH DETECTED !' on every call (AdvFrontEnd.c:202-203)."""
    sys.stderr.flush()
    saved = os.dup(2)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)
    try:
        yield
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
        os.close(saved)


@contextlib.contextmanager
def stdout_to_stderr():
    """gloo prints its rendezvous banner on fd 1; stdout must carry the one JSON line only."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def init_gloo(rank, world):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with stdout_to_stderr():
        dist.init_process_group("gloo", rank=rank, world_size=world,   # host side: barrier + two scalars
                                timeout=datetime.timedelta(seconds=600))
        dist.barrier()


def _time_cpu(lib, symbol, utts, threads, seconds=2.0):
    """Frames per second of `symbol` (an etsi_denoise-shaped C function of the ctypes library `lib`) over `utts` on
    `threads` host threads: oracle/cpu_pool.c's workers pull utterances from a shared counter (the reference harness's own
    shape, aurora_speech_enhancement.cpp:111-121), so a many-core host is not paced by a Python dispatcher.  One pass to
    warm up (tables, page faults), two for an estimate, then as many as fill about `seconds`.  Returns (frames/s, passes, outs)."""
    import ctypes
    pool = ctypes.CDLL(os.path.join(ROOT, "oracle", "libsea_cpupool.so"))
    pool.sea_cpu_pool_run.restype = ctypes.c_double
    pool.sea_cpu_pool_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                      ctypes.c_int, ctypes.c_int]
    fn = ctypes.cast(getattr(lib, symbol), ctypes.c_void_p)
    n = len(utts)
    outs = [np.zeros_like(u) for u in utts]
    pin = (ctypes.c_void_p * n)(*[u.ctypes.data for u in utts])
    pout = (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs])
    lens = (ctypes.c_long * n)(*[u.size for u in utts])
    frames = int(sum(u.size // 80 for u in utts))
    with quiet_stderr():
        pool.sea_cpu_pool_run(fn, pin, pout, lens, n, threads, 1)       # tables, page faults
        dt = pool.sea_cpu_pool_run(fn, pin, pout, lens, n, threads, 2)  # an estimate
        if dt <= 0:
            raise RuntimeError("cpu pool: thread creation failed")
        passes = int(min(64, max(2, seconds / (dt / 2))))
        dt = pool.sea_cpu_pool_run(fn, pin, pout, lens, n, threads, passes)
    return passes * frames / dt, passes, outs


def cpu_quota():
    """The CPU time this process may use per second as the cgroup says it (cpu.max), or None: a box can show 256 CPUs
    in the affinity mask and still be held to a share of them."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                t = f.read().split()
            if path.endswith("cpu.max"):
                return None if t[0] == "max" else float(t[0]) / float(t[1])
            q = float(t[0])
            if q <= 0:
                return None
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                return q / float(f.read())
        except (OSError, ValueError, IndexError):
            continue
    return None


def cpu_model():
    """The host CPU as /proc/cpuinfo names it, with the socket / core counts the box shows."""
    model, phys, cores = "unknown", set(), 0
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys.add(line.split(":", 1)[1].strip())
                elif line.startswith("processor"):
                    cores += 1
    except OSError:
        pass
    return f"{model} ({cores} logical CPUs on {max(1, len(phys))} socket(s); {len(os.sched_getaffinity(0))} usable by this process)"


def cpu_baseline(batch, n_sample, threads):
    """kind "port": the oracle's C restatement, one utterance per thread over ALL the host cores this process may run
    on (len(os.sched_getaffinity(0)); --cpu-threads overrides), on the first n_sample utterances of the shard; the
    16-thread figure of rounds 1-3 is kept beside it.  Two builds, as BASELINE.md section 3 promised: the parity build
    (-O2, no FMA contraction: oracle/libsea_oracle.so) and a speed build compiled HERE for this host (-O3
    -march=native); `value` is the faster.  Where oracle/_ref (the reference C compiled from its own sources) shipped
    with the tree it is timed the same way and reported beside as `reference_value` (etsi/cpp/AdvFrontEnd.c:125-210
    per thread)."""
    import ctypes
    from oracle import oracle as O
    host = batch.data.cpu().numpy()
    n_sample = min(n_sample, batch.n_utt)
    utts = [np.ascontiguousarray(host[o:o + l]) for o, l in
            zip(batch.host_offsets[:n_sample], batch.host_lengths[:n_sample])]
    usable = len(os.sched_getaffinity(0))
    cores = max(1, min(threads, usable)) if threads > 0 else usable
    side = 16 if cores > 16 else 0   # the figure of rounds 1-3, beside
    parity = O.Oracle()
    v, passes, outs = _time_cpu(parity.lib, "ora_etsi_denoise", utts, cores)
    res = dict(unit="frames/s", cores=cores, kind="port", parity_build_value=v,
               parity_build="gcc -O2 -ffp-contract=off (oracle/libsea_oracle.so)")
    res["cpu_model"] = cpu_model()
    quota = cpu_quota()
    res["cgroup_cpu_quota"] = quota if quota is not None else "none (cpu.max = max)"
    if side:
        res["threads16"] = {"parity_build_value": _time_cpu(parity.lib, "ora_etsi_denoise", utts, side)[0]}
    best = v
    tmp = tempfile.mkdtemp(prefix="sea_cpu_")
    try:  # the speed build must be made on the box it runs on (-march=native)
        so = os.path.join(tmp, "libsea_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-o", so,
                               os.path.join(ROOT, "oracle", "ns_oracle.c"), os.path.join(ROOT, "oracle", "resynth_oracle.c"),
                               "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        speed = ctypes.CDLL(so)
        res["speed_build_value"] = _time_cpu(speed, "ora_etsi_denoise", utts, cores)[0]
        res["speed_build"] = "gcc -O3 -march=native (FMA allowed: not bit-exact, timing only)"
        if side:
            res["threads16"]["speed_build_value"] = _time_cpu(speed, "ora_etsi_denoise", utts, side)[0]
        best = max(best, res["speed_build_value"])
    except Exception as e:  # no compiler on the box: say so, keep the parity build
        res["speed_build"] = f"unavailable ({type(e).__name__})"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)  # a loaded mapping stays valid after the file is gone
    if O.have_reference():
        ref = O.Reference()
        res["reference_value"] = _time_cpu(ref.lib, "ref_etsi_denoise", utts, cores)[0]
        res["reference_build"] = "the reference's etsi/cpp/*.c, gcc -O2 -ffp-contract=off (oracle/_ref)"
        if side:
            res["threads16"]["reference_value"] = _time_cpu(ref.lib, "ref_etsi_denoise", utts, side)[0]
    # A box can show every CPU in the affinity mask and still hold the process to a share of them (cgroup cpu.max: 16 CPUs'
    # worth on the 1-GPU box): 256 runnable threads under a 16-CPU quota are throttled and SLOWER than 16.  The baseline is
    # what the host can do for this process: the best figure over both thread counts and both builds; `cores` names the
    # thread count it was reached with, `all_cores` / `threads16` keep every measurement.
    res["all_cores"] = {"threads": cores, "parity_build_value": res["parity_build_value"],
                        "speed_build_value": res.get("speed_build_value"), "reference_value": res.get("reference_value")}
    if side:
        best16 = max(v for v in res["threads16"].values() if v)
        res["threads16"]["threads"] = side
        if best16 > best:
            best, res["cores"] = best16, side
            for k in ("parity_build_value", "speed_build_value", "reference_value"):
                if res["threads16"].get(k):
                    res[k] = res["threads16"][k]
    res["value"] = best
    frames = int(sum(len(u) // 80 for u in utts))
    res["sample"] = (f"the first {len(utts)} utterances of the shard ({frames} frames per pass), {passes}+ passes per build "
                     f"(one pass to warm up, two for an estimate, then ~2 s worth), utterances pulled from a shared counter "
                     f"(oracle/cpu_pool.c), on {usable} threads = every CPU this process may run on (all_cores) and on 16 "
                     f"(threads16, the figure of rounds 1-3); value = the fastest of the port's builds over both, reached on "
                     f"{res['cores']} threads; cgroup_cpu_quota = the CPUs' worth of time the box grants this process")
    return res, outs


# ------------------------------------------------------------------------------------------------
def source_stamp():
    """sha256 over the kernel sources: profiles/pmc_traffic.json is stamped with it so that a counter
    figure measured on an older kernel is not reported for a newer one."""
    h = hashlib.sha256()
    for fn in sorted(os.listdir(CSRC)):
        if fn.endswith((".hip", ".h", ".c")) or fn == "Makefile":
            with open(os.path.join(CSRC, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(key):
    """(bytes per launch, note) from profiles/pmc_traffic.json, None when absent or stale."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, "profiles/pmc_traffic.json absent"
    with open(path) as f:
        j = json.load(f)
    if j.get("source_stamp") != source_stamp():
        return None, (f"profiles/pmc_traffic.json was measured at kernel sources {j.get('source_stamp')}, "
                      f"this tree is {source_stamp()}: stale, not reported (tools/profile_round.sh refreshes it)")
    return j.get(key), f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at kernel sources {j['source_stamp']}"


def timed_steps(fn, steps, warmup):
    """HIP-event time per call of fn() on torch's current stream (the stream every launch of
    speech_enhancement_amd goes to); returns (mean seconds, wall seconds per step)."""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    return float(np.mean([a.elapsed_time(b) for a, b in ev])) / 1e3, wall


def also_lines(batch, ids, device, steps):
    """configs[2], configs[3], CompCeps and rfft256 on the same shard, same run."""
    import torch
    import speech_enhancement_amd as sea
    out = []
    audio_s = float(np.sum(batch.host_lengths)) / 16000.0

    def line(name, workload, units, unit, alg_bytes, ker_s, wall_s, kernel, traffic_key, extra=None):
        traffic, note = pmc_traffic(traffic_key)
        d = {"name": name, "workload": workload, "value": units / wall_s, "unit": unit, "ms_per_step": wall_s * 1e3,
             "steps": steps, "rtf": wall_s / audio_s,
             "roofline": {"bound": "hbm", "kernel": kernel, "achieved": alg_bytes / ker_s / 1e9, "peak": HBM_PEAK_GBPS,
                          "unit": "GB/s", "frac": alg_bytes / ker_s / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                          "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                          "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": ker_s * 1e3}}
        if extra:
            d.update(extra)
        out.append(d)

    # resynth, ratio mask and ideal binary mask (configs[2], configs[3])
    masks = build_masks(batch, ids, device)
    hops = int(np.sum((np.asarray(batch.host_lengths) - 320) // 160 + 1))
    scratch = torch.empty(sea.resynth_scratch_elems(batch), dtype=torch.float32, device=device)
    rs_out = torch.zeros_like(batch.data)
    for name, binary, cfg in (("resynth_64sub_ori", False, 2), ("resynth_64sub_IBM", True, 3)):
        ker, wall = timed_steps(lambda: sea.resynth_batch(batch, masks, binary=binary, out=rs_out, scratch=scratch), steps, 1)
        line(name, f"BASELINE configs[{cfg}]: {batch.n_utt} utterances, 64-band gammatone analysis/synthesis, "
                   f"{'ideal binary' if binary else 'ratio'} mask, {hops} hop-frames of 160 samples",
             hops, "hop-frames/s", hops * RS_BYTES_PER_HOP, ker, wall, "sea::resynth_fused_kernel",
             "resynth_bytes_per_launch",
             {"intermediate_bytes_per_launch": int(batch.total) * 64 * 4 * 2})
    del scratch, rs_out, masks

    # CompCeps from the float NoiseSup stream
    _, f32, first = sea.ns_denoise_batch(batch, want_f32=True)
    torch.cuda.synchronize()
    res = {}

    def run_cc():
        res["c"] = sea.compceps_batch(batch, f32, first)
    ker, wall = timed_steps(run_cc, steps, 1)
    n = int(res["c"][2].sum().item())
    line("CompCeps", f"{batch.n_utt} utterances, {n} cepstral frames (window 200, hop 80) of 14 coefficients from the float NoiseSup stream",
         n, "frames/s", n * CC_BYTES_PER_FRAME, ker, wall, "sea::compceps_kernel", "compceps_bytes_per_launch")
    del f32, res

    # subbband() (gammatone + hair cell -> 64 int16 streams) and the IRM target computed from
    # two DIFFERENT such blocks (clean = this corpus, noise = the same layout with the samples in reverse order: passing
    # one block twice would let half of the reads hit in cache and overstate the rate)
    sub = torch.zeros(batch.total * 64, dtype=torch.int16, device=device)
    sub2 = torch.zeros(batch.total * 64, dtype=torch.int16, device=device)
    keep = batch.data
    batch.data = torch.flip(keep, dims=[0]).contiguous()
    sea.subband_batch(batch, out=sub2)
    torch.cuda.synchronize()
    batch.data = keep
    ker, wall = timed_steps(lambda: sea.subband_batch(batch, out=sub), max(2, steps // 2), 1)
    samples = int(np.sum(batch.host_lengths))
    line("subbband", f"SURVEY 8(f) #1: {batch.n_utt} utterances, 64-channel gammatone + Meddis hair cell to 64 int16 streams "
                     "(2 B read + 128 B written per sample)", samples, "samples/s", samples * 130, ker, wall,
         "sea::subband_kernel", "subband_bytes_per_launch")
    ker, wall = timed_steps(lambda: sea.irm_target_batch(batch, sub, sub2), max(2, steps // 2), 1)
    line("IRM target", f"SURVEY 8(f) #2: make_single_IBM's ratio-mask target from two different subband blocks, {hops} frames x 64 channels "
                       "(2 x 128 B x 160 samples read + 256 B written per frame)", hops, "hop-frames/s",
         hops * (2 * 128 * 160 + 256), ker, wall, "sea::irm_target_kernel", "irm_bytes_per_launch")
    del sub, sub2

    # the restored feature chain (NoiseSup with speech flags -> WaveProc -> CompCeps -> PostProc -> VAD),
    # three launches on device-resident buffers (the C ABI directly: engine.afe_features_batch also unpacks on the host)
    lib = sea.load()
    n = batch.n_utt
    outb = torch.zeros_like(batch.data)
    f32 = torch.zeros(batch.total, dtype=torch.float32, device=device)
    first = torch.full((n,), -1, dtype=torch.int32, device=device)
    onset = torch.zeros(n, dtype=torch.int32, device=device)
    flags = torch.zeros(batch.total // 8, dtype=torch.uint8, device=device)
    nfr = np.asarray(batch.host_lengths) // 80
    ccum = np.concatenate(([0], np.cumsum(np.maximum(nfr - 6, 0)))).astype(np.int64)
    fcum = np.concatenate(([0], np.cumsum(nfr + 6))).astype(np.int64)
    tc, tf = int(ccum[-1]), int(fcum[-1])
    fcc = torch.zeros((tc, 14), dtype=torch.float32, device=device)
    f15 = torch.zeros((tf, 15), dtype=torch.float32, device=device)
    nfe = torch.zeros(n, dtype=torch.int32, device=device)
    d_ccum, d_fcum = torch.from_numpy(ccum).to(device), torch.from_numpy(fcum).to(device)
    st = torch.cuda.current_stream().cuda_stream

    def run_afe():
        P = lambda t: t.data_ptr()
        assert lib.sea_ns_denoise_batch_fd(P(batch.data), P(outb), P(f32), P(batch.offsets), P(batch.lengths), P(batch.order),
                                           P(first), P(flags), P(onset), n, st) == 0
        assert lib.sea_afe_features_batch(P(f32), P(flags), P(batch.offsets), P(batch.lengths), P(first), P(onset), P(d_ccum),
                                          tc, P(fcc), None, P(d_fcum), P(f15), P(nfe), None, n, st) == 0
    ker, wall = timed_steps(run_afe, max(2, steps // 2), 1)
    emitted = int(nfe.sum().item())
    line("AFE feature chain", f"SURVEY 8(f) #3: {n} utterances, {emitted} emitted feature frames of 15 floats, three launches "
                              "(NoiseSup with speech flags | WaveProc + CompCeps | PostProc + VAD + flush)", emitted,
         "feature frames/s", batch.n_frames * 320 + emitted * 60, ker, wall,
         "sea::ns_denoise_pipe_fd_kernel + sea::afe_ceps_kernel + sea::afe_vad_kernel", "afe_bytes_per_launch")
    del outb, f32, flags, fcc, f15

    # the 16 k-native NoiseSup variant behind etsi_denoise_mapping_* (round 4: four pipelined waves per stream,
    # two streams per workgroup), one stream per utterance of the shard, 400 frames of 160 int16-valued samples each
    B, nf = batch.n_utt, 400
    gen = torch.Generator(device=device)
    gen.manual_seed(16)
    fr = torch.randint(-6000, 6000, (B, nf, 160), device=device, generator=gen).float()
    o16 = torch.zeros_like(fr)
    pr16 = torch.zeros((B, nf), dtype=torch.int32, device=device)
    fl16 = torch.zeros((B, nf), dtype=torch.uint8, device=device)
    ct16 = torch.zeros((B, nf), dtype=torch.int32, device=device)
    w16 = torch.zeros((B, nf, 25), dtype=torch.float32, device=device)
    st16 = torch.zeros((B, lib.sea_ns16k_state_floats()), dtype=torch.float32, device=device)

    def run16():
        assert lib.sea_ns16k_streams_push(fr.data_ptr(), o16.data_ptr(), pr16.data_ptr(), fl16.data_ptr(), ct16.data_ptr(),
                                          w16.data_ptr(), st16.data_ptr(), B, nf, 1, st) == 0
    ker, wall = timed_steps(run16, max(2, steps // 2), 1)
    assert int(pr16.sum().item()) == B * (nf - 4)
    # the same samples as twice the streams of half the length (two rounds of workgroups instead of one)
    B2, nf2 = 2 * B, nf // 2

    def run16b():
        assert lib.sea_ns16k_streams_push(fr.data_ptr(), o16.data_ptr(), pr16.data_ptr(), fl16.data_ptr(), ct16.data_ptr(),
                                          w16.data_ptr(), st16b.data_ptr(), B2, nf2, 1, st) == 0
    st16b = torch.zeros((B2, lib.sea_ns16k_state_floats()), dtype=torch.float32, device=device)
    ker2, wall2 = timed_steps(run16b, max(2, steps // 2), 1)
    line("NoiseSup, 16 k-native variant", f"SURVEY 8(f) #4: {B} streams x {nf} frames of 160 samples through the variant behind "
         "etsi_denoise_mapping_* (window 480, rfft (x, 512, 8), 25 gammatone-shaped windows), four pipelined wavefronts per stream",
         B * nf, "frames/s", B * nf * (2 * 640 + 100 + 9), ker, wall, "sea::ns16k_pipe_kernel", "ns16k_bytes_per_launch",
         {"rtf": wall / (B * nf * 160 / 16000.0),
          "as_twice_the_streams_of_half_the_length": {"streams": B2, "frames": nf2, "value": B2 * nf2 / wall2, "unit": "frames/s",
                                                      "ms_per_step": wall2 * 1e3, "avg_launch_ms": ker2 * 1e3}})
    del fr, o16, w16, st16b

    # rfft256 on a streaming batch: 2^21 frames = 2 GiB in + 2 GiB out, 16 x the 256 MiB Infinity Cache (the HBM
    # figure), and 2^18 frames = 256 MiB + 256 MiB, which partly lives in that cache (kept for comparison with round 2)
    lib = sea.load()
    st = torch.cuda.current_stream().cuda_stream
    for nfr, name, key in ((1 << 21, "rfft256", "rfft256_2g_bytes_per_launch"), (1 << 18, "rfft256 (cache-sized)", "rfft256_bytes_per_launch")):
        x = torch.randn(nfr, 256, device=device)
        y = torch.empty_like(x)

        def run_fft():
            assert lib.sea_rfft256_batch(x.data_ptr(), y.data_ptr(), nfr, st) == 0
        ker, wall = timed_steps(run_fft, steps, 1)
        line(name, f"{nfr} frames of 256 floats (etsi/cpp/rfft.c), out of place, {nfr * 2048 / 2**30:.2f} GiB moved per launch",
             nfr, "frames/s", nfr * FFT_BYTES_PER_FRAME, ker, wall, "sea::rfft256_kernel", key)
        del x, y
    return out


def configs4_line(device, steps):
    """BASELINE configs[4] on one GPU: LPT shard 0 of the 100 000-utterance corpus (12 500 utterances, ~10 M frames),
    built on the device like the headline shard, one launch per step; the first and last 32 utterances of the shard
    are compared with the CPU oracle exactly."""
    import torch
    import speech_enhancement_amd as sea
    ids = corpus_shard_ids(100000, 8, 0)
    batch = build_shard_ids(ids, device)
    out = torch.zeros_like(batch.data)
    ker, wall = timed_steps(lambda: sea.ns_denoise_batch(batch, out=out), steps, 1)
    frames = batch.n_frames
    alg = frames * NS_BYTES_PER_FRAME
    from oracle import oracle as O
    ora = O.Oracle()
    pick = list(range(32)) + list(range(batch.n_utt - 32, batch.n_utt))
    host_in, host_out = batch.data.cpu().numpy(), out.cpu().numpy()
    bad = 0
    with quiet_stderr():
        for u in pick:
            o, l = int(batch.host_offsets[u]), int(batch.host_lengths[u])
            want = ora.etsi_denoise(np.ascontiguousarray(host_in[o:o + l]))
            bad += int(not np.array_equal(host_out[o:o + l // 80 * 80], want[:l // 80 * 80]))
    traffic, tnote = pmc_traffic("ns_big_bytes_per_launch")
    insts, _ = pmc_traffic("ns_big_valu_insts_per_launch")
    clk, _ = pmc_traffic("ns_big_valu_issue_clk_per_launch")
    d = {"name": "NoiseSup, configs[4] shard", "workload": f"BASELINE configs[4]: shard 0 of 8 (LPT by samples) of the 100000-utterance "
         f"corpus = {batch.n_utt} utterances, {frames} frames, one launch", "value": frames / wall, "unit": "frames/s",
         "ms_per_step": wall * 1e3, "steps": steps, "rtf": wall / (frames * 80 / 16000.0),
         "parity_check": f"{len(pick) - bad}/{len(pick)} sampled utterances bit-identical to the CPU oracle",
         "roofline": {"bound": "hbm", "kernel": "sea::ns_denoise_pipe_big_kernel", "achieved": alg / ker / 1e9, "peak": HBM_PEAK_GBPS,
                      "unit": "GB/s", "frac": alg / ker / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": tnote,
                      "traffic_over_algorithmic": (traffic / alg) if traffic else None,
                      "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ker * 1e3}}
    if insts:  # the big form's own counter passes (tools/profile_round.sh, same shard)
        d["roofline"]["valu_insts_per_frame"] = insts / frames
        if clk:
            d["roofline"]["valu_issue_ms"] = clk / (1024 * 2.4e9) * 1e3
            d["roofline"]["valu_issue_frac"] = clk / (1024 * 2.4e9) / ker
    del batch, out
    return d, bad


def configs4_whole(world, rank, device, dist, steps, corpus_utts=100000):
    """BASELINE configs[4] as BASELINE.json defines it for N GPUs: the WHOLE 100 000-utterance corpus over the N ranks of
    this job -- cut into N shards balanced by samples (LPT), rank r holds shard r in HBM and runs it as one launch per
    step, no collective on the data path -- so that the driver's N = 1, 2, 4, 8 runs of this script are a STRONG-scaling
    curve of one fixed corpus (the reference's shape: the shared-counter pool over files,
    aurora_speech_enhancement.cpp:311-327).  Barrier + synchronize on both sides; value = all frames / max over ranks."""
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd.shard import reduce_job
    ids = corpus_shard_ids(corpus_utts, world, rank)
    batch = build_shard_ids(ids, device)
    out = torch.zeros_like(batch.data)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    sea.ns_denoise_batch(batch, out=out)   # warm-up
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sea.ns_denoise_batch(batch, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    total, dt_max = reduce_job(batch.n_frames * steps, dt, dist if world > 1 else None)
    # size-independent property on every utterance of the shard: nothing before frame 4 of the first
    # non-silent frame; checked exactly against the oracle at N = 1 by configs4_line / the full-size tests
    d = {"name": "NoiseSup, configs[4] whole corpus", "value": total / dt_max, "unit": "frames/s", "n_gpus": world,
         "scaling": "strong", "steps": steps, "ms_per_step": dt_max / steps * 1e3,
         "rtf": dt_max / (total * 80 / 16000.0),
         "workload": f"BASELINE configs[4]: the whole {corpus_utts}-utterance synthetic corpus ({total // steps} frames) cut into "
                     f"{world} LPT shard(s), one per GPU, {batch.n_utt} utterances on rank 0, one launch per step and GPU",
         "frac_hbm": total / dt_max * NS_BYTES_PER_FRAME / 1e9 / (HBM_PEAK_GBPS * world)}
    del batch, out
    return d


def cli_wall_clock(ins):
    """The file driver end to end (etsi/deal.sh shape: ./etsi_denoise <cfg>): one WAV per utterance of the shard on
    disk in, one WAV each out; wall clock of the whole process incl. start-up, HIP initialisation, file I/O."""
    exe = os.path.join(ROOT, "speech_enhancement_amd", "host", "bin", "etsi_denoise")
    if not os.path.exists(exe):
        return {"value": None, "what": "speech_enhancement_amd/host/bin/etsi_denoise not built"}
    import struct
    tmp = tempfile.mkdtemp(prefix="sea_cli_")
    try:
        os.makedirs(os.path.join(tmp, "noisy"))
        os.makedirs(os.path.join(tmp, "out"))
        ids = [f"U{k:05d}" for k in range(len(ins))]
        for i, x in zip(ids, ins):
            body = (b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) + b"data"
                    + struct.pack("<I", 2 * len(x)) + x.astype("<i2").tobytes())
            with open(os.path.join(tmp, "noisy", i + "_noisy.wav"), "wb") as f:
                f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        with open(os.path.join(tmp, "list.txt"), "w") as f:
            f.write("".join(i + "\n" for i in ids))
        with open(os.path.join(tmp, "cfg.txt"), "w") as f:
            f.write("".join(l + "\n" for l in ["purewavDictionary= /nowhere/", f"purewavlist= {tmp}/list.txt", "numMix= 1",
                                               f"outputDictionary= {tmp}/", "save_noisy_dir= noisy/", "save_noisy_ebm_dir= ebm/",
                                               "save_noisy_sirm_dir= sirm/", "save_resynth_e_dir= out/", "save_resynth_i_dir= i/",
                                               "Log= run.log"]))
        t0 = time.perf_counter()
        r = subprocess.run([exe, os.path.join(tmp, "cfg.txt")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"value": None, "what": f"etsi_denoise <cfg> failed ({r.returncode}): {r.stderr[-200:]}"}
        n_out = len(os.listdir(os.path.join(tmp, "out")))
        frames = int(sum(len(x) // 80 for x in ins))
        return {"value": frames / wall, "unit": "frames/s", "seconds": wall, "files_written": n_out,
                "what": f"host/bin/etsi_denoise <cfg> on {len(ins)} WAV files (reader threads | device thread | writer "
                        "threads), whole process incl. start-up and HIP initialisation"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def end_to_end(batch, steps):
    """SURVEY 8(d) 'GPU timing': wall clock INCLUDING the PCIe copies, through the host-buffer drop-ins the
    reference's callers use (etsi/cpp/main.cpp:43-67, aurora_speech_enhancement.cpp:25-80).  Never the headline."""
    import ctypes
    import speech_enhancement_amd as sea
    lib = sea.load()
    host = batch.data.cpu().numpy()
    ins = [np.ascontiguousarray(host[o:o + l]) for o, l in zip(batch.host_offsets, batch.host_lengths)]
    outs = [np.zeros_like(x) for x in ins]
    n = len(ins)
    pin = (ctypes.c_void_p * n)(*[x.ctypes.data for x in ins])
    pout = (ctypes.c_void_p * n)(*[x.ctypes.data for x in outs])
    lens = (ctypes.c_long * n)(*[x.size for x in ins])
    with quiet_stderr():
        for _ in range(2):  # staging buffers, streams and the pool's threads come into being; the callers' pages get touched
            assert lib.sea_denoise_utterances(pin, pout, lens, n) == 0, lib.sea_last_error()
        per = []
        for _ in range(max(steps, 8)):
            ta = time.perf_counter()
            assert lib.sea_denoise_utterances(pin, pout, lens, n) == 0
            per.append(time.perf_counter() - ta)
        wall = float(np.median(per))
        # the packed-pinned form: the caller's samples already sit where the copy engine reads them (sea_packed_*): the run
        # below is what a file driver's device thread pays per list -- uploads, launches, downloads, no pack / unpack copies
        pk = lib.sea_packed_create()
        per_pk = []
        assert lib.sea_packed_plan(pk, lens, n) == 0, lib.sea_last_error()
        K = lib.sea_packed_slices(pk)
        for u, x in enumerate(ins):
            pi, po, cnt = (ctypes.c_void_p * K)(), (ctypes.c_void_p * K)(), (ctypes.c_long * K)()
            k = lib.sea_packed_segments(pk, u, pi, po, cnt, K)
            pos = 0
            for i in range(k):
                ctypes.memmove(pi[i], x.ctypes.data + 2 * pos, 2 * cnt[i])
                pos += cnt[i]
        for _ in range(2):
            assert lib.sea_packed_denoise(pk) == 0, lib.sea_last_error()
        for _ in range(max(steps, 8)):
            ta = time.perf_counter()
            assert lib.sea_packed_denoise(pk) == 0
            per_pk.append(time.perf_counter() - ta)
        wall_pk = float(np.median(per_pk))
        # the last utterance's pieces against what sea_denoise_utterances left for it (same bits)
        k = lib.sea_packed_segments(pk, n - 1, pi, po, cnt, K)
        got = np.concatenate([np.ctypeslib.as_array(ctypes.cast(po[i], ctypes.POINTER(ctypes.c_short)), shape=(cnt[i],)) for i in range(k)])
        packed_same = bool(np.array_equal(got, outs[n - 1][:got.size]))
        lib.sea_packed_destroy(pk)
        # the reference's own calling pattern: one etsi_denoise(short*, short*, long) per utterance
        k = min(n, 64)
        fr1 = int(sum(x.size // 80 for x in ins[:k]))
        lib.etsi_denoise(ins[0].ctypes.data, outs[0].ctypes.data, ins[0].size)
        t1 = time.perf_counter()
        for x, y in zip(ins[:k], outs[:k]):
            assert lib.etsi_denoise(x.ctypes.data, y.ctypes.data, x.size) == 0
        per_call = (time.perf_counter() - t1) / k
        # the FEParamsX plug-in slot: one DoNoiseSup call per 80-sample frame of ONE stream
        ns = sea.NoiseSup()
        x = ins[0][:80 * 2000].astype(np.float32)
        ns.DoNoiseSup(x[:80])
        t2 = time.perf_counter()
        for f in range(1, len(x) // 80):
            ns.DoNoiseSup(x[80 * f:80 * f + 80])
        per_frame = (time.perf_counter() - t2) / (len(x) // 80 - 1)
    cli = cli_wall_clock(ins)
    return {"note": "wall clock including pack, H2D, launch, D2H and unpack; never the headline value",
            "cli_files": cli,
            "denoise_utterances": {"value": batch.n_frames / wall, "unit": "frames/s", "ms_per_call": wall * 1e3,
                                   "ms_per_call_stat": f"median of {len(per)} calls after 2 warm-up calls",
                                   "ms_per_call_min_mean_max": [round(min(per) * 1e3, 3), round(float(np.mean(per)) * 1e3, 3), round(max(per) * 1e3, 3)],
                                   "what": f"sea_denoise_utterances on the {n} utterances of this shard in host memory, "
                                           f"{lib.sea_host_threads()} packing threads"},
            "denoise_packed": {"value": batch.n_frames / wall_pk, "unit": "frames/s", "ms_per_call": wall_pk * 1e3,
                               "same_bits_as_denoise_utterances": packed_same,
                               "what": f"sea_packed_denoise on the same {n} utterances written into the library's pinned staging by the "
                                       f"caller ({K} time slices): uploads, launches and downloads pipelined, no pack / unpack copies; "
                                       f"median of {len(per_pk)} calls"},
            "etsi_denoise_per_utterance": {"value": fr1 / (per_call * k), "unit": "frames/s", "ms_per_call": per_call * 1e3,
                                           "what": f"{k} sequential etsi_denoise() calls, mean {fr1 / k:.0f} frames each"},
            "ns_stream_push": {"value": per_frame * 1e6, "unit": "us per 80-sample frame",
                               "what": "sea_ns_stream_push (the DoNoiseSup plug-in slot) on one stream, through ctypes; "
                                       "the reference CPU needs ~12 us per frame (BASELINE.md section 2)"}}


def rehearse_cpu(args, world, rank):
    """The N > 1 control path without a GPU: every rank takes its shard of the corpus (ids only), the
    ranks meet at the gloo barrier and reduce (sum of frames, max of seconds) as the real run does."""
    import torch.distributed as dist
    from speech_enhancement_amd import corpus
    from speech_enhancement_amd.shard import reduce_job
    if world > 1:
        init_gloo(rank, world)
    if os.environ.get("SEA_BENCH_REHEARSE_FAIL_RANK") == str(rank):   # test hook: a rank that dies after the rendezvous
        os._exit(7)
    if args.corpus_utts > 0 and world > args.shards:
        raise SystemExit(f"--corpus-utts: {world} ranks but only {args.shards} shards (each shard is run once; use --shards {world})")
    ids = (corpus_shard_ids(args.corpus_utts, args.shards, rank) if args.corpus_utts > 0
           else list(range(rank * args.utts, (rank + 1) * args.utts)))
    frames = int(sum(corpus.utterance_length(int(u)) // 80 for u in ids))
    if world > 1:
        dist.barrier()
    total, tmax = reduce_job(frames * args.steps, 1.0 + rank, dist if world > 1 else None)
    c4 = None
    if args.corpus_utts == 0 and not args.no_configs4:   # the `configs4` object: the whole corpus over the ranks of the job
        ids4 = corpus_shard_ids(args.configs4_utts, world, rank)
        fr4 = int(sum(corpus.utterance_length(int(u)) // 80 for u in ids4))
        if world > 1:
            dist.barrier()
        tot4, t4 = reduce_job(fr4 * 2, 2.0 + rank, dist if world > 1 else None)
        c4 = {"n_gpus": world, "scaling": "strong", "total_frames": tot4, "seconds_max": t4, "utterances_rank0": len(ids4),
              "frames_rank0": fr4}
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "steps": args.steps, "total_frames": total,
                          "seconds_max": tmax, "utterances_rank0": len(ids), "first_ids_rank0": [int(u) for u in ids[:4]],
                          "configs4": c4}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_cpu:
        return rehearse_cpu(args, world, rank)
    import torch
    import torch.distributed as dist
    import speech_enhancement_amd as sea

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev  # ranks > devices only when rehearsing N ranks on a smaller box
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        init_gloo(rank, world)

    sea.load().sea_init(-1)
    if args.corpus_utts > 0:
        if world > args.shards:
            raise SystemExit(f"--corpus-utts: {world} ranks but only {args.shards} shards (each shard is run once; use --shards {world})")
        ids = corpus_shard_ids(args.corpus_utts, args.shards, rank)
        workload = (f"BASELINE configs[4]: {args.corpus_utts}-utterance synthetic corpus in {args.shards} LPT shards "
                    f"(balanced by samples), one shard per GPU; this run: {world} of {args.shards} shard(s), "
                    f"{len(ids)} utterances on rank 0")
    else:
        ids = list(range(rank * args.utts, (rank + 1) * args.utts))
        workload = (f"BASELINE configs[1]: {args.utts}-utterance batch per GPU, 256-pt rfft + two-stage Wiener "
                    "NoiseSup (etsi_denoise semantics), SURVEY 8(d) synthetic 16 kHz corpus, 2-6 s utterances")
    batch = build_shard_ids(ids, device)
    out = torch.zeros_like(batch.data)
    frames_per_step = batch.n_frames
    audio_s_per_step = frames_per_step * 80 / 16000.0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        sea.ns_denoise_batch(batch, out=out)
    barrier()

    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        events[k][0].record()
        sea.ns_denoise_batch(batch, out=out)
        events[k][1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0   # this rank's timed region; the job's is the max over ranks
    barrier()
    kernel_ms = [a.elapsed_time(b) for a, b in events]

    # whole-job aggregate: frames summed over ranks, time = max over ranks (no data-path collective)
    from speech_enhancement_amd.shard import reduce_job
    total_frames, dt_max = reduce_job(frames_per_step * args.steps, dt, dist if world > 1 else None)
    value = total_frames / dt_max

    result, rc = None, 0
    if rank == 0:
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        achieved = frames_per_step * NS_BYTES_PER_FRAME / avg_kernel_s / 1e9
        traffic, traffic_note = (pmc_traffic("ns_denoise_kernel_bytes_per_launch")
                                 if (args.corpus_utts == 0 and args.utts == UTTS_PER_GPU) else (None, "measured for configs[1] only"))
        forced = os.environ.get("SEA_NS_KERNEL")
        per_cu = batch.n_utt / 256.0
        kernel = {"single": "sea::ns_denoise_kernel", "pipe": "sea::ns_denoise_pipe_kernel", "pipe6": "sea::ns_denoise_pipe6_kernel",
                  "pipe6d": "sea::ns_denoise_pipe6_dense_kernel", "big": "sea::ns_denoise_pipe_big_kernel"}.get(forced) or (
            "sea::ns_denoise_pipe6_kernel" if per_cu <= 3 else ("sea::ns_denoise_pipe6_dense_kernel" if per_cu <= 4 else "sea::ns_denoise_pipe_big_kernel"))
        result = {
            "metric": "NoiseSup frames/sec (16 kHz, hop 80, 256-pt rfft, two-stage Wiener), batched",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.corpus_utts > 0 else "weak",  # configs[4] is ONE fixed corpus cut into shards
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "rtf": dt_max / (total_frames * 80 / 16000.0),
            "config": {
                "workload": workload,
                "utterances_per_gpu": batch.n_utt,
                "frames_per_step_per_gpu": frames_per_step,
                "audio_seconds_per_step_per_gpu": audio_s_per_step,
                "sharding": f"{world} independent shard(s), no collective on the data path (gloo barrier + 2 scalars only)",
                "devices_visible": n_dev,
            },
            "roofline": {
                "bound": "hbm",
                "limiter": "vector-instruction issue and dependent latency, not HBM (DESIGN.md 5.1); frac is the HBM fraction BASELINE asks for",
                "kernel": kernel,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": frames_per_step * NS_BYTES_PER_FRAME,
                "avg_launch_ms": avg_kernel_s * 1e3,
            },
        }
        # what actually bounds the kernel: vector-instruction issue.  Counters from the source-stamped profile passes,
        # priced with tools/valu_probe.hip's issue costs (f32 2.1 clk per wave64 instruction, packed / f64 4.2,
        # transcendental 8.4) on 1024 SIMDs at the 2.4 GHz peak engine clock.
        insts, _ = (pmc_traffic("ns_valu_insts_per_launch") if (args.corpus_utts == 0 and args.utts == UTTS_PER_GPU) else (None, ""))
        clk, _ = (pmc_traffic("ns_valu_issue_clk_per_launch") if insts else (None, ""))
        if insts:
            result["roofline"]["valu_insts_per_frame"] = insts / frames_per_step
            if clk:
                issue_s = clk / (1024 * 2.4e9)
                result["roofline"]["valu_issue_ms"] = issue_s * 1e3
                result["roofline"]["valu_issue_frac"] = issue_s / avg_kernel_s
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N=1 only
            base, outs = cpu_baseline(batch, args.cpu_utts, args.cpu_threads)
            result["cpu_baseline"] = base
            # parity on the sample, exact (the oracle as checker, never as the measured path)
            got = batch.split(out, full_frames_only=True)
            bad = sum(int(not np.array_equal(g, o[:len(g)])) for g, o in zip(got[:len(outs)], outs))
            result["parity_check"] = f"{len(outs) - bad}/{len(outs)} sampled utterances bit-identical to the CPU oracle"
            result["gpu_over_cpu"] = value / base["value"]
            if bad:
                rc = 3
        if not args.no_also and world == 1:
            del out
            if args.corpus_utts == 0 and not args.no_end_to_end:
                result["end_to_end"] = end_to_end(batch, 5)
            result["also"] = also_lines(batch, ids, device, args.also_steps)
            if args.corpus_utts == 0 and not args.no_configs4:
                del batch
                torch.cuda.empty_cache()
                c4, bad4 = configs4_line(device, 3)
                result["also"].insert(0, c4)
                if bad4:
                    rc = 3
    if args.corpus_utts == 0 and not args.no_configs4 and not args.no_also:
        # every rank takes part (N > 1: the also-array of the other kernels is an N = 1 matter, this line is not)
        batch = out = None   # this rank's configs[1] shard leaves the HBM
        torch.cuda.empty_cache()
        c4w = configs4_whole(world, rank, device, dist if world > 1 else None, 2, args.configs4_utts)
        if rank == 0:
            result["configs4"] = c4w
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return rc


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: start the ranks ourselves, from a process that never touches the GPU
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    sys.exit(run_rank(args))


if __name__ == "__main__":
    main()
