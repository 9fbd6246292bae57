#!/usr/bin/env python3
"""bench.py -- headline measurement of the NoiseSup hot path on MI355X.

One "step" = one pass of the hot path over one batch: etsi_denoise semantics (two-stage Wiener
NoiseSup incl. both 256-point rffts, int16 in -> int16 out) for every utterance of BASELINE.json's
configs[1] workload, the 1024-utterance synthetic 16 kHz corpus of SURVEY.md 8(d), resident in HBM
when the timed region starts.  Unit of throughput: NoiseSup frames (80 samples) per second.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Utterances are independent, so ranks shard the corpus with NO data-path collective (weak scaling:
every rank owns its own 1024-utterance shard); torch.distributed is used only for the barrier and
the max-over-ranks of the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NS_BYTES_PER_FRAME = 320          # 160 B int16 read + 160 B int16 written
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
UTTS_PER_GPU = 1024


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=UTTS_PER_GPU, help="utterances per GPU (default: configs[1])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-utts", type=int, default=1024, help="utterances in the bounded CPU sample (1024 = ~10 core-seconds)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline (the 1-GPU box share is 16)")
    return ap.parse_args()


def build_shard(n_utt, first, device):
    """The SURVEY 8(d) corpus, utterances first..first+n_utt-1, packed into HBM.  The harmonic part
    is evaluated with torch on the device (float64), the LCG noise on the host; formula identical to
    speech_enhancement_amd.corpus.synth_utterance."""
    import torch
    import speech_enhancement_amd as sea
    from speech_enhancement_amd import corpus
    lengths = np.array([corpus.utterance_length(u) for u in range(first, first + n_utt)], dtype=np.int64)
    offsets, total, order = sea.PackedBatch.layout(lengths)
    data = torch.zeros(max(total, 8), dtype=torch.int16, device=device)
    for k, u in enumerate(range(first, first + n_utt)):
        L = int(lengths[k])
        i = torch.arange(L, dtype=torch.float64, device=device)
        f0 = 110 + (u % 97)
        speech = torch.zeros(L, dtype=torch.float64, device=device)
        for h in range(1, 9):
            speech += torch.sin(2 * np.pi * h * f0 * i / 16000.0) / h
        speech *= 3000.0 * ((torch.arange(L, device=device) % 6400) < 3200)
        s = corpus.lcg_stream(12345 + u, L)
        noise = ((s >> np.uint32(16)) % np.uint32(1401)).astype(np.int64) - 700
        x = torch.trunc(speech + torch.from_numpy(noise).to(device)).to(torch.int16)
        if u % 5 == 0:
            x[:400] = 0
        data[int(offsets[k]):int(offsets[k]) + L] = x
    return sea.PackedBatch(data, torch.from_numpy(offsets).to(device), torch.from_numpy(lengths).to(device),
                           torch.from_numpy(order).to(device), offsets, lengths)


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


def cpu_baseline(batch, n_sample, threads):
    """The reference C itself (oracle/_ref, when it was built) or the oracle port, one utterance
    per thread over all host cores, on the first n_sample utterances of this rank's shard."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    kind = "reference" if O.have_reference() else "port"
    lib = O.Reference() if kind == "reference" else O.Oracle()
    host = batch.data.cpu().numpy()
    n_sample = min(n_sample, batch.n_utt)
    utts = [np.ascontiguousarray(host[o:o + l]) for o, l in
            zip(batch.host_offsets[:n_sample], batch.host_lengths[:n_sample])]
    frames = int(sum(len(u) // 80 for u in utts))
    cores = max(1, min(threads, len(os.sched_getaffinity(0))))
    lib.etsi_denoise(utts[0][:800])  # one-time table init outside the timed region
    outs = [None] * len(utts)

    def work(i):
        outs[i] = lib.etsi_denoise(utts[i])

    passes = 3  # ~20 core-seconds of CPU work in total
    with quiet_stderr():
        with ThreadPoolExecutor(max_workers=cores) as ex:
            t0 = time.perf_counter()
            for _ in range(passes):
                list(ex.map(work, range(len(utts))))
            dt = time.perf_counter() - t0
    return dict(value=passes * frames / dt, unit="frames/s", cores=cores, kind=kind,
                sample=f"{passes} passes over the first {len(utts)} utterances of the shard "
                       f"({passes * frames} frames, {dt:.2f} s wall on {cores} threads)"), outs


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import speech_enhancement_amd as sea

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    sea.load().sea_init(-1)
    batch = build_shard(args.utts, rank * args.utts, device)
    out = torch.zeros_like(batch.data)
    frames_per_step = batch.n_frames
    audio_s_per_step = frames_per_step * 80 / 16000.0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sea.ns_denoise_batch(batch, out=out)
    barrier()

    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        events[k][0].record()
        sea.ns_denoise_batch(batch, out=out)
        events[k][1].record()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in events]

    # whole-job aggregate: frames summed over ranks, time = max over ranks (no data-path collective)
    from speech_enhancement_amd.shard import reduce_job
    total_frames, dt_max = reduce_job(frames_per_step * args.steps, dt, dist if world > 1 else None, device)
    value = total_frames / dt_max

    result = None
    if rank == 0:
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        achieved = frames_per_step * NS_BYTES_PER_FRAME / avg_kernel_s / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path) and args.utts == UTTS_PER_GPU:
            with open(pmc_path) as f:
                traffic = json.load(f).get("ns_denoise_kernel_bytes_per_launch")
        result = {
            "metric": "NoiseSup frames/sec (16 kHz, hop 80, 256-pt rfft, two-stage Wiener), batched",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "rtf": dt_max / (audio_s_per_step * args.steps * world),
            "config": {
                "workload": "BASELINE configs[1]: 1024-utterance batch per GPU, 256-pt rfft + two-stage Wiener "
                            "NoiseSup (etsi_denoise semantics), SURVEY 8(d) synthetic 16 kHz corpus, 2-6 s utterances",
                "utterances_per_gpu": args.utts,
                "frames_per_step_per_gpu": frames_per_step,
                "audio_seconds_per_step_per_gpu": audio_s_per_step,
                "sharding": f"{world} independent shard(s), no collective on the data path",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "sea::ns_denoise_kernel" if os.environ.get("SEA_NS_KERNEL") == "single"
                          else "sea::ns_denoise_pipe_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": frames_per_step * NS_BYTES_PER_FRAME,
                "avg_launch_ms": avg_kernel_s * 1e3,
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is reported at N=1 only
            base, outs = cpu_baseline(batch, args.cpu_utts, args.cpu_threads)
            result["cpu_baseline"] = base
            # parity spot check on the sample (the oracle as checker, never as the measured path)
            got = batch.split(out, full_frames_only=True)
            bad = sum(int(np.abs(g.astype(np.int32) - o[:len(g)].astype(np.int32)).max() > 2)
                      for g, o in zip(got[:len(outs)], outs) if len(g))
            result["parity_check"] = f"{len(outs) - bad}/{len(outs)} sampled utterances within 2 LSB of the CPU {base['kind']}"
            result["gpu_over_cpu"] = value / base["value"]
    barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
