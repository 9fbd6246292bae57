"""speech_enhancement_amd -- MI355X (gfx950) engine for the per-frame noise-suppression hot path of
guokiddo1/speech_enhancement: etsi/ two-stage Wiener NoiseSup + rfft + CompCeps and the
resyth_64sub_{ori,IBM} 64-band gammatone resynthesis, as hand-written HIP kernels behind the
reference's own C entry points (include/sea_mi355x.h).

Only what that path needs lives here:
  csrc/       HIP kernels, host table builder, the C ABI (libsea_mi355x.so, built in-tree)
  engine.py   host-side mirror of the reference interface + batched HBM-resident forms
  corpus.py   the deterministic synthetic corpus the measurements run on
  shard.py    utterance sharding over the GPUs of a node (LPT / blocks) and the two-scalar job reduction
  host/       plain-C drivers with the reference's command lines (cfg -> list -> WAV in / WAV out) and the
              mask text format; deal/*.sh build them
"""
from . import corpus  # noqa: F401
from ._lib import LIB_PATH, SeaError, load  # noqa: F401
from .engine import (DoCompCeps, MaskBatch, NoiseSup, PackedBatch, afe_features_batch, compceps_batch,  # noqa: F401
                     compceps_frames, etsi_denoise, gammaToneFilter, irm_target, irm_target_batch, ns_denoise_batch,
                     ns16k_streams_push, ns16k_tables, ns_streams_push, resynth, resynth_batch, resynth_scratch_elems, rfft, rfft_any_batch, rfft_batch, subband_batch, subbband,
                     tables)
