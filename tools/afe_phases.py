#!/usr/bin/env python3
"""tools/afe_phases.py -- (GPU box, SEA_MI355X_LIB=ablate/libsea_<v>.so built with
``tools/build_variant.sh <v> speech_enhancement_amd/csrc/cc_kernel.hip -DSEA_AFE_TIMING``) shader clocks per 16-frame tile
workgroup 0 of afe_ceps_kernel spends in each step (two waves share a SIMD: wall-clock shares, not lone costs)."""
import ctypes, os, sys, json, subprocess
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import speech_enhancement_amd as sea
sea.load()
raw = ctypes.CDLL(os.environ["SEA_MI355X_LIB"])
from speech_enhancement_amd import corpus
utts = corpus.synth_corpus(256, max_len=96000)
batch = sea.PackedBatch.from_arrays(utts)
r = sea.afe_features_batch(batch)
torch.cuda.synchronize()
z = np.zeros(8, np.uint64)
raw.sea_afe_timing(z.ctypes.data_as(ctypes.c_void_p), 1)
r = sea.afe_features_batch(batch)
torch.cuda.synchronize()
raw.sea_afe_timing(z.ctypes.data_as(ctypes.c_void_p), 0)
n = int(z[7])
print("tiles", n, {k: int(z[i]) // max(n, 1) for i, k in enumerate(["stage", "energy", "smooth", "peaks", "window", "cc_tile"])})
