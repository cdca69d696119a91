"""acc_genomics_amd: MI355X-native PairHMM forward + HTC Smith-Waterman behind a C ABI.

The product is acc_genomics_amd/libaccg_hip.so (hand-written HIP for gfx950, see csrc/ and
include/accg.h).  This package is only the thin ctypes binding used by tests and bench.py; it
never computes anything itself and raises if the HIP library is missing."""
from .lib import (ACCG_PHMM_FAST, ACCG_PHMM_STRICT, HTC_WEIGHTS, AccgError, BwaswBatch, Context, PhmmBatch, PhmmRing, SmemBatch, SmemIndex, SwBatch, lib_path, load)  # noqa: F401
