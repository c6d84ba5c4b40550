"""reactranker_amd — MI355X-native (gfx950) D-MPNN reaction scorer + ranking losses.

Host-side mirror of the IannLiu/ReactRanker hot path over a C-ABI HIP library
(include/reactranker_hip.h).  Importing the package does not need a GPU; calling any op does,
and fails loudly if the native library is missing (no CPU fallback).
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
