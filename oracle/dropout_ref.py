"""numpy restatement of the dropout keep-mask stream (csrc/rr_common.h: rr_hash_group / rr_hash_lane / rr_keep).
TEST INFRASTRUCTURE (oracle): lets train-mode parity tests feed the SAME masks to the CPU oracle
that the HIP epilogues generate on the fly (hazard H2: torch's RNG stream cannot be matched)."""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _fmix32(h):
    h = h & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def hash_group(seed: int, group: np.ndarray) -> np.ndarray:
    """rr_hash_group: one word per aligned group of 4 elements."""
    group = np.asarray(group, dtype=np.uint64)
    lo, hi = group & _M32, group >> np.uint64(32)
    s0, s1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    h = _fmix32(lo ^ s0)
    h = _fmix32((h + ((hi * np.uint64(0x9E3779B1)) & _M32) + s1) & _M32)
    return h


def hash_u32(seed: int, index: np.ndarray) -> np.ndarray:
    """rr_hash_u32 = rr_hash_lane(rr_hash_group(seed, index >> 2), index & 3)."""
    index = np.asarray(index, dtype=np.uint64)
    w = hash_group(seed, index >> np.uint64(2))
    e = index & np.uint64(3)
    k = (e * np.uint64(0x9E3779B9) + np.uint64(0x7F4A7C15)) & _M32
    h = ((w ^ k) * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(15)
    return h


def threshold(p: float) -> int:
    t = float(np.float32(p)) * 4294967296.0
    if t <= 0.0:
        return 0
    if t >= 4294967295.0:
        return 4294967295
    return int(t)


def keep_mask(seed: int, index: np.ndarray, p: float) -> np.ndarray:
    return hash_u32(seed, index) >= np.uint64(threshold(p))


def site_seed(seed: int, site: int) -> int:
    """functions._site_seed."""
    return (int(seed) * 0x9E3779B97F4A7C15 + site * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
