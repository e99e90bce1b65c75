"""data::landmark::compute_descriptor (src/openvslam/data/landmark.cc:127-184), batched: oracle
against a numpy restatement on the CPU, GPU against the oracle."""
import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

POPCNT = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint16)


def make_batch(rng, sizes):
    offsets = np.zeros(len(sizes) + 1, np.uint32)
    offsets[1:] = np.cumsum(sizes)
    descs = np.zeros((int(offsets[-1]), 32), np.uint8)
    for l, n in enumerate(sizes):
        if n == 0:
            continue
        centre = rng.integers(0, 256, 32, dtype=np.uint8)
        block = np.tile(centre, (n, 1))
        for i in range(n):
            for b in rng.choice(256, rng.integers(0, 60), replace=False):
                block[i, b >> 3] ^= np.uint8(1 << (b & 7))
        if n > 3:
            block[1] = block[0]       # ties: equal medians must resolve to the first index
        descs[offsets[l]:offsets[l + 1]] = block
    return descs, offsets


def numpy_best(descs, offsets):
    out = []
    for l in range(len(offsets) - 1):
        d = descs[offsets[l]:offsets[l + 1]]
        n = len(d)
        if n == 0:
            out.append(-1)
            continue
        D = POPCNT[d[:, None, :] ^ d[None, :, :]].sum(axis=2)
        med = np.sort(D, axis=1)[:, int(0.5 * (n - 1))]
        out.append(int(np.argmin(med)))  # first minimum
    return np.array(out, np.int32)


def oracle_best(descs, offsets):
    best = np.full(len(offsets) - 1, -9, np.int32)
    oracle.check(oracle.lib().ovo_landmark_compute_descriptors(A.ptr(descs, A.u8p), A.ptr(offsets, A.u32p), len(offsets) - 1, A.ptr(best, A.i32p)))
    return best


SIZES = [1, 2, 3, 4, 0, 7, 64, 65, 130, 5, 0, 33]


def test_oracle_equals_numpy_restatement():
    descs, offsets = make_batch(np.random.default_rng(5), SIZES)
    assert np.array_equal(oracle_best(descs, offsets), numpy_best(descs, offsets))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,sizes", [(1, SIZES), (2, [300, 1, 1000, 2]), (3, list(np.random.default_rng(3).integers(0, 40, 3000)))])
def test_gpu_matches_oracle(seed, sizes):
    from openvslam_amd import match
    descs, offsets = make_batch(np.random.default_rng(seed), [int(s) for s in sizes])
    assert np.array_equal(match.landmark_compute_descriptors(descs, offsets), oracle_best(descs, offsets))


@pytest.mark.gpu
def test_gpu_rejects_oversized_landmark():
    from openvslam_amd import _lib
    from openvslam_amd import match
    descs = np.zeros((5000, 32), np.uint8)
    with pytest.raises(_lib.OvsError):
        match.landmark_compute_descriptors(descs, np.array([0, 5000], np.uint32))
