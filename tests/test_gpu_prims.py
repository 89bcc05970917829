"""GPU unit tests of the hand-written wave primitives against plain code on the same device and against numpy.  The CPU
emulator compiles their C fallbacks (FPL_EMU), so these are the only tests that see the inline-asm add-with-carry of
sliced_max (with its manual hazard nop) and the DPP forms of wave_prev_u32 / the reductions / the scans in isolation -- a
different compiler version or scheduling decision that breaks one of them shows up here, not as a parity diff somewhere."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prims():
    import torch

    assert torch.cuda.is_available()
    from tests.gpu_prims import build

    L = C.CDLL(build.build())
    L.prims_sliced_max.restype = C.c_int
    L.prims_sliced_max.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.prims_wave.restype = C.c_int
    L.prims_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.prims_kmer_stream.restype = C.c_int
    L.prims_kmer_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return L


def test_kmer_stream_from_dot4_packs_and_the_neighbours_pack(prims):
    """kmer_stream (k_stats_sorted's row set-up): a dword's four 2-bit codes packed by v_dot4, the previous lane's finished pack
    through DPP -- against the v_perm form on the same device and against numpy"""
    rng = np.random.default_rng(3)
    n = 2048
    letters = np.frombuffer(b"ACGTUNacgt", dtype=np.uint8)
    by = letters[rng.integers(0, 5, (n, 64, 8))]
    by[::5] = letters[rng.integers(0, len(letters), by[::5].shape)]
    halo_b = letters[rng.integers(0, 5, (n, 4))]
    inp = np.ascontiguousarray(by).view(np.uint32).reshape(n, 64, 2)
    halo = np.ascontiguousarray(halo_b).view(np.uint32).reshape(n)
    out = np.zeros((n, 64, 6), np.uint32)
    assert prims.prims_kmer_stream(inp.ctypes.data, halo.ctypes.data, out.ctypes.data, n) == 0
    code = lambda b: ((b & 2) | ((b >> 2) & 1)).astype(np.uint32)  # noqa: E731  (Stats::base2val for A T/U C G)
    pack = lambda c: (c[..., 0] << 6) | (c[..., 1] << 4) | (c[..., 2] << 2) | c[..., 3]  # noqa: E731
    p0, p1 = pack(code(by[:, :, 0:4])), pack(code(by[:, :, 4:8]))
    ph = np.concatenate([pack(code(halo_b))[:, None], p1[:, :-1]], axis=1)
    want = (ph << 16) | (p0 << 8) | p1
    assert np.array_equal(out[:, :, 2], p0) and np.array_equal(out[:, :, 3], p1)
    assert np.array_equal(out[:, :, 4], p0) and np.array_equal(out[:, :, 5], p1)
    assert np.array_equal(out[:, :, 1], want)
    assert np.array_equal(out[:, :, 0], want)


@pytest.mark.parametrize("nb", [6, 7])
def test_sliced_max_add_with_carry_form(prims, nb):
    rng = np.random.default_rng(nb)
    n = 4096
    inp = rng.integers(0, 1 << 32, (n, 64, 8), dtype=np.uint64).astype(np.uint32)
    # candidate masks of every density, incl. one bit and all ones; count planes with few distinct values (ties everywhere)
    dens = rng.random((n, 64, 1))
    inp[:, :, 7] = np.packbits(rng.random((n, 64, 32)) < dens, axis=-1, bitorder="little").view(np.uint32)[..., 0]
    inp[::7, :, 7] = 0xFFFFFFFF
    inp[1::7, :, 7] = (1 << rng.integers(0, 32, (len(inp[1::7]), 64))).astype(np.uint32)
    inp[2::5, :, :7] &= rng.integers(0, 1 << 32, (len(inp[2::5]), 1, 7), dtype=np.uint64).astype(np.uint32)  # (sparse planes)
    inp[3::11, :, 7] = 0  # lanes without a candidate stay out (the kernels never call it with an empty mask)
    act = rng.integers(0, 1 << 64, n, dtype=np.uint64)
    act[::3] = 0xFFFFFFFFFFFFFFFF
    act[1::9] = 1 << rng.integers(0, 64, len(act[1::9])).astype(np.uint64)
    out = np.zeros((n, 64, 4), np.uint32)
    assert prims.prims_sliced_max(inp.ctypes.data, act.ctypes.data, out.ctypes.data, n, nb) == 0
    lanes = ((act[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & 1).astype(bool) & (inp[:, :, 7] != 0)
    assert lanes.sum() > 100000
    # product form == plain loop on the device, and both == numpy
    assert np.array_equal(out[lanes][:, 0:2], out[lanes][:, 2:4])
    assert (out[~lanes] == 0xDEADBEEF).all()
    bits = ((inp[:, :, :nb, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.int64)  # [n, lane, plane, pos]
    cnt = (bits << np.arange(nb, dtype=np.int64)[None, None, :, None]).sum(axis=2)
    cand = ((inp[:, :, 7, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool)
    cnt = np.where(cand, cnt, -1)
    assert np.array_equal(out[lanes][:, 0].astype(np.int64), cnt.max(axis=-1)[lanes])
    assert np.array_equal(out[lanes][:, 1].astype(np.int64), cnt.argmax(axis=-1)[lanes])  # (argmax: the first maximum)


def test_dpp_neighbour_reductions_and_scans(prims):
    rng = np.random.default_rng(1)
    n = 2048
    v = rng.integers(0, 1 << 32, (n, 64), dtype=np.uint64).astype(np.uint32)
    v[::4] >>= 20  # (small values: ties for the minima / maxima)
    v[1::16] = 0
    v[2::16] = 0xFFFFFFFF
    out = np.zeros((n, 64, 8), np.uint32)
    assert prims.prims_wave(v.ctypes.data, out.ctypes.data, n) == 0
    prev = np.concatenate([(0xABCD0000 + np.arange(n, dtype=np.uint64)).astype(np.uint32)[:, None], v[:, :-1]], axis=1)
    assert np.array_equal(out[:, :, 0], prev)  # lane 0 keeps the value handed in; rows of 16 lanes are crossed
    assert np.array_equal(out[:, :, 1], np.broadcast_to(v.sum(axis=1, dtype=np.uint64).astype(np.uint32)[:, None], (n, 64)))
    assert np.array_equal(out[:, :, 2], np.cumsum(v.astype(np.uint64), axis=1).astype(np.uint32))
    assert np.array_equal(out[:, :, 3], np.broadcast_to(v.max(axis=1)[:, None], (n, 64)))
    assert np.array_equal(out[:, :, 4], np.broadcast_to(v.min(axis=1)[:, None], (n, 64)))
    key = (v.astype(np.uint64) << np.uint64(32)) | (63 - np.arange(64, dtype=np.uint64))[None, :]
    kmin = key.min(axis=1)
    assert np.array_equal(out[:, :, 5], np.broadcast_to((kmin & 0xFFFFFFFF).astype(np.uint32)[:, None], (n, 64)))
    assert np.array_equal(out[:, :, 6], np.broadcast_to((kmin >> np.uint64(32)).astype(np.uint32)[:, None], (n, 64)))
    assert np.array_equal(out[:, :, 7], np.broadcast_to(v[:, 37][:, None], (n, 64)))
