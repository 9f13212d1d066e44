"""Pins of the oracle's Brownian pieces against the REAL reference (golden fixtures) and Random123."""
import numpy as np
import pytest
import torch

from oracle import brownian_ref, counter
from tests import helpers

# Random123 v1.14 known-answer vectors for philox4x32_R(10): (counter, key, expected)
PHILOX_KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,expected", PHILOX_KAT)
def test_oracle_philox_known_answers(ctr, key, expected):
    assert counter.philox(ctr, key) == expected


@pytest.mark.parametrize("ctr,key,expected", PHILOX_KAT)
def test_library_philox_known_answers(ctr, key, expected):
    """The block function compiled into libtorchsde_amd.so (shared header with the kernels)."""
    import ctypes
    from torchsde_amd import _native
    lib = _native.load()
    out = (ctypes.c_uint32 * 4)()
    lib.tsde_philox4x32_10((ctypes.c_uint32 * 4)(*ctr), (ctypes.c_uint32 * 2)(*key), out)
    assert tuple(out) == expected


def test_counter_layout_agrees():
    import ctypes
    from torchsde_amd import _native
    lib = _native.load()
    for quad, cell, node, stream in [(0, 0, 0, 0), (5, 7, 9, 1), (2 ** 40 + 3, 2 ** 32 - 1, 2 ** 37 + 11, 2)]:
        out = (ctypes.c_uint32 * 4)()
        lib.tsde_noise_counter(quad, cell, node, stream, out)
        assert tuple(out) == counter.noise_counter(quad, cell, node, stream)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("levy", ["none", "space-time"])
def test_bridge_split_matches_reference(levy, tag):
    """Feed the normals the reference itself drew through the oracle's split: children must be bit-equal."""
    z = helpers.load("bridge.npz")
    pre = f"{levy}__{tag}__"
    np_dt = np.float32 if tag == "f32" else np.float64
    x = float(z[pre + "x"])
    have_h = levy != "none"
    W, H, X1, X2 = (z[pre + k] for k in ("W", "H", "X1", "X2"))
    for i in range(W.shape[0]):
        Wl, Hl, Wr, Hr = counter.bridge_split(W[i], H[i], 0.0, x, 1.0, X1[i], X2[i], have_h, dtype=np_dt)
        assert np_dt(Wl) == z[pre + "Wl"][i] and np_dt(Wr) == z[pre + "Wr"][i]
        if have_h:
            # the reference exposes U = h (W/2 + H) of each child
            Ul = np_dt(x) * (np_dt(0.5) * np_dt(Wl) + np_dt(Hl))
            Ur = np_dt(1.0 - x) * (np_dt(0.5) * np_dt(Wr) + np_dt(Hr))
            assert np_dt(Ul) == z[pre + "Ul"][i] and np_dt(Ur) == z[pre + "Ur"][i]


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_interval_merge_matches_reference(tag):
    """(W,H)[ta,u] (+) (W,H)[u,t] through the oracle's merge == the reference's own multi-interval query."""
    z = helpers.load("bridge.npz")
    pre = f"space-time__{tag}__"
    np_dt = np.float32 if tag == "f32" else np.float64
    ta, u, t = z[pre + "merge_ta_u_t"]
    Wa, Ua, Wb, Ub, Wab, Uab = (z[pre + k] for k in ("W_a", "U_a", "W_b", "U_b", "W_ab", "U_ab"))
    tol = 2e-6 if tag == "f32" else 1e-14
    for i in range(Wa.shape[0]):
        Ha = np_dt(Ua[i] / np_dt(u - ta) - np_dt(0.5) * Wa[i])
        Hb = np_dt(Ub[i] / np_dt(t - u) - np_dt(0.5) * Wb[i])
        W, H = counter.interval_merge(Wa[i], Ha, u - ta, Wb[i], Hb, t - u, True, dtype=np_dt)
        U = (t - ta) * (0.5 * W + H)
        assert abs(W - Wab[i]) <= tol and abs(U - Uab[i]) <= tol


def _seq_cases():
    z = helpers.load("brownian_seq.npz")
    return sorted({k.rsplit("__", 1)[0] for k in z.files})


@pytest.mark.parametrize("case", _seq_cases())
def test_brownian_ref_bitwise_matches_reference(case):
    """oracle/brownian_ref.py reproduces the reference's BrownianInterval bit for bit (same entropy, same
    query sequence), with and without the dt hint, with and without space-time Levy area."""
    z = helpers.load("brownian_seq.npz")
    _, levy, hint = case.split("__")
    bm = brownian_ref.BrownianIntervalRef(t0=0., t1=1., size=(3, 2), dtype=torch.float64, entropy=4321,
                                          levy_area_approximation=levy, dt=0.01 if hint == "hint" else None)
    for i, (a, b) in enumerate(z[case + "__queries"]):
        if levy == "none":
            W = bm(a, b)
        else:
            W, U = bm(a, b, return_U=True)
            assert np.array_equal(U.numpy(), z[case + "__U"][i]), (i, a, b)
        assert np.array_equal(W.numpy(), z[case + "__W"][i]), (i, a, b)
