"""The arithmetic of the LSTM kernels' latency forms (csrc/k_lstm_q8_lat.hip), restated in NumPy and
checked against the definition of the q8 arithmetic (oracle/pxo_core.c lstm_step_q: A0..A3, V, U, t) -- CPU only.

The kernels themselves are tested bit for bit on the GPU (tests/test_gpu_parity.py::test_*_latency_form_equals_tile_form);
this file pins the SCHEME they implement, independent of any device:
  * a 4-read tile whose sixteen MFMA columns are (slot, read); the B fragment of weight digit w2 / w1 / w0 starts at
    strip position 2 / 1 / 0 of [0 | 0 | h2 | h1 | h0 | 0]: three products into ONE accumulator leave level A_l in slot l;
  * the half block: two k halves under (w2 | w1) read at positions (2 | 1), then (w0 | 0) at position 0;
  * levels -> pre-activation integer inside a 16-lane row: W = (acc << 8)[lane - 4] + acc, a second tile rotated by four
    lanes into slots 2 and 0, one conversion, t = fma(F[lane - 8 mod 16], 65536, F); a third tile's t moved to slot 2.
"""
import numpy as np
import pytest

RNG = np.random.default_rng(20260930)


def digits(v):
    """balanced base-256 digits d0 + 256 d1 + 65536 d2 of an int array, each in [-128, 127] (oracle q_digits)"""
    v = np.asarray(v, dtype=np.int64)
    d0 = ((v + 128) % 256) - 128
    r1 = (v - d0) // 256
    d1 = ((r1 + 128) % 256) - 128
    d2 = (r1 - d1) // 256
    assert (np.abs(d2) <= 128).all()
    return d0, d1, d2


def reference_levels(wq, hq):
    """A0..A3 of lstm_step_q for weights wq [G][K] and hidden values hq [K][reads] (exact integers)"""
    w0, w1, w2 = digits(wq)
    h0, h1, h2 = digits(hq)
    a0 = w2 @ h2
    a1 = w2 @ h1 + w1 @ h2
    a2 = w2 @ h0 + w1 @ h1 + w0 @ h2
    a3 = w1 @ h0 + w0 @ h1
    return a0, a1, a2, a3


def t_of(V, U):
    """t = fma((float)V, 65536, (float)U) in float32 arithmetic: both conversions round, the product is exact, one
    more rounding in the sum (float64 holds the exact sum of the two float32 values)"""
    fv, fu = np.float32(V.astype(np.int32)), np.float32(U.astype(np.int32))
    return np.float32(fv.astype(np.float64) * 65536.0 + fu.astype(np.float64))


def strip(hq):
    """one k block of a hidden vector as the latency form holds it: [6 positions][K][4 reads] = [0 | 0 | h2 | h1 | h0 | 0]"""
    h0, h1, h2 = digits(hq)
    z = np.zeros_like(h0)
    return np.stack([z, z, h2, h1, h0, z])


def b_fragment(strips, first_position):
    """B [K][16 columns], column n = 4 slot + read: slot s carries strip position first_position + s"""
    return np.concatenate([strips[first_position + s] for s in range(4)], axis=1)


@pytest.mark.parametrize('K', [48, 64])
def test_three_products_leave_the_levels_in_the_slots(K):
    G, R = 16, 4
    wq = RNG.integers(-8355711, 8355712, (G, K))
    hq = RNG.integers(-(1 << 22), (1 << 22) + 1, (K, R))
    w0, w1, w2 = digits(wq)
    st = strip(hq)
    acc = w2 @ b_fragment(st, 2) + w1 @ b_fragment(st, 1) + w0 @ b_fragment(st, 0)       # one accumulator, [G][16]
    a = reference_levels(wq, hq)
    for level in range(4):
        assert np.array_equal(acc[:, 4 * level:4 * level + 4], a[level]), level
    assert np.abs(acc).max() < 2 ** 31


def test_half_block_under_two_weight_digits_per_product():
    """32 inputs in a 64-wide product: k halves (w2 | w1) with the fragment read at positions (2 | 1), then (w0 | 0)"""
    G, K, R = 16, 32, 4
    wq = RNG.integers(-8355711, 8355712, (G, K))
    hq = RNG.integers(-(1 << 22), (1 << 22) + 1, (K, R))
    w0, w1, w2 = digits(wq)
    st = strip(hq)
    a_first = np.concatenate([w2, w1], axis=1)                       # [G][64]: k 0-31 digit 2, k 32-63 digit 1
    b_first = np.concatenate([b_fragment(st, 2), b_fragment(st, 1)], axis=0)
    a_second = np.concatenate([w0, np.zeros_like(w0)], axis=1)
    b_second = np.concatenate([b_fragment(st, 0), b_fragment(st, 0)], axis=0)      # (second half: zero weights, any bytes)
    acc = a_first @ b_first + a_second @ b_second
    a = reference_levels(wq, hq)
    for level in range(4):
        assert np.array_equal(acc[:, 4 * level:4 * level + 4], a[level]), level


def row_shr(x, n):
    """DPP row_shr:n with bound_ctrl inside one 16-lane row: lane i reads lane i - n, zero from outside the row"""
    out = np.zeros_like(x)
    out[..., n:] = x[..., :16 - n]
    return out


def row_ror(x, n):
    """DPP row_ror:n: lane i reads lane (i - n) mod 16"""
    return np.roll(x, n, axis=-1)


def words(acc):
    """W = (acc << 8)[lane - 4] + acc: V = A0 256 + A1 in slot 1, U = A2 256 + A3 in slot 3"""
    return row_shr(acc.astype(np.int64) << 8, 4) + acc


def test_levels_to_preactivation_inside_the_row_three_tiles():
    """ql_levels3: tile 0 -> slot 3, tile 1 -> slot 0 (its words rotated by four lanes), tile 2 -> slot 2"""
    K, R = 96, 4
    hq = RNG.integers(-(1 << 22), (1 << 22) + 1, (K, R))
    st = strip(hq)
    tiles, want = [], []
    for _ in range(3):
        wq = RNG.integers(-8355711, 8355712, (1, K))                 # one gate row per tile is enough: rows are independent
        w0, w1, w2 = digits(wq)
        tiles.append((w2 @ b_fragment(st, 2) + w1 @ b_fragment(st, 1) + w0 @ b_fragment(st, 0))[0])
        a0, a1, a2, a3 = (x[0] for x in reference_levels(wq, hq))
        want.append(t_of(a0 * 256 + a1, a2 * 256 + a3))
    r1 = words(tiles[0])
    rot = row_ror(words(tiles[1]), 4)
    for bank in (0, 2):                                               # bank_mask 0x5: banks 0 and 2 take the rotated tile
        r1[4 * bank:4 * bank + 4] = rot[4 * bank:4 * bank + 4]
    f1 = np.float32(r1.astype(np.int32))
    f2 = np.float32(words(tiles[2]).astype(np.int32))
    t1 = np.float32(row_ror(f1, 8).astype(np.float64) * 65536.0 + f1.astype(np.float64))
    t2 = np.float32(row_shr(f2, 8).astype(np.float64) * 65536.0 + f2.astype(np.float64))
    merged = t1.copy()
    merged[8:12] = t2[12:16]                                          # row_shl:4 into bank 2
    assert np.array_equal(merged[12:16], want[0])                    # slot 3: tile 0
    assert np.array_equal(merged[0:4], want[1])                      # slot 0: tile 1
    assert np.array_equal(merged[8:12], want[2])                     # slot 2: tile 2


def test_words_fit_int32_for_the_shipped_layer_sizes():
    """V = A0 256 + A1 and U = A2 256 + A3 stay inside int32 for K <= 192 (|h2| <= 64: |q| <= 2^22)"""
    K = 160
    w = np.full((1, K), 8355711)
    h = np.full((K, 4), 1 << 22)
    a0, a1, a2, a3 = reference_levels(w, h)
    assert max(np.abs(a0 * 256 + a1).max(), np.abs(a2 * 256 + a3).max()) < 2 ** 31
    st = strip(h)
    w0, w1, w2 = digits(w)
    acc = w2 @ b_fragment(st, 2) + w1 @ b_fragment(st, 1) + w0 @ b_fragment(st, 0)
    assert np.abs(words(acc[0])).max() < 2 ** 31
