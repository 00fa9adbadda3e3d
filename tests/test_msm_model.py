"""CPU model of the Schnorr-type multi-scalar multiplication (ec_schnorr_verify_all_batch; k_msm_scal / k_msm_table_g / k_msm_loop_g /
k_msm_sum_g / k_msm_final_g of libecc_amd/csrc): the ALGORITHM the kernels implement, restated with Python integers and affine points --
the signed 4-bit recoding with its carry digit (k' = k + 0x88..8, digit = nibble - 8, one more digit 0 / 1 on top), the lane layout of the
Straus loop (lane l owns items j L + l, j < K; per window four shared doublings, the K keys, and from window 2 zlen down the K negated
signature points), the fan-in-16 tree and the final comparison with the generator's term.  It pins the design constants (2 wlen + 1 and
2 zlen + 1 = 33 windows, the order of doublings and additions, ragged last lanes) against the equation itself,
    T = [sum z_i s_i]G + sum ([z_i (q - e_i)]Y_i - [z_i]R_i),
so that a change of the kernels' indexing has a CPU-side statement to be checked against; the device code itself is tested on the GPU
(tests/test_gpu_schnorr_msm.py) and its field / group arithmetic on the host build (tests/test_u29g_host.py)."""
import numpy as np
import pytest

from oracles import CURVES, py_add, py_mul


def recode(k, nbytes):
    """signed window recoding of the kernels' recode_scalar: the 2 nbytes nibbles of k + 0x88..8 (little-endian nibble order), and the carry out"""
    kk = k + int.from_bytes(b"\x88" * nbytes, "big")
    digits = [((kk >> (4 * i)) & 15) - 8 for i in range(2 * nbytes)]
    carry = kk >> (8 * nbytes)
    assert carry in (0, 1)
    assert sum(d << (4 * i) for i, d in enumerate(digits)) + (carry << (8 * nbytes)) == k
    return digits, carry


def neg(P, p):
    return None if P is None else (P[0], (-P[1]) % p)


def straus(Y, R, w, z, K, wlen, zlen, a, p):
    """the lanes' sums as k_msm_loop_g forms them, then the tree of k_msm_sum_g"""
    n = len(Y)
    L = (n + K - 1) // K
    wwin, zwin = 2 * wlen, 2 * zlen
    rw = [recode(v, wlen) for v in w]
    rz = [recode(v, zlen) for v in z]
    tab = lambda P: [None] + [py_mul(m, P, a, p) for m in range(1, 9)]   # [1..8]P (index 0 unused)
    tY, tR = [tab(P) for P in Y], [tab(P) for P in R]
    lanes = []
    for lane in range(L):
        acc = None
        for pos in range(wwin, -1, -1):
            if pos != wwin:
                for _ in range(4):
                    acc = py_add(acc, acc, a, p)
            for j in range(K):
                i = j * L + lane
                if i >= n:
                    continue
                d = rw[i][1] if pos == wwin else rw[i][0][pos]
                if d:
                    T = tY[i][abs(d)]
                    acc = py_add(acc, T if d > 0 else neg(T, p), a, p)
            if pos <= zwin:
                for j in range(K):
                    i = j * L + lane
                    if i >= n:
                        continue
                    d = rz[i][1] if pos == zwin else rz[i][0][pos]
                    if d:
                        T = tR[i][abs(d)]
                        acc = py_add(acc, neg(T, p) if d > 0 else T, a, p)     # the signature points enter negated
        lanes.append(acc)
    while len(lanes) > 1:                                                      # fan-in 16
        lanes = [_sum(lanes[k:k + 16], a, p) for k in range(0, len(lanes), 16)]
    return lanes[0]


def _sum(pts, a, p):
    acc = None
    for P in pts:
        acc = py_add(acc, P, a, p)
    return acc


@pytest.mark.parametrize("curve,n,K", [("SECP256K1", 13, 4), ("SECP256K1", 5, 8), ("SECP192R1", 37, 3), ("SECP256R1", 16, 1)])
def test_straus_model_evaluates_the_batch_equation(curve, n, K):
    c = CURVES[curve]
    p, a, q = c["p"], c["a"], c["q"]
    G = (c["gx"], c["gy"])
    qlen = (q.bit_length() + 7) // 8
    rng = np.random.default_rng(n * 100 + K)
    rnd = lambda bits: int.from_bytes(rng.bytes(bits // 8 + 8), "big") % (1 << bits)
    x = [rnd(q.bit_length() - 1) % (q - 1) + 1 for _ in range(n)]
    k = [rnd(q.bit_length() - 1) % (q - 1) + 1 for _ in range(n)]
    e = [rnd(q.bit_length() - 1) % q for _ in range(n)]
    z = [rnd(128) or 1 for _ in range(n)]
    z[0] = (1 << 128) - 1                        # every nibble 0xf: the recoding's carry digit is 1
    z[1 % n] = 0x77777777777777777777777777777777 if n > 1 else z[0]   # digits -1 ... no carry
    Y = [py_mul(v, G, a, p) for v in x]
    R = [py_mul(v, G, a, p) for v in k]
    s = [(k[i] + e[i] * x[i]) % q for i in range(n)]
    w = [z[i] * ((q - e[i]) % q) % q for i in range(n)]
    c_sum = sum(z[i] * s[i] for i in range(n)) % q
    assert recode(z[0], 16)[1] == 1
    S = straus(Y, R, w, z, K, qlen, 16, a, p)
    cG = py_mul(c_sum, G, a, p)
    assert py_add(S, cG, a, p) is None           # valid batch: the combination vanishes
    # the direct evaluation of the same sum
    direct = None
    for i in range(n):
        direct = py_add(direct, py_mul(w[i], Y[i], a, p), a, p)
        direct = py_add(direct, neg(py_mul(z[i], R[i], a, p), p), a, p)
    assert S == direct
    # one bad s: the generator's term moves, the sum does not vanish
    c_bad = (c_sum + z[n // 2]) % q
    assert py_add(S, py_mul(c_bad, G, a, p), a, p) is not None


def test_recoding_digits_and_window_counts():
    for nbytes, k in ((16, 0), (16, 1), (16, (1 << 128) - 1), (16, 0x80000000000000000000000000000000), (32, (1 << 256) - 1), (3, 0x777777), (3, 0x888888)):
        d, cy = recode(k, nbytes)
        assert len(d) == 2 * nbytes and all(-8 <= v <= 7 for v in d) and cy in (0, 1)
    # a 128-bit z takes part in 2 * 16 + 1 = 33 windows, a qlen-byte scalar in 2 qlen + 1
    assert len(recode(5, 16)[0]) + 1 == 33
