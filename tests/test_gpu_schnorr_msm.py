"""GPU tests of ec_schnorr_verify_all_batch (SURVEY.md section 8 row f4 beyond Ed25519): the batch equation of BIP0340 / ECFSDSA
(sig/bip0340.c:905-1010, sig/ecfsdsa.c:1042-) as one multi-scalar multiplication on the radix-2^29 units.
  * valid batches are accepted, for every lane shape (K items per lane, uneven tails), with R given as points or as abscissae
    (BIP0340's lift_x on the device);
  * ONE bad item anywhere rejects the batch ("one bad signature at index k"), for every way an item can be bad: s, e, R, the key,
    s >= q, an abscissa with no point, a point off the curve;
  * the algebra is pinned independently of item validity: with a fixed seed the z_i are read back, and two items are damaged so
    that z_0 d_0 + z_1 d_1 = 0 -- the combination must still vanish, while the same damage under another seed is rejected;
  * the verdict agrees with the item form evaluated by the CPU oracle on a mixed batch.
Items are built with Python integers and the oracle (keys and nonce points as [x]G, [k]G), so nothing here trusts the path under test."""
import os

import numpy as np
import pytest

import libecc_amd
from oracles import CURVES, Oracle
from test_gpu_parity import rand_bytes

pytestmark = pytest.mark.gpu

MSM_CURVES = ["SECP256K1", "SECP256R1", "SECP384R1", "BRAINPOOLP256R1", "SECP521R1", "SECP224R1"]


def make_items(curve, n, rng, even_y=False):
    """n valid Schnorr items: s = k + e x mod q, Y = [x]G, R = [k]G (with an even y when even_y: k -> q - k otherwise), e random"""
    o = Oracle(curve)
    c = CURVES[curve]
    q, p = c["q"], c["p"]
    cl, ql = o.clen, o.qlen
    rnd = lambda: (int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1
    x = [rnd() for _ in range(n)]
    k = [rnd() for _ in range(n)]
    e = [rnd() for _ in range(n)]
    be = lambda v: v.to_bytes(ql, "big")
    Y, st = o.scalar_mult(b"".join(be(v) for v in x))
    R, st2 = o.scalar_mult(b"".join(be(v) for v in k))
    assert set(st) == {0} and set(st2) == {0}
    if even_y:
        R = bytearray(R)
        for i in range(n):
            y = int.from_bytes(R[2 * cl * i + cl:2 * cl * (i + 1)], "big")
            if y & 1:
                R[2 * cl * i + cl:2 * cl * (i + 1)] = (p - y).to_bytes(cl, "big")
                k[i] = q - k[i]
        R = bytes(R)
    s = [(k[i] + e[i] * x[i]) % q for i in range(n)]
    return {"s": b"".join(be(v) for v in s), "ne": b"".join(be((q - v) % q) for v in e), "Y": Y, "R": R,
            "rx": b"".join(R[2 * cl * i:2 * cl * i + cl] for i in range(n)), "q": q, "p": p, "cl": cl, "ql": ql, "n": n}


def patched(buf, width, i, new):
    return buf[:width * i] + new + buf[width * (i + 1):]


@pytest.fixture(autouse=True, params=["straus", "bucket"])
def msm_algo(request):
    """every test of this module runs on both evaluations of the combination: the Straus loop (round 5) and the bucket form (round 6);
    ECAMD_SCHNORR_MSM_ALGO is read by the library at every call"""
    old = os.environ.get("ECAMD_SCHNORR_MSM_ALGO")
    os.environ["ECAMD_SCHNORR_MSM_ALGO"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("ECAMD_SCHNORR_MSM_ALGO", None)
    else:
        os.environ["ECAMD_SCHNORR_MSM_ALGO"] = old


@pytest.fixture
def msm_k():
    """ECAMD_SCHNORR_MSM_K for the duration of a test (read by the library at every call)"""
    old = os.environ.get("ECAMD_SCHNORR_MSM_K")

    def setk(k):
        if k is None:
            os.environ.pop("ECAMD_SCHNORR_MSM_K", None)
        else:
            os.environ["ECAMD_SCHNORR_MSM_K"] = str(k)
    yield setk
    if old is None:
        os.environ.pop("ECAMD_SCHNORR_MSM_K", None)
    else:
        os.environ["ECAMD_SCHNORR_MSM_K"] = old


@pytest.mark.parametrize("curve", MSM_CURVES)
def test_valid_batches_accepted_and_one_bad_item_rejects(gpu_ctx, curve, msm_k):
    rng = np.random.default_rng(900 + MSM_CURVES.index(curve))
    cv = gpu_ctx.curve(curve)
    try:
        lift_ok = cv.schnorr_msm_available(1)
        assert cv.schnorr_msm_available(0)
        assert lift_ok == (CURVES[curve]["p"] % 4 == 3)
        n = 333
        it = make_items(curve, n, rng, even_y=True)
        cl, ql, q, p = it["cl"], it["ql"], it["q"], it["p"]
        for k in (None, 1, 3, 8, 64):
            msm_k(k)
            assert cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["R"], 0)
            if lift_ok:
                assert cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["rx"], 1)
            for m in (1, 2, 7, 65):         # short batches: fewer items than lanes, one item
                assert cv.schnorr_verify_all(it["s"][:ql * m], it["ne"][:ql * m], it["Y"][:2 * cl * m], it["R"][:2 * cl * m], 0)
        msm_k(None)
        be = lambda v, w: v.to_bytes(w, "big")
        for idx in (0, 1, n // 2, n - 1):
            s_i = int.from_bytes(it["s"][ql * idx:ql * (idx + 1)], "big")
            bad_s = patched(it["s"], ql, idx, be((s_i + 1) % q, ql))
            assert not cv.schnorr_verify_all(bad_s, it["ne"], it["Y"], it["R"], 0)
            ne_i = int.from_bytes(it["ne"][ql * idx:ql * (idx + 1)], "big")
            assert not cv.schnorr_verify_all(it["s"], patched(it["ne"], ql, idx, be((ne_i + 1) % q, ql)), it["Y"], it["R"], 0)
            # -R instead of R; another item's key; the key of the next item
            Ri = it["R"][2 * cl * idx:2 * cl * (idx + 1)]
            negR = Ri[:cl] + be(p - int.from_bytes(Ri[cl:], "big"), cl)
            assert not cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], patched(it["R"], 2 * cl, idx, negR), 0)
            other = it["Y"][2 * cl * ((idx + 1) % n):2 * cl * ((idx + 1) % n + 1)]
            assert not cv.schnorr_verify_all(it["s"], it["ne"], patched(it["Y"], 2 * cl, idx, other), it["R"], 0)
            # s >= q (the reference's MUST_HAVE(cmp < 0)); a point off the curve; a coordinate >= p
            if q + s_i < (1 << (8 * ql)):
                assert not cv.schnorr_verify_all(patched(it["s"], ql, idx, be(q + s_i, ql)), it["ne"], it["Y"], it["R"], 0)
            off = Ri[:cl] + be((int.from_bytes(Ri[cl:], "big") + 1) % p, cl)
            assert not cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], patched(it["R"], 2 * cl, idx, off), 0)
            assert not cv.schnorr_verify_all(it["s"], it["ne"], patched(it["Y"], 2 * cl, idx, off), it["R"], 0)
            if lift_ok:
                assert not cv.schnorr_verify_all(bad_s, it["ne"], it["Y"], it["rx"], 1)
                # an abscissa without a point (x^3 + a x + b is not a square); an abscissa >= p
                o = Oracle(curve)
                xx = int.from_bytes(Ri[:cl], "big")
                while True:
                    xx = (xx + 1) % p
                    y1, y2, st = cv.y_from_x(be(xx, cl))
                    if st[0] == 1:
                        break
                assert not cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], patched(it["rx"], cl, idx, be(xx, cl)), 1)
                if p + 1 < (1 << (8 * cl)):
                    assert not cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], patched(it["rx"], cl, idx, be(p + 1, cl)), 1)
        # the odd root instead of the even one: as points it is a valid batch only with the matching s; as abscissae the lift picks
        # the even root, so an item made for the odd root is rejected
        it2 = make_items(curve, 40, rng, even_y=False)
        assert cv.schnorr_verify_all(it2["s"], it2["ne"], it2["Y"], it2["R"], 0)
        if lift_ok:
            odd = [i for i in range(40) if it2["R"][2 * cl * i + 2 * cl - 1] & 1]
            assert odd and len(odd) < 40
            assert not cv.schnorr_verify_all(it2["s"], it2["ne"], it2["Y"], it2["rx"], 1)
            keep = [i for i in range(40) if i not in odd]
            sub = lambda b, w: b"".join(b[w * i:w * (i + 1)] for i in keep)
            assert cv.schnorr_verify_all(sub(it2["s"], ql), sub(it2["ne"], ql), sub(it2["Y"], 2 * cl), sub(it2["rx"], cl), 1)
    finally:
        cv.free()


@pytest.mark.parametrize("fold", [2, 4, 8, 16])
def test_bucket_reduction_folds(gpu_ctx, fold, msm_algo):
    """the bucket reduction folds $ECAMD_BKT_FOLD entries per lane and level (8 by default: V = C_0 + f (C_1 + f (...))): every fold gives the same
    verdicts -- a valid batch, an item damaged at either end or in the middle, the cancelling pair of test_combination_is_the_sum_with_the_dumped_z"""
    if msm_algo != "bucket":
        pytest.skip("the Straus evaluation has no bucket reduction")
    old = os.environ.get("ECAMD_BKT_FOLD")
    os.environ["ECAMD_BKT_FOLD"] = str(fold)
    rng = np.random.default_rng(77)
    cv = gpu_ctx.curve("SECP256K1")
    try:
        n = 211
        it = make_items("SECP256K1", n, rng, even_y=True)
        ql, q = it["ql"], it["q"]
        assert cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["R"], 0)
        assert cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["rx"], 1)
        for idx in (0, n // 2, n - 1):
            s_i = int.from_bytes(it["s"][ql * idx:ql * (idx + 1)], "big")
            assert not cv.schnorr_verify_all(patched(it["s"], ql, idx, ((s_i + 1) % q).to_bytes(ql, "big")), it["ne"], it["Y"], it["R"], 0)
        seed = bytes(range(32))
        acc, z, _ = cv.debug_schnorr_msm(it["s"], it["ne"], it["Y"], it["R"], 0, seed)
        assert acc
        zi = [int.from_bytes(z[16 * i:16 * (i + 1)], "little") for i in range(n)]
        d = 0x7654321
        s0 = (int.from_bytes(it["s"][:ql], "big") + d) % q
        s1 = (int.from_bytes(it["s"][ql:2 * ql], "big") - d * zi[0] * pow(zi[1], -1, q)) % q
        s_bad = s0.to_bytes(ql, "big") + s1.to_bytes(ql, "big") + it["s"][2 * ql:]
        acc, zz, _ = cv.debug_schnorr_msm(s_bad, it["ne"], it["Y"], it["R"], 0, seed)
        assert acc and zz == z
        acc, _, _ = cv.debug_schnorr_msm(s_bad, it["ne"], it["Y"], it["R"], 0, bytes(range(1, 33)))
        assert not acc
    finally:
        if old is None:
            os.environ.pop("ECAMD_BKT_FOLD", None)
        else:
            os.environ["ECAMD_BKT_FOLD"] = old
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256K1", "SECP384R1", "SECP256R1"])
def test_combination_is_the_sum_with_the_dumped_z(gpu_ctx, curve):
    """two damaged items whose errors cancel under the z_i of one seed: s_0 += d, s_1 -= d z_0 / z_1 mod q leaves
    sum z_i s_i unchanged, so the combination still vanishes with that seed and with no other"""
    rng = np.random.default_rng(5)
    cv = gpu_ctx.curve(curve)
    try:
        n = 97
        it = make_items(curve, n, rng)
        ql, q = it["ql"], it["q"]
        seed = bytes(range(32))
        acc, z, inf = cv.debug_schnorr_msm(it["s"], it["ne"], it["Y"], it["R"], 0, seed)
        assert acc
        zi = [int.from_bytes(z[16 * i:16 * (i + 1)], "little") for i in range(n)]
        assert all(0 < v < (1 << 128) for v in zi) and len(set(zi)) == n
        acc2, z2, _ = cv.debug_schnorr_msm(it["s"], it["ne"], it["Y"], it["R"], 0, seed)
        assert acc2 and z2 == z                                  # the seed determines the z_i
        d = 0x1234567
        s0 = (int.from_bytes(it["s"][:ql], "big") + d) % q
        s1 = (int.from_bytes(it["s"][ql:2 * ql], "big") - d * zi[0] * pow(zi[1], -1, q)) % q
        s_bad = s0.to_bytes(ql, "big") + s1.to_bytes(ql, "big") + it["s"][2 * ql:]
        acc, zz, _ = cv.debug_schnorr_msm(s_bad, it["ne"], it["Y"], it["R"], 0, seed)
        assert acc and zz == z                                   # cancels under these z_i: the device computed exactly this sum
        acc, zz, _ = cv.debug_schnorr_msm(s_bad, it["ne"], it["Y"], it["R"], 0, bytes(range(1, 33)))
        assert not acc and zz != z
        assert not cv.schnorr_verify_all(s_bad, it["ne"], it["Y"], it["R"], 0)
        # e_i = 0 for every item (ne = 0: the keys' scalars are all zero, their windows all "keep"): s_i = k_i, R_i = [k_i]G
        o = Oracle(curve)
        cl = it["cl"]
        m = 50
        ks = b"".join(((int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1).to_bytes(ql, "big") for _ in range(m))
        Rk, st = o.scalar_mult(ks)
        assert set(st) == {0}
        acc, _, inf = cv.debug_schnorr_msm(ks, bytes(ql) * m, it["Y"][:2 * cl * m], Rk, 0, seed)
        assert acc and inf == 0
        # every scalar of a lane zero: ne = 0 and ... z_i is never zero, so the lanes' sum is infinite only when the R_i cancel:
        # R_1 = -R_0, same z impossible -- instead a batch of ONE item with s = 0, ne = 0, R = anything: T = -[z]R != infinity
        acc, _, inf = cv.debug_schnorr_msm(bytes(ql), bytes(ql), it["Y"][:2 * cl], it["R"][:2 * cl], 0, seed)
        assert not acc and inf == 0
    finally:
        cv.free()


def test_mixed_batch_against_the_item_form(gpu_ctx):
    """a batch with a few bad items: rejected as a whole; the item form (oracle: [s]G + [ne]Y == R) says which, and the batch of the
    remaining items is accepted"""
    rng = np.random.default_rng(77)
    curve = "SECP256K1"
    cv = gpu_ctx.curve(curve)
    o = Oracle(curve)
    try:
        n = 700
        it = make_items(curve, n, rng, even_y=True)
        cl, ql, q = it["cl"], it["ql"], it["q"]
        s = bytearray(it["s"])
        bad = sorted(set(int(v) for v in rng.integers(0, n, size=9)))
        for i in bad:
            s[ql * i + ql - 1] ^= 1
        s = bytes(s)
        assert not cv.schnorr_verify_all(s, it["ne"], it["Y"], it["rx"], 1)
        sG, st = o.scalar_mult(s)
        eY, st2 = o.scalar_mult(it["ne"], it["Y"])
        W, st3 = o.pt_add(sG, eY)
        ok = [st[i] == 0 and st2[i] == 0 and st3[i] == 0 and W[2 * cl * i:2 * cl * (i + 1)] == it["R"][2 * cl * i:2 * cl * (i + 1)] for i in range(n)]
        assert [i for i in range(n) if not ok[i]] == bad
        keep = [i for i in range(n) if ok[i]]
        sub = lambda b, w: b"".join(b[w * i:w * (i + 1)] for i in keep)
        assert cv.schnorr_verify_all(sub(s, ql), sub(it["ne"], ql), sub(it["Y"], 2 * cl), sub(it["rx"], cl), 1)
    finally:
        cv.free()


@pytest.mark.parametrize("devices", [[0, 0, 0], [0] * 8])
def test_sharded_form(gpu_ctx, devices):
    """ecamd_multi_schnorr_verify_all_batch: valid iff every shard is; a bad item in the last shard rejects"""
    rng = np.random.default_rng(78)
    m = libecc_amd.Multi(devices)
    mc = m.curve("SECP256K1")
    try:
        it = make_items("SECP256K1", 203, rng, even_y=True)
        ql = it["ql"]
        assert mc.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["rx"], 1)
        assert mc.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["R"], 0)
        for idx in (0, 100, 202):
            sb = bytearray(it["s"])
            sb[ql * idx + 5] ^= 0x10
            assert not mc.schnorr_verify_all(bytes(sb), it["ne"], it["Y"], it["rx"], 1)
        assert mc.schnorr_verify_all(it["s"][:ql * 5], it["ne"][:ql * 5], it["Y"][:64 * 5], it["rx"][:32 * 5], 1)   # fewer items than ranks
    finally:
        mc.free()
        m.close()


@pytest.mark.parametrize("curve", ["WEI25519", "WEI448"])
def test_curves_with_a_cofactor_are_never_served(gpu_ctx, curve):
    """on a curve with a cofactor the random combination accepts a commitment shifted by a point D of small order
    whenever z_i D = O (probability 1 / ord(D), up to 1/2), which a loop of ec_verify rejects -- so the multi-scalar form declines such
    handles altogether: not available, and a perfectly valid batch comes back "not decided here" (the caller then runs the item form)."""
    rng = np.random.default_rng(77)
    cv = gpu_ctx.curve(curve)
    try:
        assert not cv.schnorr_msm_available(0) and not cv.schnorr_msm_available(1)
        it = make_items(curve, 40, rng)
        assert not cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["R"], 0)
    finally:
        cv.free()


def test_abscissa_beyond_the_limbs_of_the_521_bit_unit_is_rejected(gpu_ctx):
    """secp521r1 runs on 18 limbs of 29 bits (522 bits) while a coordinate has 66 octets (528 bits): r + k 2^522 must
    not lift as r, and a point whose coordinate has one of the six bits above the limbs must not import."""
    rng = np.random.default_rng(78)
    curve = "SECP521R1"
    cv = gpu_ctx.curve(curve)
    try:
        if not cv.schnorr_msm_available(1):
            pytest.skip("no lift_x on this unit")
        it = make_items(curve, 70, rng, even_y=True)
        cl = it["cl"]
        assert cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], it["rx"], 1)
        for idx in (0, 33, 69):
            for bit in (0x04, 0x80):                    # 2^522 and 2^527 of the 66-octet big-endian string
                r = bytearray(it["rx"])
                r[cl * idx] |= bit
                assert not cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], bytes(r), 1)
                R = bytearray(it["R"])
                R[2 * cl * idx] |= bit
                assert not cv.schnorr_verify_all(it["s"], it["ne"], it["Y"], bytes(R), 0)
                Y = bytearray(it["Y"])
                Y[2 * cl * idx + cl] |= bit
                assert not cv.schnorr_verify_all(it["s"], it["ne"], bytes(Y), it["R"], 0)
        # the same strings through the plain multiplication entry point: an import error (status 1), as for any coordinate >= p
        P = bytearray(it["Y"][:2 * cl])
        P[0] |= 0x04
        out, st = cv.scalar_mult(it["s"][:it["ql"]], bytes(P))
        assert st[0] == 1
    finally:
        cv.free()


def _slots(parts, stride):
    """hash-input slots: little-endian u32 length, the bytes, zero padding to `stride`"""
    out = bytearray(stride * len(parts))
    for i, b in enumerate(parts):
        out[stride * i:stride * i + 4] = len(b).to_bytes(4, "little")
        out[stride * i + 4:stride * i + 4 + len(b)] = b
    return bytes(out)


def _projective(aff, cl, p, rng):
    """X || Y || Z with a random Z of the affine points X || Y"""
    out = bytearray()
    for i in range(len(aff) // (2 * cl)):
        x, y = int.from_bytes(aff[2 * cl * i:2 * cl * i + cl], "big"), int.from_bytes(aff[2 * cl * i + cl:2 * cl * (i + 1)], "big")
        z = (int.from_bytes(rand_bytes(rng, cl + 8), "big") % (p - 1)) + 1
        out += (x * z % p).to_bytes(cl, "big") + (y * z % p).to_bytes(cl, "big") + z.to_bytes(cl, "big")
    return bytes(out)


@pytest.fixture(params=["streamed", "after_the_last_chunk"])
def schnorr_stream(request, msm_algo):
    """the bucket evaluation of ec_schnorr_verify_msg_all_batch files every staging chunk as it lands (the default) or runs whole after the
    last chunk ($ECAMD_NO_SCHNORR_STREAM, read at every call); the Straus loop has one form only"""
    if request.param != "streamed" and msm_algo != "bucket":
        pytest.skip("the Straus evaluation always runs after the last chunk")
    old = os.environ.get("ECAMD_NO_SCHNORR_STREAM")
    if request.param == "streamed":
        os.environ.pop("ECAMD_NO_SCHNORR_STREAM", None)
    else:
        os.environ["ECAMD_NO_SCHNORR_STREAM"] = "1"
    yield request.param
    if old is None:
        os.environ.pop("ECAMD_NO_SCHNORR_STREAM", None)
    else:
        os.environ["ECAMD_NO_SCHNORR_STREAM"] = old


@pytest.mark.parametrize("curve,hash_name", [("SECP256K1", "SHA256"), ("SECP256R1", "SHA512"), ("SECP384R1", "SHA384"), ("SECP224R1", "SHA256")])
def test_from_keys_signatures_and_messages(gpu_ctx, curve, hash_name, schnorr_stream):
    """ec_schnorr_verify_msg_all_batch (round 6): the same verdicts from what an application holds -- keys as generated (any y parity; affine,
    or projective with a random Z), signatures r || s / W || s, and the schemes' hash inputs with a blank where the key's x goes; the device
    hashes (SHA-256 / 384 / 512 against hashlib through the items' construction), reduces e mod q -- also when the digest is longer or
    shorter than q --, lifts the key to its even-y representative and runs the multi-scalar form.  Chunked staging (a host chunk of 64
    items) so that the batch-wide arrays are filled piece by piece."""
    import hashlib
    from oracles import HASHLIB, HASH_IDS, make_bip0340_batch
    rng = np.random.default_rng(4242 + len(curve))
    cv = gpu_ctx.curve(curve)
    old_sched = os.environ.get("ECAMD_HOST_SCHEDULE")
    os.environ["ECAMD_HOST_SCHEDULE"] = "64,100"          # three staging chunks: 64, 100, 169 items
    try:
        p, q = CURVES[curve]["p"], CURVES[curve]["q"]
        n = 333
        hid = HASH_IDS[hash_name]
        hl = hashlib.new(HASHLIB[hash_name]).digest_size
        if p % 4 == 3:
            # BIP0340
            it = make_bip0340_batch(lambda sc: cv.scalar_mult(sc), curve, n, rng, msg_len=19, hash_name=hash_name)
            cl, ql = it["cl"], it["ql"]
            tagd = hashlib.new(HASHLIB[hash_name], b"BIP0340/challenge").digest()
            parts = [tagd + tagd + it["rx"][cl * i:cl * (i + 1)] + bytes(cl) + it["msgs"][19 * i:19 * (i + 1)] for i in range(n)]
            stride = (4 + 2 * hl + 2 * cl + 19 + 3) & ~3
            slots = _slots(parts, stride)
            xo = 2 * hl + cl
            for keys, fmt in ((it["pubs"], 0), (_projective(it["pubs"], cl, p, rng), 1)):
                assert cv.schnorr_verify_msg_all(keys, fmt, it["sigs"], 1, hid, slots, stride, xo)
                for k in (0, n // 2, n - 1):
                    bad = bytearray(it["sigs"])
                    bad[(cl + ql) * k + cl + ql - 1] ^= 1                      # s_k
                    assert not cv.schnorr_verify_msg_all(keys, fmt, bytes(bad), 1, hid, slots, stride, xo)
                    bs = bytearray(slots)
                    bs[stride * k + 4 + 2 * hl + 2 * cl] ^= 0x10                 # the message: another e_k
                    assert not cv.schnorr_verify_msg_all(keys, fmt, it["sigs"], 1, hid, bytes(bs), stride, xo)
                # another item's key
                kw = (3 if fmt else 2) * cl
                swapped = keys[kw:2 * kw] + keys[:kw] + keys[2 * kw:]
                assert not cv.schnorr_verify_msg_all(swapped, fmt, it["sigs"], 1, hid, slots, stride, xo)
            # a key at infinity / off the curve: not decided
            prj = bytearray(_projective(it["pubs"], cl, p, rng))
            prj[3 * cl * 5:3 * cl * 6] = bytes(cl) + (1).to_bytes(cl, "big") + bytes(cl)
            assert not cv.schnorr_verify_msg_all(bytes(prj), 1, it["sigs"], 1, hid, slots, stride, xo)
        # ECFSDSA on every curve: e = H(W.x || W.y || m) mod q, s = k + e x, the equation [s]G + [q - e]Y = W
        o = Oracle(curve)
        cl, ql = o.clen, o.qlen
        rnd = lambda: (int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1
        x, k = [rnd() for _ in range(n)], [rnd() for _ in range(n)]
        Y, st = cv.scalar_mult(b"".join(v.to_bytes(ql, "big") for v in x))
        W, st2 = cv.scalar_mult(b"".join(v.to_bytes(ql, "big") for v in k))
        assert set(st) == {0} and set(st2) == {0}
        msgs = rand_bytes(rng, 23 * n)
        sigs, parts = bytearray(), []
        for i in range(n):
            hin = W[2 * cl * i:2 * cl * (i + 1)] + msgs[23 * i:23 * (i + 1)]
            e = int.from_bytes(hashlib.new(HASHLIB[hash_name], hin).digest(), "big") % q
            # sig/ecfsdsa.c:300-330: s = k + e x (libecc's sign convention: the verifier checks [s]G - [e]Y = W)
            sigs += W[2 * cl * i:2 * cl * (i + 1)] + ((k[i] + e * x[i]) % q).to_bytes(ql, "big")
            parts.append(hin)
        stride = (4 + 2 * cl + 23 + 3) & ~3
        slots = _slots(parts, stride)
        for keys, fmt in ((Y, 0), (_projective(Y, cl, p, rng), 1)):
            assert cv.schnorr_verify_msg_all(keys, fmt, bytes(sigs), 0, hid, slots, stride, 0xffffffff)
            bad = bytearray(sigs)
            bad[(2 * cl + ql) * 7 + 2 * cl + 3] ^= 0x20
            assert not cv.schnorr_verify_msg_all(keys, fmt, bytes(bad), 0, hid, slots, stride, 0xffffffff)
    finally:
        if old_sched is None:
            os.environ.pop("ECAMD_HOST_SCHEDULE", None)
        else:
            os.environ["ECAMD_HOST_SCHEDULE"] = old_sched
        cv.free()


@pytest.mark.parametrize("curve", ["SECP256K1", "SECP256R1", "SECP384R1"])
def test_repeated_and_opposite_keys(gpu_ctx, curve):
    """One signer, many messages -- and the signer whose key is the opposite point: every bucket of the round-6 evaluation then holds
    multiples of ONE point, "P + P" and "P + (-P)" are ordinary events (2 048 items, 16-bit windows: hundreds of equal and of opposite
    pairs meet in a bucket), and the additions must double / cancel exactly instead of giving up.  The verdicts are those of distinct keys:
    accepted when valid, not accepted with one damaged item -- on both evaluations."""
    rng = np.random.default_rng(6006)
    o = Oracle(curve)
    c = CURVES[curve]
    q, p = c["q"], c["p"]
    cl, ql = o.clen, o.qlen
    n = 2048
    x0 = (int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1
    xs = [x0 if i % 2 == 0 else q - x0 for i in range(n)]
    Y0, st = o.scalar_mult(x0.to_bytes(ql, "big"))
    assert set(st) == {0}
    Ym = Y0[:cl] + (p - int.from_bytes(Y0[cl:], "big")).to_bytes(cl, "big")
    ks = [(int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1 for _ in range(n)]
    es = [(int.from_bytes(rand_bytes(rng, ql + 8), "big") % (q - 1)) + 1 for _ in range(n)]
    R, st = o.scalar_mult(b"".join(k.to_bytes(ql, "big") for k in ks))
    assert set(st) == {0}
    s = b"".join(((ks[i] + es[i] * xs[i]) % q).to_bytes(ql, "big") for i in range(n))
    ne = b"".join(((q - e) % q).to_bytes(ql, "big") for e in es)
    Y = b"".join(Y0 if i % 2 == 0 else Ym for i in range(n))
    cv = gpu_ctx.curve(curve)
    try:
        assert cv.schnorr_verify_all(s, ne, Y, R, 0)
        for k in (0, 1, n - 1):
            sk = (int.from_bytes(s[ql * k:ql * (k + 1)], "big") + 1) % q
            assert not cv.schnorr_verify_all(patched(s, ql, k, sk.to_bytes(ql, "big")), ne, Y, R, 0)
        # the same commitment twice, and its opposite: items 0 and 1 share R_0 (k_1 := k_0), item 2 carries -R_0 (k_2 := q - k_0)
        ks[1], ks[2] = ks[0], q - ks[0]
        R0 = R[:2 * cl]
        R = R0 + R0 + R0[:cl] + (p - int.from_bytes(R0[cl:], "big")).to_bytes(cl, "big") + R[6 * cl:]
        s = b"".join(((ks[i] + es[i] * xs[i]) % q).to_bytes(ql, "big") for i in range(n))
        assert cv.schnorr_verify_all(s, ne, Y, R, 0)
    finally:
        cv.free()
