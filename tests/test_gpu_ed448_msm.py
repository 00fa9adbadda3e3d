"""Ed448 whole-batch verification as one multi-scalar multiplication on the GPU (SURVEY.md section 8, row f-4 for EDDSA448 / EDDSA448PH;
round 6): ec_eddsa_verify_all_batch / ec_eddsa_verify_all_batch_dev on the WEI448 handle -- the decoded keys and commitments as points of
the Weierstrass model, the Schnorr-type combination on the Goldilocks unit (Straus loop and bucket form), a final test cofactored as the
reference's equation is (_eddsa_verify_batch, sig/eddsa.c:2580-2860) -- against the oracle's per-item verdicts and the unmodified
reference's ec_verify_batch over the case families of tests/test_oracle.py (torsion-shifted R and A, small-order keys, undecodable and
non-canonical encodings, S >= q, R = neutral)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import oracles as O  # noqa: E402
from oracles import Oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["straus", "bucket"])
def msm_algo(request):
    """every test runs on both evaluations of the combination (ECAMD_SCHNORR_MSM_ALGO is read by the library at every call)"""
    old = os.environ.get("ECAMD_SCHNORR_MSM_ALGO")
    os.environ["ECAMD_SCHNORR_MSM_ALGO"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("ECAMD_SCHNORR_MSM_ALGO", None)
    else:
        os.environ["ECAMD_SCHNORR_MSM_ALGO"] = old


def raw_verdict(cv, pubs, sigs, hram, stream=None):
    """the combination's own byte (ec_eddsa_verify_all_batch_dev): True = the batch is valid, False = not decided here"""
    import torch
    dev = torch.device("cuda:0")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    dp, ds, dh = t(pubs), t(sigs), t(hram)
    verdict = torch.full((1,), 7, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    cv.eddsa_verify_all_dev(len(pubs) // 57, dp.data_ptr(), ds.data_ptr(), dh.data_ptr(), verdict.data_ptr(), stream)
    torch.cuda.synchronize()
    v = int(verdict.item())
    assert v in (0, 1)
    return v == 0


def test_verdict_on_the_edge_families(gpu_ctx):
    """every subset the per-item oracle accepts is accepted, a subset with one rejected item is rejected with the item's index, and where the
    reference library is here its ec_verify_batch says the same; the combination's own byte vouches for every valid subset whose
    commitments have an affine form (a commitment that decodes to the neutral element is left to the item pass) and for no other"""
    from test_oracle import ed448_cases, eddsa_subset, ED448_MSG_LEN
    rng = np.random.default_rng(172)
    pubs, sigs, msgs, hram = ed448_cases(rng, 10)
    n = len(pubs) // 57
    one = Oracle("WEI448").eddsa_verify(pubs, sigs, hram)
    good = [i for i in range(n) if one[i] == 0]
    bad = [i for i in range(n) if one[i]]
    assert len(good) >= 15 and len(bad) >= 15
    P448 = O.E4_P
    neutral_R = {i for i in good if int.from_bytes(sigs[114 * i:114 * i + 57], "little") in (1, P448 - 1)}
    assert neutral_R and len(neutral_R) < len(good)
    plain = [i for i in good if i not in neutral_R]
    cv = gpu_ctx.curve("WEI448")
    try:
        gpu_ctx.set_eddsa_msm(2, 0, 0)

        def run(idx, ref=False):
            P, S, M, H = eddsa_subset(idx, pubs, sigs, msgs, hram, 57, 114, ED448_MSG_LEN, 114)
            acc, first = cv.eddsa_verify_all(P, S, H)
            exp_first = next((k for k, i in enumerate(idx) if one[i]), len(idx))
            assert first == exp_first and acc == (exp_first == len(idx)), idx
            if ref and O.have_ref():
                assert O.ref_eddsa_verify_all(P, S, M, ED448_MSG_LEN, ed448=True) == acc, idx
            raw = raw_verdict(cv, P, S, H)
            assert raw == (acc and not (set(idx) & neutral_R)), (idx, raw, acc)
            return acc
        assert run(good, ref=True)
        assert run(plain * 5)
        for g in good:
            assert run([g], ref=True), g
        for b in bad:
            assert not run([b]), b
            assert not run(plain[:3] + [b] + plain[3:], ref=True), b
    finally:
        gpu_ctx.set_eddsa_msm(1, 0, 0)
        cv.free()


def make_items(rng, n):
    import hashlib
    pubs, sigs, hram = b"", b"", b""
    for _ in range(n):
        seed = rng.integers(0, 256, size=57, dtype=np.uint8).tobytes()
        msg = rng.integers(0, 256, size=33, dtype=np.uint8).tobytes()
        a, sg, _ = O.ed448_sign(seed, msg)
        pubs += a
        sigs += sg
        hram += hashlib.shake_256(O.ed_dom4(0, b"") + sg[:57] + a + msg).digest(114)
    return pubs, sigs, hram


def test_large_batch(gpu_ctx):
    """2^ECAMD_TEST_MSM_LOG2 (default 17) signatures of 211 signers: accepted by the combination itself; one flipped hash bit anywhere rejects
    and the item-by-item pass names the item; pieces of max_chunk items each carry their own combination and share the verdict byte"""
    import libecc_amd
    log2 = int(os.environ.get("ECAMD_TEST_MSM_LOG2", "17"))
    n = 1 << log2
    rng = np.random.default_rng(15)
    base = 211
    pubs, sigs, hram = make_items(rng, base)
    assert Oracle("WEI448").eddsa_verify(pubs, sigs, hram) == bytes(base)
    reps = (n + base - 1) // base
    P, S, H = (pubs * reps)[:57 * n], (sigs * reps)[:114 * n], bytearray((hram * reps)[:114 * n])
    cv = gpu_ctx.curve("WEI448")
    try:
        gpu_ctx.set_eddsa_msm(1, 0, 0)          # the library's own rule: the batch form from 2^17 items on
        assert cv.eddsa_verify_all(P, S, bytes(H)) == (True, n)
        assert raw_verdict(cv, P, S, bytes(H))
        for idx in (0, n // 3, n - 1):
            H[114 * idx + 5] ^= 4
            assert not raw_verdict(cv, P, S, bytes(H))
            assert cv.eddsa_verify_all(P, S, bytes(H)) == (False, idx)
            H[114 * idx + 5] ^= 4
        ctx2 = libecc_amd.Context(0)
        try:
            ctx2.set_max_chunk(n // 4 + 3)
            ctx2.set_eddsa_msm(2, 0, 0)
            c2 = ctx2.curve("WEI448")
            assert c2.eddsa_verify_all(P, S, bytes(H)) == (True, n)
            assert raw_verdict(c2, P, S, bytes(H))
            for idx in (1, n - 2):                 # the first and the last piece
                H[114 * idx] ^= 1
                assert not raw_verdict(c2, P, S, bytes(H)), idx
                assert c2.eddsa_verify_all(P, S, bytes(H)) == (False, idx)
                H[114 * idx] ^= 1
            c2.free()
        finally:
            ctx2.close()
    finally:
        gpu_ctx.set_eddsa_msm(1, 0, 0)
        cv.free()


def test_one_signer_and_repeated_items(gpu_ctx):
    """batches that fill buckets with multiples of ONE point (one key, the same signature over and over: every addition of a bucket is a doubling
    or meets its own opposite) -- the complete additions decide them exactly"""
    rng = np.random.default_rng(16)
    pubs, sigs, hram = make_items(rng, 2)
    cv = gpu_ctx.curve("WEI448")
    try:
        gpu_ctx.set_eddsa_msm(2, 0, 0)
        for n in (2, 64, 1500):
            P, S, H = pubs[:57] * n, sigs[:114] * n, hram[:114] * n
            assert raw_verdict(cv, P, S, H), n
            Hb = bytearray(H)
            Hb[114 * (n - 1) + 3] ^= 1
            assert not raw_verdict(cv, P, S, bytes(Hb)), n
            assert cv.eddsa_verify_all(P, S, bytes(Hb)) == (False, n - 1)
    finally:
        gpu_ctx.set_eddsa_msm(1, 0, 0)
        cv.free()
