#!/usr/bin/env python3
"""Write tests/golden/wycheproof_style_selfmade.json: this repository's crafted ECDSA / EdDSA / X25519 families in the SCHEMA of
the Wycheproof test-vector files (ecdsa_*_p1363_test.json, eddsa_test.json, x25519_test.json), with "result" = the verdict of
the UNMODIFIED reference (already recorded in ecdsa_crafted.json / edge_fixtures.json).  It exercises tests/wycheproof.py; it is
NOT Wycheproof data (the official files are not available in this environment).
    python tests/golden/make_wycheproof_style.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracles import clen, qlen  # noqa: E402

WNAME = {"SECP256R1": "secp256r1", "SECP384R1": "secp384r1", "SECP521R1": "secp521r1", "SECP256K1": "secp256k1",
         "BRAINPOOLP256R1": "brainpoolP256r1", "SECP192R1": "secp192r1", "SECP224R1": "secp224r1"}
WHASH = {"SHA256": "SHA-256", "SHA384": "SHA-384", "SHA512": "SHA-512", "SHA224": "SHA-224"}


def ecdh_files():
    """ecdh_<curve>_ecpoint_test.json files (schema EcdhEcpointTest: "public" = 04 || x || y or 02 / 03 || x, "private", "shared"):
    the families of Wycheproof's ECDH sets -- uncompressed and compressed peers, both roots of a compressed x, x without a point,
    points off the curve, points of ANOTHER curve with the same a (the invalid-curve attack: b + 1, and the quadratic twist),
    coordinates >= p, (0, sqrt(b)), wrong lengths and prefixes, the encoding of infinity, private keys 1 / q - 1 / q / q + 1 -- with
    the verdict and the shared secret of the UNMODIFIED reference: ecccdh_derive_secret on the point its own harness would hand over
    (wycheproof_tests/libecc_wycheproof.c:498-726: raw X || Y, or uncompress_ecc_point = aff_pt_y_from_x and one of the roots)."""
    import numpy as np
    from oracles import CURVES, RefLib, have_ref, ref_y_from_x
    assert have_ref(), "needs oracle/_ref (the unmodified reference)"
    out = {}
    rng = np.random.default_rng(0xECD4)
    for curve in ("SECP256R1", "SECP384R1", "SECP521R1", "BRAINPOOLP256R1", "SECP256K1", "SECP224R1"):
        c = CURVES[curve]
        p, a, b, q = c["p"], c["a"], c["b"], c["q"]
        cl, ql = clen(curve), qlen(curve)
        r = RefLib(curve)
        be = lambda v, n: int(v).to_bytes(n, "big")
        rnd = lambda n: int.from_bytes(rng.bytes(n + 8), "big")

        def roots(x):
            y1, y2, st = ref_y_from_x(curve, be(x, cl))
            return None if st[0] else (int.from_bytes(y1, "big"), int.from_bytes(y2, "big"))

        def on_other_curve(bb):
            while True:
                x = rnd(cl) % p
                t = (x * x * x + a * x + bb) % p
                y = pow(t, (p + 1) // 4, p) if p % 4 == 3 else None
                if y is None:   # p = 1 mod 4 (secp224r1): brute-force through the reference's own square root on a shifted curve is
                    return None  # not available; skip the family there
                if y * y % p == t:
                    return x, y
        cases = []   # (comment, public bytes, private int)
        peers = []
        for i in range(10):
            e = rnd(ql) % (q - 1) + 1
            pt, st = r.scalar_mult(be(e, ql))
            assert st == b"\0"
            peers.append((int.from_bytes(pt[:cl], "big"), int.from_bytes(pt[cl:], "big")))
        d = lambda: rnd(ql) % (q - 1) + 1
        for i, (x, y) in enumerate(peers[:6]):
            cases.append(("uncompressed peer", b"\x04" + be(x, cl) + be(y, cl), d()))
        for i, (x, y) in enumerate(peers[4:10]):
            cases.append(("compressed peer, the root with the prefix's parity", bytes([2 + (y & 1)]) + be(x, cl), d()))
            cases.append(("compressed peer, the other root (same x of the result)", bytes([3 - (y & 1)]) + be(x, cl), d()))
        x = 2
        while roots(x) is not None:
            x += 1
        cases.append(("compressed x without a point on the curve", b"\x02" + be(x, cl), d()))
        cases.append(("compressed x without a point on the curve", b"\x03" + be(x, cl), d()))
        px, py = peers[0]
        cases.append(("point off the curve: y + 1", b"\x04" + be(px, cl) + be((py + 1) % p, cl), d()))
        cases.append(("point off the curve: x and y swapped", b"\x04" + be(py, cl) + be(px, cl), d()))
        for bb, what in (((b + 1) % p, "b + 1"), ((b * 4) % p, "4 b")):
            pt = on_other_curve(bb)
            if pt:
                cases.append((f"point of the curve with {what} (invalid-curve attack)", b"\x04" + be(pt[0], cl) + be(pt[1], cl), d()))
        cases.append(("x = p", b"\x04" + be(p, cl)[-cl:] + be(py, cl), d()))
        cases.append(("y = p + y", b"\x04" + be(px, cl) + (be(p + py, cl + 1)[-cl:] if (p + py) >> (8 * cl) == 0 else b"\xff" * cl), d()))
        cases.append(("all-ones coordinates", b"\x04" + b"\xff" * (2 * cl), d()))
        r0 = roots(0)
        if r0:
            cases.append(("(0, sqrt(b))", b"\x04" + be(0, cl) + be(r0[0], cl), d()))
        cases.append(("(0, 0)", b"\x04" + bytes(2 * cl), d()))
        cases.append(("the encoding of infinity", b"\x00", d()))
        cases.append(("uncompressed point one byte short", (b"\x04" + be(px, cl) + be(py, cl))[:-1], d()))
        cases.append(("uncompressed point one byte long", b"\x04" + be(px, cl) + be(py, cl) + b"\x00", d()))
        cases.append(("prefix 05", b"\x05" + be(px, cl) + be(py, cl), d()))
        for dd, what in ((1, "private key 1"), (q - 1, "private key q - 1"), (2, "private key 2")):
            cases.append((what, b"\x04" + be(px, cl) + be(py, cl), dd))
        tests = []
        for k, (comment, pub, dd) in enumerate(cases):
            priv = be(dd, ql)
            res, shared = "invalid", b""
            aff = None
            if len(pub) == 2 * cl + 1 and pub[0] == 4:
                aff = pub[1:]
            elif len(pub) == cl + 1 and pub[0] in (2, 3):
                x = int.from_bytes(pub[1:], "big")
                rr = roots(x) if x < p else None
                if rr:
                    aff = pub[1:] + be(rr[0], cl)     # either root: only the x coordinate of the result is the secret
            if aff is not None:
                sec, st = r.ecccdh(priv, aff)
                if st[0] == 0:
                    res, shared = "valid", sec
            tests.append({"tcId": k + 1, "comment": comment, "public": pub.hex(), "private": priv.hex(), "shared": shared.hex(), "result": res,
                          "flags": []})
        out[f"ecdh_{WNAME[curve]}_ecpoint_test.json"] = {
            "algorithm": "ECDH", "schema": "ecdh_ecpoint_test_schema.json", "generatorVersion": "selfmade", "numberOfTests": len(tests),
            "notes": {"selfmade": "verdicts and secrets recorded from the unmodified libecc; not Wycheproof data"},
            "testGroups": [{"curve": WNAME[curve], "encoding": "ecpoint", "type": "EcdhEcpointTest", "tests": tests}]}
    return out


def main():
    files = {}
    crafted = json.load(open(os.path.join(HERE, "ecdsa_crafted.json")))
    for fam in crafted:
        curve, h, n = fam["curve"], fam["hash"], int(fam["n"])
        if curve not in WNAME or h not in WHASH:
            continue
        cl, ql = clen(curve), qlen(curve)
        pubs, sigs, msgs, res = (bytes.fromhex(fam[k]) for k in ("pubs", "sigs", "msgs", "reference_result"))
        ml = len(msgs) // n
        groups = {}
        for i in range(n):
            pk = pubs[2 * cl * i:2 * cl * (i + 1)]
            groups.setdefault(pk, []).append(i)
        tg = []
        tc_id = 1
        for pk, idx in groups.items():
            tests = []
            for i in idx:
                tests.append({"tcId": tc_id, "comment": "crafted family item %d" % i, "msg": msgs[ml * i:ml * (i + 1)].hex(),
                              "sig": sigs[2 * ql * i:2 * ql * (i + 1)].hex(), "result": "valid" if res[i] == 0 else "invalid", "flags": []})
                tc_id += 1
            tg.append({"key": {"curve": WNAME[curve], "type": "EcPublicKey", "uncompressed": "04" + pk.hex(), "wx": pk[:cl].hex(),
                               "wy": pk[cl:].hex()}, "sha": WHASH[h], "type": "EcdsaP1363Verify", "tests": tests})
        files[f"ecdsa_{WNAME[curve]}_{h.lower()}_p1363_test.json"] = {
            "algorithm": "ECDSA", "schema": "ecdsa_p1363_verify_schema.json", "generatorVersion": "selfmade", "numberOfTests": tc_id - 1,
            "notes": {"selfmade": "verdicts recorded from the unmodified libecc; not Wycheproof data"}, "testGroups": tg}
    edge = json.load(open(os.path.join(HERE, "edge_fixtures.json")))
    for name, curve, kl in (("ed25519_verify", "edwards25519", 32), ("ed448_verify", "edwards448", 57)):
        f = edge[name]
        pubs, sigs, msgs, res = (bytes.fromhex(f[k]) for k in ("pubs", "sigs", "msgs", "reference_result"))
        ml, n = f["msg_len"], len(res)
        tg = []
        for i in range(n):
            tg.append({"key": {"curve": curve, "keySize": 8 * kl, "pk": pubs[kl * i:kl * (i + 1)].hex(), "type": "EDDSAPublicKey"},
                       "type": "EddsaVerify",
                       "tests": [{"tcId": i + 1, "comment": "edge family item %d" % i, "msg": msgs[ml * i:ml * (i + 1)].hex(),
                                  "sig": sigs[2 * kl * i:2 * kl * (i + 1)].hex(), "result": "valid" if res[i] == 0 else "invalid", "flags": []}]})
        files["eddsa_test.json" if kl == 32 else "ed448_test.json"] = {
            "algorithm": "EDDSA", "schema": "eddsa_verify_schema.json", "generatorVersion": "selfmade", "numberOfTests": n, "testGroups": tg}
    for name, curve, ln in (("x25519", "curve25519", 32), ("x448", "curve448", 56)):
        f = edge[name]
        k, u, out, st = (bytes.fromhex(f[x]) for x in ("k", "u", "reference_out", "reference_status"))
        n = len(st)
        tests = [{"tcId": i + 1, "comment": "edge family item %d" % i, "public": u[ln * i:ln * (i + 1)].hex(),
                  "private": k[ln * i:ln * (i + 1)].hex(), "shared": out[ln * i:ln * (i + 1)].hex(),
                  "result": "valid" if st[i] == 0 else "invalid", "flags": []} for i in range(n)]
        files[f"{name}_test.json"] = {"algorithm": "XDH", "schema": "xdh_comp_schema.json", "generatorVersion": "selfmade", "numberOfTests": n,
                                      "testGroups": [{"curve": curve, "type": "XdhComp", "tests": tests}]}
    files.update(ecdh_files())
    json.dump(files, open(os.path.join(HERE, "wycheproof_style_selfmade.json"), "w"))
    print({k: v["numberOfTests"] for k, v in files.items()})


if __name__ == "__main__":
    main()
