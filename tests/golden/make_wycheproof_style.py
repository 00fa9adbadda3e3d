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
    json.dump(files, open(os.path.join(HERE, "wycheproof_style_selfmade.json"), "w"))
    print({k: v["numberOfTests"] for k, v in files.items()})


if __name__ == "__main__":
    main()
