"""Wycheproof runner: the harness logic of the reference's wycheproof_tests/libecc_wycheproof.c (ECDSA :74-152, EdDSA :158-,
XDH :278-, ECDH :542-726) over the record types of libecc_wycheproof.h:27-151, fed from the OFFICIAL Wycheproof JSON files
(testvectors/*.json of the Wycheproof project) when they are present.

The reference snapshot does not ship the generated vector header (libecc_wycheproof_tests.h) nor the JSON, and this build
environment has no network: `find_vectors()` looks in $WYCHEPROOF_VECTORS / $WYCHEPROOF_DIR (the directory itself, its testvectors/ and testvectors_v1/), tests/wycheproof/ and
/root/reference/src/wycheproof_tests/ -- when nothing is found the tests that use this module SKIP with the word UNPINNED
instead of passing.  tests/golden/wycheproof_style_selfmade.json is a file in the same schema built from this repository's own
crafted families WITH THE UNMODIFIED REFERENCE'S VERDICTS (tests/golden/make_wycheproof_style.py); it exercises the runner, it is
not Wycheproof.

Verdict rules, as the reference applies them:
  result "valid"       the operation must succeed (and give the expected secret for XDH / ECDH);
  result "invalid"     it must fail;
  result "acceptable"  either outcome is fine -- counted, never an error.
ECDSA signatures come DER encoded in ecdsa_*_test.json: the reference's generator turns them into raw r || s and SKIPS what a
strict parser cannot read (libecc_wycheproof.h:20-22: "does not handle ASN.1 parsing at all"); `der_to_raw` does the same, so a
malformed-DER case is a skip when its expected result is "invalid" and an error otherwise.  *_p1363_test.json files carry raw
r || s already.
"""
import glob
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

CURVE_NAMES = {
    "secp192r1": "SECP192R1", "secp224r1": "SECP224R1", "secp256r1": "SECP256R1", "secp384r1": "SECP384R1",
    "secp521r1": "SECP521R1", "secp256k1": "SECP256K1", "brainpoolP224r1": "BRAINPOOLP224R1",
    "brainpoolP256r1": "BRAINPOOLP256R1", "brainpoolP320r1": "BRAINPOOLP320R1", "brainpoolP384r1": "BRAINPOOLP384R1",
    "brainpoolP512r1": "BRAINPOOLP512R1", "brainpoolP224t1": "BRAINPOOLP224T1", "brainpoolP256t1": "BRAINPOOLP256T1",
    "brainpoolP320t1": "BRAINPOOLP320T1", "brainpoolP384t1": "BRAINPOOLP384T1", "brainpoolP512t1": "BRAINPOOLP512T1",
}
HASH_NAMES = {"SHA-224": "SHA224", "SHA-256": "SHA256", "SHA-384": "SHA384", "SHA-512": "SHA512",
              "SHA3-224": "SHA3_224", "SHA3-256": "SHA3_256", "SHA3-384": "SHA3_384", "SHA3-512": "SHA3_512"}


def find_vectors():
    """directories that hold Wycheproof JSON test-vector files"""
    cands = [os.path.join(HERE, "wycheproof"), "/root/reference/src/wycheproof_tests", "/root/reference/src/wycheproof_tests/testvectors"]
    for var in ("WYCHEPROOF_VECTORS", "WYCHEPROOF_DIR"):   # a checkout of the Wycheproof project, or its testvectors/ directory itself
        root = os.environ.get(var)
        if root:
            cands = [root, os.path.join(root, "testvectors"), os.path.join(root, "testvectors_v1")] + cands
    seen, out = set(), []
    for d in cands:
        if d and d not in seen and glob.glob(os.path.join(d, "*_test.json")):
            seen.add(d)
            out.append(d)
    return out


def load(pattern, dirs=None):
    out = []
    for d in (dirs if dirs is not None else find_vectors()):
        for f in sorted(glob.glob(os.path.join(d, pattern))):
            out.append((os.path.basename(f), json.load(open(f))))
    return out


def der_to_raw(sig, qlen):
    """strict DER SEQUENCE { INTEGER r, INTEGER s } -> r || s on 2 * qlen bytes, or None where a strict parser gives up
    (long-form lengths only where needed, minimal INTEGER encodings, no trailing bytes, values that fit qlen)"""
    def integer(b, pos):
        if pos + 2 > len(b) or b[pos] != 0x02:
            return None
        ln = b[pos + 1]
        if ln & 0x80 or ln == 0 or pos + 2 + ln > len(b):
            return None
        body = b[pos + 2:pos + 2 + ln]
        if body[0] & 0x80:                                   # negative
            return None
        if ln > 1 and body[0] == 0 and not (body[1] & 0x80):  # non-minimal
            return None
        v = int.from_bytes(body, "big")
        return v, pos + 2 + ln

    if len(sig) < 8 or sig[0] != 0x30:
        return None
    pos = 2
    ln = sig[1]
    if ln == 0x81:
        if len(sig) < 3 or sig[2] < 0x80:
            return None
        ln, pos = sig[2], 3
    elif ln & 0x80:
        return None
    if pos + ln != len(sig):
        return None
    r = integer(sig, pos)
    if r is None:
        return None
    s = integer(sig, r[1])
    if s is None or s[1] != len(sig):
        return None
    if r[0] >> (8 * qlen) or s[0] >> (8 * qlen):
        return None
    return r[0].to_bytes(qlen, "big") + s[0].to_bytes(qlen, "big")


class Tally:
    def __init__(self):
        self.performed = self.skipped = self.acceptable_ok = self.acceptable_nok = 0
        self.errors = []

    def judge(self, name, tc, ok):
        self.performed += 1
        res = tc["result"]
        if res == "valid" and not ok:
            self.errors.append(f"{name} tcId {tc['tcId']}: NOK while it must be valid ({tc.get('comment', '')})")
        elif res == "invalid" and ok:
            self.errors.append(f"{name} tcId {tc['tcId']}: OK while it must be invalid ({tc.get('comment', '')})")
        elif res == "acceptable":
            if ok:
                self.acceptable_ok += 1
            else:
                self.acceptable_nok += 1


def run_ecdsa(files, verify, tally=None, digest=None):
    """files: [(name, json)]; verify(curve, hash_name, pubs_affine, sigs_raw, msgs) -> bytes (0 accept / 1 reject) for a list of
    same-curve same-hash items (msgs: list of bytes).  Mirrors check_wycheproof_ecdsa."""
    from oracles import CURVES, qlen as qlen_of, clen as clen_of
    tally = tally or Tally()
    for fname, j in files:
        p1363 = "p1363" in fname or j.get("schema", "").startswith("ecdsa_p1363")
        for g in j["testGroups"]:
            key = g.get("key") or g.get("publicKey") or {}
            curve = CURVE_NAMES.get(key.get("curve"))
            h = HASH_NAMES.get(g.get("sha"))
            if curve is None or h is None or curve not in CURVES:
                tally.skipped += len(g["tests"])
                continue
            cl, ql = clen_of(curve), qlen_of(curve)
            pub = bytes.fromhex(key["wx"]).rjust(cl, b"\0")[-cl:] + bytes.fromhex(key["wy"]).rjust(cl, b"\0")[-cl:]
            items = []
            for tc in g["tests"]:
                raw = bytes.fromhex(tc["sig"])
                if p1363:
                    sig = raw if len(raw) == 2 * ql else None
                else:
                    sig = der_to_raw(raw, ql)
                if sig is None:
                    # the reference's vector generator drops what it cannot turn into r || s; such a case can only be "invalid"
                    # (or "acceptable"), otherwise the parser here is what is wrong
                    if tc["result"] == "valid":
                        tally.errors.append(f"{fname} tcId {tc['tcId']}: valid case that the strict DER reader cannot read")
                    tally.skipped += 1
                    continue
                items.append((tc, sig, bytes.fromhex(tc["msg"])))
            if not items:
                continue
            res = verify(curve, h, pub * len(items), b"".join(s for _, s, _ in items), [m for _, _, m in items])
            for (tc, _, _), r in zip(items, res):
                tally.judge(fname, tc, r == 0)
    return tally


def run_eddsa(files, verify, tally=None):
    """verify(kind, pubs, sigs, msgs) with kind 'Ed25519' / 'Ed448' -> result bytes.  Mirrors check_wycheproof_eddsa."""
    tally = tally or Tally()
    for fname, j in files:
        for g in j["testGroups"]:
            key = g.get("key") or g.get("publicKey") or {}
            kind = {"edwards25519": "Ed25519", "edwards448": "Ed448"}.get(key.get("curve"))
            if kind is None:
                tally.skipped += len(g["tests"])
                continue
            klen, slen = (32, 64) if kind == "Ed25519" else (57, 114)
            pk = bytes.fromhex(key["pk"])
            items = []
            for tc in g["tests"]:
                sig = bytes.fromhex(tc["sig"])
                if len(pk) != klen or len(sig) != slen:
                    # a signature / key of another length: ec_verify fails on the length check
                    tally.judge(fname, tc, False)
                    continue
                items.append((tc, sig, bytes.fromhex(tc["msg"])))
            if not items:
                continue
            res = verify(kind, pk * len(items), b"".join(s for _, s, _ in items), [m for _, _, m in items])
            for (tc, _, _), r in zip(items, res):
                tally.judge(fname, tc, r == 0)
    return tally


def run_xdh(files, xdh, tally=None):
    """xdh(kind, privs, pubs) with kind 'X25519' / 'X448' -> (outputs, status).  Mirrors check_wycheproof_xdh: a failure is fine
    for "acceptable" cases (e.g. public key on the twist), an error for "valid" ones; on success the secret must match."""
    tally = tally or Tally()
    for fname, j in files:
        for g in j["testGroups"]:
            kind = {"curve25519": "X25519", "curve448": "X448"}.get(g.get("curve"))
            if kind is None:
                tally.skipped += len(g["tests"])
                continue
            ln = 32 if kind == "X25519" else 56
            items = [tc for tc in g["tests"] if len(bytes.fromhex(tc["private"])) == ln and len(bytes.fromhex(tc["public"])) == ln]
            tally.skipped += len(g["tests"]) - len(items)
            if not items:
                continue
            out, st = xdh(kind, b"".join(bytes.fromhex(tc["private"]) for tc in items), b"".join(bytes.fromhex(tc["public"]) for tc in items))
            for k, tc in enumerate(items):
                ok = st[k] == 0
                if ok and out[ln * k:ln * (k + 1)] != bytes.fromhex(tc["shared"]):
                    tally.performed += 1
                    tally.errors.append(f"{fname} tcId {tc['tcId']}: shared secret differs")
                    continue
                tally.judge(fname, tc, ok)
    return tally


def run_ecdh_ecpoint(files, derive, decompress, tally=None):
    """ecdh_*_ecpoint_test.json (raw points: 04 || x || y, or 02 / 03 || x).  derive(curve, privs, peers_affine) -> (secrets, status);
    decompress(curve, compressed) -> (affine, status).  Mirrors check_wycheproof_ecdh, with the SEC 1 parity rule for
    compressed points."""
    from oracles import CURVES, qlen as qlen_of, clen as clen_of
    tally = tally or Tally()
    for fname, j in files:
        for g in j["testGroups"]:
            curve = CURVE_NAMES.get(g.get("curve"))
            if curve is None or curve not in CURVES:
                tally.skipped += len(g["tests"])
                continue
            cl, ql = clen_of(curve), qlen_of(curve)
            items = []
            for tc in g["tests"]:
                pub, priv = bytes.fromhex(tc["public"]), bytes.fromhex(tc["private"])
                priv = priv.lstrip(b"\0").rjust(ql, b"\0")
                if len(priv) != ql:
                    tally.skipped += 1
                    continue
                if len(pub) == 2 * cl + 1 and pub[0] == 4:
                    items.append((tc, priv, pub[1:], None))
                elif len(pub) == cl + 1 and pub[0] in (2, 3):
                    items.append((tc, priv, None, pub))
                else:
                    tally.judge(fname, tc, False)       # unusable encoding: the import fails
            comp = [it for it in items if it[3] is not None]
            if comp:
                aff, st = decompress(curve, b"".join(it[3] for it in comp))
                for k, it in enumerate(comp):
                    idx = items.index(it)
                    items[idx] = (it[0], it[1], aff[2 * cl * k:2 * cl * (k + 1)] if st[k] == 0 else None, it[3])
            good = [it for it in items if it[2] is not None]
            for it in items:
                if it[2] is None:
                    tally.judge(fname, it[0], False)
            if not good:
                continue
            sec, st = derive(curve, b"".join(it[1] for it in good), b"".join(it[2] for it in good))
            for k, it in enumerate(good):
                ok = st[k] == 0
                if ok and sec[cl * k:cl * (k + 1)] != bytes.fromhex(it[0]["shared"]).rjust(cl, b"\0")[-cl:]:
                    tally.performed += 1
                    tally.errors.append(f"{fname} tcId {it[0]['tcId']}: shared secret differs")
                    continue
                tally.judge(fname, it[0], ok)
    return tally
