#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors for the hot path into neutral JSON.

Run in the authoring container (reads /root/reference, which does not exist on the GPU box):

    python tests/golden/extract_kats.py

Sources (data only -- byte arrays and the record fields that bind them):
  src/tests/ecccdh_test_vectors.h   125 NIST ECC-CDH KATs (record: ec_self_tests_core.h:55-80)
  src/tests/decdsa_test_vectors.h   RFC 6979 deterministic ECDSA
  src/tests/ec_self_tests_core.h    fixed-k ECDSA (RFC 4754 style nonce callbacks), record :22-52
  src/tests/x25519_test_vectors.h, x448_test_vectors.h   RFC 7748 vectors
  src/tests/ed25519ctx_test_vectors.h, ed25519ph_test_vectors.h   RFC 8032 vectors
  src/tests/ed448_test_vectors.h, ed448ph_test_vectors.h   RFC 8032 vectors
Writes tests/golden/ecccdh_kats.json, ecdsa_kats.json, xdh_kats.json, eddsa_kats.json and eddsa448_kats.json.
"""
import json, os, re, sys

REF = "/root/reference/src/tests"
HERE = os.path.dirname(os.path.abspath(__file__))

ARR = re.compile(r"(?:static\s+)?const\s+u8\s+(\w+)\[\]\s*=\s*\{([^}]*)\}\s*;", re.S)
CASE = re.compile(r"static\s+const\s+(ec_test_case|ecdh_test_case)\s+(\w+)\s*=\s*\{(.*?)\n\};", re.S)
FIELD = re.compile(r"\.(\w+)\s*=\s*(.+?)\s*(?:,\s*\n|,?\s*$)", re.S)
NONCE = re.compile(r"static\s+int\s+(\w+)\(nn_t out, nn_src_t q\)\s*\{(.*?)\n\}", re.S)


def parse_bytes(body):
    return bytes(int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{1,2})", body))


def c_string(lit):
    # adjacent C string literals, with \x.., \n, \" escapes
    out = bytearray()
    for m in re.finditer(r'"((?:[^"\\]|\\.)*)"', lit, re.S):
        s = m.group(1)
        i = 0
        while i < len(s):
            if s[i] == "\\":
                c = s[i + 1]
                if c == "x":
                    j = i + 2
                    while j < len(s) and j < i + 4 and s[j] in "0123456789abcdefABCDEF":
                        j += 1
                    out.append(int(s[i + 2:j], 16))
                    i = j
                    continue
                out.append({"n": 10, "t": 9, "0": 0, "\\": 92, '"': 34, "r": 13}[c])
                i += 2
            else:
                out.append(ord(s[i]))
                i += 1
    return bytes(out)


def load(path):
    src = open(path, encoding="latin-1").read()
    arrays = {m.group(1): parse_bytes(m.group(2)) for m in ARR.finditer(src)}
    nonces = {}
    for m in NONCE.finditer(src):
        inner = ARR.search(m.group(2))
        if inner:
            nonces[m.group(1)] = parse_bytes(inner.group(2))
    cases = []
    for m in CASE.finditer(src):
        fields = {}
        for line in m.group(3).split("\n"):
            fm = re.match(r"\s*\.(\w+)\s*=\s*(.*?),?\s*$", line)
            if fm:
                fields[fm.group(1)] = fm.group(2).strip()
        cases.append((m.group(1), m.group(2), fields))
    return arrays, nonces, cases


def curve_of(f):
    m = re.match(r"&(\w+)_str_params", f.get("ec_str_p", ""))
    return m.group(1).upper() if m else None


def main():
    ecdh, ecdsa = [], []
    arrays, nonces, cases = load(os.path.join(REF, "ecccdh_test_vectors.h"))
    for kind, name, f in cases:
        if kind != "ecdh_test_case" or f.get("ecdh_type") != "ECCCDH":
            continue
        ecdh.append(dict(name=c_string(f["name"]).decode(), curve=curve_of(f),
                         our_priv_key=arrays[f["our_priv_key"]].hex(),
                         peer_pub_key=arrays[f["peer_pub_key"]].hex(),
                         exp_our_pub_key=arrays[f["exp_our_pub_key"]].hex(),
                         exp_shared_secret=arrays[f["exp_shared_secret"]].hex()))
    for fn in ("decdsa_test_vectors.h", "ec_self_tests_core.h"):
        arrays, nonces, cases = load(os.path.join(REF, fn))
        for kind, name, f in cases:
            if kind != "ec_test_case" or f.get("sig_type") not in ("ECDSA", "DECDSA"):
                continue
            msg = f["msg"]
            if msg.startswith('"'):
                msgb = c_string(msg)
            else:
                msgb = arrays[re.sub(r"^\(const char \*\)\s*", "", msg)]
            mlen = f.get("msglen", "")
            if mlen.isdigit():
                msgb = msgb[:int(mlen)]
            nr = f.get("nn_random", "NULL")
            ecdsa.append(dict(name=c_string(f["name"]).decode(), curve=curve_of(f),
                              sig_type=f["sig_type"], hash=f["hash_type"],
                              priv_key=arrays[f["priv_key"]].hex(),
                              k=(nonces[nr].hex() if nr != "NULL" else None),
                              msg=msgb.hex(), exp_sig=arrays[f["exp_sig"]].hex(), source=fn))
    xdh = []
    for fn in ("x25519_test_vectors.h", "x448_test_vectors.h"):
        arrays, nonces, cases = load(os.path.join(REF, fn))
        for kind, name, f in cases:
            if kind != "ecdh_test_case" or f.get("ecdh_type") not in ("X25519", "X448"):
                continue
            xdh.append(dict(name=c_string(f["name"]).decode(), kind=f["ecdh_type"], curve=curve_of(f),
                            our_priv_key=arrays[f["our_priv_key"]].hex(), peer_pub_key=arrays[f["peer_pub_key"]].hex(),
                            exp_our_pub_key=arrays[f["exp_our_pub_key"]].hex(),
                            exp_shared_secret=arrays[f["exp_shared_secret"]].hex()))
    # RFC 8032 Ed25519ctx / Ed25519ph vectors (the snapshot's ed25519_test_vectors.h is absent).  The
    # records hold only the private seed; the public key is exported by the reference itself
    # (oracle/_ref, eddsa_import_key_pair_from_priv_key_buf + eddsa_export_pub_key).
    import ctypes
    ref = ctypes.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libecc_ref.so"))
    eddsa = []
    for fn in ("ed25519ctx_test_vectors.h", "ed25519ph_test_vectors.h"):
        arrays, nonces, cases = load(os.path.join(REF, fn))
        for kind, name, f in cases:
            if kind != "ec_test_case" or f.get("sig_type") not in ("EDDSA25519CTX", "EDDSA25519PH"):
                continue
            msgb = arrays[re.sub(r"^\(const char \*\)\s*", "", f["msg"])]
            seed = arrays[f["priv_key"]]
            ad = f.get("adata", "NULL")
            pub, sig, st = ctypes.create_string_buffer(32), ctypes.create_string_buffer(64), ctypes.create_string_buffer(1)
            assert ref.refdrv_eddsa25519_sign_batch(1, seed, b"", 0, pub, sig, st) == 0 and st.raw == b"\0"
            eddsa.append(dict(name=c_string(f["name"]).decode(), sig_type=f["sig_type"], priv_key=seed.hex(),
                              pub_key=pub.raw.hex(), msg=msgb.hex(),
                              adata=(arrays[ad].hex() if ad != "NULL" else ""),
                              exp_sig=arrays[f["exp_sig"]].hex(), source=fn))
    # RFC 8032 Ed448 / Ed448ph vectors; public keys exported by the reference as above
    eddsa448 = []
    for fn in ("ed448_test_vectors.h", "ed448ph_test_vectors.h"):
        arrays, nonces, cases = load(os.path.join(REF, fn))
        for kind, name, f in cases:
            if kind != "ec_test_case" or f.get("sig_type") not in ("EDDSA448", "EDDSA448PH"):
                continue
            msgb = c_string(f["msg"]) if f["msg"].startswith('"') else arrays[re.sub(r"^\(const char \*\)\s*", "", f["msg"])]
            seed = arrays[f["priv_key"]]
            ad = f.get("adata", "NULL")
            pub, sig, st = ctypes.create_string_buffer(57), ctypes.create_string_buffer(114), ctypes.create_string_buffer(1)
            assert ref.refdrv_eddsa448_sign_batch(1, seed, b"", 0, pub, sig, st) == 0 and st.raw == b"\0"
            eddsa448.append(dict(name=c_string(f["name"]).decode(), sig_type=f["sig_type"], priv_key=seed.hex(),
                                 pub_key=pub.raw.hex(), msg=msgb.hex(),
                                 adata=(arrays[ad].hex() if ad != "NULL" else ""),
                                 exp_sig=arrays[f["exp_sig"]].hex(), source=fn))
    json.dump(eddsa448, open(os.path.join(HERE, "eddsa448_kats.json"), "w"), indent=1)
    print("EDDSA448:", [(c["sig_type"], c["name"]) for c in eddsa448])
    json.dump(eddsa, open(os.path.join(HERE, "eddsa_kats.json"), "w"), indent=1)
    print("EDDSA :", [(c["sig_type"], c["name"]) for c in eddsa])
    json.dump(xdh, open(os.path.join(HERE, "xdh_kats.json"), "w"), indent=1)
    print("XDH   :", [(c["kind"], c["name"]) for c in xdh])
    json.dump(ecdh, open(os.path.join(HERE, "ecccdh_kats.json"), "w"), indent=1)
    json.dump(ecdsa, open(os.path.join(HERE, "ecdsa_kats.json"), "w"), indent=1)
    from collections import Counter
    print("ECCCDH:", Counter(c["curve"] for c in ecdh))
    print("ECDSA :", Counter((c["curve"], c["sig_type"], c["hash"]) for c in ecdsa))


if __name__ == "__main__":
    sys.exit(main())
