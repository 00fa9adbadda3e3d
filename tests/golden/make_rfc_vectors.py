#!/usr/bin/env python3
"""tests/golden/rfc_vectors.json: vectors of RFC 8032 section 7.1 (plain Ed25519; the reference snapshot lacks the header
tests/ed25519_test_vectors.h that its self tests include, tests/ec_self_tests_core.h:4829) and of RFC 7748 section 5.2 (the
iterated X25519 / X448 test, 1 and 1 000 iterations).

The hex strings below are the RFCs' published values (public text, typed in here; RFC 8032's TEST 1024 is left out -- its
1023-byte message is not reproduced).  Because Ed25519 is deterministic, each vector is cross-checked before it is written:
the public key and the signature are recomputed from (secret key, message) by (i) the small Python signer of tests/oracles.py
and (ii) the unmodified reference (oracle/_ref: eddsa_import_key_pair_from_priv_key_buf + ec_sign), and both must give the
published bytes; the iterated X25519 / X448 values are recomputed by running the reference's x25519() / x448() 1 000 times.
Run from the repository root in the authoring container: python tests/golden/make_rfc_vectors.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracles as O  # noqa: E402

ED25519 = [
    ("RFC 8032 7.1 TEST 1", "9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60",
     "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
     "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"),
    ("RFC 8032 7.1 TEST 2", "4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb",
     "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
     "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"),
    ("RFC 8032 7.1 TEST 3", "c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7",
     "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
     "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"),
    ("RFC 8032 7.1 TEST SHA(abc)", "833fe62409237b9d62ec77587520911e9a759cec1d19755b7da901b96dca3d42",
     "ec172b93ad5e563bf4932c70e1245034c35467ef2efd4d64ebf819683467e2bf",
     "ddaf35a193617abacc417349ae20413112e6fa4e89a97ea20a9eeee64b55d39a2192992a274fc1a836ba3c23a3feebbd454d4423643ce80e2a9ac94fa54ca49f",
     "dc2a4459e7369633a52b1bf277839a00201009a3efbf3ecb69bea2186c26b58909351fc9ac90b3ecfdfbc7c66431e0303dca179c138ac17ad9bef1177331a704"),
]
XDH = {
    "x25519": {"len": 32, "start": "0900000000000000000000000000000000000000000000000000000000000000",
               "after_1": "422c8e7a6227d7bca1350b3e2bb7279f7897b87bb6854b783c60e80311ae3079",
               "after_1000": "684cf59ba83309552800ef566f2f4d3c1c3887c49360e3875f2eb94d99532c51"},
    "x448": {"len": 56, "start": "05" + "00" * 55,
             "after_1": "3f482c8a9f19b01e6c46ee9711d9dc14fd4bf67af30765c2ae2b846a4d23a8cd0db897086239492caf350b51f833868b9bc2b3bca9cf4113",
             "after_1000": "aa3b4749d55b9daf1e5b00288826c467274ce3ebbdd5c17b975e09d4af6c67cf10d087202db88286e2b79fceea3ec353ef54faa26e219f38"},
}


def main():
    ed = []
    for name, sk, pk, msg, sig in ED25519:
        skb, mb = bytes.fromhex(sk), bytes.fromhex(msg)
        p, s, _ = O.ed25519_sign(skb, mb)
        assert (p.hex(), s.hex()) == (pk, sig), f"{name}: the Python signer disagrees with the typed-in RFC value"
        rp, rs, st = O.ref_ed25519_sign(skb, mb if mb else b"\0", len(mb))
        assert st == b"\0" and (rp.hex(), rs.hex()) == (pk, sig), f"{name}: the reference disagrees with the typed-in RFC value"
        ed.append({"name": name, "secret_key": sk, "public_key": pk, "message": msg, "signature": sig})
    for name, v in XDH.items():
        k = u = bytes.fromhex(v["start"])
        for it in range(1, 1001):
            out, st = O.ref_xdh(v["len"], k, u)
            assert st == b"\0", f"{name}: the reference rejects iteration {it}"
            k, u = out, k
            if it == 1:
                assert k.hex() == v["after_1"], name
        assert k.hex() == v["after_1000"], name
    json.dump({"source": "RFC 8032 section 7.1, RFC 7748 section 5.2; cross-checked by tests/golden/make_rfc_vectors.py",
               "ed25519": ed, "xdh_iterated": XDH},
              open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rfc_vectors.json"), "w"), indent=1)
    print("wrote rfc_vectors.json:", len(ed), "Ed25519 vectors,", len(XDH), "iterated chains")


if __name__ == "__main__":
    main()
