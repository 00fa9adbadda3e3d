#!/usr/bin/env python3
"""Record what the UNMODIFIED reference (oracle/_ref/libecc_ref.so, built from /root/reference by oracle/Makefile) answers on
the crafted ECDSA family of tests/test_oracle.py::ecdsa_crafted_cases, so that the verdicts travel to machines without it.

Run in the authoring container after `make -C oracle ref`:

    python tests/golden/make_crafted.py

Writes tests/golden/ecdsa_crafted.json: per (curve, hash) the public keys, signatures, 24-byte messages and the byte the
reference's ec_pub_key_import_from_aff_buf + ec_verify returned for each item (0 accept / 1 reject).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracles as O  # noqa: E402
import test_oracle as T  # noqa: E402

CASES = [("SECP256R1", "SHA256"), ("SECP256K1", "SHA256"), ("BRAINPOOLP256R1", "SHA256"), ("SECP384R1", "SHA384"),
         ("SECP256R1", "SHA512"), ("SECP521R1", "SHA512")]


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    out = []
    for k, (curve, h) in enumerate(CASES):
        rng = np.random.default_rng(1000 + k)
        pubs, sigs, dgs, hl, exp, msgs = T.ecdsa_crafted_cases(curve, rng, h)
        ref = O.RefLib(curve).ecdsa_verify(h, pubs, sigs, msgs, 24)
        assert ref == exp, (curve, h)
        out.append({"curve": curve, "hash": h, "n": len(exp), "pubs": pubs.hex(), "sigs": sigs.hex(), "msgs": msgs.hex(),
                    "reference_result": ref.hex()})
    with open(os.path.join(HERE, "ecdsa_crafted.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", sum(c["n"] for c in out), "items")


if __name__ == "__main__":
    main()
