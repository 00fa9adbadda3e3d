#!/usr/bin/env python3
"""Record the UNMODIFIED reference's answers (oracle/_ref/libecc_ref.so) on the edge-case families the test-suite builds
for EdDSA verification, X25519 / X448 and Ed25519 signing, so that these pins travel to machines without the reference.

Run in the authoring container after `make -C oracle ref`:

    python tests/golden/make_edge_fixtures.py

Writes tests/golden/edge_fixtures.json:
  ed25519_verify / ed448_verify  keys, signatures, messages, the hash H(R || A || M) the batch verifiers take, and the byte
                                 eddsa_import_pub_key + ec_verify returned (valid, corrupted, non-canonical, small-order and
                                 torsion-shifted inputs: tests/test_oracle.py::ed25519_cases / ed448_cases)
  x25519 / x448                  scalars, u coordinates, outputs and status of x25519() / x448() (canonical, non-canonical,
                                 twist, small-order inputs: xdh_edge_inputs)
  ed25519_sign                   32-byte seeds, messages, the public keys and signatures of ec_sign (EDDSA25519)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracles as O  # noqa: E402
import test_oracle as T  # noqa: E402


def main():
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    out = {}
    rng = np.random.default_rng(2001)
    pubs, sigs, msgs, hram = T.ed25519_cases(rng, 8)
    ref = O.ref_ed25519_verify(pubs, sigs, msgs, T.ED_MSG_LEN)
    out["ed25519_verify"] = {"msg_len": T.ED_MSG_LEN, "pubs": pubs.hex(), "sigs": sigs.hex(), "msgs": msgs.hex(),
                             "hram": hram.hex(), "reference_result": ref.hex()}
    pubs, sigs, msgs, hram = T.ed448_cases(rng, 6)
    ref = O.ref_ed448_verify(pubs, sigs, msgs, T.ED448_MSG_LEN)
    out["ed448_verify"] = {"msg_len": T.ED448_MSG_LEN, "pubs": pubs.hex(), "sigs": sigs.hex(), "msgs": msgs.hex(),
                           "hram": hram.hex(), "reference_result": ref.hex()}
    for name, ln in (("x25519", 32), ("x448", 56)):
        k, u = T.xdh_edge_inputs(ln, rng)
        ro, rs = O.ref_xdh(ln, k, u)
        out[name] = {"len": ln, "k": k.hex(), "u": u.hex(), "reference_out": ro.hex(), "reference_status": rs.hex()}
    n = 16
    seeds, msgs = T.rb(rng, 32 * n), T.rb(rng, T.ED_MSG_LEN * n)
    rp, rsig, rst = O.ref_ed25519_sign(seeds, msgs, T.ED_MSG_LEN)
    assert rst == bytes(n)
    out["ed25519_sign"] = {"msg_len": T.ED_MSG_LEN, "seeds": seeds.hex(), "msgs": msgs.hex(), "reference_pubs": rp.hex(),
                           "reference_sigs": rsig.hex()}
    with open(os.path.join(HERE, "edge_fixtures.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()
