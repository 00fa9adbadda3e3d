#!/usr/bin/env python3
"""Throughput of the protocol entry points with inputs resident in HBM (the *_dev forms):
BASELINE.json configs[3] (secp256r1 ECDSA batch verification) and configs[4] (Ed25519 verification,
X25519).  Same launch contract and JSON line as bench.py (one process per GPU under
torch.distributed.run, weak scaling, contiguous shards, one RCCL all-gather of the per-rank result
bytes per step); not the driver's headline bench.

    python tools/bench_protocols.py --workload ecdsa_verify|ecdsa_sign|ecccdh|ed25519_verify|ed448_verify|x25519|x448|bip0340_msm|ed25519_msm [--gpus N --steps K --warmup W]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libecc_amd  # noqa: E402

SEED = 0x5EC9256


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=["ecdsa_verify", "ecdsa_sign", "ecccdh", "ed25519_verify", "ed448_verify", "x25519", "x448", "bip0340_msm", "ed25519_msm", "ed448_msm"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--curve", default="SECP256R1", help="ecdsa_verify / ecdsa_sign / ecccdh: any 256-bit prime-order curve libecc names")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ref-items", type=int, default=4096, help="random items of the batch checked against the unmodified reference binary (all host threads)")
    ap.add_argument("--traffic", action="store_true", help="after the timed region: HBM bytes per launch from two rocprofv3 --pmc child runs (tools/pmc.py)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--mad-peak", type=float, default=0.0, help="lane-MADs/s of the v_mad_u64_u32 streams measured by ubench (VGPR multiplier); 0: measure now")
    ap.add_argument("--mad-peak-sgpr", type=float, default=0.0, help="the same with an SGPR multiplier")
    a = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    import oracles as O
    if rank == 0:
        O.build_oracle()
    if dist is not None:
        dist.barrier()
    B = 1 << a.batch_log2
    rng = np.random.default_rng(SEED + rank)
    ctx = libecc_amd.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    def rb(n):
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()

    def t(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    t_setup = time.time()
    ref_subset = work = gate_ref = None
    if a.workload == "ecdsa_verify":
        curve = a.curve
        cv = ctx.curve(curve)
        q = O.CURVES[curve]["q"]
        ql, cl = O.qlen(curve), O.clen(curve)
        hname, hl = ("SHA256", 32) if ql <= 32 else (("SHA384", 48) if ql <= 48 else ("SHA512", 64))
        raw = rng.integers(0, 256, size=(2, B, ql + 8), dtype=np.uint8)

        def scal(rows):
            return b"".join(((int.from_bytes(rows[i].tobytes(), "big") % (q - 1)) + 1).to_bytes(ql, "big") for i in range(B))
        ML = 32
        msgs = rb(ML * B)
        privs, nonces = scal(raw[0]), scal(raw[1])
        hf = getattr(hashlib, hname.lower())
        dg = b"".join(hf(msgs[ML * i:ML * (i + 1)]).digest() for i in range(B))
        pubs, st = cv.scalar_mult(privs)
        assert set(st) == {0}
        sigs, st = cv.ecdsa_sign(privs, nonces, dg, hl)
        assert set(st) == {0}
        sigs = bytearray(sigs)
        bad = np.zeros(B, dtype=np.uint8)
        for i in range(0, B, 10):          # every 10th signature corrupted
            sigs[2 * ql * i + ql + (i % ql)] ^= 1 << (i % 8)
            bad[i] = 1
        sigs = bytes(sigs)
        ins = [t(pubs), t(sigs), t(dg)]
        d_res = torch.empty(B, dtype=torch.uint8, device=dev)

        def step():
            cv.ecdsa_verify_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), hl, d_res.data_ptr(),
                                stream.cuda_stream)
        expected = bad.tobytes()

        def oracle_subset(idx):
            o = O.Oracle(curve)
            return o.ecdsa_verify(b"".join(pubs[2 * cl * i:2 * cl * (i + 1)] for i in idx), b"".join(sigs[2 * ql * i:2 * ql * (i + 1)] for i in idx),
                                  b"".join(dg[hl * i:hl * (i + 1)] for i in idx), hl)

        def ref_subset(idx):
            r = O.RefLib(curve)
            sp, ss, sm = (b"".join(x[w * i:w * i + w] for i in idx) for x, w in ((pubs, 2 * cl), (sigs, 2 * ql), (msgs, ML)))
            return O.join_slices(O.in_slices(lambda lo, hi: r.ecdsa_verify(hname, sp[2 * cl * lo:2 * cl * hi], ss[2 * ql * lo:2 * ql * hi], sm[ML * lo:ML * hi], ML), len(idx)))
        # dominant kernel k_p256_verify_loop<COMB>: the top digit's mixed addition, 64 windows of 4 doublings + 1 mixed addition on Q's
        # table, 17 mixed additions from the comb table of G, the projective x mod q == r test (1 S + 6 M); M = 117, S = 81 MADs
        work = {"kernel": "k_p256_verify_loop<true>", "mads_per_item": 64 * (24 * 117 + 19 * 81) + 18 * (8 * 117 + 3 * 81) + 81 + 6 * 117,
                "sgpr_mads_per_item": 36 * (64 * 43 + 18 * 11 + 7)} if curve == "SECP256R1" else None
        if work:
            # the whole step: k_p256_table (44 M + 27 S) and k_p256_affine (42 M + 7 S + one Fermat inversion, 255 S + 13 M, per 8 items)
            # in front of the loop; k_ecdsa_prep / k_ecdsa_fin (mod-q words on the saturated unit) not counted
            work["step_mads_per_item"] = work["mads_per_item"] + (44 + 42 + 13 / 8) * 117 + (27 + 7 + 255 / 8) * 81
            work["alg_bytes_per_item"] = 2 * cl + 2 * ql + hl + 1
        metric, unit, cfg = "ECDSA verifications/sec (%s, %s digests, batch=2^%d)" % (curve.lower(), hname, a.batch_log2), "verifications/s", 3
    elif a.workload in ("ecdsa_sign", "ecccdh"):
        # secp256r1: signing with caller-supplied nonces (the tail of ec_sign) / ECC-CDH shared secrets
        curve = a.curve
        assert O.CURVES[curve]["p"].bit_length() == 256 and O.CURVES[curve]["q"].bit_length() == 256
        cv = ctx.curve(curve)
        q = O.CURVES[curve]["q"]
        raw = rng.integers(0, 256, size=(2, B, 40), dtype=np.uint8)

        def scal(rows):
            return b"".join(((int.from_bytes(rows[i].tobytes(), "big") % (q - 1)) + 1).to_bytes(32, "big") for i in range(B))
        privs, other = scal(raw[0]), scal(raw[1])
        d_res = torch.empty(B, dtype=torch.uint8, device=dev)
        expected = bytes(B)
        if a.workload == "ecdsa_sign":
            dg = rb(32 * B)
            ins = [t(privs), t(other), t(dg)]
            out_w = 64
            d_out = torch.empty(out_w * B, dtype=torch.uint8, device=dev)

            def step():
                cv.ecdsa_sign_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), 32, d_out.data_ptr(),
                                  d_res.data_ptr(), stream.cuda_stream)

            def oracle_subset(idx):
                cut = lambda b: b"".join(b[32 * i:32 * i + 32] for i in idx)
                return O.Oracle(curve).ecdsa_sign(cut(privs), cut(other), cut(dg), 32)
            metric, unit, cfg = "ECDSA signatures/sec (%s, nonces supplied, batch=2^%d)" % (curve.lower(), a.batch_log2), "signatures/s", 3
        else:
            peers, st = cv.scalar_mult(other)
            assert set(st) == {0}
            ins = [t(privs), t(peers)]
            out_w = 32
            d_out = torch.empty(out_w * B, dtype=torch.uint8, device=dev)

            def step():
                cv.ecccdh_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), d_out.data_ptr(), d_res.data_ptr(), stream.cuda_stream)

            def oracle_subset(idx):
                return O.Oracle(curve).ecccdh(b"".join(privs[32 * i:32 * i + 32] for i in idx),
                                              b"".join(peers[64 * i:64 * i + 64] for i in idx))
            metric, unit, cfg = "ECC-CDH shared secrets/sec (%s, batch=2^%d)" % (curve.lower(), a.batch_log2), "shared-secrets/s", 3
    elif a.workload == "ed25519_verify":
        cv = ctx.curve("WEI25519")
        if a.traffic_child:
            # the PMC passes only need the shape of the work: 512 signatures of the Python signer, tiled
            m = 512
            emsgs = [rb(32) for _ in range(m)]
            items = [O.ed25519_sign(rb(32), emsgs[j]) for j in range(m)]
            reps = B // m
            pubs = b"".join(i[0] for i in items) * reps
            sigs = bytearray(b"".join(i[1] for i in items) * reps)
            hram = b"".join(i[2] for i in items) * reps
            msgs = b"".join(emsgs) * reps
            distinct = "%d distinct signatures tiled" % m
        else:
            # B DISTINCT (key, message, signature) triples: RFC 8032 5.1.5 / 5.1.6 with the hashes on the host and the
            # three scalar-multiplication / mod-q steps on the GPU through the library's own signing entry points
            # (ec_eddsa_sign_R_batch also serves for A = [a]B: it encodes [x mod q]B for any 64-byte x)
            seeds, msgs = rb(32 * B), rb(32 * B)
            hk = [hashlib.sha512(seeds[32 * i:32 * i + 32]).digest() for i in range(B)]
            a_np = np.frombuffer(b"".join(h[:32] for h in hk), dtype=np.uint8).reshape(B, 32).copy()
            a_np[:, 0] &= 248
            a_np[:, 31] &= 127
            a_np[:, 31] |= 64
            a_le = a_np.tobytes()
            wide = np.zeros((B, 64), dtype=np.uint8)
            wide[:, :32] = a_np
            pubs, st = cv.eddsa_sign_R(wide.tobytes())
            assert set(st) == {0}
            r_hash = b"".join(hashlib.sha512(hk[i][32:] + msgs[32 * i:32 * i + 32]).digest() for i in range(B))
            Renc, st = cv.eddsa_sign_R(r_hash)
            assert set(st) == {0}
            hram = b"".join(hashlib.sha512(Renc[32 * i:32 * i + 32] + pubs[32 * i:32 * i + 32] + msgs[32 * i:32 * i + 32]).digest() for i in range(B))
            S = cv.eddsa_sign_S(r_hash, hram, a_le)
            sg = np.empty((B, 64), dtype=np.uint8)
            sg[:, :32] = np.frombuffer(Renc, dtype=np.uint8).reshape(B, 32)
            sg[:, 32:] = np.frombuffer(S, dtype=np.uint8).reshape(B, 32)
            sigs = bytearray(sg.tobytes())
            del hk, wide, sg
            distinct = "all keys, messages and signatures distinct, signed on the GPU by ec_eddsa_sign_R/S_batch"
        bad = np.zeros(B, dtype=np.uint8)
        for i in range(0, B, 10):          # every 10th signature corrupted in S (the hash binds R, A and M, not S)
            sigs[64 * i + 32 + (i % 31)] ^= 1 << (i % 8)
            bad[i] = 1
        sigs = bytes(sigs)
        ins = [t(pubs), t(sigs), t(hram)]
        d_res = torch.empty(B, dtype=torch.uint8, device=dev)

        def step():
            cv.eddsa_verify_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), d_res.data_ptr(), stream.cuda_stream)
        expected = bad.tobytes()

        def oracle_subset(idx):
            o = O.Oracle("WEI25519")
            return o.eddsa_verify(b"".join(pubs[32 * i:32 * i + 32] for i in idx), b"".join(sigs[64 * i:64 * i + 64] for i in idx),
                                  b"".join(hram[64 * i:64 * i + 64] for i in idx))

        def ref_subset(idx):
            sp, ss, sm = (b"".join(x[w * i:w * i + w] for i in idx) for x, w in ((pubs, 32), (sigs, 64), (msgs, 32)))
            return O.join_slices(O.in_slices(lambda lo, hi: O.ref_ed25519_verify(sp[32 * lo:32 * hi], ss[64 * lo:64 * hi], sm[32 * lo:32 * hi], 32), len(idx)))
        if os.environ.get("ECAMD_NO_ED_LATTICE") or os.environ.get("ECAMD_NO_ED_TAIL"):
            # the full-length form (round 4, first step): k_ed_smul_c25519<1>, 64 windows of 3 doublings (3M + 4S), 1 doubling (4M + 4S) and
            # 1 addition without its T (7M); M = 81 + 16 = 97, S = 45 + 16 = 61 MADs, 16 of them with a constant multiplier
            work = {"kernel": "k_ed_smul_c25519<1>", "mads_per_item": 64 * (20 * 97 + 16 * 61), "sgpr_mads_per_item": 64 * 36 * 16}
            # whole step: decode 59 M + 522 S, window table 49 M + 16 S, the loop, k_ed_tail_c25519 162 M + 13 S
            work["step_mads_per_item"] = (59 + 49 + 64 * 20 + 162) * 97 + (522 + 16 + 64 * 16 + 13) * 61
        else:
            # half-length scalars (k_ed_lat): dominant kernel k_ed_smul2_c25519<1, 33>, 33 windows of 3 doublings (3M + 4S), 1 doubling with T
            # (4M + 4S), 1 addition with T (8M) and 1 without (7M) over the tables of A and R
            work = {"kernel": "k_ed_smul2_c25519<1, 33>", "mads_per_item": 33 * (28 * 97 + 16 * 61), "sgpr_mads_per_item": 33 * 44 * 16}
            # the whole step: k_ed_decode_ed_c25519 (two square roots: 2 x (255 S + 25 M), the key's cofactor doublings: 59 M + 522 S),
            # k_ed_lat (word arithmetic mod q and the Euclidean loop: no MADs of this unit), the two window tables (2 x (49 M + 16 S): four
            # doublings with T, three additions, eight entries of one multiplication each), the loop, k_ed_tail2_c25519 (two comb passes
            # of 17 mixed additions, the comparisons, one addition, three doublings: 264 M + 13 S)
            work["step_mads_per_item"] = (59 + 98 + 33 * 28 + 264) * 97 + (522 + 32 + 33 * 16 + 13) * 61
        work["alg_bytes_per_item"] = 32 + 64 + 64 + 1
        metric, unit, cfg = "Ed25519 verifications/sec (batch=2^%d, %s)" % (a.batch_log2, distinct), "verifications/s", 4
    elif a.workload in ("bip0340_msm", "ed25519_msm", "ed448_msm"):
        # the reference's whole-batch verification (ec_verify_batch -> bip0340_verify_batch sig/bip0340.c:1296,
        # eddsa_verify_batch sig/eddsa.c:2904) as ONE multi-scalar multiplication.  A step is one verdict over B VALID signatures resident in
        # HBM (a batch with a bad item comes back "not decided" in the same time and the caller then runs the item form: that path is
        # the ed25519_verify / item-form workload).  Gates: the verdict for the valid batch, the verdict with one item damaged at a random
        # index, and the unmodified reference's own batch function on random contiguous pieces of the same batch (see oracles.py:
        # its Bos-Coster / no-memory verifiers take minutes per 2^17 items on one thread, so the batch is sampled in pieces).
        msm_bad = None
        d_res = torch.full((1,), 7, dtype=torch.uint8, device=dev)
        expected = bytes(1)
        if a.workload == "bip0340_msm":
            curve = "SECP256K1" if a.curve == "SECP256R1" else a.curve
            cv = ctx.curve(curve)
            assert cv.schnorr_msm_available(1)
            if a.traffic_child:
                # the PMC passes only need the shape of the work: valid points and scalars below q; the equation need not hold
                cl, ql = cv.clen, cv.qlen
                raw = rng.integers(0, 256, size=(4, B * ql), dtype=np.uint8)
                raw[:, ::ql] &= 0x7f
                Pk, _ = cv.scalar_mult(raw[0].tobytes())
                Rk, _ = cv.scalar_mult(raw[1].tobytes())
                it = {"s": raw[2].tobytes(), "ne": raw[3].tobytes(), "keys": Pk, "cl": cl, "ql": ql,
                      "rx": np.frombuffer(Rk, dtype=np.uint8).reshape(B, 2 * cl)[:, :cl].tobytes()}
            else:
                it = O.make_bip0340_batch(lambda sc: cv.scalar_mult(sc), curve, B, rng)
            cl, ql = it["cl"], it["ql"]
            ins = [t(it["s"]), t(it["ne"]), t(it["keys"]), t(it["rx"])]
            bad_i = int(rng.integers(0, B))
            bad_s = bytearray(it["s"][ql * bad_i:ql * (bad_i + 1)])
            bad_s[ql - 1] ^= 1

            def step():
                cv.schnorr_verify_all_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), ins[3].data_ptr(), 1, d_res.data_ptr(),
                                          stream.cuda_stream)

            def msm_bad():
                good = ins[0][ql * bad_i:ql * (bad_i + 1)].clone()
                ins[0][ql * bad_i:ql * (bad_i + 1)] = t(bytes(bad_s))
                step()
                torch.cuda.synchronize()
                v = int(d_res.item())
                ins[0][ql * bad_i:ql * (bad_i + 1)] = good
                return v, bad_i

            def ref_pieces(pieces):
                def one(lo, hi):
                    return all(O.ref_sig_verify_all(curve, "BIP0340", "SHA256", it["pubs"][2 * cl * l:2 * cl * h], it["sigs"][(cl + ql) * l:(cl + ql) * h],
                                                    cl + ql, it["msgs"][32 * l:32 * h], 32) for l, h in pieces[lo:hi])
                return all(O.in_slices(one, len(pieces)))
            K = max(1, min(8, B >> 18))
            nl, M, S, red = 9, 101, 65, 20       # secp256k1's flavour (ecamd_u29g.h: 81 + 20 / 45 + 20 MADs)
            if curve != "SECP256K1":
                import bench as _b
                nl, M, S, red = _b.field_mads(O.CURVES[curve]["p"])
            dbl = (3, 4) if O.CURVES[curve]["a"] == 0 else (4, 4)
            # k_msm_loop_g: per lane 64 windows x 4 doublings shared by K items; per item 64 additions of key multiples and 33 of R's
            # (z_i has 128 bits), Jacobian + Jacobian 12M + 4S each
            loop = (256 / K) * (dbl[0] * M + dbl[1] * S) + 97 * (12 * M + 4 * S)
            # k_msm_table_g per item: lift_x of r (a square root: ~ |p| squarings + 12 M) and the import of Y, then 2 x (4 doublings + 3 additions)
            pb = O.CURVES[curve]["p"].bit_length()
            table = (pb * S + 12 * M) + 2 * (4 * (dbl[0] * M + dbl[1] * S) + 3 * (12 * M + 4 * S)) + 8 * M
            work = {"kernel": "k_msm_loop_g", "mads_per_item": loop, "sgpr_mads_per_item": loop * red / M,
                    "step_mads_per_item": loop + table, "alg_bytes_per_item": ql + ql + 2 * cl + cl}
            how = "Straus, K = %d items per lane" % K
            qb = O.CURVES[curve]["q"].bit_length()
            algo = os.environ.get("ECAMD_SCHNORR_MSM_ALGO") or ("bucket" if B >= (1 << 17) else "straus")
            if algo == "bucket" and qb - 16 * ((qb - 1) // 16) >= 12:
                # round 6, the bucket evaluation (ecamd_host.cpp:schnorr_msm_use_buckets): per item 8 ql / 16 windows of the key's scalar and 8 of
                # z_i, one complete addition with an affine operand each (add_aff: 10M + 3S) in k_bkt_accum_g, the dominant kernel; the step
                # adds the import of the two points (the key: on-curve check; r: a square root), and the bucket reduction -- 2^16 buckets
                # per window, 2 Jacobian additions (12M + 4S) each, whatever the batch size
                pairs = (8 * ql + 15) // 16 + 8
                acc = pairs * (10 * M + 3 * S)
                front = (pb * S + 14 * M) + 5 * M + 2 * S
                reduce_ = ((8 * ql + 15) // 16) * 65536 * 2 * (12 * M + 4 * S) / B
                work = {"kernel": "k_bkt_accum_g", "mads_per_item": acc, "sgpr_mads_per_item": acc * red / M,
                        "step_mads_per_item": acc + front + reduce_, "alg_bytes_per_item": ql + ql + 2 * cl + cl}
                how = "buckets, 16-bit windows, %d additions per item" % pairs
            metric, unit, cfg = "BIP0340 signatures/sec in whole-batch verification (%s, one multi-scalar multiplication per 2^%d-item batch: %s)" % (curve.lower(), a.batch_log2, how), "verifications/s", "f4"
            ref_what = "ec_verify_batch (BIP0340: bip0340_verify_batch, no scratch pad)"
        elif a.workload == "ed448_msm":
            # round 6: EDDSA448's batch equation on the Weierstrass model WEI448 -- 2^batch_log2 DISTINCT signatures made here (keys, [r]B and S on
            # the device, SHAKE256 by hashlib), the Schnorr-type combination on the Goldilocks unit with the cofactored final test
            cv = ctx.curve("WEI448")
            dom = O.ed_dom4(0, b"")
            shake = lambda x: hashlib.shake_256(x).digest(114)
            if a.traffic_child:
                pubs, _ = cv.eddsa_sign_R(rb(114 * B))
                Renc, _ = cv.eddsa_sign_R(rb(114 * B))
                sg = np.zeros((B, 114), dtype=np.uint8)
                sg[:, :57] = np.frombuffer(Renc, dtype=np.uint8).reshape(B, 57)
                sg[:, 57:112] = rng.integers(0, 256, size=(B, 55), dtype=np.uint8)
                sigs, hram, msgs = sg.tobytes(), rb(114 * B), b""
            else:
                seeds, msgs = rb(57 * B), rb(32 * B)
                hk = [shake(seeds[57 * i:57 * i + 57]) for i in range(B)]
                a_np = np.frombuffer(b"".join(h[:57] for h in hk), dtype=np.uint8).reshape(B, 57).copy()
                a_np[:, 0] &= 252
                a_np[:, 55] |= 128
                a_np[:, 56] = 0
                wide = np.zeros((B, 114), dtype=np.uint8)
                wide[:, :57] = a_np
                pubs, st = cv.eddsa_sign_R(wide.tobytes())
                assert set(st) == {0}
                r_hash = b"".join(shake(dom + hk[i][57:] + msgs[32 * i:32 * i + 32]) for i in range(B))
                Renc, st = cv.eddsa_sign_R(r_hash)
                assert set(st) == {0}
                hram = b"".join(shake(dom + Renc[57 * i:57 * i + 57] + pubs[57 * i:57 * i + 57] + msgs[32 * i:32 * i + 32]) for i in range(B))
                Sb = cv.eddsa_sign_S(r_hash, hram, a_np.tobytes())
                sg = np.empty((B, 114), dtype=np.uint8)
                sg[:, :57] = np.frombuffer(Renc, dtype=np.uint8).reshape(B, 57)
                sg[:, 57:] = np.frombuffer(Sb, dtype=np.uint8).reshape(B, 57)
                sigs = sg.tobytes()
                del hk, wide, sg
            ins = [t(pubs), t(sigs), t(hram)]
            bad_i = int(rng.integers(0, B))
            ctx.set_eddsa_msm(2, 0, 0)

            def step():
                cv.eddsa_verify_all_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), d_res.data_ptr(), stream.cuda_stream)

            def msm_bad():
                pos = 114 * bad_i + 70
                ins[1][pos] ^= 1
                step()
                torch.cuda.synchronize()
                v = int(d_res.item())
                ins[1][pos] ^= 1
                return v, bad_i

            def ref_pieces(pieces):
                def one(lo, hi):
                    return all(O.ref_eddsa_verify_all(pubs[57 * l:57 * h], sigs[114 * l:114 * h], msgs[32 * l:32 * h], 32, ed448=True) for l, h in pieces[lo:hi])
                return all(O.in_slices(one, len(pieces)))
            import bench as _b
            nl, M, S, red = _b.field_mads(O.CURVES["WEI448"]["p"])
            K = max(1, min(8, B >> 18))
            loop = (448 / K) * (4 * M + 4 * S) + (112 + 33) * (12 * M + 4 * S)
            # the front end: two decodings (a square root and two inversions-by-exponentiation each: ~ 3 x 448 S + 60 M), [4]A, S and h mod q
            front = 2 * (3 * 448 * S + 60 * M) + 2 * (4 * M + 4 * S)
            work = {"kernel": "k_msm_loop_g", "mads_per_item": loop, "sgpr_mads_per_item": loop * red / M, "step_mads_per_item": loop + front,
                    "alg_bytes_per_item": 57 + 114 + 114}
            how = "Straus, K = %d items per lane" % K
            algo = os.environ.get("ECAMD_SCHNORR_MSM_ALGO") or ("bucket" if B >= (1 << 17) else "straus")
            if algo == "bucket":
                pairs = 28 + 8
                acc = pairs * (10 * M + 3 * S)
                reduce_ = 28 * 65536 * 2 * (12 * M + 4 * S) / B
                work = {"kernel": "k_bkt_accum_g", "mads_per_item": acc, "sgpr_mads_per_item": acc * red / M, "step_mads_per_item": acc + front + reduce_,
                        "alg_bytes_per_item": 57 + 114 + 114}
                how = "buckets, 16-bit windows, %d additions per item" % pairs
            metric, unit, cfg = "Ed448 signatures/sec in whole-batch verification (one multi-scalar multiplication per 2^%d-item batch: %s)" % (a.batch_log2, how), "verifications/s", "f4"
            ref_what = "ec_verify_batch (EDDSA448: eddsa_verify_batch, no scratch pad)"
        else:
            cv = ctx.curve("WEI25519")
            if a.traffic_child:
                # the PMC passes only need the shape of the work: encodings that decode, S below q; the equation need not hold
                pubs, _ = cv.eddsa_sign_R(rb(64 * B))
                Renc, _ = cv.eddsa_sign_R(rb(64 * B))
                sg = np.empty((B, 64), dtype=np.uint8)
                sg[:, :32] = np.frombuffer(Renc, dtype=np.uint8).reshape(B, 32)
                sg[:, 32:] = rng.integers(0, 256, size=(B, 32), dtype=np.uint8)
                sg[:, 63] &= 0x0f
                sigs, hram, msgs = sg.tobytes(), rb(64 * B), b""
            else:
                seeds, msgs = rb(32 * B), rb(32 * B)
                hk = [hashlib.sha512(seeds[32 * i:32 * i + 32]).digest() for i in range(B)]
                a_np = np.frombuffer(b"".join(h[:32] for h in hk), dtype=np.uint8).reshape(B, 32).copy()
                a_np[:, 0] &= 248
                a_np[:, 31] &= 127
                a_np[:, 31] |= 64
                wide = np.zeros((B, 64), dtype=np.uint8)
                wide[:, :32] = a_np
                pubs, st = cv.eddsa_sign_R(wide.tobytes())
                assert set(st) == {0}
                r_hash = b"".join(hashlib.sha512(hk[i][32:] + msgs[32 * i:32 * i + 32]).digest() for i in range(B))
                Renc, st = cv.eddsa_sign_R(r_hash)
                assert set(st) == {0}
                hram = b"".join(hashlib.sha512(Renc[32 * i:32 * i + 32] + pubs[32 * i:32 * i + 32] + msgs[32 * i:32 * i + 32]).digest() for i in range(B))
                Sb = cv.eddsa_sign_S(r_hash, hram, a_np.tobytes())
                sg = np.empty((B, 64), dtype=np.uint8)
                sg[:, :32] = np.frombuffer(Renc, dtype=np.uint8).reshape(B, 32)
                sg[:, 32:] = np.frombuffer(Sb, dtype=np.uint8).reshape(B, 32)
                sigs = sg.tobytes()
                del hk, wide, sg
            ins = [t(pubs), t(sigs), t(hram)]
            bad_i = int(rng.integers(0, B))
            ctx.set_eddsa_msm(2, 0, 0)      # the multi-scalar form whatever the size

            def step():
                cv.eddsa_verify_all_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), d_res.data_ptr(), stream.cuda_stream)

            def msm_bad():
                pos = 64 * bad_i + 40
                ins[1][pos] ^= 1
                step()
                torch.cuda.synchronize()
                v = int(d_res.item())
                ins[1][pos] ^= 1
                return v, bad_i

            def ref_pieces(pieces):
                def one(lo, hi):
                    return all(O.ref_eddsa_verify_all(pubs[32 * l:32 * h], sigs[64 * l:64 * h], msgs[32 * l:32 * h], 32) for l, h in pieces[lo:hi])
                return all(O.in_slices(one, len(pieces)))
            K = max(1, min(8, B >> 16))
            M, S = 97, 61
            # k_edmsm_loop: per lane 64 windows of 3 doublings (3M + 4S), 1 doubling with T (4M + 4S) and one addition from B's table (8M),
            # shared by K items; per item 64 additions from A's table and 33 from R's, 8M each
            loop = (64 / K) * (13 * M + 16 * S + 8 * M) + 97 * 8 * M
            # k_edmsm_prep: two decodings (a square root each: 255 S + 25 M), [8]A (3 doublings), two window tables (49 M + 16 S each)
            prep = 2 * (255 * S + 25 * M) + 3 * (4 * M + 4 * S) + 2 * (49 * M + 16 * S)
            work = {"kernel": "k_edmsm_loop", "mads_per_item": loop, "sgpr_mads_per_item": loop * 16 / 97, "step_mads_per_item": loop + prep,
                    "alg_bytes_per_item": 32 + 64 + 64}
            how = "Straus, K = %d items per lane" % K
            algo = os.environ.get("ECAMD_ED_MSM_ALGO") or ("bucket" if B >= (1 << 18) else "straus")
            if algo == "bucket":
                # round 6, the bucket evaluation (ecamd_host.cpp:eddsa_bkt_dev_locked): 16 windows of z h mod q, 8 of z, and a copy of B per 64
                # items with 16 windows: one addition with an affine precomputed operand (ed_madd: 7M) per pair in k_edbkt_accum; the step adds
                # the decoding (two square roots, the key's cofactor doublings, two precomputed entries of 3M), and the reduction -- 2^16
                # buckets per window, 2 unified additions (8M + the operand's conversion 1M) each, whatever the batch size
                pairs = 16 + 8 + 16 / 64
                acc = pairs * 7 * M
                front = 2 * (255 * S + 25 * M) + 3 * (4 * M + 4 * S) + 2 * 3 * M
                reduce_ = 16 * 65536 * 2 * 9 * M / B
                work = {"kernel": "k_edbkt_accum", "mads_per_item": acc, "sgpr_mads_per_item": acc * 16 / 97, "step_mads_per_item": acc + front + reduce_,
                        "alg_bytes_per_item": 32 + 64 + 64}
                how = "buckets, 16-bit windows, %.2f additions per item" % pairs
            metric, unit, cfg = "Ed25519 signatures/sec in whole-batch verification (one multi-scalar multiplication per 2^%d-item batch: %s)" % (a.batch_log2, how), "verifications/s", "f4"
            ref_what = "ec_verify_batch (EDDSA25519: eddsa_verify_batch, no scratch pad)"
    elif a.workload == "ed448_verify":
        cv = ctx.curve("WEI448")
        m = 128
        items = [O.ed448_sign(rb(57), rb(32)) for _ in range(m)]
        reps = B // m
        pubs = b"".join(i[0] for i in items) * reps
        sigs = b"".join(i[1] for i in items) * reps
        hram = bytearray(b"".join(i[2] for i in items) * reps)
        bad = np.zeros(B, dtype=np.uint8)
        for i in range(0, B, 10):
            hram[114 * i + (i % 114)] ^= 1 << (i % 8)
            bad[i] = 1
        hram = bytes(hram)
        ins = [t(pubs), t(sigs), t(hram)]
        d_res = torch.empty(B, dtype=torch.uint8, device=dev)

        def step():
            cv.eddsa_verify_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), ins[2].data_ptr(), d_res.data_ptr(), stream.cuda_stream, 114)
        expected = bad.tobytes()

        def oracle_subset(idx):
            o = O.Oracle("WEI448")
            return o.eddsa_verify(b"".join(pubs[57 * i:57 * i + 57] for i in idx), b"".join(sigs[114 * i:114 * i + 114] for i in idx),
                                  b"".join(hram[114 * i:114 * i + 114] for i in idx))
        metric, unit, cfg = "Ed448 verifications/sec (batch=2^%d, %d distinct signatures tiled)" % (a.batch_log2, m), "verifications/s", "4], Ed448 counterpart [not in BASELINE"
    else:
        x448 = a.workload == "x448"
        xw, xcurve = (56, "WEI448") if x448 else (32, "WEI25519")
        cv = ctx.curve(xcurve)
        k1, k2 = rb(xw * B), rb(xw * B)
        pub, st = cv.xdh(k1, (5 if x448 else 9).to_bytes(xw, "little") * B)   # peers' public keys: u on the curve
        assert set(st) == {0}
        ins = [t(k2), t(pub)]
        d_out = torch.empty(xw * B, dtype=torch.uint8, device=dev)
        d_res = torch.empty(B, dtype=torch.uint8, device=dev)

        def step():
            cv.xdh_dev(B, ins[0].data_ptr(), ins[1].data_ptr(), d_out.data_ptr(), d_res.data_ptr(), stream.cuda_stream)
        expected = bytes(B)

        def oracle_subset(idx):
            o = O.Oracle(xcurve)
            return o.xdh(b"".join(k2[xw * i:xw * i + xw] for i in idx), b"".join(pub[xw * i:xw * i + xw] for i in idx))
        out_w = xw

        def ref_subset(idx):
            sk, su = (b"".join(x[xw * i:xw * i + xw] for i in idx) for x in (k2, pub))
            return O.join_slices(O.in_slices(lambda lo, hi: O.ref_xdh(xw, sk[xw * lo:xw * hi], su[xw * lo:xw * hi]), len(idx)))
        if x448:
            # dominant kernel k_x448_ladder: 448 steps of 5 multiplications, 4 squarings and a24 e as sixteen MADs (round 3: a full
            # product) on the Goldilocks unit (16 limbs of 28 bits: M = 256, S = 136 MADs)
            work = {"kernel": "k_x448_ladder", "mads_per_item": 448 * (5 * 256 + 4 * 136 + 16) + 256, "sgpr_mads_per_item": 448 * 16}
            metric, unit, cfg = "X448 shared secrets/sec (batch=2^%d)" % a.batch_log2, "shared-secrets/s", "5], X448 counterpart [not in BASELINE"
        else:
            # dominant kernel k_x25519_ladder: 255 steps of 5 multiplications, 4 squarings and a24 e as nine MADs (round 3: a full
            # product); M = 81 + 16 = 97, S = 45 + 16 = 61 MADs (ecamd_u29g.h:mul_p25519)
            work = {"kernel": "k_x25519_ladder", "mads_per_item": 255 * (5 * 97 + 4 * 61 + 9) + 97, "sgpr_mads_per_item": 255 * (9 * 16 + 9)}
            # the whole step: k_xdh_prep_c25519 (the on-curve square root and the small-order doublings: 267 S + 35 M), the ladder,
            # k_x25519_fin (one inversion per 8 items: 32 S + 6 M per item)
            work["step_mads_per_item"] = work["mads_per_item"] + (35 + 6) * 97 + (267 + 32) * 61
            work["alg_bytes_per_item"] = 32 + 32 + 32 + 1
            metric, unit, cfg = "X25519 shared secrets/sec (batch=2^%d)" % a.batch_log2, "shared-secrets/s", 4
    gathered = torch.empty(world * d_res.numel(), dtype=torch.uint8, device=dev) if world > 1 else None

    def full_step():
        step()
        if world > 1:
            dist.all_gather_into_tensor(gathered, d_res)

    if a.traffic_child:
        for _ in range(a.warmup + a.steps):
            full_step()
        torch.cuda.synchronize()
        cv.free()
        ctx.close()
        return
    # ---- parity gate: the whole result against what the construction implies, 128 items against the oracle ----
    full_step()
    torch.cuda.synchronize()
    res = d_res.cpu().numpy().tobytes()
    if res != expected:
        raise SystemExit("PARITY FAILURE: accept/reject bits differ from the construction of the batch")
    msm = a.workload in ("bip0340_msm", "ed25519_msm", "ed448_msm")
    if msm:
        v, where = msm_bad()
        if v != 1:
            raise SystemExit("PARITY FAILURE: the whole-batch form accepted a batch with a damaged item")
        full_step()
        torch.cuda.synchronize()
        if int(d_res.item()) != 0:
            raise SystemExit("PARITY FAILURE: the whole-batch form rejected the restored batch")
        gate = f"valid batch of 2^{a.batch_log2} accepted; rejected with item {where} damaged; accepted again once restored"
        if O.have_ref() and rank == 0 and a.ref_items > 0:
            # the unmodified reference's batch function on random contiguous pieces of 256 items of the same batch, one piece list per host thread
            nt = O.host_threads()
            npieces = max(nt, min(a.ref_items, B) // 256)
            starts = sorted(int(x) for x in np.random.default_rng(2).choice(max(1, B // 256), size=min(npieces, max(1, B // 256)), replace=False))
            pieces = [(256 * x, min(B, 256 * x + 256)) for x in starts]
            tr0 = time.time()
            if not ref_pieces(pieces):
                raise SystemExit("PARITY FAILURE: the unmodified reference's batch verifier rejects a piece of the batch the GPU accepted")
            el = time.time() - tr0
            items = sum(h - l for l, h in pieces)
            gate_ref = {"items": items, "seconds": el, "cores": nt, "what": ref_what}
            gate += f"; {len(pieces)} random pieces of 256 items ({items} items) accepted by the unmodified reference's {ref_what} on {nt} threads, {el:.1f} s"
    idx = [int(i) for i in np.random.default_rng(1).choice(B, size=128, replace=False)] if not msm else []
    exp = oracle_subset(idx) if not msm else None
    payload = a.workload in ("x25519", "x448", "ecdsa_sign", "ecccdh")
    if payload:
        out = d_out.cpu().numpy().tobytes()
        got = (b"".join(out[out_w * i:out_w * i + out_w] for i in idx), bytes(res[i] for i in idx))
    elif not msm:
        got = bytes(res[i] for i in idx)
    if not msm and got != exp:
        raise SystemExit("PARITY FAILURE: GPU output differs from the CPU oracle")
    if not msm:
        gate = "all accept/reject bits as constructed; 128 random items identical to the CPU oracle"
    if ref_subset is not None and O.have_ref() and rank == 0 and a.ref_items > 0:
        tg = time.time()
        ridx = [int(i) for i in np.sort(np.random.default_rng(2).choice(B, size=min(B, a.ref_items), replace=False))]
        tr0 = time.time()
        rexp = ref_subset(ridx)
        gate_ref = {"items": len(ridx), "seconds": time.time() - tr0, "cores": O.host_threads()}
        if payload:
            rgot = (b"".join(out[out_w * i:out_w * i + out_w] for i in ridx), bytes(res[i] for i in ridx))
        else:
            rgot = bytes(res[i] for i in ridx)
        if rgot != rexp:
            raise SystemExit("PARITY FAILURE: GPU output differs from the unmodified reference binary")
        gate += f"; {len(ridx)} random items identical to the unmodified reference (oracle/_ref) on {O.host_threads()} threads, {time.time() - tg:.1f} s"
    setup_s = time.time() - t_setup

    for _ in range(a.warmup):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        full_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # roofline of the dominant kernel: one more step with the library's own HIP events around it (on the launch stream)
    roof = None
    if rank == 0 and work is not None:
        try:
            ctx.enable_kernel_timing(True)
            kms = []
            for _ in range(3):
                full_step()
                torch.cuda.synchronize()
                kms.append(ctx.dominant_kernel_ms())
            ctx.enable_kernel_timing(False)
            kernel_ms = float(np.mean(kms))
            pv, ps = a.mad_peak, a.mad_peak_sgpr
            if not pv:
                ub = json.loads(subprocess.run([os.path.join(ROOT, "libecc_amd", "lib", "ubench"), "2000"], capture_output=True, text=True, timeout=120).stdout)
                pv, ps = ub["v_mad_u64_u32"]["lane_ops_per_s"], ub.get("v_mad_u64_u32_sgpr", {}).get("lane_ops_per_s", 0.0)
            fs = work["sgpr_mads_per_item"] / work["mads_per_item"] if ps else 0.0
            peak = 1.0 / ((1.0 - fs) / pv + (fs / ps if ps else 0.0))     # operand-mix weighted stream (VGPR / SGPR multiplier)
            rate = B * work["mads_per_item"] / (kernel_ms * 1e-3)
            roof = {"bound": "valu-int-mad (v_mad_u64_u32 issue)", "kernel": work["kernel"], "kernel_ms": kernel_ms,
                    "kernel_mads_per_item": work["mads_per_item"], "sgpr_multiplier_share": fs, "achieved": rate / 1e9, "peak": peak / 1e9,
                    "unit": "GMAD/s (one GPU)", "frac": rate / peak, "peak_vgpr_stream": pv / 1e9, "peak_sgpr_stream": ps / 1e9,
                    "kernel_share_of_step": kernel_ms / (1e3 * elapsed / a.steps)}
            if work.get("step_mads_per_item"):
                step_ms = 1e3 * elapsed / a.steps
                roof["step_ms"] = step_ms
                roof["mads_per_item"] = work["step_mads_per_item"]
                roof["pipeline_frac"] = (B * work["step_mads_per_item"] / (step_ms * 1e-3)) / peak
                hb = B * work["alg_bytes_per_item"] / (step_ms * 1e-3)
                roof["hbm"] = {"achieved": hb / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": hb / 8e12,
                               "algorithmic_bytes_per_item": work["alg_bytes_per_item"]}
        except Exception as e:   # the timing hook only covers the fast paths
            roof = {"error": str(e)}
    cpu = None
    # HBM bytes per launch from PMC counters: two profiled child runs of this workload (tools/pmc.py), after the timed region
    if rank == 0 and world == 1 and a.traffic and isinstance(roof, dict) and "kernel" in roof:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc
        tt = time.time()
        child = [sys.executable, os.path.abspath(__file__), "--workload", a.workload, "--curve", a.curve, "--batch-log2", str(a.batch_log2),
                 "--traffic-child", "--steps", "2", "--warmup", "1"]
        by_kernel, note = pmc.hbm_bytes_per_launch(child, timeout=300)
        roof["traffic_by_kernel"] = by_kernel
        dom = [v for k, v in (by_kernel or {}).items() if roof["kernel"].split("<")[0] in k]
        roof["traffic"] = max(dom) if dom else None
        roof["traffic_note"] = f"{note}; {time.time() - tt:.0f} s"
        if by_kernel and work.get("alg_bytes_per_item"):
            roof["step_traffic"] = sum(by_kernel.values())
            roof["traffic_over_algorithmic"] = roof["step_traffic"] / (B * work["alg_bytes_per_item"])
    if rank == 0 and world == 1 and not a.no_cpu_baseline and gate_ref and (gate_ref["seconds"] >= 3.0 or msm):
        # the parity gate already ran the unmodified reference over a random subset of this batch on every host thread: that run
        # IS the CPU baseline of the workload (ec_verify / x25519() of the reference beside configs 3-5)
        what = {"ecdsa_verify": "ec_pub_key_import_from_aff_buf + ec_verify (ECDSA)", "ecdsa_sign": "ec_sign (ECDSA, nonce supplied)",
                "ecccdh": "ecccdh_derive_secret", "ed25519_verify": "eddsa_import_pub_key + ec_verify (EDDSA25519)",
                "ed448_verify": "eddsa_import_pub_key + ec_verify (EDDSA448)", "x25519": "x25519()", "x448": "x448()",
                "bip0340_msm": gate_ref.get("what"), "ed25519_msm": gate_ref.get("what"), "ed448_msm": gate_ref.get("what")}[a.workload]
        cpu = {"value": gate_ref["items"] / gate_ref["seconds"], "unit": unit, "cores": gate_ref["cores"], "kind": "reference",
               "sample": f"the parity gate's own run: {gate_ref['items']} random items of the same batch through {what} of the unmodified "
                         f"reference (oracle/_ref) on {gate_ref['cores']} threads, {gate_ref['seconds']:.1f} s wall"}
    elif rank == 0 and world == 1 and not a.no_cpu_baseline and O.have_ref() and not msm:
        # the unmodified reference on this host, one thread, on a bounded sample of the same inputs
        m = 1536 if a.workload != "x25519" else 3072
        if a.workload == "x448":
            m = 512
        t0 = time.time()
        if a.workload == "ecdsa_verify":
            # ec_verify hashes the message itself: time it on messages of the digest's length (SHA-256 of 32 bytes
            # is noise next to the two scalar multiplications); accept bits are not compared here
            O.RefLib(curve).ecdsa_verify(hname, pubs[:2 * cl * m], sigs[:2 * ql * m], msgs[:32 * m], 32)
            what = "ec_pub_key_import_from_aff_buf + ec_verify (ECDSA, SHA-256 over 32-byte messages)"
        elif a.workload == "ecdsa_sign":
            O.RefLib(curve).ecdsa_sign("SHA256", privs[:32 * m], other[:32 * m], dg[:32 * m], 32)
            what = "ec_key_pair_import_from_priv_key_buf + ec_sign (ECDSA, SHA-256 over 32-byte messages, nonce supplied)"
        elif a.workload == "ecccdh":
            O.RefLib(curve).ecccdh(privs[:32 * m], peers[:64 * m])
            what = "ecccdh_derive_secret"
        elif a.workload == "ed25519_verify":
            O.ref_ed25519_verify(pubs[:32 * m], sigs[:64 * m], msgs[:32 * m], 32)
            what = "eddsa_import_pub_key + ec_verify (EDDSA25519, 32-byte messages)"
        elif a.workload == "ed448_verify":
            m = 256
            O.ref_ed448_verify(pubs[:57 * m], sigs[:114 * m], hram[:114 * m], 114)
            what = "eddsa_import_pub_key + ec_verify (EDDSA448, SHAKE256 over 114-byte messages)"
        else:
            O.ref_xdh(xw, k2[:xw * m], pub[:xw * m])
            what = "x448()" if x448 else "x25519()"
        el = time.time() - t0
        cpu = {"value": m / el, "unit": unit, "cores": 1, "kind": "reference",
               "sample": f"first {m} items of the same batch through {what} of the unmodified reference (oracle/_ref), "
                         f"1 thread, {el:.1f} s"}
    if rank == 0:
        print(json.dumps({
            "metric": metric, "value": B * world * a.steps / elapsed, "unit": unit, "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 (%d-bit limbs, v_mad_u64_u32 integer MAD, u64 accumulators)" % (28 if a.workload in ("x448", "ed448_verify", "ed448_msm") else 29),
            "data": ("synthetic (seeded), inputs resident in HBM; every signature valid (the form vouches for valid batches)" if msm else
                     "synthetic (seeded), inputs resident in HBM; 10 % of the signatures corrupted") if not payload
                    else "synthetic (seeded), inputs resident in HBM; valid keys",
            "config": {"workload": (f"{a.workload} (SURVEY.md section 8 row f4), batch 2^{a.batch_log2} per GPU" if msm else
                                    f"{a.workload} (BASELINE.json configs[{cfg}]), batch 2^{a.batch_log2} per GPU"),
                       "sharding": "contiguous per-rank shards" + (", RCCL all_gather of result bytes per step" if world > 1 else ""),
                       "parity_gate": gate},
            "roofline": roof, "cpu_baseline": cpu, "setup_s": setup_s}))
    cv.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
