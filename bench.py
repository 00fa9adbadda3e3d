#!/usr/bin/env python3
"""bench.py -- scalar-mults/sec of the batched prj_pt_mul hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
torch.distributed.run with one rank per GPU (RCCL).  A "step" is one pass of the hot path over
one batch: B = 2^20 secp256r1 variable-base scalar multiplications per GPU (BASELINE.json
configs[1]), scalars uniform in [1, q-1], base points P_i = [t_i]G, affine X||Y in / out --
inputs resident in HBM before the timed region.  For N > 1 the batch is sharded by rank (weak
scaling: every rank owns 2^20 items) and the step ends with one RCCL all-gather of the
output points, as north_star specifies.

Rank 0 prints ONE JSON line with the driver's fields plus
  roofline     executed 32x32 multiply-accumulates per second against the v_mad_u64_u32 issue
               peak MEASURED on this device by libecc_amd/lib/ubench, and
               the HBM view (algorithmic bytes / s against 8 TB/s) beside it;
  cpu_baseline the reference's own prj_pt_mul + prj_pt_unique (oracle/_ref, "reference") or the
               C restatement ("port") timed on this box's host cores on a bounded sample.
Before anything is timed a random subset of the GPU output is compared byte-for-byte with the
CPU oracle; a mismatch aborts the run.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libecc_amd  # noqa: E402

CURVE = "SECP256R1"
SEED = 0x5EC9256


def popcount(x):
    return bin(x).count("1")


def field_mads(p):
    """(limbs, MADs per multiplication, MADs per squaring, of which with a wave-uniform multiplier) of the generic radix-2^29 field
    code for the prime p (ecamd_u29g.h; tests/test_u29g_host.py counts the MADs the host build of that header really executes)."""
    pbits = p.bit_length()
    # dense Montgomery: NL = ceil((|p| + 16) / 29) limbs, NL^2 products + NL^2 reduction MADs (the digits of p sit in SGPRs);
    # squaring NL (NL + 1) / 2 products + NL^2 reduction MADs
    nl = (pbits + 16 + 28) // 29
    M, S, red = 2 * nl * nl, nl * (nl + 1) // 2 + nl * nl, nl * nl
    if p == 2**521 - 1:                              # secp521r1: plain residues on 18 limbs, 2^522 = 2 folded inside the columns
        nl = 18
        M, S, red = nl * nl, nl * (nl + 1) // 2, 0
    if p == 2**384 - 2**128 - 2**96 + 2**32 - 1:     # secp384r1: m_k (p + 1) as four signed MADs per quotient digit
        M, S, red = nl * nl + 4 * nl, nl * (nl + 1) // 2 + 4 * nl, 4 * nl
    if p == 2**224 - 2**96 + 1 or p == 2**192 - 2**64 - 1:   # secp224r1 / secp192r1: two signed MADs per quotient digit
        M, S, red = nl * nl + 2 * nl, nl * (nl + 1) // 2 + 2 * nl, 2 * nl
    if p == 2**255 - 19:                             # 2^255 - 19: 9 limbs; the eight high columns fold as their two register halves,
        nl = 9                                       # 2 x 8 MADs with wave-uniform multipliers riding in the low columns (round 4)
        M, S, red = nl * nl + 16, nl * (nl + 1) // 2 + 16, 16
    if p == 2**448 - 2**224 - 1:                     # Goldilocks: 16 limbs of 28 bits, phi^2 = phi + 1 folded inside the columns;
        M, S, red = nl * nl, 2 * 36 + 64, 0          # squaring a0^2 + a1^2 (36 each) and a1 (2 a0 + a1) (64), ecamd_u29g.h:mul_p448
    if p == 2**256 - 2**32 - 977:                    # secp256k1: 9 limbs, 9 + 8 + 2 = 19 fold MADs riding in the low columns + 977 q
        nl = 9
        M, S, red = nl * nl + 20, nl * (nl + 1) // 2 + 20, 20
    return nl, M, S, red


def work_model(curve_params, nw, slen, batch=1 << 20):
    """Field multiplications and 32x32 MADs (v_mad_u64_u32) executed per item, derived from the
    kernels' own parameters.  secp256r1 takes the hand-specialised radix-2^29 Jacobian path
    (ecamd_p256_kernel.hip: multiplication 81 product + 36 reduction MADs, squaring 45 + 36), every
    other curve the generic radix-2^29 path (ecamd_g29_kernel.hip), scalars longer than the field the
    saturated complete-formula kernel k_smul<NW>."""
    p = curve_params["p"]
    pbits = p.bit_length()
    nwin = 2 * slen
    if p == 2**256 - 2**224 + 2**192 + 2**96 - 1 and slen <= 32:
        M, S = 117, 81
        dbl, madd = (4, 4), (8, 3)                       # (mults, squarings); mixed addition: affine table
        fin_k = aff_k = 8 if batch >= (1 << 19) else (4 if batch >= (1 << 17) else 2)   # items per lane sharing one inversion (p256_items_per_inversion)
        # k_p256_table: to Montgomery 2M, on-curve 2M+2S, 4 dbl + 3 madd
        nm = 2 + 2 + 4 * dbl[0] + 3 * madd[0]
        ns = 2 + 4 * dbl[1] + 3 * madd[1]
        # k_p256_affine: 7 entries x (prefix 1M, back-substitution 2M, affine 3M + 1S) + (255S + 13M)/aff_k
        nm += 7 * 6 + 13 / aff_k
        ns += 7 * 1 + 255 / aff_k
        # k_p256_loop: nwin x (4 dbl + 1 madd)
        nm += nwin * (4 * dbl[0] + madd[0])
        ns += nwin * (4 * dbl[1] + madd[1])
        # k_p256_finalize: prefix 1M, (255S + 13M)/fin_k, back-substitution 2M, affine 1S + 3M, from Montgomery 2M
        nm += 1 + 13 / fin_k + 2 + 3 + 2
        ns += 255 / fin_k + 1
        loop_mads = nwin * ((4 * dbl[0] + madd[0]) * M + (4 * dbl[1] + madd[1]) * S)
        # 36 of a multiplication's 117 (a squaring's 81) MADs are the reduction products m_k x (p + 1)'s four digits: an SGPR
        # multiplier (ecamd_u29.h), which issues faster than the VGPR x VGPR form (ubench: v_mad_u64_u32_sgpr)
        work_model.loop_sgpr_share = 36.0 * (4 * dbl[0] + madd[0] + 4 * dbl[1] + madd[1]) * nwin / loop_mads
        return nm + ns, nm * M + ns * S, "k_p256_loop", loop_mads
    fin_k = 8
    work_model.loop_sgpr_share = 0.0
    if slen <= 4 * ((pbits + 31) // 32):
        # generic radix-2^29 Jacobian kernels k_smul_g<|p|> + k_finalize_g<|p|> (ecamd_g29_kernel.hip):
        # NL = ceil((|p| + 16) / 29) limbs, multiplication NL^2 products + NL^2 reduction MADs,
        # squaring NL (NL + 1) / 2 products + NL^2 reduction MADs
        nl, M, S, red = field_mads(p)
        am3 = curve_params["a"] == p - 3 or iso_to_am3(p, curve_params["a"])
        dbl = (3, 4) if curve_params["a"] == 0 else ((4, 4) if am3 else (4, 6))
        add = (12, 4)
        # jacg::inv: 2-bit windows over p - 2 with the table x, x^2, x^3
        e, top = p - 2, (pbits - 1) | 1
        inv_s = top + 1 + 1
        inv_m = 1 + sum(1 for i in range(top, 0, -2) if (e >> (i - 1)) & 3)
        if p == 2**256 - 2**32 - 977 or p == 2**255 - 19:
            # the two nine-limb flavours keep the one-kernel Jacobian-table path k_smul_g (ecamd_g29_kernel.hip, launcher)
            nm = 2 + 2 + 4 * dbl[0] + 3 * add[0] + 7 + nwin * (4 * dbl[0] + add[0]) + 1   # import, table (+7 Y normalisations), loop, Z test
            ns = 2 + 4 * dbl[1] + 3 * add[1] + nwin * (4 * dbl[1] + add[1])
            nm += 1 + inv_m / fin_k + 2 + 3 + 2
            ns += inv_s / fin_k + 1
            loop_mads = (nm - (1 + inv_m / fin_k + 2 + 3 + 2)) * M + (ns - (inv_s / fin_k + 1)) * S
            return nm + ns, nm * M + ns * S, f"k_smul_g<{pbits}>", loop_mads
        # affine-table pipeline k_table_g / k_affine_g / k_loop_g / k_finalize_g (round 2)
        madd = (8, 3)
        aff_k = 8 if batch >= (1 << 19) else (4 if batch >= (1 << 17) else 2)
        nm = 2 + 2 + 4 * dbl[0] + 3 * add[0] + 2                    # import, on-curve, 2P..8P, exact test of the last Z's
        ns = 2 + 4 * dbl[1] + 3 * add[1]
        nm += 7 * 6 + inv_m / aff_k                                  # per entry: prefix 1M, back-substitution 2M, Z^-3 1M, x 1M, y 1M
        ns += 7 * 1 + inv_s / aff_k                                  # + Z^-2 1S; one Fermat inversion per aff_k items
        loop_m, loop_s = nwin * (4 * dbl[0] + madd[0]) + 1, nwin * (4 * dbl[1] + madd[1])   # + the exact test of the final Z
        nm += loop_m
        ns += loop_s
        nm += 1 + inv_m / fin_k + 2 + 3 + 2
        ns += inv_s / fin_k + 1
        # the reduction MADs multiply by digits of p held in __constant__ memory (scalar registers)
        work_model.loop_sgpr_share = (loop_m + loop_s) * red / (loop_m * M + loop_s * S)
        return nm + ns, nm * M + ns * S, f"k_loop_g<{pbits}>", loop_m * M + loop_s * S
    mm_add, mm_dbl = 17, 16                              # RCB Alg. 1 / Alg. 3, generic a
    mm = 2 + 3                                           # to Montgomery (x, y) + on-curve check
    mm += 14 * mm_add                                    # table [2..15]P
    mm += (nwin - 1) * (4 * mm_dbl + mm_add)             # windows
    mm += pbits + popcount(p - 2)                        # Fermat inversion (square-and-multiply)
    mm += 2 + 2                                          # X/Z, Y/Z, from Montgomery
    mads_per_mm = 2 * nw * nw + nw                       # FIPS Montgomery multiplication, 32-bit words
    return mm, mm * mads_per_mm, f"k_smul<{nw}>", mm * mads_per_mm


def iso_to_am3(p, a):
    """does ecamd_host.cpp:upload_g29 move the curve onto an isomorphic one with a = -3 (u^4 a = -3, p = 3 mod 4)?"""
    if p % 4 != 3 or a == 0 or p == 2**255 - 19:
        return False
    t = (-3 * pow(a, p - 2, p)) % p
    s1 = pow(t, (p + 1) // 4, p)
    if s1 * s1 % p != t:
        return False
    return any(pow(c, (p + 1) // 4, p) ** 2 % p == c for c in (s1, p - s1))


def ref_equiv_mads():
    """the reference algorithm's work per P-256 scalar mult (W_ref)"""
    return 8724 * 136


def measured_mad_peak():
    exe = os.path.join(ROOT, "libecc_amd", "lib", "ubench")
    try:
        out = subprocess.run([exe, "2000"], capture_output=True, text=True, timeout=120).stdout
        j = json.loads(out)
        return j["v_mad_u64_u32"]["lane_ops_per_s"], j
    except Exception:
        return None, None


def mix_peak(ub, sgpr_share, signed=False):
    """The MAD issue ceiling for a kernel whose MADs are `sgpr_share` SGPR-multiplier and the rest VGPR-multiplier: the two
    measured streams weighted by their share of the instruction count (issue time adds up).
    signed: the SGPR-multiplier MADs are v_mad_i64_i32 (secp384r1's reduction) -- its own measured stream when ubench has it."""
    pv = ub["v_mad_u64_u32"]["lane_ops_per_s"]
    ps = ub.get("v_mad_u64_u32_sgpr", {}).get("lane_ops_per_s")
    if signed:
        ps = ub.get("v_mad_i64_i32_sgpr", {}).get("lane_ops_per_s") or ps
    if not ps or sgpr_share <= 0.0:
        return pv
    return 1.0 / ((1.0 - sgpr_share) / pv + sgpr_share / ps)


def secondary_lines(ub):
    """BASELINE.json configs[2]-[4] as records behind the headline line, each with the keys of the headline (VERDICT round 3,
    item 2): secp384r1 and secp521r1 scalar multiplication, secp256r1 ECDSA verification, Ed25519 verification, X25519 -- a fresh
    process per workload at 2^20 items, 10 timed steps, a 2^16-item gate against the unmodified reference binary on every host
    thread (whose own wall time is the workload's `cpu_baseline`: ec_verify / x25519() / prj_pt_mul of the reference on the same
    inputs), the dominant kernel's and the whole step's fraction of the mix-weighted MAD stream, and the PMC HBM bytes per launch."""
    env = dict(os.environ)
    pv = ub["v_mad_u64_u32"]["lane_ops_per_s"] if ub else 0.0
    ps = (ub.get("v_mad_u64_u32_sgpr", {}).get("lane_ops_per_s", 0.0)) if ub else 0.0
    common = ["--steps", "10", "--warmup", "2"]
    jobs = [("configs[2] secp384r1", [sys.executable, os.path.abspath(__file__), "--curve", "SECP384R1"]),
            ("configs[2] secp521r1", [sys.executable, os.path.abspath(__file__), "--curve", "SECP521R1"])]
    jobs = [(n, c + ["--no-secondary", "--parity-items", "65536"] + common) for n, c in jobs]
    tool = os.path.join(ROOT, "tools", "bench_protocols.py")
    for name, w in (("configs[3] ECDSA verify secp256r1", "ecdsa_verify"), ("configs[4] Ed25519 verify", "ed25519_verify"),
                    ("configs[4] X25519", "x25519")):
        jobs.append((name, [sys.executable, tool, "--workload", w, "--ref-items", "65536", "--traffic", "--mad-peak", repr(pv),
                            "--mad-peak-sgpr", repr(ps)] + common))
    # the whole-batch verification as one multi-scalar multiplication, with the
    # unmodified reference's own batch verifier (ec_verify_batch -> bip0340_verify_batch / eddsa_verify_batch) on pieces of the same batch
    # as gate and cpu_baseline
    for name, w in (("f4 BIP0340 whole-batch verification secp256k1 (one multi-scalar multiplication)", "bip0340_msm"),
                    ("f4 Ed25519 whole-batch verification (one multi-scalar multiplication)", "ed25519_msm")):
        jobs.append((name, [sys.executable, tool, "--workload", w, "--ref-items", "16384", "--traffic", "--mad-peak", repr(pv),
                            "--mad-peak-sgpr", repr(ps), "--steps", "10", "--warmup", "2"]))
    # ... and the same row for EDDSA448 (round 6: the combination on the Weierstrass model WEI448 with the cofactored final test); no PMC passes
    # (the default run stays within minutes)
    jobs.append(("f4 Ed448 whole-batch verification (one multi-scalar multiplication)",
                 [sys.executable, tool, "--workload", "ed448_msm", "--ref-items", "8192", "--mad-peak", repr(pv), "--mad-peak-sgpr", repr(ps),
                  "--steps", "6", "--warmup", "2"]))
    out = []
    for name, cmd in jobs:
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            roof = line.get("roofline") or {}
            out.append({"config": name, "metric": line["metric"], "value": line["value"], "unit": line["unit"], "steps": line["steps"],
                        "ms_per_step": line["ms_per_step"], "parity_gate": line["config"].get("parity_gate"),
                        "kernel": roof.get("kernel"), "kernel_ms": roof.get("kernel_ms"), "kernel_mads_per_item": roof.get("kernel_mads_per_item"),
                        "frac": roof.get("frac"), "peak": roof.get("peak"), "pipeline_frac": roof.get("pipeline_frac"),
                        "mads_per_item": roof.get("mads_per_item"), "step_ms": roof.get("step_ms"),
                        "traffic": roof.get("traffic"), "traffic_by_kernel": roof.get("traffic_by_kernel"),
                        "traffic_over_algorithmic": roof.get("traffic_over_algorithmic"),
                        "cpu_baseline": line.get("cpu_baseline"), "wall_s": time.time() - t0})
        except Exception as e:
            out.append({"config": name, "error": f"{type(e).__name__}: {e}"[:300], "wall_s": time.time() - t0})
    out.append(typed_boundary(out))
    return out


def typed_boundary(device_records):
    """The drop-in in libecc's own types as a driver-timed record: `libecc_amd/lib/compat_check benchj 20`, a
    libecc APPLICATION linked against libsign_amd.so, calls ec_verify_batch / ec_sign_batch on 2^20 libecc structures (ec_pub_key **,
    u8 **; hashing, marshalling, copies, kernels: everything between the call and its return) and, in the same process, libecc's own
    ec_verify / ec_sign / ec_verify_batch on every host thread over a sample of the same structures (the `cpu_baseline` of each call).
    device_ratio = the call's rate over the device-resident rate of the same verification measured above."""
    t0 = time.time()
    exe = os.path.join(ROOT, "libecc_amd", "lib", "compat_check")
    rec = {"config": "typed boundary: libsign_amd.so (include/libecc_amd_compat.h), 2^20 libecc structures in, results out"}
    try:
        r = subprocess.run([exe, "benchj", "20"], capture_output=True, text=True, timeout=420)
        txt = r.stdout[r.stdout.index("{"):]
        j = json.loads(txt)
        dev = {}
        for d in device_records:
            if "value" in d:
                if "ECDSA verify" in d["config"]:
                    dev["ECDSA"] = d["value"]
                elif "Ed25519 verify" in d["config"]:
                    dev["EDDSA25519"] = d["value"]
                elif "BIP0340" in d["config"]:
                    dev["BIP0340"] = d["value"]
        calls = []
        for c in j["records"]:
            e = dict(c)
            if "rate" in c and c["call"].startswith("ec_verify_batch"):
                for k, v in dev.items():
                    if k in c["call"]:
                        e["device_resident_rate"] = v
                        e["device_ratio"] = c["rate"] / v
            for k in ("cpu", "cpu_batch"):
                if k in c:
                    e[k] = {"value": c[k]["rate"], "unit": "items/s", "cores": c[k]["threads"], "kind": "reference",
                            "sample": f"{c[k]['what']}: {c[k]['items']} items on {c[k]['threads']} threads, {c[k]['seconds']:.1f} s"}
            if "cpu" in e:
                e["cpu_baseline"] = e.pop("cpu")
            calls.append(e)
        rec.update({"metric": "items/s through the libecc-typed batch entry points (end to end)", "unit": "items/s",
                    "items": j["items"], "host_threads": j["host_threads"], "calls": calls})
        head = [c for c in calls if c.get("call", "").startswith("ec_verify_batch ECDSA")]
        if head:
            rec["value"] = head[0]["rate"]
            rec["cpu_baseline"] = head[0].get("cpu_baseline")
    except Exception as e:
        rec["error"] = f"{type(e).__name__}: {e}"[:300]
    rec["wall_s"] = time.time() - t0
    return rec


def pmc_traffic(kernel, batch_log2, curve):
    """HBM bytes per launch of the pipeline's kernels, MEASURED IN THIS RUN: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (tools/pmc.py: separate passes, nothing else traced, as
    MI355X_MICROARCH.md's HBM section prescribes; gfx950 correction 2 x FETCH_SIZE, both counters in KiB).  The children run the
    same batch on seeded inputs without the parity gate.  Returns (bytes per launch of `kernel`, {kernel: bytes per launch}, note);
    nulls with a note when rocprofv3 is missing or a pass fails -- never an estimate.  What it counts is window-table scratch
    (the loop's 64 look-ups x 64 B per item and the staging of the table kernels), not re-reads of the 160 algorithmic bytes."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--curve", curve, "--batch-log2", str(batch_log2),
             "--steps", "2", "--warmup", "1"]
    by_kernel, note = pmc.hbm_bytes_per_launch(child, timeout=240)
    if by_kernel is None:
        return None, None, note
    dom = [v for k, v in by_kernel.items() if kernel.split("<")[0] in k and "verify" not in k]
    return (max(dom) if dom else None), by_kernel, note


def cpu_baseline(curve, scalars, points, slen, gate_timing=None):
    """Reference CPU path on this box's host cores, bounded to ~10-20 s of wall time.  gate_timing: the parity gate already ran
    the unmodified reference over a random subset of the same batch on every host thread and timed it -- when that sample is
    at least 3 s of work it IS the baseline (no second run), and only the one-thread probe is added."""
    from oracles import Oracle, RefLib, have_ref
    cores = host_cores()
    nmax = len(scalars) // slen
    if have_ref():
        r = RefLib(curve)
        probe = 64
        t0 = time.time()
        r.scalar_mult(scalars[:probe * slen], points[:probe * 2 * r.clen], slen)
        rate1 = probe / (time.time() - t0)
        if gate_timing and gate_timing["seconds"] >= 3.0:
            g = gate_timing
            return {"value": g["items"] / g["seconds"], "unit": "scalar-mults/s", "cores": g["cores"], "kind": "reference",
                    "one_core_value": rate1,
                    "sample": f"the parity gate's own run: {g['items']} random items of the same batch, prj_pt_mul+prj_pt_unique of the "
                              f"unmodified reference (oracle/_ref, default flags) on {g['cores']} pthreads, {g['seconds']:.1f} s wall; "
                              f"1-thread probe on 64 items {rate1:.0f}/s"}
        # multi-thread probe (4 items per thread) to size the real sample: hosts rarely scale linearly
        n0 = min(nmax, 4 * cores)
        _, _, el0, _ = r.scalar_mult(scalars[:n0 * slen], points[:n0 * 2 * r.clen], slen, nthreads=cores, timing=True)
        n = int(min(nmax, max(n0, (n0 / el0) * 12.0)))
        _, st, el, _ = r.scalar_mult(scalars[:n * slen], points[:n * 2 * r.clen], slen, nthreads=cores, timing=True)
        return {"value": n / el, "unit": "scalar-mults/s", "cores": cores, "kind": "reference",
                "one_core_value": rate1,
                "sample": f"first {n} items of the same batch, prj_pt_mul+prj_pt_unique of the unmodified "
                          f"reference (oracle/_ref, default flags) on {cores} pthreads, {el:.1f} s wall; "
                          f"1-thread probe on 64 items {rate1:.0f}/s"}
    o = Oracle(curve)
    n = min(nmax, 4096)
    t0 = time.time()
    o.scalar_mult(scalars[:n * slen], points[:n * 2 * o.clen], slen)
    el = time.time() - t0
    return {"value": n / el, "unit": "scalar-mults/s", "cores": 1, "kind": "port",
            "sample": f"first {n} items of the same batch through oracle/ecc_oracle.c, 1 thread"}


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:  # honour the container's CPU quota (cgroup v2): "max" or "<quota> <period>"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def edge_slice(curve_params, slen, clen, pts_h, n=4096):
    """The 4096-item edge slice of SURVEY.md 8d cfg-2: m in {0, 1, 2, q-1, q, q+1, 2^(8 slen)-1} on P in {G, -G} and on
    points of the batch; coordinates >= p; points off the curve (all must be rejected); the rest random scalars."""
    p, q = curve_params["p"], curve_params["q"]
    plen = 2 * clen
    top = (1 << (8 * slen)) - 1
    G = curve_params["gx"].to_bytes(clen, "big") + curve_params["gy"].to_bytes(clen, "big")
    mG = curve_params["gx"].to_bytes(clen, "big") + (p - curve_params["gy"]).to_bytes(clen, "big")
    ms = [0, 1, 2, q - 1, q, q + 1, top, (q + 2) & top, q - 2, 3]
    rng = np.random.default_rng(4096)
    sc, pt = bytearray(), bytearray()
    for i in range(n):
        P = pts_h[plen * i:plen * (i + 1)]
        m = int.from_bytes(rng.integers(0, 256, size=slen, dtype=np.uint8).tobytes(), "big")
        kind = i % 8
        if kind < 2:
            m = ms[(i // 8) % len(ms)]
            P = (G, mG)[kind]
        elif kind == 2:
            m = ms[(i // 8) % len(ms)]
        elif kind == 3:                                   # x >= p (p itself, p + small, all ones)
            x = (p, p + 1 + (i % 5), (1 << (8 * clen)) - 1)[(i // 8) % 3]
            if x >> (8 * clen):
                x = (1 << (8 * clen)) - 1
            P = x.to_bytes(clen, "big") + P[clen:]
        elif kind == 4:                                   # y >= p
            P = P[:clen] + ((1 << (8 * clen)) - 1 - (i % 3)).to_bytes(clen, "big")
        elif kind == 5:                                   # off the curve: y + 1, or x and y swapped
            y = int.from_bytes(P[clen:], "big")
            P = (P[:clen] + ((y + 1) % p).to_bytes(clen, "big")) if (i // 8) % 2 else (P[clen:] + P[:clen])
        elif kind == 6:                                   # (0, 0) and (0, y)
            P = bytes(clen) + (bytes(clen) if (i // 8) % 2 else P[clen:])
        sc += m.to_bytes(slen, "big")
        pt += P
    return bytes(sc), bytes(pt)


def parity_gate(curve, cv, scalars_h, pts_h, out_h, slen, plen, B, nrand):
    from oracles import CURVES, Oracle, RefLib, have_ref
    clen = plen // 2
    use_ref = have_ref()
    if not use_ref:
        nrand = min(nrand, 2048)
    nrand = min(nrand, B)
    idx = np.sort(np.random.default_rng(1).choice(B, size=nrand, replace=False))
    sub_s = b"".join(scalars_h[slen * i:slen * i + slen] for i in idx)
    sub_p = b"".join(pts_h[plen * i:plen * i + plen] for i in idx)
    got = b"".join(out_h[plen * i:plen * i + plen] for i in idx)
    e_sc, e_pt = edge_slice(CURVES[curve], slen, clen, pts_h, 4096 if B >= 4096 else B)
    e_got = cv.scalar_mult(e_sc, e_pt, slen)
    t0 = time.time()
    ref_timing = None
    if use_ref:
        r = RefLib(curve)
        cores = host_cores()
        exp, est, el_ref, _ = r.scalar_mult(sub_s, sub_p, slen, nthreads=cores, timing=True)
        ref_timing = {"items": nrand, "seconds": el_ref, "cores": cores}   # the gate's own CPU run doubles as the cpu_baseline sample
        e_exp = tuple(r.scalar_mult(e_sc, e_pt, slen, nthreads=cores)[:2])
        who = f"the unmodified reference (oracle/_ref) on {cores} threads"
    else:
        o = Oracle(curve)
        exp, est = o.scalar_mult(sub_s, sub_p, slen)
        e_exp = tuple(o.scalar_mult(e_sc, e_pt, slen))
        who = "the C restatement oracle (oracle/_ref absent)"
    if got != exp or set(est) != {0}:
        raise SystemExit("PARITY FAILURE: GPU output differs from the CPU reference on the random subset")
    if tuple(e_got) != e_exp:
        bad = [i for i in range(len(e_exp[1])) if e_got[1][i] != e_exp[1][i] or
               e_got[0][plen * i:plen * (i + 1)] != e_exp[0][plen * i:plen * (i + 1)]]
        raise SystemExit(f"PARITY FAILURE: edge slice differs from the CPU reference at items {bad[:8]}")
    st = e_exp[1]
    return (f"{nrand} random items of the timed batch + {len(st)} edge items ({st.count(1)} rejected, {st.count(2)} at infinity) "
            f"byte-identical to {who}, {time.time() - t0:.1f} s"), ref_timing


def traffic_child(args):
    """the workload of the timed region alone (no gate, no timing), for the rocprofv3 --pmc passes of pmc_traffic()"""
    from oracles import CURVES
    cp = CURVES[args.curve]
    B = 1 << args.batch_log2
    slen, clen = (cp["q"].bit_length() + 7) // 8, (cp["p"].bit_length() + 7) // 8
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = libecc_amd.Context(0)
    cv = ctx.curve(args.curve)
    rng = np.random.default_rng(SEED)
    raw = rng.integers(0, 256, size=(2, B * slen), dtype=np.uint8)
    raw[:, ::slen] &= 0x7f                       # any scalar value is fine for the traffic; keep them below 2^(8 slen - 1)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    d_s, d_t = torch.from_numpy(raw[0]).to(dev), torch.from_numpy(raw[1]).to(dev)
    d_p = torch.empty(B * 2 * clen, dtype=torch.uint8, device=dev)
    d_o = torch.empty(B * 2 * clen, dtype=torch.uint8, device=dev)
    d_st = torch.empty(B, dtype=torch.uint8, device=dev)
    cv.scalar_mult_dev(B, d_t.data_ptr(), slen, None, d_p.data_ptr(), d_st.data_ptr(), stream.cuda_stream)
    for _ in range(args.warmup + args.steps):
        cv.scalar_mult_dev(B, d_s.data_ptr(), slen, d_p.data_ptr(), d_o.data_ptr(), d_st.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    cv.free()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)   # the first launches after the setup run at ramping clocks (profiles/r1f_bench_kernels.md)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ubench-json", default=None, help="also write the raw micro-benchmark output (instruction rates, sustained clock) here")
    ap.add_argument("--parity-items", type=int, default=1 << 16, help="random items of the batch checked against the CPU reference before timing")
    ap.add_argument("--curve", default=CURVE, help="ad-hoc runs on another built-in curve (the driver uses the default)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes that measure the HBM bytes per launch")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-secondary", action="store_true", help="skip the records of BASELINE configs[2]-[4] that follow the headline measurement at N = 1")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): every rank owns 2^batch-log2 items; strong: 2^batch-log2 items in total, "
                         "split into contiguous shards of 2^batch-log2 / N per rank (BASELINE.json's metric read literally: batch = 2^20 at 1/2/4/8 GPUs)")
    args = ap.parse_args()

    if args.traffic_child:
        return traffic_child(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) and hand back
        # their exit status -- never fall through to a silent 1-GPU run
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible")
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or let "
                         "`python bench.py --gpus N` spawn them)")
    # DRY RUN of the N > 1 code path on a box with fewer GPUs than ranks (development only, never a measurement): with
    # ECAMD_BENCH_DRY_RUN_SHARED_GPU=1 every rank uses cuda:0 and the process group is gloo (RCCL refuses two ranks on one device); the
    # JSON line then says so in "dry_run" and carries no claim about scaling
    dry_shared = world > 1 and os.environ.get("ECAMD_BENCH_DRY_RUN_SHARED_GPU") == "1"
    if dry_shared:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but {torch.cuda.device_count()} are visible")
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry_shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from oracles import CURVES, Oracle, build_oracle
    # the CPU checker is compiled on first use: let rank 0 do it alone, the others wait
    if rank == 0:
        build_oracle()
    if dist is not None:
        dist.barrier()
    curve = args.curve
    cp = CURVES[curve]
    q = cp["q"]
    B = 1 << args.batch_log2
    if args.scaling == "strong":
        if B % world:
            raise SystemExit(f"bench.py: --scaling strong needs 2^{args.batch_log2} items to divide by {world} ranks")
        B //= world                      # contiguous shard of this rank; the job's batch stays 2^batch_log2
    slen, clen = (q.bit_length() + 7) // 8, (cp["p"].bit_length() + 7) // 8
    plen = 2 * clen

    ctx = libecc_amd.Context(local_rank)
    cv = ctx.curve(curve)

    # ---- synthetic inputs (seeded; rank-dependent shard) ----
    rng = np.random.default_rng(SEED + rank)

    # uniform in [1, q-1]: rejection-free by reducing 320 random bits (bias 2^-64)
    t_setup = time.time()
    raw = rng.integers(0, 256, size=(2, B, slen + 8), dtype=np.uint8)
    qm1 = q - 1

    def reduce_rows(rows):
        out = bytearray(B * slen)
        for i in range(B):
            v = (int.from_bytes(rows[i].tobytes(), "big") % qm1) + 1
            out[slen * i:slen * i + slen] = v.to_bytes(slen, "big")
        return bytes(out)

    scalars_h = reduce_rows(raw[0])
    t_h = reduce_rows(raw[1])
    # a real (non-null) stream: the C ABI maps a NULL stream to the context's own stream, and the
    # HIP events below must be recorded on the stream the kernel is launched on
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    d_scalars = torch.frombuffer(bytearray(scalars_h), dtype=torch.uint8).to(dev)
    d_t = torch.frombuffer(bytearray(t_h), dtype=torch.uint8).to(dev)
    d_points = torch.empty(B * plen, dtype=torch.uint8, device=dev)
    d_status = torch.empty(B, dtype=torch.uint8, device=dev)
    # base points P_i = [t_i]G, produced on the GPU by the same engine (fixed base), untimed
    cv.scalar_mult_dev(B, d_t.data_ptr(), slen, None, d_points.data_ptr(), d_status.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    assert int(d_status.max().item()) == 0, "base-point generation produced an error status"
    # N > 1: every step ends with ONE all-gather of the output shards, issued asynchronously so that it
    # overlaps the next step's kernels; outputs are double-buffered (libecc_amd/shard.py:OverlappedGather,
    # covered by a world-size-2 gloo test).  All gathers are waited for inside the timed region.
    from libecc_amd.shard import OverlappedGather
    og = OverlappedGather(world, B * plen, dev)
    d_outs = og.bufs

    def step():
        buf = og.next_buffer()
        cv.scalar_mult_dev(B, d_scalars.data_ptr(), slen, d_points.data_ptr(), buf.data_ptr(),
                           d_status.data_ptr(), stream.cuda_stream)
        og.submit(buf)

    drain = og.drain

    # ---- parity gate: >= 2^16 random items of the timed batch AND a 4096-item edge slice, byte for
    #      byte against the unmodified reference on all host threads (oracle/_ref); without it, the C restatement on
    #      a smaller sample ----
    step()
    drain()
    torch.cuda.synchronize()
    out_h = d_outs[0].cpu().numpy().tobytes()
    pts_h = d_points.cpu().numpy().tobytes()
    st_h = d_status.cpu().numpy().tobytes()
    assert set(st_h) == {0}, "unexpected status in the synthetic batch"
    gate, gate_timing = parity_gate(curve, cv, scalars_h, pts_h, out_h, slen, plen, B, args.parity_items if rank == 0 else 256)
    setup_s = time.time() - t_setup

    # ---- warmup, then exactly K timed steps ----
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ctx.enable_kernel_timing(True)   # HIP events inside the library, on the stream the kernels run on
    ktimes = []
    t0 = time.perf_counter()
    ev[0].record(stream)
    for k in range(args.steps):
        step()
        ev[k + 1].record(stream)
        try:
            ktimes.append(ctx.kernel_times())   # [table, affine, loop, finalize] ms of this launch
        except libecc_amd.EcamdError:
            pass
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [ev[k].elapsed_time(ev[k + 1]) for k in range(args.steps)]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # which device every rank ran on (the RCCL world, as the job saw it)
    props = torch.cuda.get_device_properties(dev)
    mine = f"rank {rank}: cuda:{local_rank} {props.name} ({getattr(props, 'gcnArchName', '?')}, {props.multi_processor_count} CUs)"
    if world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
    else:
        rank_devices = [mine]
    if rank == 0:
        total_items = B * world * args.steps
        value = total_items / elapsed
        mm, mads, kname, kmads = work_model(cp, cv.words, slen, B)
        step_ms = float(np.mean(kern_ms))             # HIP-event time of one whole step on the launch stream
        if ktimes:
            kt = np.mean(np.array(ktimes), axis=0)     # per-kernel averages over the timed steps
            launch_ms = float(kt[2])                   # the dominant kernel: the window loop
            knames = ["table", "affine", "loop", "finalize"]
        else:
            kt, launch_ms, kmads = None, step_ms, mads
        mad_rate = B * kmads / (launch_ms * 1e-3)      # executed lane-MADs per second of the dominant kernel
        # the MAD issue peak is a per-GPU number: measured on rank 0's GPU after the timed region
        peak_v, ub = measured_mad_peak()
        sg = getattr(work_model, "loop_sgpr_share", 0.0) if kt is not None else 0.0
        peak = mix_peak(ub, sg, signed=(cp["p"] == 2**384 - 2**128 - 2**96 + 2**32 - 1)) if ub else None
        nominal_quarter = 256 * 4 * 16 * 2.4e9 / 4.0   # SURVEY.md 8d planning figure (quarter rate)
        alg_bytes = float(slen + 2 * plen)             # scalar + affine point in + affine point out (160 B for P-256)
        hbm_rate = B * alg_bytes / (step_ms * 1e-3)
        line = {
            "metric": f"scalar-mults/sec ({curve.lower()}, batch=2^{args.batch_log2}, variable base, affine out, bit-exact vs CPU)",
            "value": value, "unit": "scalar-mults/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "u32 (%d-bit limbs, v_mad_u64_u32 integer MAD, u64 accumulators)" % (28 if cp["p"] == 2**448 - 2**224 - 1 else 29),
            "data": "synthetic (seeded): scalars uniform in [1,q-1], base points P_i=[t_i]G",
            "config": {"workload": f"{curve} prj_pt_mul+prj_pt_unique, batch 2^{args.batch_log2} " +
                                   ("per GPU" if args.scaling == "weak" else f"in total ({B} per GPU)") + " (BASELINE.json configs[1])",
                       "batch_per_gpu": B, "scalar_len": slen, "window": "signed fixed w=4",
                       "sharding": "contiguous per-rank shards" + (", one RCCL all_gather of the output shards per step, overlapped with the next step's kernels" if world > 1 else ""),
                       "parity_gate": gate,
                       "world_size": world, "rank_devices": rank_devices},
            "roofline": {
                "bound": "valu-int-mad (v_mad_u64_u32 issue; not hbm, not mfma -- SURVEY.md 8d)",
                "achieved": mad_rate / 1e9, "peak": (peak or nominal_quarter) / 1e9, "unit": "GMAD/s (one GPU)",
                "frac": mad_rate / (peak or nominal_quarter),
                "peak_source": ("measured live by libecc_amd/lib/ubench: the v_mad_u64_u32 streams with a VGPR and with an SGPR multiplier, "
                                "weighted by the kernel's operand mix") if peak else "nominal quarter-rate estimate",
                "sgpr_multiplier_share": sg, "peak_vgpr_stream": (peak_v or 0) / 1e9,
                "frac_of_vgpr_stream": mad_rate / peak_v if peak_v else None,
                "kernel": kname, "kernel_ms": launch_ms, "kernel_mads_per_item": kmads,
                "pipeline_ms": ({k_: float(v) for k_, v in zip(knames, kt)} if kt is not None else None),
                "step_ms": step_ms, "pipeline_frac": (B * mads / (step_ms * 1e-3)) / (peak or nominal_quarter),
                "field_mults_per_item": mm, "mads_per_item": mads,
                "ref_equivalent_mads_per_item": ref_equiv_mads(),
                "traffic": None,
                "hbm": {"achieved": hbm_rate / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": hbm_rate / 8e12, "algorithmic_bytes_per_item": alg_bytes},
            },
            "setup_s": setup_s,
        }
        if dry_shared:
            line["dry_run"] = "ECAMD_BENCH_DRY_RUN_SHARED_GPU=1: all ranks on cuda:0 over gloo -- exercises the N > 1 code path only, NOT a measurement"
        if ub:
            line["ubench"] = {k: (v["cycles_per_wave_instr_per_simd"] if isinstance(v, dict) else v)
                              for k, v in ub.items() if k.startswith("v_") or k.startswith("mix_")}
            # the analytic ceiling beside the measured one: 4 cycles per wave64 v_mad_u64_u32 on every SIMD at the part's MAXIMUM clock (no
            # stream reaches it: the instruction measures 5 cycles and the part sustains 2.1 - 2.3 GHz under it, profiles/r3a_effective_clock.md)
            ana = ub.get("analytic_mad_peak_at_max_clock")
            if ana:
                line["roofline"]["peak_analytic_4_cycles_at_max_clock"] = ana / 1e9
                line["roofline"]["frac_of_analytic_peak"] = mad_rate / ana
            if args.ubench_json:
                try:
                    json.dump(ub, open(args.ubench_json, "w"), indent=1)
                except OSError:
                    pass
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(curve, scalars_h, pts_h, slen, gate_timing)
        else:
            line["cpu_baseline"] = None
    cv.free()
    ctx.close()
    if rank == 0:
        # after the headline region, with this process's device memory released:
        # (i) the HBM bytes per launch from PMC counters, measured on this very batch by two profiled child runs
        if world == 1 and not args.no_traffic:
            t0 = time.time()
            tr, by_kernel, note = pmc_traffic(kname, args.batch_log2, curve)
            line["roofline"]["traffic"] = tr
            line["roofline"]["traffic_by_kernel"] = by_kernel
            line["roofline"]["traffic_note"] = f"{note}; {time.time() - t0:.0f} s"
            if tr:
                line["roofline"]["traffic_over_algorithmic"] = tr / (B * alg_bytes)
        # (ii) the other BASELINE configs, one fresh process each
        if world == 1 and not args.no_secondary and curve == CURVE:
            line["secondary"] = secondary_lines(ub)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
