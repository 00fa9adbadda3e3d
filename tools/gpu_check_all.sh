#!/bin/bash
# One budget-conscious GPU pass that re-measures everything README.md / DESIGN.md quote (about 14 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_check_all.sh'
# Outputs under gpurun_out/check/ (copy what is to be kept into profiles/).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/check
mkdir -p $O
cd $R
# 1. the whole GPU suite as the driver runs it (~7 min), smoke, the bench line
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 300 python bench.py --ubench-json $O/ubench.json > $O/bench.json 2> $O/bench.err
# 2. the other field sizes (batch 2^20)
B="python $R/bench.py --no-cpu-baseline --no-traffic --no-secondary --parity-items 4096 --steps 5 --warmup 2"
for c in SECP384R1 SECP521R1 WEI448 BRAINPOOLP256R1 BRAINPOOLP512R1 SECP256K1 WEI25519 SECP224R1 SECP192R1; do
  timeout 200 $B --curve $c > $O/bench_$c.json 2> $O/bench_$c.err
done
# 3. protocol workloads, the whole-batch EdDSA predicate, secret mode / blinding, the libecc-typed boundary end to end
for w in ecdsa_verify ed25519_verify x25519 ed448_verify x448 ecdsa_sign ecccdh; do
  timeout 200 python tools/bench_protocols.py --workload $w --no-cpu-baseline > $O/proto_$w.json 2> $O/proto_$w.err
done
MSM_LOG2=16,17,18,20 MSM_K=0 timeout 200 python tools/bench_msm.py > $O/bench_msm.json 2> $O/bench_msm.err
timeout 200 python tools/bench_secret_mode.py > $O/secret_mode.json 2> $O/secret_mode.err
timeout 100 libecc_amd/lib/compat_check bench 18 > $O/compat_bench_18.txt 2>&1
# 4. rocprofv3: kernel trace of the bench command, then the two PMC passes (separate runs, nothing else traced)
cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 10 --warmup 3 > $O/prof_bench.json 2> $O/prof_bench.err
db=$(ls -S $(find $O/prof -name '*.db') | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py kernels $db > $O/bench_kernels.md && python $R/tools/rocpd_summary.py dispatches $db k_p256_loop > $O/bench_loop_dispatches.md
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $c -d $O/pmc_$c -- python $R/bench.py --no-cpu-baseline --parity-items 1024 --steps 2 --warmup 1 > $O/pmc_$c.json 2> $O/pmc_$c.err
  db=$(ls -S $(find $O/pmc_$c -name '*.db') | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py pmc $db > $O/pmc_$c.md
done
find $O -name '*.db' -delete; find $O -size +1M -delete
tail -n 4 $O/pytest.log; tail -n 1 $O/smoke.log; head -c 300 $O/bench.json
