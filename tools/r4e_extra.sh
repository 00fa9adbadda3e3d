cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f
mkdir -p $O
for c in SECP384R1 SECP521R1 SECP256R1; do
  for kp in 4 8 16 32; do
    ECAMD_PREP_KP=$kp timeout 200 python tools/bench_protocols.py --workload ecdsa_verify --curve $c --no-cpu-baseline --ref-items 0 --steps 6 --warmup 2 > $O/ecdsa_${c}_kp$kp.json 2> /dev/null
    python - $O/ecdsa_${c}_kp$kp.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],3), 'ms')
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
  done
done
