#!/bin/bash
# dev (GPU box): second pass of round 3 -- full GPU suite, instanced A/B, headline, WRITE_SIZE of the batched RGBA stores
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3c; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/suite.log; tail -4 $O/suite.log
run() { "$@" timeout 300 python bench.py --workload carpet_instanced --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4))"; }
run env A=1; run env NERFTEX_NO_DIR_HOIST=1; run env A=2
for w in carpet fur_sharded; do
  timeout 400 python bench.py --steps 5 --warmup 1 --workload $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['roofline']['kernel_ms'],2), round(d['roofline']['frac'],4), d['parity']['rel_linf_f32'])"
done
ONLY="kt pmc2 pmc5" bash tools/profile_bench.sh pw_fur_sharded --workload fur_sharded > /dev/null 2>&1
ONLY="kt pmc2 pmc5" bash tools/profile_bench.sh pw_carpet --workload carpet > /dev/null 2>&1
ONLY="pmc2 pmc5" bash tools/profile_bench.sh pw_gfs --workload grass_filtered_sharded > /dev/null 2>&1
find gpurun_out -name "*.db" -delete 2>/dev/null
