#!/bin/bash
# dev (GPU box): A/B of two builds of the library on the same box, interleaved.  usage: dev_ab.sh <other.so name in build_dev/r2> [bench args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab; mkdir -p $O; OTHER=$1; shift   # one name, or several separated by commas
for i in 1 2 3 4; do for lib in default ${OTHER//,/ }; do
  if [ $lib = default ]; then unset NERFTEX_LIB; else export NERFTEX_LIB=$R/build_dev/r2/$lib.so; fi
  timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AB $lib', d['roofline']['kernel_ms'])"
done; done | tee $O/ab.log
python - <<'PY'
import collections,os
d=collections.defaultdict(list)
for l in open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/ab/ab.log"):
    _,k,v=l.split(); d[k].append(float(v))
for k,v in d.items(): print(k, "mean %.2f min %.2f max %.2f"%(sum(v)/len(v), min(v), max(v)))
PY
