#!/bin/bash
out=gpurun_out/${1:-r5c}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-secondary"
run() { name=$1; shift; timeout 300 "$@" > $out/$name.json 2> $out/$name.err; python - $out/$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d["config"]
    print(f"{sys.argv[2]:28s} {d['value']:9.1f} frames/s {d['ms_per_step']} ms  engine={c.get('d2h_engine')} two_queues={c.get('two_queues',{}).get('on')}")
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run a_default $B
run b_nosample $B --sample-every 100000
DIF_BENCH_NO_PRIME=1 run c_noprime $B
DIF_BENCH_NO_PRIME=1 run d_noprime_nosample $B --sample-every 100000
run e_ov0 $B --overlap 0
run f_ov0_nosample $B --overlap 0 --sample-every 100000
run g_none $B --d2h none
run h_none_nosample $B --d2h none --sample-every 100000
