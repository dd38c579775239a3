cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in "0:0-255" "0:0-63" "0:0-31" "0:0-15" "0:0-7"; do
  echo "== HSA_CU_MASK=$m"
  HSA_CU_MASK=$m timeout 200 python bench.py --no-cpu-baseline --no-secondary --d2h none --steps 30 --warmup 5 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('other_ms_per_frame'))
except Exception as e: print('no json', e)"
done
