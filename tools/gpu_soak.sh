#!/bin/bash
# a long seeded soak of the differential fuzzers (seeds the suite does not use) and the repeat-determinism check; the tails go to gpurun_out/soak/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/soak; mkdir -p $out
# the tests the driver's `-m gpu` step leaves out (other fuzz seeds, the one-queue run of the 25-frame C3 window, the C2 window)
timeout 1500 python -m pytest tests -m soak -x -q --durations=15 > $out/pytest_soak.log 2>&1; echo "pytest -m soak rc=$? $(tail -1 $out/pytest_soak.log)"
timeout 900 python tools/soak_overlap.py ${SOAK_OVERLAP_RUNS:-12} 40 > $out/overlap.log 2>&1; echo "two-queue soak rc=$? $(tail -1 $out/overlap.log)"
for seed in ${SOAK_SEEDS:-101 102 103}; do
  timeout 1200 python tools/fuzz_integrate.py --cases 120 --seed $seed > $out/integrate_$seed.log 2>&1; echo "integrate $seed rc=$? $(tail -1 $out/integrate_$seed.log)"
  timeout 900 python tools/fuzz_mc.py --cases 300 --seed $seed > $out/mc_$seed.log 2>&1; echo "mc $seed rc=$? $(tail -1 $out/mc_$seed.log)"
  timeout 900 python tools/fuzz_cloud.py --cases 150 --seed $seed > $out/cloud_$seed.log 2>&1; echo "cloud $seed rc=$? $(tail -1 $out/cloud_$seed.log)"
  timeout 900 python tools/fuzz_group.py --cases 25 --seed $seed > $out/group_$seed.log 2>&1; echo "group $seed rc=$? $(tail -1 $out/group_$seed.log)"
done
timeout 900 python tools/determinism_stress.py 60 > $out/determinism.log 2>&1; echo "determinism rc=$? $(tail -1 $out/determinism.log)"
timeout 900 python tools/determinism_stress.py 30 --mc-grid-cap 3 > $out/determinism_ticket.log 2>&1; echo "determinism (ticket mode) rc=$? $(tail -1 $out/determinism_ticket.log)"
timeout 900 python tools/stress_mc_ticket.py --n 128 --reps 2 > $out/mc_ticket_128.log 2>&1; echo "mc ticket 128 rc=$? $(tail -1 $out/mc_ticket_128.log)"
