# usage: bash tools/install_profiles.sh <gpurun_out tag> <profiles prefix>     copies the summaries of tools/refresh_profiles.sh into profiles/
set -e
S=gpurun_out/$1; P=profiles/$2
for f in bench_n1 bench_k20 bench_c1 bench_c2 bench_tiled_n1 bench_tiled_loopback8_delta bench_tiled_loopback8_full bench_tiled_loopback2_delta \
         bench_cloud bench_query pmc_hbm pmc_hbm_k20 pmc_mfma_stream pmc_mfma_k20 bench_batch5 bench_rccl_1rank bench_rccl_1rank_before_clock bench_n1_one_queue \
         bench_k20_one_queue; do
  [ -s $S/$f.json ] && cp $S/$f.json ${P}_$f.json
done
for n in 2 4 8; do for f in bench_s$n pmc_mfma_s$n pmc_mfma_s${n}_k20 pmc_hbm_s$n; do [ -s $S/$f.json ] && cp $S/$f.json ${P}_$f.json; done; done
for f in kernel_stats_s4.md kernel_stats_s4_steady.md timeline_s4.txt kernel_stats_query.md kernel_stats_cloud.md kernel_stats_frows.md; do [ -s $S/$f ] && cp $S/$f ${P}_$f; done
[ -s $S/bench_frows.json ] && cp $S/bench_frows.json ${P}_bench_frows.json
[ -s $S/stress_full.json ] && cp $S/stress_full.json ${P}_stress_full_occupancy.json
[ -s $S/stress_integrate.json ] && cp $S/stress_integrate.json ${P}_stress_integrate.json
[ -s $S/sweep_decode.jsonl ] && cp $S/sweep_decode.jsonl ${P}_sweep_decode.jsonl
for f in kernel_stats.md kernel_stats_steady.md kernel_stats_steady_one_queue.md kernel_stats_k20.md kernel_stats_tiled_loopback8.md kernel_stats_tiled_n1.md timeline_direct.txt timeline_overlap.txt; do [ -s $S/$f ] && cp $S/$f ${P}_$f; done
# every profiles/*.json must parse: a bench run under torch.distributed prints the gloo / RCCL banners before its line — keep the line only
python - <<'PY'
import json, pathlib
for p in sorted(pathlib.Path("profiles").glob("*.json")):
    t = p.read_text()
    try:
        json.loads(t)
        continue
    except Exception:
        pass
    for line in reversed(t.strip().splitlines()):
        try:
            json.loads(line)
        except Exception:
            continue
        p.write_text(line + "\n")
        print("stripped to its JSON line:", p)
        break
    else:
        print("NOT JSON:", p)
PY
ls -la profiles | grep "$2" | wc -l
