# per-kernel average durations of the evaluation step: bash profiles/scripts/kstats.sh <tag> [env assignments ...]
R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/ks_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-solve > $O/bench.json 2> $O/bench.err
python - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
  n = r["Name"]
  if "mcba" in n and int(r["Calls"]) > 100: print("$T", n.split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2))
PY
grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1
cd $R
