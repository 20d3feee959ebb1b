# HBM counters of the LM solve kernels: bash profiles/scripts/solve_pmc.sh <cfg>   -> gpurun_out/spmc/<cfg>_*.csv
R=$GRAFT_REPO_ROOT; CFG=$1; O=$R/gpurun_out/spmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  PYTHONPATH=$R:$R/tests rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${CFG}_$C -o pmc -- python $R/profiles/scripts/prof_cfg.py $CFG > $O/${CFG}_$C.log 2>&1
done
