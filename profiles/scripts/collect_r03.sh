# round-3 evidence: bash profiles/scripts/collect_r03.sh   (on the GPU box; results under gpurun_out/r3p, summaries copied by
# `python profiles/make_summary.py r03 gpurun_out/r3p/trace gpurun_out/r3p/pmc` afterwards)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/bench.py --steps 200 --warmup 20 > $O/bench_traced.json 2> $O/bench_traced.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_fetch -- python $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc1.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc -o pmc_write -- python $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc2.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc -o pmc_mfma -- python $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-solve > /dev/null 2> $O/pmc3.err
for CFG in cfg3 cfg4 cfg2; do
  PYTHONPATH=$R:$R/tests rocprofv3 --kernel-trace --stats --output-format csv -d $O/solve_$CFG -o solve -- python $R/tests/prof_cfg.py $CFG > $O/solve_$CFG.log 2>&1
done
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
MCBA_FUSED=0 python bench.py --no-cpu-baseline > $O/bench_table_form.json 2>/dev/null
MCBA_GRAPH=1 python bench.py --no-cpu-baseline > $O/bench_graph.json 2>/dev/null
python tests/prof_linearize.py > $O/lin_phases.log 2>&1; tail -12 $O/lin_phases.log
MCBA_TIMING=1 python tests/prof_workspace.py cfg3 > $O/workspace_cfg3.log 2>&1; grep "calibrate ms" $O/workspace_cfg3.log
MCBA_TIMING=1 python tests/prof_workspace.py cfg4 2>&1 | grep 'calibrate ms'
MCBA_TIMING=1 python tests/prof_workspace.py cfg2 2>&1 | grep 'calibrate ms'
python tests/prof_lin_cfgs.py cfg2 cfg3 cfg4 cfg5 > $O/lin_cfgs.log 2>&1; cat $O/lin_cfgs.log
python tests/prof_cholp.py > $O/cholp_phases.log 2>&1; cat $O/cholp_phases.log
python tests/prof_chol.py > $O/chol_paths.log 2>&1; cat $O/chol_paths.log
python tests/prof_xcd.py > $O/xcd_probe.log 2>&1; cat $O/xcd_probe.log
python tests/prof_init.py cfg2 cfg3 cfg4 > $O/init.log 2>&1; cat $O/init.log
ls $O $O/pmc
