#!/bin/bash
# GPU box helper: rocprofv3 kernel stats + PMC passes of one bench workload -> gpurun_out/prof_<tag>_<env>_<envs>
# (counters are collected in their own passes, never together with --kernel-trace/--stats tracing domains)
#   tools/profile.sh <tag> [<env id> [<envs>]]      default: the metric's workload, AntUMaze-v0 / 4096
tag=${1:-r03}
envid=${2:-AntUMaze-v0}
nenv=${3:-4096}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_${envid}_${nenv}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the default bench command (100 settle + 100 warm-up + 1000 timed steps), minus the CPU leg
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-pmc --sustained 0 --env $envid --envs $nenv ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $CMD > $out/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $CMD > $out/bench_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $out/pmc_sq -- $CMD > $out/bench_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_lds -- $CMD > $out/bench_lds.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $out/pmc_busy -- $CMD > $out/bench_busy.log 2>&1
find $out -name "*.csv" | head -40
