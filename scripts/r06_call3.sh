#!/bin/bash
# Round 6, GPU call 3: what the update chain's GEMMs would gain from pre-split operands -- a TIMING ABLATION build whose split3 makes no residual
# pieces (wrong results, same loads / LDS / MFMAs) next to the round-5 library and this round's variants, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call3; rm -rf $O; mkdir -p $O; cd $R
NB="--no-cpu-baseline --no-verify --steps 110 --repeats 3"
run() { tag=$1; shift; timeout 200 python bench.py $NB "$@" > $O/$tag.json 2> $O/$tag.err; python -c "
import json
try:
    d=json.load(open('$O/$tag.json')); print('$tag', d.get('ms_per_step', d.get('diagnostic_ms_per_step')), d['ms_per_step_runs'])
except Exception as e: print('$tag FAILED', open('$O/$tag.err').read()[-600:])"; }
L=$R/serl_amd/lib
for v in "head SERL_MI355_LIB=$L/libserl_mi355_head.so" "ablate SERL_MI355_LIB=$L/libserl_mi355_ablate_split.so" "neither SERL_DENSE_ROWS=0 SERL_GEMM_BIG=0" "bigenc SERL_DENSE_ROWS=0 SERL_ENC_SPLIT_BUDGET=4096"; do
  set -- $v; m=$1; shift
  for e in "$@"; do export $e; done
  run upd_$m --farm-role updater
  run serial_$m --no-pipeline
  run emu8_$m --emulate-world 8
  unset SERL_DENSE_ROWS SERL_GEMM_BIG SERL_MI355_LIB SERL_ENC_SPLIT_BUDGET
done
SERL_MI355_LIB=$L/libserl_mi355_head.so run small_head --encoder small --steps 40
SERL_MI355_LIB=$L/libserl_mi355_ablate_split.so run small_ablate --encoder small --steps 40
SERL_DENSE_ROWS=0 SERL_GEMM_BIG=0 run small_neither --encoder small --steps 40
cd /tmp && export TMPDIR=/tmp
for m in head ablate_split; do
SERL_MI355_LIB=$L/libserl_mi355_$m.so timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -o t -- python $R/bench.py --no-cpu-baseline --no-verify --fill 3000 --steps 12 --warmup 5 --repeats 1 --farm-role updater > $O/trace_$m.log 2>&1
(cd $R && python scripts/chain_trace.py $O/trace_$m > $O/launches_$m.txt 2>&1)
done
find $O -name '*.db' -delete; find $O -name '*.csv' -size +2M -delete
paste -d'|' <(cut -c1-60,95-108 $O/launches_head.txt) <(cut -c95-108 $O/launches_ablate_split.txt) | tail -56
