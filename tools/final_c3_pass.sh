#!/bin/bash
# Round-end refresh of the config-3 evidence on the GPU box (from the repo root): kernel stats and HBM traffic of K5.
export TMPDIR=/tmp
R=r03; OUT=gpurun_out/$R; mkdir -p $OUT
B="--steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extras"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/kt_c3 -o kt -f csv -- python3 bench.py $B --workload criteo_c3_attn > $OUT/kt_c3.json 2> $OUT/kt_c3.err
MS=$(tail -1 $OUT/kt_c3.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')
python3 tools/profile_summary.py $OUT/kt_c3 auto $OUT/${R}_kernel_stats_c3.md "bench.py $B --workload criteo_c3_attn under rocprofv3 --kernel-trace --stats; bench line of this run: $MS"
export XDFM_HIP_GRAPH=0
PB3="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-extras --vocab-preset mid --workload criteo_c3_attn"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f3 -o pmc -f csv -- python3 bench.py $PB3 > $OUT/pmc_f3.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w3 -o pmc -f csv -- python3 bench.py $PB3 > $OUT/pmc_w3.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_f3 $OUT/pmc_w3 - > $OUT/${R}_pmc_c3_traffic.txt
grep "attn" $OUT/${R}_pmc_c3_traffic.txt
head -12 $OUT/${R}_kernel_stats_c3.md | tail -5 | cut -c1-150
