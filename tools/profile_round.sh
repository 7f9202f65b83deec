#!/bin/bash
# Profiles of one round, on the GPU box, from the repo root:  bash tools/profile_round.sh r02
# kernel-trace statistics of bench.py for the headline workload (criteo-card), the mid vocabulary, config 3 and config 5,
# then counter passes (separate runs, --kernel-trace only next to --pmc).  The bench.py counter passes run with
# XDFM_HIP_GRAPH=0 (eager launches): round 1's counter pass over the graph-replayed step hung.
set -u
R=${1:-r03}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
B="--steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extras"
run_kt() {  # name, bench args
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -f csv -- python3 bench.py $B "$@" > $OUT/kt_$name.json 2> $OUT/kt_$name.err || { echo "kt $name failed"; return 1; }
  python3 tools/profile_summary.py $OUT/kt_$name auto $OUT/${R}_kernel_stats_$name.md "bench.py $B $* under rocprofv3 --kernel-trace --stats (steps in the trace: 4 pre-capture + 5 warm-up + 30 timed + 2 + 10 event-bracketed eager ones); bench line of this run: $(tail -1 $OUT/kt_$name.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"
}
run_kt card && run_kt mid --vocab-preset mid && run_kt c3 --workload criteo_c3_attn && run_kt c5 --workload avazu_c5 && run_kt pro --workload criteo_pro || exit 1
echo "== MFMA pipe counters of the CIN kernels"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_m -o pmc -f csv -- python3 tools/pmc_cin.py > $OUT/pmc_m.log 2>&1 && python3 tools/pmc_summary.py $OUT/pmc_m cin_ > $OUT/${R}_pmc_cin_mfma.txt || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_s -o pmc -f csv -- python3 tools/pmc_cin.py > $OUT/pmc_s.log 2>&1 && python3 tools/pmc_summary.py $OUT/pmc_s cin_ > $OUT/${R}_pmc_cin_waves.txt || exit 1
echo "== HBM traffic counters over bench.py itself (eager launches)"
export XDFM_HIP_GRAPH=0
PB="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-extras --vocab-preset mid"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o pmc -f csv -- python3 bench.py $PB > $OUT/pmc_f.log 2>&1 || { echo "FETCH_SIZE pass over bench.py failed / timed out"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o pmc -f csv -- python3 bench.py $PB > $OUT/pmc_w.log 2>&1 || { echo "WRITE_SIZE pass over bench.py failed / timed out"; exit 1; }
python3 tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w $OUT/${R}_pmc_bench_traffic.json > $OUT/${R}_pmc_bench_traffic.txt
echo "== the attention kernels (config 3), the gather / scatter at larger batches, the SFG heads' kernels"
PB3="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-extras --vocab-preset mid --workload criteo_c3_attn"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f3 -o pmc -f csv -- python3 bench.py $PB3 > $OUT/pmc_f3.log 2>&1 || { echo "FETCH_SIZE pass (c3) failed"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w3 -o pmc -f csv -- python3 bench.py $PB3 > $OUT/pmc_w3.log 2>&1 || { echo "WRITE_SIZE pass (c3) failed"; exit 1; }
python3 tools/pmc_traffic.py $OUT/pmc_f3 $OUT/pmc_w3 - > $OUT/${R}_pmc_c3_traffic.txt
export XDFM_GATHER_B=4096,8192,65536
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fg -o pmc -f csv -- python3 tools/gather_scale.py > $OUT/pmc_fg.log 2>&1 || { echo "FETCH_SIZE pass (gather) failed"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_wg -o pmc -f csv -- python3 tools/gather_scale.py > $OUT/pmc_wg.log 2>&1 || { echo "WRITE_SIZE pass (gather) failed"; exit 1; }
python3 tools/pmc_traffic.py $OUT/pmc_fg $OUT/pmc_wg - grid > $OUT/${R}_pmc_gather_traffic.txt
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mv -o pmc -f csv -- python3 tools/vocab_ce_probe.py 1024 100000 26 fused > $OUT/pmc_mv.log 2>&1 && python3 tools/pmc_summary.py $OUT/pmc_mv vx_ > $OUT/${R}_pmc_vocab_ce_mfma.txt || exit 1
echo "profiles done"
