#!/bin/bash
# Round-end refresh of the xDeepFMPro evidence on the GPU box (from the repo root): full GPU test suite, kernel stats and bench
# line of --workload criteo_pro, MFMA-busy of the K9 kernels.
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
B="--steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-extras"
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/kt_pro -o kt -f csv -- python3 bench.py $B --workload criteo_pro > gpurun_out/r03/kt_pro.json 2> gpurun_out/r03/kt_pro.err
MS=$(tail -1 gpurun_out/r03/kt_pro.json | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')
python3 tools/profile_summary.py gpurun_out/r03/kt_pro auto gpurun_out/r03/r03_kernel_stats_pro.md "bench.py $B --workload criteo_pro under rocprofv3 --kernel-trace --stats; bench line of this run: $MS"
python bench.py --workload criteo_pro --no-cpu-baseline --no-alt --steps 20 > gpurun_out/r03_bench_pro.json 2> gpurun_out/r03_bench_pro.err; echo "pro rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/r03/pmc_mv -o pmc -f csv -- python3 tools/vocab_ce_probe.py 1024 100000 26 fused > gpurun_out/r03/pmc_mv.log 2>&1 && python3 tools/pmc_summary.py gpurun_out/r03/pmc_mv vx_ > gpurun_out/r03/r03_pmc_vocab_ce_mfma.txt
grep -A5 "vx_hs\|vx_ws" gpurun_out/r03/r03_pmc_vocab_ce_mfma.txt | grep "^_Z\|utilisation"
