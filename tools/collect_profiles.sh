#!/bin/bash
# Run ON THE GPU BOX (via gpurun): bench + rocprofv3 kernel stats + PMC passes for the headline step kernel.
# Usage: tools/collect_profiles.sh <tag>     (writes under gpurun_out/<tag>/)
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots":
# FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined with sys/hip/hsa tracing).
set -o pipefail
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 500 --warmup 50 --min-seconds 0.1 --no-cpu-baseline > $OUT/stats.log 2>&1 || exit 1
B="python3 $R/bench.py --steps 100 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-fused"      # short: the counter CSVs hold one row per launch and counter
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $B > $OUT/pmc_sq2.log 2>&1 || exit 1
for wl in boat_race_ex safe_interruptibility boat_race firemaker_ex_ma island_navigation_ex_ma mixed island_navigation distributional_shift absent_supervisor side_effects_sokoban conveyor_belt rocks_diamonds tomato_watering friend_foe whisky_gold aintelope_savanna; do
  python3 $R/bench.py --workload $wl --steps 1000 --warmup 100 --no-cpu-baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err || exit 1
done
for n in 131072 262144 1048576; do
  python3 $R/bench.py --envs $n --steps 500 --warmup 50 --no-cpu-baseline > $OUT/bench_island_n$n.json 2> $OUT/bench_island_n$n.err || exit 1
done
if [ -x $R/tools/diag/stamp_probe.bin ]; then $R/tools/diag/stamp_probe.bin 65536 > $OUT/phase_stamps.txt 2>&1; $R/tools/diag/stamp_probe.bin 1048576 >> $OUT/phase_stamps.txt 2>&1; fi
# round 3: the side kernels of the step path (derived statistics, RGB / layers, agent windows) on their own, the firemaker round
# kernel with the agents' windows written by the same launch, and the mixed suite as one heterogeneous launch per step
python3 $R/tools/diag/side_probe.py $OUT/side_probe.json > $OUT/side_probe.txt 2>&1 || exit 1       # HIP-event timings, unprofiled
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_side -- python3 $R/tools/diag/side_probe.py > $OUT/side_probe_rocprof.txt 2>&1 || exit 1   # kernel durations (the probe's own event timings are inflated under the profiler)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_firemaker -- python3 $R/bench.py --workload firemaker_ex_ma --steps 300 --warmup 30 --min-seconds 0.2 --no-cpu-baseline --no-fused > $OUT/stats_firemaker.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_mixed -- python3 $R/bench.py --workload mixed --steps 500 --warmup 50 --min-seconds 0.2 --no-cpu-baseline > $OUT/stats_mixed.log 2>&1 || exit 1
python3 $R/tools/diag/host_probe.py > $OUT/host_probe.txt 2>&1          # host vs end-to-end time of every way to issue one step
python3 $R/tools/diag/agent_views_probe.py > $OUT/agent_views_probe.txt 2>&1
python3 $R/tools/diag/zoo_vector_probe.py > $OUT/zoo_vector_probe.txt 2>&1
for wl in firemaker_ex_ma aintelope_savanna; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_$wl -- python3 $R/bench.py --workload $wl --steps 60 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-fused > $OUT/pmc_$wl.log 2>&1 || exit 1
done
echo done
