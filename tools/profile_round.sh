#!/bin/bash
# profile_round.sh - GPU-BOX TOOLING: the rocprofv3 passes behind the numbers of a round.
#   bash tools/profile_round.sh r02
# writes gpurun_out/prof_<tag>/ (raw) and gpurun_out/<tag>_*.{csv,json} (summaries to copy into
# profiles/). Counter passes are separate runs with --pmc only (no trace domains beside them).
set -o pipefail
TAG=${1:-r05}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-order --no-latency"
# 1. kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
STATS=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" gpurun_out/${TAG}_kernel_stats.csv
cp $OUT/bench_under_rocprof.json gpurun_out/${TAG}_bench_line_under_rocprof.json
echo "stats done"
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in passes of their own
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
echo "write done"
python3 tools/pmc_summary.py $OUT/fetch $OUT/write 32000 "bench.py --steps 5 --warmup 2 ($TAG); K1a/K1b/K3: 32000 propagator steps per nominal dispatch" > gpurun_out/${TAG}_pmc_hbm.json
# 3. SQ counters (two passes of <= 8 counters)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $CMD > /dev/null 2> $OUT/sq1.err
echo "sq1 done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
echo "sq2 done"
python3 tools/pmc_counters.py $OUT/sq1 $OUT/sq2 > gpurun_out/${TAG}_pmc_sq.json
# 4. what the passes were taken from: bench.py labels the committed counters stale once these change
python3 -c "import bench, json; print(json.dumps({'k1a_sources': list(bench.K1A_SOURCES), 'k1a_sources_sha16': bench.source_digest(), 'command': '$CMD'}))" > gpurun_out/${TAG}_profile_meta.json
ls -la gpurun_out/${TAG}_*
