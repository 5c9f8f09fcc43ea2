#!/bin/bash
# GPU box: the c4 form of bench.py (3.1 Gbp, 100 M x 100 bp, search + locate) on one GPU: plain run,
# kernel trace, and the two PMC passes for HBM traffic of its search kernel.
set -u
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -u bench.py --config c4 --steps 20 --warmup 2 > $OUT/c4_bench.json 2> $OUT/c4_bench.err || { tail -5 $OUT/c4_bench.err; exit 1; }
python3 - <<PY
import json; d=json.load(open("$OUT/c4_bench.json"))
print("c4 reads/s %.4g ms/step %.2f frac %.3f checks %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["checks"]))
print("c4 build", d["sa_build"]["wall_ms"], d["sa_build"]["search_arrays_ms"], "two calls", d["search_then_locate"])
PY
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 20 --warmup 2 > $OUT/trace_bench.json 2> $OUT/trace.err
echo "trace rc=$?"
timeout -s KILL 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 3 --warmup 1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "pmc fetch rc=$?"
timeout -s KILL 280 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 3 --warmup 1 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
echo "pmc write rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/tools_pmc_summary.py $OUT k_search k_locate_sa k_leaf_sort k_part_scatter k_part_hist k_isa_deal k_isa_fill k_kmer_level 2>&1 | tail -30
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/c4_kernel_stats.csv
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
