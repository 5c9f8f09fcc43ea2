#!/bin/bash
# usage (GPU box): tools_profile_r3.sh <tag>  -> kernel trace + PMC passes of the 46 Mbp legs of the default bench
# (the 3.1 Gbp leg has a run of its own: tools_c4.sh), plus SQ / LDS passes for the build kernels
TAG=$1
ARGS="--steps 20 --warmup 3 --no-c5 --no-c4"
bash tools/tools_prof.sh $TAG $ARGS || exit 1
bash tools/tools_pmc.sh $TAG fetch "FETCH_SIZE" $ARGS --no-callers || exit 1
bash tools/tools_pmc.sh $TAG write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" $ARGS --no-callers || exit 1
bash tools/tools_pmc.sh $TAG sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" $ARGS --no-callers || exit 1
bash tools/tools_pmc.sh $TAG lds "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" $ARGS --no-callers || echo "lds pass failed"
cd $GRAFT_REPO_ROOT
python3 tools/tools_profile_digest.py gpurun_out/$TAG $TAG
python3 tools/tools_pmc_summary.py gpurun_out/$TAG k_leaf_sort_wave k_leaf_sort k_part_scatter k_part_hist k_part_classify k_emit_planes k_rank k_build_sym k_make_text2 k_prepare_text k_isa_deal k_isa_fill k_record_text k_kmer_level k_kmer_last_two k_search > gpurun_out/$TAG/digest/${TAG}_pmc_summary.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete; find gpurun_out/$TAG -name "*_kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*counter_collection.csv" -size +8M -delete
