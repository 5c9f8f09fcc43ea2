#!/bin/bash
# usage (GPU box): tools_profile_all.sh <tag>   -> kernel trace + two PMC passes of the default bench
TAG=$1
bash tools/tools_prof.sh $TAG --steps 5 --warmup 2 || exit 1
bash tools/tools_pmc.sh $TAG fetch "FETCH_SIZE" --steps 5 --warmup 2 || exit 1
bash tools/tools_pmc.sh $TAG write "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" --steps 5 --warmup 2 || exit 1
bash tools/tools_pmc.sh $TAG sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" --steps 5 --warmup 2 || exit 1
cd $GRAFT_REPO_ROOT
python3 tools/tools_profile_digest.py gpurun_out/$TAG $TAG
