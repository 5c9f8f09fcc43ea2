#!/bin/bash
# usage (GPU box): tools_asan_driver.sh  -> single_align and search_reads drivers built with host AddressSanitizer,
# run on the files of the program test; the reference-API layer (host/fmi_host.c) is compiled in, instrumented too
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/asan; mkdir -p $OUT
for d in single_align search_reads build_index; do
  gcc -O1 -g -fsanitize=address -fno-omit-frame-pointer -std=gnu11 -Iinclude -o $OUT/$d bwt_amd/drivers/$d.c bwt_amd/host/fmi_host.c \
      -Lbwt_amd -lbwt_mi355x -Wl,-rpath,$GRAFT_REPO_ROOT/bwt_amd -lpthread || exit 1
done
python3 - <<'PY' || exit 1
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import load_oracle
from helpers import single_align_inputs
os.makedirs("gpurun_out/asan/files", exist_ok=True)
single_align_inputs(load_oracle(), "gpurun_out/asan/files")
PY
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=0
F=$OUT/files
timeout -k 10 120 $OUT/build_index $F/genome.txt $F/genome.idx > $OUT/build.out 2> $OUT/build.err; echo "build_index rc=$?"
timeout -k 10 120 $OUT/single_align $F/genome.txt $F/genome.idx $F/reads.txt > $OUT/single.out 2> $OUT/single.err; echo "single_align rc=$?"
timeout -k 10 120 $OUT/search_reads $F/genome.txt $F/genome.idx $F/reads.txt > $OUT/search.out 2> $OUT/search.err; echo "search_reads rc=$?"
grep -l "AddressSanitizer" $OUT/*.err; sha256sum $OUT/single.out; tail -2 $OUT/single.err
