#!/bin/bash
# usage (GPU box): tools_reference_programs.sh -> the reference's filetest, searchtest, index_test and fmitest, compiled from
# its sources and linked with libbwt_mi355x.so (oracle/Makefile: refprogs), on a 1 Mbp genome
set -u
cd $GRAFT_REPO_ROOT
W=gpurun_out/refprogs; mkdir -p $W
python3 - <<'PY' || exit 1
import sys; sys.path.insert(0, '.')
from oracle import load_oracle
from bwt_amd import synth
o = load_oracle(); n = 1000000
t = o.gen_genome(n, 77)
open('gpurun_out/refprogs/g1m.txt', 'w').write("".join("ACGT"[b] for b in synth.unpack(t, n)))
PY
timeout -k 10 60 bwt_amd/drivers/build_index $W/g1m.txt $W/g1m.idx > $W/build.out 2> $W/build.err || exit 1
for p in filetest searchtest index_test fmitest; do
  args="$W/g1m.txt"; [ $p = index_test ] && args="$W/g1m.txt $W/g1m.idx"; [ $p = fmitest ] && args="1000000"
  S=$(date +%s%N)
  timeout -k 10 170 oracle/_ref/${p}_on_mi355x $args > $W/$p.out 2> $W/$p.err; rc=$?
  E=$(date +%s%N)
  echo "$p: rc=$rc, $(( (E - S) / 1000000 )) ms wall, stdout: $(grep -c Warning $W/$p.out) 'Warning: multiple matches' lines, $(grep -c 'Ruh roh' $W/$p.out) 'Ruh roh' lines, $(grep -v -c -e Warning -e 'Ruh roh' $W/$p.out) other lines"
  tail -2 $W/$p.err; [ $p = fmitest ] && cat $W/$p.out
done
