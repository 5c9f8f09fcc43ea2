#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from oracle import load_oracle
from bwt_amd import synth
o=load_oracle(); n=1000000
t=o.gen_genome(n, 77)
open('gpurun_out/g1m.txt','w').write("".join("ACGT"[b] for b in synth.unpack(t,n)))
PY
for p in filetest searchtest; do
  S=$(date +%s.%N); timeout -k 10 170 oracle/_ref/${p}_on_mi355x gpurun_out/g1m.txt > gpurun_out/$p.out 2> gpurun_out/$p.err; echo "$p rc=$? $(echo "$(date +%s.%N) - $S" | bc) s"
  tail -3 gpurun_out/$p.err; grep -c Warning gpurun_out/$p.out; grep -c "Ruh roh" gpurun_out/$p.out
done
