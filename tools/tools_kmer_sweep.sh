for K in 13 14 15; do
  BWT_MI355X_KMER=$K timeout -k 10 200 python -u bench.py --no-cpu --no-c5 --no-c4 --no-callers --steps 100 > gpurun_out/kexp_$K.json 2> gpurun_out/kexp_$K.err || { tail -3 gpurun_out/kexp_$K.err; exit 1; }
  python - <<PY
import json; d=json.load(open("gpurun_out/kexp_$K.json"))
print("K=$K reads/s %.4g ms/step %.3f kernel_ms %.3f  arrays_ms %.2f index_bytes %.0f MB loads %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["sa_build"]["search_arrays"]["wall_ms"], d["index_bytes"]/1e6, d["roofline"]["loads"]))
PY
done
