mkdir -p gpurun_out/r02j
B="python bench.py --steps 4 --warmup 2 --no-boundary --no-cpu-baseline --no-two-in-flight"
for T in 1 4 3 2; do
  echo "== tables $T" >> gpurun_out/r02j/tables.txt
  timeout 300 $B --tables $T 2>gpurun_out/r02j/err_$T.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['value_uniform'], d['uniform']['ms_per_step'], d['checked']['ok'], d['checked']['proofs'], d['phases_ms_per_proof'])" >> gpurun_out/r02j/tables.txt 2>&1
done
cat gpurun_out/r02j/tables.txt; tail -3 gpurun_out/r02j/err_4.txt
