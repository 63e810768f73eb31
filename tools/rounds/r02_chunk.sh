mkdir -p gpurun_out/r02m
for C in 8 32 128 512; do
  echo "== copy_chunk_mb $C" >> gpurun_out/r02m/chunk.txt
  timeout 250 python bench.py --steps 3 --warmup 1 --uniform-steps 0 --no-cpu-baseline --no-two-in-flight --no-check --copy-chunk-mb $C 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['boundary']; print(d['ms_per_step'], b['ms_per_proof'], b['frac_of_resident_value'], b['one_caller_ms_per_proof'])" >> gpurun_out/r02m/chunk.txt 2>&1
done
cat gpurun_out/r02m/chunk.txt
