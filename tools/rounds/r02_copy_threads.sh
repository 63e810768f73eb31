mkdir -p gpurun_out/r02h
for T in 2 4 8 12; do
  for CFG in zkpor50_1380 zkpor500_200; do
    echo "== copy_threads $T $CFG" >> gpurun_out/r02h/copy_threads.txt
    timeout 200 python bench.py --steps 3 --warmup 1 --config $CFG --uniform-steps 0 --no-cpu-baseline --copy-threads $T 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); b=d['boundary']; print(d['ms_per_step'], b['ms_per_proof'], b['frac_of_resident_value'], b['one_caller_ms_per_proof'], b['checked_ok'], b['proofs'])" >> gpurun_out/r02h/copy_threads.txt 2>&1
  done
done
cat gpurun_out/r02h/copy_threads.txt
