mkdir -p gpurun_out/r02c
tools/bin/fe52_bench > gpurun_out/r02c/fe52.txt 2>&1
python tools/cpu_probe.py 22 > gpurun_out/r02c/cpu_probe.txt 2>&1
B="python bench.py --steps 3 --warmup 1 --no-check --no-boundary --no-cpu-baseline --uniform-steps 0"
for cfg in "--chunk 32" "--chunk 48" "--chunk 64" "--window 21" "--window 22" "--window 21 --chunk 64" "--streams 2"; do
  echo "== $cfg" >> gpurun_out/r02c/sweep.txt
  timeout 200 $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['phases_ms_per_proof'])" >> gpurun_out/r02c/sweep.txt 2>&1
done
timeout 200 $B --config zkpor500_200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zkpor500_200', d['value'], d['ms_per_step'], d['phases_ms_per_proof'])" >> gpurun_out/r02c/sweep.txt 2>&1
cat gpurun_out/r02c/fe52.txt gpurun_out/r02c/cpu_probe.txt gpurun_out/r02c/sweep.txt
