set -u
OUT=gpurun_out/r04u
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 100 python bench.py --gpus 2 --share-device --circuit 5,20,6 --steps 2 --warmup 1 --no-cpu-baseline --no-boundary --e2e-steps 2 > $OUT/two_ranks.json 2> $OUT/two_ranks.err; echo rc=$?
tail -3 $OUT/two_ranks.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04u/two_ranks.json"))
print(d["n_gpus"], d["value"], d["per_rank_ms_per_step"], d["checked"]["per_rank_ok_of_total"], d["end_to_end"]["value"], d["end_to_end"]["checked"], d["configs"].keys() if "configs" in d else None)
PY
