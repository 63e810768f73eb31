# where do the level-1 kernels wait?  SQ wave-cycle counters for the G1 / G2 accumulation and the NTT passes (one PMC pass each)
OUT=gpurun_out/r02o
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TA_[A-Z_0-9]+)\b" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
grep -E "SQ_WAIT|SQ_ACTIVE|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_INST_CYCLES|SQ_INSTS_VMEM|SQ_INSTS_LDS|SQ_INSTS_SALU|SQ_INSTS_VALU|LATENCY|TCP_PENDING|TCC_EA_RDREQ|TCC_HIT|TCC_MISS|TA_BUSY|TCP_TCC_READ" $OUT/counters.txt | tr '\n' ' '
for SET in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  T=$(echo $SET | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/$T -o pmc -- python bench.py --log2 26 --steps 1 --warmup 0 --timed-only > /dev/null 2> $OUT/$T.err
  python - "$OUT/$T/pmc_counter_collection.csv" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
try:
    for r in csv.DictReader(open(sys.argv[1])):
        for k in ("k_acc_level1_fp29", "k_acc_level1_g2pair29", "k_ntt_pass29", "k_reduce_level29", "k_acc_levelN29"):
            if k in r["Kernel_Name"]:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        print(k, {c: (round(v / 1e9, 3), cnt[(k, c)]) for c, v in d.items()})
except Exception as e:
    print("no data:", e)
PY
done
