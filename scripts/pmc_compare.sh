# SQ counters of one weight-gradient layer against one input-gradient layer of the same tile structure (diagnostic; GPU box)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CNT="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
for spec in "U1.wgrad 0" "U1.wgrad 0x40000" "U1.dgrad 0" "U2.fwd 0"; do
  set -- $spec
  tag=$(echo "$1_$2" | tr '.' '_')
  timeout -k 10 120 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/scripts/pmc_layer.py $1 $2 4 > $R/gpurun_out/pmc_$tag.log 2>&1
  f=$(ls $R/gpurun_out/pmc_$tag/*/*counter_collection.csv | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys, collections
f, tag = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    if "tapgemm" in k or "wgrad" in k or "halo" in k:
        wc = v.get("SQ_WAVE_CYCLES", 1)
        print(tag, k[:48], " ".join("%s=%.3g" % (c.replace("SQ_", ""), v[c]) for c in sorted(v)),
              "| conflict/lds_active=%.3f wait_any/wave=%.2f lds_wait/wave=%.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_LDS", 0) / wc))
PY
  rm -rf $R/gpurun_out/pmc_$tag
done
