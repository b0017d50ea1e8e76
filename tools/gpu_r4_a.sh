#!/bin/bash
# Round 4, call A: new GPU tests (reference suite under the product library, tilted fixtures), the face-kernel
# scheduling / ordering lab with PMC passes for the default and the best variant, and a baseline bench line.
export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r4a
mkdir -p $O
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" >> $O/timeline.log; }
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "reference_suite or tilted" > $O/pytest_subset.log 2>&1
echo "pytest exit $?" >> $O/pytest_subset.log; tail -5 $O/pytest_subset.log; stamp tests
timeout 600 python tools/face_lab.py 69 > $O/face_lab.log 2> $O/face_lab.err
cat $O/face_lab.log; tail -3 $O/face_lab.err; stamp lab
best=$(grep -v "^base" $O/face_lab.log | grep "bit-identical" | sort -k3 -n | head -1 | awk '{print $1}')
echo "best variant: $best" | tee -a $O/face_lab.log
cd /tmp
for v in base $best; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -c1-9 | tr ' ' '_')
    PFV_LAB_ONE=$v timeout 300 rocprofv3 --pmc $set -d $R/$O/pmc_${v}_$tag -o c --output-format csv -- python $R/tools/face_lab.py 69 > $R/$O/pmc_${v}_$tag.log 2>&1
  done
done
cd "$R"
python - "$O" <<'PY' > $O/pmc_face.txt 2>&1
import collections, csv, glob, sys
o = sys.argv[1]
for d in sorted(glob.glob(o + "/pmc_*/")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            kn = row.get("Kernel_Name", "")
            key = "face" if "k_face_pipe" in kn else ("node40" if "launch_node_class_reg<64, 3, 40" in kn else None)
            if key:
                agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: (sorted(v)[len(v) // 2], len(v)) for c, v in cs.items()})
PY
cat $O/pmc_face.txt
rm -rf $O/pmc_*/
stamp pmc
timeout 400 python bench.py --no-cpu-baseline --no-extra-configs --steps 5 > $O/bench_base.json 2> $O/bench_base.err
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
try:
    d = json.loads([l for l in open(f"{o}/bench_base.json") if l.startswith("{")][-1])
    ph = {k[:-3]: round(v, 2) for k, v in d["assembly"]["phases_ms"].items()}
    print(f"bench ms/step {d['ms_per_step']:.2f} its {d['config']['iterations']} asm {d['assembly']['ms']:.2f} {ph} amg_setup {d['config']['amg']['setup_ms']:.2f}")
except Exception as e:
    print("bench FAILED", e, open(f"{o}/bench_base.err").read()[-800:])
PY
stamp bench
cat $O/timeline.log
