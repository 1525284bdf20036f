#!/bin/bash
# PMC counters of the local-energy A/B probe (own pass, no tracing): tools/pmc_eloc.sh <tag> <counters...>
# (FF_ELOC_KERNEL selects the kernel; appends one JSON object per matching kernel to gpurun_out/pmc_eloc.jsonl)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/probes/eloc_ab.py > $out.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv, glob, collections, json, os
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
with open(os.path.join("$GRAFT_REPO_ROOT", "gpurun_out", "pmc_eloc.jsonl"), "a") as fo:
    for k in acc:
        if ("eloc_rows" in k or "eloc_mfma" in k or "ode_fwd_kernel<6, 2, 2" in k) and "true" in k:
            d = {"kernel": k, "tag": "$tag", "per_launch": {c: v / n[(k, c)] for c, v in acc[k].items()}}
            print(json.dumps(d)); fo.write(json.dumps(d) + "\n")
PY
