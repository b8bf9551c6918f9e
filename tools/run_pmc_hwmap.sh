#!/bin/bash
# L2 hit / miss counters of hw_kernel for the two workgroup orders (hw_map 0 / 1) -> gpurun_out/$1/hwmap_pmc.txt
O=gpurun_out/${1:-r02r}; mkdir -p $O; cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "0:0" "1:0"; do
  export CONFIGS=$cfg
  tag=map${cfg%%:*}
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_$tag -o p -- python3 tools/bench_hw_map.py 1 3 > $O/pmc_$tag.log 2>&1
  f=$(find $O/pmc_$tag -name "p_counter_collection.csv" | head -1)
  python3 - "$f" $tag >> $O/hwmap_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "hw_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[2], {k: (v[0] / v[1], v[1]) for k, v in acc.items()})
PY
done
cat $O/hwmap_pmc.txt
