#!/bin/bash
# What could an XCD-aware item mapping win at most?  With HA_BENCH_SAME_BATCH=1 every step names the SAME batch: a key's apply item
# sits at the same list position in every launch, i.e. in the same workgroup, i.e. on the same XCD (workgroups go to the XCDs
# round-robin), and the row it reads is the row its predecessor WROTE one launch earlier on that XCD -- perfect affinity without
# touching the queue builder.  If the L2 read hits do not rise even then, no mapping can raise them: between two touches of a row
# ~5 MB of gradient and output rows pass through every 4 MiB L2.  Counters of ha::qapply_kernel, one --pmc pass per group.
O=$GRAFT_REPO_ROOT/gpurun_out/l2aff; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 64 --warmup 32 --no-cpu-baseline --no-kernel-pass --no-cache-tier --no-cold-tier --no-laia --no-wide"
for v in 0 1; do
  export HA_BENCH_SAME_BATCH=$v
  for g in "TCC_HIT TCC_MISS TCC_REQ TCC_READ" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_64B"; do
    n=$(echo $g | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $g --output-format csv -d $O/d${v}_$n -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $O/d${v}_$n.log 2>&1
  done
  cd $GRAFT_REPO_ROOT
  for i in 1 2; do python3 bench.py --no-cpu-baseline --no-cache-tier --no-cold-tier --no-laia --no-wide 2>/dev/null | python3 tools/ab_line.py same_batch_$v long >> $O/times.txt; done
  cd /tmp
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee gpurun_out/l2aff/summary.txt
import csv, glob, collections
for v in (0, 1):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/l2aff/d%d_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "qapply_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    m = {k: a[0] / a[1] for k, a in acc.items()}
    print("same batch every step: %d: per launch " % v + "  ".join("%s %.0f" % (k, m[k]) for k in sorted(m)))
    if "TCC_EA0_RDREQ" in m:
        print("    fabric reads %.1f MB (128-B requests), fabric writes %.1f MB (64 B), L2 read requests %.0f k -> read hit rate ~ %.0f %%"
              % (m["TCC_EA0_RDREQ"] * 128 / 1e6, m.get("TCC_EA0_WRREQ", 0) * 64 / 1e6, m.get("TCC_READ", 0) / 1e3,
                 100 * max(0.0, 1 - m["TCC_EA0_RDREQ"] / max(m.get("TCC_READ", 1), 1))))
print(open("gpurun_out/l2aff/times.txt").read())
PY
find $O -name "*.csv" -size +3M -delete
