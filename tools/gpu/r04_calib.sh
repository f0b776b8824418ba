# FETCH_SIZE against a known byte count for the access shapes of the solve kernels (tools/microbench/fetch_calib.hip)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04f
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pmc_cal && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_cal -o run -- $GRAFT_REPO_ROOT/tools/microbench/fetch_calib > $GRAFT_REPO_ROOT/$OUT/calib_stdout.txt 2>&1
cd $GRAFT_REPO_ROOT
python - > $OUT/fetch_calib.txt <<'PY'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob('/tmp/pmc_cal/**/*.db', recursive=True)[0])
rows = db.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection order by dispatch_id").fetchall()
true = {"k_wide16": 1 << 30, "k_flat8": 1 << 30, "k_slab<8>": 1 << 30, "k_slab<16>": 1 << 30}
seen = {}
print("kernel (second launch of each shape)            FETCH_SIZE counter   bytes (x1024?)   true bytes   counter/true  duration_us  GB/s(true)")
for name, cname, val, dur, did in rows:
    m = re.search(r"k_(wide16|flat8|slab<\d+>)", name)
    if not m or cname != "FETCH_SIZE": continue
    k = "k_" + m.group(1)
    seen[k] = seen.get(k, 0) + 1
    tag = k
    tb = true[k]
    if k == "k_slab<8>":
        idx = seen[k]
        if idx % 2 == 0: tag, tb = "k_slab<8> +32B", (4096 - 8) * 32768 * 8
    # FETCH_SIZE is reported in kilobytes by rocprofv3's derived metric
    for unit, f in (("KB", 1024.0),):
        print("%-44s %18.1f %16.0f %12d %12.3f %12.1f %10.1f" % (tag + " #%d" % seen[k], val, val * f, tb, val * f / tb, dur / 1e3, tb / max(dur, 1)))
PY
cat $OUT/fetch_calib.txt
