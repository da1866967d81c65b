cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/w8prof; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w8prof -- python3 $GRAFT_REPO_ROOT/tools/time_attn.py $@ > /tmp/w8.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("/tmp/w8prof/**/*kernel_stats.csv",recursive=True)
print(open("/tmp/w8.log").read()[-600:] if not f else "")
for r in list(csv.DictReader(open(f[0])))[:10]:
    print(r["Name"][:80], r["Calls"], r["AverageNs"])
PY
