cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "" $@; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_track$L -- python $R/bench.py --stage track --steps 20 --warmup 6 --no-cpu-baseline --no-dense --pipeline 1 ${L:+--lib build_exp/liblcs_$L.so} > $R/gpurun_out/kst_track$L.log 2>&1
find $R/gpurun_out/kst_track$L -type f ! -name "*kernel_stats.csv" -delete
echo "== ${L:-in-tree}"; python - <<PY
import csv,glob,os
fn=sorted(glob.glob("$R/gpurun_out/kst_track$L/*/*kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in list(csv.reader(open(fn)))[1:5]:
    print("%-30s calls %5s avg %9.1f us"%(r[0][:30], r[1], float(r[3])/1e3))
PY
done
