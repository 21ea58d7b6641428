R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SDR_OV_TRACE=1 timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d /tmp/fmtr -- python3 $R/tools/exp_fm_ov.py 2>&1 | grep -E "FM|side-stream"
python3 - <<'PY'
import csv, glob
f = max(glob.glob("/tmp/fmtr/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getsize(p))
rows = [r for r in csv.DictReader(open(f))]
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("mfma2", "k_frontend_win", "k_fir_fff", "k_deemph", "k_resample", "k_deemph_resample", "k_tail"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-60]["Start_Timestamp"])
for r in rows[-60:]:
    print("%-22s q%-3s start %9.1f us  dur %7.1f us" % (r["Kernel_Name"][:22], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
