"""One steady-state step of a rocprofv3 --kernel-trace run as a timeline (start offset, duration, queue, kernel).
  python tools/step_timeline.py <dir with *_kernel_trace.csv> [step index]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void coattn_fwd_kernel")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
main_q = rows[a]["Queue_Id"]
busy = gap = 0.0
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if r["Queue_Id"] == main_q:
        if prev_end is not None:
            gap += max(0, s - prev_end) / 1e3
        busy += (e - s) / 1e3
        prev_end = e
    print("%8.1f %7.1f q=%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], r["Kernel_Name"][:70]))
n_main = sum(1 for r in rows[a:b] if r["Queue_Id"] == main_q)
print("step %.1f us; main queue: %d kernels, busy %.1f us, gaps %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, n_main, busy, gap))
