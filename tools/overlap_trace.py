import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
last = rows[-60:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = 0; ov = 0
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("ctl::", "").replace("void ", "")[:34]
    print("%-34s q%s  start %9.1f us  dur %8.1f us  %s" % (name, r.get("Queue_Id", "?"), s / 1e3, (e - s) / 1e3, "OVERLAP %.1f us" % ((prev_end - s) / 1e3) if s < prev_end else ""))
    prev_end = max(prev_end, e)
