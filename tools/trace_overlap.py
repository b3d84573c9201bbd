"""Prints start/end (ms, relative) of every solve kernel in a rocprofv3 --kernel-trace CSV: how many batches really overlap."""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "solve_kernel" in r["Kernel_Name"]]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    print(f"queue {r.get('Queue_Id','?'):>3} stream {r.get('Stream_Id','?'):>3}  start {s:8.2f}  end {e:8.2f}  dur {e - s:7.2f} ms")
