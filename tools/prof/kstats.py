"""Summarise a rocprofv3 kernel_stats csv: name (short), calls, avg us, pct."""
import csv, sys, glob
for f in sys.argv[1:]:
    for path in glob.glob(f, recursive=True):
        print("==", path)
        for r in csv.DictReader(open(path)):
            nm = r["Name"].split("(")[0][-60:]
            print(f'{nm:60s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.2f} min {float(r["MinNs"])/1e3:8.2f} pct {r["Percentage"]}')
