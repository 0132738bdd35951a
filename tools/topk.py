import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = r["Name"]
    n = n[:90]
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["Percentage"]):6.2f}% calls={r["Calls"]:>6} avg={float(r["AverageNs"])/1e3:8.1f} us  {n}')
print("total ms", tot / 1e6)
