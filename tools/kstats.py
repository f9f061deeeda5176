"""Print name / calls / average ms of the top kernels of a rocprofv3 kernel_stats.csv (argv[1] = directory or file)."""
import csv, glob, os, sys
p = sys.argv[1]
f = p if os.path.isfile(p) else sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= int(sys.argv[2]) if len(sys.argv) > 2 else i >= 6:
        break
    name = r["Name"].split("(")[1 if r["Name"].startswith("(anonymous") else 0][:48] if not r["Name"].startswith("void") else r["Name"][5:60]
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:9.4f} ms')
