# per-kernel averages of one rocprofv3 --pmc pass:  python profiles/tools/counter_summary.py DIR [kernel-substring]
import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
want = sys.argv[2] if len(sys.argv) > 2 else '_lm'
disp = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if want in k:
        disp[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
agg = collections.defaultdict(list)
for (k, c, d), v in disp.items():
    agg[(k, c)].append(v)
for (k, c), vals in sorted(agg.items()):
    print("%-40s %-28s %.4e  (%d launches)" % (k, c, sum(vals) / len(vals), len(vals)))
