import csv,collections,sys,glob
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0].replace("void ","")
    if "lc" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in v.items(): print("   ",c, "%.3e"%(sum(vals)/len(vals)))
