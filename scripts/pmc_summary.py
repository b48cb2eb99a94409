import csv, glob, collections, sys
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            agg[r["Kernel_Name"][:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if "dm_" not in k: continue
            print(d.split('/')[-1],k,{c:round(sum(x)/len(x)) for c,x in v.items()})
