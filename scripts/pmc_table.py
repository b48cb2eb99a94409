"""Averages rocprofv3 --pmc counter_collection CSVs per kernel: python scripts/pmc_table.py <pattern> <dir> [<dir> ...]"""
import collections, csv, glob, sys
pat = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if pat in k:
                print(d.split("/")[-1], k, {c: round(sum(x) / len(x)) for c, x in v.items()})
