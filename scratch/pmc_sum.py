"""Sum a PMC counter per kernel family from rocprofv3 csv output: python scratch/pmc_sum.py <dir> <counter>"""
import csv, glob, sys, collections
d, counter = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        fam = "conv_igemm" if "conv_igemm" in k else k.split("(")[0].split("::")[-1][:40]
        agg[fam][0] += 1
        agg[fam][1] += float(r["Counter_Value"])
for fam, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{fam:42s} launches {n:6d}  sum {v:16.1f}  per launch {v / n:14.2f}")
