"""Average PMC counters over the conv_igemm dispatches of a rocprofv3 csv dir: python scratch/pmc_kernel.py <dir>"""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:36s} {v / n:16.1f}")
