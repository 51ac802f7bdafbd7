"""Per-layer HBM traffic of the conv launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units;
FETCH_SIZE doubled: on gfx950 it counts 128-byte requests as 64 bytes for wide coalesced loads).
usage: python scratch/pmc_conv.py <fetch_dir> <write_dir> <convs_per_pass> <first_full_pass> <n_full_passes> [out.json]"""
import csv, glob, json, sys
fd, wd, n, first, npass = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])

def load(d, counter):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("conv_igemm" in r["Kernel_Name"] or "stem_conv" in r["Kernel_Name"]):
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"]))
    rows.sort()
    return rows

fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
assert len(fe) == len(wr) and len(fe) % n == 0, (len(fe), len(wr))
per_layer = []
for j in range(n):
    f = sum(fe[(first + p) * n + j][1] for p in range(npass)) / npass * 1024 * 2
    w = sum(wr[(first + p) * n + j][1] for p in range(npass)) / npass * 1024
    per_layer.append((j, f, w, fe[first * n + j][2]))
tot_f = sum(x[1] for x in per_layer); tot_w = sum(x[2] for x in per_layer)
print(f"conv launches per pass {n}; HBM read {tot_f/1e9:.3f} GB  write {tot_w/1e9:.3f} GB  per launch {(tot_f+tot_w)/n/1e6:.1f} MB")
for j, f, w, k in per_layer:
    print(f"{j:3d} read {f/1e6:9.1f} MB  write {w/1e6:9.1f} MB  {k[k.find('<'):k.find('>')+1]}")
if len(sys.argv) > 6:
    json.dump(dict(convs_per_pass=n, read_bytes=tot_f, write_bytes=tot_w, bytes_per_launch=(tot_f + tot_w) / n,
                   note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; KB units; FETCH_SIZE doubled "
                        "(gfx950 counts the 128-byte requests of wide coalesced loads as 64 bytes)"), open(sys.argv[6], "w"), indent=1)
