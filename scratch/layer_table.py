"""Joins the four rocprofv3 runs of scratch/layer_counters.sh into one per-layer table.  Dispatches are matched by
their order inside a pass over the program (the program launches the same kernels in the same order every time);
the LAST pass of every run is used.  python3 scratch/layer_table.py <dir>"""
import csv, glob, json, sys
D = sys.argv[1]
meta = json.load(open(D + "/ops.json"))
ops = meta["ops"]
n = len(ops)


def rows(sub, pattern):
    out = []
    for f in glob.glob(f"{D}/{sub}/**/{pattern}", recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


tr = rows("trace", "*kernel_trace.csv")
tr.sort(key=lambda r: int(r["Start_Timestamp"]))


def ours(name: str) -> bool:
    """kernels of libyolo_amd.so (not torch's fills / random numbers, not the runtime's copy kernels)"""
    return "at::" not in name and "__amd_rocclr" not in name and ("yamd::" in name or "anonymous namespace" in name or "_GLOBAL__N_" in name)


mine = [r for r in tr if ours(r["Kernel_Name"])]
assert len(mine) % n == 0 and len(mine) >= n, (len(mine), n)
last = mine[-n:]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in last]     # us


def counters(sub):
    rs = rows(sub, "*counter_collection.csv")
    by = {}
    for r in rs:
        if not ours(r["Kernel_Name"]):
            continue
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)
    assert len(ids) % n == 0 and len(ids) >= n, (sub, len(ids), n)
    return [by[i] for i in ids[-n:]]


sq, fe, wr = counters("pmc_sq"), counters("pmc_fetch"), counters("pmc_write")
print(f"# {meta['model']} batch {meta['batch']}: per-launch table of the conv program (last of the profiled passes)")
print("# us = kernel-trace duration; TF = algorithmic 2*MACs / us; mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (us * 2.4 GHz * 1024 SIMDs);")
print("# clk = GRBM_GUI_ACTIVE / 8 / duration of the counter pass (GHz, reads high on short launches); wait% = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls), ldsw% = SQ_WAIT_INST_LDS, park% = SQ_WAIT_ANY (s_waitcnt / barrier), act% = SQ_ACTIVE_INST_ANY;")
print("# hbmR / hbmW = FETCH_SIZE * 2 (gfx950: 128-byte requests counted as 64) / WRITE_SIZE, KB units, in MB; alg = algorithmic bytes in MB")
print(f"{'kind':5s} {'name':34s} {'us':>8s} {'TF':>7s} {'mfma%':>6s} {'wait%':>6s} {'ldsw%':>6s} {'park%':>6s} {'act%':>6s} {'hbmR':>8s} {'hbmW':>8s} {'alg':>8s} {'GB/s':>7s}  shape")
tot = dict(us=0.0, fl=0.0, busy=0.0, r=0.0, w=0.0, alg=0.0)
for op, us, s, f, w in zip(ops, dur, sq, fe, wr):
    busy = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    mf = busy / (us * 1e-6 * 2.4e9 * 1024) * 100
    wc = s.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    wait = s.get("SQ_WAIT_INST_ANY", 0.0) / wc * 100
    ldsw = s.get("SQ_WAIT_INST_LDS", 0.0) / wc * 100
    park = s.get("SQ_WAIT_ANY", 0.0) / wc * 100
    act = s.get("SQ_ACTIVE_INST_ANY", 0.0) / wc * 100
    r = f.get("FETCH_SIZE", 0.0) * 1024 * 2 / 1e6
    wv = w.get("WRITE_SIZE", 0.0) * 1024 / 1e6
    is_conv = op["kind"] in ("conv", "conv_cat2", "stem", "stem2", "bneck", "c3")
    if is_conv:
        for k, v in (("us", us), ("fl", op["flops"]), ("busy", busy), ("r", r), ("w", wv), ("alg", op["bytes"] / 1e6)):
            tot[k] += v
    print(f"{op['kind']:5s} {op['name'][:34]:34s} {us:8.1f} {op['flops'] / us / 1e6:7.1f} {mf:6.1f} {wait:6.1f} {ldsw:6.1f} {park:6.1f} {act:6.1f} {r:8.1f} {wv:8.1f} "
          f"{op['bytes'] / 1e6:8.1f} {(r + wv) / us * 1e3:7.0f}  {op['desc']}")
n_conv = sum(1 for op in ops if op["kind"] in ("conv", "conv_cat2", "stem", "stem2", "bneck", "c3"))
if len(sys.argv) > 2:
    json.dump(dict(model=meta["model"], batch=meta["batch"], convs_per_pass=n_conv, conv_launches=meta.get("conv_launches"), read_bytes=tot["r"] * 1e6, write_bytes=tot["w"] * 1e6,
                   bytes_per_launch=(tot["r"] + tot["w"]) * 1e6 / n_conv, conv_us=tot["us"], conv_flops=tot["fl"],
                   mfma_busy_frac_at_2p4GHz=tot["busy"] / (tot["us"] * 1e-6 * 2.4e9 * 1024),
                   note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate passes over the conv "
                        "program of the benchmark configuration (scratch/layer_counters.sh); KB units; FETCH_SIZE doubled (gfx950 "
                        "counts the 128-byte requests of wide coalesced loads as 64 bytes); durations from a --kernel-trace pass"),
              open(sys.argv[2], "w"), indent=1)
print(f"# conv launches: {tot['us'] / 1e3:.3f} ms, {tot['fl'] / tot['us'] / 1e6:.1f} TFLOP/s, MFMA busy {tot['busy'] / (tot['us'] * 1e-6 * 2.4e9 * 1024) * 100:.1f} %, "
      f"HBM read {tot['r'] / 1e3:.2f} GB + write {tot['w'] / 1e3:.2f} GB vs algorithmic {tot['alg'] / 1e3:.2f} GB")
