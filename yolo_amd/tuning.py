"""Per-layer conv tilings measured on the device (``Program.autotune``) and kept in a JSON table next to this
file.  A layer without an entry uses the default rule of ``yamd_conv_desc_init``.

  python -m yolo_amd.tuning --model v5s [--batch 256] [--out yolo_amd/tuned_gfx950.json]
"""
import json
import os
from typing import Dict, List, Optional, Tuple

# YAMD_TUNING_TABLE: another table file (A/B runs); YAMD_NO_TUNING: the built-in rule for every layer
PATH = os.environ.get("YAMD_TUNING_TABLE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json")
_table: Optional[Dict[str, List[int]]] = None
enabled = os.environ.get("YAMD_NO_TUNING", "") == ""


def key(k: int, s: int, cin: int, cout: int, ho: int, wo: int) -> str:
    return f"k{k}s{s}_{cin}>{cout}@{ho}x{wo}"


def table() -> Dict[str, List[int]]:
    global _table
    if _table is None:
        _table = {}
        if os.path.exists(PATH):
            with open(PATH) as f:
                _table = json.load(f)
    return _table


def lookup(k: int, s: int, cin: int, cout: int, ho: int, wo: int) -> Optional[Tuple[int, ...]]:
    if not enabled:
        return None
    t = table().get(key(k, s, cin, cout, ho, wo))
    if t is None:
        return None
    return (int(t[0]), int(t[1]), int(t[2]), int(t[3]) if len(t) > 3 else 4, int(t[4]) if len(t) > 4 else 1)


def candidates(k: int, s: int, cin: int, cout: int) -> List[Tuple[int, ...]]:
    groups = (cout + 31) // 32
    out = []
    for mrep in (1, 2):
        for nrep in (1, 2, 4):
            if nrep > 1 and nrep // 2 >= groups:       # more than one empty 32-channel group
                continue
            for ck in (16, 32, 64):
                if cin % ck == 0:
                    for nwaves in (4, 8):
                        for stages in (1, 2, 3):
                            if stages == 3 and groups <= nrep:       # one cout tile: the same launch as stages 2
                                continue
                            out.append((mrep, nrep, ck, nwaves, stages))
    return out


def save(entries: Dict[str, Tuple[int, ...]], path: str = PATH) -> None:
    t = dict(table())
    t.update({k_: list(v) for k_, v in entries.items()})
    with open(path, "w") as f:
        json.dump(dict(sorted(t.items())), f, indent=0)


def main() -> None:
    import argparse
    import torch
    from . import models
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="v5s")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--out", default=PATH)
    ap.add_argument("--reps", type=int, default=10, help="timed launches per candidate")
    ap.add_argument("--in-situ", action="store_true", help="time candidates inside whole-program runs (realistic cache state)")
    ap.add_argument("--missing-only", action="store_true", help="measure only the layer shapes the table does not hold yet")
    ap.add_argument("--only", default="", help="measure only the layer shapes whose key starts with this (e.g. k3s2)")
    args = ap.parse_args()
    global enabled
    if args.missing_only:
        from . import engine
        engine.FUSE["upsample_cat"] = False            # their readers are measured as plain convs of the same shape
    elif not args.in_situ:
        enabled = False                                # start every layer from the default rule
    p, m = models.create(args.model)
    dev = torch.device("cuda", 0)
    prog = m.program()
    imgs = torch.rand(args.batch, 3, p.img_size, p.img_size, device=dev)
    prog.run(imgs)                                     # fill the buffers with real activations
    if args.in_situ:
        best = prog.autotune_in_situ(imgs, reps=args.reps, log=print, only=args.only)
    else:
        skip = set(table()) if args.missing_only else set()
        if args.only:
            for op in prog.ops:
                if op[0] == "conv" and not op[1].desc.out_f32:
                    k_ = key(op[1]._args[0], op[1]._args[1], op[2].c, op[3].c + (op[5].c if op[5] is not None else 0), op[3].h, op[3].w)
                    if not k_.startswith(args.only):
                        skip.add(k_)
        best = prog.autotune(args.batch, dev, reps=args.reps, log=print, skip=skip)
    save(best, args.out)
    print(f"{len(best)} layer shapes -> {args.out}")


if __name__ == "__main__":
    main()
