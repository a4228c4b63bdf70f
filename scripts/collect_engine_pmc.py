"""attribute `rocprofv3 --pmc` passes over `scripts/engine_layers.py --pmc K` (the ENGINE'S OWN launches of one config-3 train step:
its strides, its ReLU bit planes, its fused bias sums, its automatic dispatch - each call replayed K times in step order) to
(layer, direction).  The script runs 3 whole steps first and then the replays, so the LAST len(plan) * K main-kernel dispatches are
the replays, K per plan entry; helper kernels (split-K finalize, row / slab reductions, derived bit planes, the head's finish) between
two main dispatches are counted with the entry in front of them.

    traffic :  python scripts/collect_engine_pmc.py traffic <plan.json> <out.json> <fetch_dir> <write_dir> [<l2_dir>]
               gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KiB) counts 128-B requests at 64 B -> bytes = 2048 v;
               WRITE_SIZE (KiB) is exact for 16-byte streaming stores -> bytes = 1024 v; Infinity-Cache hits are included in both.
    mfma    :  python scripts/collect_engine_pmc.py mfma <plan.json> <out.txt> <dir with SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ...>
               MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 (the
               counter is summed over the 8 XCDs); cross-check: busy cycles / (16 cycles x MFMA instructions the launch must issue).
"""
import csv, glob, json, re, sys

MAIN = re.compile(r"tapgemm_kernel|halo_convT_kernel|wgrad256q_kernel|wgrad_kernel|rgb_fwd_kernel|rgb_wgrad_kernel")
HELP = re.compile(r"tapgemm_finalize_kernel|dbpart_reduce_kernel|wgrad_reduce|relu_bits_kernel|dense_head_finish_kernel")


def rows_of(d):
    rows = list(csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def per_entry(d, counter, K, n, main_only=False):
    """mean per launch of `counter` for each of the n plan entries; also the kernel name of each entry"""
    rows = [r for r in rows_of(d) if r["Counter_Name"] == counter]
    assert rows, (d, counter)
    idx_of = []                                    # plan slot of every row, or -1
    idx = -1
    for r in rows:
        name = r["Kernel_Name"]
        if MAIN.search(name) and not HELP.search(name):
            idx += 1
            idx_of.append(("main", idx))
        elif HELP.search(name):
            idx_of.append(("help", idx))
        else:
            idx_of.append(("other", -1))
    total_main = idx + 1
    first = total_main - n * K                     # the replays are the last n * K main dispatches
    assert first >= 0, (total_main, n, K)
    tot, names = [0.0] * n, [""] * n
    for r, (kind, i) in zip(rows, idx_of):
        if kind == "other" or i < first:
            continue
        if kind == "help" and main_only:
            continue
        slot = (i - first) // K
        tot[slot] += float(r["Counter_Value"])
        if kind == "main":
            names[slot] = re.sub(r"^void \(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
    return [t / K for t in tot], names


mode, plan_path, out_path = sys.argv[1:4]
plan = json.load(open(plan_path))
K, entries = plan["K"], plan["plan"]
n = len(entries)
if mode == "traffic":
    fetch, names = per_entry(sys.argv[4], "FETCH_SIZE", K, n)
    write, _ = per_entry(sys.argv[5], "WRITE_SIZE", K, n)
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (/ --pmc TCC_HIT_sum TCC_MISS_sum) in SEPARATE runs of "
                     "scripts/engine_layers.py --pmc %d: the engine's own launches of one config-3 train step (3x128x128, batch 64, bf16; its "
                     "strides incl. the dense dR_0, ReLU bit planes on, fused bias sums, automatic dispatch), each replayed %d times; FETCH_SIZE "
                     "KiB x 2048, WRITE_SIZE KiB x 1024 (gfx950 corrections, MI355X_MICROARCH.md); helper kernels (split-K finalize, row / slab "
                     "reductions, head finish) counted with their GEMM; alg = every operand touched once (masks as bits)" % (K, K),
           "layers": []}
    hit = miss = None
    if len(sys.argv) > 6:
        try:
            hit, _ = per_entry(sys.argv[6], "TCC_HIT_sum", K, n)
            miss, _ = per_entry(sys.argv[6], "TCC_MISS_sum", K, n)
        except Exception as exc:
            print("no L2 hit rates:", exc)
    for i, (e, f, w) in enumerate(zip(entries, fetch, write)):
        rd, wr = f * 2048, w * 1024
        row = dict(layer=e["layer"], dir=e["dir"], form=e["form"], kernel=names[i], alg_MB=round(e["alg_bytes"] / 1e6, 1), read_MB=round(rd / 1e6, 1),
                   write_MB=round(wr / 1e6, 1), ratio=round((rd + wr) / e["alg_bytes"], 2), gflop=round(e["flops"] / 1e9, 1), plane=e.get("plane", False))
        if hit is not None and hit[i] + miss[i] > 0:
            row["l2_hit"] = round(hit[i] / (hit[i] + miss[i]), 3)
        out["layers"].append(row)
    fams = {}
    for l in out["layers"]:
        f = fams.setdefault(l["form"], dict(launches=0, alg_MB=0.0, read_MB=0.0, write_MB=0.0))
        f["launches"] += 1
        for k in ("alg_MB", "read_MB", "write_MB"):
            f[k] = round(f[k] + l[k], 1)
    for f in fams.values():
        f["ratio"] = round((f["read_MB"] + f["write_MB"]) / f["alg_MB"], 2)
        f["per_launch_MB"] = round((f["read_MB"] + f["write_MB"]) / f["launches"], 1)
    out["families"] = fams
    json.dump(out, open(out_path, "w"), indent=1)
    for l in out["layers"]:
        print("%-3s %-9s %-6s alg %7.1f MB  read %7.1f  write %7.1f  x%.2f%s" % (l["layer"], l["dir"], l["form"], l["alg_MB"], l["read_MB"], l["write_MB"],
                                                                              l["ratio"], ("  L2 hit %.3f" % l["l2_hit"]) if "l2_hit" in l else ""))
    print(json.dumps(fams))
else:
    d = sys.argv[4]
    busy, names = per_entry(d, "SQ_VALU_MFMA_BUSY_CYCLES", K, n, main_only=True)
    gui, _ = per_entry(d, "GRBM_GUI_ACTIVE", K, n, main_only=True)
    try:
        sqb, _ = per_entry(d, "SQ_BUSY_CYCLES", K, n, main_only=True)
    except AssertionError:
        sqb = [0.0] * n
    with open(out_path, "w") as f:
        f.write("# MFMA utilisation of the engine's own launches (config 3, bf16; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over\n"
                "# scripts/engine_layers.py --pmc %d; main kernel of every call only).  util = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8);\n"
                "# issue = busy cycles / (16 x MFMA 16x16x32 instructions of the launch) - 1.0 means the counter sees exactly the multiplies the layer needs;\n"
                "# clock = GRBM_GUI_ACTIVE / 8 / duration is NOT given: it reads high on dispatches this short (guide, DVFS give-back) - see r04_kernel_clock.txt\n" % K)
        f.write("%-4s %-10s %-64s %14s %14s %14s %7s %6s\n" % ("lay", "dir", "kernel", "MFMA_BUSY", "SQ_BUSY", "GUI_ACTIVE/8", "util", "issue"))
        for e, b, g8, sb, nm in zip(entries, busy, gui, sqb, names):
            cyc = g8 / 8.0
            n_mfma = e["flops"] / (2 * 16 * 16 * 32)
            line = "%-4s %-10s %-64s %14.4g %14.4g %14.4g %7.3f %6.3f" % (e["layer"], e["dir"], nm[:64], b, sb, cyc, b / (1024 * cyc) if cyc else 0.0,
                                                                       b / (16 * n_mfma) if n_mfma else 0.0)
            f.write(line + "\n")
            print(line)
