"""attribute the HBM traffic of two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md) over
scripts/traffic_layers.py to (layer, direction): the script launches K times per plan entry in a fixed order, so the dispatches of
the main GEMM kernels, taken in dispatch order, are chunks of K per entry; helper kernels (split-K finalize, partial-row and slab
reductions) between two main dispatches are added to the entry before them.  gfx950 corrections from the guide: FETCH_SIZE (KiB)
counts 128-B requests at 64 B -> bytes = 2048 v; WRITE_SIZE (KiB) is exact for 16-byte streaming stores -> bytes = 1024 v.
Optional third pass (--pmc TCC_HIT_sum TCC_MISS_sum): the L2 hit rate of every entry, TCC_HIT / (TCC_HIT + TCC_MISS) over its launches.
usage: python scripts/collect_traffic_layers.py <fetch_dir> <write_dir> <plan.json> <out.json> [<l2_dir>]"""
import csv, glob, json, re, sys

MAIN = re.compile(r"tapgemm_kernel|halo_convT_kernel|halo_conv_kernel|wgrad256[pq]?_kernel|wgrad2x_kernel|wgrad_kernel")
HELP = re.compile(r"tapgemm_finalize_kernel|dbpart_reduce_kernel|wgrad_reduce_kernel")

def per_entry(d, counter, K, n):
    rows = [r for r in csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    tot = [0.0] * n
    idx = -1
    for r in rows:
        name = r["Kernel_Name"]
        if HELP.search(name):
            if idx >= 0:
                tot[idx // K] += float(r["Counter_Value"])
        elif MAIN.search(name):
            idx += 1
            if idx // K < n:
                tot[idx // K] += float(r["Counter_Value"])
    assert idx + 1 == n * K, (idx + 1, n * K)
    return [t / K for t in tot]

plan = json.load(open(sys.argv[3]))
K, entries = plan["K"], plan["plan"]
fetch = per_entry(sys.argv[1], "FETCH_SIZE", K, len(entries))
write = per_entry(sys.argv[2], "WRITE_SIZE", K, len(entries))
out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of scripts/traffic_layers.py (config-3 shapes, bf16, "
                 "standalone launches, dense views); FETCH_SIZE KiB x 2048, WRITE_SIZE KiB x 1024; helper kernels (split-K finalize, "
                 "row / slab reductions) counted with their GEMM; alg = every operand touched once", "layers": []}
hit = miss = None
if len(sys.argv) > 5:
    try:
        hit, miss = per_entry(sys.argv[5], "TCC_HIT_sum", K, len(entries)), per_entry(sys.argv[5], "TCC_MISS_sum", K, len(entries))
        out["method"] += "; l2_hit = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) of a third pass"
    except Exception as exc:                                  # counters not collected on this box: the traffic table still stands
        print("no L2 hit rates:", exc)
for i, (e, f, w) in enumerate(zip(entries, fetch, write)):
    rd, wr = f * 2048, w * 1024
    row = dict(layer=e["layer"], dir=e["dir"], form=e["form"], alg_MB=round(e["alg_bytes"] / 1e6, 1), read_MB=round(rd / 1e6, 1),
               write_MB=round(wr / 1e6, 1), ratio=round((rd + wr) / e["alg_bytes"], 2), gflop=round(e["flops"] / 1e9, 1))
    if hit is not None and hit[i] + miss[i] > 0:
        row["l2_hit"] = round(hit[i] / (hit[i] + miss[i]), 3)
    out["layers"].append(row)
json.dump(out, open(sys.argv[4], "w"), indent=1)
for l in out["layers"]:
    print("%-3s %-6s %-6s alg %7.1f MB  read %7.1f  write %7.1f  x%.2f%s" % (l["layer"], l["dir"], l["form"], l["alg_MB"], l["read_MB"], l["write_MB"],
                                                                             l["ratio"], ("  L2 hit %.3f" % l["l2_hit"]) if "l2_hit" in l else ""))
