import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_ortho9d" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "k_heads_l23" in r["Kernel_Name"]]
k = len(ends) // 2
t0 = int(rows[ends[k]]["End_Timestamp"])
step = rows[ends[k] + 1:ends[k + 1] + 1]
span = int(step[-1]["End_Timestamp"]) - t0
# union of busy intervals, and time-weighted concurrency
iv = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0) for r in step)
busy = 0; cur_s, cur_e = iv[0]
gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("step span %.3f ms, some kernel running %.3f ms, idle %.3f ms (first kernel starts at %.1f us)" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, iv[0][0] / 1e3))
gaps.sort(key=lambda g: g[0] - g[1])
print("largest idle gaps (us, at us):", [(round((b - a) / 1e3, 1), round(a / 1e3)) for a, b in gaps[:12]])
tot = sum(e - s for s, e in iv)
print("sum of kernel durations %.3f ms" % (tot / 1e6))
# phases
def first(name): 
    for r in step:
        if name in r["Kernel_Name"]: return (int(r["Start_Timestamp"]) - t0) / 1e3
def last(name):
    v = None
    for r in step:
        if name in r["Kernel_Name"]: v = (int(r["End_Timestamp"]) - t0) / 1e3
    return v
for nm in ("k_mask_chain64", "k_enumerate_sets", "k_sparse_conv_stem", "k_sparse_conv_dma", "k_three_nn_grid_levels", "k_three_interpolate_levels", "Cijk", "k_cross_attn_dma", "k_ortho9d"):
    print("%-28s first start %8.1f  last end %8.1f" % (nm, first(nm) or -1, last(nm) or -1))
