#!/usr/bin/env python3
"""How many of the 27 kernel-offset steps does a 128-row tile of a conv layer really use under different row orders?
(natural order; the product's 9-bit plane key sorted inside 8192-row windows; finer keys) -- the chunk units an ordered
launch issues are proportional to it.  usage: order_keys.py [b]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dcl = importlib.import_module("dcl-net_amd")
ops, sp = dcl.ops, dcl.spconv.ops
b, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 64
data = dcl.synth.make_batch(b, 1024, 64)
aset = ops.grid_from_indices(data["inp"]["occupied_voxels"].int().cuda().contiguous(), b, S)

def used_steps(valid, perm, tile=128):
    n = valid.shape[1]
    v = valid[:, perm]
    nt = (n + tile - 1) // tile
    pad = nt * tile - n
    if pad:
        v = torch.cat([v, torch.zeros(27, pad, dtype=torch.bool, device=v.device)], 1)
    return float(v.view(27, nt, tile).any(2).sum(0).float().mean())

def windowed_sort(key, win=8192):
    n = key.shape[0]
    return torch.cat([w0 + torch.argsort(key[w0:w0 + win], stable=True) for w0 in range(0, n, win)])

for lvl in range(4):
    out, nbr1 = sp.build_rulebook(aset, 3, 1, 1, False)
    _, nbr2 = sp.build_rulebook(out, 3, 1, 1, True)
    pool, _ = sp.build_rulebook(out, 3, 2, 1, False)
    for name, nbr in (("conv", nbr1), ("subm", nbr2)):
        valid = nbr[:, :out.n] >= 0
        n = out.n
        ident = torch.arange(n, device="cuda")
        v3 = valid.view(3, 3, 3, n)
        plane = torch.zeros(n, dtype=torch.int64, device="cuda")
        bit = 0
        for ax in range(3):
            for q in range(3):
                plane |= v3.select(ax, q).reshape(9, n).any(0).long() << bit
                bit += 1
        full = (valid.long() << torch.arange(27, device="cuda").view(27, 1)).sum(0)
        pc = valid.sum(0).long()
        # edge key: the 12 axis-pair "lines" (any neighbour in a row of 3 along z for every (x,y) -> 9 bits) -- a finer 9-bit key
        rowz = v3.any(2).reshape(9, n)                      # (kx,ky) columns: any kz present
        linekey = (rowz.long() << torch.arange(9, device="cuda").view(9, 1)).sum(0)
        res = {"natural": used_steps(valid, ident), "plane key / 8192 windows (product)": used_steps(valid, windowed_sort(plane)),
               "plane key, global": used_steps(valid, torch.argsort(plane, stable=True)),
               "plane + (x,y)-column key (18 bits) / 8192": used_steps(valid, windowed_sort(plane * 512 + linekey)),
               "full 27-bit mask / 8192": used_steps(valid, windowed_sort(full)),
               "full 27-bit mask, global": used_steps(valid, torch.argsort(full, stable=True)),
               "per-row mean (lower bound of any order)": float(pc.float().mean())}
        print("L%d %s rows %d:" % (lvl, name, n))
        for k, v in res.items():
            print("    %-46s %5.2f of 27" % (k, v))
    aset = pool
