#!/usr/bin/env python3
"""Feasibility probe: CU-masked HIP streams (hipExtStreamCreateWithCUMask) as a spatial partition between a saturating
GEMM stream and a chain of small latency-bound kernels.  Prints the chain's duration alone, beside the GEMMs on plain
streams, and beside the GEMMs with complementary CU masks (and what the masks cost the GEMMs)."""
import ctypes as C, sys, time
import torch
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
every = int(sys.argv[1]) if len(sys.argv) > 1 else 16           # 1 CU in `every` goes to the side partition
side_bits = sum(1 << i for i in range(256) if i % every == 0)
main_bits = ((1 << 256) - 1) ^ side_bits
torch.zeros(1, device="cuda")
A = torch.randn(32768, 1024, device="cuda"); B = torch.randn(1024, 1024, device="cuda")
x = torch.randn(1 << 16, device="cuda")
def gemms(n=20):
    for _ in range(n): torch.mm(A, B)
def chain(n=200):
    y = x
    for _ in range(n): y = y * 1.0001 + 0.5
    return y
def run(sm, ss, with_gemm=True):
    torch.cuda.synchronize()
    e0, e1, g0, g1 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    if with_gemm:
        with torch.cuda.stream(sm):
            g0.record(sm); gemms(); g1.record(sm)
    with torch.cuda.stream(ss):
        e0.record(ss); chain(); e1.record(ss)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), (g0.elapsed_time(g1) if with_gemm else float("nan"))
plain_m, plain_s = torch.cuda.Stream(), torch.cuda.Stream()
mask_m, mask_s = masked_stream(main_bits), masked_stream(side_bits)
for name, sm, ss, wg in (("chain alone, plain stream", plain_m, plain_s, False), ("chain alone, side mask", mask_m, mask_s, False),
                         ("plain streams", plain_m, plain_s, True), ("complementary CU masks", mask_m, mask_s, True)):
    run(sm, ss, wg)
    c, g = run(sm, ss, wg)
    print("%-28s chain of 200 small kernels %.3f ms, 20 GEMMs %.3f ms" % (name, c, g), flush=True)
