"""Register / LDS / scratch use of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), demangled.
usage: python tools/kernel_resources.py dcl-net_amd/csrc/sparse_conv.hip [extra hipcc flags]"""
import re
import subprocess
import sys

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Rpass-analysis=kernel-resource-usage", "-c"]


def main():
    src, extra = sys.argv[1], sys.argv[2:]
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [src, "-o", "/dev/null"], capture_output=True, text=True).stderr
    name, d = None, {}
    for l in out.splitlines():
        m = re.search(r" Name: (\S+)", l)
        if m:
            name = m.group(1)
            d[name] = {}
            continue
        m = re.search(r"remark: +(TotalSGPRs|VGPRs|AGPRs|ScratchSize|Occupancy|LDS Size|VGPRs Spill|SGPRs Spill)[^:]*: (\d+)", l)
        if m and name:
            d[name].setdefault(m.group(1), int(m.group(2)))
        elif "error" in l:
            print(l)
    for k, v in d.items():
        dn = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(anonymous namespace\)::", "", dn)
        dn = re.sub(r"\(.*", "", dn)
        print("%-52s sgpr %3d vgpr %3d agpr %3d scratch %4d occ %d lds %6d" % (
            dn[:52], v.get("TotalSGPRs", -1), v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1),
            v.get("Occupancy", -1), v.get("LDS Size", -1)))


if __name__ == "__main__":
    main()
